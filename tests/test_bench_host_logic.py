"""bench.py's host logic, on the CPU: which committed profile a bench line may quote, what the plain command measures at N > 1,
and the command lines of the standalone apps.  No GPU call anywhere: the library is loaded only for mc_build_id()."""
import argparse
import copy
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(**kw):
    base = dict(width=None, height=None, spp=None, math="fast")
    base.update(kw)
    return argparse.Namespace(**base)


def _summary(build, library="vulkan-compute-tests_amd/lib/libmc_compute.so", launches=1):
    """A PMC summary shaped like tools/summarize_prof.py's output, stamped with `build`."""
    doc = {"_build": dict(build, library=library)}
    for i in range(launches):
        doc[f"pathtrace_pool_kernel<1, 16, 3, true>{'' if i == 0 else ' tail'}"] = {
            "per_launch_mean": {"SQ_INSTS_VALU_ADD_F32": 10.0, "SQ_INSTS_VALU_MUL_F32": 20.0, "SQ_INSTS_VALU_FMA_F32": 30.0,
                                "SQ_INSTS_VALU_TRANS_F32": 5.0},
            "derived": {"active_lanes_per_valu_inst": 40.0, "hbm_write_bytes": 8640000.0, "kernel_clock_ghz": 2.3}}
    return doc


@pytest.fixture()
def fake_profiles(B, tmp_path, monkeypatch):
    """bench.ROOT pointed at an empty tree with a profiles/ directory; returns (write(doc), this library's build id)."""
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "_STALE", {})
    monkeypatch.delenv("MC_LIB_PATH", raising=False)

    def write(doc, name="r06_pt_fast_pmc_summary.json"):
        (tmp_path / "profiles" / name).write_text(json.dumps(doc))

    return write, B.build_id()


def test_a_summary_of_this_build_is_quoted(fake_profiles):
    write, mine = fake_profiles
    write(_summary(mine))
    entries, fname = bench.profiled_summary("K2", _args(), 1)
    assert fname == "r06_pt_fast_pmc_summary.json" and len(entries) == 1 and "_build" not in entries
    traffic, note = bench.profiled_traffic("K2", _args(), 1)
    assert traffic == 8640000.0 and "1 launch" in note
    flops, src = bench.profiled_executed_lane_flops("K2", _args(), 1)
    assert flops == (10.0 + 20.0 + 2 * 30.0 + 5.0) * 40.0 and fname in src        # add + mul + 2 fma + transcendental, x active lanes
    assert bench._STALE == {}


def test_two_launches_of_a_step_are_summed_and_the_tail_reread_is_added(fake_profiles):
    write, mine = fake_profiles
    write(_summary(mine, launches=2))
    traffic, note = bench.profiled_traffic("K2", _args(), 1)
    assert traffic == 2 * 8640000.0 + 900 * 600 * 16 and "accumulator read" in note
    flops, _ = bench.profiled_executed_lane_flops("K2", _args(), 1)
    assert flops == 2 * (10.0 + 20.0 + 60.0 + 5.0) * 40.0


def test_a_summary_of_another_build_is_refused_with_the_reason(fake_profiles):
    """ADVICE r5 / DESIGN §5 "Evidence names its build": a figure measured on other code is not this run's."""
    write, mine = fake_profiles
    other = dict(mine, pt="0123456789abcdef")
    write(_summary(other))
    assert bench.profiled_summary("K2", _args(), 1) == (None, None)
    assert "pt=0123456789abcdef" in bench._STALE["K2"] and f"pt={mine['pt']}" in bench._STALE["K2"] and "not quoted" in bench._STALE["K2"]
    assert bench.profiled_traffic("K2", _args(), 1) == (None, None)
    assert bench.profiled_executed_lane_flops("K2", _args(), 1) == (None, None)
    # the Mandelbrot family's id does not rescue a path-tracer summary, and the other way round
    write(_summary(dict(mine, mandel="0123456789abcdef")))
    assert bench.profiled_summary("K2", _args(), 1)[1] == "r06_pt_fast_pmc_summary.json"


def test_an_unstamped_summary_is_refused(fake_profiles):
    write, _ = fake_profiles
    doc = _summary({})
    del doc["_build"]
    write(doc)
    assert bench.profiled_summary("K2", _args(), 1) == (None, None) and "unstamped" in bench._STALE["K2"]


def test_a_summary_of_a_diagnostic_library_is_never_this_builds(fake_profiles, monkeypatch):
    write, mine = fake_profiles
    write(_summary(mine, library="vulkan-compute-tests_amd/lib/libmc_compute_stats.so"))
    assert bench.profiled_summary("K2", _args(), 1) == (None, None) and "libmc_compute_stats.so" in bench._STALE["K2"]
    # ... and neither is any summary while a diagnostic library is the one loaded
    write(_summary(mine))
    monkeypatch.setenv("MC_LIB_PATH", "/somewhere/libmc_compute_exp_a.so")
    assert bench.profiled_summary("K2", _args(), 1) == (None, None) and "libmc_compute_exp_a.so" in bench._STALE["K2"]


def test_only_the_profiled_configuration_quotes_a_profile(fake_profiles):
    write, mine = fake_profiles
    write(_summary(mine))
    assert bench.profiled_summary("K2", _args(), 2) == (None, None)                   # N > 1
    assert bench.profiled_summary("K2", _args(spp=16), 1) == (None, None)             # overridden sizes
    assert bench.profiled_summary("K2", _args(height=64), 1) == (None, None)
    assert bench.profiled_summary("K2", _args(math="careful"), 1) == (None, None)     # no profile tag for the careful tier
    assert bench.profiled_summary("K2", _args(math="strict"), 1) == (None, None)      # (its own file: r06_pt_strict_…, absent here)
    assert bench._STALE == {}                                                         # none of these is a stale profile


def test_newest_round_first(fake_profiles):
    write, mine = fake_profiles
    write(_summary(mine), "r05_pt_fast_pmc_summary.json")
    assert bench.profiled_summary("K2", _args(), 1)[1] == "r05_pt_fast_pmc_summary.json"
    write(_summary(dict(mine, pt="feedfeedfeedfeed")), "r06_pt_fast_pmc_summary.json")
    # the newest round's file decides: a stale r06 summary is not papered over by an older one that happens to match
    assert bench.profiled_summary("K2", _args(), 1) == (None, None) and "r06_pt_fast" in bench._STALE["K2"]


def test_committed_summaries_are_either_of_this_build_or_refused(B, monkeypatch):
    """The real profiles/ directory: every configuration's newest summary either carries the loaded library's id for its kernel
    family (and is quoted) or is refused with a reason — never quoted across builds."""
    monkeypatch.setattr(bench, "_STALE", {})
    monkeypatch.delenv("MC_LIB_PATH", raising=False)
    mine = B.build_id()
    for cfg, math in (("K2", "fast"), ("K2", "strict"), ("K1", "fast"), ("K1ds", "fast"), ("K3", "fast"), ("K4", "fast")):
        entries, fname = bench.profiled_summary(cfg, _args(math=math), 1)
        family = "pt" if bench.CONFIGS[cfg]["kind"] == "pt" else "mandel"
        if entries:
            doc = json.load(open(os.path.join(ROOT, "profiles", fname)))
            assert doc["_build"][family] == mine[family] and doc["_build"]["library"].endswith("lib/libmc_compute.so")
        else:
            assert cfg in bench._STALE or fname is None


def _parse(monkeypatch, *argv):
    monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
    return bench.parse()


def test_what_the_plain_command_measures(monkeypatch):
    """N > 1 with nothing overridden adds BASELINE's two 8-GPU configurations; anything else is the headline only."""
    a = _parse(monkeypatch)
    assert (a.gpus, a.config, a.steps, a.warmup, a.math, a.multi) == (1, "K2", 10, 2, "fast", [])
    assert _parse(monkeypatch, "--gpus", "8").multi == ["K3", "K4"]
    assert _parse(monkeypatch, "--gpus", "8", "--steps", "5", "--warmup", "1").multi == ["K3", "K4"]     # the driver's flags
    for extra in (["--no-multi"], ["--config", "K2"], ["--config", "K3"], ["--workload", "pathtrace"], ["--spp", "16"],
                  ["--width", "64"], ["--height", "64"]):
        assert _parse(monkeypatch, "--gpus", "8", *extra).multi == [], extra
    a = _parse(monkeypatch, "--config", "K3")
    assert (a.steps, a.warmup) == (3, 1)                      # a K3 step is 4e10 samples
    assert _parse(monkeypatch, "--workload", "mandelbrot_ds").config == "K1ds"
    with pytest.raises(SystemExit):
        _parse(monkeypatch, "--config", "K1", "--workload", "pathtrace")
    with pytest.raises(SystemExit):
        _parse(monkeypatch, "--math", "carefull")


def test_app_command_lines_follow_the_reference_surface():
    """src/main.cpp:22-24: argv[1] = spp, argv[2] = resy, resx = resy * 3 / 2 — never reinterpreted; everything else is an option."""
    cmd = bench.app_command("K2", "host_buffer", "/tmp/x.png", "fast")
    assert os.path.basename(cmd[0]) == "pathtracer" and cmd[1:3] == ["500", "600"] and cmd[cmd.index("--math") + 1] == "fast"
    assert "--gpu-postprocess" not in cmd and cmd[-2:] == ["--out", "/tmp/x.png"]
    cmd = bench.app_command("K3", "rgba8", "/tmp/y.png", "strict", extra=("--full-teardown",))
    assert cmd[1:3] == ["4096", "2560"] and "--gpu-postprocess" in cmd and cmd[-1] == "--full-teardown"
    cmd = bench.app_command("K4", "rgba8", "/tmp/z.png")
    assert os.path.basename(cmd[0]) == "mandelbrot" and cmd[cmd.index("--precision") + 1] == "ds"
    assert [float(v) for v in cmd[cmd.index("--centre") + 1:cmd.index("--centre") + 3]] == list(bench.K4_VIEW["centre"])
    assert [float(v) for v in cmd[cmd.index("--scale") + 1:cmd.index("--scale") + 3]] == list(bench.K4_VIEW["scale"])
    cmd = bench.app_command("K1", "host_buffer", "/tmp/w.png")
    assert "--precision" not in cmd and cmd[cmd.index("--max-iter") + 1] == "1000"


def test_frozen_work_figures():
    """The figures `roofline.achieved` is built from: changing one changes every reported fraction."""
    assert bench.FLOPS_PER_PIXEL_ITER_F32 == 8 and bench.FLOPS_PER_PIXEL_ITER_DS == 3 * 32 + 4 * 11 + 2
    assert bench.FLOPS_PER_SAMPLE_PT == 1469 + 2072 + 146 + 90 + 32
    assert bench.PEAK_FP32_TFLOPS == pytest.approx(256 * 4 * 32 * 2 * 2.4e9 / 1e12, rel=1e-3)
    assert bench.PEAK_LANE_OPS == pytest.approx(bench.PEAK_FP32_TFLOPS * 1e12 / 2, rel=1e-3)
    assert bench.K4_PIXEL_ITERS == 41176259776
    k = copy.deepcopy(bench.CONFIGS)
    assert (k["K2"]["W"], k["K2"]["H"], k["K2"]["spp"]) == (900, 600, 500) and (k["K3"]["W"], k["K3"]["H"], k["K3"]["spp"]) == (3840, 2560, 4096)
    assert (k["K1"]["W"], k["K1"]["H"], k["K1"]["M"]) == (3200, 2400, 1000) and (k["K4"]["W"], k["K4"]["H"], k["K4"]["M"]) == (7680, 5120, 50000)
    assert k["K2"]["scaling"] == k["K1"]["scaling"] == "weak" and k["K3"]["scaling"] == k["K4"]["scaling"] == "strong"
