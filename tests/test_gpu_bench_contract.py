"""bench.py on the GPU box: the one JSON line the driver parses (contract of the task statement + tier ④), on a reduced
workload so the test takes seconds.  The headline workload itself is what `python bench.py` runs; here we check the
shape of the line, that the timed kernel is the HIP one, and the N = 2 code path (gloo rehearsal on one GPU: same
sharding / gather / re-assembly code, RCCL replaced by a host-staged gather) with --verify."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, launcher=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    cmd = [sys.executable] + (launcher or []) + [os.path.join(ROOT, "bench.py")] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # exactly ONE JSON line on stdout
    return json.loads(lines[0])


def check_common(d, n, steps, warmup, scaling="weak"):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == n and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == scaling and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"):
        assert k in r, k
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["value"] > 0 and d["ms_per_step"] > 0


def test_default_workload_line_reduced_spp():
    d = run_bench(["--steps", "6", "--warmup", "2", "--spp", "20"])
    check_common(d, 1, 6, 2)
    assert d["metric"] == "path-traced samples/s" and d["unit"] == "samples/s"
    # value = units / wall: 900*600*20 samples per step
    assert abs(d["value"] - 900 * 600 * 20 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # kernel time from HIP events on the launch stream is a large part of the step (nothing else is in the timed region; the lower
    # bound is loose: a 1 ms kernel on a shared host — one slow launch of six used to fail 0.5)
    assert 0.2 * d["ms_per_step"] < d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.05
    assert d["roofline"]["traffic"] is None            # not the profiled default configuration
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "samples/s" and cb["sample"]
    assert d["value"] > 20 * cb["value"]                # a GPU, not a fallback, produced the headline
    assert d["secondary"]["metric"] == "Mandelbrot pixel-iters/s" and d["secondary"]["pixel_iters"] == 2158756620
    assert "strict_math" not in d                       # reported for the un-overridden headline workload only


@pytest.mark.parametrize("workload,unit", [("mandelbrot", "pixel-iters/s"), ("mandelbrot_ds", "pixel-iters/s")])
def test_other_workloads(workload, unit):
    d = run_bench(["--steps", "2", "--warmup", "1", "--workload", workload, "--no-cpu-baseline"])
    check_common(d, 1, 2, 1)
    assert d["unit"] == unit
    r = d["roofline"]
    assert r["hbm"]["algorithmic_bytes"] == 3200 * 2400 * 20
    # Profiled configurations: `traffic` / `executed` come from the PMC summary committed under profiles/ — but only from one taken on
    # THIS build (tools/summarize_prof.py stamps mc_build_id(); a figure measured on other code is not quoted).
    assert r["build_id"].startswith("pt=") and " mandel=" in r["build_id"]
    if r["traffic"] is not None:
        assert abs(r["traffic"] - r["hbm"]["algorithmic_bytes"]) / r["traffic"] < 0.02
        assert r["executed_frac"] is not None and 0.05 < r["executed_frac"] < 1.0 and "profiles/r0" in r["executed_source"]
    else:
        assert r["executed"] is None and r["kernel_clock_ghz"] is None
        assert r["executed_source"] is None or "not quoted" in r["executed_source"]


def test_two_rank_rehearsal_is_bit_identical_to_one_gpu():
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29533"]
    d = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--spp", "16", "--verify"],
                  env_extra={"MC_BENCH_BACKEND": "gloo"}, launcher=launcher)
    check_common(d, 2, 1, 1)
    assert d["config"]["image"] == [900, 1200] and d["config"]["rows_per_gpu"] == 600
    assert d["config"]["verified_equal_to_single_gpu"] is True
    assert "cpu_baseline" not in d                      # rank 0 at N = 1 only


def test_plain_command_with_two_gpus_starts_its_own_ranks():
    """the driver's N = 1 command shape with --gpus 2 — no launcher around it — must not die on the launcher:
    bench.py starts `torch.distributed.run` as a child process before touching the GPU and relays rank 0's one line (gloo rehearsal
    on the one GPU; every rank saw world size 2; --verify: the gathered image equals the single-GPU render)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--spp", "16", "--verify"],
                       cwd=ROOT, env=dict(env, MC_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]
    d = json.loads(lines[0])
    check_common(d, 2, 1, 1)
    c = d["config"]
    assert c["world_size"] == 2 and [r["world_size_seen"] for r in c["ranks"]] == [2, 2] and c["verified_equal_to_single_gpu"] is True
    # under a profiler's preload it refuses, with the launcher line as the hint, before any child is started
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--spp", "16"],
                       cwd=ROOT, env=dict(env, ROCPROFILER_TEST_MARK="1"), capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "torch.distributed.run" in p.stderr and not p.stdout.strip()


def test_baseline_8gpu_configs_run_from_the_driver_command():
    """BASELINE's 8-GPU configurations are reachable as `bench.py --config K3|K4` (strong scaling: the whole image on N
    ranks).  K4 at full size on one GPU; K3 at reduced spp (its 4096-spp step is 4e10 samples, ~4 s)."""
    d = run_bench(["--config", "K4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    check_common(d, 1, 2, 1, scaling="strong")
    assert d["config"]["baseline_config"] == "K4" and d["config"]["image"] == [7680, 5120]
    assert d["metric"] == "Mandelbrot pixel-iters/s"
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 41176259776) < 1.0          # the frozen checksum of executed loop bodies
    d = run_bench(["--config", "K3", "--steps", "1", "--warmup", "1", "--spp", "8"])
    check_common(d, 1, 1, 1, scaling="strong")
    assert d["config"]["baseline_config"] == "K3" and d["config"]["image"] == [3840, 2560]
    assert abs(d["value"] - 3840 * 2560 * 8 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "one sample" in d["cpu_baseline"]["sample"] or "samples 0.." in d["cpu_baseline"]["sample"]


def test_four_rank_rehearsal_strong_scaling_with_per_rank_evidence():
    """K3 geometry cut down (384 x 250: 32 row blocks, the last one partial -> ranks own 64/64/64/58 rows) over 4 ranks
    (gloo rehearsal on the one GPU): per-rank rows / devices / kernel times arrive on rank 0, every rank saw world size 4,
    and the gathered image is bit-identical to the single-GPU render."""
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                "--master-port", "29541"]
    d = run_bench(["--gpus", "4", "--steps", "1", "--warmup", "1", "--config", "K3", "--spp", "6", "--width", "384", "--height", "250",
                   "--verify"], env_extra={"MC_BENCH_BACKEND": "gloo"}, launcher=launcher)
    check_common(d, 4, 1, 1, scaling="strong")
    c = d["config"]
    assert c["image"] == [384, 250] and c["world_size"] == 4 and c["verified_equal_to_single_gpu"] is True
    assert [r["rank"] for r in c["ranks"]] == [0, 1, 2, 3] and all(r["world_size_seen"] == 4 for r in c["ranks"])
    assert [r["rows"] for r in c["ranks"]] == [64, 64, 64, 58] and sum(r["units_per_step"] for r in c["ranks"]) == 384 * 250 * 6
    assert all(r["kernel_ms"] > 0 for r in c["ranks"]) and c["gather_ms_rank0"] > 0


def test_two_rank_rehearsal_mandelbrot_exchanges_iteration_counts():
    """K4 geometry cut down (768 x 520, M = 50 000) over 2 ranks (gloo rehearsal): the ranks send 16-bit iteration counts — 2 B
    per pixel, an eighth of the vec4 tile — rank 0 rebuilds the storage buffer through the colour table, and the result is
    bit-identical to the single-GPU render."""
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29551"]
    d = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "K4", "--width", "768", "--height", "520", "--verify"],
                  env_extra={"MC_BENCH_BACKEND": "gloo"}, launcher=launcher)
    check_common(d, 2, 2, 1, scaling="strong")
    c = d["config"]
    assert c["verified_equal_to_single_gpu"] is True and "uint16" in c["exchange"]
    assert c["gather_bytes_per_rank"] == 264 * 768 * 2      # rank 0's tile: 33 of the 65 row blocks
    assert sum(r["units_per_step"] for r in c["ranks"]) == round(d["value"] * d["ms_per_step"] * 1e-3)


MULTI_KEYS = ("workload", "metric", "unit", "scaling", "value", "ms_per_step", "image", "n_gpus", "world_size", "ranks", "exchange",
              "exchange_async", "gather_ms_rank0", "gather_bytes_per_rank", "single_gpu", "retained_per_gpu", "roofline",
              "equal_to_single_gpu", "checks")


@pytest.mark.parametrize("n", [2, 4])
def test_plain_command_carries_the_8gpu_configurations(n):
    """the driver's ONE command — `python bench.py --gpus N`, nothing else — must measure BASELINE's two 8-GPU
    configurations too.  After the (weak-scaled) K2 headline the line carries a `multi` block: K3 (3840 x 2560 x 4096 spp, RGBA8
    exchange: 4 B/pixel) and K4 (7680 x 5120, M = 50 000, two-float, 16-bit counts) at FULL size, strong-scaled over the N ranks, each
    with per-rank evidence, a bit-equality check against rank 0's own single-GPU render in the same job (K4 also against the frozen
    checksum 41 176 259 776) and retained_per_gpu.  gloo rehearsal on the one GPU: the ranks share it, so the TIMINGS mean nothing
    (the block says so) — the keys, the sharding, the exchange formats and the equalities are what is asserted."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], cwd=ROOT, env=dict(env, MC_BENCH_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]
    d = json.loads(lines[0])
    check_common(d, n, 10, 2)
    assert d["config"]["baseline_config"] == "K2" and d["config"]["image"] == [900, 600 * n]      # nothing about the headline changes
    assert d["config"]["gather_bytes_per_rank"] == 600 * 900 * 16 and d["config"]["exchange_fell_back"] is False
    mb = d["multi"]
    assert mb["failed_equality"] == [] and "REHEARSAL" in mb["note"]
    for name, image in (("K3", [3840, 2560]), ("K4", [7680, 5120])):
        e = mb[name]
        assert not [k for k in MULTI_KEYS if k not in e], [k for k in MULTI_KEYS if k not in e]
        assert e["image"] == image and e["n_gpus"] == n and e["world_size"] == n and e["scaling"] == "strong"
        assert [r["rank"] for r in e["ranks"]] == list(range(n)) and all(r["world_size_seen"] == n for r in e["ranks"])
        assert sum(r["rows"] for r in e["ranks"]) == image[1] and all(r["kernel_ms"] > 0 for r in e["ranks"])
        assert e["equal_to_single_gpu"] is True and all(v for v in e["checks"].values() if isinstance(v, bool))
        assert e["exchange_async"] is False                                      # (the rehearsal backend stages through the host)
        assert e["single_gpu"]["value"] > 0 and abs(e["retained_per_gpu"] - e["value"] / n / e["single_gpu"]["value"]) < 1e-9
        rf = e["roofline"]      # north_star: the fraction of the fp32-ALU roofline at every N (whole job over N x the one-GPU peak)
        assert rf["peak"] == pytest.approx(157.3 * n) and rf["frac"] == pytest.approx(e["value"] * rf["flops_per_unit"] / 1e12 / (157.3 * n))
        assert rf["frac"] == pytest.approx(rf["single_gpu_frac"] * e["retained_per_gpu"])
    k3, k4 = mb["K3"], mb["K4"]
    assert k3["gather_bytes_per_rank"] == 4 * 3840 * (2560 // n) and "RGBA8" in k3["exchange"]           # 4 B/pixel
    assert set(k3["checks"]) == {"rgba8_image", "f32_storage_buffer"}
    assert abs(k3["value"] - 3840 * 2560 * 4096 / (k3["ms_per_step"] * 1e-3)) / k3["value"] < 1e-6
    assert k4["gather_bytes_per_rank"] == 2 * 7680 * (5120 // n) and "uint16" in k4["exchange"]
    assert k4["checks"]["pixel_iters"] == k4["checks"]["pixel_iters_frozen"] == 41176259776 and k4["checks"]["pixel_iters_match"] is True
    assert sum(r["units_per_step"] for r in k4["ranks"]) == 41176259776


def test_rgba8_exchange_of_the_headline_and_its_verify():
    """--exchange rgba8 on the K2 geometry (2 ranks, gloo rehearsal) with --verify: the RGBA8 image equals the single-GPU render's,
    and the fp32 tiles — gathered once more, untimed — re-assemble to the single-GPU storage buffer bit for bit."""
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29561"]
    d = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "16", "--exchange", "rgba8", "--verify"],
                  env_extra={"MC_BENCH_BACKEND": "gloo"}, launcher=launcher)
    check_common(d, 2, 2, 1)
    c = d["config"]
    assert c["verified_equal_to_single_gpu"] is True and "RGBA8" in c["exchange"] and c["gather_bytes_per_rank"] == 600 * 900 * 4
    assert "multi" not in d                                                      # overridden sizes: the headline only


def test_app_default_math_is_reported_beside_the_headline():
    """the headline is `--math fast`; what the C ABI's defaults, bin/pathtracer and route B render without being
    asked is strict.  The un-overridden N = 1 line says so in `app_default`, next to `strict_math`."""
    d = run_bench(["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end"])
    a = d["app_default"]
    assert a["math"] == "strict" and a["headline_math"] == "fast" and a["kernel_ms"] == d["strict_math"]["kernel_ms"]
    assert a["kernel_ms"] > 1.5 * a["headline_kernel_ms"] and 0.1 < a["frac"] < d["roofline"]["frac"]
    r = d["roofline"]
    assert "variant=shipped" in r["build_id"]


def test_a_failing_multi_block_keeps_the_headline_and_fails_the_job():
    """Round 6: whatever goes wrong inside the `multi` block on the first real multi-GPU run must not cost the headline that was already
    measured, and must not pass for success: the line is printed with the error in `multi`, the job exits 5, and the plain command
    relays both (injected failure; 2 ranks, gloo rehearsal)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT,
                       env=dict(env, MC_BENCH_BACKEND="gloo", MC_BENCH_INJECT_MULTI_FAILURE="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "the multi block failed" in p.stderr, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    check_common(d, 2, 2, 1)
    assert "injected failure" in d["multi"]["error"] and d["config"]["image"] == [900, 1200]
