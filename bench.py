#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X (driver contract).

  python bench.py --gpus N --steps K --warmup W [--config K2|K1|K1ds|K3|K4]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   — one rank per GPU;
   the plain form `python bench.py --gpus N` starts exactly that launcher itself, as a CHILD process, and relays its one line)

A "step" is one pass of the hot path over one synthetic batch (BASELINE.json configs):
  K2 (default, the headline): path trace 900 x 600, 500 spp, default scene.      WEAK scaling: image W x (600 N).
  K1: Mandelbrot 3200 x 2400, M = 1000, fp32;  K1ds: its two-float variant.       WEAK scaling: image W x (2400 N).
  K3: path trace 3840 x 2560, 4096 spp — BASELINE's 8-GPU path-trace config.      STRONG scaling: the whole image on N ranks.
  K4: Mandelbrot deep zoom 7680 x 5120, M = 50 000, two-float — the 8-GPU one.    STRONG scaling.
For N > 1 every rank renders its interleaved row blocks (mc_row_block() = 8 rows; every pixel is keyed by its absolute
coordinates, so tiling never changes a pixel's arithmetic; samples are never split across GPUs — the fp32 accumulation
order is part of the parity contract), then the fp32 tiles are gathered to rank 0 over RCCL and re-assembled there.
`--verify` re-renders the whole image on rank 0 after the timed region and requires bit-identity with the gathered one.

Output buffers are resident in HBM (torch tensors); there are no inputs besides 432 B of scene constants.  The timed
region is bracketed by barrier + torch.cuda.synchronize() on both sides; value = whole-job units / max-over-ranks time.
One JSON line is printed by rank 0.

`roofline`: fp32 VALU (neither HBM nor MFMA bounds this path: 16 B written per pixel, no contraction).
   achieved = algorithmic fp32 flops per step (flops/unit from the oracle's op counters) / mean kernel time of a
   step, measured with HIP events on the launch stream; peak = 157.3 TFLOP/s (FMA-counted, MI355X_MICROARCH.md).
   `lane_ops` adds the issue-slot view (78.6e12 lane-ops/s at 2.4 GHz): parity forbids contraction, so every flop occupies
   a slot and 0.5 of the FMA-counted peak is the ceiling.  `traffic` = HBM bytes per step from the rocprofv3 PMC pass
   committed under profiles/ (summed over ALL launches of a step), with `traffic_source` naming the file.
`cpu_baseline`: the CPU oracle (a port, not the reference's Vulkan build — lavapipe/glslang are absent on the GPU box, which
   receives only this repository) timed on this host's cores over a bounded sample of the same workload.
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic work per unit (frozen from the oracle's per-op counters).
FLOPS_PER_PIXEL_ITER_F32 = 8        # mandelbrot.comp:43-44 in reuse-optimal form
FLOPS_PER_PIXEL_ITER_DS = 142       # 3 ds_mul(32) + 4 ds_add/sub(11) + 2
FLOPS_PER_SAMPLE_PT = 3809.0        # oracle counters, 900x600 default scene: add 1469 + mul 2072 + div 146 + sqrt 90 + trig 32
PEAK_FP32_TFLOPS = 157.3            # MI355X vector fp32, FMA counted as 2 (v_pk_fma_f32 only)
PEAK_LANE_OPS = 78.6e12             # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (v_add/v_mul issue rate)

K4_VIEW = dict(centre=(-0.7436438870371587, 0.13182590420531198), scale=(1e-8, 1e-8 * 2.0 / 3.0))
CONFIGS = {
    #        kind          W     H     spp / M      two-float  scaling
    "K2":   dict(kind="pt", W=900, H=600, spp=500, scaling="weak"),
    "K1":   dict(kind="mandel", W=3200, H=2400, M=1000, ds=False, scaling="weak"),
    "K1ds": dict(kind="mandel", W=3200, H=2400, M=1000, ds=True, scaling="weak"),
    "K3":   dict(kind="pt", W=3840, H=2560, spp=4096, scaling="strong"),
    "K4":   dict(kind="mandel", W=7680, H=5120, M=50000, ds=True, scaling="strong"),
}
WORKLOAD_ALIAS = {"pathtrace": "K2", "mandelbrot": "K1", "mandelbrot_ds": "K1ds"}
PROFILE_TAG = {"K2": {"fast": "pt_fast", "strict": "pt_strict"}, "K1": "mandel", "K1ds": "mandel_ds", "K4": "k4",
               "K3": {"fast": "k3", "strict": "k3_strict"}}


PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02", "r01d", "r01c")
_STALE = {}      # profile file -> why it is not quoted (build id of another library)


def profiled_summary(cfg_name, args, n):
    """The committed rocprofv3 PMC summary (profiles/<round>_<tag>_pmc_summary.json) of this exact configuration, newest round
    first; None when the run is not a profiled configuration (N > 1, overridden sizes) — or when the summary was taken on ANOTHER
    BUILD: tools/summarize_prof.py stamps each summary with mc_build_id() of the library it profiled (the source hash of the kernel
    family), and a figure measured on other code is not this run's (`roofline.executed_source` then says so instead of quoting it)."""
    if n != 1 or args.width or args.height or args.spp or cfg_name not in PROFILE_TAG:
        return None, None
    tag = PROFILE_TAG[cfg_name]
    if isinstance(tag, dict):
        tag = tag.get(args.math)
        if tag is None:
            return None, None
    family = "pt" if CONFIGS[cfg_name]["kind"] == "pt" else "mandel"
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{tag}_pmc_summary.json")
        if os.path.exists(path):
            try:
                doc = json.load(open(path))
            except (ValueError, OSError):
                return None, None
            import __graft_entry__ as entry
            mine = entry.load_package().bindings.build_id().get(family)
            theirs = (doc.get("_build") or {}).get(family)
            taken_on = (doc.get("_build") or {}).get("library", "vulkan-compute-tests_amd/lib/libmc_compute.so")
            shipped = os.path.normpath(taken_on).endswith(os.path.join("lib", "libmc_compute.so"))
            if not shipped or os.environ.get("MC_LIB_PATH"):
                # a summary of a diagnostic library (make stats / variants / exp), or a diagnostic library loaded now: never this build's
                _STALE[cfg_name] = (f"profiles/{os.path.basename(path)} was taken on {taken_on}" +
                                    (f", the loaded library is {os.environ['MC_LIB_PATH']}" if os.environ.get("MC_LIB_PATH") else "") + ": not quoted")
                return None, None
            if theirs != mine:
                _STALE[cfg_name] = (f"profiles/{os.path.basename(path)} was taken on build {family}={theirs or 'unstamped'}, the loaded library "
                                    f"is {family}={mine}: not quoted")
                return None, None
            entries = {k: v for k, v in doc.items() if "pathtrace" in k or "mandelbrot_kernel" in k}
            return (entries, os.path.basename(path)) if entries else (None, None)
    return None, None


def profiled_traffic(cfg_name, args, n):
    """HBM bytes per STEP from the rocprofv3 PMC pass committed under profiles/ (WRITE_SIZE in its own --pmc pass, exact for
    16-B-per-lane stores, MI355X_MICROARCH.md §HBM), summed over every launch of a step (a K2 step may be two launches: the
    S = 16 rounds and the 4-sample tail, which also re-reads the 8.64 MB accumulator it continues — added as algorithmic
    bytes, FETCH_SIZE being uncalibrated for lane-sparse loads).  (None, None) when the run is not a profiled configuration."""
    entries, fname = profiled_summary(cfg_name, args, n)
    if not entries:
        return None, None
    total, launches = 0.0, 0
    for entry in entries.values():
        b = entry.get("derived", {}).get("hbm_write_bytes")
        if b:
            total += b
            launches += 1
    if not launches:
        return None, None
    note = f"profiles/{fname}: WRITE_SIZE summed over the {launches} launch(es) of a step"
    if cfg_name == "K2" and launches == 2:
        total += 900 * 600 * 16
        note += " + the tail launch's 8.64 MB accumulator read (algorithmic)"
    return total, note


def profiled_executed_lane_flops(cfg_name, args, n):
    """EXECUTED fp32 lane-flops per step from the committed PMC instruction mix: (add + mul + 2 fma + transcendental
    wave-instructions) x the active lanes per VALU instruction, summed over the launches of a step.  This is what the ALUs
    did, as opposed to the algorithmic count of the reference's arithmetic that `roofline.achieved` is defined on."""
    entries, fname = profiled_summary(cfg_name, args, n)
    if not entries:
        return None, None
    total = 0.0
    for entry in entries.values():
        c, d = entry.get("per_launch_mean", {}), entry.get("derived", {})
        lanes = d.get("active_lanes_per_valu_inst")
        if lanes is None or "SQ_INSTS_VALU_ADD_F32" not in c:
            return None, None
        total += (c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + 2.0 * c["SQ_INSTS_VALU_FMA_F32"] +
                  c["SQ_INSTS_VALU_TRANS_F32"]) * lanes
    return total, f"profiles/{fname}: (ADD + MUL + 2 FMA + TRANS wave-instructions) x active lanes per VALU instruction"


def lavapipe_probe():
    """BASELINE.md §3 plan 1 asks for the reference's own Vulkan build on lavapipe as the CPU baseline when the tooling exists.
    It needs a Vulkan loader, a lavapipe ICD, a GLSL compiler AND the reference checkout; the GPU box receives only this
    repository, so the probe documents what is missing and the baseline stays the oracle port (kind "port")."""
    import ctypes.util
    import glob
    import shutil
    found = {
        "libvulkan": bool(ctypes.util.find_library("vulkan")),
        "lavapipe_icd": bool(glob.glob("/usr/share/vulkan/icd.d/lvp_icd*.json") + glob.glob("/etc/vulkan/icd.d/lvp_icd*.json")),
        "glsl_compiler": bool(shutil.which("glslangValidator") or shutil.which("glslc")),
        "reference_checkout": os.path.isdir(os.environ.get("MC_REFERENCE_DIR", "/root/reference")),
    }
    found["usable"] = all(found.values())
    return found


APP_DIR = os.path.join(ROOT, "vulkan-compute-tests_amd", "bin")


def app_command(cfg_name, route, out_png, math="fast", extra=()):
    """The standalone app's command line for a BASELINE configuration (the reference's main.cpp surface + this repo's options)."""
    cfg = CONFIGS[cfg_name]
    if cfg["kind"] == "pt":
        cmd = [os.path.join(APP_DIR, "pathtracer"), str(cfg["spp"]), str(cfg["H"]), "--math", math]
        assert cfg["W"] == cfg["H"] * 3 // 2        # main.cpp:24: resx = resy * 3 / 2
    else:
        cmd = [os.path.join(APP_DIR, "mandelbrot"), "--width", str(cfg["W"]), "--height", str(cfg["H"]), "--max-iter", str(cfg["M"])]
        if cfg["ds"]:
            cmd += ["--precision", "ds", "--centre", repr(K4_VIEW["centre"][0]), repr(K4_VIEW["centre"][1]),
                    "--scale", repr(K4_VIEW["scale"][0]), repr(K4_VIEW["scale"][1])]
    cmd += ["--quiet", "--timing-json", "--out", out_png]
    if route == "rgba8":
        cmd += ["--gpu-postprocess"]
    return cmd + list(extra)


def end_to_end(cfg_names=("K2", "K4"), math="fast", probe=None, extra=(), reps=3):
    """"end-to-end seconds incl. gather, D2H, convert, PNG (reported separately)" — the standalone apps (the reference's
    main.cpp flow: init, preRun, run, saveRenderedImage) run as child processes with --timing-json, through both routes:
      host_buffer: run() fills the application's pinned 16-B/pixel storage buffer (mc_*_render), saveRenderedImage converts on the host
                   (float -> u8, + the path tracer's rotation; row stripes on all cores) and writes the PNG  — the reference's own flow;
      rgba8:       --gpu-postprocess: conversion (+ rotation) on the device, 4 B/pixel cross PCIe (mc_*_render_rgba8).
    Times are milliseconds of ONE cold process each — the best of `reps` by total (HIP start-up is `init`); `total` is the app's own clock
    from main() to the file written, `wall` this process's clock around the child (spawn to exit) = before_main (loading, static
    initialisers) + total + after_file (teardown: 1 ms, the apps leave with _Exit once the file is written; 45 - 50 with --full-teardown);
    d2h_gbps = bytes copied / the copy's device time, beside `probe` = a pinned hipMemcpy of the same size (GB/s) when the caller measured one."""
    import subprocess
    import tempfile
    import time
    out = {}
    with tempfile.TemporaryDirectory(prefix="mc_e2e_") as tmp:
        for name in cfg_names:
            cfg = CONFIGS[name]
            entry = {"image": [cfg["W"], cfg["H"]]}
            for route in ("host_buffer", "rgba8"):
                png = os.path.join(tmp, f"{name}_{route}.png")
                # `init` (HIP start-up + context) varies from process to process on one box — 70 ms or 150 - 350, every other run —
                # whatever ran before (profiles/r06_init_spread_probe.txt: neither the route, nor a pause, nor the parent's own context
                # explains it): `reps` processes per entry, the one with the best total is reported, all totals beside it
                runs, err = [], None
                for _ in range(reps):
                    try:
                        t_spawn = time.monotonic()
                        p = subprocess.run(app_command(name, route, png, math, extra), capture_output=True, text=True, timeout=600)
                        t_exit = time.monotonic()
                    except (OSError, subprocess.SubprocessError) as e:     # the app is not built / did not finish: say so, keep the line
                        err = repr(e)[-300:]
                        break
                    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{"timing_ms"')]
                    if p.returncode != 0 or not line:
                        err = (p.stdout + p.stderr)[-300:]
                        break
                    j = json.loads(line[0])
                    t = j["timing_ms"]
                    t["wall"] = (t_exit - t_spawn) * 1e3
                    if "main_at_ms" in j:      # the app's CLOCK_MONOTONIC at main() and at its end: the same clock as time.monotonic()
                        t["before_main"] = j["main_at_ms"] - t_spawn * 1e3
                        t["after_file"] = t_exit * 1e3 - j["end_at_ms"]
                    runs.append(t)
                if err is not None or not runs:
                    entry[route] = {"error": err}
                    continue
                t = min(runs, key=lambda r: r["total"])
                t["total_all"] = [r["total"] for r in runs]
                nbytes = cfg["W"] * cfg["H"] * (16 if route == "host_buffer" else 4)
                t["d2h_bytes"] = nbytes
                # (a streamed save — bin/mandelbrot at K4 — overlaps its copies with the kernels: `copy` is then what was NOT hidden, no rate)
                t["d2h_gbps"] = nbytes / (t["copy"] * 1e-3) / 1e9 if t["copy"] > 0 and not t.get("streamed_bands") else None
                if probe and probe.get(nbytes):
                    t["d2h_probe_gbps"] = probe[nbytes]
                    t["d2h_vs_probe"] = t["d2h_gbps"] / probe[nbytes] if t["d2h_gbps"] else None
                t["png_bytes"] = os.path.getsize(png) if os.path.exists(png) else None
                entry[route] = t
            out[name] = entry
    out["note"] = ("one cold app process per entry (bin/pathtracer, bin/mandelbrot --timing-json); ms; init = HIP start-up + context, alloc = "
                   "the pinned storage buffer (mc_host_alloc), run = the blocking render call = kernel + copy (device time) + launch / sync, "
                   "convert = host float -> u8 (+ rotation; 0 when done on the device), png = encode + write, total = process wall time.  "
                   "Round 6: the apps warm the kernel family up on a helper thread from init() (warmup; warmup_wait = what run() still waited "
                   "for it): `kernel` no longer contains the code "
                   "object's first use; and the host_buffer route converts inside the PNG writer's stripe workers: `convert` is 0, `png` contains it.  "
                   "Three processes per entry, the best total reported (total_all = all three): `init` alternates between 70 and 150 - 350 ms on one "
                   "box.  wall = the parent's clock around the process = before_main + total + after_file (the apps leave with _Exit once the "
                   "file is written: 1 ms; --full-teardown: 45 - 50).  streamed_bands > 0 (bin/mandelbrot where W x H x M >= 1e11): the image was rendered in "
                   "pipelined row bands and encoded meanwhile — kernel = first launch to the last kernel's end, copy = what of the copies was not hidden, "
                   "png = what was left to encode after run()")
    return out


def pinned_copy_probe(sizes, torch):
    """Device -> pinned host memory, GB/s per size (best of 5 after a warm-up): the rate the storage buffer's trip could reach."""
    res = {}
    for n in sizes:
        src = torch.empty(n, dtype=torch.uint8, device="cuda")
        dst = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        best = 1e9
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dst.copy_(src, non_blocking=True)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                best = min(best, e0.elapsed_time(e1))
        res[n] = n / (best * 1e-3) / 1e9
        del src, dst
    return res


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 10; 3 for K3, whose step is 4e10 samples)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 2; 1 for K3)")
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="BASELINE configuration (default K2)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOAD_ALIAS), help="round-1 spelling of --config")
    ap.add_argument("--math", default="fast", choices=["fast", "careful", "strict"],
                    help="path tracer math mode: fast = gfx950 hardware transcendentals + contraction (toleranced parity, "
                         "tests/test_gpu_fullsize.py pins it at K2), careful = the fast mode's second tier asked for explicitly (no "
                         "contraction, division / sqrt / rsq rounded as the reference rounds them; the host selects it by itself from "
                         "four spheres on), strict = IEEE + mc math (bit-identical to the oracle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end_to_end block (the standalone apps timed as child processes)")
    ap.add_argument("--spp", type=int, default=None, help="override spp (diagnostics only; invalidates the headline)")
    ap.add_argument("--width", type=int, default=None, help="override image width (diagnostics only)")
    ap.add_argument("--height", type=int, default=None, help="override image height (per GPU for weak scaling; diagnostics only)")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region, rank 0 re-renders the whole image on its own GPU and checks that the "
                         "gathered N-rank image is bit-identical (multi-GPU == single-GPU invariant, SURVEY §8e)")
    ap.add_argument("--exchange", default=None, choices=["f32", "rgba8"],
                    help="path tracer, N > 1: what the ranks send to rank 0 — f32: the fp32 vec4 tiles (16 B/pixel, the storage-buffer contract; "
                         "default for K2), rgba8: tiles converted by their owners, 4 B/pixel (SURVEY 8(f)1; default for K3)")
    ap.add_argument("--no-multi", action="store_true",
                    help="N > 1: skip the `multi` block (BASELINE's 8-GPU configurations K3 and K4 after the headline)")
    ap.add_argument("--allow-sync-exchange", action="store_true",
                    help="N > 1 on RCCL: accept a run whose asynchronous exchange fell back to the synchronous path (default: exit 3)")
    a = ap.parse_args()
    # the `multi` block: N > 1 from the plain command — the default headline with nothing overridden
    a.multi = [] if (a.no_multi or a.config or a.workload or a.spp or a.width or a.height or a.gpus < 2) else ["K3", "K4"]
    if a.config and a.workload and WORKLOAD_ALIAS[a.workload] != a.config:
        ap.error("--config and --workload disagree")
    a.config = a.config or (WORKLOAD_ALIAS[a.workload] if a.workload else "K2")
    if a.steps is None:
        a.steps = 3 if a.config == "K3" else 10
    if a.warmup is None:
        a.warmup = 1 if a.config == "K3" else 2
    return a


def profiler_preload():
    """A profiler's preloaded library (rocprofv3 --pmc and friends) initialises the GPU before Python starts; every child this
    process starts would then be an exec from a GPU-initialised process image, which this pool forbids."""
    env = os.environ
    if "rocprof" in env.get("LD_PRELOAD", "").lower():
        return "LD_PRELOAD=" + env["LD_PRELOAD"]
    for k in env:
        if k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_TOOL_", "ROCTRACER_")):
            return k
    return None


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (the driver's N = 1 command shape): start the one-rank-per-GPU
    launcher as a CHILD process — never os.exec*, and before this process has imported torch.cuda or made any HIP call — wait for
    it, relay rank 0's single JSON line and the exit code."""
    import socket
    import subprocess
    hint = (f"launch it as: python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
            f"--master-port <port> bench.py --gpus {args.gpus} ...")
    why = profiler_preload()
    if why:
        sys.exit(f"bench.py --gpus {args.gpus}: a profiler preload is active ({why}); not starting the ranks from a process it has "
                 f"initialised the GPU in — {hint} (the program after `--` must be the launcher itself)")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this image
    env["MC_BENCH_SPAWNED_BY"] = str(os.getpid())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)      # stderr passes through
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    for ln in p.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if p.returncode != 0:
        for ln in lines:            # a line printed before the failure (exit 3 / 4 / 5: sync fallback, failed equality, multi block) is relayed
            print(ln, flush=True)
        sys.exit(f"bench.py --gpus {args.gpus}: the launcher exited with {p.returncode}" if p.returncode > 0 else p.returncode)
    if len(lines) != 1:
        sys.exit(f"bench.py --gpus {args.gpus}: expected one JSON line from rank 0, got {len(lines)}")
    print(lines[0], flush=True)
    return 0


class Job:
    """This process's place in the job: rank, world size, device, context, launch stream (set up once by main())."""


def measure(job, cfg_name, steps, warmup, rank, n, args, exchange="f32", verify=False, keep_image=False):
    """Times `steps` steps of BASELINE configuration `cfg_name` on the ranks (rank, n) — n = 1 with rank 0 is a plain single-GPU
    measurement with no collective anywhere (rank 0's own one-GPU rate inside a multi-rank job).  Returns a dict of what was
    measured; `image` (rank 0: the assembled result, fp32 storage buffer or — exchange "rgba8" — the RGBA8 image) stays on the GPU
    when keep_image is set.  Collective when n > 1: every rank of the job calls it with the same arguments."""
    import numpy as np
    torch, dist, B, S, ctx, stream = job.torch, job.dist, job.B, job.S, job.ctx, job.stream
    ROW_BLOCK = S.ROW_BLOCK
    cfg = CONFIGS[cfg_name]
    is_pt = cfg["kind"] == "pt"
    weak = cfg["scaling"] == "weak"
    W = args.width or cfg["W"]
    H = (args.height or cfg["H"]) * (n if weak else 1)
    m = {"cfg_name": cfg_name, "cfg": cfg, "is_pt": is_pt, "W": W, "H": H, "n": n, "rank": rank, "steps": steps, "warmup": warmup}
    if is_pt:
        spp = args.spp or cfg["spp"]
        math_mode = {"fast": B.PT_MATH_FAST, "careful": B.PT_MATH_FAST_CAREFUL, "strict": B.PT_MATH_STRICT}[args.math]
        pt_flags = int(os.environ.get("MC_PT_FLAGS", "0"), 0)    # experiments only (mc_pathtrace_params.flags)
        p = S.shard(B.pathtrace_params(W, H, spp, math_mode=math_mode, flags=pt_flags), rank, n)
        units_per_step = W * H * spp                         # samples
        m.update(flops_per_unit=FLOPS_PER_SAMPLE_PT, metric="path-traced samples/s", unit="samples/s", spp=spp,
                 workload_name=f"{cfg_name}: pathtrace {W}x{H} spp{spp} default-scene math={args.math}")
    else:
        M, ds = cfg["M"], cfg["ds"]
        kw = dict(max_iter=M)
        if ds:
            kw.update(precision=B.PRECISION_DS, centre=K4_VIEW["centre"], scale=K4_VIEW["scale"])
        p = S.shard(B.mandelbrot_params(W, H, **kw), rank, n)
        units_per_step = None                                # pixel-iters: data dependent, counted after the run
        m.update(flops_per_unit=FLOPS_PER_PIXEL_ITER_DS if ds else FLOPS_PER_PIXEL_ITER_F32, metric="Mandelbrot pixel-iters/s",
                 unit="pixel-iters/s", workload_name=f"{cfg_name}: mandelbrot{'_ds' if ds else ''} {W}x{H} M{M}")
    m["p"] = p
    owns = S.owns_rows(p)                                     # False for a rank beyond the last row block (renders nothing)
    rows_local = B.tile_rows(p)
    rows_padded = S.padded_tile_rows(H, n) if n > 1 else rows_local
    m.update(rows_local=rows_local, rows_padded=rows_padded)
    # What a rank renders into, and what it sends to rank 0 (S.Exchange: receive buffers allocated once, the gather asynchronous,
    # rank 0's re-assembly on a side stream, two buffer sets — step i's exchange overlaps step i + 1's render):
    #   path tracer, exchange "f32": the fp32 vec4 tile (16 B/pixel; samples are never split across ranks);
    #   path tracer, exchange "rgba8" (SURVEY 8(f)1): every rank converts its tile on its own GPU (pathtracerApp.h:212-219), 4 B/pixel
    #   travel, rank 0 de-interleaves and applies the point reflection on bytes (:236-243) — the image saveRenderedImage would write;
    #   Mandelbrot, N > 1: ONLY the iteration counts — uint16 for max_iter <= 65535 (2 B/pixel) — from which rank 0 rebuilds the
    #   vec4 storage buffer through the colour table (the colour is a function of the count alone); N = 1: vec4 + counts.
    narrow = (not is_pt) and n > 1 and p.max_iter <= 65535
    rgba8 = is_pt and n > 1 and exchange == "rgba8"
    m.update(narrow=narrow, rgba8=rgba8)
    f32_tile = None
    if rgba8:
        f32_tile = torch.zeros((rows_padded, W, 4), dtype=torch.float32, device="cuda")
        ex = S.Exchange(rank, n, (rows_padded, W, 4), torch.uint8, "cuda")
    elif is_pt or n == 1:
        ex = S.Exchange(rank, n, (rows_padded, W, 4), torch.float32, "cuda")
    elif narrow:
        p.flags |= B.MANDEL_ITERS_U16
        ex = S.Exchange(rank, n, (rows_padded, W, 2), torch.uint8, "cuda")
    else:
        ex = S.Exchange(rank, n, (rows_padded, W), torch.int32, "cuda")
    iters_t = torch.zeros((rows_padded, W), dtype=torch.int32, device="cuda") if (not is_pt and n == 1) else None
    root = n > 1 and rank == 0
    full = torch.empty((H, W, 4), dtype=torch.uint8 if rgba8 else torch.float32, device="cuda") if root else None
    full_iters = torch.empty((H, W), dtype=torch.int32, device="cuda") if (root and not is_pt) else None

    ev_k0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ev_k1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ev_g1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]

    def whole(q):
        q = copy.copy(q)
        q.row_begin, q.row_end, q.row_block, q.row_stride = 0, H, 0, 0
        return q

    def assemble(recv, stream_handle):                        # rank 0, on the exchange's side stream
        if rgba8:
            ctx.assemble_rgba8_device(recv.data_ptr(), W, H, n, ROW_BLOCK, rows_padded, True, full.data_ptr(), stream=stream_handle)
        elif is_pt:
            S.assemble_device(ctx, recv, W, H, n, full, stream=stream_handle)
        else:
            S.assemble_mandelbrot_device(ctx, whole(p), recv, n, full, full_iters, stream=stream_handle)

    def step(i=None, k=0):
        tile = ex.tile(k)
        if i is not None:
            ev_k0[i].record()
        if owns:
            if rgba8:      # render, then this rank's own float -> u8 (scale 1, no rotation: tile rows are not image rows yet)
                ctx.pathtrace_device(p, f32_tile.data_ptr(), stream=stream)
                ctx.convert_rgba8_device(f32_tile.data_ptr(), W, rows_local, 1.0, False, tile.data_ptr(), stream=stream)
            elif is_pt:
                ctx.pathtrace_device(p, tile.data_ptr(), stream=stream)
            elif n > 1:
                ctx.mandelbrot_device(p, 0, tile.data_ptr(), stream=stream)
            else:
                ctx.mandelbrot_device(p, tile.data_ptr(), iters_t.data_ptr(), stream=stream)
        if i is not None:
            ev_k1[i].record()
        if n > 1:
            ex.submit(k, assemble, ev_g1[i] if i is not None else None)

    def fence():
        ex.finish()
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(warmup):
        step(None, w)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i, warmup + i)
    fence()
    dt = time.perf_counter() - t0
    if n > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev_k0, ev_k1)]))
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev_k1, ev_g1)])) if n > 1 else 0.0
    tile = ex.tiles[(warmup + steps - 1) % len(ex.tiles)]   # the last step's tile (units are counted from it)

    # ---- units ------------------------------------------------------------------------------------
    if is_pt:
        local_units = W * rows_local * p.spp
    else:
        if n == 1:
            it = iters_t[:rows_local].to(torch.int64)
        elif narrow:
            it = tile[:rows_local].contiguous().view(torch.int16).to(torch.int64).squeeze(-1) & 0xffff
        else:
            it = tile[:rows_local].to(torch.int64)
        M = p.max_iter
        local_units = int(torch.where(it < M, it + 1, torch.full_like(it, M)).sum().item())   # loop bodies of the REFERENCE algorithm
        if n > 1:
            tot = torch.tensor([local_units], dtype=torch.int64, device="cuda")
            dist.all_reduce(tot)
            units_per_step = int(tot.item())
        else:
            units_per_step = local_units
    m.update(dt=dt, kernel_ms=kernel_ms, gather_ms=gather_ms, gather_bytes=int(ex.bytes_per_rank) if n > 1 else 0,
             local_units=local_units, units_per_step=units_per_step, value=units_per_step * steps / dt,
             ms_per_step=dt / steps * 1e3, tile=tile, sync_mode=bool(ex.sync_mode), fell_back=bool(getattr(ex, "fell_back", False)),
             exchange_async=bool(n > 1 and not ex.sync_mode and job.backend == "nccl"))

    # ---- per-rank evidence: who rendered what on which device, how long (gathered to rank 0) ----------------------
    m["ranks_info"] = None
    if n > 1:
        mine = torch.tensor([rank, job.device_index, rows_local, local_units, kernel_ms, gather_ms, dist.get_world_size(),
                             1.0 if m["fell_back"] else 0.0], dtype=torch.float64, device="cuda" if job.backend == "nccl" else "cpu")
        allv = [torch.empty_like(mine) for _ in range(n)] if rank == 0 else None
        dist.gather(mine, allv, dst=0)
        if rank == 0:
            m["ranks_info"] = [{"rank": int(v[0]), "device": int(v[1]), "rows": int(v[2]), "units_per_step": int(v[3]),
                                "kernel_ms": round(float(v[4]), 4), "gather_ms": round(float(v[5]), 4),
                                "world_size_seen": int(v[6]), **({"exchange_fell_back": True} if v[7] else {})}
                               for v in (t.cpu() for t in allv)]
        # one rank's fallback is everyone's business: the collective below tells every rank (exit codes must agree)
        fb = torch.tensor([1.0 if m["fell_back"] else 0.0], dtype=torch.float64, device="cuda" if job.backend == "nccl" else "cpu")
        dist.all_reduce(fb, op=dist.ReduceOp.MAX)
        m["any_fell_back"] = bool(fb.item())

    # ---- optional self-check: the N-rank result == rank 0's own single-GPU render, bit for bit (outside the timed region) -------
    # exchange "rgba8": the RGBA8 images must agree byte for byte AND — so that nothing hides behind the quantisation — the ranks'
    # fp32 tiles, gathered once more here (untimed, 16 B/pixel), must re-assemble to the single-GPU storage buffer bit for bit.
    m["verified"] = None
    if verify and n > 1:
        gathered_f32 = S.gather_tiles(f32_tile, rank, n) if rgba8 else None            # collective: every rank
        if rank == 0:
            q = whole(p)
            if not is_pt:
                q.flags &= ~B.MANDEL_ITERS_U16
            single = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
            if is_pt:
                ctx.pathtrace_device(q, single.data_ptr(), stream=stream)
            else:
                ctx.mandelbrot_device(q, single.data_ptr(), 0, stream=stream)
            torch.cuda.synchronize()
            if rgba8:
                single_u8 = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
                ctx.convert_rgba8_device(single.data_ptr(), W, H, 1.0, True, single_u8.data_ptr(), stream=stream)
                again = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
                S.assemble_device(ctx, gathered_f32, W, H, n, again, stream=stream)
                torch.cuda.synchronize()
                ok = bool(torch.equal(single_u8, full)) and bool(torch.equal(single.view(torch.int32), again.view(torch.int32)))
                del single_u8, again
            else:
                ok = bool(torch.equal(single.view(torch.int32), full.view(torch.int32)))
            m["verified"] = ok
            del single
        del gathered_f32
    m["image"] = full if (keep_image and root) else (tile if (keep_image and n == 1) else None)
    m["image_iters"] = full_iters if (keep_image and root) else (iters_t if (keep_image and n == 1) else None)
    m["f32_tile"] = f32_tile if keep_image else None
    return m


K4_PIXEL_ITERS = 41176259776      # the frozen checksum of executed loop bodies of K4 (7680 x 5120, M = 50 000, two-float)


def multi_block(job, args):
    """VERDICT r5 item 1 — what the driver's one command measures at N > 1 besides the (weak-scaled) K2 headline: BASELINE's two 8-GPU
    configurations, K3 and K4, strong-scaled over the N ranks, each with
      value                 whole-job units / max-over-ranks wall time of the timed steps (the headline's own method);
      ranks                 per rank: device, rows, kernel_ms, gather_ms, the world size it saw;
      exchange_async        whether the exchange overlapped the next step's render (False on the rehearsal backend);
      equal_to_single_gpu   the assembled N-rank result against rank 0's OWN whole-image render in the same job, bit for bit — K3: the
                            RGBA8 image and the re-gathered fp32 storage buffer; K4: the storage buffer, the iteration plane and the
                            frozen checksum of loop bodies (41 176 259 776);
      retained_per_gpu      (value / N) / rank 0's one-GPU rate on the same configuration measured in the same job, same timing method.
    Collective: every rank calls it.  Returns (block on rank 0 / None elsewhere, whether any rank's asynchronous exchange fell back)."""
    torch, dist = job.torch, job.dist
    rank, n = job.rank, job.n
    if os.environ.get("MC_BENCH_INJECT_MULTI_FAILURE") == "1":      # tests: the failure path of main() (headline kept, exit 5)
        raise RuntimeError("injected failure of the multi block")
    plan = {"K3": dict(steps=3, warmup=1, single_steps=1, single_warmup=0, exchange=args.exchange or "rgba8"),
            "K4": dict(steps=10, warmup=2, single_steps=5, single_warmup=1, exchange="f32")}
    block = {}
    bad = []
    fell_back = False
    margs = argparse.Namespace(width=None, height=None, spp=None, math=args.math)
    for name in args.multi:
        pl = plan[name]
        mm = measure(job, name, pl["steps"], pl["warmup"], rank, n, margs, exchange=pl["exchange"], verify=False, keep_image=True)
        fell_back = fell_back or bool(mm.get("any_fell_back"))
        f32_again = job.S.gather_tiles(mm["f32_tile"], rank, n) if mm["rgba8"] else None   # (untimed; collective)
        entry = None
        if rank == 0:
            single = measure(job, name, pl["single_steps"], pl["single_warmup"], 0, 1, margs, keep_image=True)   # rank 0 alone: no collective
            W, H = mm["W"], mm["H"]
            checks = {}
            if mm["is_pt"]:
                if mm["rgba8"]:
                    u8 = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
                    job.ctx.convert_rgba8_device(single["image"].data_ptr(), W, H, 1.0, True, u8.data_ptr(), stream=job.stream)
                    again = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
                    job.S.assemble_device(job.ctx, f32_again, W, H, n, again, stream=job.stream)
                    torch.cuda.synchronize()
                    checks["rgba8_image"] = bool(torch.equal(u8, mm["image"]))
                    checks["f32_storage_buffer"] = bool(torch.equal(again.view(torch.int32), single["image"].view(torch.int32)))
                    del u8, again
                else:
                    checks["f32_storage_buffer"] = bool(torch.equal(mm["image"].view(torch.int32), single["image"].view(torch.int32)))
            else:
                checks["f32_storage_buffer"] = bool(torch.equal(mm["image"].view(torch.int32), single["image"].view(torch.int32)))
                checks["iteration_plane"] = bool(torch.equal(mm["image_iters"], single["image_iters"][:H]))
                checks["pixel_iters"] = mm["units_per_step"]
                if name == "K4":
                    checks["pixel_iters_frozen"] = K4_PIXEL_ITERS
                    checks["pixel_iters_match"] = mm["units_per_step"] == K4_PIXEL_ITERS == single["units_per_step"]
            equal = all(v for k, v in checks.items() if isinstance(v, bool))
            if not equal:
                bad.append(name)
            entry = {"workload": mm["workload_name"], "metric": mm["metric"], "unit": mm["unit"], "scaling": "strong",
                     "value": mm["value"], "ms_per_step": mm["ms_per_step"], "steps": pl["steps"], "warmup": pl["warmup"],
                     "image": [W, H], "n_gpus": n, "world_size": dist.get_world_size(), "ranks": mm["ranks_info"],
                     "exchange": ("RGBA8 tiles converted by their owners (4 B/pixel)" if mm["rgba8"] else
                                  "fp32 vec4 tiles (16 B/pixel)" if mm["is_pt"] else
                                  ("uint16" if mm["narrow"] else "uint32") + " iteration counts"),
                     "exchange_async": mm["exchange_async"], "gather_ms_rank0": round(mm["gather_ms"], 4),
                     "gather_bytes_per_rank": mm["gather_bytes"],
                     "single_gpu": {"value": single["value"], "ms_per_step": single["ms_per_step"], "kernel_ms": single["kernel_ms"],
                                    "steps": pl["single_steps"], "device": job.device_index,
                                    "note": "rank 0 alone on its GPU, the whole image, same job, same timing method; the other ranks idle"},
                     "retained_per_gpu": (mm["value"] / n) / single["value"],
                     # north_star: "as absolute numbers and as fraction of the fp32-ALU roofline" at every N — the whole job's
                     # algorithmic flops per second (wall time of the timed steps, exchange included) over N x the one-GPU peak
                     "roofline": {"bound": "valu", "unit": "TFLOP/s", "achieved": mm["value"] * mm["flops_per_unit"] / 1e12,
                                  "peak": PEAK_FP32_TFLOPS * n, "frac": mm["value"] * mm["flops_per_unit"] / 1e12 / (PEAK_FP32_TFLOPS * n),
                                  "single_gpu_frac": single["value"] * mm["flops_per_unit"] / 1e12 / PEAK_FP32_TFLOPS,
                                  "flops_per_unit": mm["flops_per_unit"],
                                  "basis": "algorithmic flops of all ranks / max-over-ranks wall time of the timed steps (exchange "
                                           "included), over N x 157.3 TFLOP/s"},
                     "equal_to_single_gpu": equal, "checks": checks}
            del single
        del mm, f32_again
        torch.cuda.empty_cache()
        dist.barrier()            # the other ranks wait here while rank 0 renders alone
        if rank == 0:
            block[name] = entry
    if rank == 0:
        block["note"] = ("BASELINE's 8-GPU configurations from the plain command: strong scaling over the job's ranks, interleaved 8-row "
                         "blocks; retained_per_gpu = (value / N) / single_gpu.value; target >= 0.9 at N = 8 (north_star)" +
                         ("; REHEARSAL backend (ranks share GPUs, gather staged through the host): timings are not a measurement"
                          if job.backend != "nccl" else ""))
        block["failed_equality"] = bad
        return block, fell_back
    return None, fell_back


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry
    pkg = entry.load_package()
    B, S = pkg.bindings, pkg.sharding
    ROW_BLOCK = S.ROW_BLOCK

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    if world != n:   # (N > 1 without a launcher never gets here: spawn_ranks)
        n = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback for the product path)")
    # MC_BENCH_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than ranks: the ranks share the visible
    # devices and the gather is staged through host memory (RCCL refuses two ranks on one GPU).  It exercises the
    # same sharding / gather / re-assembly code; its timings are meaningless and are labelled as such.
    backend = os.environ.get("MC_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(device_index)
    if n > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            import datetime   # (180 s: the longest legitimate wait is rank 0's single-GPU K3 render, 2 s; a stuck collective ends the job)
            dist.init_process_group("nccl", rank=rank, world_size=n, device_id=torch.device("cuda", device_index),
                                    timeout=datetime.timedelta(seconds=int(os.environ.get("MC_BENCH_PG_TIMEOUT_S", "180"))))
        else:
            dist.init_process_group(backend, rank=rank, world_size=n)
    ctx = B.Context(device_index)
    dev_name, cus, _ = ctx.device_info()
    # a non-default torch stream: its handle is non-zero, so the C ABI launches on exactly this stream and the
    # torch.cuda.Event pairs below bracket the kernel (a NULL handle would select the context's own stream)
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    job = Job()
    job.torch, job.dist, job.B, job.S, job.ctx, job.stream = torch, dist, B, S, ctx, stream
    job.rank, job.n, job.backend, job.device_index = rank, n, backend, device_index

    # ---- the headline workload ---------------------------------------------------------------------
    cfg_name = args.config
    cfg = CONFIGS[cfg_name]
    is_pt = cfg["kind"] == "pt"
    exchange = args.exchange or ("rgba8" if cfg_name == "K3" else "f32")
    m = measure(job, cfg_name, args.steps, args.warmup, rank, n, args, exchange=exchange, verify=args.verify)
    W, H, p = m["W"], m["H"], m["p"]
    rows_local, narrow, rgba8 = m["rows_local"], m["narrow"], m["rgba8"]
    dt, kernel_ms, gather_ms, gather_bytes = m["dt"], m["kernel_ms"], m["gather_ms"], m["gather_bytes"]
    local_units, value, tile = m["local_units"], m["value"], m["tile"]
    metric, unit, flops_per_unit, workload_name = m["metric"], m["unit"], m["flops_per_unit"], m["workload_name"]
    ranks_info, verified = m["ranks_info"], m["verified"]
    spp = m.get("spp")
    sclk_mhz = ctx.measure_clock() if rank == 0 else None   # the clock this box holds under VALU load (boxes differ by >10 %)
    if args.verify and n > 1 and rank == 0 and not verified:
        sys.exit("bench.py --verify: the gathered multi-rank image differs from the single-GPU render")
    fell_back = bool(m.get("any_fell_back"))

    out = None
    if rank == 0:
        achieved_tflops = local_units * flops_per_unit / (kernel_ms * 1e-3) / 1e12
        if is_pt:   # the kernel the host selected for this request (mc_pathtrace_select_kernel: the same decision the launch made)
            ki = B.pathtrace_select_kernel(p)
            fast_ran = ki.math_mode != B.PT_MATH_STRICT
            pl, sp = B.default_scene()
            disjoint = (not fast_ran) or bool(B.pathtrace_scene_class(pl, sp) & B.PT_SCENE_SPHERES_DISJOINT)   # (template default: true)
            # (the math tier is an int template parameter since round 5 — 0 strict, 1 fast, 2 careful — and rocprofv3 prints it so)
            kern = (f"pathtrace_pool_kernel<{ki.math_mode}, {ki.lanes_per_pixel}, 3, {'true' if disjoint else 'false'}>"
                    if ki.kernel == B.PT_KERNEL_POOL
                    else f"pathtrace_kernel<{B.PT_KERNEL_NAMES[ki.kernel]}, tier={ki.math_mode}, S={ki.lanes_per_pixel}>")
        else:
            kern = "mandelbrot_kernel<StateDS>" if cfg["ds"] else "mandelbrot_kernel<StateF32>"
        traffic, traffic_source = profiled_traffic(cfg_name, args, n)
        exec_flops, exec_source = profiled_executed_lane_flops(cfg_name, args, n)
        # the algorithmic bytes are the 16-B storage-buffer entry per pixel, written ONCE (the Mandelbrot kernel
        # also writes its 4-B iteration count, the parity object).  More launches per step or an accumulator re-read show up
        # in `traffic` (PMC) and in `traffic_ratio`, not here.
        alg_bytes = W * rows_local * (16 if is_pt else 20)
        # The clock the kernel itself held: GRBM_GUI_ACTIVE / 8 XCDs / the duration of the SAME profiled launches (tools/summarize_prof.py,
        # `kernel_clock_ghz`), quoted only from a summary of this build (profiled_summary).  NOT mc_context_measure_clock: that probe
        # reads the clock under ITS OWN dense FMA chain — 2.15 GHz after a K3 step during which the path tracer held 2.38 GHz
        # (profiles/r04_k3_clock.txt) — and is kept only to compare boxes.
        prof_entries, prof_file = profiled_summary(cfg_name, args, n)
        prof_ghz = None
        if prof_entries and len(prof_entries) == 1:
            prof_ghz = next(iter(prof_entries.values())).get("derived", {}).get("kernel_clock_ghz")
        contracted = is_pt and args.math == "fast"
        if is_pt and args.math == "careful":
            note = ("the careful tier of fast math: no contraction, division / sqrt / rsq rounded as the reference rounds them, identities "
                    "of exact arithmetic not executed where they are free of side effects (toleranced parity, p99.9 0.44 at K2)")
        elif is_pt and args.math == "fast":
            note = ("fast math: hardware transcendentals, a*b+c contraction, identities of exact arithmetic not executed (toleranced "
                    "parity); `achieved` counts the REFERENCE's arithmetic per sample, `executed` what the ALUs did; the strict "
                    "kernel, reported beside it, forbids all three")
        elif is_pt:
            note = "parity forbids contraction: one issue slot per flop"
        elif cfg["ds"]:
            note = ("142 = the reference composition's literal flop count per pixel-iteration; the kernel gets the same bits from "
                    "~87 issued instructions (Dekker error term = one fma, exact), so the lane-op fraction may exceed 1")
        else:
            note = ("pixel-iters are loop bodies of the REFERENCE algorithm (n + 1 per escaping pixel, M per interior pixel): converged "
                    "tiles leave early through exact cycle detection, so `achieved` is reference-equivalent work per second and may "
                    "exceed the issue ceiling; `executed` is the arithmetic that actually ran")
        sync_exchange = m["sync_mode"] or backend != "nccl"
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name, "baseline_config": cfg_name, "image": [W, H], "rows_per_gpu": rows_local,
                       "tiling": "whole image" if n == 1 else f"interleaved {ROW_BLOCK}-row blocks, RCCL gather to rank 0",
                       **({"exchange": (("RGBA8 tiles converted by their owners (4 B/pixel); rank 0 de-interleaves and point-reflects the bytes"
                                         if rgba8 else "fp32 vec4 tiles (16 B/pixel)") if is_pt else
                                        ("uint16" if narrow else "uint32") + " iteration counts; rank 0 rebuilds the vec4 buffer through the colour table") +
                                       ("; synchronous gather on the render stream" if sync_exchange else
                                        "; asynchronous gather + re-assembly on a side stream, overlapping the next step's render")}
                          if n > 1 else {}),
                       **({"exchange_async": m["exchange_async"], "exchange_fell_back": fell_back} if n > 1 else {}),
                       "device": dev_name, "compute_units": cus, "sclk_mhz_probe_kernel": round(sclk_mhz, 1),
                       "sclk_note": "clock under the PROBE kernel's dense FMA chain (compares boxes); the timed kernel's own clock is roofline.kernel_clock_ghz",
                       **({"unit_note": "reference-equivalent pixel-iterations (see roofline.lane_ops.note)"} if not is_pt else {}),
                       **({"backend": backend, "world_size": dist.get_world_size(), "ranks": ranks_info,
                           "gather_ms_rank0": round(gather_ms, 4), "gather_bytes_per_rank": gather_bytes,
                           "gather_note": "kernel end -> gathered + re-assembled on rank 0 (side stream; overlaps the next render); includes waiting for the slowest rank"}
                          if n > 1 else {}),
                       **({"rehearsal": "MC_BENCH_BACKEND=gloo: ranks share GPUs, gather staged through the host; "
                                        "timings are NOT a measurement"} if backend != "nccl" and n > 1 else {}),
                       **({"verified_equal_to_single_gpu": verified} if verified is not None else {})},
            "roofline": {"bound": "valu", "kernel": kern, "achieved": achieved_tflops, "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tflops / PEAK_FP32_TFLOPS,
                         "kernel_clock_ghz": prof_ghz,   # of the profiled launches themselves (same pass: cycles / duration)
                         "kernel_clock_source": (f"profiles/{prof_file}: GRBM_GUI_ACTIVE / 8 / the profiled launches' own duration" if prof_ghz else None),
                         **({"frac_at_kernel_clock": achieved_tflops * 1e12 / (2.0 * cus * 4 * 32 * prof_ghz * 1e9)} if prof_ghz else {}),
                         # `frac` is defined on the REFERENCE's arithmetic (algorithmic flops per unit); `executed_frac` is the hardware's
                         # view — the fp32 lane-flops the ALUs performed (PMC instruction mix x active lanes) over the same time
                         # executed fp32 lane-flops (committed PMC instruction mix) over the live kernel time
                         "executed": (exec_flops / (kernel_ms * 1e-3) / 1e12) if exec_flops else None,
                         "executed_frac": (exec_flops / (kernel_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS) if exec_flops else None,
                         "executed_source": exec_source or _STALE.get(cfg_name),
                         "build_id": B.lib().mc_build_id().decode(),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_ratio": (traffic / alg_bytes) if traffic else None,
                         **({"traffic_note": "the strict sample-pool kernel spills three values per lane around its batch refill (the 80-VGPR budget of "
                                             "6 waves per SIMD), none reloaded inside an iteration since round 6: ~5 MB per launch beside the 8.64 MB "
                                             "storage buffer, which is written once (rounds 3-5: ten values, 94-106 MB)"}
                            if (traffic and is_pt and args.math == "strict" and traffic / alg_bytes > 1.2) else {}),
                         "kernel_ms": kernel_ms, "flops_per_unit": flops_per_unit,
                         "hbm": {"algorithmic_bytes": alg_bytes, "gbps": alg_bytes / (kernel_ms * 1e-3) / 1e9, "peak_gbps": 8000.0},
                         # the issue-slot view — one slot per flop — only where the kernel does not contract (strict path tracer,
                         # Mandelbrot): for the fast path tracer the quotient of REFERENCE flops and unfused slots means nothing
                         "lane_ops": ({"note": note} if contracted else
                                      {"achieved": achieved_tflops * 1e12, "peak": PEAK_LANE_OPS,
                                       "frac": achieved_tflops * 1e12 / PEAK_LANE_OPS, "note": note})},
        }
    del m

    # ---- BASELINE's 8-GPU configurations (K3, K4) from the plain command: N > 1, the default headline, nothing overridden ------------
    multi_failed = []
    multi_error = None
    if n > 1 and args.multi:
        torch.cuda.empty_cache()
        # An exception in here (a rank that cannot allocate, a collective the backend refuses ...) must not cost the headline that was
        # already measured: the line is printed with the error in `multi`, and the job exits 5.  (The other ranks of a stuck collective
        # are ended by the launcher when this rank exits, or by the process group's timeout — 180 s, set at init.)
        try:
            blk, multi_fell_back = multi_block(job, args)
        except Exception as e:      # noqa: BLE001
            import traceback
            multi_error = f"rank {rank}: {e!r}"
            print(f"[bench.py] the multi block failed on rank {rank}:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
            blk, multi_fell_back = ({"error": multi_error, "failed_equality": []} if rank == 0 else None), False
        fell_back = fell_back or multi_fell_back
        if rank == 0:
            out["multi"] = blk
            multi_failed = blk["failed_equality"]

    # ---- secondary metric + strict-math leg: rank 0, N = 1, default headline only, outside the timed region ----------
    if rank == 0 and n == 1 and not args.no_secondary and cfg_name == "K2":
        k1 = CONFIGS["K1"]
        q = B.mandelbrot_params(k1["W"], k1["H"], max_iter=k1["M"])
        rg = torch.empty((k1["H"], k1["W"], 4), dtype=torch.float32, device="cuda")
        itr = torch.empty((k1["H"], k1["W"]), dtype=torch.int32, device="cuda")
        for _ in range(3):
            ctx.mandelbrot_device(q, rg.data_ptr(), itr.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ctx.mandelbrot_device(q, rg.data_ptr(), itr.data_ptr(), stream=stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        it64 = itr.to(torch.int64)
        pi = int(torch.where(it64 < k1["M"], it64 + 1, torch.full_like(it64, k1["M"])).sum().item())
        tf = pi * FLOPS_PER_PIXEL_ITER_F32 / (ms * 1e-3) / 1e12
        k1_args = argparse.Namespace(width=None, height=None, spp=None, math=args.math)
        k1_exec, k1_exec_src = profiled_executed_lane_flops("K1", k1_args, 1)
        out["secondary"] = {"metric": "Mandelbrot pixel-iters/s", "value": pi / (ms * 1e-3), "unit": "pixel-iters/s",
                            "workload": f"K1: mandelbrot {k1['W']}x{k1['H']} M{k1['M']} fp32", "pixel_iters": pi, "kernel_ms": ms,
                            "unit_note": "reference-equivalent pixel-iterations (n + 1 per escaping pixel, M per interior pixel)",
                            "roofline": {"bound": "valu", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                         "frac": tf / PEAK_FP32_TFLOPS, "lane_ops_frac": tf * 1e12 / PEAK_LANE_OPS,
                                         "frac_note": "reference-equivalent work per second over the peak: converged tiles leave early (exact cycle "
                                                      "detection), so this is NOT a utilisation figure and may exceed 1 in lane-op terms; the hardware "
                                                      "fraction is executed_frac",
                                         "executed_frac": (k1_exec / (ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS) if k1_exec else None,
                                         "executed_source": k1_exec_src or _STALE.get("K1")}}
        if args.math == "fast" and not (args.width or args.height or args.spp):
            # the same workload with MC_PT_MATH_STRICT (IEEE divide/sqrt + mc math: bit-identical to the oracle) — which is what the C
            # ABI's default parameters, both standalone apps and a reference tree bound to this library run unless they ask for
            # fast math (mc_pathtrace_default_params, host/main.cpp --math): `app_default` says so beside the headline
            ps = B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_STRICT)
            ctx.pathtrace_device(ps, tile.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            reps = 5
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ctx.pathtrace_device(ps, tile.data_ptr(), stream=stream)
            e1.record()
            torch.cuda.synchronize()
            sms = e0.elapsed_time(e1) / reps
            out["strict_math"] = {"metric": metric, "value": W * H * spp / (sms * 1e-3), "unit": unit, "kernel_ms": sms,
                                  "note": "bit-identical to the CPU oracle (tests/test_gpu_parity.py)"}
            out["app_default"] = {"math": "strict", "kernel_ms": sms, "value": W * H * spp / (sms * 1e-3), "unit": unit,
                                  "frac": W * H * spp * FLOPS_PER_SAMPLE_PT / (sms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                                  "headline_math": args.math, "headline_kernel_ms": kernel_ms,
                                  "note": "what a drop-in user gets without asking: mc_pathtrace_default_params, bin/pathtracer and route B all "
                                          "default to MC_PT_MATH_STRICT (bit-identical to the oracle); the headline is `--math fast` "
                                          "(toleranced parity, tests/test_gpu_fullsize.py)"}
    del tile

    # ---- end to end: the standalone apps as child processes, K2 and K4, both routes; rank 0, N = 1, headline only ----
    if (rank == 0 and n == 1 and cfg_name == "K2" and not args.no_end_to_end and not (args.width or args.height or args.spp)
            and not profiler_preload()):      # (under a profiler's preload no child process is started: spawn_ranks)
        sizes = sorted({CONFIGS[c]["W"] * CONFIGS[c]["H"] * b for c in ("K2", "K4") for b in (16, 4)})
        out["end_to_end"] = end_to_end(("K2", "K4"), args.math, pinned_copy_probe(sizes, torch))

    # ---- CPU baseline: rank 0, N = 1, outside the timed region, a bounded sample of the same workload ----------------
    if rank == 0 and n == 1 and not args.no_cpu_baseline:
        O = entry.load_oracle()   # TEST INFRASTRUCTURE, used here only as the timed CPU baseline
        threads = O.hardware_threads()   # CPUs this process may run on: min(affinity, cgroup quota) - 16 on a one-GPU box
        if is_pt:
            # probe on a band of rows (one oracle row per thread), then size the sample for ~10-20 s of CPU work
            band = min(H, max(threads, 16))
            t = time.perf_counter()
            O.pathtrace(W, H, p.spp, math_mode=O.MATH_LIBM, sample_begin=0, sample_end=1, row_begin=0, row_end=band, nthreads=threads)
            per_sample = (time.perf_counter() - t) / (W * band)
            budget = 20.0      # (the one-sample probe over-estimates the per-sample cost: the sample then takes about half of this)
            if W * H * per_sample <= budget:           # whole image, several samples per pixel
                s_spp = int(max(1, min(p.spp, budget / (W * H * per_sample))))
                rows = (0, H)
            else:                                      # K3-sized image: one sample per pixel over a band of rows
                s_spp = 1
                rows = (0, max(threads, int(budget / (W * per_sample))))
            t = time.perf_counter()
            O.pathtrace(W, H, p.spp, math_mode=O.MATH_LIBM, sample_begin=0, sample_end=s_spp, row_begin=rows[0], row_end=rows[1],
                        nthreads=threads)
            cdt = time.perf_counter() - t
            nsamp = W * (rows[1] - rows[0]) * s_spp
            out["cpu_baseline"] = {"value": nsamp / cdt, "unit": unit, "cores": threads, "kind": "port",
                                   "sample": f"samples 0..{s_spp - 1} of {p.spp} over rows {rows[0]}..{rows[1] - 1} of the "
                                             f"{W}x{H} image ({nsamp} samples, {cdt:.1f} s)"}
        else:
            stride = 1 if p.max_iter <= 1000 else 4    # K1: the whole image (0.5 s of 16 cores); K4: every 4th row (about 10 s): same mix as the image
            rows = list(range(0, H, stride))
            from concurrent.futures import ThreadPoolExecutor
            view = O.make_view(*(K4_VIEW["centre"] + K4_VIEW["scale"])) if cfg["ds"] else O.REF_VIEW

            def one_row(r):   # ctypes releases the GIL: one oracle row per worker thread
                itc = O.mandelbrot_iters(W, H, p.max_iter, view=view, precision=int(cfg["ds"]), row_begin=r,
                                         row_end=r + 1, nthreads=1)
                return O.mandel_pixel_iters(itc, p.max_iter)

            t = time.perf_counter()
            with ThreadPoolExecutor(threads) as tex:
                tot = sum(tex.map(one_row, rows))
            cdt = time.perf_counter() - t
            out["cpu_baseline"] = {"value": tot / cdt, "unit": unit, "cores": threads, "kind": "port",
                                   "sample": f"every {stride}th row of the {W}x{H} image ({tot} pixel-iters, {cdt:.1f} s)"}

    if rank == 0 and "cpu_baseline" in out:
        out["cpu_baseline"]["lavapipe_probe"] = lavapipe_probe()   # all four must exist for kind "reference"; they do not here
    if rank == 0:
        print(json.dumps(out), flush=True)
    # Refusing to degrade silently.  A job that was meant to overlap its exchange (RCCL, not the explicit
    # MC_BENCH_SYNC_EXCHANGE=1) but fell back to the synchronous path has measured something else: the line above says so
    # (`exchange_async`: false, `exchange_fell_back`: true) and the job EXITS NON-ZERO unless --allow-sync-exchange was given.  So does
    # a multi block whose N-rank result differs from rank 0's single-GPU render.  Every rank takes the same exit (the flags were
    # all-reduced), after the process group is torn down.
    code = 0
    if multi_error is not None:     # (no further collective: the ranks may no longer be in step)
        if rank == 0:
            print(f"bench.py: the multi block failed ({multi_error}); the headline above stands, exit 5", file=sys.stderr, flush=True)
        ctx.close()
        os._exit(5)                 # not sys.exit: destroying a process group with a collective in flight can block
    if n > 1:
        flags = torch.tensor([1.0 if fell_back else 0.0, float(len(multi_failed))], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flags, op=dist.ReduceOp.MAX)
        if flags[0].item() and backend == "nccl" and not args.allow_sync_exchange:
            code = 3
            if rank == 0:
                print("bench.py: the asynchronous exchange fell back to the synchronous path on at least one rank — the line above is "
                      "not the overlapped measurement; exit 3 (pass --allow-sync-exchange to accept it)", file=sys.stderr, flush=True)
        if flags[1].item():
            code = 4
            if rank == 0:
                print(f"bench.py: multi-rank result differs from the single-GPU render for {multi_failed}; exit 4", file=sys.stderr, flush=True)
    ctx.close()
    if n > 1:
        dist.destroy_process_group()
    if code:
        sys.exit(code)


if __name__ == "__main__":
    main()
