#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X (driver contract).

  python bench.py --gpus N --steps K --warmup W [--workload pathtrace|mandelbrot|mandelbrot_ds]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one synthetic batch:
  pathtrace (default, BASELINE config K2): render the 900 x 600 default scene at 500 spp — one fused
      HIP launch per rank — and, for N > 1, gather the fp32 row tiles to rank 0 over RCCL and
      re-assemble the storage buffer there.
  mandelbrot (BASELINE config K1): 3200 x 2400, M = 1000, fp32.  mandelbrot_ds: the two-float variant.
Weak scaling: per-GPU work is fixed, the image grows to W x (H*N) rows, interleaved row blocks (mc_row_block() = 8 rows) per
rank (every pixel is keyed by its absolute coordinates, so tiling never changes a pixel's arithmetic;
samples are never split across GPUs — the fp32 accumulation order is part of the parity contract).

Output buffers are resident in HBM (torch tensors); there are no inputs besides 432 B of scene
constants.  The timed region is bracketed by barrier + torch.cuda.synchronize() on both sides; the
value is whole-job units / max-over-ranks time.  One JSON line is printed by rank 0.

`roofline`: fp32 VALU (neither HBM nor MFMA bounds this path: 16 B written per pixel, no contraction).
   achieved = algorithmic fp32 flops per launch (flops/unit from the oracle's op counters) /
   mean kernel time measured with HIP events on the launch stream; peak = 157.3 TFLOP/s (FMA-counted,
   MI355X_MICROARCH.md).  `lane_ops` adds the issue-slot view (78.6e12 lane-ops/s at 2.4 GHz): with
   no contraction allowed every flop occupies a slot, so 0.5 of the FMA-counted peak is the ceiling.
`cpu_baseline`: the CPU oracle (a port, not the reference's Vulkan build — lavapipe/glslang are absent)
   timed on this host's cores over a bounded sample of the same workload.
"""
import argparse
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

K2 = dict(W=900, H=600, spp=500)
K1 = dict(W=3200, H=2400, M=1000)
K4_VIEW = dict(centre=(-0.7436438870371587, 0.13182590420531198), scale=(1e-8, 1e-8 * 2.0 / 3.0))


def profiled_traffic(wl, args, n):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC pass committed under profiles/
    (WRITE_SIZE in its own --pmc pass; exact for 16-B-per-lane stores, MI355X_MICROARCH.md §HBM; these kernels
    read nothing but a <1 MB LUT / 432 B of scene).  None when the run is not the profiled default configuration."""
    if n != 1 or args.width or args.height or args.spp:
        return None
    tag = {"pathtrace": "pt_fast" if args.math == "fast" else "pt_strict", "mandelbrot": "mandel",
           "mandelbrot_ds": "mandel_ds"}[wl]
    for rnd in ("r01d", "r01c", "r01b"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{tag}_pmc_summary.json")
        if os.path.exists(path):
            try:
                for entry in json.load(open(path)).values():
                    b = entry.get("derived", {}).get("hbm_write_bytes")
                    if b:
                        return b
            except (ValueError, OSError):
                return None
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="pathtrace", choices=["pathtrace", "mandelbrot", "mandelbrot_ds"])
    ap.add_argument("--math", default="fast", choices=["fast", "strict"],
                    help="path tracer math mode: fast = gfx950 hardware transcendentals (toleranced parity), "
                         "strict = IEEE + mc math (bit-identical to the oracle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--spp", type=int, default=None, help="override spp (diagnostics only; invalidates the headline)")
    ap.add_argument("--width", type=int, default=None, help="override image width (diagnostics only)")
    ap.add_argument("--height", type=int, default=None, help="override per-GPU image height (diagnostics only)")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region, rank 0 re-renders the whole image on its own GPU and checks that the "
                         "gathered N-rank image is bit-identical (multi-GPU == single-GPU invariant, SURVEY §8e)")
    return ap.parse_args()


def main():
    args = parse()
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
    if world != n:
        if world == 1 and n > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
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
            dist.init_process_group("nccl", rank=rank, world_size=n, device_id=torch.device("cuda", device_index))
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

    # ---- workload ---------------------------------------------------------------------------------
    wl = args.workload
    if wl == "pathtrace":
        W, Hbase, spp = args.width or K2["W"], args.height or K2["H"], args.spp or K2["spp"]
        H = Hbase * n
        math_mode = B.PT_MATH_FAST if args.math == "fast" else B.PT_MATH_STRICT
        pt_flags = int(os.environ.get("MC_PT_FLAGS", "0"), 0)    # experiments only (mc_pathtrace_params.flags)
        p = S.shard(B.pathtrace_params(W, H, spp, math_mode=math_mode, flags=pt_flags), rank, n)
        units_per_step = W * H * spp                         # samples
        flops_per_unit = FLOPS_PER_SAMPLE_PT
        metric, unit = "path-traced samples/s", "samples/s"
        workload_name = f"pathtrace {W}x{H} spp{spp} default-scene math={args.math}"
    else:
        W, Hbase, M = args.width or K1["W"], args.height or K1["H"], K1["M"]
        H = Hbase * n
        ds = wl == "mandelbrot_ds"
        kw = dict(max_iter=M)
        if ds:
            kw.update(precision=B.PRECISION_DS, centre=K4_VIEW["centre"], scale=K4_VIEW["scale"])
        p = S.shard(B.mandelbrot_params(W, H, **kw), rank, n)
        units_per_step = None                                # pixel-iters: data dependent, counted after the run
        flops_per_unit = FLOPS_PER_PIXEL_ITER_DS if ds else FLOPS_PER_PIXEL_ITER_F32
        metric, unit = "Mandelbrot pixel-iters/s", "pixel-iters/s"
        workload_name = f"mandelbrot{'_ds' if ds else ''} {W}x{H} M{M}"

    rows_local = B.tile_rows(p)
    rows_padded = S.padded_tile_rows(H, n) if n > 1 else rows_local
    tile = torch.zeros((rows_padded, W, 4), dtype=torch.float32, device="cuda")
    iters_t = torch.zeros((rows_padded, W), dtype=torch.int32, device="cuda") if wl != "pathtrace" else None
    full = torch.empty((H, W, 4), dtype=torch.float32, device="cuda") if n > 1 and rank == 0 else None

    ev_k0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_k1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            ev_k0[i].record()
        if wl == "pathtrace":
            ctx.pathtrace_device(p, tile.data_ptr(), stream=stream)
        else:
            ctx.mandelbrot_device(p, tile.data_ptr(), iters_t.data_ptr(), stream=stream)
        if i is not None:
            ev_k1[i].record()
        if n > 1:
            gathered = S.gather_tiles(tile, rank, n)          # RCCL gather of the fp32 tiles to rank 0
            if rank == 0:
                S.assemble_device(ctx, gathered, W, H, n, full, stream=stream)

    def fence():
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if n > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev_k0, ev_k1)]))

    # ---- units ------------------------------------------------------------------------------------
    if wl == "pathtrace":
        local_units = W * rows_local * p.spp
    else:
        it = iters_t[:rows_local].to(torch.int64)
        M = p.max_iter
        local_units = int(torch.where(it < M, it + 1, torch.full_like(it, M)).sum().item())   # executed loop bodies
        if n > 1:
            tot = torch.tensor([local_units], dtype=torch.int64, device="cuda")
            dist.all_reduce(tot)
            units_per_step = int(tot.item())
        else:
            units_per_step = local_units
    value = units_per_step * args.steps / dt

    # ---- optional self-check: N-rank image == single-GPU image, bit for bit (outside the timed region) ----------
    verified = None
    if args.verify and n > 1 and rank == 0:
        import copy
        q = copy.copy(p)
        q.row_begin, q.row_end, q.row_block, q.row_stride = 0, H, 0, 0
        single = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        if wl == "pathtrace":
            ctx.pathtrace_device(q, single.data_ptr(), stream=stream)
        else:
            ctx.mandelbrot_device(q, single.data_ptr(), 0, stream=stream)
        torch.cuda.synchronize()
        verified = bool(torch.equal(single.view(torch.int32), full.view(torch.int32)))
        if not verified:
            sys.exit("bench.py --verify: the gathered multi-rank image differs from the single-GPU render")

    out = None
    if rank == 0:
        achieved_tflops = local_units * flops_per_unit / (kernel_ms * 1e-3) / 1e12
        kern = {"pathtrace": "pathtrace_kernel", "mandelbrot": "mandelbrot_kernel<StateF32>",
                "mandelbrot_ds": "mandelbrot_kernel<StateDS>"}[wl]
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name, "image": [W, H], "rows_per_gpu": rows_local,
                       "tiling": "whole image" if n == 1 else f"interleaved {ROW_BLOCK}-row blocks, RCCL gather to rank 0",
                       "device": dev_name, "compute_units": cus,
                       **({"rehearsal": "MC_BENCH_BACKEND=gloo: ranks share GPUs, gather staged through the host; "
                                        "timings are NOT a measurement"} if backend != "nccl" and n > 1 else {}),
                       **({"verified_equal_to_single_gpu": verified} if verified is not None else {})},
            "roofline": {"bound": "valu", "kernel": kern, "achieved": achieved_tflops, "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tflops / PEAK_FP32_TFLOPS,
                         "traffic": profiled_traffic(wl, args, n),
                         "kernel_ms": kernel_ms, "flops_per_unit": flops_per_unit,
                         # HBM is not the bound: the algorithmic bytes are the 16-B storage-buffer entry per pixel
                         # (+4 B iteration count for Mandelbrot), written once
                         "hbm": {"algorithmic_bytes": W * rows_local * (16 if wl == "pathtrace" else 20),
                                 "gbps": W * rows_local * (16 if wl == "pathtrace" else 20) / (kernel_ms * 1e-3) / 1e9,
                                 "peak_gbps": 8000.0},
                         "lane_ops": {"achieved": achieved_tflops * 1e12, "peak": PEAK_LANE_OPS,
                                      "frac": achieved_tflops * 1e12 / PEAK_LANE_OPS,
                                      "note": ("142 = the reference composition's literal flop count; the kernel gets the same "
                                               "bits from ~87 issued instructions (Dekker error term = one fma, exact), "
                                               "so this fraction may exceed 1") if wl == "mandelbrot_ds" else
                                              ("fast math: hardware transcendentals and a*b+c contraction (toleranced parity); "
                                               "the strict kernel, reported beside it, forbids both") if wl == "pathtrace" and args.math == "fast"
                                              else "parity forbids contraction: one issue slot per flop"}},
        }

    # ---- secondary metric + CPU baseline: rank 0, N = 1 only, outside the timed region ------------------
    if rank == 0 and n == 1 and not args.no_secondary and wl == "pathtrace":
        q = B.mandelbrot_params(K1["W"], K1["H"], max_iter=K1["M"])
        rg = torch.empty((K1["H"], K1["W"], 4), dtype=torch.float32, device="cuda")
        itr = torch.empty((K1["H"], K1["W"]), dtype=torch.int32, device="cuda")
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
        pi = int(torch.where(it64 < K1["M"], it64 + 1, torch.full_like(it64, K1["M"])).sum().item())
        tf = pi * FLOPS_PER_PIXEL_ITER_F32 / (ms * 1e-3) / 1e12
        out["secondary"] = {"metric": "Mandelbrot pixel-iters/s", "value": pi / (ms * 1e-3), "unit": "pixel-iters/s",
                            "workload": f"mandelbrot {K1['W']}x{K1['H']} M{K1['M']} fp32", "pixel_iters": pi, "kernel_ms": ms,
                            "roofline": {"bound": "valu", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                         "frac": tf / PEAK_FP32_TFLOPS, "lane_ops_frac": tf * 1e12 / PEAK_LANE_OPS}}
        if args.math == "fast" and not (args.width or args.height or args.spp):
            # the same workload with MC_PT_MATH_STRICT (IEEE divide/sqrt + mc_math: bit-identical to the oracle)
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

    if rank == 0 and n == 1 and not args.no_cpu_baseline:
        O = entry.load_oracle()   # TEST INFRASTRUCTURE, used here only as the timed CPU baseline
        threads = O.hardware_threads()   # CPUs this process may run on: min(affinity, cgroup quota) - 16 on a one-GPU box
        if wl == "pathtrace":
            t = time.perf_counter()   # probe, then size the sample for ~10-20 s of CPU work
            O.pathtrace(W, H, p.spp, math_mode=O.MATH_LIBM, sample_begin=0, sample_end=1, nthreads=threads)
            probe = time.perf_counter() - t
            s_spp = int(max(1, min(p.spp, 12.0 / max(probe, 1e-3))))
            t = time.perf_counter()
            O.pathtrace(W, H, p.spp, math_mode=O.MATH_LIBM, sample_begin=0, sample_end=s_spp, nthreads=threads)
            cdt = time.perf_counter() - t
            out["cpu_baseline"] = {"value": W * H * s_spp / cdt, "unit": unit, "cores": threads, "kind": "port",
                                   "sample": f"samples 0..{s_spp - 1} of {p.spp} over the full {W}x{H} image "
                                             f"({W * H * s_spp} samples, {cdt:.1f} s)"}
        else:
            rows = list(range(0, H, 4))   # every 4th row: same interior/exterior mix as the full image
            from concurrent.futures import ThreadPoolExecutor
            view = O.make_view(*(K4_VIEW["centre"] + K4_VIEW["scale"])) if wl == "mandelbrot_ds" else O.REF_VIEW

            def one_row(r):   # ctypes releases the GIL: one oracle row per worker thread
                itc = O.mandelbrot_iters(W, H, p.max_iter, view=view, precision=int(wl == "mandelbrot_ds"), row_begin=r,
                                         row_end=r + 1, nthreads=1)
                return O.mandel_pixel_iters(itc, p.max_iter)

            t = time.perf_counter()
            with ThreadPoolExecutor(threads) as ex:
                tot = sum(ex.map(one_row, rows))
            cdt = time.perf_counter() - t
            out["cpu_baseline"] = {"value": tot / cdt, "unit": unit, "cores": threads, "kind": "port",
                                   "sample": f"every 4th row of the {W}x{H} image ({tot} pixel-iters, {cdt:.1f} s)"}

    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if n > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
