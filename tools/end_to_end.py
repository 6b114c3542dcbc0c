#!/usr/bin/env python3
"""Where the wall time of a whole app run goes ("end-to-end seconds incl. gather, D2H, convert, PNG,
reported separately").  Runs the standalone apps — the reference's main.cpp flow: init, preRun, run, saveRenderedImage — as child
processes with --timing-json for K2 (path trace 900 x 600 x 500), K1 and K4 (Mandelbrot 3200 x 2400 / 7680 x 5120 two-float), through
both routes (bench.py: end_to_end), five times each (the best total is shown), next to a pinned device -> host copy of the same size.
`wall` is this process's clock around the child: what `total` (main() to the file written) cannot contain — `pre` = loading and static
initialisers before main(), `post` = teardown and exit (the apps leave with _Exit once the file is written; K2 also with --full-teardown).
Round 6: every configuration twice — the apps' overlapped start (kernel family warmed up on a helper thread from init(); VERDICT r5
item 2) and `--serial-start` (the first launch, with the code object's load, inside run()), on the same build.
    GPU box:  python tools/end_to_end.py > gpurun_out/r06_end_to_end.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    cfgs = ("K2", "K1", "K4")
    sizes = sorted({bench.CONFIGS[c]["W"] * bench.CONFIGS[c]["H"] * b for c in cfgs for b in (16, 4)})
    probe = bench.pinned_copy_probe(sizes, torch)
    print("# pinned device -> host copy (torch, best of 5): " + ", ".join(f"{n / 1e6:.1f} MB {g:.1f} GB/s" for n, g in probe.items()))
    print("# one cold app process per row, best total of 5; milliseconds.  init = HIP start-up + context; alloc = the pinned storage buffer; kernel / copy = device")
    print("# time of the render (+ on-device conversion) and of the device -> host copy; convert = host float -> u8 (+ rotation) as a pass of its own: 0 since")
    print("# round 6 — the host_buffer route converts inside the PNG writer's stripe workers; png = (convert +) filter + deflate + write; total = main() to the file written")
    print("# start: overlap = round 6 (mc_context_warmup_* on a helper thread from init(), joined by run()), serial = --serial-start (first launch inside run());")
    print("# alloc = the storage buffer (mc_host_alloc, in preRun()); warm = the warm-up call on its helper thread, w.wait = what run() still waited for it")
    print("# wall = this process's clock around the child = pre (before main()) + total + post (after the file is written)")
    print(f"# {'config':6s} {'route':12s} {'start':8s} {'init':>7s} {'alloc':>7s} {'warm':>6s} {'w.wait':>6s} {'run':>8s} {'kernel':>8s} {'copy':>7s} {'GB/s':>6s} "
          f"{'convert':>8s} {'png':>7s} {'total':>8s} {'pre':>5s} {'post':>5s} {'wall':>7s}   png bytes")
    for name in cfgs:
        starts = [("overlap", ()), ("serial", ("--serial-start",))] + ([("teardown", ("--full-teardown",))] if name == "K2" else [])
        for start, extra in starts:
            r = bench.end_to_end((name,), "fast", probe, extra, reps=5)
            for route in ("host_buffer", "rgba8"):
                t = r[name][route]
                if "error" in t:
                    print(f"  {name:6s} {route:12s} {start:8s} FAILED {t}")
                    continue
                print(f"  {name:6s} {route:12s} {start:8s} {t['init']:7.1f} {t['alloc']:7.1f} {t['warmup']:6.1f} {t['warmup_wait']:6.1f} "
                      f"{t['run']:8.2f} {t['kernel']:8.2f} {t['copy']:7.2f} {t['d2h_gbps'] or 0:6.1f} {t['convert']:8.1f} {t['png']:7.1f} {t['total']:8.1f} "
                      f"{t['before_main']:5.1f} {t['after_file']:5.1f} {t['wall']:7.1f}   {t['png_bytes']}", flush=True)


if __name__ == "__main__":
    main()
