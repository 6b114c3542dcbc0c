#!/usr/bin/env python3
"""Where the wall time of a whole app run goes ("end-to-end seconds incl. gather, D2H, convert, PNG,
reported separately").  Runs the standalone apps — the reference's main.cpp flow: init, preRun, run, saveRenderedImage — as child
processes with --timing-json for K2 (path trace 900 x 600 x 500), K1 and K4 (Mandelbrot 3200 x 2400 / 7680 x 5120 two-float), through
both routes (bench.py: end_to_end), three times each (the best total is shown), next to a pinned device -> host copy of the same size.
    GPU box:  python tools/end_to_end.py > gpurun_out/r05_end_to_end.txt"""
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
    print("# one cold app process per row, best total of 3; milliseconds.  init = HIP start-up + context; alloc = the pinned storage buffer; kernel / copy = device")
    print("# time of the render (+ on-device conversion) and of the device -> host copy; convert = host float -> u8 (+ rotation), row stripes on all cores;")
    print("# png = encode + write (stripe-parallel zlib); total = process wall time from main() to the file being written")
    print(f"# {'config':6s} {'route':12s} {'init':>8s} {'alloc':>8s} {'kernel':>9s} {'copy':>8s} {'GB/s':>6s} {'vs probe':>8s} {'convert':>8s} {'png':>8s} {'total':>9s}   png bytes")
    for name in cfgs:
        runs = [bench.end_to_end((name,), "fast", probe) for _ in range(3)]
        for route in ("host_buffer", "rgba8"):
            ok = [r[name][route] for r in runs if "error" not in r[name][route]]
            if not ok:
                print(f"  {name:6s} {route:12s} FAILED {runs[0][name][route]}")
                continue
            t = min(ok, key=lambda x: x["total"])
            print(f"  {name:6s} {route:12s} {t['init']:8.1f} {t['alloc']:8.1f} {t['kernel']:9.2f} {t['copy']:8.2f} {t['d2h_gbps'] or 0:6.1f} "
                  f"{t.get('d2h_vs_probe') or 0:8.2f} {t['convert']:8.1f} {t['png']:8.1f} {t['total']:9.1f}   {t['png_bytes']}", flush=True)


if __name__ == "__main__":
    main()
