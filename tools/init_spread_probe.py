"""How `init` (HIP start-up + context) of a cold app process varies from one process to the next on one box, and what it depends on — and
what a process costs OUTSIDE the app's own `total` (main() to the file written): the parent reads CLOCK_MONOTONIC around the process, the
app prints the same clock at main() and at its end.  The same K2 command started 10 times back to back (the app's default: it leaves
with _Exit once the file is written; and with --full-teardown), 10 times with 0.5 s between an exit and the next start, and both again while THIS process
holds an idle GPU context of its own (as bench.py does when it measures its end_to_end block).  ms.
  python tools/init_spread_probe.py > gpurun_out/r06_init_spread_probe.txt"""
import json
import os
import subprocess
import sys
import tempfile
import time

EXTRA = sys.argv[1:]          # e.g. --fast-exit
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def series(n, gap, tmp, extra=()):
    rows = []
    for k in range(n):
        cmd = bench.app_command("K2", "rgba8", os.path.join(tmp, "x.png"), "fast", list(EXTRA) + list(extra))
        t0 = time.monotonic()
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        t1 = time.monotonic()
        line = [ln for ln in p.stdout.splitlines() if ln.startswith('{"timing_ms"')]
        j = json.loads(line[0])
        t = j["timing_ms"]
        rows.append((t["init"], t["total"], (t1 - t0) * 1e3, j["main_at_ms"] - t0 * 1e3, t1 * 1e3 - j["end_at_ms"]))
        if gap:
            time.sleep(gap)
    return rows


def show(label, rows):
    print(f"{label:78s} init  " + " ".join(f"{r[0]:6.0f}" for r in rows))
    print(f"{'':78s} total " + " ".join(f"{r[1]:6.0f}" for r in rows))
    print(f"{'':78s} wall  " + " ".join(f"{r[2]:6.0f}" for r in rows) + "   (parent's clock: spawn to exit)")
    print(f"{'':78s} before main() " + " ".join(f"{r[3]:6.0f}" for r in rows) + "   (spawn, loading, static initialisers)")
    print(f"{'':78s} after the file " + " ".join(f"{r[4]:6.0f}" for r in rows) + "   (teardown, exit, the parent's wait)")


def main():
    with tempfile.TemporaryDirectory(prefix="mc_init_") as tmp:
        show("no GPU context in the parent, back to back", series(10, 0.0, tmp))
        show("the same with --full-teardown (destructors + the runtime's exit handlers)", series(10, 0.0, tmp, ["--full-teardown"]))
        show("no GPU context in the parent, 0.5 s between processes", series(10, 0.5, tmp))
        import torch
        x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
        show("parent holds an idle GPU context, back to back", series(10, 0.0, tmp))
        show("parent holds an idle GPU context, 0.5 s between", series(10, 0.5, tmp))
        del x


if __name__ == "__main__":
    main()
