"""bin/mandelbrot's streamed save (row bands rendered while the PNG workers encode the previous ones) against --no-streamed-save, with the
overlapped and the serial start, K1 and K4, both routes: every run's kernel / copy / png / total, not the best one.  ms.
  python tools/streamed_save_probe.py [reps] > gpurun_out/r06_streamed_save_probe.txt"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    extra_all = sys.argv[2:]
    with tempfile.TemporaryDirectory(prefix="mc_stream_") as tmp:
        for name in ("K1", "K4"):
            for route in ("host_buffer", "rgba8"):
                for start in ((), ("--serial-start",)):
                    for mode in (("--streamed-save",), ("--no-streamed-save",)):
                        rows = []
                        for _ in range(reps):
                            cmd = bench.app_command(name, route, os.path.join(tmp, "x.png"), "fast", list(start) + list(mode) + extra_all)
                            p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                            t = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{"timing_ms"')][0])["timing_ms"]
                            rows.append(t)
                        label = f"{name} {route:11s} {'serial ' if start else 'overlap'} {'unstreamed' if mode[0].startswith('--no') else 'streamed  '}"
                        for key in ("kernel", "copy", "png", "run", "total"):
                            print(f"{label if key == 'kernel' else '':44s} {key:7s}" + " ".join(f"{r[key]:8.1f}" for r in rows), flush=True)


if __name__ == "__main__":
    main()
