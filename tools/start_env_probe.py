"""Does any HIP / ROCr environment switch shorten a cold run of bin/pathtracer (K2, fast)?  Seven cold processes per setting; init / run / total from
--timing-json.  GPU box:  python tools/start_env_probe.py > gpurun_out/r06_start_env_probe.txt  (record: profiles/r06_start_env_probe.txt)"""
import os, subprocess, json, statistics, tempfile, sys
APP = "vulkan-compute-tests_amd/bin/pathtracer"
envs = [("baseline", {}), ("HSA_ENABLE_SDMA=0", {"HSA_ENABLE_SDMA": "0"}), ("GPU_ENABLE_COOP_GROUPS=0", {"GPU_ENABLE_COOP_GROUPS": "0"}),
        ("HSA_ENABLE_INTERRUPT=0", {"HSA_ENABLE_INTERRUPT": "0"}), ("GPU_MAX_HW_QUEUES=1", {"GPU_MAX_HW_QUEUES": "1"}),
        ("HIP_VISIBLE_DEVICES=0", {"HIP_VISIBLE_DEVICES": "0"}), ("AMD_DIRECT_DISPATCH=0", {"AMD_DIRECT_DISPATCH": "0"}),
        ("HSA_DISABLE_FRAGMENT_ALLOCATOR=1", {"HSA_DISABLE_FRAGMENT_ALLOCATOR": "1"}), ("baseline again", {})]
with tempfile.TemporaryDirectory() as tmp:
    for name, e in envs:
        inits, totals, runs = [], [], []
        for _ in range(7):
            r = subprocess.run([APP, "500", "600", "--math", "fast", "--quiet", "--timing-json", "--out", os.path.join(tmp, "o.png")], capture_output=True, text=True, env=dict(os.environ, **e))
            if r.returncode: print(name, "FAILED", r.stdout[-200:]); break
            t = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"timing_ms"')][0])["timing_ms"]
            inits.append(t["init"]); totals.append(t["total"]); runs.append(t["run"])
        else:
            print(f"{name:36s} init best {min(inits):6.1f} median {statistics.median(inits):6.1f}   run best {min(runs):6.1f}   total best {min(totals):6.1f} median {statistics.median(totals):6.1f}", flush=True)
