#!/bin/bash
# PMC comparison of experiment libraries on K2 fast: tools/pmc_libs.sh <tag> lib1.so lib2.so ...  -> gpurun_out/pmc_<tag>.txt
set -o pipefail
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/pmc_${tag}
mkdir -p $out
for lib in "$@"; do
  n=$(basename $lib .so)
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $out/${n}_a -- python3 tools/run_k2.py $lib ${MATH:-fast} 3 > $out/${n}_a.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_IFETCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $out/${n}_b -- python3 tools/run_k2.py $lib ${MATH:-fast} 3 > $out/${n}_b.log 2>&1 || exit 1
done
python3 - "$out" "$@" > gpurun_out/pmc_${tag}.txt <<'PY'
import csv, glob, sys, os, collections, re
out = sys.argv[1]
# the S = 16 path tracer kernel of either family: pathtrace_pool_kernel<fast, 16, spheres> / pathtrace_kernel<..., 16, ...>
KERNEL = re.compile(r"pathtrace(_pool)?_kernel<[^>]*\b16\b[^>]*>")
rows = {}
for lib in sys.argv[2:]:
    n = os.path.basename(lib)[:-3]
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{n}_[ab]/*/*_counter_collection.csv"):   # (_a / _b: the two counter passes of THIS library)
        for r in csv.DictReader(open(f)):
            if KERNEL.search(r["Kernel_Name"]):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not agg:
        sys.exit(f"pmc_libs: no path tracer kernel rows for {n} (kernel renamed?)")
    rows[n] = {k: sum(v) / len(v) for k, v in agg.items()}
names = sorted({k for r in rows.values() for k in r})
print(f"{'counter':28s}" + "".join(f"{n[-14:]:>16s}" for n in rows))
for k in names:
    print(f"{k:28s}" + "".join(f"{rows[n].get(k, float('nan')):16.5g}" for n in rows))
print(f"{'cycles/SIMD (GUI/8)':28s}" + "".join(f"{rows[n].get('GRBM_GUI_ACTIVE', 0) / 8:16.5g}" for n in rows))
print(f"{'VALU per SIMD':28s}" + "".join(f"{rows[n].get('SQ_INSTS_VALU', 0) / 1024:16.5g}" for n in rows))
print(f"{'cycles per VALU inst':28s}" + "".join(f"{rows[n].get('GRBM_GUI_ACTIVE', 0) / 8 / (rows[n].get('SQ_INSTS_VALU', 1) / 1024):16.4f}" for n in rows))
PY
[ $? -eq 0 ] || exit 1
cat gpurun_out/pmc_${tag}.txt
