cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out/pmc_pool; rm -rf $out; mkdir -p $out
L=vulkan-compute-tests_amd/lib/libmc_compute.so
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/a -- python3 tools/run_k2.py $L fast 2 > $out/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC --output-format csv -d $out/b -- python3 tools/run_k2.py $L fast 2 > $out/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_pool/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pathtrace" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {sum(x)/len(x):.5g}")
PY
