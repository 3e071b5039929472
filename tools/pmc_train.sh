cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/t4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/t4/p1 -- python3 $R/tools/train_bench.py --only hip --steps 3 --warmup 2 > $R/gpurun_out/t4/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/t4/p2 -- python3 $R/tools/train_bench.py --only hip --steps 3 --warmup 2 > $R/gpurun_out/t4/p2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/t4/p1", "gpurun_out/t4/p2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(d + "_summary.txt", "w") as f:
        for n, c in sorted(agg.items()):
            if not n.startswith("aft::") and "aft" not in n: continue
            f.write(n + "  n=%d\n" % len(next(iter(c.values()))))
            for k, v in sorted(c.items()):
                f.write("    %-28s %14.0f\n" % (k, sum(v) / len(v)))
PY
rm -rf gpurun_out/t4/p1 gpurun_out/t4/p2
