set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/abwd_pmc; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/p$i" -- python3 "$REPO/tools/train_bench.py" --only hip --steps 2 --warmup 1 > "$OUT/p$i.log" 2>&1
done
cd "$REPO"
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/abwd_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        for key in ("attn_bwd_kernel", "attn_bwd_kv_kernel", "attn_train_fwd_kernel", "chain_bwd_kernel"):
            if key in k:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}  n={len(v)}")
PY
find "$OUT" -name "*.csv" -size +1M -delete
