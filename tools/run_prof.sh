#!/bin/bash
# GPU-box profiling recipe (run via gpurun).  Separate passes: --kernel-trace/--stats never
# together with --pmc; FETCH_SIZE and WRITE_SIZE in their own passes (TCC slot limits).
set -u
OUT=${1:-gpurun_out/prof}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/trace" -- python3 "$REPO/tools/prof_kernels.py" > "$REPO/$OUT/trace.log" 2>&1
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAVES" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$REPO/$OUT/pmc$i" -- python3 "$REPO/tools/prof_kernels.py" > "$REPO/$OUT/pmc$i.log" 2>&1
done
cd "$REPO"
python3 tools/summarize_prof.py "$OUT/trace" "$OUT"/pmc* > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete
tail -80 "$OUT/summary.txt"
