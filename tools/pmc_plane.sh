set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r03; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export AFT_REPS=10 AFT_FWD=1 AFT_ONLY=encoder_plane
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_plane_fetch" -- python3 "$REPO/tools/prof_kernels.py" > "$OUT/pmc_plane_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_plane_write" -- python3 "$REPO/tools/prof_kernels.py" > "$OUT/pmc_plane_write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d "$OUT/pmc_plane_sq" -- python3 "$REPO/tools/prof_kernels.py" > "$OUT/pmc_plane_sq.log" 2>&1
cd "$REPO"; python3 tools/summarize_prof.py "$OUT"/pmc_plane_* 2>&1 | grep -A12 "encoder_plane" | head -40
find "$OUT" -name "*.csv" -size +1M -delete
