#!/bin/bash
# usage: run_pmc.sh OUTDIR "CTR CTR ..." ["CTR ..."]...   (one rocprofv3 --pmc pass per quoted set)
set -u
OUT=$1; shift
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$REPO/$OUT/pmc$i" -- python3 "$REPO/tools/prof_kernels.py" > "$REPO/$OUT/pmc$i.log" 2>&1
done
cd "$REPO"
python3 tools/summarize_prof.py "$OUT"/pmc* > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete
cat "$OUT/summary.txt" | grep -v rocclr
