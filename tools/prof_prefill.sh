#!/bin/bash
# rocprofv3 kernel stats of resident Q4_0 prefill chunk replays: tools/prof_prefill.sh <T> <outname>
set -o pipefail
T=${1:-32}; NAME=${2:-prefill_q4_$T}
OUT=gpurun_out/r3/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/prefill_run.py llama2-7b --T $T --reps 20 --kinds q4 > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(6), f'{float(r["AverageNs"])/1e3:8.2f} us', r["Percentage"])
PY
tail -1 $OUT.log
