#!/bin/bash
# rocprofv3 kernel trace of resident Q4_0 prefill chunk replays, aggregated per (kernel, grid): tools/prof_prefill_by_grid.sh <T> <outname> [kinds]
set -o pipefail
T=${1:-128}; NAME=${2:-prefill${T}_by_grid}; KINDS=${3:-q4}
OUT=gpurun_out/${ROUND_DIR:-r4}/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/prefill_run.py llama2-7b --T $T --reps 10 --kinds $KINDS > $OUT.log 2>&1
python3 tools/trace_by_grid.py $OUT 16 > $OUT.txt 2>&1
find $OUT -name "*.csv" -size +20M -delete
cat $OUT.txt
tail -2 $OUT.log
