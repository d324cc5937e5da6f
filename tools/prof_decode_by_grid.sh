#!/bin/bash
# rocprofv3 kernel trace of a resident decode run, aggregated per (kernel, grid): tools/prof_decode_by_grid.sh <model> <steps> <outname> [env...]
set -o pipefail
MODEL=$1; STEPS=$2; NAME=$3
OUT=gpurun_out/${ROUND_DIR:-r4}/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/decode_run.py $MODEL $STEPS > $OUT.log 2>&1
python3 tools/trace_by_grid.py $OUT 16 > $OUT.txt 2>&1
find $OUT -name "*.csv" -size +20M -delete
cat $OUT.txt
