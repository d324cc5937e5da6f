#!/bin/bash
# one table that reconciles the three clocks on the 4096^2 Q4_0 mat-vec chain (run through gpurun from the repo root):
#   (a) HIP events over the timed region, un-profiled (what bench.py reports);
#   (b) ONE `rocprofv3 --kernel-trace` pass of the same program (placed directly after `--`): per dispatch begin / end -> kernel
#       duration, begin->begin period, end->begin gap — and the program's own event figure while it is being profiled;
#   (c) in-kernel stamps of the diagnostics build (libzgml_hip_trace.so).
# Writes gpurun_out/<tag>/reconcile/*; tools/reconcile_chain.py turns them into profiles/<tag>_qmatvec_chain_reconcile.txt.
set -e -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/$TAG/reconcile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/chain_bench.py 4096 2048 > $OUT/events_unprofiled.json
python3 tools/chain_bench.py 4096 2048 >> $OUT/events_unprofiled.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/chain_bench.py 4096 2048 > $OUT/events_profiled.json 2> $OUT/rocprof.log
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); cp "$f" $OUT/kernel_trace.csv
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv
rm -rf $OUT/trace
ZGML_HIP_LIB=zgml_amd/lib/libzgml_hip_trace.so ZGML_HIP_QMV_TRACE=1 python3 tools/chain_bench.py 4096 512 > $OUT/stamps.txt 2>&1
for v in 0 1 2 0x200108; do echo "variant $v"; ZGML_COPY_VARIANT=$v python3 -c "
import sys; sys.path.insert(0,'.')
from zgml_amd import Backend
be=Backend(0)
cp=be._lib.zgml_hip_copy_bench(be.ctx, 1<<30, 3, 20)
print(round((1 if '$v'=='2' else 2)*(1<<30)/cp/1e3,1), 'GB/s', 'read only' if '$v'=='2' else 'read+write')
"; done > $OUT/copy_variants.txt 2>&1
python3 tools/reconcile_chain.py $OUT > $OUT/reconcile.txt
cat $OUT/reconcile.txt
