#!/bin/bash
# GPU-side timeline of the vtable path (rocprofv3 kernel + memory-copy trace, no counters)
O=gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/vt_trace
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/vt_trace -- python3 tools/decode_vtable_run.py smollm-135m 96 dyn > $O/vt_trace.log 2>&1
tail -2 $O/vt_trace.log
python3 tools/vtable_timeline.py $O/vt_trace | tee $O/vt_timeline.txt
find $O/vt_trace -name '*_memory_copy_trace.csv' | head -1 | xargs head -3
find $O/vt_trace -name '*.csv' -size +2M -delete
