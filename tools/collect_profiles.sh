#!/bin/bash
# Collect the rocprofv3 artefacts of a round on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the decode loops, the mat-vec ring and the prefill chunk, and a separate
#   --pmc FETCH_SIZE pass of the mat-vec ring. Summaries land under gpurun_out/prof_<tag>/ ; copy the
#   ones to be judged into profiles/ (tools/summarise_profiles.py does that).
set -e -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# The decode loops are profiled on the GRAPH path (what bench.py times). Round 1 ran them with ZGML_HIP_GRAPH=0 after
# rocprofv3 crashed inside hipGraphLaunch of the per-token graph; the crash did not reproduce in round 2 (current or
# rebuilt round-1 library), see DESIGN.md section 5.
run() { # name, command...
  local name=$1; shift
  echo "== $name"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1); test -n "$f" && cp "$f" $OUT/${name}_kernel_stats.csv
}
run smollm135m_decode python3 tools/decode_run.py smollm-135m 64
run llama2_7b_decode python3 tools/decode_run.py llama2-7b 32
run qmatvec_bench python3 tools/bench_matvec.py --shapes 4096x4096 --iters 512
run qmatvec_chain python3 tools/chain_bench.py 4096 1024
run llama2_7b_prefill32 python3 tools/prefill_run.py llama2-7b --T 32 --reps 6 --kinds q4,f16
run llama2_7b_prefill128 python3 tools/prefill_run.py llama2-7b --T 128 --reps 4 --kinds q4,f16
echo "== pmc"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc -- python3 tools/bench_matvec.py --shapes 4096x4096 --iters 128 > $OUT/pmc.log 2>&1
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1); test -n "$f" && cp "$f" $OUT/pmc_counter_collection.csv
ls -la $OUT/*.csv
