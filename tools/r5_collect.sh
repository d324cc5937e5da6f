#!/bin/bash
# round 5: the committed profiles — reconciliation of the three clocks on the 4096^2 chain, decode traces by grid, the bench line
O=gpurun_out/r5; mkdir -p $O
bash tools/reconcile_chain.sh r5 > $O/reconcile.log 2>&1; tail -3 $O/reconcile.log
ROUND_DIR=r5 bash tools/prof_decode_by_grid.sh smollm-135m 64 smollm135m_decode_by_grid > /dev/null 2>&1; head -8 $O/smollm135m_decode_by_grid.txt
ROUND_DIR=r5 bash tools/prof_decode_by_grid.sh llama2-7b 32 llama2_7b_decode_by_grid > /dev/null 2>&1; head -10 $O/llama2_7b_decode_by_grid.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','vtable_tok_s','x_cpu_baseline','ms_per_step')})
print(d['roofline']['frac'], d['roofline']['frac_events'], d['roofline']['us_per_launch'])
print(d['roofline_smollm_decode']['frac'], d['roofline_llama7b_decode'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['fastest'], {k:v['value'] for k,v in d['cpu_baseline']['variants'].items()})
print(d['extra'].get('vtable_path_full_refresh_tok_s'), d['extra'].get('launches_per_token'))
PY
