#!/bin/bash
# PMC passes over the stand-alone decode attention at a long context, f32 rows against int8 rows (through gpurun from the repo root):
#   tools/pmc_attn_kvq.sh [model] [pos]   -> per-kernel means of the counters, one counter group per pass
set -e -o pipefail
MODEL=${1:-llama2-7b}; POS=${2:-1900}
OUT=gpurun_out/pmc_attn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ZGML_HIP_FUSE_QKV_ATTN_KON=0 ZGML_HIP_FUSE_QKV_ATTN=0
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/decode_kvq_run.py $MODEL 8 $POS > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1); test -n "$f" && cp "$f" $OUT/p${i}_counters.csv
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_attn/p*_counters.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attention_decode_kernel" not in k: continue
        acc[k.split("attention_decode_kernel")[1][:24]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("attention_decode_kernel" + k)
    for c, v in d.items(): print(f"   {c:28s} {sum(v)/len(v):14.1f}  (n={len(v)})")
PY
