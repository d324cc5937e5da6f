#!/bin/bash
# PMC passes over the dense f16 M = 32 tile kernel (run through gpurun from the repo root); one counter group per pass
set -e -o pipefail
OUT=gpurun_out/r4/pmc_f16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/bench_matvec.py --fmts ${FMT:-f16} --shapes ${SHAPE:-4096x22016} --M ${M:-32} --iters 64 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1); test -n "$f" && cp "$f" $OUT/p${i}_counters.csv
  rm -rf $OUT/p$i
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*_counters.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("zgml::(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if not any(s in k for s in ("dense_f16_tile", "qmatvec_kon", "qmatmul_xdl")): continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n {len(v)}")
PY
