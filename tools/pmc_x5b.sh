#!/bin/bash
# MFMA / VALU co-execution counters of the shared-A tile kernel (through gpurun from the repo root)
set -e -o pipefail
OUT=gpurun_out/pmc_x5b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/bench_matvec.py --fmts q4 --shapes ${SHAPE:-4096x4096} --M ${M:-32} --iters 64 > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1); test -n "$f" && cp "$f" $OUT/p${i}_counters.csv
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_x5b/p*_counters.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "xdl" not in k: continue
        acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items(): print(f"   {c:32s} {sum(v)/len(v):14.1f}  (n={len(v)})")
PY
