#!/bin/bash
# PMC passes over the M <= 32 shared-A tile kernel (through gpurun from the repo root), one counter group per pass;
# prints per-kernel-name means. ENV (e.g. ZGML_QMM_XDL5_MIN_RUN=32) is inherited by the benchmark.
set -e -o pipefail
OUT=gpurun_out/pmc_x5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/bench_matvec.py --fmts q4 --shapes ${SHAPE:-4096x4096} --M ${M:-32} --iters 64 > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1); test -n "$f" && cp "$f" $OUT/p${i}_counters.csv
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_x5/p*_counters.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "xdl" not in k and "split_a" not in k: continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items(): print(f"   {c:32s} {sum(v)/len(v):14.1f}  (n={len(v)})")
PY
