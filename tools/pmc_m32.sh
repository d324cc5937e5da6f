#!/bin/bash
# PMC before / after of the M = 32 quantized tile kernels on one shape (through gpurun from the repo root): the per-workgroup-A
# form (ZGML_QMM_XDL5=0: qmatmul_xdl2_kernel) against the shared-A form (qmatmul_xdl5_kernel). One counter group per pass.
#   tools/pmc_m32.sh [KxN]   -> gpurun_out/pmc_m32/summary.txt
set -e -o pipefail
SHAPE=${1:-4096x22016}
OUT=gpurun_out/pmc_m32
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TA_TA_BUSY_sum"; do
    i=$((i+1))
    ZGML_QMM_XDL5=$v timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/x$v/p$i -- python3 tools/bench_matvec.py --fmts q4 --shapes $SHAPE --M 32 --iters 64 > $OUT/x${v}_p$i.log 2>&1
    f=$(find $OUT/x$v/p$i -name "*counter_collection.csv" | head -1); test -n "$f" && cp "$f" $OUT/x${v}_p${i}_counters.csv
    t=$(find $OUT/x$v/p$i -name "*kernel_trace.csv" | head -1); test -n "$t" && cp "$t" $OUT/x${v}_p${i}_trace.csv
  done
done
python3 - "$SHAPE" <<'PY' | tee gpurun_out/pmc_m32/summary.txt
import csv, glob, collections, sys
print(f"M = 32 Q4_0 tile kernels on {sys.argv[1]} (tools/pmc_m32.sh): per-launch means over the ring's launches; SQ_* cycle counters in quad-cycles")
for v, label in ((0, "ZGML_QMM_XDL5=0"), (1, "default (shared A)")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(f"gpurun_out/pmc_m32/x{v}_p*_counters.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "qmatmul_xdl" not in k: continue
            acc[k[:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(glob.glob(f"gpurun_out/pmc_m32/x{v}_p1_trace.csv")):
        for r in csv.DictReader(open(f)):
            if "qmatmul_xdl" in r["Kernel_Name"]: dur[r["Kernel_Name"][:75]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, d in acc.items():
        print(f"== {label}: {k}")
        if dur[k]: print(f"   {'duration under the profiler (us)':36s} {sum(dur[k])/len(dur[k]):14.2f}")
        for c, vals in d.items(): print(f"   {c:36s} {sum(vals)/len(vals):14.1f}  (n={len(vals)})")
PY
