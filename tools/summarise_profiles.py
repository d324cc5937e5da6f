"""Copy the rocprofv3 summaries collected by tools/collect_profiles.sh into profiles/ (tracked) and
reduce the PMC pass to per-kernel averages + the traffic figure bench.py reports."""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = ROOT / "gpurun_out" / f"prof_{tag}", ROOT / "profiles"
for name in ("smollm135m_decode", "llama2_7b_decode", "qmatvec_bench", "qmatvec_chain", "llama2_7b_prefill32", "llama2_7b_prefill128"):
    f = src / f"{name}_kernel_stats.csv"
    if f.exists():
        shutil.copy(f, dst / f"{tag}_{name}_kernel_stats.csv")
        print("copied", f.name)
pmc = src / "pmc_counter_collection.csv"
if pmc.exists():
    agg = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(pmc)):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        k = r["Kernel_Name"]
        agg[k].append(float(r["Counter_Value"]))
        meta[k] = (r["Grid_Size"], r["Workgroup_Size"])
    out = dst / f"{tag}_qmatvec_bench_pmc_fetch_size.csv"
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Dispatches", "Avg_Counter_Value", "Min", "Max"])
        for k, v in agg.items():
            w.writerow([k, int(meta[k][0]), int(meta[k][1]), "FETCH_SIZE", len(v), round(sum(v) / len(v), 3), min(v), max(v)])
    print("wrote", out.name)
    copy = [v for k, v in agg.items() if "copy_f4" in k]
    # round 3: the Q4_0 mat-vec of the roofline shape is the K-on-lanes kernel; before: qmatvec_kernel with Q4 = true (4th template argument)
    q4 = [(k, v) for k, v in agg.items() if "qmatvec_kon_kernel<" in k]
    if not q4:
        q4 = [(k, v) for k, v in agg.items() if "qmatvec_kernel<" in k and k.split("qmatvec_kernel<")[1].split(",")[3].strip() == "true"]
    if q4 and copy:
        k, v = max(q4, key=lambda kv: len(kv[1]))
        kb = sum(v) / len(v)
        cal = sum(copy[0]) / len(copy[0])
        import subprocess
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
        js = {"source_commit": commit + " (HEAD when the summary was written; the kernel of the pass is the tree's)",
              "4096x4096_q4_0": {
            "FETCH_SIZE_KB_avg": round(kb, 3), "traffic_bytes_per_launch": int(round(kb * 1024 * 2)),
            "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams -> x2 "
                          f"(MI355X_MICROARCH.md, HBM); calibrated in the same run: the copy kernel of 1 GiB reads FETCH_SIZE = {cal:.0f} KB = 0.5 GiB",
            "write_side": "16 KB per launch (y), not collected (separate --pmc WRITE_SIZE pass)",
            "kernel": k, "source": f"profiles/{out.name}",
            "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/bench_matvec.py --shapes 4096x4096 --iters 128"}}
        (dst / f"{tag}_qmatvec_pmc.json").write_text(json.dumps(js, indent=1))
        print("traffic bytes/launch", js["4096x4096_q4_0"]["traffic_bytes_per_launch"])

# the SQ counter series of tools/pmc_matvec.sh (one --pmc pass per group): per shape and counter, dispatches and the average
ser = ROOT / "gpurun_out" / f"pmc_mv_{tag}"
if ser.exists():
    rows = []
    for f in sorted(ser.glob("*_counters.csv")):
        shape = f.name.split("_p")[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "qmatvec_kon_kernel" in r["Kernel_Name"] or "qmatvec_kernel" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            rows.append([shape, k, c, len(v), round(sum(v) / len(v), 1)])
    if rows:
        out = dst / f"{tag}_qmatvec_pmc_series.csv"
        with open(out, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["shape", "kernel", "counter", "dispatches", "avg_per_dispatch"])
            w.writerows(rows)
        print("wrote", out.name)
