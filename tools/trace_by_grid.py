"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel name, grid size): mean / min duration and call count.
Separates the q/k/v, gate/up, down and LM-head launches that share one kernel instantiation."""
import csv
import collections
import glob
import sys

agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.replace("zgml::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0] if not short.startswith("(") else short
        agg[(short[:88], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':90s} {'grid':>9s} {'wg':>5s} {'calls':>6s} {'mean_us':>8s} {'min_us':>8s} {'share':>6s}")
for (name, grid, wg), v in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{name:90s} {grid:>9s} {wg:>5s} {len(v):6d} {sum(v) / len(v) / 1e3:8.2f} {min(v) / 1e3:8.2f} {100 * sum(v) / tot:5.1f}%")
