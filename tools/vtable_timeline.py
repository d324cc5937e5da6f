"""GPU-side timeline of the drop-in (vtable) decode path from a rocprofv3 --kernel-trace --memory-copy-trace run of
tools/decode_vtable_run.py: per token, where the device waits for the host. Prints the median over the tokens of
  previous token's D2H end -> this token's H2D start   (host turn-around: sync wake-up, unpack, argmax, refresh, pack, submit)
  H2D start -> end, -> scatter kernel start -> end, -> first program kernel start
  first program kernel start -> last program kernel end (the token itself)
  last kernel end -> D2H start -> D2H end
usage: python tools/vtable_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv
import glob
import statistics
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"]))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", ""))))
ev.sort()
# a token = [H2D copy?] scatter_words_kernel ... kernels ... D2H copy
toks, cur = [], None
for s, e, k, n in ev:
    if k == "K" and "scatter_words_kernel" in n:
        cur = {"scatter": (s, e), "kern": [], "h2d": None, "d2h": None}
        # the H2D right in front of it
        toks.append(cur)
    elif cur is not None and k == "K":
        cur["kern"].append((s, e, n))
    elif k == "C":
        if "HOST_TO_DEVICE" in n.upper() or "H2D" in n.upper():
            last_h2d = (s, e)
            pending_h2d = (s, e)
        elif cur is not None and cur["d2h"] is None and cur["kern"]:
            cur["d2h"] = (s, e)
# attach the H2D that ended last before each scatter
h2ds = [(s, e) for s, e, k, n in ev if k == "C" and ("HOST_TO_DEVICE" in n.upper() or "H2D" in n.upper())]
for t in toks:
    c = [h for h in h2ds if h[1] <= t["scatter"][0]]
    t["h2d"] = c[-1] if c else None
rows = []
for i in range(1, len(toks)):
    t, p = toks[i], toks[i - 1]
    if not t["kern"] or not t["d2h"] or not p["d2h"]:
        continue
    first, last = t["kern"][0], max(t["kern"], key=lambda x: x[1])
    h = t["h2d"] if t["h2d"] and t["h2d"][0] >= p["d2h"][1] else None
    h0 = h[0] if h else t["scatter"][0]
    rows.append({
        "turnaround(prev D2H end -> H2D/scatter start)": h0 - p["d2h"][1],
        "H2D": (h[1] - h[0]) if h else 0,
        "H2D end -> scatter start": t["scatter"][0] - (h[1] if h else h0),
        "scatter": t["scatter"][1] - t["scatter"][0],
        "scatter end -> first kernel": first[0] - t["scatter"][1],
        "token (first kernel start -> last kernel end)": last[1] - first[0],
        "last kernel end -> D2H start": t["d2h"][0] - last[1],
        "D2H": t["d2h"][1] - t["d2h"][0],
        "period (D2H end -> D2H end)": t["d2h"][1] - p["d2h"][1],
        "kernels": len(t["kern"]),
    })
rows = rows[len(rows) // 4:]  # (warm tokens)
print(f"{len(rows)} tokens; medians in us")
for k in rows[0]:
    v = [r[k] for r in rows]
    if k == "kernels":
        print(f"  {k:52s} {statistics.median(v):9.0f}")
    else:
        print(f"  {k:52s} {statistics.median(v) / 1e3:9.2f}   (p10 {sorted(v)[len(v) // 10] / 1e3:.2f}, p90 {sorted(v)[9 * len(v) // 10] / 1e3:.2f})")
