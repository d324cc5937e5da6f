#!/usr/bin/env python3
"""Reconcile the clocks on the 4096^2 Q4_0 mat-vec chain (tools/reconcile_chain.sh collects the inputs): rocprofv3's per-dispatch
begin / end stamps against the HIP-event figure of the same program and the in-kernel stamps of the diagnostics build.

    python tools/reconcile_chain.py gpurun_out/r04/reconcile > profiles/r04_qmatvec_chain_reconcile.txt
"""
import csv
import json
import statistics as st
import sys
from pathlib import Path

BYTES = 4096 * 4096 // 32 * 18 + 4 * 4096 + 4 * 4096


def main():
    d = Path(sys.argv[1])
    rows = []
    with open(d / "kernel_trace.csv") as f:
        for r in csv.DictReader(f):
            if "qmatvec_kon_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    # the timed region is the LAST run of >= 2048 back-to-back dispatches (warm-up replays come first); gaps > 50 us split runs
    runs, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if b[0] - a[1] > 50_000:
            runs.append(cur)
            cur = []
        cur.append(b)
    runs.append(cur)
    # every back-to-back run of the ring (one graph replay = 64 dispatches); the first two runs are warm-up replays
    runs = [r for r in runs if len(r) >= 8][2:] or runs
    dur = [e - s for r in runs for s, e in r]
    period = [b[0] - a[0] for r in runs for a, b in zip(r, r[1:])]
    gap = [b[0] - a[1] for r in runs for a, b in zip(r, r[1:])]
    span = sum((r[-1][1] - r[0][0]) for r in runs) / sum(len(r) for r in runs)
    run = [x for r in runs for x in r]

    def line(name, v):
        return f"{name:<44} mean {st.mean(v) / 1e3:7.3f}  median {st.median(v) / 1e3:7.3f}  p10 {sorted(v)[len(v) // 10] / 1e3:7.3f}  p90 {sorted(v)[len(v) * 9 // 10] / 1e3:7.3f}  (us)"

    ev_un = [json.loads(x) for x in (d / "events_unprofiled.json").read_text().splitlines() if x.startswith("{")]
    ev_pr = [json.loads(x) for x in (d / "events_profiled.json").read_text().splitlines() if x.startswith("{")]
    print("# 4096 x 4096 Q4_0 mat-vec on the data-dependent chain (tools/chain_bench.py 4096 2048): three clocks, one table")
    print(f"# algorithmic bytes per launch: {BYTES}; dispatches in back-to-back runs of the ring (graph replays, warm-up replays dropped): {len(run)} of {len(rows)} in the trace")
    print()
    print("## (a) HIP events over the timed region (cudaEvent-style, on the launch stream; ring replayed from a hipGraph)")
    for e in ev_un:
        print(f"un-profiled run : {e['us']:.3f} us per launch -> {BYTES / e['us'] / 1e3:7.1f} GB/s = {BYTES / e['us'] / 1e3 / 8000:.3f} of 8 TB/s")
    for e in ev_pr:
        print(f"UNDER rocprofv3 : {e['us']:.3f} us per launch -> {BYTES / e['us'] / 1e3:7.1f} GB/s = {BYTES / e['us'] / 1e3 / 8000:.3f} of 8 TB/s   <- the same program while every dispatch is being timestamped")
    print()
    print("## (b) rocprofv3 --kernel-trace, per dispatch of that one profiled pass")
    print(line("kernel duration  (end - begin)", dur))
    print(line("period           (begin -> next begin)", period))
    print(line("gap              (end -> next begin)", gap))
    print(f"{'whole run / dispatches':<44} {span / 1e3:7.3f} us")
    k = st.mean(dur)
    print(f"frac from the profiler's mean duration : {BYTES / k / 8000:.3f};  from the profiler's mean PERIOD: {BYTES / st.mean(period) / 8000:.3f}")
    print()
    print("## (c) in-kernel stamps (diagnostics build; the stamps add ~0.4 us per launch)")
    for ln in (d / "stamps.txt").read_text().splitlines():
        if "stamps" in ln or ln.startswith("{"):
            print(ln)
    print()
    print("## calibration streams on the same box (1 GiB; ZGML_COPY_VARIANT 0 = grid-stride loop, 1 = one float4 per thread, 2 = READ-only 8 in flight, 0x200108 = 8 in flight nt 32 blocks/CU)")
    print((d / "copy_variants.txt").read_text().rstrip())
    import re
    spans = [int(m) for m in re.findall(r"first start -> last end (\d+)", (d / "stamps.txt").read_text())]
    summary = {"stamps_span_us": round(sum(spans) / len(spans) / 1e3, 3) if spans else None, "bytes_per_launch": BYTES, "events_unprofiled_us": [e["us"] for e in ev_un], "events_profiled_us": [e["us"] for e in ev_pr],
               "rocprof_mean_duration_us": k / 1e3, "rocprof_median_duration_us": st.median(dur) / 1e3,
               "rocprof_mean_period_us": st.mean(period) / 1e3, "rocprof_mean_gap_us": st.mean(gap) / 1e3, "dispatches": len(run)}
    (d / "reconcile.json").write_text(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
