#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 --kernel-trace csv of the multi-stream bench.

For the steady-state window (last `--frac` of the trace) prints: wall span, GPU-busy union,
average number of kernels in flight, and per kernel: calls, average duration, workgroups,
LDS, and its share of 'CU-time' (duration x min(workgroups, 256) / 256) -- the quantity the
whole-job throughput is bound by once enough independent compressions are in flight."""
import argparse
import csv
import collections
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--frac", type=float, default=0.5)
    ap.add_argument("--top", type=int, default=28)
    a = ap.parse_args()
    rows = [r for r in csv.DictReader(open(a.trace)) if r["Kernel_Name"].startswith(("void rc::", "rc::", "(anonymous"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
    lo = t1 - (t1 - t0) * a.frac
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= lo]
    span = max(int(r["End_Timestamp"]) for r in rows) - int(rows[0]["Start_Timestamp"])
    ev = []
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), 1))
        ev.append((int(r["End_Timestamp"]), -1))
    ev.sort()
    busy = 0
    inflight_ns = 0
    depth = 0
    last = ev[0][0]
    for t, d in ev:
        if depth > 0:
            busy += t - last
        inflight_ns += depth * (t - last)
        depth += d
        last = t
    agg = collections.OrderedDict()
    for r in rows:
        wgs = 1
        for ax in "XYZ":
            wgs *= max(1, int(r[f"Grid_Size_{ax}"]) // max(1, int(r[f"Workgroup_Size_{ax}"])))
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void rc::", "").replace("rc::", ""))
        key = (name, wgs)
        e = agg.setdefault(key, [0, 0, 0.0, int(r["LDS_Block_Size"]), int(r["Workgroup_Size_X"])])
        e[0] += 1
        e[1] += dur
        e[2] += dur * min(wgs, 256) / 256.0
    tot_cu = sum(e[2] for e in agg.values())
    print(f"window {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms ({busy/span:.2%})  mean kernels in flight {inflight_ns/span:.2f}  "
          f"sum CU-time {tot_cu/1e6:.2f} ms ({tot_cu/span:.2%} of the window)")
    print(f"{'kernel':<58}{'wgs':>7}{'thr':>6}{'lds':>8}{'calls':>7}{'avg us':>10}{'CU-time %':>11}")
    for (name, wgs), e in sorted(agg.items(), key=lambda kv: -kv[1][2])[: a.top]:
        print(f"{name[:57]:<58}{wgs:>7}{e[4]:>6}{e[3]:>8}{e[0]:>7}{e[1]/e[0]/1e3:>10.1f}{100*e[2]/tot_cu:>10.1f}%")


if __name__ == "__main__":
    main()
