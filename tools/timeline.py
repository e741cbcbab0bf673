#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of bench.py: per forward, where the trunks end and how long the tail
(sum-pool ... grasp head) runs.  usage: timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# a forward starts at its first nchw3_to_nhwc4 kernel
starts = [i for i, e in enumerate(ev) if "nchw3_to_nhwc4" in e[2] and (i == 0 or "nchw3_to_nhwc4" not in ev[i - 1][2])]
starts = [s for k, s in enumerate(starts) if k % 1 == 0]
fw = []
for a, b in zip(starts, starts[1:] + [len(ev)]):
    seg = ev[a:b]
    if len(seg) < 100:
        continue
    t0 = seg[0][0]
    sp = [e for e in seg if "sumpool" in e[2]]
    if not sp:
        continue
    t_tail = sp[0][0]
    t_end = max(e[1] for e in seg)
    busy = sum(e[1] - e[0] for e in seg)
    tail_busy = sum(e[1] - e[0] for e in seg if e[0] >= t_tail)
    fw.append(((t_end - t0) / 1e6, (t_tail - t0) / 1e6, (t_end - t_tail) / 1e6, busy / (t_end - t0), tail_busy / max(1, t_end - t_tail)))
for f in fw[-4:]:
    print("forward %.2f ms: trunks %.2f ms, tail %.2f ms; mean concurrency %.2f (tail %.2f)" % f)
