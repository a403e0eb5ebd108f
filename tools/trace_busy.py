"""GPU-busy analysis of a rocprofv3 kernel trace: union of dispatch intervals, per-kernel totals, and the largest idle gaps.
    python tools/trace_busy.py <kernel_trace.csv> [t0_frac t1_frac]   (optional window as fractions of the trace)"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_lo, t_hi = ev[0][0], max(e[1] for e in ev)
if len(sys.argv) >= 4:
    a, b = float(sys.argv[2]), float(sys.argv[3])
    lo, hi = t_lo + a * (t_hi - t_lo), t_lo + b * (t_hi - t_lo)
    ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
    t_lo, t_hi = ev[0][0], max(e[1] for e in ev)
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
for s, e, n in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t_lo))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t_hi - t_lo
print(f"span {span / 1e6:.3f} ms, GPU busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %), {len(ev)} dispatches")
per = defaultdict(lambda: [0, 0])
for s, e, n in ev:
    k = n.split("(")[0].replace("void amsm::", "")[:60]
    per[k][0] += e - s
    per[k][1] += 1
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"  {k:60s} {c:6d} x {t / c / 1e3:9.1f} us = {t / 1e6:8.3f} ms")
gaps.sort(reverse=True)
print("largest idle gaps (us @ ms offset):", ", ".join(f"{g / 1e3:.0f}@{o / 1e6:.2f}" for g, o in gaps[:12]))
print(f"idle in gaps > 20 us: {sum(g for g, _ in gaps if g > 20000) / 1e6:.3f} ms; gaps <= 20 us: {sum(g for g, _ in gaps if g <= 20000) / 1e6:.3f} ms")
