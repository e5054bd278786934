"""Summarise a rocprofv3 --kernel-trace CSV of a concurrent run: how much of the wall time at least one NON-resident kernel was
executing, the mean number of kernels in flight, per-queue busy fractions and the gap between consecutive kernels of a queue.
usage: trace_overlap.py kernel_trace.csv [t_lo_fraction t_hi_fraction]"""
import csv, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
hi = t0 + (t1 - t0) * float(sys.argv[3]) if len(sys.argv) > 3 else t1
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
wall = hi - lo
resident = ("sc_grid_layer_kernel", "grid256_layer_kernel", "sc_small_layer_kernel")
ev = []
for s, e, n, q in rows:
    if any(x in n for x in resident):
        continue
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur, last, busy, area = 0, lo, 0, 0
for t, d in ev:
    if cur > 0:
        busy += t - last
    area += cur * (t - last)
    last = t
    cur += d
print("window %.3f s, %d kernels (%d non-resident)" % (wall / 1e9, len(rows), len(ev) // 2))
print("non-resident kernels: some kernel executing %.1f %% of the time, mean in flight %.2f" % (100.0 * busy / wall, area / wall))
perq = defaultdict(list)
for s, e, n, q in rows:
    perq[q].append((s, e, n))
gaps, qbusy = [], []
for q, l in perq.items():
    l.sort()
    b = sum(e - s for s, e, _ in l)
    qbusy.append(b / wall)
    for (s0, e0, n0), (s1, e1, n1) in zip(l, l[1:]):
        if not any(x in n0 for x in resident) and s1 > e0:
            gaps.append(s1 - e0)
gaps.sort()
print("queues %d; per-queue busy fraction (incl. resident kernels) min %.2f median %.2f max %.2f" % (len(perq), min(qbusy), sorted(qbusy)[len(qbusy) // 2], max(qbusy)))
if gaps:
    print("gap after a non-resident kernel to the next kernel of its queue: median %.1f us, p90 %.1f us, mean %.1f us" % (gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, sum(gaps) / len(gaps) / 1e3))
dur = defaultdict(lambda: [0, 0])
for s, e, n, q in rows:
    k = n.split("(")[0][-40:]
    dur[k][0] += 1; dur[k][1] += e - s
for k, (c, t) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-42s %7d calls  %8.1f ms total  %7.1f us avg" % (k, c, t / 1e6, t / c / 1e3))
