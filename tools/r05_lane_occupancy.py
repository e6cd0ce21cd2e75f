"""Round 5: how full are the lanes? Reads a rocprofv3 kernel trace (csv) of bench.py; per hardware queue inside the steady window: kernel
time, gaps between consecutive kernels of the queue (by size class), launches. usage: python tools/r05_lane_occupancy.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
big = [r for r in rows if "zgemm3m_dma_kernel<2, 2, true>" in r[3]]
t0 = big[len(big) // 5][0]; t1 = big[4 * len(big) // 5][1]
win = (t1 - t0) / 1e6
nsys = sum(1 for r in big if t0 <= r[0] <= t1) / 25.0
print("steady window %.1f ms = %.1f systems (%.2f ms per system)" % (win, nsys, win / nsys))
byq = collections.defaultdict(list)
for r in rows:
    if r[0] >= t0 and r[1] <= t1: byq[r[2]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(b - a for a, b, _, _ in rs) / 1e6
    gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
    g_small = sum(g for g in gaps if 0 < g <= 20000) / 1e6; n_small = sum(1 for g in gaps if 0 < g <= 20000)
    g_mid = sum(g for g in gaps if 20000 < g <= 200000) / 1e6; n_mid = sum(1 for g in gaps if 20000 < g <= 200000)
    g_big = sum(g for g in gaps if g > 200000) / 1e6; n_big = sum(1 for g in gaps if g > 200000)
    ov = sum(-g for g in gaps if g < 0) / 1e6
    print("queue %s: %6d kernels, busy %.1f%% | gaps <=20us %.1f%% (%d, mean %.1f us) | 20-200us %.1f%% (%d) | >200us %.1f%% (%d) | overlap %.1f%%" % (
        q, len(rs), 100 * busy / win, 100 * g_small / win, n_small, 1e3 * g_small / max(n_small, 1), 100 * g_mid / win, n_mid, 100 * g_big / win, n_big, 100 * ov / win))
    top = collections.Counter()
    for a, b, _, n in rs: top[n.split("(")[0][-40:]] += (b - a)
    print("     ", ", ".join("%s %.1f%%" % (k, 100 * v / 1e6 / win) for k, v in top.most_common(5)))
