"""Diagnostic: print windows of a rocprofv3 --kernel-trace CSV as a timeline, plus per-kernel totals.
usage: python tools/kernel_timeline.py <kernel_trace.csv> [window_us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 2.6e6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ma::", "")[:30],
             r.get("Grid_Size_X", r.get("Grid_Size", ""))) for r in rows)
tot = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    tot[e[3]][0] += 1; tot[e[3]][1] += (e[1] - e[0]) / 1e3
print("kernel totals (count, total us, avg us)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-30s %6d %12.1f %9.1f" % (k, v[0], v[1], v[1] / v[0]))
p = [i for i, e in enumerate(ev) if "lu_panel" in e[3]]
for frac in (0.1, 0.6, 0.9):
    base = p[int(len(p) * frac)]
    t0 = ev[base][0]
    print("---- window at panel launch %d of %d" % (int(len(p) * frac), len(p)))
    for e in ev[base: base + 80]:
        if e[0] - t0 > win: break
        print("%9.1f us  dur %8.1f us  q%s  %-30s grid %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3], e[4]))
