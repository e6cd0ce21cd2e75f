"""Per-queue and per-kernel summary of a rocprofv3 --kernel-trace CSV of the staged LU pipeline: how busy each hardware queue is
over the run, the average duration of each kernel, and for the big-update queue the gaps between consecutive kernels."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
# the timed region: skip the first 35 % (library load, warm-up)
lo = t0 + int(0.35 * (t1 - t0))
byq = defaultdict(list)
byk = defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < lo:
        continue
    name = r["Kernel_Name"].split("(")[0][:70]
    byq[r["Queue_Id"]].append((s, e, name))
    byk[name].append(e - s)
span = (t1 - lo) / 1e6
print("window %.1f ms" % span)
for q, v in sorted(byq.items()):
    v.sort()
    busy = sum(e - s for s, e, _ in v) / 1e6
    # union of intervals (kernels of one queue can overlap)
    u, cs, ce = 0, None, None
    for s, e, _ in v:
        if cs is None or s > ce:
            if cs is not None:
                u += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    u += ce - cs
    names = defaultdict(float)
    for s, e, nme in v:
        names[nme] += (e - s) / 1e6
    top = sorted(names.items(), key=lambda kv: -kv[1])[:4]
    print("queue %s: %d kernels, sum %.1f ms, covered %.1f ms (%.2f of window); top: %s" % (q, len(v), busy, u / 1e6, u / 1e6 / span, "; ".join("%s %.1f" % (a[:40], b) for a, b in top)))
print()
print("%-72s %8s %10s %10s" % ("kernel", "calls", "avg us", "total ms"))
for name, d in sorted(byk.items(), key=lambda kv: -sum(kv[1]))[:22]:
    print("%-72s %8d %10.1f %10.1f" % (name, len(d), sum(d) / len(d) / 1e3, sum(d) / 1e6))
