"""Diagnostic: where a sweep step's time goes, from a rocprofv3 --kernel-trace CSV of `bench.py`.
Per stream (queue): kernels, busy time, gaps between consecutive kernels; per slot chain: time inside panel kernels against
time between them (the small launches and their dispatch gaps); chip level: time with at least one trailing update running.
usage: python tools/chain_analysis.py <kernel_trace.csv> <out.json> [lo_frac hi_frac]"""
import csv, sys, json, collections

rows = list(csv.DictReader(open(sys.argv[1])))
lo_f = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
hi_f = float(sys.argv[4]) if len(sys.argv) > 4 else 0.75


def short(nm):
    return nm.split("(")[0].replace("void ", "").replace("ma::", "").split("<")[0]


ev = []
for r in rows:
    q = r.get("Stream_Id") or r.get("Queue_Id")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], q, short(r["Kernel_Name"]), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
               int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
ev.sort()
T0, T1 = ev[0][0], max(e[1] for e in ev)
lo, hi = T0 + lo_f * (T1 - T0), T0 + hi_f * (T1 - T0)
win = [e for e in ev if e[0] >= lo and e[1] <= hi]
span = (hi - lo) / 1e6  # ms
out = {"window_ms": span, "kernels_in_window": len(win)}
n_far = sum(1 for e in win if e[4] == "tbem_far_kernel")
out["frequencies_in_window"] = n_far
out["ms_per_frequency_in_window"] = span / max(n_far, 1)


def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot / 1e6


per = collections.defaultdict(lambda: [0, 0.0])
for e in win:
    per[e[4]][0] += 1; per[e[4]][1] += (e[1] - e[0]) / 1e6
out["per_kernel"] = {k: {"launches_per_frequency": v[0] / max(n_far, 1), "avg_us": 1e3 * v[1] / v[0], "ms_per_frequency": v[1] / max(n_far, 1)}
                     for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}
out["chip"] = {
    "any_kernel_busy_frac": union([(e[0], e[1]) for e in win]) / span,
    "zgemm_running_frac": union([(e[0], e[1]) for e in win if "zgemm" in e[4]]) / span,
    "panel_running_frac": union([(e[0], e[1]) for e in win if "lu_panel" in e[4]]) / span,
    "far_running_frac": union([(e[0], e[1]) for e in win if "tbem_far" in e[4]]) / span,
}
# concurrency of zgemm launches: time-weighted histogram of how many run at once
pts = []
for e in win:
    if "zgemm" in e[4]:
        pts.append((e[0], 1)); pts.append((e[1], -1))
pts.sort()
hist = collections.defaultdict(float); c = 0; prev = lo
for t, d in pts:
    hist[c] += (t - prev) / 1e6; prev = t; c += d
hist[c] += (hi - prev) / 1e6
out["zgemm_concurrency_frac"] = {str(k): v / span for k, v in sorted(hist.items())}

# per stream
streams = collections.defaultdict(list)
for e in win:
    streams[e[3]].append(e)
out["streams"] = {}
for q, lst in streams.items():
    lst.sort()
    busy = sum(e[1] - e[0] for e in lst) / 1e6
    gaps = [(lst[i + 1][0] - lst[i][1]) / 1e3 for i in range(len(lst) - 1)]
    gaps_pos = sorted(g for g in gaps if g > 0)
    names = collections.Counter(e[4] for e in lst)
    d = {"kernels": len(lst), "busy_ms": busy, "busy_frac": busy / span, "kinds": dict(names.most_common(6))}
    if gaps_pos:
        d["gap_us_median"] = gaps_pos[len(gaps_pos) // 2]; d["gap_us_p90"] = gaps_pos[int(len(gaps_pos) * 0.9)]
        d["gap_ms_total"] = sum(gaps_pos) / 1e3
        d["gap_ms_total_below_200us"] = sum(g for g in gaps_pos if g < 200.0) / 1e3
    # chain structure on a slot's stream: panel kernels and what lies between them
    pidx = [i for i, e in enumerate(lst) if e[4].startswith("lu_panel")]
    if len(pidx) > 10:
        pan = sum(lst[i][1] - lst[i][0] for i in pidx) / 1e6
        between = []; between_k = []; between_busy = []
        for a, b in zip(pidx[:-1], pidx[1:]):
            between.append((lst[b][0] - lst[a][1]) / 1e3)
            between_k.append(b - a - 1)
            between_busy.append(sum(lst[i][1] - lst[i][0] for i in range(a + 1, b)) / 1e3)
        srt = sorted(between)
        d["chain"] = {"panels": len(pidx), "panel_ms": pan, "panel_avg_us": 1e3 * pan / len(pidx),
                      "between_panels_ms": sum(between) / 1e3, "between_median_us": srt[len(srt) // 2],
                      "kernels_between_avg": sum(between_k) / len(between_k),
                      "kernel_time_between_ms": sum(between_busy) / 1e3,
                      "gap_time_between_ms": (sum(between) - sum(between_busy)) / 1e3}
    out["streams"][str(q)] = d
# small kernels: launch-to-launch latency by kind (start of this kernel minus end of the previous kernel on the same stream)
lat = collections.defaultdict(list)
for q, lst in streams.items():
    for i in range(1, len(lst)):
        g = (lst[i][0] - lst[i - 1][1]) / 1e3
        if 0 < g < 500:
            lat[lst[i][4]].append(g)
out["dispatch_gap_before_kernel_us"] = {k: {"n": len(v), "median": sorted(v)[len(v) // 2], "mean": sum(v) / len(v)} for k, v in lat.items() if len(v) >= 5}
# large gaps on a stream (the stream waits for an event, for admission or for the host): by the kernel that ends the gap
big = collections.defaultdict(lambda: [0, 0.0])
for q, lst in streams.items():
    for i in range(1, len(lst)):
        g = (lst[i][0] - lst[i - 1][1]) / 1e3
        if g >= 100:
            key = "stream %s: %s after %s" % (q if str(q) == "0" else "lane", lst[i][4], lst[i - 1][4])
            big[key][0] += 1; big[key][1] += g / 1e3
out["gaps_over_100us"] = {k: {"n": v[0], "ms_per_frequency": v[1] / max(n_far, 1), "avg_us": 1e3 * v[1] / v[0]} for k, v in sorted(big.items(), key=lambda kv: -kv[1][1])[:25]}
# panel kernels: duration per column against what ran beside them
def ovl(e, kind):
    tot = 0
    for f in win:
        if f is e or kind not in f[4]: continue
        a, b = max(e[0], f[0]), min(e[1], f[1])
        if b > a: tot += b - a
    return tot / max(e[1] - e[0], 1)
pan = [e for e in win if e[4].startswith("lu_panel")]
rec = []
for e in pan[::3]:
    rec.append(((e[1] - e[0]) / 1e3, e[5] // max(e[6], 1), ovl(e, "lu_panel"), ovl(e, "zgemm"), ovl(e, "tbem_far")))
tab = collections.defaultdict(lambda: [0, 0.0])
for d, wg, op, oz, of in rec:
    key = "wgs %3d-%3d | other panels %s | zgemm %s | far %s" % (wg // 60 * 60, wg // 60 * 60 + 59, "0" if op < 0.1 else ("<0.6" if op < 0.6 else ">=0.6"),
                                                                   "<0.3" if oz < 0.3 else ("<0.8" if oz < 0.8 else ">=0.8"), "y" if of > 0.2 else "n")
    tab[key][0] += 1; tab[key][1] += d
out["panel_us_by_company"] = {k: {"n": v[0], "avg_us": v[1] / v[0]} for k, v in sorted(tab.items())}
# a stretch of one lane's timeline (start offset, duration, gap before, kernel, workgroups), mid-window
tl = []
for q, lst in streams.items():
    if str(q) == "0" or len(lst) < 500: continue
    i0 = len(lst) // 2
    while i0 < len(lst) - 130 and not lst[i0][4].startswith("lu_panel"): i0 += 1
    t0 = lst[i0][0]
    for i in range(i0, min(i0 + 120, len(lst))):
        e = lst[i]
        tl.append("%9.1f %8.1f gap %7.1f  %-24s wgs %d" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, (e[0] - lst[i - 1][1]) / 1e3, e[4], e[5] // max(e[6], 1)))
    break
out["lane_timeline"] = tl
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("window_ms", "frequencies_in_window", "ms_per_frequency_in_window", "chip", "zgemm_concurrency_frac")}, indent=1))
