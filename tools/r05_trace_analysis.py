"""Round 5: what do the lanes run while the update stream is idle? Reads a rocprofv3 kernel trace (csv) of bench.py."""
import csv, sys, collections
path = sys.argv[1]
rows = []
with open(path) as f:
    rd = csv.DictReader(f)
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
def short(n):
    for k in ("lu_spec_block_kernel<32, false>", "lu_spec_block_kernel<32, true>", "lu_spec_finish_kernel<32, false>", "lu_spec_finish_kernel<32, true>", "zgemm3m_dma_kernel<2, 2, true>", "zgemm3m_dma_kernel<2, 2, false>",
              "lu_lane_step2", "lu_lane_step_kernel", "lu_trsm64", "zgemv_sub", "lu_trsv", "lu_gather", "lu_scatter", "tbem_far", "tbem_near", "tbem_self", "incident", "fill_zero", "copyBuffer", "fillBuffer"):
        if k in n: return k
    return n[:40]
big = [r for r in rows if "zgemm3m_dma_kernel<2, 2, true>" in r[3]]
bq = collections.Counter(r[2] for r in big).most_common(1)[0][0]
onbig = [r for r in rows if r[2] == bq]
t0 = big[len(big) // 5][0]; t1 = big[4 * len(big) // 5][1]          # steady window
print("queues:", collections.Counter(r[2] for r in rows))
print("update queue", bq, "window ms", (t1 - t0) / 1e6)
# idle periods of the update queue inside the window
idle = []
prev = None
for r in onbig:
    if r[1] < t0 or r[0] > t1: continue
    if prev is not None and r[0] > prev: idle.append((prev, r[0]))
    prev = max(prev, r[1]) if prev else r[1]
tot_idle = sum(b - a for a, b in idle)
print("update queue idle in window: %.1f ms in %d gaps" % (tot_idle / 1e6, len(idle)))
# what runs on the other queues during those idle periods
lanes = [r for r in rows if r[2] != bq and r[1] >= t0 and r[0] <= t1]
acc = collections.Counter(); nothing = 0
import bisect
starts = [r[0] for r in lanes]
for a, b in idle:
    if b - a < 20000: continue
    covered = []
    i = bisect.bisect_left(starts, a - 5_000_000)
    for r in lanes[i:]:
        if r[0] >= b: break
        lo, hi = max(a, r[0]), min(b, r[1])
        if hi > lo: acc[short(r[3])] += hi - lo
print("kernel time on the other queues inside the update queue's idle periods (ms; three lanes overlap, so the sum can exceed the idle time):")
for k, v in acc.most_common(14): print("   %-40s %8.1f" % (k, v / 1e6))
# per-lane-queue busy fraction
for q in sorted(set(r[2] for r in lanes)):
    ql = [r for r in lanes if r[2] == q]
    busy = sum(min(r[1], t1) - max(r[0], t0) for r in ql)
    print("queue %s: %d kernels, busy %.1f ms of %.1f" % (q, len(ql), busy / 1e6, (t1 - t0) / 1e6))
    gaps = [ql[i + 1][0] - ql[i][1] for i in range(len(ql) - 1)]
    small = [g for g in gaps if 0 <= g < 30000]
    print("     launch gaps < 30 us: %d, mean %.1f us, total %.1f ms; gaps >= 30 us: %d, total %.1f ms" % (len(small), sum(small) / max(1, len(small)) / 1e3, sum(small) / 1e6, len([g for g in gaps if g >= 30000]), sum(g for g in gaps if g >= 30000) / 1e6))
