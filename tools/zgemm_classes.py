"""Diagnostic: split the trailing-update launches of a kernel trace into their roles in the block schedule.

usage: python tools/zgemm_classes.py gpurun_out/tl/trace_kernel_trace.csv [skip_fraction]
Classes (by grid shape; tiles are 64 x 64): big (K = 256 trailing update), narrow (next block column, <= 4 tile columns,
many rows), mid (<= 3 tile rows: the U12 rows of the later panels of a block), lane (look-ahead lane, <= 3 tile columns).
"""
import csv, sys, collections

path = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Stream_Id"]) if "Stream_Id" in r else 0))
t0 = min(r[1] for r in rows); t1 = max(r[2] for r in rows)
cut = t0 + skip * (t1 - t0)
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for name, s, e, gx, gy, sid in rows:
    if s < cut or "zgemm" not in name:
        continue
    if gy <= 3 and gx > 4:
        cls = "mid (M<=192)"
    elif gx <= 3:
        cls = "lane (N<=192)"
    elif gx == 4:
        cls = "narrow (N=256)"
    else:
        cls = "big"
    a = acc[cls]; a[0] += 1; a[1] += (e - s) * 1e-6; a[2] += gx * gy
tot = sum(a[1] for a in acc.values())
for cls, (cnt, ms, tiles) in sorted(acc.items()):
    print(f"{cls:16s} launches {cnt:6d}  total {ms:9.2f} ms ({100 * ms / tot:5.1f} %)  avg {1e3 * ms / cnt:8.1f} us  tiles/launch {tiles / cnt:9.1f}")
