"""Discrete-event model of the staged LU pipeline (lu_plan.hip Stage / bench.py run_pipeline): in-order streams, event dependencies,
measured kernel times. Explores what bounds a sweep step before anything is built: panel time, chain length, look-ahead depth,
number of slots, CU split. usage: python tools/lu_pipeline_model.py"""
import sys, itertools, json

N = 10000; NBLK = 256


def simulate(P=0.75, C=0.11, MW=0.55, slots=3, nsys=12, asm=4.7, rate=77e12 * 0.85, panels_per_block=4, depth2=False, cu_frac=1.0, spacing=None, verbose=False,
             backsub=1.5, mw_scales=True, order="rounds"):
    G = (N + NBLK - 1) // NBLK
    spacing = spacing if spacing is not None else max(1, (G + slots) // (slots + 1))

    def big_ms(g, part="all"):
        e = min(N, (g + 1) * NBLK); e2 = min(N, (g + 2) * NBLK); e3 = min(N, (g + 3) * NBLK)
        if part == "all":
            fl = 8.0 * (N - e) * (N - e2) * NBLK
        elif part == "rest":       # rows [e, n) x cols [e3, n)
            fl = 8.0 * (N - e) * (N - e3) * NBLK
        elif part == "n2":         # cols [e2, e3)
            fl = 8.0 * (N - e) * (e3 - e2) * NBLK
        return fl / (rate * cu_frac) * 1e3
    tasks = {}     # name -> dict(stream, dur, deps)
    order_on = {}  # stream -> [names]

    def add(name, stream, dur, deps=()):
        tasks[name] = dict(stream=stream, dur=dur, deps=[d for d in deps if d in tasks or True])
        order_on.setdefault(stream, []).append(name)
    # host enqueue order = bench.py run_pipeline (rounds)
    off = [s * spacing for s in range(slots)]
    r = 0
    while True:
        live = False; sl = []
        for s in range(slots):
            lr = r - off[s]
            if lr < 0:
                live = True; continue
            sysno, g = divmod(lr, G)
            idx = s + slots * sysno
            if idx >= nsys:
                continue
            live = True
            if g == 0:
                prev = "back_%d_%d" % (s, sysno - 1)
                add("asm_%d" % idx, "main", asm, [prev] if sysno > 0 else [])
                add("lane_%d_%d_0" % (s, sysno), "lane%d" % s, panels_per_block * (P + C), ["asm_%d" % idx])
            sl.append((s, sysno, g))
        if not live:
            break
        # mwork of every slot in the round, then the bigs (smallest first)
        for (s, sysno, g) in sl:
            rows_left = max(0, N - (g + 1) * NBLK)
            mw = MW * (0.35 + 0.65 * rows_left / N) if mw_scales else MW
            deps = ["lane_%d_%d_%d" % (s, sysno, g)]
            if not depth2:
                if g > 0: deps.append("big_%d_%d_%d" % (s, sysno, g - 1))
                add("mwork_%d_%d_%d" % (s, sysno, g), "lane%d" % s, mw, deps)
                if g + 1 < G:
                    add("lane_%d_%d_%d" % (s, sysno, g + 1), "lane%d" % s, panels_per_block * (P + C), [])
            else:
                # depth 2: mworkA (cols of block g+1) + N1, lane(g+1), then wait rest(g-1): mworkA2 + N2 + mworkB
                add("mworkA_%d_%d_%d" % (s, sysno, g), "lane%d" % s, 0.25 * mw, deps)
                if g + 1 < G:
                    add("lane_%d_%d_%d" % (s, sysno, g + 1), "lane%d" % s, panels_per_block * (P + C), [])
                d2 = ["big_%d_%d_%d" % (s, sysno, g - 1)] if g > 0 else []
                add("mwork_%d_%d_%d" % (s, sysno, g), "lane%d" % s, 0.75 * mw + big_ms(g, "n2") * 1.3, d2)
        if order == "rounds":
            sl2 = sorted(sl, key=lambda t: -t[2])
        else:
            sl2 = sl
        for (s, sysno, g) in sl2:
            if g + 1 < G:
                add("big_%d_%d_%d" % (s, sysno, g), "main", big_ms(g, "rest" if depth2 else "all"), ["mwork_%d_%d_%d" % (s, sysno, g)])
            else:
                tasks["big_%d_%d_%d" % (s, sysno, g)] = dict(stream=None, dur=0.0, deps=["mwork_%d_%d_%d" % (s, sysno, g)])
        for (s, sysno, g) in sl:
            if g == G - 1:
                add("back_%d_%d" % (s, sysno), "lane%d" % s, backsub, ["mwork_%d_%d_%d" % (s, sysno, g)])
        r += 1
    # run
    done = {}
    free = {st: 0.0 for st in order_on}
    head = {st: 0 for st in order_on}
    busy = {st: 0.0 for st in order_on}
    progress = True
    def ready(name):
        t = tasks[name]
        return all((d in done) or (d not in tasks) for d in t["deps"])
    # zero-stream tasks resolve lazily
    def resolve_virtual():
        ch = True
        while ch:
            ch = False
            for nm, t in tasks.items():
                if t["stream"] is None and nm not in done and all((d in done) or (d not in tasks) for d in t["deps"]):
                    done[nm] = max([done[d] for d in t["deps"] if d in done] + [0.0]); ch = True
    while progress:
        progress = False
        resolve_virtual()
        for st, names in order_on.items():
            while head[st] < len(names):
                nm = names[head[st]]
                if not ready(nm):
                    break
                t = tasks[nm]
                start = max([free[st]] + [done[d] for d in t["deps"] if d in done])
                done[nm] = start + t["dur"]; free[st] = done[nm]; busy[st] += t["dur"]; head[st] += 1
                progress = True
                resolve_virtual()
    total = max(done.values())
    if any(head[st] < len(order_on[st]) for st in order_on):
        return None
    return dict(ms_per_system=total / nsys, main_busy=busy["main"] / total, lane_busy=[busy["lane%d" % s] / total for s in range(slots)], total=total)


if __name__ == "__main__":
    print("round-2 schedule, panel 0.75 ms loaded (64 columns), chain 0.11, mwork 0.55:", simulate(nsys=24))
    for P in (0.75, 0.5, 0.3, 0.15):
        print(" P=%.2f" % P, {k: (round(v, 3) if not isinstance(v, list) else [round(x, 2) for x in v]) for k, v in simulate(P=P, nsys=24).items()})
    print("depth-2 look-ahead:")
    for P in (0.75, 0.5, 0.3):
        print(" P=%.2f" % P, {k: (round(v, 3) if not isinstance(v, list) else [round(x, 2) for x in v]) for k, v in simulate(P=P, nsys=24, depth2=True).items()})
    print("slots (P = 0.75):")
    for s_ in (2, 3, 4, 5, 6):
        print(" slots", s_, round(simulate(P=0.75, slots=s_, nsys=24)["ms_per_system"], 2), " depth2", round(simulate(P=0.75, slots=s_, nsys=24, depth2=True)["ms_per_system"], 2))
    print("register panel + lane step, 32 columns (8 per block): P = 0.205 loaded on masked CUs, chain 0.045, update on 216 of 256 CUs:")
    for d2 in (False, True):
        print(" depth2", d2, round(simulate(P=0.205, C=0.045, MW=0.8, panels_per_block=8, cu_frac=216 / 256.0, nsys=24, depth2=d2)["ms_per_system"], 2))
