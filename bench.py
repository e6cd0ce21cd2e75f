#!/usr/bin/env python3
"""bench.py — BEM frequency-sweep throughput on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2]): S10 = UV sphere r = 0.1 m,
n_theta = 51, n_phi = 100 -> 10 000 Tri3 panels; frequencies taken from the 64 log-spaced
points 100 Hz .. 8 kHz; rigid BC, beta = 4i/k (BemSolver default), plane wave +z.
One step = one frequency: TBEM assembly (far + near + self kernels) + incident RHS + dense
complex LU solve, everything resident in HBM (geometry and the near-pair plan are uploaded once
before the timed region; nothing crosses PCIe inside it).

  python bench.py --gpus N --steps K --warmup W
      N > 1 without WORLD_SIZE in the environment: bench.py starts its own N ranks (one per GPU) with
      torch.distributed.run before anything touches a GPU, and exits with their status;
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
      (the driver's form: RANK / LOCAL_RANK / WORLD_SIZE come from the environment).

Frequencies shard over ranks (rank r takes points r, r+N, ...): no data-path collective;
"scaling": "weak" (K frequencies per GPU whatever N is). Rank 0 prints ONE JSON line.
PyTorch is plumbing here: device buffers, the stream handle and torch.distributed.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C_SOUND = 343.0
RADIUS = 0.1
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6       # MI355X datasheet FP64 matrix (= FP64 vector) peak, dense


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/*_pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate runs, FETCH doubled as the gfx950 guide prescribes). PMC
    counters cannot be read from inside the timed run; null if no profile has been committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            k = json.load(open(f))["kernels"].get(kernel)
            if k:
                best = k["traffic_bytes_per_launch"]
        except Exception:
            pass
    return best


def big_update_line(n, G, seconds):
    """the K = kb x 64 updates on the caller's stream alone (the launches on the look-ahead lanes are K = 32 / 64 and run beside
    them): algorithmic flops of the staged schedule's big updates / their summed time"""
    Q = (n + 63) // 64
    kb = (Q + G - 1) // G
    fl, cnt = 0.0, 0
    for g in range(G):
        a0 = g * kb * 64
        e = min(n, a0 + kb * 64)
        enext = min(n, e + kb * 64)
        if n - e > 0 and n - enext > 0:
            fl += 8.0 * (n - e) * (n - enext) * (e - a0); cnt += 1
    return {"launches_per_step": cnt, "K": kb * 64, "algorithmic_flops_per_step": fl, "ms_per_step": seconds * 1e3,
            "achieved": fl / seconds / 1e12, "unit": "TFLOP/s", "frac": fl / seconds / 1e12 / FP64_MFMA_PEAK_TF}


def lu_flops(n):
    return (8.0 / 3.0) * n ** 3 + 8.0 * n ** 2          # zgetrf + zgetrs, real flops


def gemm_flops(n, nb=128):
    """Real flops of the trailing updates A22 -= L21 U12 of a right-looking LU with panel width nb."""
    f, k0 = 0.0, 0
    while k0 < n:
        w = min(nb, n - k0)
        r = n - k0 - w
        f += 8.0 * w * r * r
        k0 += w
    return f


def cpu_baseline(n_theta, n_phi, freq, seconds_target=12.0):
    """Reference algorithm on this host's cores: the C restatement (oracle, kind "port") assembles a
    bounded strip of rows of the SAME mesh at one sweep frequency with every core (rows over threads =
    rayon par_iter over rows); the dense solve is timed with the host LAPACK (zgesv family via SciPy /
    NumPy = what lu_solve calls, lu.rs:145) on a smaller system and scaled by the flop count."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = os.cpu_count() or 1
    om = O.uv_sphere(RADIUS, n_theta, n_phi)
    n = om.n_elem
    k = O.wave_number(freq, C_SOUND)
    beta = O.beta_scaled(k, 4.0)
    A = np.zeros((n, n), dtype=np.complex128)
    rhs = np.zeros(n, dtype=np.complex128)
    rows = min(n, 16 * cores)
    t0 = time.perf_counter()
    O.build_tbem_system_with_beta(om, k, beta, nthreads=cores, rows=(0, rows), A=A, rhs=rhs)
    t_probe = time.perf_counter() - t0
    rate = rows * n / t_probe
    rows2 = int(min(n, max(rows, rate * seconds_target * 0.6 / n)))
    if rows2 > rows:
        t0 = time.perf_counter()
        O.build_tbem_system_with_beta(om, k, beta, nthreads=cores, rows=(0, rows2), A=A, rhs=rhs)
        t_probe = time.perf_counter() - t0
        rows = rows2
    asm_pairs_per_s = rows * n / t_probe
    # the reference assembles on ONE thread (tbem.rs:96-222 is a serial double loop): the same strip kernel on one core
    rows1 = max(4, min(64, int(asm_pairs_per_s / cores * 3.0 / n)))
    t0 = time.perf_counter()
    O.build_tbem_system_with_beta(om, k, beta, nthreads=1, rows=(0, rows1), A=A, rhs=rhs)
    asm_1thread = rows1 * n / (time.perf_counter() - t0)
    del A
    # dense solve: host LAPACK (zgetrf + zgetrs) on a system of the workload's OWN size (n = 10 000: 2.67 TFLOP) unless a probe at
    # n = 2000 says that would take more than 90 s, in which case the largest size that fits ~30 s is timed and scaled by flops
    def lapack(ns_):
        rng = np.random.default_rng(0)
        M = rng.standard_normal((ns_, ns_)) + 1j * rng.standard_normal((ns_, ns_)); M[np.arange(ns_), np.arange(ns_)] += ns_
        b = rng.standard_normal(ns_) + 0j
        try:
            import scipy.linalg as sl
            t0_ = time.perf_counter(); lu_, piv_ = sl.lu_factor(M, check_finite=False, overwrite_a=True); sl.lu_solve((lu_, piv_), b, check_finite=False)
            return time.perf_counter() - t0_, "scipy.linalg.lu_factor/lu_solve (LAPACK zgetrf/zgetrs)"
        except Exception:
            t0_ = time.perf_counter(); np.linalg.solve(M, b)
            return time.perf_counter() - t0_, "numpy.linalg.solve (LAPACK zgesv)"
    t_probe_lu, lib = lapack(min(n, 2000))
    predicted = t_probe_lu * lu_flops(n) / lu_flops(min(n, 2000))
    ns = n if predicted <= 90.0 else int(max(2000, min(n, 2000 * (30.0 / t_probe_lu) ** (1.0 / 3.0))))
    t_lu, lib = lapack(ns) if ns > 2000 else (t_probe_lu, lib)
    solve_gflops = lu_flops(ns) / t_lu / 1e9
    t_step = n * n / asm_pairs_per_s + lu_flops(n) / (solve_gflops * 1e9)
    return {
        "value": n * n / t_step, "unit": "panel-pairs/s", "cores": cores, "kind": "port",
        "sample": "oracle C restatement of build_tbem_system_with_beta: %d of %d rows at %.0f Hz on %d threads (%.1f s) -> %.3e pairs/s; "
                  "dense solve: %s, n=%d%s, %.2f s -> %.1f GFLOP/s; step time = N^2/asm + ((8/3)N^3+8N^2)/solve" % (
                      rows, n, freq, cores, t_probe, asm_pairs_per_s, lib, ns, " (the workload's own size)" if ns == n else " (scaled by flops to n=%d)" % n, t_lu, solve_gflops),
        "solve_n": ns,
        "assembly_pairs_per_s": asm_pairs_per_s, "solve_gflops": solve_gflops,
        "assembly_pairs_per_s_one_thread": asm_1thread,
    }


def fem_measure(steps, warmup, fem_n, cpu=True):
    """BASELINE.json configs[3]: F1M = box 5 x 4 x 2.5 m, n^3 nodes P1 Kuhn tets,
    k = 2 pi 100 / 343: `steps` x 10 CSR SpMVs, Jacobi (omega 0.8) and l1-Jacobi sweeps, device-resident.
    HBM-bound: achieved = algorithmic bytes (nnz 20 B + N 36 B per SpMV, + N 64 B per smoother update) / time.
    Returns (kernels, roofline, config, cpu_baseline or None)."""
    import torch
    import math_audio_amd as ma
    from math_audio_amd import fem
    dev = torch.device("cuda", torch.cuda.current_device())
    m = fem_n - 1
    t0 = time.perf_counter()
    nodes, rp, ci, K, M = fem.helmholtz_box(m, m, m)
    n, nnz = len(rp) - 1, len(ci)
    t_gen = time.perf_counter() - t0
    op = ma.CsrOperator(rp, ci, K=K, M=M, device=torch.cuda.current_device())
    k = 2.0 * math.pi * 100.0 / C_SOUND
    op.set_wavenumber(complex(k, 0.01))
    i = torch.arange(n, dtype=torch.float64, device=dev)
    x = torch.complex(torch.sin(0.1 * i), torch.cos(0.2 * i)); b = torch.ones(n, dtype=torch.complex128, device=dev)
    y = torch.empty_like(x); tmp = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    byts = {"spmv": nnz * 20.0 + n * 36.0, "jacobi": nnz * 20.0 + n * 100.0, "l1_jacobi": nnz * 20.0 + n * 92.0}
    for name in ("spmv", "jacobi", "l1_jacobi"):
        def run(reps):
            if name == "spmv":
                for _ in range(reps):
                    op.spmv_dev(x.data_ptr(), y.data_ptr(), st)
            elif name == "jacobi":
                op.jacobi_dev(y.data_ptr(), b.data_ptr(), 0.8, reps, tmp.data_ptr(), st)
            else:
                op.l1_jacobi_dev(y.data_ptr(), b.data_ptr(), reps, tmp.data_ptr(), st)
        y.copy_(x); run(max(2, warmup * 2)); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        reps = steps * 10
        e0.record(); run(reps); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[name] = {"ms": ms, "GB/s": byts[name] / (ms * 1e-3) / 1e9, "frac_of_8TBs": byts[name] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # The same kernels with the working set ROTATED over three operators and vector pairs (3 x 331 MB + vectors > the 256 MiB Infinity
    # Cache): inside a V-cycle or a Krylov step other levels and vectors pass through the cache between two passes over one operator,
    # so the single-matrix figures above (the matrix partly served from the Infinity Cache) flatter the kernels.
    ops3 = [op] + [ma.CsrOperator(rp, ci, K=K, M=M, device=torch.cuda.current_device()) for _ in range(2)]
    for o3 in ops3[1:]:
        o3.set_wavenumber(complex(k, 0.01))
    xs3 = [x, x.clone(), x.clone()]; ys3 = [y, torch.empty_like(x), torch.empty_like(x)]; ts3 = [tmp, torch.empty_like(x), torch.empty_like(x)]
    def rot(name, q):
        o3, xx, yy, tt = ops3[q % 3], xs3[q % 3], ys3[q % 3], ts3[q % 3]
        if name == "spmv":
            o3.spmv_dev(xx.data_ptr(), yy.data_ptr(), st)
        else:
            o3.jacobi_dev(yy.data_ptr(), b.data_ptr(), 0.8, 1, tt.data_ptr(), st)
    for name in ("spmv", "jacobi"):
        for q in range(6):
            rot(name, q)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        reps3 = 3 * max(10, steps * 3)
        e0.record()
        for q in range(reps3):
            rot(name, q)
        e1.record(); torch.cuda.synchronize()
        ms3 = e0.elapsed_time(e1) / reps3
        res[name + "_rotating_3_operators"] = {"ms": ms3, "GB/s": byts[name] / (ms3 * 1e-3) / 1e9, "frac_of_8TBs": byts[name] / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               "working_set_MB": 3 * (byts[name] / 1e6)}
    for o3 in ops3[1:]:
        o3.close()
    # symmetric Gauss-Seidel (amg.rs:932-978; the FEM smoother's default family): the sequential sweep's result by dependency levels,
    # ONE persistent launch per direction in which a row's new value is its own flag (csr_gs_flags_kernel): chain-bound, not HBM-bound
    y.copy_(x); op.sym_gauss_seidel_dev(y.data_ptr(), b.data_ptr(), 1, st); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); op.sym_gauss_seidel_dev(y.data_ptr(), b.data_ptr(), 2, st); e1.record(); torch.cuda.synchronize()
    lev = op.gauss_seidel_levels()
    res["sym_gauss_seidel"] = {"ms": e0.elapsed_time(e1) / 2, "levels_forward_backward": list(lev),
                               "GB/s": 2 * (nnz * 20.0 + n * 68.0) / (e0.elapsed_time(e1) / 2 * 1e-3) / 1e9}
    config = {"workload": "F1M-family box 5x4x2.5 m, %d^3 nodes, P1 Kuhn tets: N=%d, nnz=%d; A = K - k^2 M fused; k = 2 pi 100/343 + 0.01i" % (fem_n, n, nnz),
              "host_generation_s": t_gen}
    roof = {"kernel": "sell_rows_kernel (SpMV, sliced-ELLPACK copy of the CSR operator, 16-bit relative columns)", "bound": "hbm", "achieved": res["spmv"]["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": res["spmv"]["frac_of_8TBs"], "traffic": pmc_traffic("ma::sell_rows_kernel<true, 0, true>"),
            "algorithmic_bytes_per_launch": byts["spmv"],
            "achieved_rotating": res["spmv_rotating_3_operators"]["GB/s"], "frac_rotating": res["spmv_rotating_3_operators"]["frac_of_8TBs"],
            "jacobi_achieved_rotating": res["jacobi_rotating_3_operators"]["GB/s"], "jacobi_frac_rotating": res["jacobi_rotating_3_operators"]["frac_of_8TBs"],
            "note": "achieved / frac: one operator applied back to back (its 331 MB partly live in the 256 MiB Infinity Cache); *_rotating: three operators "
                    "and vector pairs in rotation (1 GB working set), the figure to expect inside a V-cycle or a Krylov iteration"}
    cpu_b = None
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        cores = os.cpu_count() or 1
        vals = O.helmholtz_values(K, M, complex(k, 0.01)); xh = x.cpu().numpy()
        O.csr_matvec(rp, ci, vals, xh, nthreads=cores)
        t0 = time.perf_counter(); reps = 5
        for _ in range(reps):
            O.csr_matvec(rp, ci, vals, xh, nthreads=cores)
        tc = (time.perf_counter() - t0) / reps
        cpu_b = {"value": byts["spmv"] / tc / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
                 "sample": "oracle row-parallel CSR matvec (csr.rs:273-292), %d threads, %d reps of the same matrix" % (cores, reps)}
    op.close()
    return res, roof, config, cpu_b


def fem_workload(args):
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X")
    res, roof, config, cpu_b = fem_measure(args.steps, args.warmup, args.fem_n, cpu=not args.no_cpu_baseline)
    out = {"metric": "fem_csr_spmv_gbs", "value": res["spmv"]["GB/s"], "unit": "GB/s", "n_gpus": 1, "steps": args.steps * 10, "warmup": args.warmup,
           "ms_per_step": res["spmv"]["ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 (complex128 x, real K/M)",
           "data": "synthetic", "config": config, "kernels": res, "roofline": roof}
    if cpu_b:
        out["cpu_baseline"] = cpu_b
    print(json.dumps(out))


def config5_measure():
    """BASELINE.json configs[4] on ONE GPU: the 50 172-panel closed box 0.30 x 0.40 x 0.60 m at 1 kHz, one apply
    of each dense-free operator the reference offers -- matrix-free TBEM (13-point rule recomputed per apply), the single-level FMM
    operator (slfmm.rs) and the multi-level one (mlfmm.rs) -- with algorithmic work and bytes. Build times are host + device wall time."""
    import torch
    import math_audio_amd as ma
    from math_audio_amd import mesh as mm
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fmm_clusters import grid_clusters                    # input builder of build_slfmm_system (clusters are given to it), plain numpy
    dev = torch.device("cuda", torch.cuda.current_device())
    m = mm.generate_box_mesh(0.30, 0.40, 0.60, 46, 61, 91)
    n = m.n_elem
    k = mm.wave_number(1000.0); beta = mm.burton_miller_beta_scaled(k, 4.0)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.ones(n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
    plan = ma.BemPlan(m, device=torch.cuda.current_device())
    out = {"workload": "closed box 0.30 x 0.40 x 0.60 m, 46 x 61 x 91 cells -> %d Tri3 panels, 1 kHz, one GPU" % n, "panels": n}

    def timed(fn, reps):
        fn(x.data_ptr(), y.data_ptr(), st); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn(x.data_ptr(), y.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def near_entries(sizes, near_ptr, near_idx, nc):
        nb = 0
        for c in range(nc):
            nb += int(sizes[c]) ** 2 + sum(int(sizes[c]) * int(sizes[j]) for j in near_idx[near_ptr[c]:near_ptr[c + 1]] if j > c)
        return nb
    op = ma.LinearOperator.tbem(plan, k, beta)
    ms = timed(op.apply_dev, 2)
    out["matrix_free_tbem"] = {"apply_ms": ms, "pairs_per_s": n * n / (ms * 1e-3), "algorithmic_flops": 1.2e3 * n * n, "fp64_valu_tflops_equiv": 1.2e3 * n * n / (ms * 1e-3) / 1e12,
                               "algorithmic_bytes": 32.0 * n, "note": "the 13-point rule of every pair recomputed per apply (about 1.2 kflop per pair, SURVEY 8d K5); FP64-VALU-bound"}
    op.close()
    t0 = time.perf_counter(); cl = grid_clusters(m.center, 0.05); t_cl = time.perf_counter() - t0
    sizes = np.diff(cl.elem_ptr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    op = ma.LinearOperator.slfmm(plan, cl, k, 8, 16, 6)
    torch.cuda.synchronize(); t_build = time.perf_counter() - t0
    ms = timed(op.apply_dev, 20)
    nb = near_entries(sizes, cl.near_ptr, cl.near_idx, cl.n)
    out["slfmm"] = {"clusters": cl.n, "sphere_points": 128, "near_entries": nb, "algorithmic_bytes": nb * 16.0, "cluster_build_host_s": t_cl, "operator_build_s": t_build,
                    "apply_ms": ms, "apply_near_GBs": nb * 16.0 / (ms * 1e-3) / 1e9, "frac_of_8TBs": nb * 16.0 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "note": "SlfmmSystem::matvec; algorithmic bytes = the stored near blocks read once (16 B per entry); T / D / S stages ride on top"}
    op.close()
    t0 = time.perf_counter(); tree = ma.ClusterTree(m, 64, k); t_tree = time.perf_counter() - t0
    leaf = tree.level(tree.num_levels() - 1)
    nbm = near_entries(np.diff(leaf["elem_ptr"]), leaf["near_ptr"], leaf["near_idx"], leaf["n_clusters"])
    try:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        op = ma.LinearOperator.mlfmm(plan, tree, k)
        torch.cuda.synchronize(); t_build = time.perf_counter() - t0
        ms = timed(op.apply_dev, 10)
        out["mlfmm"] = {"levels": tree.num_levels(), "leaves": leaf["n_clusters"], "near_entries": nbm, "algorithmic_bytes": nbm * 16.0, "tree_host_s": t_tree, "operator_build_s": t_build,
                        "apply_ms": ms, "apply_near_GBs": nbm * 16.0 / (ms * 1e-3) / 1e9, "frac_of_8TBs": nbm * 16.0 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "note": "MlfmmSystem::matvec on the tree of build_cluster_tree (64 elements per leaf)"}
        op.close()
    except ma.MaError as e:
        out["mlfmm"] = {"refused": str(e)}
    out["finite"] = bool(torch.isfinite(torch.view_as_real(y)).all())
    return out


def inlib_sweep(args):
    """The sweep as ONE call into the library per phase (ma_bem_solve_sweep_multi_timed): frequency i of the list belongs to device
    i mod N; device d therefore solves the frequencies the rank form gives rank d (freqs[(d + s N) mod 64], s = 0..steps-1).
    Warm-up = one call with `warmup` frequencies per device; the timed call carries `steps` per device. value = all pairs / the
    slowest device's sweep time (plan creation -- geometry upload, near list -- is per call and reported next to it)."""
    import math_audio_amd as ma
    from math_audio_amd import mesh as mm
    N = args.gpus
    devices = [int(t) for t in args.devices.split(",")] if args.devices else list(range(N))
    if len(devices) != N:
        raise SystemExit("bench.py --inlib: --devices lists %d devices, --gpus says %d" % (len(devices), N))
    mesh = mm.generate_sphere_mesh(RADIUS, args.n_theta, args.n_phi)
    n = mesh.n_elem
    freqs = mm.log_space(100.0, 8000.0, 64)

    def flist(first, count):
        return [freqs[(d + (first + s_) * N) % len(freqs)] for s_ in range(count) for d in range(N)]
    S = max(1, min(args.slots, 4))
    if args.warmup > 0:
        ma.solve_sweep_multi_timed(mesh, devices, flist(0, args.warmup), speed_of_sound=C_SOUND, beta_scale=4.0, slots=S)
    t0 = time.perf_counter()
    X, st, secs, setup, cnt = ma.solve_sweep_multi_timed(mesh, devices, flist(args.warmup, args.steps), speed_of_sound=C_SOUND, beta_scale=4.0, slots=S)
    wall = time.perf_counter() - t0
    if not np.all(st == 0):
        raise SystemExit("a frequency of the sweep failed: status %s" % sorted(set(int(v) for v in st)))
    if not np.all(np.isfinite(X.view(np.float64))):
        raise SystemExit("non-finite solution")
    if any(int(c) != args.steps for c in cnt):
        raise SystemExit("device frequency counts %s, expected %d each" % (list(map(int, cnt)), args.steps))
    elapsed = float(max(secs))
    K = args.steps
    out = {"metric": "bem_sweep_panel_pairs_per_s", "value": float(n) * n * K * N / elapsed, "unit": "panel-pairs/s", "n_gpus": N, "steps": K, "warmup": args.warmup,
           "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
           "config": {"workload": "S10 UV-sphere r=0.1 n_theta=%d n_phi=%d -> %d Tri3 panels; 64 log-spaced frequencies 100 Hz-8 kHz, frequency i on device i mod N; "
                                  "one process, ma_bem_solve_sweep_multi (a host thread per device inside the library)" % (args.n_theta, args.n_phi, n),
                      "panels": n, "frequencies_per_gpu": K, "frequencies_in_flight_per_gpu": S, "mode": "inlib", "devices": devices,
                      "sharding": "frequency sweep, no data-path collective"},
           "per_device_ms_per_step": [float(v) / K * 1e3 for v in secs], "per_device_frequencies": [int(c) for c in cnt],
           "per_device_plan_setup_s": [float(v) for v in setup], "wall_s_of_the_call": wall,
           "note": "value uses the slowest device's sweep time (solutions copied back to the host included); the call's wall time also holds each device's plan creation"}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["bem", "fem"], default="bem", help="bem = the headline sweep (default); fem = CSR SpMV / smoother bandwidth")
    ap.add_argument("--fem-n", type=int, default=100, help="nodes per box edge for --workload fem")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-theta", type=int, default=51)
    ap.add_argument("--n-phi", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-timing", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--inlib", action="store_true", help="ONE process: ma_bem_solve_sweep_multi over --gpus N devices (a host thread per device inside the library: what a "
                                                          "Rust caller gets) instead of one torch.distributed rank per GPU")
    ap.add_argument("--devices", default="", help="--inlib: comma-separated device list instead of 0..N-1 (test hook MA_TEST_ALLOW_DUPLICATE_DEVICES=1 allows repeats)")
    ap.add_argument("--no-extras", action="store_true", help="skip the passes after the timed region that put configs #4 (FEM SpMV / smoother) and #5 (50k-panel operators) into the line")
    ap.add_argument("--schedule", choices=["auto", "pipeline", "batch"], default="auto",
                    help="pipeline: slots at staggered block indices (staged plan API); batch: lock-step batches of --slots systems; "
                         "auto: pipeline from 12 steps on (below that its fill and drain cost more than the lock step does)")
    ap.add_argument("--slots", type=int, default=int(os.environ.get("MA_SWEEP_SLOTS", "3")),
                    help="frequencies in flight per GPU (systems factored as one interleaved batch, 1..4; with --group-size g: a multiple of g, up to 8)")
    ap.add_argument("--group-size", type=int, default=int(os.environ.get("MA_SWEEP_GROUP", "0")),
                    help="pipeline schedule: slots in groups of this size share one panel kernel per panel and move in lock step (0 = every slot on its own)")
    args = ap.parse_args()
    if args.schedule == "auto":
        args.schedule = "pipeline" if args.steps >= 6 else "batch"   # 6 / 9 / 12 / 20 steps: 54.6 / 52.5 / 51.5 / 50.1 ms staged against 56.2 / 56.8 / 57.0 / 57.8 in lock step (3 steps: 61.8 against 55.9)
    if args.workload == "fem":
        return fem_workload(args)
    if args.inlib:
        return inlib_sweep(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks here, as child processes, BEFORE this process touches a GPU
        # (nothing GPU-related has been imported yet), and hand their status back. One rank per GPU over RCCL, rendezvous on
        # 127.0.0.1; the ranks re-enter this file with RANK / LOCAL_RANK / WORLD_SIZE set.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    import math_audio_amd as ma
    from math_audio_amd import mesh as mm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (one rank per GPU; start it as `python bench.py --gpus N` or with torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False and there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    mesh = mm.generate_sphere_mesh(RADIUS, args.n_theta, args.n_phi)
    n = mesh.n_elem
    freqs = mm.log_space(100.0, 8000.0, 64)
    # Frequencies are independent, and 288 GB of HBM holds many 1.6 GB systems: S systems are kept in flight and
    # factored as ONE interleaved batch, so one frequency's latency-bound panel factorisation (one chip-wide
    # gather per column) runs underneath another's MFMA-bound trailing update. A step is still one frequency.
    gsz = args.group_size if (args.group_size >= 2 and args.schedule == "pipeline") else 0
    S = max(1, min(args.slots, args.steps, 8 if gsz else int(os.environ.get("MA_BENCH_MAX_SLOTS", "4"))))
    if gsz:
        S = max(gsz, (S // gsz) * gsz)
    plan = ma.BemPlan(mesh, device=local_rank)
    lu = ma.LuPlan(n, device=local_rank)
    As = [torch.empty(n * n, dtype=torch.complex128, device=dev) for _ in range(S)]
    xs_ = [torch.empty(n, dtype=torch.complex128, device=dev) for _ in range(S)]
    stream = torch.cuda.current_stream().cuda_stream
    cu_split = 0
    if lu.main_stream():
        # the chip is split (MA_LU_CU_SPLIT): the plan's big updates run on a CU-masked stream, which is a blocking stream (the only
        # kind hipExtStreamCreateWithCUMask makes) and would serialise against work on the NULL stream. The bench's own launches
        # (assemblies) go onto that stream too: no extra hardware queue
        stream = lu.main_stream()
        torch.cuda.set_stream(torch.cuda.ExternalStream(stream, device=dev))
        cu_split = int(os.environ.get("MA_LU_CU_SPLIT", "64"))
    asm_ms = np.zeros(3); lu_ms = np.zeros(8); upd = np.zeros(3); bigupd = np.zeros(2)
    timing = False

    def batch(first_step, count):
        """`count` (<= S) consecutive frequencies: assemble each, then one interleaved factor+solve."""
        for i in range(count):
            f = freqs[(rank + (first_step + i) * world) % len(freqs)]
            k = mm.wave_number(f, C_SOUND)
            beta = mm.burton_miller_beta_scaled(k, 4.0)
            plan.assemble_dev(k, beta, As[i].data_ptr(), xs_[i].data_ptr(), stream=stream)
            plan.incident_rhs_dev(k, beta, xs_[i].data_ptr(), kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, accumulate=True, stream=stream)
            if timing:
                asm_ms[:] += plan.last_timing()
        lu.factor_solve_batch_dev([a.data_ptr() for a in As[:count]], [v.data_ptr() for v in xs_[:count]], 1, stream=stream)
        if timing:
            lu_ms[:] += lu.last_timing()
            upd[:] += lu.last_update_stats(); bigupd[:] += lu.last_big_update_stats()

    def run_batches(first, nsteps):
        s = 0
        while s < nsteps:
            c = min(S, nsteps - s)
            batch(first + s, c)
            s += c

    def assemble_into(step, slot, on=None):
        on = stream if on is None else on
        f = freqs[(rank + step * world) % len(freqs)]
        k = mm.wave_number(f, C_SOUND)
        beta = mm.burton_miller_beta_scaled(k, 4.0)
        plan.assemble_dev(k, beta, As[slot].data_ptr(), xs_[slot].data_ptr(), stream=on)
        plan.incident_rhs_dev(k, beta, xs_[slot].data_ptr(), kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, accumulate=True, stream=on)

    # The pipeline assembles AHEAD: the systems of the next `ahead` steps in one call (ma_bem_plan_assemble_multi_dev: the far pairs
    # of up to three systems share one pass over the quadrature points) into spare matrices; a slot that begins a system swaps its
    # matrix with the spare that holds it. Same work inside the timed region, 1.6 GB more of HBM per system assembled ahead.
    ahead = max(1, min(3, int(os.environ.get("MA_BENCH_ASM_AHEAD", "3"))))
    spare_A, spare_x = [], []
    ready = {}                                                   # step -> index of the spare that holds its system

    def take_system(step, slot, last_step):
        """the system of `step` into slot `slot` (assembled on `stream`)"""
        if os.environ.get("MA_BENCH_NO_ASM_BOUND"):
            # diagnostic, NOT a benchmark: three systems assembled once, every step copies one (0.65 ms) -- what the schedule would
            # run at if the assemblies cost the caller's stream nothing
            if not spare_A:
                for q in range(3):
                    spare_A.append(torch.empty(n * n, dtype=torch.complex128, device=dev)); spare_x.append(torch.empty(n, dtype=torch.complex128, device=dev))
                    f = freqs[(rank + q * world) % len(freqs)]; k = mm.wave_number(f, C_SOUND); b = mm.burton_miller_beta_scaled(k, 4.0)
                    plan.assemble_dev(k, b, spare_A[q].data_ptr(), spare_x[q].data_ptr(), stream=stream)
                    plan.incident_rhs_dev(k, b, spare_x[q].data_ptr(), kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, accumulate=True, stream=stream)
            As[slot].copy_(spare_A[step % 3]); xs_[slot].copy_(spare_x[step % 3])
            return
        if ahead <= 1:
            return assemble_into(step, slot)
        if not sets:
            for _ in range(2):
                sets.append({"A": [torch.empty(n * n, dtype=torch.complex128, device=dev) for _ in range(ahead)],
                             "x": [torch.empty(n, dtype=torch.complex128, device=dev) for _ in range(ahead)],
                             "steps": [], "taken": set(), "part": 0, "ks": [], "bs": []})
        have = [t for t in sets if step in t["steps"] and step not in t["taken"]]
        if not have:                                             # (first use, or the driver jumped): assemble it and its successors now
            t = next(t for t in sets if len(t["taken"]) == len(t["steps"]))
            start_job(t, step, last_step)
            have = [t]
            other = sets[1] if t is sets[0] else sets[0]
            if len(other["taken"]) == len(other["steps"]) and step + ahead < last_step:
                start_job(other, step + ahead, last_step)      # the set after this one: in pieces, from now on
        t = have[0]
        while t["part"] < nparts:                                # not finished in the gaps: the rest now
            issue_part(t)
        i = t["steps"].index(step)
        As[slot], t["A"][i] = t["A"][i], As[slot]
        xs_[slot], t["x"][i] = t["x"][i], xs_[slot]
        t["taken"].add(step)
        if len(t["taken"]) == len(t["steps"]):                   # the set is free again: the systems after the other set's, in pieces
            other = sets[1] if t is sets[0] else sets[0]
            nxt = (max(other["steps"]) + 1) if other["steps"] and max(other["steps"]) >= step else step + 1
            if nxt < last_step:
                start_job(t, nxt, last_step)

    # Assembly AHEAD, in PIECES: two sets of `ahead` spare systems. While the slots consume one set (a slot that begins a system
    # swaps its matrix with the spare that holds it), the other set's systems are assembled by ma_bem_plan_assemble_multi_part_dev
    # in `nparts` pieces of the far pairs' rows (the far pairs of the set's systems share one pass over the quadrature points),
    # one piece per round in the rounds just before a slot begins -- where the sum of the three slots' updates is smallest and
    # the caller's stream would wait for the slots' panel chains. Same work inside the timed region.
    sets = []
    ppp = max(1, int(os.environ.get("MA_BENCH_ASM_PIECES", "4")))   # pieces per period (= per begin of a slot)
    nparts = ppp * ahead

    def start_job(t, first_step, last_step):
        t["steps"] = [s_ for s_ in range(first_step, min(first_step + ahead, last_step))]
        t["taken"] = set(); t["part"] = 0; t["ks"] = []; t["bs"] = []
        for s_ in t["steps"]:
            f = freqs[(rank + s_ * world) % len(freqs)]
            k = mm.wave_number(f, C_SOUND); t["ks"].append(k); t["bs"].append(mm.burton_miller_beta_scaled(k, 4.0))

    def issue_part(t):
        m = len(t["steps"])
        plan.assemble_multi_part_dev(t["ks"], t["bs"], [a.data_ptr() for a in t["A"][:m]], [x.data_ptr() for x in t["x"][:m]], t["part"], nparts, stream=stream)
        t["part"] += 1
        if t["part"] == nparts:
            for i in range(m):
                plan.incident_rhs_dev(t["ks"][i], t["bs"][i], t["x"][i].data_ptr(), kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, accumulate=True, stream=stream)

    def assembly_tick(r, spacing):
        """after the updates of round r: one piece of the set being assembled, in the last `ppp` rounds before a slot begins"""
        if ahead <= 1 or not sets:
            return
        if (r % spacing) < spacing - ppp:
            return
        for t in sets:
            if t["steps"] and t["part"] < nparts and not t["taken"]:
                issue_part(t)
                return

    host_round = [] if os.environ.get("MA_BENCH_HOST_ROUNDS") else None   # host seconds per stage_round call (diagnostic)

    def run_pipeline(first, nsteps):
        """The same frequencies through the staged schedule: slot s works on steps s, s + S, ... and starts a quarter of a
        factorisation after slot s - 1, so every round of block updates carries a bigger, a medium and a smaller one and no
        slot's latency-bound panel chain is ever the only thing running. No host synchronisation inside."""
        slots = max(1, min(S, nsteps))
        if gsz:
            return run_pipeline_groups(first, nsteps)
        G = lu.num_blocks()
        spacing = int(os.environ.get("MA_STAGE_SPACING", "0")) or lu.stage_spacing(slots)   # rounds between the starts of two slots (ma_lu_plan_stage_spacing: G/3 with the register pair panels, G/4 with the LDS panels)
        off = [s * spacing for s in range(slots)]
        lu.stage_reset(stream)
        asm_lane = os.environ.get("MA_BENCH_ASM_LANE", "0") != "0"
        lanes = [lu.slot_stream(s) for s in range(slots)]
        r = 0
        while True:
            sl, bl, live = [], [], False
            for s in range(slots):
                lr = r - off[s]
                if lr < 0:
                    live = True
                    continue
                sysno, g = divmod(lr, G)
                idx = s + slots * sysno
                if idx >= nsteps:
                    continue
                live = True
                own = lanes[s] if asm_lane else stream
                if g == 0:
                    if asm_lane:
                        assemble_into(first + idx, s, own)
                    else:
                        take_system(first + idx, s, first + nsteps)
                    lu.stage_begin(s, As[s].data_ptr(), xs_[s].data_ptr(), 1, own)
                sl.append(s); bl.append(g)
            if not live:
                break
            if sl:
                if host_round is not None:
                    th = time.perf_counter()
                lu.stage_round(sl, bl, stream)
                if host_round is not None:
                    host_round.append((time.perf_counter() - th, len(sl)))
            for s, g in zip(sl, bl):
                if g == G - 1:
                    lu.stage_finish(s, lanes[s] if asm_lane else stream)
            if not asm_lane:
                assembly_tick(r, spacing)
            r += 1
        if timing:
            lu_ms[:] += lu.last_timing()
            upd[:] += lu.last_update_stats(); bigupd[:] += lu.last_big_update_stats()

    def run_pipeline_model(first, nsteps):
        """The staged schedule WITHOUT rounds. The caller's stream carries the big updates (and the assemblies) of all slots in
        ONE order; in rounds -- every slot one block per round -- a slot near the end of its factorisation (tiny update, 2 ms
        of chain per block) waits each round for the updates of the slots near their start (2.5 ms of matrix-core work each).
        Here the order comes from a small model of the pipeline (lane chain, per-panel work, update time ~ (rows left)^2): the
        next job on the caller's stream is always the one that becomes ready first, so slots advance at their own pace."""
        slots = max(1, min(S, nsteps))
        G = lu.num_blocks()
        lanes = [lu.slot_stream(s_) for s_ in range(slots)]
        Lm = float(os.environ.get("MA_MODEL_LANE", "2.6")); Mm = float(os.environ.get("MA_MODEL_MWORK", "0.9"))
        Bm = float(os.environ.get("MA_MODEL_BIG", "3.0")); Am = float(os.environ.get("MA_MODEL_ASM", "5.5")); Km = float(os.environ.get("MA_MODEL_BACK", "1.5"))
        gap0 = float(os.environ.get("MA_MODEL_STAGGER", "0")) or (G * (Lm + Mm) + Am) / slots
        blk = 256.0
        def big_ms(g):
            left = max(0.0, n - blk * (g + 1))
            return Bm * (left / n) ** 2
        lu.stage_reset(stream)
        t_main = 0.0
        # per slot: next job ("asm" or block index), when it becomes eligible, lane end time of the block, end of the previous big
        sysno = [0] * slots; job = ["asm"] * slots; elig = [s_ * gap0 for s_ in range(slots)]
        t_lane = [0.0] * slots; t_bigdone = [0.0] * slots
        active = [s_ < nsteps for s_ in range(slots)]
        while any(active):
            s_ = min((i for i in range(slots) if active[i]), key=lambda i: elig[i])
            idx = s_ + slots * sysno[s_]
            if job[s_] == "asm":
                start = max(t_main, elig[s_]); t_main = start + Am
                assemble_into(first + idx, s_, stream)
                lu.stage_begin(s_, As[s_].data_ptr(), xs_[s_].data_ptr(), 1, stream)
                t_lane[s_] = t_main + Lm; t_bigdone[s_] = t_main
                job[s_] = 0
                elig[s_] = max(t_lane[s_], t_bigdone[s_]) + Mm          # t_mid of block 0
            else:
                g = job[s_]
                t_mid = elig[s_]
                start = max(t_main, t_mid); dur = big_ms(g); t_main = start + dur
                lu.stage_round([s_], [g], stream)
                t_bigdone[s_] = t_main if dur > 0 else t_mid
                t_lane[s_] = t_mid + Lm
                if g == G - 1:
                    lu.stage_finish(s_, stream)
                    sysno[s_] += 1
                    job[s_] = "asm"; elig[s_] = t_mid + Km
                    if s_ + slots * sysno[s_] >= nsteps:
                        active[s_] = False
                else:
                    job[s_] = g + 1
                    elig[s_] = max(t_lane[s_], t_bigdone[s_]) + Mm
        if timing:
            lu_ms[:] += lu.last_timing()
            upd[:] += lu.last_update_stats(); bigupd[:] += lu.last_big_update_stats()

    if gsz and (args.steps % gsz or args.warmup % gsz):
        raise SystemExit("bench.py: with --group-size %d, --steps and --warmup must be multiples of it (no frequency may be skipped)" % gsz)

    def run_pipeline_groups(first, nsteps):
        """Groups of gsz slots in lock step (one panel kernel per panel for the whole group: a wavefront per system), the groups
        staggered against each other: a group's latency-bound chain runs under the other groups' trailing updates."""
        U = S // gsz
        lu.stage_set_group(gsz)
        G = lu.num_blocks()
        spacing = int(os.environ.get("MA_STAGE_SPACING", "0")) or max(1, G // U)
        off = [u * spacing for u in range(U)]
        lu.stage_reset(stream)
        r = 0
        while True:
            sl, bl, live = [], [], False
            for u in range(U):
                lr = r - off[u]
                if lr < 0:
                    live = True
                    continue
                sysno, g = divmod(lr, G)
                base = (u + U * sysno) * gsz                 # first step index of this group's current systems
                if base + gsz > nsteps:
                    continue
                live = True
                if g == 0:
                    for t in range(gsz):
                        assemble_into(first + base + t, u * gsz + t)
                        lu.stage_begin(u * gsz + t, As[u * gsz + t].data_ptr(), xs_[u * gsz + t].data_ptr(), 1, stream)
                    lu.stage_begin_group(u * gsz, stream)
                for t in range(gsz):
                    sl.append(u * gsz + t); bl.append(g)
            if not live:
                break
            if sl:
                lu.stage_round(sl, bl, stream)
            for s_, g in zip(sl, bl):
                if g == G - 1:
                    lu.stage_finish(s_, stream)
            r += 1
        if timing:
            lu_ms[:] += lu.last_timing()
            upd[:] += lu.last_update_stats(); bigupd[:] += lu.last_big_update_stats()

    run = run_pipeline if args.schedule == "pipeline" else run_batches
    if args.schedule == "pipeline" and not gsz and os.environ.get("MA_SWEEP_ORDER", "rounds") == "model":
        run = run_pipeline_model

    run(0, args.warmup)
    torch.cuda.synchronize()
    if lu.status(stream) != ma.MA_OK:
        raise SystemExit("warm-up solve failed: %s" % ma.lib().ma_last_error_string().decode())

    if os.environ.get("MA_BENCH_HOST_PROBE"):
        # how fast can the host enqueue? three systems into empty queues on an idle device: no back-pressure yet
        th = time.perf_counter()
        run(args.warmup, min(S, 3))
        th = time.perf_counter() - th
        torch.cuda.synchronize()
        sys.stderr.write("host probe: %.1f ms to enqueue %d systems into empty queues\n" % (th * 1e3, min(S, 3)))
    timing = not args.no_timing
    plan.set_timing(timing); lu.set_timing(2 if (timing and args.schedule == "pipeline") else timing)
    if timing and args.schedule == "pipeline":
        lu.reserve_events(1800 * args.steps)       # the event pool must not grow inside the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    t_enqueued = time.perf_counter() - t0            # host time to enqueue the timed steps (the device may still be running)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    per_rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        mine = torch.tensor([elapsed, float(args.steps)], dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank_ms = [float(v[0].item()) / args.steps * 1e3 for v in allv]
        if any(int(v[1].item()) != args.steps for v in allv):
            raise SystemExit("ranks disagree on the number of steps: %s" % [int(v[1].item()) for v in allv])
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if lu.status(stream) != ma.MA_OK:
        raise SystemExit("solve failed: %s" % ma.lib().ma_last_error_string().decode())
    if timing and args.schedule == "pipeline" and os.environ.get("MA_BENCH_DUMP_UPDATES"):
        # diagnostic: where does the caller's stream wait? (start, end) of every big update of the timed region
        lu.last_timing()
        iv = lu.dump_intervals(3)
        np.save(os.environ["MA_BENCH_DUMP_UPDATES"], iv)
        gaps = iv[1:, 0] - iv[:-1, 1]
        dur = iv[:, 1] - iv[:, 0]
        span = iv[-1, 1] - iv[0, 0]
        sys.stderr.write("big updates: %d, busy %.1f ms of %.1f ms (%.3f); gaps: total %.1f ms; > 0.05 ms: %d (%.1f ms); > 0.5 ms: %d (%.1f ms); > 2 ms: %d (%.1f ms)\n"
                         % (len(iv), dur.sum(), span, dur.sum() / span, gaps.sum(), (gaps > 0.05).sum(), gaps[gaps > 0.05].sum(), (gaps > 0.5).sum(), gaps[gaps > 0.5].sum(),
                            (gaps > 2).sum(), gaps[gaps > 2].sum()))
    if timing and args.schedule == "pipeline":
        # per-call assembly events would need a host synchronisation per system inside the pipeline: the assembly phases
        # are timed in a separate pass over the same frequencies, after the timed region
        if ahead > 1 and sets:                     # as the timed region assembled them: `ahead` systems per pass over the quadrature points
            spare_A, spare_x = sets[0]["A"], sets[0]["x"]
            for i in range(0, args.steps, ahead):
                steps_ = list(range(args.warmup + i, min(args.warmup + i + ahead, args.warmup + args.steps)))
                ks_ = [mm.wave_number(freqs[(rank + s_ * world) % len(freqs)], C_SOUND) for s_ in steps_]
                plan.assemble_multi_dev(ks_, [mm.burton_miller_beta_scaled(k, 4.0) for k in ks_], [spare_A[q].data_ptr() for q in range(len(steps_))],
                                        [spare_x[q].data_ptr() for q in range(len(steps_))], stream=stream)
                asm_ms[:] += plan.last_timing()
        else:
            for i in range(args.steps):
                assemble_into(args.warmup + i, 0)
                asm_ms[:] += plan.last_timing()
        # the other phases of the factorisation (panel, interchanges, U12, substitutions) are bracketed in one lock-step batch
        # outside the timed region: inside it only the trailing-update launches carry events (every event sits on a
        # latency-bound chain; all of them cost 2.4 ms per frequency)
        keep = (lu_ms.copy(), upd.copy(), asm_ms.copy(), bigupd.copy())
        lu_ms[:] = 0; upd[:] = 0
        lu.set_timing(1)
        batch(args.warmup, min(S, 4))
        torch.cuda.synchronize()
        diag_ms = lu_ms / min(S, 4)
        lu_ms[:], upd[:], asm_ms[:], bigupd[:] = keep
    for v in xs_:
        if not np.all(np.isfinite(v.cpu().numpy().view(np.float64))):
            raise SystemExit("non-finite solution")

    if rank == 0 and host_round:
        hr = sorted(t for t, c in host_round if c == max(c2 for _, c2 in host_round))
        sys.stderr.write("host time per full stage_round call: min %.3f ms, median %.3f ms, max %.3f ms over %d calls (%d rounds per system)\n"
                         % (hr[0] * 1e3, hr[len(hr) // 2] * 1e3, hr[-1] * 1e3, len(hr), lu.num_blocks()))
    if rank == 0:
        K = args.steps
        sys.stderr.write("host enqueue %.1f ms per step of %.1f ms per step\n" % (t_enqueued * 1e3 / K, elapsed * 1e3 / K))
        total_pairs = float(n) * n * K * world
        out = {
            "metric": "bem_sweep_panel_pairs_per_s", "value": total_pairs / elapsed, "unit": "panel-pairs/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
            "config": {"workload": "S10 UV-sphere r=0.1 n_theta=%d n_phi=%d -> %d Tri3 panels; 64 log-spaced frequencies 100 Hz-8 kHz sharded "
                                   "f -> rank f mod N; rigid BC, beta=4i/k, plane wave +z; step = TBEM assembly + incident RHS + dense complex LU "
                                   "solve (zgesv) of one frequency, device-resident" % (args.n_theta, args.n_phi, n),
                       "panels": n, "frequencies_per_gpu": K, "frequencies_in_flight_per_gpu": S, "schedule": args.schedule,
                       "sharding": "frequency sweep, no data-path collective", "mode": "ranks"},
            "per_rank_ms_per_step": per_rank_ms, "per_rank_frequencies": [K] * world,
        }
        if timing:
            asm_t = asm_ms.sum() / K * 1e-3
            gemm_t = (lu_ms[3] + lu_ms[7]) / K * 1e-3   # every zgemm launch: main lane + look-ahead lanes
            lu_t = lu_ms[6] / K * 1e-3                 # whole factor+solve on the caller's stream (panel overlaps zgemm)
            if args.schedule == "pipeline":            # the span also holds the assemblies, which share the caller's stream with the big updates
                lu_t = max(lu_t - asm_t, 1e-9)
            n_gemm = max(1.0, upd[0] / K)
            gf = upd[1] / K
            out["assembly_pairs_per_s"] = n * n / asm_t
            out["solve_gflops"] = lu_flops(n) / lu_t / 1e9
            out["phase_ms_per_step"] = {"assembly_far": asm_ms[0] / K, "assembly_near": asm_ms[1] / K, "assembly_self": asm_ms[2] / K,
                                        "lu_panel": lu_ms[0] / K, "lu_swaps": lu_ms[1] / K, "lu_trsm": lu_ms[2] / K, "lu_zgemm": lu_ms[3] / K,
                                        "lu_rhs_and_triangular": lu_ms[4] / K, "lu_zgemm_lookahead_lanes": lu_ms[7] / K, "lu_total": lu_ms[6] / K,
                                        "note": "lu_panel and lu_zgemm_lookahead_lanes run on the look-ahead streams concurrently with the main lane; lu_panel intervals include queueing behind other systems' panels"}
            if args.schedule == "pipeline":
                ph = out["phase_ms_per_step"]
                for key, idx in (("lu_panel", 0), ("lu_swaps", 1), ("lu_trsm", 2), ("lu_rhs_and_triangular", 4)):
                    ph[key] = diag_ms[idx]
                ph["note"] += "; pipeline schedule: lu_panel / lu_swaps / lu_trsm / lu_rhs_and_triangular and the assembly phases come from separate passes after the timed region, lu_zgemm* and lu_total from events inside it (lu_total spans the assemblies too)"
            ach = gf / gemm_t / 1e12
            all_launches = {"kernels": "zgemm3m_dma_kernel<2, 2, true> + <2, 2, false> (every update launch the library counts: the big updates on the caller's stream and the K = 64 in-block updates on the look-ahead lanes, all timed under co-tenancy)",
                            "achieved": ach, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TF, "launches_per_step": n_gemm, "avg_launch_ms": gemm_t / n_gemm * 1e3,
                            "algorithmic_flops_per_step": gf, "raw_mfma_frac": 0.75 * ach / FP64_MFMA_PEAK_TF}
            # the dominant kernel: the big trailing updates on the caller's stream (their own instantiation of the update kernel, so
            # that the committed kernel statistics show them apart): nine tenths of a step's flops
            n_big = max(1.0, bigupd[0] / K)
            big_f = bigupd[1] / K
            big_t = lu_ms[3] / K * 1e-3
            if big_f <= 0.0:                                  # lock-step schedule on an old library: fall back to all launches
                n_big, big_f, big_t = n_gemm, gf, gemm_t
            bach = big_f / big_t / 1e12
            out["roofline"] = {"kernel": "zgemm3m_dma_kernel<2, 2, true> (the big LU trailing updates on the caller's stream, K = %d, v_mfma_f64_16x16x4_f64)" % big_update_line(n, lu.num_blocks(), 1.0)["K"],
                               "bound": "mfma", "achieved": bach, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": bach / FP64_MFMA_PEAK_TF,
                               "traffic": pmc_traffic("ma::zgemm3m_dma_kernel<2, 2, true>") or pmc_traffic("ma::zgemm3m_dma_kernel<2, 2>") or pmc_traffic("ma::zgemm3m_sub_kernel"),
                               "raw_mfma_frac": 0.75 * bach / FP64_MFMA_PEAK_TF,
                               "raw_mfma_note": "achieved/frac count ALGORITHMIC flops (8 M N K per complex update); the 3M kernel issues 3 real products per complex product, i.e. 3/4 of them on the matrix cores",
                               "traffic_note": "HBM-side bytes per launch from a separate rocprofv3 --pmc pass (profiles/); the algorithmic C read+write is %.3g B per launch on average" % (32.0 * big_f / (8.0 * big_update_line(n, lu.num_blocks(), 1.0)["K"]) / n_big),
                               "launches_per_step": n_big, "avg_launch_ms": big_t / n_big * 1e3,
                               "algorithmic_flops_per_step": big_f, "share_of_update_flops": big_f / gf if gf > 0 else None,
                               "all_update_launches": all_launches,
                               "cus_note": ("these updates run on a stream masked to %d of 256 CUs (the other %d are left to the panel kernels: lu_plan.hip, MA_LU_CU_SPLIT); "
                                            "peak is the whole chip's" % (256 - cu_split, cu_split)) if cu_split else "updates on the whole chip"}
            far_t = asm_ms[0] / K * 1e-3
            out["roofline_assembly"] = {"kernel": "tbem_far_kernel<%d, velocity-only> (far pairs of %d systems per pass)" % (ahead, ahead), "bound": "hbm", "achieved": 16.0 * n * n / far_t / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": 16.0 * n * n / far_t / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("ma::tbem_far_kernel<%d, true>" % ahead) or pmc_traffic("ma::tbem_far_kernel"), "systems_per_pass": ahead,
                                        "traffic_note": "bytes per LAUNCH: a launch writes one piece (1 / %d of the rows) of the matrices of `systems_per_pass` systems (16 B x N^2 each)" % nparts,
                                        "note": "16 B written per pair; the kernel is FP64-VALU/transcendental bound (SURVEY §8d): ~1.2 kflop per pair",
                                        "fp64_valu_tflops_equiv": 1.2e3 * n * n / far_t / 1e12}
            try:
                out["mfma_f64_probe_tflops"] = ma.probe_mfma_f64(local_rank)
            except Exception as e:      # diagnostics only
                out["mfma_f64_probe_tflops"] = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.n_theta, args.n_phi, freqs[32])
        if world == 1 and not args.no_extras and timing:
            # BASELINE configs #4 and #5, after the timed region like the diagnostic passes above: the sweep's buffers are released first
            torch.cuda.synchronize()
            torch.cuda.set_stream(torch.cuda.default_stream(dev))
            del As[:], xs_[:]
            lu.close(); plan.close()
            torch.cuda.empty_cache()
            try:
                t0x = time.perf_counter()
                res, roof, config, _ = fem_measure(10, 2, 100, cpu=False)
                out["roofline_fem"] = dict(roof, config=config["workload"], jacobi_ms=res["jacobi"]["ms"], l1_jacobi_ms=res["l1_jacobi"]["ms"], spmv_ms=res["spmv"]["ms"],
                                           spmv_rotating_ms=res["spmv_rotating_3_operators"]["ms"], jacobi_rotating_ms=res["jacobi_rotating_3_operators"]["ms"],
                                           sym_gauss_seidel_ms=res["sym_gauss_seidel"]["ms"], seconds=time.perf_counter() - t0x)
                t0x = time.perf_counter()
                out["config5"] = config5_measure()
                out["config5"]["seconds"] = time.perf_counter() - t0x
            except Exception as e:      # the extras must not take the headline line with them
                out["extras_error"] = repr(e)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
