#!/usr/bin/env python3
"""bench.py — BEM frequency-sweep throughput on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2]): S10 = UV sphere r = 0.1 m,
n_theta = 51, n_phi = 100 -> 10 000 Tri3 panels; frequencies taken from the 64 log-spaced
points 100 Hz .. 8 kHz; rigid BC, beta = 4i/k (BemSolver default), plane wave +z.
One step = one frequency: TBEM assembly (far + near + self kernels) + incident RHS + dense
complex LU solve, everything resident in HBM (geometry and the near-pair plan are uploaded once
before the timed region; nothing crosses PCIe inside it).

The timed region is ONE call of the library's own frequency loop per rank -- ma_bem_sweep_run on a reusable handle
(include/mathaudio_hip.h; the C-ABI entry a Rust caller binds in place of room_simulator_bem.rs:328-360) -- so the number of
record comes from the boundary, not from orchestration in this file.

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


def lu_flops(n):
    return (8.0 / 3.0) * n ** 3 + 8.0 * n ** 2          # zgetrf + zgetrs, real flops


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
    # how many threads the host's BLAS used for that (threadpoolctl reads the loaded libraries; the environment may cap them)
    try:
        from threadpoolctl import threadpool_info
        blas = "; ".join("%s %s: %s threads" % (i.get("internal_api"), i.get("version"), i.get("num_threads")) for i in threadpool_info() if i.get("user_api") == "blas") or "no BLAS pool reported"
    except Exception as e:
        blas = "thread count unknown (%s)" % type(e).__name__
    blas += ", OMP_NUM_THREADS=%s" % os.environ.get("OMP_NUM_THREADS", "unset")
    return {
        "value": n * n / t_step, "unit": "panel-pairs/s", "cores": cores, "kind": "port",
        "sample": "oracle C restatement of build_tbem_system_with_beta: %d of %d rows at %.0f Hz on %d threads (%.1f s) -> %.3e pairs/s; "
                  "dense solve: %s [%s], n=%d%s, %.2f s -> %.1f GFLOP/s; step time = N^2/asm + ((8/3)N^3+8N^2)/solve" % (
                      rows, n, freq, cores, t_probe, asm_pairs_per_s, lib, blas, ns, " (the workload's own size)" if ns == n else " (scaled by flops to n=%d)" % n, t_lu, solve_gflops),
        "blas_threads": blas,
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


def inlib_multi(args):
    """--inlib-multi: ONE process, ma_bem_solve_sweep_multi_timed over --gpus N devices (a host thread, BEM plan and sweep handle per
    device inside the library: what a single Rust process gets on a node). Frequency i of the list belongs to device i mod N."""
    import math_audio_amd as ma
    from math_audio_amd import mesh as mm
    N = args.gpus
    devices = [int(t) for t in args.devices.split(",")] if args.devices else list(range(N))
    if len(devices) != N:
        raise SystemExit("bench.py --inlib-multi: --devices lists %d devices, --gpus says %d" % (len(devices), N))
    mesh = mm.generate_sphere_mesh(RADIUS, args.n_theta, args.n_phi)
    n = mesh.n_elem
    freqs = mm.log_space(100.0, 8000.0, 64)

    def flist(first, count):
        return [freqs[(d + (first + s_) * N) % len(freqs)] for s_ in range(count) for d in range(N)]
    # the reusable handle (ma_bem_sweep_multi_create: a BEM plan and a sweep handle per device) is made once, outside the timed region,
    # as the single-device handle is; the timed region is ONE ma_bem_sweep_multi_run
    ts = time.perf_counter()
    handle = ma.BemSweepMulti(mesh, devices, max(args.steps, args.warmup, 1) * N, slots=args.slots)
    setup = [time.perf_counter() - ts] * N
    if args.warmup > 0:
        handle.run(flist(0, args.warmup), speed_of_sound=C_SOUND, beta_scale=4.0)
    t0 = time.perf_counter()
    X, st = handle.run(flist(args.warmup, args.steps), speed_of_sound=C_SOUND, beta_scale=4.0)
    wall = time.perf_counter() - t0
    secs, cnt = handle.last_timing()
    handle.close()
    if not np.all(st == 0) or not np.all(np.isfinite(X.view(np.float64))) or any(int(c) != args.steps for c in cnt):
        raise SystemExit("the multi-device sweep failed: status %s, counts %s" % (sorted(set(int(v) for v in st)), list(map(int, cnt))))
    elapsed, K = float(max(secs)), args.steps
    print(json.dumps({"metric": "bem_sweep_panel_pairs_per_s", "value": float(n) * n * K * N / elapsed, "unit": "panel-pairs/s", "n_gpus": N, "steps": K, "warmup": args.warmup,
                      "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
                      "config": {"workload": workload_text(args, n), "panels": n, "frequencies_per_gpu": K, "frequencies_in_flight_per_gpu": args.slots,
                                 "mode": "in-library, one process: ma_bem_sweep_multi_run on a reusable handle (a host thread per device)", "devices": devices,
                                 "sharding": "frequency sweep, no data-path collective"},
                      "per_device_ms_per_step": [float(v) / K * 1e3 for v in secs], "per_device_plan_setup_s": [float(v) for v in setup], "wall_s_of_the_call": wall,
                      "note": "value uses the slowest device's run (solutions copied back included); the handle's creation (all devices in parallel, outside the timed region) is in per_device_plan_setup_s"}))


def workload_text(args, n):
    return ("S10 UV-sphere r=0.1 n_theta=%d n_phi=%d -> %d Tri3 panels; 64 log-spaced frequencies 100 Hz-8 kHz sharded f -> rank f mod N; rigid BC, "
            "beta=4i/k, plane wave +z; step = TBEM assembly + incident RHS + dense complex LU solve (zgesv) of one frequency, device-resident"
            % (args.n_theta, args.n_phi, n))


def residual_check(ma, mm, torch, plan, n, dev, freqs_run, X, count):
    """After the timed region: the last `count` systems re-assembled on the single-system path (the factorisation destroyed the
    sweep's copies) and ||A x - b|| / ||b|| of the sweep's solutions, in torch on the device. The checker, not the product."""
    A = torch.empty(n * n, dtype=torch.complex128, device=dev); b = torch.empty(n, dtype=torch.complex128, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    worst, which = 0.0, []
    for i in range(len(freqs_run) - count, len(freqs_run)):
        k = mm.wave_number(freqs_run[i], C_SOUND); beta = mm.burton_miller_beta_scaled(k, 4.0)
        plan.assemble_dev(k, beta, A.data_ptr(), b.data_ptr(), stream=st)
        plan.incident_rhs_dev(k, beta, b.data_ptr(), accumulate=True, stream=st)
        x = torch.from_numpy(X[i]).to(dev)
        res = float(torch.linalg.norm(A.view(n, n) @ x - b) / torch.linalg.norm(b))
        worst = max(worst, res); which.append(round(float(freqs_run[i]), 1))
    del A
    return {"max_rel_residual": worst, "frequencies_hz": which, "bound": 1e-10,
            "note": "||A x - b|| / ||b|| of the sweep's last solutions against systems re-assembled by ma_bem_plan_assemble_dev after the timed region"}


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
    ap.add_argument("--no-timing", action="store_true", help="no HIP events around the update launches and the assembly pieces inside the timed region")
    ap.add_argument("--no-extras", action="store_true", help="skip the passes after the timed region that put configs #4 (FEM SpMV / smoother) and #5 (50k-panel operators) into the line")
    ap.add_argument("--no-check", action="store_true", help="skip the residual check of the last systems after the timed region")
    ap.add_argument("--inlib-multi", action="store_true", help="ONE process: ma_bem_solve_sweep_multi over --gpus N devices (a host thread per device inside the library) "
                                                                "instead of one torch.distributed rank per GPU")
    ap.add_argument("--devices", default="", help="--inlib-multi: comma-separated device list instead of 0..N-1 (repeats need the diagnostic build: MA_LIB_PATH=.../libmathaudio_hip_diag.so MA_TEST_ALLOW_DUPLICATE_DEVICES=1)")
    ap.add_argument("--slots", type=int, default=int(os.environ.get("MA_SWEEP_SLOTS", "3")), help="frequencies in flight per GPU (1..4)")
    ap.add_argument("--dump-updates", default="", help="diagnostic: save (start, end) ms of every big update of the timed run to this .npy and print the stream's gaps")
    args = ap.parse_args()
    if args.workload == "fem":
        return fem_workload(args)
    if args.inlib_multi:
        return inlib_multi(args)
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
    K, W = args.steps, args.warmup
    S = max(1, min(args.slots, 4, K))
    # rank r solves the list's points r, r + N, ...: its s-th step is point (r + s N) mod 64 (the list wraps for long runs)
    mine = lambda first, count: [freqs[(rank + (first + s_) * world) % len(freqs)] for s_ in range(count)]
    plan = ma.BemPlan(mesh, device=local_rank)
    # The library's frequency loop behind its reusable handle: LU plan, streams, S systems in flight, the spare systems of the
    # assembly-ahead and the parked solutions are allocated HERE, once, outside the timed region (inputs resident in HBM).
    proxy = world == 1 and not args.no_extras
    sweep = ma.BemSweep(plan, max(K, W, 64 if proxy else 1), slots=S)
    info = sweep.info()
    lu = sweep.lu_plan()
    if W > 0:
        _, st_w = sweep.run(mine(0, W), speed_of_sound=C_SOUND, beta_scale=4.0)
        if not np.all(st_w == 0):
            raise SystemExit("warm-up sweep failed: status %s" % sorted(set(int(v) for v in st_w)))
    timing = not args.no_timing
    sweep.set_timing(timing)                        # events around the update launches and the assembly pieces; the pools are filled now, not inside the timed region
    run_freqs = mine(W, K)
    spec0 = lu.speculation_stats()                  # (synchronises: outside the timed region)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    X, status = sweep.run(run_freqs, speed_of_sound=C_SOUND, beta_scale=4.0)      # exactly K steps; returns with the K solutions on the host
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [elapsed / K * 1e3]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allv, t)
        per_rank_ms = [float(v.item()) / K * 1e3 for v in allv]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not np.all(status == 0):
        raise SystemExit("a frequency of the sweep failed: status %s (%s)" % (sorted(set(int(v) for v in status)), ma.lib().ma_last_error_string().decode()))
    if not np.all(np.isfinite(X.view(np.float64))):
        raise SystemExit("non-finite solution")
    tm = sweep.last_timing()
    spec1 = lu.speculation_stats()
    if rank == 0:
        out = {
            "metric": "bem_sweep_panel_pairs_per_s", "value": float(n) * n * K * world / elapsed, "unit": "panel-pairs/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
            "config": {"workload": workload_text(args, n), "panels": n, "frequencies_per_gpu": K, "frequencies_in_flight_per_gpu": info["slots"],
                       "mode": "in-library: one ma_bem_sweep_run call per rank on a reusable ma_bem_sweep_t handle (the C-ABI frequency loop; staged pipeline = %s, "
                               "%d blocks per factorisation, slots %d rounds apart, %d systems assembled ahead)" % (info["staged"], info["blocks"], info["spacing"], info["systems_ahead"]),
                       "sharding": "frequency sweep, no data-path collective"},
            "per_rank_ms_per_step": per_rank_ms, "per_rank_frequencies": [K] * world,
            "sweep_call": {"wall_s_inside_the_call": tm["wall_s"], "device_ms_first_to_last_event": tm["device_ms"],
                           "note": "the timed region is the ma_bem_sweep_run call plus the bracket's synchronisations; solutions (K x N complex128) come back to the host inside it"},
        }
        out["lu_panels"] = {"pivoting": lu.pivoting(), "speculation": lu.speculation(),
                            "half_panels_accepted_per_step": (spec1[0] - spec0[0]) / K, "half_panels_accepted_widened_per_step": (spec1[1] - spec0[1]) / K,
                            "half_panels_rejected_per_step": (spec1[2] - spec0[2]) / K,
                            "note": "lu_spec.hip: a half-panel (32 columns) is first factored with partial pivoting inside its top 32 rows and checked against every row below "
                                    "(accepted = LAPACK's panel, two short launches); rows that fail the check join the candidates of a second, widened attempt; a half-panel both give up is factored "
                                    "by the plan's own panel kernel (verified mode) or its system is solved again (optimistic mode: counted in sweep_redone)"}
        if timing and info["staged"]:
            l8 = lu.last_timing()
            n_all, f_all, _ = lu.last_update_stats()
            t_all = (l8[3] + l8[7]) * 1e-3                         # every update launch: the big ones on the sweep's stream + the lanes' K = 32 / 64 / 384 ones
            n_big, f_big, t_big = max(1, tm["big_update_launches"]), tm["big_update_flops"], tm["big_update_ms"] * 1e-3
            asm_t = tm["assembly_ms"] * 1e-3
            cu_split = lu.cu_split()[0]
            bach = f_big / t_big / 1e12
            ach_all = f_all / t_all / 1e12
            e2e = lu_flops(n) * K / elapsed / 1e12
            out["roofline"] = {
                "kernel": "zgemm3m_dma_kernel<2, 2, true> (the big LU trailing updates on the sweep's stream, K = 64 x panels per block, v_mfma_f64_16x16x4_f64)",
                "bound": "mfma", "achieved": bach, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": bach / FP64_MFMA_PEAK_TF,
                "traffic": pmc_traffic("ma::zgemm3m_dma_kernel<2, 2, true>") or pmc_traffic("ma::zgemm3m_sub_kernel"),
                "launches_per_step": n_big / K, "avg_launch_ms": t_big / n_big * 1e3, "algorithmic_flops_per_launch": f_big / n_big,
                "algorithmic_flops_per_step": f_big / K, "share_of_update_flops": f_big / f_all if f_all > 0 else None,
                "raw_mfma_frac": 0.75 * bach / FP64_MFMA_PEAK_TF,
                "raw_mfma_note": "achieved / frac count ALGORITHMIC flops (8 M N K per complex update); the 3M kernel issues 3 real products per complex product, i.e. 3/4 of them on the matrix cores",
                "all_update_launches": {"achieved": ach_all, "unit": "TFLOP/s", "frac": ach_all / FP64_MFMA_PEAK_TF, "launches_per_step": n_all / K, "avg_launch_ms": t_all / max(1.0, n_all) * 1e3,
                                        "algorithmic_flops_per_step": f_all / K,
                                        "note": "every update launch (big ones + the lanes' K = 32 / 64 / 384 ones), each timed by its own events under co-tenancy: the basis of rounds 1-2"},
                "end_to_end_frac": e2e / FP64_MFMA_PEAK_TF, "end_to_end_tflops": e2e,
                "end_to_end_note": "((8/3) N^3 + 8 N^2) x steps / the timed region / peak: assembly, panels, waits and ramps included -- the figure that compares like for like across rounds",
                "cus_note": ("the big updates run on a stream masked to %d of 256 CUs (the other %d are left to the kernels of the slots' lanes: lu_plan.hip, MA_LU_CU_SPLIT); peak is the whole chip's"
                             % (256 - cu_split, cu_split)) if cu_split else "updates on the whole chip"}
            out["solve_gflops"] = lu_flops(n) * K / max(elapsed - asm_t, 1e-9) / 1e9
            out["assembly_pairs_per_s"] = float(n) * n * K / asm_t if asm_t > 0 else None
            out["phase_ms_per_step"] = {"assembly_in_the_timed_region": tm["assembly_ms"] / K, "assembly_pieces_per_step": tm["assembly_pieces"] / K,
                                        "big_updates": tm["big_update_ms"] / K, "lane_updates": l8[7] / K, "stream_neither": (tm["device_ms"] - tm["assembly_ms"] - tm["big_update_ms"]) / K,
                                        "note": "events on the sweep's stream INSIDE the timed region: assembly pieces (far pairs of up to 3 systems per pass, near, self, incident right-hand sides) and "
                                                "big updates share that stream, so device_ms - assembly - big updates is what the stream spends waiting for the slots' chains, on parking copies and in ramps; "
                                                "lane_updates run on the slots' streams beside it"}
            # 16 B written per pair; the far kernel is FP64-VALU / transcendental bound. Work per pair: ~1.2 kflop alone; with three
            # systems per pass the geometric third of a quadrature point's ~120 instructions is shared: ~93 per system -> ~0.93 kflop per pair
            per_pair = 1.2e3 * (93.0 / 120.0) if info["systems_ahead"] >= 3 else 1.2e3
            out["roofline_assembly"] = {"kernel": "tbem_far_kernel<%d, velocity-only> + near + self + incident RHS, as the timed region ran them (on the update stream's CUs, beside the lanes)" % max(1, info["systems_ahead"]),
                                        "bound": "hbm", "achieved": 16.0 * n * n * K / asm_t / 1e9 if asm_t > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": 16.0 * n * n * K / asm_t / 1e9 / HBM_PEAK_GBS if asm_t > 0 else None,
                                        "traffic": pmc_traffic("ma::tbem_far_kernel<3, true>") or pmc_traffic("ma::tbem_far_kernel"),
                                        "fp64_valu_tflops_equiv": per_pair * n * n * K / asm_t / 1e12 if asm_t > 0 else None, "flops_per_pair_assumed": per_pair,
                                        "note": "algorithmic bytes: 16 B per pair; FP64-VALU-bound, not HBM-bound (DESIGN 4)"}
            if args.dump_updates:
                iv = lu.dump_intervals(3)
                np.save(args.dump_updates, iv)
                gaps = iv[1:, 0] - iv[:-1, 1]; dur = iv[:, 1] - iv[:, 0]; span = iv[-1, 1] - iv[0, 0]
                sys.stderr.write("big updates: %d, busy %.1f ms of %.1f ms (%.3f); gaps: total %.1f ms; > 0.5 ms: %d (%.1f ms); > 2 ms: %d (%.1f ms)\n"
                                 % (len(iv), dur.sum(), span, dur.sum() / span, gaps.sum(), (gaps > 0.5).sum(), gaps[gaps > 0.5].sum(), (gaps > 2).sum(), gaps[gaps > 2].sum()))
        if proxy:
            # Strong-scaling proxy of BASELINE config #3 (64 frequencies over 1 / 2 / 4 / 8 GPUs; room_simulator_bem.rs:328-360): this handle at
            # the per-rank loads of N = 1, 2, 4, 8 -- K = 64, 32, 16, 8 frequencies of the list, strided as rank 0's share would be. A rank's
            # time at N GPUs is the K = 64 / N run's, so the projected efficiency is T(64) / (N T(64 / N)). NOT a multi-GPU measurement.
            sweep.set_timing(False)
            tk = {}
            for Np in (1, 2, 4, 8):
                Kp = 64 // Np
                fk = [freqs[(s_ * Np) % len(freqs)] for s_ in range(Kp)]
                torch.cuda.synchronize()
                tq = time.perf_counter()
                _, st_k = sweep.run(fk, speed_of_sound=C_SOUND, beta_scale=4.0)
                torch.cuda.synchronize()
                tk[Np] = time.perf_counter() - tq
                if not np.all(st_k == 0):
                    raise SystemExit("strong-scaling proxy: a frequency failed")
            out["strong_scaling_proxy"] = {
                "frequencies_per_rank": {str(Np): 64 // Np for Np in tk}, "ms_per_step": {str(Np): tk[Np] / (64 // Np) * 1e3 for Np in tk},
                "wall_s": {str(Np): tk[Np] for Np in tk}, "projected_efficiency": {str(Np): tk[1] / (Np * tk[Np]) for Np in tk},
                "note": "one GPU running the per-rank load of an N-GPU sweep of the 64-point list (K = 64 / N frequencies through the same handle): "
                        "projected_efficiency[N] = T(64) / (N T(64 / N)); the fill and drain of the three-slot pipeline are what a short run pays. A projection, not a multi-GPU measurement"}
        sweep.close()                                   # 9 matrices of 1.6 GB go back before the checker and the extras allocate theirs
        if not args.no_check:
            out["check"] = residual_check(ma, mm, torch, plan, n, dev, run_freqs, X, min(S, K))
            if not out["check"]["max_rel_residual"] <= 1e-10:
                raise SystemExit("residual check failed: %r" % out["check"])
        if timing:
            try:
                out["mfma_f64_probe_tflops"] = ma.probe_mfma_f64(local_rank)     # 3 x 6.5 ms of back-to-back v_mfma_f64_16x16x4_f64 on every CU (tools/mfma_peak_probe.py reads the same)
            except Exception:                                                  # diagnostics only
                out["mfma_f64_probe_tflops"] = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.n_theta, args.n_phi, freqs[32])
        if world == 1 and not args.no_extras and timing:
            # BASELINE configs #4 and #5, after the timed region: the sweep's buffers have been released
            plan.close()
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
    else:
        sweep.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
