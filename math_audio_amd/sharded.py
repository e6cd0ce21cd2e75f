"""Row-sharded operator and GMRES for systems that do not fit one GPU as a dense matrix
(BASELINE configs[4], SURVEY §2b C2: the 50k-panel matrix-free path on 8 GPUs).

Every rank keeps the full x, owns a contiguous block of collocation ROWS of the operator, produces its slice of
y = A x with the HIP operator kernels and the slices are exchanged with ONE all-gather per apply (RCCL over xGMI with
the "nccl" backend; 16 B x N = 0.8 MB at 50k panels: latency-bound, no reduction, no atomics). The Krylov vectors
are replicated: every rank runs the same restarted GMRES (math-solvers/src/iterative/gmres.rs:105-277) on identical data,
so no other collective is needed and all ranks agree bit for bit.

On a GPU the solver is the LIBRARY's device GMRES (`ma_gmres` / `ma_gmres_preconditioned`): the row block becomes an ordinary
`ma_op_t` through `ma_op_create_gathered`, whose exchange callback runs the all-gather of this module on the communicator the
caller holds (round 3; before that the iteration below ran as torch arithmetic with a host synchronisation per inner product).
The torch iteration remains for operators without a library handle (the CPU rehearsal of the exchange with gloo).
"""
import math
import numpy as np


def row_block(n, rank, world):
    """Rows [r0, r1) of rank `rank`: equal blocks of ceil(n / world) rows, the last one shorter (or empty)."""
    per = (n + world - 1) // world
    r0 = min(n, rank * per)
    return r0, min(n, r0 + per)


class ShardedOperator:
    """y = A x with the rows of A spread over the ranks of `dist`.

    local_apply(x_full, y_block): writes rows [r0, r1) of A x into y_block (a tensor of r1 - r0 entries).
    """

    def __init__(self, n, local_apply, dist=None, device="cpu"):
        import torch
        self.n = n
        self.dist = dist if dist is not None and dist.is_initialized() else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.r0, self.r1 = row_block(n, self.rank, self.world)
        self.per = (n + self.world - 1) // self.world
        self.local_apply = local_apply
        self.device = device
        self._block = torch.zeros(self.per, dtype=torch.complex128, device=device)
        self._all = torch.zeros(self.per * self.world, dtype=torch.complex128, device=device)
        self.applies = 0

    def apply(self, x):
        import torch
        self.applies += 1
        self.local_apply(x, self._block[: self.r1 - self.r0])
        if self.world == 1:
            return self._block[: self.n].clone()
        # complex tensors travel as their real view
        self.dist.all_gather_into_tensor(torch.view_as_real(self._all), torch.view_as_real(self._block))
        return self._all[: self.n].clone()          # blocks are contiguous and only the last may be short


def tbem_sharded_operator(plan, k, beta, dist=None, device=None):
    """The matrix-free TBEM operator (ma_op_create_tbem with a row range) as a ShardedOperator on this rank's GPU."""
    import torch
    import math_audio_amd as ma
    n = plan.num_dofs
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    r0, r1 = row_block(n, rank, world)
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    op = ma.LinearOperator.tbem(plan, k, beta, rows=(r0, r1)) if r1 > r0 else None
    y_full = torch.zeros(n, dtype=torch.complex128, device=dev)

    def local_apply(x, y_block):
        if op is None:
            return
        op.apply_dev(x.data_ptr(), y_full.data_ptr(), torch.cuda.current_stream().cuda_stream)   # writes rows [r0, r1) of y_full
        y_block.copy_(y_full[r0:r1])
    so = ShardedOperator(n, local_apply, dist=dist, device=dev)
    so._keep = (op, y_full)
    # the same row block as an ma_op_t of the library: its exchange callback is this module's all-gather
    if op is not None:
        so.lib_op = library_operator(so, op, n, r0, r1, dev)
    return so


def library_operator(so, inner, n, r0, r1, dev):
    """ma_op_create_gathered over `inner` (rows [r0, r1) of y): the callback all-gathers the ranks' blocks into d_y in place.
    With the "nccl" backend (RCCL) the collective runs on device tensors; other backends (gloo) go through host staging."""
    import torch
    import math_audio_amd as ma
    per, world, dist = so.per, so.world, so.dist
    block = torch.zeros(per, dtype=torch.complex128, device=dev)
    allb = torch.zeros(per * world, dtype=torch.complex128, device=dev)
    on_device = dist is not None and dist.get_backend() == "nccl"

    def gather(d_y, n_, row0, row1, stream):
        if world == 1:
            return 0
        ext = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.default_stream(dev)
        with torch.cuda.stream(ext):
            y = _as_tensor(d_y, n_, dev)
            block.zero_(); block[: row1 - row0].copy_(y[row0:row1])
            if on_device:
                dist.all_gather_into_tensor(torch.view_as_real(allb), torch.view_as_real(block))
            else:
                hb = torch.view_as_real(block).cpu(); ha = torch.empty(per * world, 2, dtype=torch.float64)
                dist.all_gather_into_tensor(ha, hb)
                torch.view_as_real(allb).copy_(ha)
            y.copy_(allb[:n_])
        return 0
    lop = ma.LinearOperator.gathered(inner, r0, r1, gather)
    lop._keep2 = (block, allb)
    return lop


def _as_tensor(ptr, n, dev):
    """complex128 tensor view of n entries of device memory at `ptr` (no copy)."""
    import torch

    class _Arr:
        __cuda_array_interface__ = {"shape": (2 * n,), "typestr": "<f8", "data": (ptr, False), "version": 3}
    return torch.view_as_complex(torch.as_tensor(_Arr(), device=dev).view(n, 2))


def gmres(op, b, x0=None, restart=30, max_iterations=100, tol=1e-6):
    """Restarted GMRES, step for step gmres_with_guess (gmres.rs:105-277): modified Gram-Schmidt, conjugated Givens
    rotations, relative tolerance on ||b||, breakdown at 1e-14; converged inside the Arnoldi loop returns at once, running
    out of restarts returns the true residual with converged = False. `op.apply(x)` is the only distributed step.
    Returns (x, info) with info = dict(iterations, restarts, converged, residual)."""
    import torch
    if getattr(op, "lib_op", None) is not None:
        # the library's device GMRES on the rank-sharded ma_op_t: same steps, vectors and scalars stay on the device
        import math_audio_amd as ma
        xh, info = ma.gmres(op.lib_op, b.cpu().numpy(), None if x0 is None else x0.cpu().numpy(), restart=restart, max_iterations=max_iterations, tol=tol)
        return torch.tensor(xh, device=b.device), dict(iterations=info.iterations, restarts=info.restarts, converged=bool(info.converged), residual=info.residual)
    n = b.shape[0]
    x = torch.zeros(n, dtype=torch.complex128, device=b.device) if x0 is None else x0.clone()
    b_norm = float(torch.linalg.vector_norm(b))
    if b_norm < 1e-15:
        return torch.zeros_like(b), dict(iterations=0, restarts=0, converged=True, residual=0.0)
    m = restart
    total, restarts = 0, 0
    for _outer in range(max_iterations):
        r = b - op.apply(x)
        beta = float(torch.linalg.vector_norm(r))
        rel = beta / b_norm
        if rel < tol:
            return x, dict(iterations=total, restarts=restarts, converged=True, residual=rel)
        V = [r * (1.0 / beta)]
        H = np.zeros((m + 1, m), dtype=np.complex128)
        cs, sn = [], []
        g = np.zeros(m + 1, dtype=np.complex128); g[0] = beta
        for j in range(m):
            total += 1
            w = op.apply(V[j])
            for i in range(j + 1):                         # modified Gram-Schmidt (gmres.rs:181-185)
                hij = complex(torch.vdot(V[i], w))         # inner_product = sum conj(v) w
                H[i, j] = hij
                w = w - hij * V[i]
            wn = float(torch.linalg.vector_norm(w))
            H[j + 1, j] = wn
            inner_converged = wn < 1e-14                   # breakdown (gmres.rs:190-199)
            if not inner_converged:
                V.append(w + (1.0 / wn - 1.0) * w)
            for i in range(j):                             # previous rotations (gmres.rs:202-206)
                t = np.conj(cs[i]) * H[i, j] + np.conj(sn[i]) * H[i + 1, j]
                H[i + 1, j] = 0.0 - sn[i] * H[i, j] + cs[i] * H[i + 1, j]
                H[i, j] = t
            a, bb = H[j, j], H[j + 1, j]                   # givens_rotation (gmres.rs:589-603)
            if abs(bb) < 1e-30:
                c, s = 1.0 + 0j, 0j
            elif abs(a) < 1e-30:
                c, s = 0j, 1.0 + 0j
            else:
                rr = math.sqrt(abs(a) ** 2 + abs(bb) ** 2)
                c, s = a * (1.0 / rr), bb * (1.0 / rr)
            cs.append(c); sn.append(s)
            H[j, j] = np.conj(c) * H[j, j] + np.conj(s) * H[j + 1, j]
            H[j + 1, j] = 0.0
            t = np.conj(c) * g[j] + np.conj(s) * g[j + 1]
            g[j + 1] = 0.0 - s * g[j] + c * g[j + 1]
            g[j] = t
            rel = abs(g[j + 1]) / b_norm
            if rel < tol or inner_converged:
                y = _solve_upper_triangular(H, g, j + 1)
                for i in range(j + 1):
                    x = x + complex(y[i]) * V[i]
                return x, dict(iterations=total, restarts=restarts, converged=True, residual=rel)
        y = _solve_upper_triangular(H, g, m)
        for i in range(m):
            x = x + complex(y[i]) * V[i]
        restarts += 1
    rt = float(torch.linalg.vector_norm(b - op.apply(x))) / b_norm
    return x, dict(iterations=total, restarts=restarts, converged=False, residual=rt)


def _solve_upper_triangular(H, g, k):
    """gmres.rs:606-621."""
    y = np.zeros(k, dtype=np.complex128)
    for i in range(k - 1, -1, -1):
        acc = g[i]
        for j in range(i + 1, k):
            acc = acc - H[i, j] * y[j]
        if abs(H[i, i]) > 1e-30:
            y[i] = acc * (1.0 / H[i, i])
    return y
