"""Frequency-sweep sharding across GPUs (one process per GPU, torch.distributed).

The reference's sweeps are loops over independent frequencies (math-bem/bin/room_simulator_bem.rs:329;
the FEM CLI runs them on rayon workers, math-fem/bin/room_simulator_fem.rs:1143-1146). Sharding is
therefore frequency f -> rank f mod N with NO data-path collective; the only exchange is the final
gather of per-frequency results (C1 in SURVEY.md §2b). Works with the "nccl" backend (RCCL over xGMI on
MI355X) and with "gloo" on CPU for the tests.
"""
import numpy as np


def shard_frequencies(num_freqs, rank, world):
    """Indices of the frequencies rank `rank` owns: rank, rank + world, ..."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, num_freqs, world))


def gather_results(local_idx, local_vals, num_freqs, width, dist=None, device=None):
    """All ranks end with the (num_freqs, width) complex table; local_vals[i] belongs to frequency local_idx[i].
    Uses one all_gather of equally padded blocks (ranks own ceil(F/N) or floor(F/N) frequencies)."""
    import torch
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    per = (num_freqs + world - 1) // world
    dev = device if device is not None else "cpu"
    buf = torch.zeros((per, width), dtype=torch.complex128, device=dev)
    idx = torch.full((per,), -1, dtype=torch.int64, device=dev)
    for i, (f, v) in enumerate(zip(local_idx, local_vals)):
        buf[i] = torch.as_tensor(np.asarray(v), dtype=torch.complex128, device=dev)
        idx[i] = f
    if world == 1:
        bufs, idxs = [buf], [idx]
    else:
        # complex tensors travel as their real view (RCCL/gloo reduce and copy real dtypes)
        rb = [torch.zeros((per, width, 2), dtype=torch.float64, device=dev) for _ in range(world)]
        ib = [torch.zeros((per,), dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(rb, torch.view_as_real(buf).contiguous())
        dist.all_gather(ib, idx)
        bufs = [torch.view_as_complex(r.contiguous()) for r in rb]; idxs = ib
    out = np.zeros((num_freqs, width), dtype=np.complex128)
    for bsrc, isrc in zip(bufs, idxs):
        bi = isrc.cpu().numpy(); bv = bsrc.cpu().numpy()
        for j, f in enumerate(bi):
            if f >= 0:
                out[f] = bv[j]
    return out


def max_over_ranks(seconds, dist=None, device=None):
    """bench.py contract: the timed region's wall time is the MAX over ranks."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
