"""Diagnostic: S independent 10k systems factored (a) as one interleaved batch on one stream, (b) one plan and stream each."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import math_audio_amd as ma
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
A0 = [(torch.randn(n, n, dtype=torch.float64, generator=g) + 1j * torch.randn(n, n, dtype=torch.float64, generator=g)).to(dev) for _ in range(S)]
b0 = torch.ones(n, dtype=torch.complex128, device=dev)
plans = [ma.LuPlan(n) for _ in range(S)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
for mode in ("batch", "streams", "batch", "streams"):
    A = [a.clone() for a in A0]; b = [b0.clone() for _ in range(S)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "batch":
        plans[0].factor_solve_batch_dev([a.data_ptr() for a in A], [x.data_ptr() for x in b], 1, torch.cuda.current_stream().cuda_stream)
    else:
        for i in range(S):
            plans[i].factor_solve_dev(A[i].data_ptr(), b[i].data_ptr(), 1, streams[i].cuda_stream)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("%-8s %d systems: %.1f ms per system" % (mode, S, (t1 - t0) * 1e3 / S))
    for i in range(S):
        st = streams[i].cuda_stream if mode == "streams" else torch.cuda.current_stream().cuda_stream
        assert plans[i if mode == "streams" else 0].status(st) == 0
