"""Round 5: one 10 000-row factorisation with the look-ahead lanes off (every kernel alone on the chip) -- run under
rocprofv3 --kernel-trace --stats to read the idle durations of the panel kernels. argv[1]: partial | tournament."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MA_LU_LOOKAHEAD"] = "0"
import torch
import math_audio_amd as ma

mode = sys.argv[1] if len(sys.argv) > 1 else "tournament"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.randn(n, n, dtype=torch.complex128, device=dev, generator=g)
b = torch.randn(n, dtype=torch.complex128, device=dev, generator=g)
lu = ma.LuPlan(n, pivoting=mode)
st = lu.main_stream() or torch.cuda.current_stream().cuda_stream
for rep in range(2):
    dA = A.clone().reshape(-1); db = b.clone()
    torch.cuda.synchronize()
    lu.factor_solve_dev(dA.data_ptr(), db.data_ptr(), 1, st)
    assert lu.status(st) == ma.MA_OK
    torch.cuda.synchronize()
res = float(torch.linalg.norm(A @ db - b) / torch.linalg.norm(b))
print(mode, n, "residual", res)
lu.close()
