"""Diagnostic: what the f64 matrix cores sustain on this chip (ma_diag_mfma_burn: 12 independent accumulators per wavefront, no
memory traffic), at 1 / 2 / 3 wavefronts per SIMD. The update kernel's fraction of peak is priced against 78.6 TFLOP/s."""
import ctypes as C
import os
import sys
# ma_diag_mfma_burn exists only in the diagnostic build (make -C math_audio_amd/csrc diag); the shipped library has ma_probe_mfma_f64
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MA_LIB_PATH", os.path.join(_root, "math_audio_amd", "lib", "libmathaudio_hip_diag.so"))
sys.path.insert(0, _root)
import torch
import math_audio_amd as ma

dev = torch.device("cuda", 0)
lib = ma.lib()
lib.ma_diag_mfma_burn.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
out = torch.zeros(256 * 4096, dtype=torch.float64, device=dev)
for blocks in (256, 512, 768, 1024):
    iters = 20000
    ma.check(lib.ma_diag_mfma_burn(C.c_void_p(out.data_ptr()), blocks, 2000, 1, C.c_void_p(0)))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    ma.check(lib.ma_diag_mfma_burn(C.c_void_p(out.data_ptr()), blocks, iters, 3, C.c_void_p(0)))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    flops = blocks * 4.0 * iters * 12.0 * 2048.0
    print("%4d workgroups of 4 wavefronts (%.0f per SIMD): %7.2f ms  %6.1f TFLOP/s  (%.1f ns per MFMA per SIMD)" % (blocks, blocks / 256.0, ms, flops / ms / 1e9, ms * 1e6 / (blocks / 256.0 * iters * 12.0)))
    sys.stdout.flush()
