// Diagnostic (round 3): which ingredient of the trailing-update kernel's main loop costs the f64 matrix cores their last 20 %?
// The loop of zgemm3m_body (24 v_mfma_f64_16x16x4 per stage of 8 k) rebuilt from its parts, one variant adding one part:
//   0  12 MFMAs per k-step on fixed operands (registers), nothing else
//   1  + the operands of every k-step read from LDS (4 ds_read_b128, waits) and the two sums (4 v_add_f64)
//   2  + one workgroup barrier per stage
//   3  + 4 ds_write_b128 per stage (register staging to the next LDS buffer)
//   4  + 4 global_load_dwordx4 per stage (an L2-resident source), waited for before the stores
//   5  variant 2 + 4 global_load_lds_dwordx4 per stage (LDS-DMA, no registers, no ds_write), one stage left in flight across the
//      raw barrier (s_waitcnt vmcnt(4))
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_loop_probe.hip -o tools/mfma_loop_probe.bin ; run: tools/mfma_loop_probe.bin [workgroups per CU]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
struct dc { double re, im; };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define BK 8
#define BM 64
#define BN 64
#define STAGES 3
// the instruction written out: the builtin form made the compiler shuttle the accumulators between the two register files
#define MFMA(acc, x, y) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))

// LDS-DMA written out (the builtin makes the compiler drain vmcnt before the next LDS read): 16 B per lane to m0 + lane * 16
__device__ __forceinline__ void glds16(const void* g, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_off(const void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const void*)p);
}

template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void loop_kernel(const dc* __restrict__ src, double* __restrict__ out, int nstage) {
  __shared__ __attribute__((aligned(16))) dc As[STAGES][BK][BM];
  __shared__ __attribute__((aligned(16))) dc Bs[STAGES][BK][BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 15, lk = lane >> 4;
  for (int i = tid; i < STAGES * BK * BM; i += 256) { (&As[0][0][0])[i] = {1.0 + 1e-3 * i, 1.0 - 1e-3 * i}; (&Bs[0][0][0])[i] = {1.0 - 2e-3 * i, 0.5 + 1e-3 * i}; }
  __syncthreads();
  v4d t1[2][2], t2[2][2], t3[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) { t1[a][b] = (v4d){0, 0, 0, 0}; t2[a][b] = (v4d){0, 0, 0, 0}; t3[a][b] = (v4d){0, 0, 0, 0}; }
  dc ra[2] = {{1.0, 2.0}, {3.0, 4.0}}, rb[2] = {{1.5, 2.5}, {3.5, 4.5}};
  dc af0[2] = {{1.0 + lane, 2.0}, {3.0, 4.0 + lane}}, bf0[2] = {{1.5, 2.5 + lane}, {3.5 + lane, 4.5}};
  const dc* gsrc = src + (size_t)(blockIdx.x % 64) * 4096 + tid;
  int buf = 0;
  for (int st = 0; st < nstage; ++st) {
    const int nxt = buf == STAGES - 1 ? 0 : buf + 1;
    if (V == 5) {
      const int nn = nxt == STAGES - 1 ? 0 : nxt + 1;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        glds16(gsrc + (st & 3) * 1024 + 256 * c, lds_off(&As[nn][0][0] + (wave * 2 + c) * 64));
        glds16(gsrc + (st & 3) * 1024 + 512 + 256 * c, lds_off(&Bs[nn][0][0] + (wave * 2 + c) * 64));
      }
    }
    if (V >= 3 && V <= 4) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int e = tid + 256 * s;
        As[nxt][e & 7][(e >> 3) ^ (e & 7)] = ra[s];
        Bs[nxt][e >> 6][e & 63] = rb[s];
      }
    }
    if (V == 4) {
#pragma unroll
      for (int s = 0; s < 2; ++s) { ra[s] = gsrc[(st & 3) * 1024 + 256 * s]; rb[s] = gsrc[(st & 3) * 1024 + 512 + 256 * s]; }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = ks * 4 + lk;
      dc af[2], bf[2];
      double as[2], bs[2];
      if (V >= 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a) { af[a] = As[buf][kk][(wm * 32 + a * 16 + li) ^ kk]; as[a] = af[a].re + af[a].im; }
#pragma unroll
        for (int b = 0; b < 2; ++b) { bf[b] = Bs[buf][kk][wn * 32 + b * 16 + li]; bs[b] = bf[b].re + bf[b].im; }
      } else {
#pragma unroll
        for (int a = 0; a < 2; ++a) { af[a] = af0[a]; as[a] = af0[a].re * 0.5; bf[a] = bf0[a]; bs[a] = bf0[a].im * 0.5; }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          MFMA(t1[a][b], af[a].re, bf[b].re);
          MFMA(t2[a][b], af[a].im, bf[b].im);
          MFMA(t3[a][b], as[a], bs[b]);
        }
    }
    if (V >= 2 && V <= 4) __syncthreads();
    if (V == 5) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    buf = nxt;
  }
  double s = 0.0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += t1[a][b][r] + t2[a][b][r] + t3[a][b][r];
  out[(size_t)blockIdx.x * 256 + tid] = s + ra[0].re + rb[1].im;
}

template <int V>
static void run(const dc* src, double* out, int blocks, int nstage) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(loop_kernel<V>, dim3(blocks), dim3(256), 0, 0, src, out, 200);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  hipLaunchKernelGGL(loop_kernel<V>, dim3(blocks), dim3(256), 0, 0, src, out, nstage);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, a, b));
  const double flops = (double)blocks * 4.0 * nstage * 24.0 * 2048.0;
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(loop_kernel<V>)));
  printf("variant %d: %4d workgroups, %d stages: %8.3f ms  %6.1f TFLOP/s on the matrix cores (%d registers, %zu B LDS)\n", V, blocks, nstage, ms, flops / ms / 1e9, fa.numRegs, (size_t)fa.sharedSizeBytes);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int per_cu = argc > 1 ? atoi(argv[1]) : 3;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * per_cu, nstage = 4000;
  dc* src; double* out;
  CK(hipMalloc(&src, sizeof(dc) * 64 * 4096 + sizeof(dc) * 8192));
  CK(hipMemset(src, 0, sizeof(dc) * 64 * 4096 + sizeof(dc) * 8192));
  CK(hipMalloc(&out, sizeof(double) * 256 * (size_t)blocks));
  run<0>(src, out, blocks, nstage);
  run<1>(src, out, blocks, nstage);
  run<2>(src, out, blocks, nstage);
  run<3>(src, out, blocks, nstage);
  run<4>(src, out, blocks, nstage);
  run<5>(src, out, blocks, nstage);
  return 0;
}
