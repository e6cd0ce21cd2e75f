// Diagnostic: tile-shape variants of the 3-product complex trailing update (C -= A B) on v_mfma_f64_16x16x4_f64.
// build: hipcc -O3 --offload-arch=gfx950 -I math_audio_amd/csrc tools/zgemm_variants.hip -L math_audio_amd/lib -lmathaudio_hip -o tools/zgemm_variants.bin
#include "lu_kernels.hpp"
#include "ma_device_math.hpp"
#include <cstdio>
#include <vector>
using namespace ma;
typedef double v4d __attribute__((ext_vector_type(4)));

template <int BM, int BN, int BK, int PAD, int WPE>
__global__ __launch_bounds__(256, WPE) void z3_kernel(int M, int N, int K, const dc* __restrict__ A, size_t lda,
                                                                                           const dc* __restrict__ B, size_t ldb, dc* __restrict__ C, size_t ldc) {
  constexpr int TA = BM / 32, TB = BN / 32;
  constexpr int NA = BM * BK / 256, NB = BK * BN / 256;
  __shared__ __attribute__((aligned(16))) dc As[2][BK][BM + ((PAD == 1 || PAD == 4) ? 1 : 0)];
  __shared__ __attribute__((aligned(16))) dc Bs[2][BK][BN + ((PAD == 1 || PAD == 3) ? 1 : 0)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (PAD == 5) {   // XCD-aware order: 1D grid, XCD x (= bid % 8) owns a contiguous run of 8x8-tile super-blocks
    const int TX = (N + BN - 1) / BN, TY = (M + BM - 1) / BM;
    const int SBX = (TX + 7) / 8;
    const int total = gridDim.x, chunk = total / 8;
    const int bid = blockIdx.x;
    const int L = (bid % 8) * chunk + bid / 8;
    const int sb = L >> 6, w = L & 63;
    const int tx = (sb % SBX) * 8 + (w & 7), ty = (sb / SBX) * 8 + (w >> 3);
    if (tx >= TX || ty >= TY) return;
    m0 = ty * BM; n0 = tx * BN;
  }
  const int li = lane & 15, lk = lane >> 4;
  v4d t1[TA][TB], t2[TA][TB], t3[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) { t1[a][b] = (v4d){0, 0, 0, 0}; t2[a][b] = (v4d){0, 0, 0, 0}; t3[a][b] = (v4d){0, 0, 0, 0}; }
  dc ra[NA], rb[NB];
  auto load_stage = [&](int k0) {
#pragma unroll
    for (int s = 0; s < NA; ++s) {
      const int e = tid + 256 * s;
      const int row = e / BK, kk = e % BK;
      const int gm = m0 + row, gk = k0 + kk;
      ra[s] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : dc_make(0.0, 0.0);
    }
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      const int e = tid + 256 * s;
      const int bk = e / BN, bn = e % BN;
      const int gn = n0 + bn, gk2 = k0 + bk;
      rb[s] = (gn < N && gk2 < K) ? B[(size_t)gk2 * ldb + gn] : dc_make(0.0, 0.0);
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int s = 0; s < NA; ++s) { const int e = tid + 256 * s; As[buf][e % BK][(PAD == 2 || PAD == 3 || PAD == 5) ? ((e / BK) ^ (e % BK)) : (e / BK)] = ra[s]; }
#pragma unroll
    for (int s = 0; s < NB; ++s) { const int e = tid + 256 * s; Bs[buf][e / BN][e % BN] = rb[s]; }
  };
  const int nstage = (K + BK - 1) / BK;
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    if (st + 1 < nstage) load_stage((st + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      const int kk = ks * 4 + lk;
      dc af[TA], bf[TB];
      double as[TA], bs[TB];
#pragma unroll
      for (int a = 0; a < TA; ++a) { af[a] = As[buf][kk][(PAD == 2 || PAD == 3 || PAD == 5) ? ((wm * (BM / 2) + a * 16 + li) ^ kk) : (wm * (BM / 2) + a * 16 + li)]; as[a] = af[a].re + af[a].im; }
#pragma unroll
      for (int b = 0; b < TB; ++b) { bf[b] = Bs[buf][kk][wn * (BN / 2) + b * 16 + li]; bs[b] = bf[b].re + bf[b].im; }
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) {
          t1[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].re, bf[b].re, t1[a][b], 0, 0, 0);
          t2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].im, bf[b].im, t2[a][b], 0, 0, 0);
          t3[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[a], bs[b], t3[a][b], 0, 0, 0);
        }
    }
    if (st + 1 < nstage) store_stage(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm * (BM / 2) + a * 16 + lk + 4 * r;
        const int gn = n0 + wn * (BN / 2) + b * 16 + li;
        if (gm < M && gn < N) {
          dc* pc = C + (size_t)gm * ldc + gn;
          dc c = *pc;
          const double p1 = t1[a][b][r], p2 = t2[a][b][r];
          c.re -= p1 - p2; c.im -= t3[a][b][r] - p1 - p2;
          *pc = c;
        }
      }
}

template <int BM, int BN, int BK, int PAD, int WPE>
static void run(const char* name, int n, int K, const c64* pa, const c64* pb, c64* pc, size_t ld, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  dim3 grid((n + BN - 1) / BN, (n + BM - 1) / BM);
  if (PAD == 5) { const int SBX = ((n + BN - 1) / BN + 7) / 8, SBY = ((n + BM - 1) / BM + 7) / 8; grid = dim3(((SBX * SBY * 64 + 7) / 8) * 8, 1); }
  auto go = [&] { hipLaunchKernelGGL((z3_kernel<BM, BN, BK, PAD, WPE>), grid, dim3(256), 0, st, n, n, K, (const dc*)pa, ld, (const dc*)pb, ld, (dc*)pc, ld); };
  for (int it = 0; it < 2; ++it) go();
  hipEventRecord(e0, st);
  const int reps = 5;
  for (int it = 0; it < reps; ++it) go();
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("%-22s n=%d K=%3d  %.3f ms  %.1f TFLOP/s  (%s)\n", name, n, K, ms, 8.0 * n * (double)n * K / ms * 1e-9, hipGetErrorString(hipGetLastError()));
}

// software-pipelined variant: fragments of the second k-slice are read under the first slice's products, the next
// stage is written to LDS mid-stage and its first fragments are read under the second slice's products.
template <int WPE>
__global__ __launch_bounds__(256, WPE) void z3p_kernel(int M, int N, int K, const dc* __restrict__ A, size_t lda,
                                                     const dc* __restrict__ B, size_t ldb, dc* __restrict__ C, size_t ldc) {
  constexpr int BM = 64, BN = 64, BK = 8;
  __shared__ __attribute__((aligned(16))) dc As[2][BK][BM];
  __shared__ __attribute__((aligned(16))) dc Bs[2][BK][BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int li = lane & 15, lk = lane >> 4;
  v4d t1[2][2], t2[2][2], t3[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) { t1[a][b] = (v4d){0, 0, 0, 0}; t2[a][b] = (v4d){0, 0, 0, 0}; t3[a][b] = (v4d){0, 0, 0, 0}; }
  dc ra[2], rb[2];
  auto load_stage = [&](int k0) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int e = tid + 256 * s;
      const int row = e >> 3, kk = e & 7;
      const int gm = m0 + row, gk = k0 + kk;
      ra[s] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : dc_make(0.0, 0.0);
      const int bk = e >> 6, bn = e & 63;
      const int gn = n0 + bn, gk2 = k0 + bk;
      rb[s] = (gn < N && gk2 < K) ? B[(size_t)gk2 * ldb + gn] : dc_make(0.0, 0.0);
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int e = tid + 256 * s;
      As[buf][e & 7][(e >> 3) ^ (e & 7)] = ra[s];
      Bs[buf][e >> 6][e & 63] = rb[s];
    }
  };
  dc af0[2], bf0[2], af1[2], bf1[2];
  auto read_frag = [&](int buf, int ks, dc (&af)[2], dc (&bf)[2]) {
    const int kk = ks * 4 + lk;
#pragma unroll
    for (int a = 0; a < 2; ++a) af[a] = As[buf][kk][(wm * 32 + a * 16 + li) ^ kk];
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[b] = Bs[buf][kk][wn * 32 + b * 16 + li];
  };
  auto products = [&](const dc (&af)[2], const dc (&bf)[2]) {
    double as[2], bs[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) as[a] = af[a].re + af[a].im;
#pragma unroll
    for (int b = 0; b < 2; ++b) bs[b] = bf[b].re + bf[b].im;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        t1[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].re, bf[b].re, t1[a][b], 0, 0, 0);
        t2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].im, bf[b].im, t2[a][b], 0, 0, 0);
        t3[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[a], bs[b], t3[a][b], 0, 0, 0);
      }
  };
  const int nstage = (K + BK - 1) / BK;
  load_stage(0);
  store_stage(0);
  if (nstage > 1) load_stage(BK);
  __syncthreads();
  read_frag(0, 0, af0, bf0);
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    read_frag(buf, 1, af1, bf1);
    __builtin_amdgcn_sched_barrier(0);
    products(af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    if (st + 1 < nstage) store_stage(buf ^ 1);
    __syncthreads();
    if (st + 2 < nstage) load_stage((st + 2) * BK);
    if (st + 1 < nstage) read_frag(buf ^ 1, 0, af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    products(af1, bf1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm * 32 + a * 16 + lk + 4 * r;
        const int gn = n0 + wn * 32 + b * 16 + li;
        if (gm < M && gn < N) {
          dc* pc = C + (size_t)gm * ldc + gn;
          dc c = *pc;
          const double p1 = t1[a][b][r], p2 = t2[a][b][r];
          c.re -= p1 - p2; c.im -= t3[a][b][r] - p1 - p2;
          *pc = c;
        }
      }
}

template <int WPE>
static void runp(const char* name, int n, int K, const c64* pa, const c64* pb, c64* pc, size_t ld, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  dim3 grid((n + 63) / 64, (n + 63) / 64);
  auto go = [&] { hipLaunchKernelGGL((z3p_kernel<WPE>), grid, dim3(256), 0, st, n, n, K, (const dc*)pa, ld, (const dc*)pb, ld, (dc*)pc, ld); };
  for (int it = 0; it < 2; ++it) go();
  hipEventRecord(e0, st);
  const int reps = 5;
  for (int it = 0; it < reps; ++it) go();
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("%-22s n=%d K=%3d  %.3f ms  %.1f TFLOP/s  (%s)\n", name, n, K, ms, 8.0 * n * (double)n * K / ms * 1e-9, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 10000;
  c64* A;
  const size_t ld = (size_t)n + 512;
  hipMalloc(&A, sizeof(c64) * ld * ld);
  {
    std::vector<double> h(2 * ld * 64);
    unsigned long long sd = 88172645463325252ull;
    for (auto& v : h) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; v = ((double)(sd >> 11) / 9007199254740992.0 - 0.5) * 2e-3; }
    for (size_t r = 0; r < ld; r += 64) hipMemcpy(A + r * ld, h.data(), sizeof(c64) * ld * (r + 64 <= ld ? 64 : ld - r), hipMemcpyHostToDevice);
  }
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int K : {64, 128, 256}) {
    const c64* pa = A + (size_t)K * ld; const c64* pb = A + K; c64* pc = A + (size_t)K * ld + K;
    run<64, 64, 8, 1, 2>("(warm-up)", n, K, pa, pb, pc, ld, st, e0, e1);
    run<64, 64, 8, 2, 2>("64x64 swz A", n, K, pa, pb, pc, ld, st, e0, e1);
    run<64, 64, 8, 5, 2>("64x64 swz A xcd-remap", n, K, pa, pb, pc, ld, st, e0, e1);
  }
  return 0;
}
