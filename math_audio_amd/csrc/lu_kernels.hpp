// lu_kernels.hpp — launchers of the dense LU kernels.
#pragma once
#include "ma_common.hpp"

#define LU_NB_MAX 128
#define LU_REG_NB 32            /* panel width of lu_panel_reg_kernel: a row's entries live in 4 * LU_REG_NB vector registers */
#define LU_BATCH_MAX 8          /* systems (slots) a plan keeps resources for */
#define LU_GROUP_MAX 4          /* lock-step batch size of the public API */

namespace ma {

// Global scratch shared by the co-resident workgroups of lu_panel_kernel. `counter` lives in a
// 16-byte block that is zeroed before every launch (the arrival counter is monotonic within a
// launch); `info` (first zero pivot, 1-based; 0 = none) and `timeout` persist over a factorisation.
struct LuPanelWs {
  unsigned* counter;            // arrivals
  unsigned* timeout;            // the plan's poison word: 1 an exchange wait expired (or the test hook fired), 2 lu_perm_kernel met a pivot
                                // outside [k0 + c, n); every poll of every panel kernel of the plan reads it and leaves when it is set
  int* info;
  unsigned long long* cand;     // [2][max_blocks] granules {|re|+|im| bits, tag, row}, LU_GRANULE_STRIDE words apart
  unsigned long long* candrow;  // [2][max_blocks][2*LU_NB_MAX] candidate row of the panel
  unsigned long long* diagrow;  // [2][2*LU_NB_MAX]            current diagonal row of the panel
  int max_blocks;
  int test_abort_col;           // diagnostic build only (MA_LU_TEST_ABORT_COL): the last workgroup gives up at this global column; -1 = off
};

// Admission of a kernel whose workgroups wait for one another (all of them must be resident): see "Residency" in lu_kernels.hip.
//   SpinLaunch g; if ((rc = g.admit(stream, workgroups, lds bytes, registers per lane, CUs the stream may use))) return rc;
//   <launch>; return g.commit();
// admit() refuses (MA_ERR_UNSUPPORTED) a grid that cannot be co-resident on its own, and makes the stream wait for older spinning
// grids of other streams until everything in flight fits; the sequencer stays locked until commit() / destruction.
struct SpinLaunch {
  bool locked = false; int dev = 0; hipStream_t st = nullptr; int nblk = 0; size_t lds = 0; int regs = 0; int ncu = 0;
  int admit(hipStream_t st, int nblk, size_t lds, int regs, int ncu);
  int commit();
  void abandon();
  ~SpinLaunch();
};
unsigned* spin_error_word();            // device word raised by any spinning kernel that abandons a wait (NULL if it cannot be allocated)
int spin_error_check(const char* what); // MA_ERR_HIP (and clears the word) if it is set; synchronous 4-byte copy

size_t lu_panel_granule_bytes(int max_blocks);
// the spinning partial-pivoting panel kernel (rows in registers, 256 rows per workgroup, <= LU_REG_NB columns); run_if_nonzero: a device
// word, 0 = the kernel returns at once (the speculative panel ahead of it was accepted)
int lu_launch_panel_reg(c64* A, int n, int k0, int nb, int nblk, int ncu, const LuPanelWs& ws, int* ipiv, int* lists, bool clear_tags, hipStream_t st,
                        c64* lrows = nullptr, int lcol0 = 0, const int* run_if_nonzero = nullptr);
int lu_launch_lane_step2(c64* A, int n, int k0, int nb, const int* lists1, const int* lists2, int x0, int ncols, const int* ipiv, int* lists64, c64* invd, unsigned* poison,
                         const c64* l10, hipStream_t st);
int lu_launch_lane_step(c64* A, int n, int k0, int nb, const int* lists, int x0, int ncols, c64* invd, const unsigned* poison, hipStream_t st);
int lu_panel_reg_admissible(int nblk, int ncu);
// Tournament pivoting (lu_calu.hip): the same panel, the same outputs, no workgroup waits for another. Per slot: cand[nodes][LU_REG_NB]
// row indices and one arrival counter per tree node (zero between launches).
struct LuCaluWs { int* cand = nullptr; unsigned* counters = nullptr; int max_nodes = 0; };
int lu_calu_tree_nodes(int leaves);
int lu_launch_panel_calu(c64* A, int n, int k0, int nb, const LuCaluWs& ws, int* info, int* ipiv, int* lists, hipStream_t st, c64* lrows = nullptr, int lcol0 = 0,
                         const int* run_if_nonzero = nullptr);   // run_if_nonzero: a device word; 0 = the kernels return at once (the speculative panel was accepted)
// Speculative panel (lu_spec.hip): partial pivoting restricted to the panel's top 32 rows, VERIFIED against every row below -- accepted, it
// is LAPACK's factorisation of the panel in two launches without any exchange between workgroups; rejected (verdict word != 0), the
// panel's columns are restored and the caller's fallback panel (launched with run_if_nonzero = the verdict word) factors it.
struct LuSpecWs { c64* u11 = nullptr; c64* rinv = nullptr; double* pivmag = nullptr; int* ctl = nullptr; /* [0] the verdict: 0 = factored */ int* vlist = nullptr; int* ext = nullptr;
                  c64* backup = nullptr; int rows = 0;
                  unsigned long long* stats = nullptr; /* the plan's counters: [0] half-panels tried, [1] rejected by the first attempt, [2] of those accepted by the widened one */ };
int lu_launch_panel_spec(c64* A, int n, int k0, int nb, const LuSpecWs& ws, int* ipiv, int* lists, hipStream_t st, c64* lrows = nullptr, int lcol0 = 0,
                         bool optimistic = false, int* reject_info = nullptr);
void lu_panel_forget_stream(int dev, hipStream_t st);
int lu_launch_perm(const c64* A, int n, int k0, int nb, const int* ipiv, int* lists, c64* invd, unsigned* poison, hipStream_t st);
int lu_panel_slots_per_cu(size_t lds, int regs);
int lu_panel_regs();           /* vector registers per lane of lu_panel_reg_kernel */
int lu_launch_row_moves(c64* A, int n, int nb, const int* lists, c64* tmp, int tstride, int x0, int x1, int y0, int y1, c64* B, int nrhs, hipStream_t st);
int lu_launch_swaps(c64* A, int n, int k0, int nb, const int* ipiv, int* lists, c64* tmp, int tstride, int x0, int x1, int y0, int y1, c64* B, int nrhs,
                    c64* invd, unsigned* poison, hipStream_t st);
int lu_launch_trsm_mfma(const c64* T, int ldt, int nb, const c64* invd, c64* X, size_t ldx, int ncols, c64* B, size_t ldb, int nrhs, hipStream_t st);
int lu_trsm_configure();
int lu_launch_trsv(bool upper, const c64* T, int ldt, int nb, c64* B, size_t ldb, int nrhs, hipStream_t st);
int lu_launch_zgemm_sub(int M, int N, int K, const c64* A, size_t lda, const c64* B, size_t ldb, c64* C, size_t ldc, hipStream_t st, bool big = false, bool dma = true);
int lu_launch_zgemv_sub(int M, int K, const c64* A, size_t lda, const c64* x, c64* y, hipStream_t st);
int lu_launch_mfma_probe(double* out, int blocks, int iters, hipStream_t st);
int lu_cumask_selfcheck(hipStream_t masked, int expect_cus, bool* ok);   // the masked stream uses exactly expect_cus CUs, spread evenly over the 8 XCDs

}  // namespace ma
