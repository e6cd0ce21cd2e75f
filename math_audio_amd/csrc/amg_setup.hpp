// Host side of AmgPreconditioner::from_csr (math-solvers/src/preconditioners/amg.rs:276-372): the hierarchy is built on the host,
// as in the reference, and handed to the device V-cycle of op_plan.hip.
#pragma once
#include <cstdint>
#include <vector>
#include "ma_common.hpp"
#include "../../include/mathaudio_hip.h"

namespace ma {

struct HostCsr {                                 // CsrMatrix<Complex64> (sparse/csr.rs:18-35)
  int64_t nr = 0, nc = 0;
  std::vector<int64_t> ptr, col;
  std::vector<c64> val;
};

// levels[0].A is the input; for l < levels-1, P[l] (n_l x n_{l+1}) and R[l] = P[l]^T; complexities of amg.rs:837-853
int amg_setup_host(const HostCsr& A, const ma_amg_config_t& cfg, std::vector<HostCsr>& As, std::vector<HostCsr>& Ps, std::vector<HostCsr>& Rs,
                   double* grid_complexity, double* operator_complexity);

}  // namespace ma
