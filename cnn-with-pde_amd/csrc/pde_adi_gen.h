// K1 for any line length (pde_adi_gen.hip): what pde_adi.hip's entry points call when the fused kernels do not cover N.
#pragma once
#include "pde_common.h"

namespace pde {

bool gen_n_ok(int N);                                                       // 2 <= N <= PDE_MAX_N_GENERIC
size_t gen_forward_workspace_bytes(const PdeAdiDesc* d);                    // factorisation + device copy of the schedule
size_t gen_backward_workspace_bytes(const PdeAdiDesc* d, int num_checkpoints);
// per-sweep coefficient maxima alone
int gen_kappa_max(const PdeAdiDesc* d, const float* ab, const float* bb, const float* as, const float* bs, float* kmax,
                  hipStream_t st);
// factorise every sweep into `workspace` (and the maxima into kmax when not null)
int gen_factor(const PdeAdiDesc* d, const float* ab, const float* bb, const float* as, const float* bs, float* kmax,
               void* workspace, hipStream_t st);
// all sweeps of d on u -> y with the factorisation in `workspace`
int gen_forward_sweeps(const PdeAdiDesc* d, const void* u, void* y, const void* workspace, hipStream_t st);
// adjoint + the four parameter gradients; nck / Sf from the checkpoint mask as in the fused path
int gen_backward(const PdeAdiDesc* d, const void* gy, const void* y, const void* u, const uint64_t ckpt_mask[2], int nck,
                 int Sf, void* gu, const float* ab, const float* bb, const float* as, const float* bs, float* g_ab,
                 float* g_bb, float* g_as, float* g_bs, const void* fwd_workspace, void* workspace, hipStream_t st);

}  // namespace pde
