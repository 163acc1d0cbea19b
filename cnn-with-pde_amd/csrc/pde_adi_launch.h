// Launch entry points of the per-N instantiation units (pde_adi_inst.hip, one object per
// line length).  Plain functions so that every unit can be compiled in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace pde {
struct SweepArgsOpaque;          // = SweepArgs of pde_adi_dev.h, passed by pointer across units

#ifndef PDE_JF
#define PDE_JF 4
#endif
#ifndef PDE_JB
#define PDE_JB 2
#endif
constexpr int kJFwd = PDE_JF;    // planes per lane in the forward kernel
constexpr int kJBwd = PDE_JB;    // planes per lane in the backward kernel

// return 0 on success, PDE_E_LAUNCH otherwise
#define PDE_DECLARE_N(NN)                                                                              \
    int adi_launch_fwd_##NN(int io, int split, const void* args, int grid, size_t lds, hipStream_t st); \
    int adi_launch_bwd_##NN(int io, int split, const void* args, int grid, hipStream_t st);
PDE_DECLARE_N(8) PDE_DECLARE_N(12) PDE_DECLARE_N(16) PDE_DECLARE_N(20) PDE_DECLARE_N(24) PDE_DECLARE_N(28) PDE_DECLARE_N(32)
#undef PDE_DECLARE_N

// whole-layer kernels for C <= 4 (pde_adi_small.h), instantiated for the line lengths below
#define PDE_SMALL_N_LIST PDE_SMALL_CASE(16) PDE_SMALL_CASE(28) PDE_SMALL_CASE(32)
#define PDE_SMALL_CASE(NN)                                                                                \
    int adi_launch_small_fwd_##NN(int io, int split, const void* args, int grid, size_t lds, hipStream_t st); \
    int adi_launch_small_bwd_##NN(int io, int split, const void* args, int grid, size_t lds, hipStream_t st);
PDE_SMALL_N_LIST
#undef PDE_SMALL_CASE

// whole-layer forward for C = 32 / 64 fp32 (pde_adi_wide.h)
#define PDE_WIDE_N_LIST PDE_WIDE_CASE(28) PDE_WIDE_CASE(32)
#define PDE_WIDE_CASE(NN) int adi_launch_wide_fwd_##NN(int C, int split, const void* args, int grid, hipStream_t st);
PDE_WIDE_N_LIST
#undef PDE_WIDE_CASE
}  // namespace pde
