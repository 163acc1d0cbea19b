// One translation unit per line length: compile with -DPDE_INST_N=<N>.
#include "pde_adi_dev.h"
#include "pde_adi_launch.h"
#if PDE_INST_N == 16 || PDE_INST_N == 28 || PDE_INST_N == 32
#define PDE_INST_SMALL 1
#include "pde_adi_small.h"
#endif
#if PDE_INST_N == 28 || PDE_INST_N == 32
#define PDE_INST_WIDE 1
#include "pde_adi_wide.h"
#endif

#ifndef PDE_INST_N
#error "compile with -DPDE_INST_N=<line length>"
#endif

#define PDE_CAT2(a, b) a##b
#define PDE_CAT(a, b) PDE_CAT2(a, b)

namespace pde {
namespace {

template <typename K>
int launch(K kernel, const SweepArgs& sa, int grid, size_t lds, hipStream_t st) {
    static unsigned long long configured = 0;       // one static per kernel instantiation
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)lds, configured) != PDE_OK) return PDE_E_LAUNCH;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), lds, st, sa);
    return check_launch();
}

template <typename IO>
int fwd_io(int split, const SweepArgs& sa, int grid, size_t lds, hipStream_t st) {
    constexpr int N = PDE_INST_N;
    switch (split) {
        case kSplitStrang: return launch(adi_fwd_kernel<N, kJFwd, IO, kSplitStrang>, sa, grid, lds, st);
        case kSplitLie: return launch(adi_fwd_kernel<N, kJFwd, IO, kSplitLie>, sa, grid, lds, st);
        default: return launch(adi_fwd_kernel<N, kJFwd, IO, kSplitAny>, sa, grid, lds, st);
    }
}

template <bool MASKED, int SPLIT>
constexpr size_t bwd_lds() { return (size_t)(BwdStage<MASKED, SPLIT>::kFloats + kWaves * kImage) * sizeof(float); }
template <int SPLIT>
constexpr size_t bwd_lds_dual() {
    return bwd_lds<false, SPLIT>() > bwd_lds<true, kSplitAny>() ? bwd_lds<false, SPLIT>() : bwd_lds<true, kSplitAny>();
}

// grid = 2 * C * G: fast and masked variant in one launch (see adi_bwd_kernel)
template <typename IO>
int bwd_io(int split, const SweepArgs& sa, int grid, hipStream_t st) {
    constexpr int N = PDE_INST_N;
    switch (split) {
        case kSplitStrang: return launch(adi_bwd_kernel<N, kJBwd, IO, kSplitStrang>, sa, grid, bwd_lds_dual<kSplitStrang>(), st);
        case kSplitLie: return launch(adi_bwd_kernel<N, kJBwd, IO, kSplitLie>, sa, grid, bwd_lds_dual<kSplitLie>(), st);
        default: return launch(adi_bwd_kernel<N, 1, IO, kSplitAny>, sa, grid, bwd_lds_dual<kSplitAny>(), st);
    }
}

}  // namespace

int PDE_CAT(adi_launch_fwd_, PDE_INST_N)(int io, int split, const void* args, int grid, size_t lds, hipStream_t st) {
    const SweepArgs& sa = *static_cast<const SweepArgs*>(args);
    return io == PDE_IO_F32 ? fwd_io<float>(split, sa, grid, lds, st) : fwd_io<bf16_t>(split, sa, grid, lds, st);
}

int PDE_CAT(adi_launch_bwd_, PDE_INST_N)(int io, int split, const void* args, int grid, hipStream_t st) {
    const SweepArgs& sa = *static_cast<const SweepArgs*>(args);
    return io == PDE_IO_F32 ? bwd_io<float>(split, sa, grid, st) : bwd_io<bf16_t>(split, sa, grid, st);
}

#ifdef PDE_INST_SMALL
int PDE_CAT(adi_launch_small_fwd_, PDE_INST_N)(int io, int split, const void* args, int grid, size_t lds, hipStream_t st) {
    const SmallArgs& sa = *static_cast<const SmallArgs*>(args);
    return io == PDE_IO_F32 ? small_fwd_io<PDE_INST_N, float>(split, sa, grid, lds, st)
                            : small_fwd_io<PDE_INST_N, bf16_t>(split, sa, grid, lds, st);
}
int PDE_CAT(adi_launch_small_bwd_, PDE_INST_N)(int io, int split, const void* args, int grid, size_t lds, hipStream_t st) {
    const SmallArgs& sa = *static_cast<const SmallArgs*>(args);
    return io == PDE_IO_F32 ? small_bwd_io<PDE_INST_N, float>(split, sa, grid, lds, st)
                            : small_bwd_io<PDE_INST_N, bf16_t>(split, sa, grid, lds, st);
}
#endif

#ifdef PDE_INST_WIDE
int PDE_CAT(adi_launch_wide_fwd_, PDE_INST_N)(int C, int split, const void* args, int grid, hipStream_t st) {
    return wide_fwd_launch<PDE_INST_N>(C, split, *static_cast<const WideArgs*>(args), grid, st);
}
#endif

}  // namespace pde
