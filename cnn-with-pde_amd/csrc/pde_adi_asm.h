// Launcher of the hand-scheduled backward sweep kernels (gen_adi_bwd_asm.py): gfx950 assembly, assembled at build time into
// code objects that are embedded in the library and loaded once per device.
#pragma once
#include <hip/hip_runtime.h>

namespace pde {

// kernel-argument block of adi_bwd_asm_n32_w<NW> (104 bytes; the kernel reads it with two scalar loads)
struct AsmBwdArgs {
    const void* gy;         // upstream gradient (B,C,32,32) fp32
    const void* y;          // layer output
    void* gu;               // gradient w.r.t. the layer input
    const float* coef;      // [S][C][kRecStride] records of the factorisation
    float* part;            // [G][C][4][kImage] partial parameter-gradient sums
    const void* tab;        // SweepTab
    const int* varying;     // [C]: channels with a time-dependent clamp mask are skipped (the masked HIP body owns them)
    int B, C, S, G;
    float gu_scale;         // (1+eps)^-S
    int acc_part;           // flags: bit 0 = add to what `part` holds, bit 1 = the schedule's twin x sweeps (last of a step,
                            // first of the next) have identical records (strang_pairs_identical)
    int cz_mul;             // channel = blockIdx.x + cz_mul * blockIdx.z
    int K;                  // time steps = S / 3
    int nchunk;             // ceil(B / planes per workgroup pass)
    int pad;                // diagnostic stop stage (0 in production)
    void* dbg;              // diagnostic builds: cycle stamps [wave][16] (null otherwise)
};
static_assert(sizeof(AsmBwdArgs) == 104, "kernel-argument layout");

// waves per workgroup of the variant that will run (0: none available / disabled by PDE_ASM_BWD=0)
int asm_bwd_waves();
// planes a workgroup of that variant takes per pass
inline int asm_bwd_planes(int nw) { return 2 * nw; }
// returns PDE_OK / PDE_E_LAUNCH
int asm_bwd_launch(int nw, const AsmBwdArgs& a, int xcd_map, hipStream_t st);

// The forward counterpart (gen_adi_fwd_asm.py: 16 waves, four planes per lane).  Same argument block: gy = the input u,
// y = the output, coef, B, C, S, G, K, nchunk = ceil(B / 64), acc_part bit 1 = twin records; the other fields unused.
bool asm_fwd_enabled();                        // opt-in: PDE_ASM_FWD=1 (no faster than the HIP forward, see pde_adi_asm.hip)
constexpr int kAsmFwdPlanes = 64;              // planes a workgroup takes per pass
int asm_fwd_launch(const AsmBwdArgs& a, int xcd_map, hipStream_t st);

}  // namespace pde
