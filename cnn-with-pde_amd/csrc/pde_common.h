// Shared device/host helpers for libpdecnn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pdecnn.h"

namespace pde {

// ---- LDS image geometry shared by the factor kernel and the sweep kernels ---------
// A "line image" holds one scalar per (line, position) of a <=32x32 plane as
// [32 lines][36 floats]: the first 16 floats of a row are the half-line seen from
// the low end (k = 0 at index 0 ... k = m-1), the next 16 the half-line seen from
// the high end (k = 0 at index N-1), 4 floats of padding.  The 144-byte row stride
// makes a ds_read_b128 by 16 consecutive lines hit 16 distinct 16-byte bank slots.
constexpr int kLineStride = 36;
constexpr int kImage = 32 * kLineStride;            // 1152 floats = 4608 B
constexpr int kHalfPad = 16;

// Coefficient record of one (sweep, channel) in global memory, stored exactly as it sits in LDS:
//   [INV image][JN 32][E image][INVB image][KAPX image][MASKX image]
// INV, E, INVB are indexed by the sweep's own lines; KAPX and MASKX always by rows (x layout).
//   INV   1/den of the two-sided factorisation            (forward)
//   JN    1/(1 - e_t e_b), the junction factor per line   (forward, backward)
//   E     kap/den                                          (forward, backward)
//   INVB  (1+eps)/den: the adjoint solve carries one factor (1+eps) per sweep so that the state
//         can be rebuilt with ONE fma, y_{s-1} = y_s + KAPX (L y_s), on the rescaled state
//         y_s = x_s (1+eps)^-(S-1-s)        (backward)
//   KAPX  kap/(1+eps)                                      (backward)
//   MASKX 1 where the clamp lets the gradient through      (masked backward only)
// The forward stages the prefix [INV|JN|E]; the backward the suffix starting at JN.
constexpr int kG_Inv = 0;
constexpr int kG_Jn = kImage;
constexpr int kG_E = kImage + 32;
constexpr int kG_InvB = 2 * kImage + 32;
constexpr int kG_KapX = 3 * kImage + 32;
constexpr int kG_MaskX = 4 * kImage + 32;
constexpr int kRecStride = 5 * kImage + 32;          // record stride in global memory (floats)
// forward view (staged from kG_Inv)
constexpr int kRecFwd = 2 * kImage + 32;
constexpr int kRecFwdPad = ((kRecFwd / 4 + 63) / 64) * 256;   // LDS slot: whole 1-KB LDS-DMA pieces
constexpr int kF_Inv = 0, kF_Jn = kImage, kF_E = kImage + 32;
// backward view (staged from kG_Jn)
constexpr int kBwdOff = kG_Jn;
constexpr int kRecBwd = 3 * kImage + 32;             // JN, E, INVB, KAPX
constexpr int kRecBwdMasked = 4 * kImage + 32;       // ... + MASKX
constexpr int kB_Jn = 0, kB_E = 32, kB_Inv = kImage + 32, kB_KapX = 2 * kImage + 32, kB_MaskX = 3 * kImage + 32;

// Lane-major copy of a record for the kernels that read coefficients straight from memory (pde_adi_wide.h): every
// image as [4 quads][64 lanes][4 floats] — lane (line, half) finds its 16 values of an image at 4 x 16 bytes that are
// CONTIGUOUS ACROSS LANES (a wave-level 16-byte load covers 1 KB; the [32][36] image gives 64 different cache lines).
//   [JN 64 (32 used)][INV][E][INVB][KAPX], INV/E/INVB by the sweep's own lines, KAPX by rows, as above
constexpr int kWideImage = 4 * 64 * 4;
constexpr int kW_Jn = 0, kW_Inv = 64, kW_E = 64 + kWideImage, kW_InvB = 64 + 2 * kWideImage, kW_KapX = 64 + 3 * kWideImage;
constexpr int kWideRec = 64 + 4 * kWideImage;         // 4160 floats

// position of global index j inside a line image row
__host__ __device__ inline int half_pos(int j, int N) { return (j < N / 2) ? j : kHalfPad + (N - 1 - j); }

// bf16 tensors through the bf16 MFMA (pde_mix_bf16.hip); C = 64 / 128, HW a multiple of 64
bool mix_bf16_ok(int C, int HW);
int mix_bf16_splits(int B, int C, int HW);
int mix_bf16_apply(int B, int C, int HW, const void* u, const float* M, void* out, int trans, hipStream_t st);
int mix_bf16_backward(int B, int C, int HW, const void* u, const void* g, const float* M, void* gu, float* part, int nsplit,
                      int accp, hipStream_t st);

// fp32 tensors through the bf16 MFMA with every operand as three bf16 pieces (pde_mix_bf16.hip); C = 64, HW a multiple of 64
bool mix_split_ok(int C, int HW);
int mix_split_splits(int B, int C, int HW);
int mix_split_backward(int B, int C, int HW, const void* u, const void* g, const float* M, void* gu, float* part, int nsplit,
                       int accp, hipStream_t st);

// fp32 -> bf16, round to nearest even: one v_cvt_pk_bf16_f32 per pair on gfx950 (the bit-twiddling form costs
// five VALU instructions per element, and a VALU instruction is what the sweep kernels run out of)
__device__ __forceinline__ unsigned short f32_to_bf16_hw(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

struct Timing {
    bool on = false;
    double fwd_ms = 0, bwd_ms = 0;
    long long fwd_n = 0, bwd_n = 0;
};
Timing& timing();

// Peek, not Get: the caller's framework (PyTorch) owns the thread's sticky error state; the library only
// reports that one of its launches failed and leaves the state for the owner to read.
inline int check_launch() { return hipPeekAtLastError() == hipSuccess ? PDE_OK : PDE_E_LAUNCH; }

// The dynamic-LDS limit is a per-device attribute of a kernel: set it once per (kernel, device).  `done` is the
// caller's per-kernel bit mask (bit d: configured on device d); one process-wide mutex guards all of them
// (autograd runs one backward thread per device, so two devices may arrive here together).
int ensure_dynamic_lds(const void* kernel, int bytes, unsigned long long& done);

}  // namespace pde
