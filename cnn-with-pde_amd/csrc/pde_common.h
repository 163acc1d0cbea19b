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

// Coefficient record of one (sweep, channel), exactly as it sits in LDS:
//   [JN 32][INV image][E image][KAPX image][MASKX image]
// INV/E are indexed by the sweep's own lines; KAPX (the coefficient itself) and MASKX (1 where
// the clamp lets the gradient through, else 0) always by rows (x layout).
constexpr int kRecJn = 0;
constexpr int kRecInv = 32;
constexpr int kRecE = 32 + kImage;
constexpr int kRecKapX = 32 + 2 * kImage;
constexpr int kRecMaskX = 32 + 3 * kImage;
constexpr int kRecFwd = 32 + 2 * kImage;            // floats staged by the forward
constexpr int kRecBwd = 32 + 3 * kImage;            // floats staged by the backward (constant masks)
constexpr int kRecStride = 32 + 4 * kImage;         // record stride in global memory; staged whole by the masked backward

// position of global index j inside a line image row
__host__ __device__ inline int half_pos(int j, int N) { return (j < N / 2) ? j : kHalfPad + (N - 1 - j); }

struct Timing {
    bool on = false;
    double fwd_ms = 0, bwd_ms = 0;
    long long fwd_n = 0, bwd_n = 0;
};
Timing& timing();

inline int check_launch() { return hipGetLastError() == hipSuccess ? PDE_OK : PDE_E_LAUNCH; }

}  // namespace pde
