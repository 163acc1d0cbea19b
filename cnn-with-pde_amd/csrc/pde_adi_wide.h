// Whole-layer forward for the implicit layers with a channel operator between the time steps at C = 32 / 64 fp32
// channels: cifar10.EnhancedDiffusionLayer (u <- M u BEFORE every step, cifar10.py:84-112) and SVHN.DiffusionLayer
// (u <- K u AFTER every step, SVHN.py:55-72) at the widths of BASELINE.json's cfg2 (with mixing) and cfg3-32ch.
//
// The per-step path (pde_adi_mixed_forward's loop: one MFMA mixing launch + one sweep launch per step) moves every
// plane through HBM four times per time step.  Here ONE workgroup owns ALL channels of a sample for the whole time
// loop: C/4 waves, wave w holds channels 4w..4w+3 in registers (4 x N/2 VGPRs, the row layout of pde_adi_dev.h), so a
// plane is read once and each step writes only what the backward keeps.
//
//   sweeps   per channel, exactly the code of the other kernels (solve_fwd_rows / relayout).  A workgroup needs the
//            records of all C channels of a sweep (C x 9.3 KB: they do not fit in LDS and nothing is shared between
//            the planes of a wave any more), so every lane loads its own 2 x 64 bytes of (E, INV) per plane and
//            sweep straight from global memory (L2: all workgroups walk the same records at about the same time),
//            from the lane-major copy of the records the factor kernel writes for this path (pde_common.h).
//   mixing   v_mfma_f32_32x32x2_f32 through an LDS exchange image, half of a lane's elements at a time (the whole
//            sample, 4 KB x C, is larger than LDS): owners write float4 = (their 4 channels) per (element, lane);
//            wave w multiplies the [C x C] operator into 2 tiles of [32 channels x 32 pixels] (B operand: one
//            ds_read_b128 per 4 k-steps, the k order permuted so that lanes 32-63 read the next channel group;
//            A operand: fragments of the operator pre-arranged in LDS) and writes the products back in place; owners
//            read their float4s back.  4 workgroup barriers per mixing.  A wave's private re-layout image lies in
//            its own channel group's slice of the exchange image, which nobody else touches outside the MFMA phase.
#pragma once
#include "pde_adi_small.h"

namespace pde {
namespace {

struct WideArgs {
    const void* u;          // (B,C,N,N)
    void* states;           // [K][2][B][C][N][N]: the slots of pde_adi_mixed_forward; only the sweep output of every
                            // step is written (slot 2k+1 for mode 1, 2k for mode 2) and the layer output (2K-1)
    const float* coef;      // [S][C][kWideRec]: lane-major records (pde_common.h)
    const float* M;         // [C][C]
    int B, K, mode, keep;   // keep = 0: inference, only the layer output is written
    void* last;             // nullptr, or where the layer output (slot 2K-1) goes instead
};

typedef float wide_f32x16 __attribute__((ext_vector_type(16)));
constexpr int kWideGroup = 8 * 64 * 4;                    // floats of one channel group's slice of the exchange image

// PDE_WIDE_SPLIT (default 1): the operator product on the bf16 matrix cores with every fp32 operand as three bf16 pieces
// (pde_mix_bf16.hip: six exact piece products per product, fp32-level accuracy, 6/16 of the fp32 MFMA's time); 0: the
// fp32 MFMA (v_mfma_f32_32x32x2_f32).
#ifndef PDE_WIDE_SPLIT
#define PDE_WIDE_SPLIT 1
#endif
typedef short wide_v8s __attribute__((ext_vector_type(8)));
typedef __bf16 wide_v8bf __attribute__((ext_vector_type(8)));

template <int WAVES> constexpr size_t wide_lds_bytes() {
    // exchange image + operator fragments: fp32 [CT][KP][64][4], or three bf16 pieces [3][CT][KS][64][8]
    return (size_t)WAVES * kWideGroup * sizeof(float) +
           (PDE_WIDE_SPLIT ? (size_t)3 * (4 * WAVES / 32) * (4 * WAVES / 16) * 64 * 16
                           : (size_t)(4 * WAVES / 32) * (4 * WAVES / 8) * 256 * sizeof(float));
}

__device__ __forceinline__ void wide_split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    hi = f32_to_bf16_hw(x);
    const float r1 = x - __uint_as_float((unsigned)hi << 16);
    mid = f32_to_bf16_hw(r1);
    lo = f32_to_bf16_hw(r1 - __uint_as_float((unsigned)mid << 16));
}
// eight fp32 values (two float4) -> three bf16 operand fragments
__device__ __forceinline__ void wide_split8(const float4& x0, const float4& x1, wide_v8bf (&pc)[3]) {
    const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    wide_v8s s[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned short a, b, c;
        wide_split3(v[j], a, b, c);
        s[0][j] = (short)a; s[1][j] = (short)b; s[2][j] = (short)c;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) pc[q] = __builtin_bit_cast(wide_v8bf, s[q]);
}
// operator fragments for the bf16 MFMA: Af3[piece][ct][ks][lane] = 8 bf16: Op[32 ct + (lane & 31)][16 ks + 8 (lane >> 5) + j]
template <int WAVES>
__device__ __forceinline__ void wide_fill_operator3(float* Af, const float* Mg, int tid) {
    constexpr int C = 4 * WAVES, KS = C / 16, CT = C / 32, FR = CT * KS * 64;
    uint4* dst = reinterpret_cast<uint4*>(Af);
    for (int e = tid; e < FR; e += 64 * WAVES) {
        const int ln = e & 63, ks = (e >> 6) % KS, ct = (e >> 6) / KS;
        const int i = 32 * ct + (ln & 31), k0 = 16 * ks + 8 * (ln >> 5);
        unsigned short pc[3][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wide_split3(Mg[i * C + k0 + j], pc[0][j], pc[1][j], pc[2][j]);
#pragma unroll
        for (int q = 0; q < 3; ++q)
            dst[q * FR + e] = make_uint4(pc[q][0] | (pc[q][1] << 16), pc[q][2] | (pc[q][3] << 16), pc[q][4] | (pc[q][5] << 16),
                                         pc[q][6] | (pc[q][7] << 16));
    }
}
#define PDE_WIDE_MFMA6(ACC, A, B)                                                                   \
    do {                                                                                            \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], ACC, 0, 0, 0);                    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], ACC, 0, 0, 0);                    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], ACC, 0, 0, 0);                    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], ACC, 0, 0, 0);                    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], ACC, 0, 0, 0);                    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], ACC, 0, 0, 0);                    \
    } while (0)

// operator fragments: Af[ct][p][lane][q] = M[32 ct + (lane & 31)][8 p + 4 (lane >> 5) + q]   (TRANS: M^T)
template <int WAVES, bool TRANS>
__device__ __forceinline__ void wide_fill_operator(float* Af, const float* Mg, int tid) {
    constexpr int C = 4 * WAVES, KP = C / 8, CT = C / 32;
    for (int e = tid; e < CT * KP * 256; e += 64 * WAVES) {
        const int q = e & 3, ln = (e >> 2) & 63, p = (e >> 8) % KP, ct = (e >> 8) / KP;
        const int i = 32 * ct + (ln & 31), k = 8 * p + 4 * (ln >> 5) + q;
        Af[e] = TRANS ? Mg[k * C + i] : Mg[i * C + k];
    }
}

// v[j][.] <- sum_c Op[4w+j][c] v_c[.] over the C channels of the workgroup's sample (v_c: the planes held by the
// owners); MM = elements per lane (N/2); all waves of the workgroup call it together.
template <int WAVES, int MM>
__device__ __forceinline__ void wide_mix(float (&v)[4][MM], float* X, const float* Af, int w, int lane) {
    constexpr int C = 4 * WAVES, KP = C / 8;
    const int lh = lane >> 5, ln = lane & 31;
    sfor<0, 2>([&](auto HC) __attribute__((always_inline)) {
        constexpr int h = decltype(HC)::value;
        {
            float4* dst = reinterpret_cast<float4*>(X) + (size_t)(w * 8) * 64 + lane;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = 8 * h + kk;
                float4 x;
                x.x = (k < MM) ? v[0][k < MM ? k : 0] : 0.f;
                x.y = (k < MM) ? v[1][k < MM ? k : 0] : 0.f;
                x.z = (k < MM) ? v[2][k < MM ? k : 0] : 0.f;
                x.w = (k < MM) ? v[3][k < MM ? k : 0] : 0.f;
                dst[kk * 64] = x;
            }
        }
        __syncthreads();
        wide_f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // my two [32 x 32] tiles: C = 64: pixel tile w, output-channel tiles 0 and 1;  C = 32: pixel tiles 2w, 2w+1
        const int kk0 = (C == 64) ? (w >> 1) : w;
        const int l0 = (C == 64) ? 32 * (w & 1) : 0;
        const float4* Xq = reinterpret_cast<const float4*>(X);
        const float4* Aq = reinterpret_cast<const float4*>(Af);
#if PDE_WIDE_SPLIT
        {
            // contraction group ks = 16 channels = channel groups 4 ks .. 4 ks + 3; lanes 0-31 take the first two (k = 0..7),
            // lanes 32-63 the other two (k = 8..15): B[k = 8 lh + j][pixel ln] from two float4 of the exchange image
            constexpr int KS = C / 16, CT = C / 32, FR = CT * KS * 64;
            const uint4* A3 = reinterpret_cast<const uint4*>(Af);
            auto afrag = [&](int ct, int ks, wide_v8bf (&pc)[3]) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < 3; ++q) pc[q] = __builtin_bit_cast(wide_v8bf, A3[q * FR + (ct * KS + ks) * 64 + lane]);
            };
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                wide_v8bf b0[3];
                wide_split8(Xq[((4 * ks + 2 * lh) * 8 + kk0) * 64 + l0 + ln], Xq[((4 * ks + 2 * lh + 1) * 8 + kk0) * 64 + l0 + ln], b0);
                if constexpr (C == 64) {
                    wide_v8bf a0[3], a1[3];
                    afrag(0, ks, a0);
                    afrag(1, ks, a1);
                    PDE_WIDE_MFMA6(acc[0], a0, b0);
                    PDE_WIDE_MFMA6(acc[1], a1, b0);
                } else {
                    wide_v8bf b1[3], a0[3];
                    wide_split8(Xq[((4 * ks + 2 * lh) * 8 + kk0) * 64 + 32 + ln], Xq[((4 * ks + 2 * lh + 1) * 8 + kk0) * 64 + 32 + ln], b1);
                    afrag(0, ks, a0);
                    PDE_WIDE_MFMA6(acc[0], a0, b0);
                    PDE_WIDE_MFMA6(acc[1], a0, b1);
                }
            }
        }
#else
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            const float4 b0 = Xq[((2 * p + lh) * 8 + kk0) * 64 + l0 + ln];
            if constexpr (C == 64) {
                const float4 a0 = Aq[p * 64 + lane], a1 = Aq[(KP + p) * 64 + lane];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, acc[1], 0, 0, 0);
            } else {
                const float4 b1 = Xq[((2 * p + lh) * 8 + kk0) * 64 + 32 + ln];
                const float4 a0 = Aq[p * 64 + lane];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b1.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b1.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b1.w, acc[1], 0, 0, 0);
            }
        }
#endif
        // products back in place: D row = (r & 3) + 8 (r >> 2) + 4 lh -> channel group 8 ct + 2 (r >> 2) + lh, member r & 3
        {
            float4* Xw = reinterpret_cast<float4*>(X);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ct = (C == 64) ? t : 0;
                const int lt = (C == 64) ? l0 : 32 * t;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    Xw[((8 * ct + 2 * g + lh) * 8 + kk0) * 64 + lt + ln] =
                        make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
            }
        }
        __syncthreads();
        {
            const float4* src = reinterpret_cast<const float4*>(X) + (size_t)(w * 8) * 64 + lane;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = 8 * h + kk;
                if (k < MM) {
                    const float4 x = src[kk * 64];
                    v[0][k < MM ? k : 0] = x.x; v[1][k < MM ? k : 0] = x.y; v[2][k < MM ? k : 0] = x.z; v[3][k < MM ? k : 0] = x.w;
                }
            }
        }
    });
}

// my 16 values of a lane-major image: 4 loads of 16 bytes, each contiguous across the wave
template <int MM>
__device__ __forceinline__ void wide_load(const float* img, int lane, float (&dst)[MM]) {
    const float4* src = reinterpret_cast<const float4*>(img) + lane;
#pragma unroll
    for (int i = 0; i < (MM + 3) / 4; ++i) {
        const float4 x = src[i * 64];
        dst[4 * i] = x.x;
        if (4 * i + 1 < MM) dst[4 * i + 1] = x.y;
        if (4 * i + 2 < MM) dst[4 * i + 2] = x.z;
        if (4 * i + 3 < MM) dst[4 * i + 3] = x.w;
    }
}

template <int N, int WAVES, int SPLIT>
__global__ __launch_bounds__(64 * WAVES) void adi_wide_fwd_kernel(WideArgs a) {
    constexpr int MM = Geo<N>::M, C = 4 * WAVES;
    constexpr int SPS = SPLIT == kSplitStrang ? 3 : 2;
    static_assert(SPLIT == kSplitStrang || SPLIT == kSplitLie, "step pattern must be known");
    static_assert(WAVES == 8 || WAVES == 16, "C = 32 or 64");
    static_assert(kImage <= kWideGroup, "a wave's private image lies inside its slice of the exchange image");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = lane >> 5, l = lane & 31;
    float* X = smem;
    float* T = X + (size_t)w * kWideGroup;
    float* Af = smem + (size_t)WAVES * kWideGroup;
    if constexpr (PDE_WIDE_SPLIT) wide_fill_operator3<WAVES>(Af, a.M, tid);
    else wide_fill_operator<WAVES, false>(Af, a.M, tid);
    for (int e = lane; e < kWideGroup; e += 64) T[e] = 0.f;             // idle lanes (N < 32) never meet uninitialised LDS
    const float* u = static_cast<const float*>(a.u);
    float* st = static_cast<float*>(a.states);
    const size_t tens = (size_t)a.B * C * N * N;
    const int K = a.K, mode = a.mode;
    __syncthreads();

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        float v[4][MM];
        sfor<0, 4>([&](auto JC) __attribute__((always_inline)) {
            constexpr int j = decltype(JC)::value;
            small_load<N, 0, float>(u, b, C, 4 * w + j, lane, l, hf, T, v[j]);
        });
        for (int k = 0; k < K; ++k) {
            if (mode == 1) wide_mix<WAVES, MM>(v, X, Af, w, lane);       // cifar10.py:91
            sfor<0, SPS>([&](auto SI) __attribute__((always_inline)) {
                constexpr int si = decltype(SI)::value;
                constexpr int AX = (si == 1) ? PDE_AXIS_Y : PDE_AXIS_X;   // Strang x,y,x / Lie x,y
                const float* rec0 = a.coef + ((size_t)(k * SPS + si) * C + 4 * w) * kWideRec;
                sfor<0, 4>([&](auto JC) __attribute__((always_inline)) {
                    constexpr int j = decltype(JC)::value;
                    float e[MM], inv[MM];
                    const float* rec = rec0 + (size_t)j * kWideRec;
                    wide_load<MM>(rec + kW_E, lane, e);
                    wide_load<MM>(rec + kW_Inv, lane, inv);
                    const float jn = rec[kW_Jn + l];
                    if (AX == PDE_AXIS_Y) relayout<N, 0>(v[j], T, l, hf);
                    solve_fwd_rows<MM, 1>(v[j], e, inv, jn, hf);
                    if (AX == PDE_AXIS_Y) relayout<N, 0>(v[j], T, l, hf);
                });
            });
            if (a.keep || (mode == 1 && k == K - 1)) {                   // the step's sweep output, for the backward
                const int slot = mode == 1 ? 2 * k + 1 : 2 * k;
                float* dst = (slot == 2 * K - 1 && a.last != nullptr) ? static_cast<float*>(a.last) : st + (size_t)slot * tens;
                sfor<0, 4>([&](auto JC) __attribute__((always_inline)) {
                    constexpr int j = decltype(JC)::value;
                    small_store<N, 0, float>(dst, b, C, 4 * w + j, lane, l, hf, T, v[j]);
                });
            }
            if (mode == 2) wide_mix<WAVES, MM>(v, X, Af, w, lane);       // SVHN.py:71
        }
        if (mode == 2) {
            float* dst = a.last != nullptr ? static_cast<float*>(a.last) : st + (size_t)(2 * K - 1) * tens;
            sfor<0, 4>([&](auto JC) __attribute__((always_inline)) {
                constexpr int j = decltype(JC)::value;
                small_store<N, 0, float>(dst, b, C, 4 * w + j, lane, l, hf, T, v[j]);
            });
        }
    }
}

template <int N>
int wide_fwd_launch(int C, int split, const WideArgs& wa, int grid, hipStream_t st) {
    static unsigned long long cfg[4] = {0, 0, 0, 0};
#define PDE_WIDE_GO(WV, SP, slot)                                                                                       \
    {                                                                                                                   \
        if (ensure_dynamic_lds((const void*)adi_wide_fwd_kernel<N, WV, SP>, (int)wide_lds_bytes<WV>(), cfg[slot]) != PDE_OK) \
            return PDE_E_LAUNCH;                                                                                        \
        hipLaunchKernelGGL((adi_wide_fwd_kernel<N, WV, SP>), dim3(grid), dim3(64 * WV), wide_lds_bytes<WV>(), st, wa);   \
        return check_launch();                                                                                          \
    }
    if (C == 64 && split == kSplitStrang) PDE_WIDE_GO(16, kSplitStrang, 0)
    if (C == 64 && split == kSplitLie) PDE_WIDE_GO(16, kSplitLie, 1)
    if (C == 32 && split == kSplitStrang) PDE_WIDE_GO(8, kSplitStrang, 2)
    if (C == 32 && split == kSplitLie) PDE_WIDE_GO(8, kSplitLie, 3)
#undef PDE_WIDE_GO
    return PDE_E_BADARG;
}

}  // namespace
}  // namespace pde
