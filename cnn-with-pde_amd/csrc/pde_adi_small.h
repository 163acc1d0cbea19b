// Whole-layer kernels for the implicit layers with a channel operator between the time steps, C <= 4 channels:
// cifar10.EnhancedDiffusionLayer / cifar_2version.LearnableDiffusionLayer (u <- M u BEFORE every step,
// cifar10.py:84-112) and SVHN.DiffusionLayer (u <- K u AFTER every step, then the sigmoid skip blend with the
// input, SVHN.py:55-76) at the channel counts the reference itself uses (C = 3).  ONE launch runs the whole time
// loop forward, one the whole adjoint; the per-step path (pde_adi_mixed_*) needs ~110 launches for the three
// cifar10 layers and is bound by the host's launch rate there.
//
// Layout: a workgroup is C waves, wave c owns channel c of the workgroup's sample (J planes per lane = J samples);
// inside a step every wave runs the ordinary sweep code on its channel (solve_fwd / solve_adj / state_x / state_y
// of pde_adi_dev.h, coefficient records in a wave-PRIVATE two-slot LDS ring filled by LDS-DMA: no workgroup
// barrier for coefficients).  At a step boundary the waves exchange their planes through LDS images (same lane
// positions, ds_write_b128 / ds_read_b128) and every lane applies its row of the C x C operator in registers:
// C^2 FMAs per element and step.  Forward parks the sweep output of every step (what autograd keeps in the
// reference); the backward reads them, rebuilds the states inside a step backwards as the big kernel does, and
// accumulates the matrix gradient g u^T (its row c in C scalars per lane) and the skip-weight gradient next to the
// four coefficient sums.  Sums over the batch are written per workgroup and added in a fixed order by
// adi_pgrad_kernel (no float atomics).
#pragma once
#include "pde_adi_dev.h"

namespace pde {
namespace {

constexpr int kSmallMaxC = 4;
constexpr int kSmallRecB = ((kRecBwd / 4 + 63) / 64) * 256;      // backward record slot (whole 1-KB DMA pieces)

struct SmallArgs {
    const void* u;          // layer input (B,C,N,N)
    const void* gy;         // bwd: dL/dy
    void* out;              // fwd: y (null: checkpoint pre-pass only); bwd: gu
    void* states;           // [K][B][C][N][N] of the tensor type: sweep output of every step (fwd: null = do not keep)
    float* ckpt;            // [K][nck][B][C][N][N] fp32: states inside a step (null: none)
    const float* coef;      // [S][C][kRecStride]
    const SweepTab* tabs;   // [K] one table per step
    const int* varying;     // [C]
    const float* M;         // [C][C] channel_mixing / channel_coupling
    const float* skip_w;    // SVHN skip_weight (device scalar); null: no skip blend
    float* part;            // bwd: [grid][C][4][kImage]
    float* gm_part;         // bwd: [grid][C][kSmallMaxC + 1]: row c of the matrix gradient, then the skip term
    unsigned long long ck[2];   // bit i: the state after sweep i of EVERY step is checkpointed
    int nck;
    int B, C, K, mode;      // mode 1: operator before every step, 2: after
    int smooth3;
    float step_scale;       // (1+eps)^-(sweeps per step): undoes the factor the adjoint solves carry
};

__device__ __forceinline__ float sigmoid_f(float w) { return 1.0f / (1.0f + expf(-w)); }

template <int N, int C, typename IO, class P>
__device__ __forceinline__ void small_load(const IO* base, int b, int nch, int c, int lane, int l, int hf, float* T, P (&v)[N / 2]) {
    float4 raw[Geo<N>::kLoads];
    plane_fetch<N, IO>(base + ((size_t)b * nch + c) * (size_t)(N * N), true, lane, raw);
    plane_to_rows<N, C>(raw, T, lane, l, hf, v);
}
template <int N, int C, typename IO, class P>
__device__ __forceinline__ void small_store(IO* base, int b, int nch, int c, int lane, int l, int hf, float* T, const P (&v)[N / 2]) {
    rows_to_plane<N, C, IO>(v, T, lane, l, hf, base + ((size_t)b * nch + c) * (size_t)(N * N), true);
}

// my half row <-> an exchange image (rows in natural order, columns in half order: the layout of plane I/O)
template <int M>
__device__ __forceinline__ void image_put(float* img, int l, int hf, const float (&v)[M]) {
    float* dst = img + l * kLineStride + hf * kHalfPad;
#pragma unroll
    for (int i = 0; i < (M + 3) / 4; ++i) {
        float4 x;
        x.x = v[4 * i];
        x.y = (4 * i + 1 < M) ? v[4 * i + 1] : 0.f;
        x.z = (4 * i + 2 < M) ? v[4 * i + 2] : 0.f;
        x.w = (4 * i + 3 < M) ? v[4 * i + 3] : 0.f;
        *reinterpret_cast<float4*>(dst + 4 * i) = x;
    }
}
template <int M>
__device__ __forceinline__ void image_get(const float* img, int l, int hf, float (&v)[M]) {
    load_half<M>(img + l * kLineStride + hf * kHalfPad, v);
}

template <typename IO> __device__ __forceinline__ float round_io(float v) { return v; }
template <> __device__ __forceinline__ float round_io<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// ---- forward ---------------------------------------------------------------------------------------------
template <int N, typename IO, int SPLIT>
__global__ __launch_bounds__(64 * kSmallMaxC) void adi_small_fwd_kernel(SmallArgs a) {
    constexpr int M = Geo<N>::M;
    constexpr int SPS = SPLIT == kSplitStrang ? 3 : 2;
    static_assert(SPLIT == kSplitStrang || SPLIT == kSplitLie, "step pattern must be known");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, c = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = lane >> 5, l = lane & 31;
    const int nC = a.C, S = a.K * SPS;
    float* ring = smem + (size_t)c * 2 * kRecFwdPad;                      // wave-private [2][kRecFwdPad]
    float* imgs = smem + (size_t)nC * 2 * kRecFwdPad;                     // [C][kImage]: plane I/O, re-layout, exchange
    float* T = imgs + (size_t)c * kImage;
    for (int e = lane; e < kImage; e += 64) T[e] = 0.f;                  // rows >= N stay zero: idle lanes read zeros
    const IO* u = static_cast<const IO*>(a.u);
    IO* y = static_cast<IO*>(a.out);
    IO* st = static_cast<IO*>(a.states);
    const size_t tens = (size_t)a.B * nC * N * N;
    float mrow[kSmallMaxC];                                               // my row of the operator
#pragma unroll
    for (int j = 0; j < kSmallMaxC; ++j) mrow[j] = (j < nC) ? a.M[c * nC + j] : 0.f;
    const float sk = a.skip_w ? sigmoid_f(*a.skip_w) : 0.f;

    auto dma_rec = [&](int slot, int s) __attribute__((always_inline)) {
        const float* rec = a.coef + ((size_t)s * nC + c) * kRecStride + kG_Inv;
#pragma unroll
        for (int p = 0; p < kRecFwdPad / 256; ++p) {
            const int f = p * 64 + lane;
            if (f < kRecFwd / 4) lds_dma16_s(rec + p * 256, 16u * lane, ring + (size_t)slot * kRecFwdPad + p * 256);
        }
    };
    // v <- sum_j M[c][j] v_j over the channels of my sample: planes through the waves' images
    auto mix = [&](float (&v)[M]) __attribute__((always_inline)) {
        image_put<M>(T, l, hf, v);
        __syncthreads();
        float acc[M];
#pragma unroll
        for (int k = 0; k < M; ++k) acc[k] = 0.f;
#pragma unroll
        for (int j = 0; j < kSmallMaxC; ++j) {
            if (j < nC) {
                float o[M];
                image_get<M>(imgs + (size_t)j * kImage, l, hf, o);
#pragma unroll
                for (int k = 0; k < M; ++k) acc[k] = fmaf(mrow[j], o[k], acc[k]);
            }
        }
        __syncthreads();                                                  // everyone has read: the images are free again
#pragma unroll
        for (int k = 0; k < M; ++k) v[k] = acc[k];
    };

    int cur = 0;
    dma_rec(0, 0);
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        float v[M];
        small_load<N, 0, IO>(u, b, nC, c, lane, l, hf, T, v);
        for (int k = 0; k < a.K; ++k) {
            if (a.mode == 1) mix(v);                                      // cifar10.py:91
            sfor<0, SPS>([&](auto SI) __attribute__((always_inline)) {
                constexpr int si = decltype(SI)::value;
                constexpr int AX = (si == 1) ? PDE_AXIS_Y : PDE_AXIS_X;   // Strang x,y,x / Lie x,y
                const int s = k * SPS + si;
                dma_wait_all();                                           // my record s has landed (wave-private ring)
                __builtin_amdgcn_wave_barrier();
                int sn = s + 1;
                if (sn == S) sn = 0;                                      // first record of my next sample
                if (sn != 0 || b + (int)gridDim.x < a.B) dma_rec(cur ^ 1, sn);
                const float* rec = ring + (size_t)cur * kRecFwdPad;
                if (AX == PDE_AXIS_Y) relayout<N, 0>(v, T, l, hf);
                solve_fwd<M, 1>(v, rec, l, hf);
                if (AX == PDE_AXIS_Y) relayout<N, 0>(v, T, l, hf);
                if (a.ckpt != nullptr && si < SPS - 1 && ck_bit(a.ck, si)) {        // backward pre-pass: park this state
                    float* slot = a.ckpt + ((size_t)k * a.nck + ck_slot(a.ck, si)) * tens;
                    small_store<N, 0, float>(slot, b, nC, c, lane, l, hf, T, v);
                }
                cur ^= 1;
            });
            if (st != nullptr) {                                          // the step's sweep output, for the backward
                small_store<N, 0, IO>(st + (size_t)k * tens, b, nC, c, lane, l, hf, T, v);
                if (sizeof(IO) < 4) {                                     // go on from what the backward will read
#pragma unroll
                    for (int q = 0; q < M; ++q) v[q] = round_io<IO>(v[q]);
                }
            }
            if (a.mode == 2) mix(v);                                      // SVHN.py:71
        }
        if (y != nullptr) {
            if (a.skip_w != nullptr) {                                    // SVHN.py:74  sigmoid(w) u0 + (1 - sigmoid(w)) u
                float u0[M];
                small_load<N, 0, IO>(u, b, nC, c, lane, l, hf, T, u0);
#pragma unroll
                for (int q = 0; q < M; ++q) v[q] = sk * u0[q] + (1.0f - sk) * v[q];
            }
            small_store<N, 0, IO>(y, b, nC, c, lane, l, hf, T, v);
        }
    }
    dma_wait_all();
}

// ---- backward --------------------------------------------------------------------------------------------
// One adjoint sweep on my plane: r <- (A + eps I)^-T r, the coefficient-gradient contribution of this sweep added
// to (A, Tm) with weight 1 and tau, the state x rebuilt to the sweep's input.  Same arithmetic as the body of
// adi_bwd_body; the time-weighted sum is accumulated directly (G is per sweep, not per plane: 2 x 16 FMAs).
template <int N, int AX, bool MASKED>
__device__ __forceinline__ void small_adj_sweep(float (&r)[N / 2], float (&x)[N / 2], float (&A)[N / 2], float (&Tm)[N / 2],
                                                const float* rec, const float* rec_mask, float tau, float* T, int l, int hf,
                                                int smooth) {
    constexpr int M = N / 2;
    float ce[M], cinv[M], ckap[M], G[M];
#pragma unroll
    for (int k = 0; k < M; ++k) G[k] = 0.f;
    const float* crow = rec + l * kLineStride + hf * kHalfPad;
    const float cjn = rec[kB_Jn + l];
    if (AX == PDE_AXIS_Y) {
        relayout<N, 0>(r, T, l, hf);
        load_half<M>(crow + kB_E, ce);
        load_half<M>(crow + kB_Inv, cinv);
        solve_adj<M, 1>(r, ce, cinv, cjn, hf);
        relayout<N, 0>(r, T, l, hf);
        load_half<M>(crow + kB_KapX, ckap);
        state_y<N, 1, MASKED>(r, x, G, ckap, rec_mask, l, hf, smooth);
    } else {
        const float xin = xchg_half(x[M - 1], hf);
        load_half<M>(crow + kB_E, ce);
        load_half<M>(crow + kB_Inv, cinv);
        solve_adj<M, 1>(r, ce, cinv, cjn, hf);
        load_half<M>(crow + kB_KapX, ckap);
        state_x<M, 1, MASKED>(r, x, G, xin, ckap, rec_mask, l, hf, smooth);
    }
#pragma unroll
    for (int k = 0; k < M; ++k) { A[k] += G[k]; Tm[k] = fmaf(tau, G[k], Tm[k]); }
}

template <int N, typename IO, int SPLIT>
__global__ __launch_bounds__(64 * kSmallMaxC) void adi_small_bwd_kernel(SmallArgs a) {
    constexpr int M = Geo<N>::M;
    constexpr int SPS = SPLIT == kSplitStrang ? 3 : 2;
    constexpr int RECP = kSmallRecB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, c = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = lane >> 5, l = lane & 31;
    const int nC = a.C, S = a.K * SPS;
    float* ring = smem + (size_t)c * 2 * RECP;                            // wave-private [2][RECP]
    float* imgR = smem + (size_t)nC * 2 * RECP;                           // [C][kImage]: adjoints (also my re-layout image)
    float* imgX = imgR + (size_t)nC * kImage;                             // [C][kImage]: states
    float* T = imgR + (size_t)c * kImage;
    float* TX = imgX + (size_t)c * kImage;
    for (int e = lane; e < kImage; e += 64) { T[e] = 0.f; TX[e] = 0.f; }
    const IO* u = static_cast<const IO*>(a.u);
    const IO* gy = static_cast<const IO*>(a.gy);
    const IO* st = static_cast<const IO*>(a.states);
    IO* gu = static_cast<IO*>(a.out);
    const size_t tens = (size_t)a.B * nC * N * N;
    const bool masked = as_const(a.varying)[c] != 0;                      // wave-uniform: my channel's clamp mask moves in time
    float mrow[kSmallMaxC], mcol[kSmallMaxC];
#pragma unroll
    for (int j = 0; j < kSmallMaxC; ++j) {
        mrow[j] = (j < nC) ? a.M[c * nC + j] : 0.f;
        mcol[j] = (j < nC) ? a.M[j * nC + c] : 0.f;
    }
    const float sk = a.skip_w ? sigmoid_f(*a.skip_w) : 0.f;

    float Ax[M], Tx[M], Ay[M], Ty[M], gm[kSmallMaxC], gskip = 0.f;
#pragma unroll
    for (int k = 0; k < M; ++k) Ax[k] = Tx[k] = Ay[k] = Ty[k] = 0.f;
#pragma unroll
    for (int j = 0; j < kSmallMaxC; ++j) gm[j] = 0.f;

    auto dma_rec = [&](int slot, int s) __attribute__((always_inline)) {
        const float* rec = a.coef + ((size_t)s * nC + c) * kRecStride + kBwdOff;
#pragma unroll
        for (int p = 0; p < RECP / 256; ++p) {
            const int f = p * 64 + lane;
            if (f < kRecBwd / 4) lds_dma16_s(rec + p * 256, 16u * lane, ring + (size_t)slot * RECP + p * 256);
        }
    };
    // adjoint of v = M w at a step boundary: r holds dL/dv (my channel), xw the operator's input w (my channel):
    //   gM[c][j] += sum r * w_j,   r <- sum_i M[i][c] r_i
    auto mix_adjoint = [&](float (&r)[M], const float (&xw)[M], bool skip_term, const float (&gsk)[M], int b)
                           __attribute__((always_inline)) {
        image_put<M>(T, l, hf, r);
        image_put<M>(TX, l, hf, xw);
        __syncthreads();
        const float live = (l < N) ? 1.0f : 0.0f;                        // idle lanes (N < 32) carry no data
        float racc[M], vfull[M];
#pragma unroll
        for (int k = 0; k < M; ++k) { racc[k] = 0.f; vfull[k] = 0.f; }
#pragma unroll
        for (int j = 0; j < kSmallMaxC; ++j) {
            if (j < nC) {
                float o[M], w[M];
                image_get<M>(imgR + (size_t)j * kImage, l, hf, o);
                image_get<M>(imgX + (size_t)j * kImage, l, hf, w);
                float d = 0.f;
#pragma unroll
                for (int k = 0; k < M; ++k) {
                    racc[k] = fmaf(mcol[j], o[k], racc[k]);
                    d = fmaf(r[k], w[k], d);
                    vfull[k] = fmaf(mrow[j], w[k], vfull[k]);             // the operator's output, recomputed (skip term)
                }
                gm[j] = fmaf(live, d, gm[j]);
            }
        }
        __syncthreads();
        if (skip_term) {                                                  // d/dw of sigmoid(w) u0 + (1 - sigmoid(w)) b_K
            float u0[M];
            small_load<N, 0, IO>(u, b, nC, c, lane, l, hf, T, u0);
#pragma unroll
            for (int k = 0; k < M; ++k) gskip = fmaf(live * gsk[k], u0[k] - vfull[k], gskip);
        }
#pragma unroll
        for (int k = 0; k < M; ++k) r[k] = racc[k];
    };

    int cur = 0;
    dma_rec(0, S - 1);
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        float r[M], x[M], gsk[M];
        small_load<N, 0, IO>(gy, b, nC, c, lane, l, hf, T, r);
#pragma unroll
        for (int k = 0; k < M; ++k) { gsk[k] = sk * r[k]; r[k] = (1.0f - sk) * r[k]; }     // sk = 0 without a skip blend
        if (a.mode == 1) small_load<N, 0, IO>(st + (size_t)(a.K - 1) * tens, b, nC, c, lane, l, hf, TX, x);
        for (int k = a.K - 1; k >= 0; --k) {
            if (a.mode == 2) {                                            // SVHN: the coupling came after the sweeps
                small_load<N, 0, IO>(st + (size_t)k * tens, b, nC, c, lane, l, hf, TX, x);
                mix_adjoint(r, x, a.skip_w != nullptr && k == a.K - 1, gsk, b);
            }
            const ConstTab tab = as_const(a.tabs + k);
            sfor<0, SPS>([&](auto SI) __attribute__((always_inline)) {
                constexpr int si = SPS - 1 - decltype(SI)::value;         // newest sweep of the step first
                constexpr int AX = (si == 1) ? PDE_AXIS_Y : PDE_AXIS_X;
                const int s = k * SPS + si;
                dma_wait_all();
                __builtin_amdgcn_wave_barrier();
                int sn = s - 1;
                if (sn < 0) sn = S - 1;                                   // newest record of my next sample
                if (sn != S - 1 || b + (int)gridDim.x < a.B) dma_rec(cur ^ 1, sn);
                const float* rec = ring + (size_t)cur * RECP;
                const float* recg = a.coef + ((size_t)s * nC + c) * kRecStride + kBwdOff;     // mask image: read from memory
                const float tau = (si == 0) ? tab->dts[0] : (si == 1 ? tab->dts[1] : tab->t_last[0]);
                if (AX == PDE_AXIS_Y) {
                    if (masked) small_adj_sweep<N, PDE_AXIS_Y, true>(r, x, Ay, Ty, rec, recg, tau, T, l, hf, a.smooth3);
                    else small_adj_sweep<N, PDE_AXIS_Y, false>(r, x, Ay, Ty, rec, recg, tau, T, l, hf, a.smooth3);
                } else {
                    if (masked) small_adj_sweep<N, PDE_AXIS_X, true>(r, x, Ax, Tx, rec, recg, tau, T, l, hf, a.smooth3);
                    else small_adj_sweep<N, PDE_AXIS_X, false>(r, x, Ax, Tx, rec, recg, tau, T, l, hf, a.smooth3);
                }
                // x is the rebuilt state after sweep si-1 of this step; take the checkpoint instead if there is one
                if (si > 0 && a.ckpt != nullptr && ck_bit(a.ck, si - 1)) {
                    const float* slot = a.ckpt + ((size_t)k * a.nck + ck_slot(a.ck, si - 1)) * tens;
                    small_load<N, 0, float>(slot, b, nC, c, lane, l, hf, TX, x);
                    const float sc = tab->ysc[si - 1];
#pragma unroll
                    for (int q = 0; q < M; ++q) x[q] *= sc;
                }
                cur ^= 1;
            });
#pragma unroll
            for (int q = 0; q < M; ++q) r[q] *= a.step_scale;            // the (1+eps) carried by every adjoint solve
            if (a.mode == 1) {                                            // cifar10: the mixing came before the sweeps
                if (k > 0) small_load<N, 0, IO>(st + (size_t)(k - 1) * tens, b, nC, c, lane, l, hf, TX, x);
                else small_load<N, 0, IO>(u, b, nC, c, lane, l, hf, TX, x);
                mix_adjoint(r, x, false, gsk, b);                         // x stays: it is the sweep output of step k-1
            }
        }
#pragma unroll
        for (int q = 0; q < M; ++q) r[q] += gsk[q];                      // the skip branch's share of dL/du
        small_store<N, 0, IO>(gu, b, nC, c, lane, l, hf, T, r);
    }
    dma_wait_all();

    // my channel's sums: this workgroup's slot of the partial buffers (one wave per channel: nothing to add up here)
    float* dst = a.part + ((size_t)blockIdx.x * nC + c) * 4 * kImage + l * kLineStride + hf * kHalfPad;
#pragma unroll
    for (int arr = 0; arr < 4; ++arr) {
#pragma unroll
        for (int i = 0; i < (M + 3) / 4; ++i) {
            float4 v4;
            auto at = [&](int k) { return k < M ? ((arr == 0) ? Ax[k] : (arr == 1) ? Tx[k] : (arr == 2) ? Ay[k] : Ty[k]) : 0.f; };
            v4.x = at(4 * i); v4.y = at(4 * i + 1); v4.z = at(4 * i + 2); v4.w = at(4 * i + 3);
            *reinterpret_cast<float4*>(dst + arr * kImage + 4 * i) = v4;
        }
    }
    float* gd = a.gm_part + ((size_t)blockIdx.x * nC + c) * (kSmallMaxC + 1);
#pragma unroll
    for (int j = 0; j <= kSmallMaxC; ++j) {
        float v = (j < kSmallMaxC) ? gm[j] : gskip;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) gd[j] = v;
    }
}

template <int N, typename IO>
int small_fwd_io(int split, const SmallArgs& sa, int grid, size_t lds, hipStream_t st) {
    static unsigned long long cfg[2] = {0, 0};
    if (split == kSplitStrang) {
        if (ensure_dynamic_lds((const void*)adi_small_fwd_kernel<N, IO, kSplitStrang>, (int)lds, cfg[0]) != PDE_OK) return PDE_E_LAUNCH;
        hipLaunchKernelGGL((adi_small_fwd_kernel<N, IO, kSplitStrang>), dim3(grid), dim3(64 * sa.C), lds, st, sa);
    } else {
        if (ensure_dynamic_lds((const void*)adi_small_fwd_kernel<N, IO, kSplitLie>, (int)lds, cfg[1]) != PDE_OK) return PDE_E_LAUNCH;
        hipLaunchKernelGGL((adi_small_fwd_kernel<N, IO, kSplitLie>), dim3(grid), dim3(64 * sa.C), lds, st, sa);
    }
    return check_launch();
}
template <int N, typename IO>
int small_bwd_io(int split, const SmallArgs& sa, int grid, size_t lds, hipStream_t st) {
    static unsigned long long cfg[2] = {0, 0};
    if (split == kSplitStrang) {
        if (ensure_dynamic_lds((const void*)adi_small_bwd_kernel<N, IO, kSplitStrang>, (int)lds, cfg[0]) != PDE_OK) return PDE_E_LAUNCH;
        hipLaunchKernelGGL((adi_small_bwd_kernel<N, IO, kSplitStrang>), dim3(grid), dim3(64 * sa.C), lds, st, sa);
    } else {
        if (ensure_dynamic_lds((const void*)adi_small_bwd_kernel<N, IO, kSplitLie>, (int)lds, cfg[1]) != PDE_OK) return PDE_E_LAUNCH;
        hipLaunchKernelGGL((adi_small_bwd_kernel<N, IO, kSplitLie>), dim3(grid), dim3(64 * sa.C), lds, st, sa);
    }
    return check_launch();
}

}  // namespace
}  // namespace pde
