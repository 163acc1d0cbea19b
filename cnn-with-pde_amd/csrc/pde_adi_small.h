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
constexpr int kSmallMaxL = 4;                                     // layers that share an input in one launch
constexpr int kSmallRecB = ((kRecBwd / 4 + 63) / 64) * 256;      // backward record slot (whole 1-KB DMA pieces)
constexpr int kGmStride = kSmallMaxC + 2;                         // row c of the matrix gradient | skip term | <g, y_i>

// One layer of the launch.  Several layers that read the SAME input (cifar10.py:272-274: three, cifar_2version.py:
// 287-288: two) run one after the other inside the launch: out = sum_i w_i y_i (cifar10.py:277-280 without the
// attention gates), and in the backward gu = sum_i gu_i.
struct SmallLayer {
    const float* coef;      // [S][C][kRecStride]
    const SweepTab* tabs;   // [K] one table per step
    const int* varying;     // [C]
    const float* M;         // [C][C] channel_mixing / channel_coupling
    const float* skip_w;    // SVHN skip_weight (device scalar); null: no skip blend
    void* states;           // [K][B][C][N][N] of the tensor type: sweep output of every step (fwd: null = do not keep)
    float* ckpt;            // [K][nck][B][C][N][N] fp32: states inside a step (null: none)
    const void* gys;        // bwd: dL/dy_i on top of w_i dL/dout (null: none)
    const float* roff;      // bwd: [B][C] dL/d(plane sum of y_i), added to every element of the plane's gradient (null: none)
    float* psum;            // fwd: [B][C] sum of y_i over the plane (the average pool of cifar10.py:239; null: not wanted)
    float* part;            // bwd: [grid][C][4][kImage]
    float* gm_part;         // bwd: [grid][C][kGmStride]
    unsigned long long ck[2];   // bit i: the state after sweep i of EVERY step is checkpointed
    int nck;
    int K, mode;            // mode 1: operator before every step, 2: after
    int smooth3;
    float step_scale;       // (1+eps)^-(sweeps per step): undoes the factor the adjoint solves carry
    float w;                // weight of this layer's output in `out` ...
    const float* wp;        // ... or, when not null, a device scalar holding it
    void* slab;             // par: (B,C,N,N) of the tensor type: this layer's term of `out` (fwd) / `gu` (bwd)
};
struct SmallArgs {
    const void* u;          // the layers' common input (B,C,N,N)
    const void* gy;         // bwd: dL/dout (null: only the gys of the layers)
    void* out;              // fwd: sum_i w_i y_i (null: checkpoint pre-pass only); bwd: gu
    int B, C, L;
    int par;                // 1: the layers run side by side, workgroup x works on layer x % L of sample x / L, and writes
                            // its term of `out` / `gu` to the layer's slab; small_combine_kernel adds the slabs up
    SmallLayer layer[kSmallMaxL];
};
// The layer descriptors are read where they lie, in the kernel-argument segment, with scalar loads at a run-time
// index (an argument array indexed at run time would be held whole in SGPRs and spilled).
typedef const __attribute__((address_space(4))) SmallLayer* ConstLayer;
__device__ __forceinline__ ConstLayer small_layer(int i) {
    const __attribute__((address_space(4))) char* base =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    return (ConstLayer)(base + __builtin_offsetof(SmallArgs, layer)) + i;
}

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
    const int nC = a.C;
    float* ring = smem + (size_t)c * 2 * kRecFwdPad;                      // wave-private [2][kRecFwdPad]
    const unsigned ring_lds = lds_byte_address(smem) + (unsigned)c * 2u * kRecFwdPad * 4u;
    float* imgs = smem + (size_t)nC * 2 * kRecFwdPad;                     // [C][kImage]: plane I/O, re-layout, exchange
    float* T = imgs + (size_t)c * kImage;
    for (int e = lane; e < kImage; e += 64) T[e] = 0.f;                  // rows >= N stay zero: idle lanes read zeros
    const IO* u = static_cast<const IO*>(a.u);
    IO* y = static_cast<IO*>(a.out);
    const size_t tens = (size_t)a.B * nC * N * N;

    auto dma_rec = [&](int slot, const float* coef, int s) __attribute__((always_inline)) {
        const float* rec = coef + ((size_t)s * nC + c) * kRecStride + kG_Inv;
#pragma unroll
        for (int p = 0; p < kRecFwdPad / 256; ++p) {
            const int f = p * 64 + lane;
            if (f < kRecFwd / 4) lds_dma16_a(rec + p * 256, 16u * lane, ring_lds + ((unsigned)slot * kRecFwdPad + p * 256u) * 4u);
        }
    };

    // layers one after the other in every workgroup, or (par) side by side: one layer per workgroup
    const int lsel = a.par ? (int)(blockIdx.x % (unsigned)a.L) : -1;
    const int b0 = a.par ? (int)(blockIdx.x / (unsigned)a.L) : (int)blockIdx.x;
    const int bstep = a.par ? (int)(gridDim.x / (unsigned)a.L) : (int)gridDim.x;
    int cur = 0;
    dma_rec(0, small_layer(lsel < 0 ? 0 : lsel)->coef, 0);
    for (int li = 0; li < a.L; ++li) {
        if (lsel >= 0 && li != lsel) continue;
        const ConstLayer Lp = small_layer(li);
        const float* coef = Lp->coef;
        const float* Mp = Lp->M;
        const float* skp = Lp->skip_w;
        IO* st = static_cast<IO*>(Lp->states);
        float* ckpt = Lp->ckpt;
        float* psum = Lp->psum;
        const unsigned long long ck[2] = {Lp->ck[0], Lp->ck[1]};
        const int nck = Lp->nck, K = Lp->K, mode = Lp->mode, S = K * SPS;
        const float* wpp = Lp->wp;
        const float wl = wpp ? *wpp : Lp->w;
        const bool last_layer = li + 1 == a.L || lsel >= 0;
        const float* coef_next = last_layer ? small_layer(0)->coef : small_layer(li + 1)->coef;
        IO* yl = (lsel >= 0 && y != nullptr) ? static_cast<IO*>(Lp->slab) : y;      // where my term of the output goes
        float mrow[kSmallMaxC];                                           // my row of the operator
#pragma unroll
        for (int j = 0; j < kSmallMaxC; ++j) mrow[j] = (j < nC) ? Mp[c * nC + j] : 0.f;
        const float sk = skp ? sigmoid_f(*skp) : 0.f;
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
            __syncthreads();                                              // everyone has read: the images are free again
#pragma unroll
            for (int k = 0; k < M; ++k) v[k] = acc[k];
        };

        for (int b = b0; b < a.B; b += bstep) {
            const bool last_sample = b + bstep >= a.B;
            float v[M], u0[M];
            small_load<N, 0, IO>(u, b, nC, c, lane, l, hf, T, u0);
#pragma unroll
            for (int q = 0; q < M; ++q) v[q] = u0[q];
            for (int k = 0; k < K; ++k) {
                if (mode == 1) mix(v);                                    // cifar10.py:91
                sfor<0, SPS>([&](auto SI) __attribute__((always_inline)) {
                    constexpr int si = decltype(SI)::value;
                    constexpr int AX = (si == 1) ? PDE_AXIS_Y : PDE_AXIS_X;   // Strang x,y,x / Lie x,y
                    const int s = k * SPS + si;
                    dma_wait_all();                                       // my record s has landed (wave-private ring)
                    __builtin_amdgcn_wave_barrier();
                    // next record: of this sample, of my next sample, or of the next layer's first sample
                    if (s + 1 < S) dma_rec(cur ^ 1, coef, s + 1);
                    else if (!last_sample) dma_rec(cur ^ 1, coef, 0);
                    else if (!last_layer) dma_rec(cur ^ 1, coef_next, 0);
                    const float* rec = ring + (size_t)cur * kRecFwdPad;
                    if (AX == PDE_AXIS_Y) relayout<N, 0>(v, T, l, hf);
                    solve_fwd<M, 1>(v, rec, l, hf);
                    if (AX == PDE_AXIS_Y) relayout<N, 0>(v, T, l, hf);
                    if (ckpt != nullptr && si < SPS - 1 && ck_bit(ck, si)) {          // backward pre-pass: park this state
                        float* slot = ckpt + ((size_t)k * nck + ck_slot(ck, si)) * tens;
                        small_store<N, 0, float>(slot, b, nC, c, lane, l, hf, T, v);
                    }
                    cur ^= 1;
                });
                if (st != nullptr) {                                      // the step's sweep output, for the backward
                    small_store<N, 0, IO>(st + (size_t)k * tens, b, nC, c, lane, l, hf, T, v);
                    if (sizeof(IO) < 4) {                                 // go on from what the backward will read
#pragma unroll
                        for (int q = 0; q < M; ++q) v[q] = round_io<IO>(v[q]);
                    }
                }
                if (mode == 2) mix(v);                                    // SVHN.py:71
            }
            if (y != nullptr) {
                if (skp != nullptr) {                                     // SVHN.py:74  sigmoid(w) u0 + (1 - sigmoid(w)) u
#pragma unroll
                    for (int q = 0; q < M; ++q) v[q] = sk * u0[q] + (1.0f - sk) * v[q];
                }
                if (psum != nullptr) {                                    // spatial sum of y_i: SpatialAttention's pool for free
                    float t = 0.f;
#pragma unroll
                    for (int q = 0; q < M; ++q) t += v[q];
                    t = (l < N) ? t : 0.f;
                    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
                    if (lane == 0) psum[(size_t)b * nC + c] = t;
                }
                if (a.L > 1) {                                            // out = sum_i w_i y_i: my own earlier store, re-read
                    if (li > 0 && lsel < 0) {
                        float o[M];
                        small_load<N, 0, IO>(y, b, nC, c, lane, l, hf, T, o);
#pragma unroll
                        for (int q = 0; q < M; ++q) v[q] = fmaf(wl, v[q], o[q]);
                    } else {
#pragma unroll
                        for (int q = 0; q < M; ++q) v[q] = wl * v[q];
                    }
                }
                small_store<N, 0, IO>(yl, b, nC, c, lane, l, hf, T, v);
            }
        }
    }
    dma_wait_all();
}

// out = slab_0 + slab_1 + ... in this order (the layers' terms of the weighted sum / of the input gradient)
struct CombineArgs { const void* slab[kSmallMaxL]; void* out; int L; size_t n4; };
template <typename IO>
__global__ __launch_bounds__(256) void small_combine_kernel(CombineArgs a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n4) return;
    float4 s = IoTraits<IO>::load4(static_cast<const IO*>(a.slab[0]) + 4 * i);
    for (int k = 1; k < a.L; ++k) {
        const float4 t = IoTraits<IO>::load4(static_cast<const IO*>(a.slab[k]) + 4 * i);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    IoTraits<IO>::store4(static_cast<IO*>(a.out) + 4 * i, s);
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
    const int nC = a.C;
    float* ring = smem + (size_t)c * 2 * RECP;                            // wave-private [2][RECP]
    const unsigned ring_lds = lds_byte_address(smem) + (unsigned)c * 2u * RECP * 4u;
    float* imgR = smem + (size_t)nC * 2 * RECP;                           // [C][kImage]: adjoints (also my re-layout image)
    float* imgX = imgR + (size_t)nC * kImage;                             // [C][kImage]: states
    float* T = imgR + (size_t)c * kImage;
    float* TX = imgX + (size_t)c * kImage;
    for (int e = lane; e < kImage; e += 64) { T[e] = 0.f; TX[e] = 0.f; }
    const IO* u = static_cast<const IO*>(a.u);
    const IO* gy = static_cast<const IO*>(a.gy);
    IO* gu = static_cast<IO*>(a.out);
    const size_t tens = (size_t)a.B * nC * N * N;
    const bool live = l < N;                                              // idle lanes (N < 32) carry no data — and whatever they
                                                                          // hold (Inf, NaN) must not reach a sum: select, never multiply

    auto dma_rec = [&](int slot, const float* coef, int s) __attribute__((always_inline)) {
        const float* rec = coef + ((size_t)s * nC + c) * kRecStride + kBwdOff;
#pragma unroll
        for (int p = 0; p < RECP / 256; ++p) {
            const int f = p * 64 + lane;
            if (f < kRecBwd / 4) lds_dma16_a(rec + p * 256, 16u * lane, ring_lds + ((unsigned)slot * RECP + p * 256u) * 4u);
        }
    };

    const int lsel = a.par ? (int)(blockIdx.x % (unsigned)a.L) : -1;
    const int b0 = a.par ? (int)(blockIdx.x / (unsigned)a.L) : (int)blockIdx.x;
    const int bstep = a.par ? (int)(gridDim.x / (unsigned)a.L) : (int)gridDim.x;
    int cur = 0;
    dma_rec(0, small_layer(lsel < 0 ? 0 : lsel)->coef, small_layer(lsel < 0 ? 0 : lsel)->K * SPS - 1);
    for (int li = 0; li < a.L; ++li) {
        if (lsel >= 0 && li != lsel) continue;
        const ConstLayer Lp = small_layer(li);
        const float* coef = Lp->coef;
        const SweepTab* tabs = Lp->tabs;
        const float* Mp = Lp->M;
        const float* skp = Lp->skip_w;
        const IO* st = static_cast<const IO*>(Lp->states);
        const IO* gys = static_cast<const IO*>(Lp->gys);
        const float* roff = Lp->roff;
        const float* ckpt = Lp->ckpt;
        const unsigned long long ck[2] = {Lp->ck[0], Lp->ck[1]};
        const int nck = Lp->nck, K = Lp->K, mode = Lp->mode, S = K * SPS, smooth = Lp->smooth3;
        const float* wpp = Lp->wp;
        const float wl = wpp ? *wpp : Lp->w, step_scale = Lp->step_scale;
        const bool last_layer = li + 1 == a.L || lsel >= 0;
        const ConstLayer Ln = small_layer(last_layer ? 0 : li + 1);
        IO* gul = lsel >= 0 ? static_cast<IO*>(Lp->slab) : gu;             // where my term of the input gradient goes
        const float* coef_next = Ln->coef;
        const int s_next = Ln->K * SPS - 1;
        const bool masked = as_const(Lp->varying)[c] != 0;                // wave-uniform: my channel's clamp mask moves in time
        float mrow[kSmallMaxC], mcol[kSmallMaxC];
#pragma unroll
        for (int j = 0; j < kSmallMaxC; ++j) {
            mrow[j] = (j < nC) ? Mp[c * nC + j] : 0.f;
            mcol[j] = (j < nC) ? Mp[j * nC + c] : 0.f;
        }
        const float sk = skp ? sigmoid_f(*skp) : 0.f;

        float Ax[M], Tx[M], Ay[M], Ty[M], gm[kSmallMaxC], gskip = 0.f, wsum = 0.f;
#pragma unroll
        for (int k = 0; k < M; ++k) Ax[k] = Tx[k] = Ay[k] = Ty[k] = 0.f;
#pragma unroll
        for (int j = 0; j < kSmallMaxC; ++j) gm[j] = 0.f;

        // adjoint of v = M w at a step boundary: r holds dL/dv (my channel), xw the operator's input w (my channel):
        //   gM[c][j] += sum r * w_j,   r <- sum_i M[i][c] r_i
        auto mix_adjoint = [&](float (&r)[M], const float (&xw)[M], bool skip_term, const float (&gsk)[M], int b)
                               __attribute__((always_inline)) {
            image_put<M>(T, l, hf, r);
            image_put<M>(TX, l, hf, xw);
            __syncthreads();
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
                        vfull[k] = fmaf(mrow[j], w[k], vfull[k]);         // the operator's output, recomputed (skip term)
                    }
                    gm[j] += live ? d : 0.f;
                }
            }
            __syncthreads();
            if (skip_term) {                                              // d/dw of sigmoid(w) u0 + (1 - sigmoid(w)) b_K
                float u0[M];
                small_load<N, 0, IO>(u, b, nC, c, lane, l, hf, T, u0);
#pragma unroll
                for (int k = 0; k < M; ++k) gskip += live ? gsk[k] * (u0[k] - vfull[k]) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < M; ++k) r[k] = racc[k];
        };

        for (int b = b0; b < a.B; b += bstep) {
            const bool last_sample = b + bstep >= a.B;
            float r[M], x[M], gsk[M], g0[M];
            if (gy != nullptr) small_load<N, 0, IO>(gy, b, nC, c, lane, l, hf, T, g0);
            else {
#pragma unroll
                for (int k = 0; k < M; ++k) g0[k] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < M; ++k) r[k] = wl * g0[k];                // out = sum_i w_i y_i
            if (gys != nullptr) {                                         // ... plus what arrives at y_i itself
                float gi[M];
                small_load<N, 0, IO>(gys, b, nC, c, lane, l, hf, T, gi);
#pragma unroll
                for (int k = 0; k < M; ++k) r[k] += gi[k];
            }
            if (roff != nullptr) {                                        // ... and at its spatial sum (the average pool)
                const float ro = live ? roff[(size_t)b * nC + c] : 0.f;
#pragma unroll
                for (int k = 0; k < M; ++k) r[k] += ro;
            }
#pragma unroll
            for (int k = 0; k < M; ++k) { gsk[k] = sk * r[k]; r[k] = (1.0f - sk) * r[k]; }     // sk = 0 without a skip blend
            if (mode == 1) {
                small_load<N, 0, IO>(st + (size_t)(K - 1) * tens, b, nC, c, lane, l, hf, TX, x);   // = y_i
#pragma unroll
                for (int k = 0; k < M; ++k) wsum += live ? g0[k] * x[k] : 0.f;                   // d out / d w_i = y_i
            }
            for (int k = K - 1; k >= 0; --k) {
                if (mode == 2) {                                          // SVHN: the coupling came after the sweeps
                    small_load<N, 0, IO>(st + (size_t)k * tens, b, nC, c, lane, l, hf, TX, x);
                    mix_adjoint(r, x, skp != nullptr && k == K - 1, gsk, b);
                }
                const ConstTab tab = as_const(tabs + k);
                sfor<0, SPS>([&](auto SI) __attribute__((always_inline)) {
                    constexpr int si = SPS - 1 - decltype(SI)::value;     // newest sweep of the step first
                    constexpr int AX = (si == 1) ? PDE_AXIS_Y : PDE_AXIS_X;
                    const int s = k * SPS + si;
                    dma_wait_all();
                    __builtin_amdgcn_wave_barrier();
                    // next record: of this sample, of my next sample, or of the next layer's first sample
                    if (s > 0) dma_rec(cur ^ 1, coef, s - 1);
                    else if (!last_sample) dma_rec(cur ^ 1, coef, S - 1);
                    else if (!last_layer) dma_rec(cur ^ 1, coef_next, s_next);
                    const float* rec = ring + (size_t)cur * RECP;
                    const float* recg = coef + ((size_t)s * nC + c) * kRecStride + kBwdOff;     // mask image: read from memory
                    const float tau = (si == 0) ? tab->dts[0] : (si == 1 ? tab->dts[1] : tab->t_last[0]);
                    if (AX == PDE_AXIS_Y) {
                        if (masked) small_adj_sweep<N, PDE_AXIS_Y, true>(r, x, Ay, Ty, rec, recg, tau, T, l, hf, smooth);
                        else small_adj_sweep<N, PDE_AXIS_Y, false>(r, x, Ay, Ty, rec, recg, tau, T, l, hf, smooth);
                    } else {
                        if (masked) small_adj_sweep<N, PDE_AXIS_X, true>(r, x, Ax, Tx, rec, recg, tau, T, l, hf, smooth);
                        else small_adj_sweep<N, PDE_AXIS_X, false>(r, x, Ax, Tx, rec, recg, tau, T, l, hf, smooth);
                    }
                    // x is the rebuilt state after sweep si-1 of this step; take the checkpoint instead if there is one
                    if (si > 0 && ckpt != nullptr && ck_bit(ck, si - 1)) {
                        const float* slot = ckpt + ((size_t)k * nck + ck_slot(ck, si - 1)) * tens;
                        small_load<N, 0, float>(slot, b, nC, c, lane, l, hf, TX, x);
                        const float sc = tab->ysc[si - 1];
#pragma unroll
                        for (int q = 0; q < M; ++q) x[q] *= sc;
                    }
                    cur ^= 1;
                });
#pragma unroll
                for (int q = 0; q < M; ++q) r[q] *= step_scale;          // the (1+eps) carried by every adjoint solve
                if (mode == 1) {                                          // cifar10: the mixing came before the sweeps
                    if (k > 0) small_load<N, 0, IO>(st + (size_t)(k - 1) * tens, b, nC, c, lane, l, hf, TX, x);
                    else small_load<N, 0, IO>(u, b, nC, c, lane, l, hf, TX, x);
                    mix_adjoint(r, x, false, gsk, b);                     // x stays: it is the sweep output of step k-1
                }
            }
#pragma unroll
            for (int q = 0; q < M; ++q) r[q] += gsk[q];                  // the skip branch's share of dL/du
            if (li > 0 && lsel < 0) {                                     // gu = sum over the layers: my own earlier store
                float o[M];
                small_load<N, 0, IO>(gu, b, nC, c, lane, l, hf, T, o);
#pragma unroll
                for (int q = 0; q < M; ++q) r[q] += o[q];
            }
            small_store<N, 0, IO>(gul, b, nC, c, lane, l, hf, T, r);
        }

        // my channel's sums of this layer: this workgroup's slot of the partial buffers (one wave per channel: nothing
        // to add up here)
        float* dst = Lp->part + ((size_t)b0 * nC + c) * 4 * kImage + l * kLineStride + hf * kHalfPad;
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
        float* gd = Lp->gm_part + ((size_t)b0 * nC + c) * kGmStride;
#pragma unroll
        for (int j = 0; j < kGmStride; ++j) {                 // (the lanes' sums meet in double: these scalars cancel heavily)
            double v = (j < kSmallMaxC) ? gm[j] : (j == kSmallMaxC ? gskip : wsum);
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) gd[j] = (float)v;
        }
    }
    dma_wait_all();
}

template <typename IO>
int small_combine_io(const CombineArgs& ca, hipStream_t st) {
    hipLaunchKernelGGL(small_combine_kernel<IO>, dim3((unsigned)((ca.n4 + 255) / 256)), dim3(256), 0, st, ca);
    return check_launch();
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
