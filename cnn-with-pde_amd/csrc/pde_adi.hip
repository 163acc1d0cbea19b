// K1 — fused implicit ADI diffusion time-stepper for gfx950 (MI355X).
//
// Replaces, for one (B,C,N,N) tensor and a whole run of sweeps, the reference's
//   forward time loop            mnist_test.py:44-65   cifar10.py:74-114
//   diffuse_x / diffuse_y        mnist_test.py:67-133  cifar10.py:124-177
//   smooth_coefficients          mnist_test.py:135-149
//   thomas_solver_batch          mnist_test.py:151-198 cifar10.py:179-211
// and the autograd backward of all of the above.
//
// Design:
//   * one wave = one plane: lane (hf,l) owns HALF of line l (hf = 0 low end, 1 high
//     end) and keeps its N/2 unknowns in VGPRs for the whole time loop; J planes per
//     lane give independent recurrences for ILP and amortise coefficient reads;
//   * every line is solved from both ends towards the middle (two-sided / "twisted"
//     Thomas elimination): the two half-lines run the same code on mirrored
//     coefficients and meet in one cross-lane exchange, so a 64-wide wave is exactly
//     32 lines x 2 halves;
//   * the factorisation depends on (sweep, channel, line) only, never on the sample:
//     a small kernel computes it once per call, the sweep kernels stage one record
//     per sweep into LDS (double buffered, one barrier per sweep) and all waves of a
//     workgroup — which share a channel — read it from there;
//   * x <-> y re-layout goes through a wave-private LDS image, 4 B written and read
//     per element, no workgroup barrier;
//   * backward: adjoint two-sided solves; the states needed by the coefficient
//     gradients are rebuilt backwards from the output, x_{s-1} = (A_s + eps I) x_s,
//     so nothing is stored per sweep; the state never leaves the row layout (the
//     y-direction second difference is taken across lanes with DPP), only the adjoint
//     is re-laid out.  Parameter gradients are accumulated over the batch in registers
//     and reduced deterministically (no float atomics).
#include "pde_common.h"

#include <mutex>
#include <type_traits>
#include <vector>

namespace pde {

Timing& timing() {
    static Timing t;
    return t;
}

namespace {

// ------------------------------------------------------------------------------------
// factorisation kernel: one thread per (sweep, channel, line)
// ------------------------------------------------------------------------------------
// Per-launch sweep table read by the sweep kernels with scalar loads (keeping it in the
// kernel arguments makes hipcc hold all of it in SGPRs and spill them).
struct SweepTab {
    int axis[PDE_MAX_SWEEPS];
    float dts[PDE_MAX_SWEEPS];      // bwd: t_s minus t of the previous (earlier) sweep of the same axis
    int first_s[2];                 // bwd: earliest sweep of each axis (-1: none)
    float t_last[2];                // bwd: time of the latest sweep of each axis
};

struct FactorArgs {
    SweepTab* tab;
    int* varying;           // [C] set to 1 when a clamp mask of the channel differs between sweeps (may be null)
    float t_first[2];       // time of the earliest sweep of each axis
    const float* ab;
    const float* bb;
    const float* as;
    const float* bs;
    float* coef;            // [S][C][kRecAll]
    float* kmax;            // optional [S] (atomic max of coeff), may be null
    int C, N, S;
    int smooth3, has_max;
    float cmax, eps;
    PdeSweep sweep[PDE_MAX_SWEEPS];
};

__global__ __launch_bounds__(64) void adi_factor_kernel(FactorArgs a) {
    // one wave = two records (sweep s, channels 2p and 2p+1); lane = (record, line).  Records are
    // built in LDS and written out as whole 16-byte rows: a line-per-thread store straight to
    // global memory is a 4-byte scatter and ran 10x slower.
    __shared__ __attribute__((aligned(16))) float rec_s[2][kRecStride];
    const int N = a.N, m = N / 2;
    const int pairs = (a.C + 1) / 2;
    const int s = blockIdx.x / pairs;
    const int which = threadIdx.x >> 5;
    const int c = 2 * (blockIdx.x % pairs) + which;
    const int line = threadIdx.x & 31;
    if (blockIdx.x == 0 && threadIdx.x < a.S) {   // publish the sweep table
        const int idx = threadIdx.x;
        float tprev = 0.f, tlast = 0.f;
        int first = -1;
        const int ax = a.sweep[idx].axis;
        for (int q = 0; q < a.S; ++q) {
            if (a.sweep[q].axis != ax) continue;
            if (first < 0) first = q;
            if (q < idx) tprev = a.sweep[q].t;
            tlast = a.sweep[q].t;
        }
        a.tab->axis[idx] = ax;
        a.tab->dts[idx] = a.sweep[idx].t - tprev;
        a.tab->first_s[ax] = first;
        a.tab->t_last[ax] = tlast;
    }
    float* rec = rec_s[which];
    const PdeSweep sw = a.sweep[s];
    const bool xax = sw.axis == PDE_AXIS_X;
    const bool live = (c < a.C) && (line < N);
    // idle lines get zero rows: lanes beyond the plane run the same instruction stream on zeros,
    // so nothing they compute can leak a NaN through a lane exchange
    rec[kRecJn + line] = 0.f;
    for (int i = 0; i < kLineStride; ++i) {
        rec[kRecInv + line * kLineStride + i] = 0.f;
        rec[kRecE + line * kLineStride + i] = 0.f;
        rec[kRecKapX + line * kLineStride + i] = 0.f;
        rec[kRecMaskX + line * kLineStride + i] = 0.f;
    }
    __syncthreads();
    if (live) {
        const float* base = xax ? a.ab : a.bb;
        const float* slope = xax ? a.as : a.bs;
        const size_t cbase = (size_t)c * N * N;
        const int st = xax ? 1 : N;                 // stride between consecutive unknowns of my line
        const int o0 = xax ? line * N : line;
        float kap[PDE_MAX_N];
        float pass[PDE_MAX_N];
        bool differs = false;
        // theta = clamp(base + slope*t, eps[, max])          mnist_test.py:33-42
        for (int i = 0; i < N; ++i) {
            const float bs = base[cbase + o0 + i * st], sl = slope[cbase + o0 + i * st];
            float th = bs + sl * sw.t;
            const bool ps = (th >= a.eps) && (!a.has_max || th <= a.cmax);     // clamp passes the gradient
            const float th0 = bs + sl * a.t_first[sw.axis];
            const bool ps0 = (th0 >= a.eps) && (!a.has_max || th0 <= a.cmax);
            differs |= (ps != ps0);
            pass[i] = ps ? 1.0f : 0.0f;
            th = fmaxf(th, a.eps);
            if (a.has_max) th = fminf(th, a.cmax);
            kap[i] = th;
        }
        if (differs && a.varying) atomicOr(&a.varying[c], 1);
        if (a.smooth3) {                             // mnist_test.py:135-149 (replicate ends)
            const float third = 1.0f / 3.0f;
            float prev = kap[0];
            for (int i = 0; i < N; ++i) {
                const float cur = kap[i];
                const float nxt = kap[i + 1 < N ? i + 1 : N - 1];
                kap[i] = (prev * third + cur * third) + nxt * third;
                prev = cur;
            }
        }
        for (int i = 0; i < N; ++i) kap[i] = (kap[i] * sw.delta) / sw.h2;     // coeff = theta*dt/dx**2  mnist_test.py:83
        float* inv_row = rec + kRecInv + line * kLineStride;
        float* e_row = rec + kRecE + line * kLineStride;
        float e_in[2];
        // two-sided elimination of (A + eps I): rows i = k (hf 0) and i = N-1-k (hf 1),
        // den_k = b_k - kap_k * (kap_{k-1}/den_{k-1}) + eps       mnist_test.py:169,177
        for (int hf = 0; hf < 2; ++hf) {
            float e = 0.f;                           // kap_{k-1}/den_{k-1} of the outer neighbour
            for (int k = 0; k < m; ++k) {
                const int i = hf ? N - 1 - k : k;
                const float kp = kap[i];
                const float b = (k == 0) ? 1.0f + kp : 1.0f + 2.0f * kp;    // Neumann ends, mnist_test.py:88-93
                const float den = (b - kp * e) + a.eps;
                const float inv = 1.0f / den;
                e = kp * inv;
                inv_row[hf * kHalfPad + k] = inv;
                e_row[hf * kHalfPad + k] = e;
            }
            e_in[hf] = e;
        }
        rec[kRecJn + line] = 1.0f / (1.0f - e_in[0] * e_in[1]);
        // coefficient and clamp mask in row layout (always): element (h,w) at row h, half_pos(w)
        float* kx = rec + kRecKapX;
        float* mx = rec + kRecMaskX;
        if (xax) {
            for (int i = 0; i < N; ++i) {
                kx[line * kLineStride + half_pos(i, N)] = kap[i];
                mx[line * kLineStride + half_pos(i, N)] = pass[i];
            }
        } else {
            const int p = half_pos(line, N);
            for (int i = 0; i < N; ++i) {
                kx[i * kLineStride + p] = kap[i];
                mx[i * kLineStride + p] = pass[i];
            }
        }
    }
    __syncthreads();
    for (int w = 0; w < 2; ++w) {
        const int cw = 2 * (blockIdx.x % pairs) + w;
        if (cw >= a.C) break;
        float4* dst = reinterpret_cast<float4*>(a.coef + ((size_t)s * a.C + cw) * kRecStride);
        const float4* src = reinterpret_cast<const float4*>(rec_s[w]);
        for (int f = threadIdx.x; f < kRecStride / 4; f += 64) dst[f] = src[f];
    }
}

// max over the tensor of the coefficient of every sweep (no factorisation): feeds the host-side
// choice of checkpoints.  One wave per (sweep, pair of channels): lane = (channel of the pair, line);
// the wave reduces its maximum and issues ONE atomic.
__global__ __launch_bounds__(64) void adi_kmax_kernel(FactorArgs a) {
    const int N = a.N;
    const int pairs = (a.C + 1) / 2;
    const int s = blockIdx.x / pairs;
    const int c = 2 * (blockIdx.x % pairs) + (threadIdx.x >> 5);
    const int line = threadIdx.x & 31;
    float km = 0.f;
    if (c < a.C && line < N) {
        const PdeSweep sw = a.sweep[s];
        const bool xax = sw.axis == PDE_AXIS_X;
        const float* base = xax ? a.ab : a.bb;
        const float* slope = xax ? a.as : a.bs;
        const size_t cbase = (size_t)c * N * N;
        const int st = xax ? 1 : N;
        const int o0 = xax ? line * N : line;
        float th[PDE_MAX_N];
        for (int i = 0; i < N; ++i) {
            float v = base[cbase + o0 + i * st] + slope[cbase + o0 + i * st] * sw.t;
            v = fmaxf(v, a.eps);
            if (a.has_max) v = fminf(v, a.cmax);
            th[i] = v;
        }
        const float third = 1.0f / 3.0f;
        for (int i = 0; i < N; ++i) {
            float v = th[i];
            if (a.smooth3) v = (th[i > 0 ? i - 1 : 0] * third + th[i] * third) + th[i + 1 < N ? i + 1 : N - 1] * third;
            km = fmaxf(km, (v * sw.delta) / sw.h2);
        }
    }
    for (int o = 32; o > 0; o >>= 1) km = fmaxf(km, __shfl_xor(km, o, 64));
    if (threadIdx.x == 0) atomicMax((unsigned int*)&a.kmax[s], __float_as_uint(km));   // coefficients are > 0
}

// ------------------------------------------------------------------------------------
// sweep kernels
// ------------------------------------------------------------------------------------
constexpr int kWaves = 8;                         // waves per workgroup (512 threads)
constexpr int kThreads = kWaves * 64;

struct SweepArgs {
    const void* in0;        // fwd: u        bwd: gy
    const void* in1;        // fwd: -        bwd: y
    void* out;              // fwd: y        bwd: gu
    const float* coef;      // [S][C][kRecAll]
    float* part;            // bwd: [G][C][4][kImage] partial parameter-gradient sums
    const SweepTab* tab;
    const int* varying;     // bwd: [C] per-channel "clamp mask changes with time" flag
    const void* in2;        // bwd with checkpoints: u
    float* ckpt;            // bwd with checkpoints: [slots][B][C][N][N] fp32
    unsigned long long ck[2];   // bit s: the state after sweep s is checkpointed
    int Sf;                 // bwd: sweeps 0..Sf-1 are recomputed forward first (0: no checkpoints)
    int smooth3;
    int B, C, S, G;
    float one_eps;          // 1 + eps
};

__device__ __forceinline__ float xchg_half(float v) {     // value held by lane ^ 32
    return __shfl_xor(v, 32, 64);
}

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int kDppWaveShl1 = 0x130;   // lane i <- lane i+1
constexpr int kDppWaveShr1 = 0x138;   // lane i <- lane i-1

// Stage COUNT floats of a coefficient record global -> registers -> LDS: up to three 16-byte
// pieces per thread, held in plain locals of the kernel (a struct here ends up in scratch).
struct Staged { float4 r0, r1, r2; };
template <int COUNT>
__device__ __forceinline__ void stage_load(const float* src, int tid, float4& r0, float4& r1, float4& r2) {
    constexpr int F4 = COUNT / 4;
    static_assert(F4 <= 3 * kThreads, "record too large for three pieces per thread");
    r0 = reinterpret_cast<const float4*>(src)[tid];
    if constexpr (F4 > kThreads) {
        if (F4 >= 2 * kThreads || tid + kThreads < F4) r1 = reinterpret_cast<const float4*>(src)[tid + kThreads];
    }
    if constexpr (F4 > 2 * kThreads) {
        if (tid + 2 * kThreads < F4) r2 = reinterpret_cast<const float4*>(src)[tid + 2 * kThreads];
    }
}
template <int COUNT>
__device__ __forceinline__ void stage_store(float* dst, int tid, const float4& r0, const float4& r1, const float4& r2) {
    constexpr int F4 = COUNT / 4;
    reinterpret_cast<float4*>(dst)[tid] = r0;
    if constexpr (F4 > kThreads) {
        if (F4 >= 2 * kThreads || tid + kThreads < F4) reinterpret_cast<float4*>(dst)[tid + kThreads] = r1;
    }
    if constexpr (F4 > 2 * kThreads) {
        if (tid + 2 * kThreads < F4) reinterpret_cast<float4*>(dst)[tid + 2 * kThreads] = r2;
    }
}

// ---- plane I/O through the wave's LDS image (natural [h][w] rows, stride 36) --------
template <typename IO> struct IoTraits;
template <> struct IoTraits<float> {
    static constexpr int kVec = 4;                      // elements per 16-byte access
    __device__ static __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};
struct bf16_t { unsigned short v; };
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {      // round to nearest even, NaN kept
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
template <> struct IoTraits<bf16_t> {
    __device__ static __forceinline__ float4 load4(const bf16_t* p) {
        const ushort4 q = *reinterpret_cast<const ushort4*>(p);
        return make_float4(bf16_to_f32(q.x), bf16_to_f32(q.y), bf16_to_f32(q.z), bf16_to_f32(q.w));
    }
    __device__ static __forceinline__ void store4(bf16_t* p, float4 v) {
        ushort4 q;
        q.x = f32_to_bf16(v.x); q.y = f32_to_bf16(v.y); q.z = f32_to_bf16(v.z); q.w = f32_to_bf16(v.w);
        *reinterpret_cast<ushort4*>(p) = q;
    }
};

template <int N>
struct Geo {
    static constexpr int M = N / 2;
    static constexpr int NN4 = N * N / 4;               // float4 per plane
    static constexpr int R4 = N / 4;                    // float4 per row
    static constexpr int kLoads = (NN4 + 63) / 64;      // per-lane 16-byte accesses per plane
};

// global -> registers (issue only).  `valid` must be wave-uniform.
template <int N, typename IO>
__device__ __forceinline__ void plane_fetch(const IO* gp, bool valid, int lane, float4 (&q)[Geo<N>::kLoads]) {
#pragma unroll
    for (int i = 0; i < Geo<N>::kLoads; ++i) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
#pragma unroll
        for (int i = 0; i < Geo<N>::kLoads; ++i) {
            const int f = i * 64 + lane;
            if ((i + 1) * 64 <= Geo<N>::NN4 || f < Geo<N>::NN4) q[i] = IoTraits<IO>::load4(gp + 4 * f);
        }
    }
}

// compile-time loop: every index below is a constant in the AST, so register arrays are
// scalarised before any select-of-loads folding can turn them into dynamic indexing
template <int I, int E, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, E>(f);
    }
}

// Plane I/O goes through a wave-private LDS image whose rows are in natural order and whose
// columns are in half order (half_pos): the coalesced side (float4 = two pairs of columns)
// puts each pair where its half expects it, swapping the two floats of a pair that lands in
// the mirrored high half; the row side then moves whole half rows with ds_*_b128.
template <int N>
__device__ __forceinline__ int pair_slot(int row, int w, bool& swapped) {
    swapped = w >= N / 2;                       // w is even and N/2 is even-or-odd*2: pairs never straddle
    return row * kLineStride + (swapped ? kHalfPad + (N - 2 - w) : w);
}

template <int N>
__device__ __forceinline__ void plane_to_rows(const float4 (&q)[Geo<N>::kLoads], float* T, int lane, int l, int hf,
                                              float (&v)[Geo<N>::M]) {
    constexpr int M = Geo<N>::M;
#pragma unroll
    for (int i = 0; i < Geo<N>::kLoads; ++i) {
        const int f = i * 64 + lane;
        if ((i + 1) * 64 <= Geo<N>::NN4 || f < Geo<N>::NN4) {
            const int row = f / Geo<N>::R4, w0 = 4 * (f % Geo<N>::R4);
            bool s0, s1;
            const int a0 = pair_slot<N>(row, w0, s0), a1 = pair_slot<N>(row, w0 + 2, s1);
            *reinterpret_cast<float2*>(&T[a0]) = s0 ? make_float2(q[i].y, q[i].x) : make_float2(q[i].x, q[i].y);
            *reinterpret_cast<float2*>(&T[a1]) = s1 ? make_float2(q[i].w, q[i].z) : make_float2(q[i].z, q[i].w);
        }
    }
    __builtin_amdgcn_wave_barrier();
    const float* src = T + l * kLineStride + hf * kHalfPad;
#pragma unroll
    for (int i = 0; i < (M + 3) / 4; ++i) {
        const float4 x = *reinterpret_cast<const float4*>(src + 4 * i);
        v[4 * i] = x.x;
        if (4 * i + 1 < M) v[4 * i + 1] = x.y;
        if (4 * i + 2 < M) v[4 * i + 2] = x.z;
        if (4 * i + 3 < M) v[4 * i + 3] = x.w;
    }
    __builtin_amdgcn_wave_barrier();
}

template <int N, typename IO>
__device__ __forceinline__ void rows_to_plane(const float (&v)[Geo<N>::M], float* T, int lane, int l, int hf,
                                              IO* gp, bool valid) {
    constexpr int M = Geo<N>::M;
    if (l < N) {
        float* dst = T + l * kLineStride + hf * kHalfPad;
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
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < Geo<N>::kLoads; ++i) {
        const int f = i * 64 + lane;
        if ((i + 1) * 64 <= Geo<N>::NN4 || f < Geo<N>::NN4) {
            const int row = f / Geo<N>::R4, w0 = 4 * (f % Geo<N>::R4);
            bool s0, s1;
            const int a0 = pair_slot<N>(row, w0, s0), a1 = pair_slot<N>(row, w0 + 2, s1);
            const float2 p0 = *reinterpret_cast<const float2*>(&T[a0]);
            const float2 p1 = *reinterpret_cast<const float2*>(&T[a1]);
            float4 x;
            x.x = s0 ? p0.y : p0.x; x.y = s0 ? p0.x : p0.y;
            x.z = s1 ? p1.y : p1.x; x.w = s1 ? p1.x : p1.y;
            if (valid) IoTraits<IO>::store4(gp + 4 * f, x);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// row layout <-> column layout of one plane (an involution; same code both ways).
// The image used here is private to this exchange, so its rows and its columns are both kept
// in half order (row index hf*M + k, column index half_pos): every access is then
// lane base + compile-time offset.
template <int N>
__device__ __forceinline__ void relayout(float (&v)[Geo<N>::M], float* T, int l, int hf) {
    constexpr int M = Geo<N>::M;
    if (l < N) {
        const int mypos = (l < M) ? l : kHalfPad + (N - 1 - l);
        float* dst = T + hf * M * kLineStride + mypos;
#pragma unroll
        for (int k = 0; k < M; ++k) dst[k * kLineStride] = v[k];
    }
    __builtin_amdgcn_wave_barrier();
    const int myrow = (l < M) ? l : M + (N - 1 - l);
    const float* src = T + myrow * kLineStride + hf * kHalfPad;
#pragma unroll
    for (int i = 0; i < (M + 3) / 4; ++i) {
        const float4 x = *reinterpret_cast<const float4*>(src + 4 * i);
        v[4 * i] = x.x;
        if (4 * i + 1 < M) v[4 * i + 1] = x.y;
        if (4 * i + 2 < M) v[4 * i + 2] = x.z;
        if (4 * i + 3 < M) v[4 * i + 3] = x.w;
    }
    __builtin_amdgcn_wave_barrier();
}

template <int M>
__device__ __forceinline__ void load_half(const float* src, float (&dst)[M]) {
#pragma unroll
    for (int i = 0; i < (M + 3) / 4; ++i) {
        const float4 x = *reinterpret_cast<const float4*>(src + 4 * i);
        dst[4 * i] = x.x;
        if (4 * i + 1 < M) dst[4 * i + 1] = x.y;
        if (4 * i + 2 < M) dst[4 * i + 2] = x.z;
        if (4 * i + 3 < M) dst[4 * i + 3] = x.w;
    }
}

// ---- forward: (A + eps I) x = d on J planes, two-sided ---------------------------------
template <int M, int J>
__device__ __forceinline__ void solve_fwd(float (&v)[J][M], const float* rec, int l, int hf) {
    float e[M], inv[M];
    load_half<M>(rec + kRecE + l * kLineStride + hf * kHalfPad, e);
    load_half<M>(rec + kRecInv + l * kLineStride + hf * kHalfPad, inv);
    const float jn = rec[kRecJn + l];
    // elimination from my end inwards: D_k = d_k*inv_k + e_k*D_{k-1}
#pragma unroll
    for (int k = 0; k < M; ++k) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float t = v[j][k] * inv[k];
            v[j][k] = (k == 0) ? t : fmaf(e[k], v[j][k - 1], t);
        }
    }
    // junction: x_in = (D_in + e_in * D_in(partner)) / (1 - e_t e_b)
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const float other = xchg_half(v[j][M - 1]);
        v[j][M - 1] = fmaf(e[M - 1], other, v[j][M - 1]) * jn;
    }
    // substitution outwards: x_k = D_k + e_k*x_{k+1}
#pragma unroll
    for (int k = M - 2; k >= 0; --k) {
#pragma unroll
        for (int j = 0; j < J; ++j) v[j][k] = fmaf(e[k], v[j][k + 1], v[j][k]);
    }
}

__device__ __forceinline__ int ck_slot(const unsigned long long (&ck)[2], int s) {
    return s < 64 ? __builtin_popcountll(ck[0] & ((1ull << s) - 1ull))
                  : __builtin_popcountll(ck[0]) + __builtin_popcountll(ck[1] & ((1ull << (s - 64)) - 1ull));
}

template <int N, int J, typename IO>
__global__ __launch_bounds__(kThreads) void adi_fwd_kernel(SweepArgs a) {
    constexpr int M = Geo<N>::M;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cbuf = smem;                                   // [2][kRecFwd]
    float* tbuf = smem + 2 * kRecFwd;                     // [kWaves][kImage]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = lane >> 5, l = lane & 31;
    const int c = blockIdx.x % a.C, g = blockIdx.x / a.C;
    float* T = tbuf + wave * kImage;
    const IO* u = static_cast<const IO*>(a.in0);
    IO* y = static_cast<IO*>(a.out);
    constexpr int PPI = kWaves * J;                       // planes per workgroup iteration
    const int nchunk = (a.B + PPI - 1) / PPI;
    const size_t plane = (size_t)N * N;

    // rows >= N of the wave images are never written: zero them once so idle lanes read zeros
    for (int e = tid; e < kWaves * kImage; e += kThreads) tbuf[e] = 0.f;
    float4 st0 = make_float4(0.f, 0.f, 0.f, 0.f), st1 = st0, st2 = st0;
    unsigned n = 0;                                       // running sweep counter (buffer parity)
    stage_load<kRecFwd>(a.coef + ((size_t)0 * a.C + c) * kRecStride, tid, st0, st1, st2);
    stage_store<kRecFwd>(cbuf, tid, st0, st1, st2);
    __syncthreads();

    for (int q = g; q < nchunk; q += a.G) {
        float v[J][M];
        {
            float4 raw[J][Geo<N>::kLoads];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int b = q * PPI + wave * J + j;
                plane_fetch<N, IO>(u + ((size_t)b * a.C + c) * plane, b < a.B, lane, raw[j]);
            }
#pragma unroll
            for (int j = 0; j < J; ++j) plane_to_rows<N>(raw[j], T, lane, l, hf, v[j]);
        }
        const bool more = q + a.G < nchunk;
        for (int s = 0; s < a.S; ++s, ++n) {
            const int snext = (s + 1 < a.S) ? s + 1 : 0;
            const bool pre = (s + 1 < a.S) || more;
            if (pre) stage_load<kRecFwd>(a.coef + ((size_t)snext * a.C + c) * kRecStride, tid, st0, st1, st2);
            const float* rec = cbuf + (n & 1) * kRecFwd;
            const int axs = a.tab->axis[s];
            if (axs == PDE_AXIS_Y) {
#pragma unroll
                for (int j = 0; j < J; ++j) relayout<N>(v[j], T, l, hf);
            }
            solve_fwd<M, J>(v, rec, l, hf);
            if (axs == PDE_AXIS_Y) {
#pragma unroll
                for (int j = 0; j < J; ++j) relayout<N>(v[j], T, l, hf);
            }
            if ((a.ck[s >> 6] >> (s & 63)) & 1ull) {      // backward pre-pass: park this state (fp32)
                float* slot = a.ckpt + (size_t)ck_slot(a.ck, s) * a.B * a.C * plane;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const int b = q * PPI + wave * J + j;
                    rows_to_plane<N, float>(v[j], T, lane, l, hf, slot + ((size_t)b * a.C + c) * plane, b < a.B);
                }
            }
            if (pre) stage_store<kRecFwd>(cbuf + ((n + 1) & 1) * kRecFwd, tid, st0, st1, st2);
            __syncthreads();
        }
        if (y != nullptr) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int b = q * PPI + wave * J + j;
                rows_to_plane<N, IO>(v[j], T, lane, l, hf, y + ((size_t)b * a.C + c) * plane, b < a.B);
            }
        }
    }
}

// ---- backward -------------------------------------------------------------------------
// adjoint two-sided solve (A + eps I)^T g = r on J planes, in place.
template <int M, int J>
__device__ __forceinline__ void solve_adj(float (&r)[J][M], const float* rec, int l, int hf) {
    __builtin_amdgcn_sched_barrier(0);
    float e[M];
    load_half<M>(rec + kRecE + l * kLineStride + hf * kHalfPad, e);
    const float jn = rec[kRecJn + l];
    // H_k = r_k + e_{k-1} H_{k-1}
#pragma unroll
    for (int k = 1; k < M; ++k) {
#pragma unroll
        for (int j = 0; j < J; ++j) r[j][k] = fmaf(e[k - 1], r[j][k - 1], r[j][k]);
    }
    // junction: G_in = (H_in + e_in(partner) H_in(partner)) / (1 - e_t e_b)
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const float pv = xchg_half(e[M - 1] * r[j][M - 1]);
        r[j][M - 1] = (r[j][M - 1] + pv) * jn;
    }
    // G_k = H_k + e_{k+1} G_{k+1};   g_k = inv_k G_k
#pragma unroll
    for (int k = M - 2; k >= 0; --k) {
#pragma unroll
        for (int j = 0; j < J; ++j) r[j][k] = fmaf(e[k + 1], r[j][k + 1], r[j][k]);
    }
    __builtin_amdgcn_sched_barrier(0);
    float inv[M];
    load_half<M>(rec + kRecInv + l * kLineStride + hf * kHalfPad, inv);
#pragma unroll
    for (int k = 0; k < M; ++k) {
#pragma unroll
        for (int j = 0; j < J; ++j) r[j][k] *= inv[k];
    }
    __builtin_amdgcn_sched_barrier(0);
}

// After an x sweep has been undone on the adjoint (g in r[]), use the sweep's OUTPUT state
// x to (1) add g.(Lx) to the coefficient-gradient sums, (2) rebuild the sweep's input
// x_prev = (1+eps) x + kap.(Lx).  L = Neumann second difference along the row.
// MASKED: the clamp mask of this channel changes with time, so the sum over sweeps cannot be
// masked (and un-smoothed) once at the end: do both here, per sweep.
template <int M, int J, bool MASKED>
__device__ __forceinline__ void state_x(const float (&g)[J][M], float (&x)[J][M], float (&acc)[M],
                                        const float* rec, int l, int hf, float one_eps, int smooth) {
    float kap[M];
    load_half<M>(rec + kRecKapX + l * kLineStride + hf * kHalfPad, kap);
    float msk[MASKED ? M : 1];
    if constexpr (MASKED) load_half<M>(rec + kRecMaskX + l * kLineStride + hf * kHalfPad, msk);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        float gq[MASKED ? M : 1];
        float xo_next = xchg_half(x[j][M - 1]);          // inner neighbour of k = M-1
#pragma unroll
        for (int k = M - 1; k >= 0; --k) {
            const float xo = x[j][k];
            float q = (k == 0) ? xo - xo_next : fmaf(2.0f, xo, -x[j][k - 1]) - xo_next;
            if constexpr (MASKED) gq[k] = g[j][k] * q;
            else acc[k] = fmaf(g[j][k], q, acc[k]);
            x[j][k] = fmaf(kap[k], q, xo * one_eps);
            xo_next = xo;
        }
        if constexpr (MASKED) {
            if (smooth) {                                // transpose of the replicate 3-tap average (x1/3 later)
                const float gin = xchg_half(gq[M - 1]);
#pragma unroll
                for (int k = 0; k < M; ++k) {
                    float z = (k == 0) ? 2.0f * gq[0] : gq[k] + gq[k - 1];
                    z += (k < M - 1) ? gq[k + 1] : gin;
                    acc[k] = fmaf(msk[k], z, acc[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < M; ++k) acc[k] = fmaf(msk[k], gq[k], acc[k]);
            }
        }
    }
}

// Same for a y sweep, with the state (and g) in ROW layout: the second difference (and, when
// MASKED, the transposed smoothing) runs across lanes (rows h-1, h+1 = lanes l-1, l+1 of the
// same half).
template <int N, int J, bool MASKED>
__device__ __forceinline__ void state_y(const float (&g)[J][N / 2], float (&x)[J][N / 2], float (&acc)[N / 2],
                                        const float* rec, int l, int hf, float one_eps, int smooth) {
    constexpr int M = N / 2;
    float kap[M];
    load_half<M>(rec + kRecKapX + l * kLineStride + hf * kHalfPad, kap);
    float msk[MASKED ? M : 1];
    if constexpr (MASKED) load_half<M>(rec + kRecMaskX + l * kLineStride + hf * kHalfPad, msk);
    const bool edge = (l == 0 || l == N - 1);
    const float kk = edge ? 1.0f : 2.0f;
    const float kz = edge ? 2.0f : 1.0f;
    const float mu = (l > 0) ? 1.0f : 0.0f;
    const float md = (l < N - 1) ? 1.0f : 0.0f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int k = 0; k < M; ++k) {
            const float xo = x[j][k];
            const float up = dpp_move<kDppWaveShr1>(xo);
            const float dn = dpp_move<kDppWaveShl1>(xo);
            float q = kk * xo;
            q = fmaf(-mu, up, q);
            q = fmaf(-md, dn, q);
            if constexpr (MASKED) {
                const float gq = g[j][k] * q;
                float z = gq;
                if (smooth) {
                    const float zu = dpp_move<kDppWaveShr1>(gq);
                    const float zd = dpp_move<kDppWaveShl1>(gq);
                    z = kz * gq;
                    z = fmaf(mu, zu, z);
                    z = fmaf(md, zd, z);
                }
                acc[k] = fmaf(msk[k], z, acc[k]);
            } else {
                acc[k] = fmaf(g[j][k], q, acc[k]);
            }
            x[j][k] = fmaf(kap[k], q, xo * one_eps);
        }
    }
}

template <int N, int J, typename IO>
__device__ __forceinline__ void load_planes(const IO* base, int q, int wave, int lane, int l, int hf, int B, int C,
                                            int c, float* T, float (&v)[J][N / 2]) {
    constexpr int PPI = kWaves * J;
    float4 raw[J][Geo<N>::kLoads];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int b = q * PPI + wave * J + j;
        plane_fetch<N, IO>(base + ((size_t)b * C + c) * (size_t)(N * N), b < B, lane, raw[j]);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) plane_to_rows<N>(raw[j], T, lane, l, hf, v[j]);
}

template <int N, int J, typename IO, bool MASKED>
__global__ __launch_bounds__(kThreads) void adi_bwd_kernel(SweepArgs a) {
    constexpr int M = Geo<N>::M;
    constexpr int REC = MASKED ? kRecStride : kRecBwd;
    const int c = blockIdx.x % a.C, g = blockIdx.x / a.C;
    if ((a.varying[c] != 0) != MASKED) return;            // the other instantiation owns this channel
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cbuf = smem;                                   // [2][REC]
    float* tbuf = smem + 2 * REC;                         // [kWaves][kImage]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = lane >> 5, l = lane & 31;
    float* T = tbuf + wave * kImage;
    const IO* gy = static_cast<const IO*>(a.in0);
    const IO* yy = static_cast<const IO*>(a.in1);
    IO* gu = static_cast<IO*>(a.out);
    constexpr int PPI = kWaves * J;
    const int nchunk = (a.B + PPI - 1) / PPI;
    const size_t plane = (size_t)N * N;
    auto ck_bit = [&](int s) { return (int)((a.ck[s >> 6] >> (s & 63)) & 1ull); };

    float Ax[M], Tx[M], Ay[M], Ty[M];
#pragma unroll
    for (int k = 0; k < M; ++k) Ax[k] = Tx[k] = Ay[k] = Ty[k] = 0.f;

    for (int e = tid; e < kWaves * kImage; e += kThreads) tbuf[e] = 0.f;
    float4 st0 = make_float4(0.f, 0.f, 0.f, 0.f), st1 = st0, st2 = st0;
    unsigned n = 0;
    stage_load<REC>(a.coef + ((size_t)(a.S - 1) * a.C + c) * kRecStride, tid, st0, st1, st2);
    stage_store<REC>(cbuf, tid, st0, st1, st2);
    __syncthreads();

    for (int q = g; q < nchunk; q += a.G) {
        float r[J][M], x[J][M];
        load_planes<N, J, IO>(gy, q, wave, lane, l, hf, a.B, a.C, c, T, r);
        load_planes<N, J, IO>(yy, q, wave, lane, l, hf, a.B, a.C, c, T, x);
        // The time-weighted sums use summation by parts over the whole processing sequence
        // (all chunks, sweeps in decreasing time):  sum_i tau_i G_i = sum_i (tau_i - tau_{i+1}) R_i
        // with R_i the running sum of g.q and tau_{i+1} the time of the next processed sweep of
        // the same axis (0 after the very last one).  So Ax/Ay double as R and are never reset.
        const bool more = q + a.G < nchunk;
        for (int s = a.S - 1; s >= 0; --s, ++n) {
            const int snext = (s > 0) ? s - 1 : a.S - 1;
            const bool pre = (s > 0) || more;
            if (pre) stage_load<REC>(a.coef + ((size_t)snext * a.C + c) * kRecStride, tid, st0, st1, st2);
            const float* rec = cbuf + (n & 1) * REC;
            const int axs = a.tab->axis[s];
            float dts = a.tab->dts[s];
            if (more && s == a.tab->first_s[axs]) dts -= a.tab->t_last[axs];
            if (axs == PDE_AXIS_Y) {
#pragma unroll
                for (int j = 0; j < J; ++j) relayout<N>(r[j], T, l, hf);
                solve_adj<M, J>(r, rec, l, hf);
#pragma unroll
                for (int j = 0; j < J; ++j) relayout<N>(r[j], T, l, hf);
                state_y<N, J, MASKED>(r, x, Ay, rec, l, hf, a.one_eps, a.smooth3);
                if (dts != 0.f) {
#pragma unroll
                    for (int k = 0; k < M; ++k) Ty[k] = fmaf(dts, Ay[k], Ty[k]);
                }
            } else {
                solve_adj<M, J>(r, rec, l, hf);
                state_x<M, J, MASKED>(r, x, Ax, rec, l, hf, a.one_eps, a.smooth3);
                if (dts != 0.f) {
#pragma unroll
                    for (int k = 0; k < M; ++k) Tx[k] = fmaf(dts, Ax[k], Tx[k]);
                }
            }
            // x now holds the rebuilt state after sweep s-1; take the checkpoint instead if there is one
            if (s > 0 && ck_bit(s - 1)) {
                const float* slot = a.ckpt + (size_t)ck_slot(a.ck, s - 1) * a.B * a.C * plane;
                load_planes<N, J, float>(slot, q, wave, lane, l, hf, a.B, a.C, c, T, x);
            }
            if (pre) stage_store<REC>(cbuf + ((n + 1) & 1) * REC, tid, st0, st1, st2);
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int b = q * PPI + wave * J + j;
            rows_to_plane<N, IO>(r[j], T, lane, l, hf, gu + ((size_t)b * a.C + c) * plane, b < a.B);
        }
    }

    // deterministic reduction of the four sums over the waves of this workgroup
    __syncthreads();
    float* dst = a.part + ((size_t)g * a.C + c) * 4 * kImage;
#pragma unroll
    for (int arr = 0; arr < 4; ++arr) {
        float* row = T + l * kLineStride + hf * kHalfPad;
#pragma unroll
        for (int k = 0; k < M; ++k) row[k] = (arr == 0) ? Ax[k] : (arr == 1) ? Tx[k] : (arr == 2) ? Ay[k] : Ty[k];
        __syncthreads();
        for (int e = tid; e < kImage; e += kThreads) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) sum += tbuf[w * kImage + e];
            dst[arr * kImage + e] = sum;
        }
        __syncthreads();
    }
}

// ---- parameter-gradient epilogue: one workgroup per channel ------------------------------
struct PgradArgs {
    const float* part;      // [G][C][4][kImage]
    const float* ab;
    const float* bb;
    const float* as;
    const float* bs;
    float* g_ab;
    float* g_bb;
    float* g_as;
    float* g_bs;
    const int* varying;     // [C] 1: the masked kernel already applied mask and transposed smoothing per sweep
    int C, N, S, G;
    int smooth3, has_max, accumulate;
    float cmax, eps;
    float wx, wy;           // delta/h2 of the x / y sweeps
    float t_first[2];
    int have_axis[2];
};

__global__ __launch_bounds__(1024) void adi_pgrad_kernel(PgradArgs a) {
    __shared__ float sm[4][PDE_MAX_N][PDE_MAX_N + 1];
    const int N = a.N, c = blockIdx.x;
    const int tid = threadIdx.x;
    const int h = tid / N, w = tid % N;
    const bool act = tid < N * N;
    if (act) {
        const int e = h * kLineStride + half_pos(w, N);
        for (int arr = 0; arr < 4; ++arr) {
            float sum = 0.f;
            for (int g = 0; g < a.G; ++g) sum += a.part[(((size_t)g * a.C + c) * 4 + arr) * kImage + e];
            sm[arr][h][w] = sum;
        }
    }
    __syncthreads();
    if (!act) return;
    const size_t off = ((size_t)c * N + h) * N + w;
    for (int ax = 0; ax < 2; ++ax) {
        const float wgt = -(ax == 0 ? a.wx : a.wy);
        float gb, gs;
        {
            const float(*A)[PDE_MAX_N + 1] = sm[2 * ax];
            const float(*Tm)[PDE_MAX_N + 1] = sm[2 * ax + 1];
            if (a.varying[c]) {
                const float f = a.smooth3 ? wgt * (1.0f / 3.0f) : wgt;
                gb = A[h][w] * f;
                gs = Tm[h][w] * f;
            } else if (a.smooth3) {
                // transpose of the replicate-padded 3-tap average along the solve axis:
                // theta_bar_j = (1/3)(c_j q_j + q_{j-1} + q_{j+1}),  c_j = 2 at the two ends, else 1
                const int i = (ax == 0) ? w : h;
                auto at = [&](const float(*Q)[PDE_MAX_N + 1], int ii) { return (ax == 0) ? Q[h][ii] : Q[ii][w]; };
                const float cj = (i == 0 || i == N - 1) ? 2.0f : 1.0f;
                float sa = at(A, i) * cj, stt = at(Tm, i) * cj;
                if (i > 0) { sa += at(A, i - 1); stt += at(Tm, i - 1); }
                if (i < N - 1) { sa += at(A, i + 1); stt += at(Tm, i + 1); }
                gb = sa * (1.0f / 3.0f) * wgt;
                gs = stt * (1.0f / 3.0f) * wgt;
            } else {
                gb = A[h][w] * wgt;
                gs = Tm[h][w] * wgt;
            }
        }
        // clamp pass-through mask, the same for every sweep of a channel that gets here unflagged
        if (!a.varying[c]) {
            const float base = (ax == 0 ? a.ab : a.bb)[off];
            const float slope = (ax == 0 ? a.as : a.bs)[off];
            const float th = base + slope * a.t_first[ax];
            const bool pass = (th >= a.eps) && (!a.has_max || th <= a.cmax);
            if (!pass || !a.have_axis[ax]) { gb = 0.f; gs = 0.f; }
        }
        float* ob = (ax == 0) ? a.g_ab : a.g_bb;
        float* os = (ax == 0) ? a.g_as : a.g_bs;
        if (a.accumulate) { ob[off] += gb; os[off] += gs; }
        else { ob[off] = gb; os[off] = gs; }
    }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int check_desc(const PdeAdiDesc* d) {
    if (!d) return PDE_E_BADARG;
    if (d->B <= 0 || d->C <= 0 || d->num_sweeps <= 0) return PDE_E_BADARG;
    if (d->N < 8 || d->N > PDE_MAX_N || (d->N % 4) != 0) return PDE_E_UNSUPPORTED_N;
    if (d->num_sweeps > PDE_MAX_SWEEPS) return PDE_E_TOO_MANY_SWEEPS;
    if (d->io_dtype != PDE_IO_F32 && d->io_dtype != PDE_IO_BF16) return PDE_E_BADARG;
    for (int s = 0; s < d->num_sweeps; ++s)
        if (d->sweep[s].axis != PDE_AXIS_X && d->sweep[s].axis != PDE_AXIS_Y) return PDE_E_BADARG;
    return PDE_OK;
}

size_t coef_bytes(const PdeAdiDesc* d) {
    return align_up((size_t)d->num_sweeps * d->C * kRecStride * sizeof(float), 256);
}
size_t tab_bytes() { return align_up(sizeof(SweepTab), 256); }
size_t flag_bytes(const PdeAdiDesc* d) { return align_up((size_t)d->C * sizeof(int), 256); }

constexpr int kJFwd = 4;
constexpr int kJBwd = 1;

int groups_per_channel(const PdeAdiDesc* d, int planes_per_iter, int wg_per_cu) {
    const int nchunk = (d->B + planes_per_iter - 1) / planes_per_iter;
    int G = (256 * wg_per_cu + d->C - 1) / d->C;           // fill the chip once
    if (G < 1) G = 1;
    if (G > nchunk) G = nchunk;
    return G;
}

void fill_factor_args(FactorArgs& fa, const PdeAdiDesc* d, const float* ab, const float* bb, const float* as,
                      const float* bs) {
    fa.tab = nullptr; fa.varying = nullptr; fa.coef = nullptr; fa.kmax = nullptr;
    fa.ab = ab; fa.bb = bb; fa.as = as; fa.bs = bs;
    fa.C = d->C; fa.N = d->N; fa.S = d->num_sweeps;
    fa.smooth3 = d->smooth3; fa.has_max = d->has_clamp_max; fa.cmax = d->clamp_max; fa.eps = d->eps;
    fa.t_first[0] = fa.t_first[1] = 0.f;
    bool seen[2] = {false, false};
    for (int s = 0; s < d->num_sweeps; ++s) {
        fa.sweep[s] = d->sweep[s];
        const int ax = d->sweep[s].axis;
        if (!seen[ax]) { seen[ax] = true; fa.t_first[ax] = d->sweep[s].t; }
    }
}

int launch_factor(const PdeAdiDesc* d, const float* ab, const float* bb, const float* as, const float* bs,
                  float* coef, SweepTab* tab, int* varying, hipStream_t st) {
    FactorArgs fa;
    fill_factor_args(fa, d, ab, bb, as, bs);
    fa.coef = coef; fa.tab = tab; fa.varying = varying;
    hipLaunchKernelGGL(adi_factor_kernel, dim3(d->num_sweeps * ((d->C + 1) / 2)), dim3(64), 0, st, fa);
    return check_launch();
}

struct PendingEvent { hipEvent_t e0, e1; bool fwd; };
std::vector<PendingEvent>& pending() {
    static std::vector<PendingEvent> v;
    return v;
}
std::mutex& launch_mutex() {
    static std::mutex m;
    return m;
}

template <typename K>
int launch_sweep(K kernel, const SweepArgs& sa, int grid, size_t lds, hipStream_t st, bool is_fwd, bool timed) {
    static std::vector<const void*> configured;
    {
        std::lock_guard<std::mutex> lk(launch_mutex());
        const void* key = reinterpret_cast<const void*>(kernel);
        bool seen = false;
        for (auto p : configured) seen |= (p == key);
        if (!seen) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return PDE_E_LAUNCH;
            configured.push_back(key);
        }
    }
    Timing& tm = timing();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool rec = tm.on && timed;
    if (rec) {
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, st);
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), lds, st, sa);
    const int rc = check_launch();
    if (rec) {
        (void)hipEventRecord(e1, st);                     // resolved later, in pde_timing_read
        std::lock_guard<std::mutex> lk(launch_mutex());
        pending().push_back({e0, e1, is_fwd});
    }
    return rc;
}

#define PDE_N_LIST PDE_CASE(8) PDE_CASE(12) PDE_CASE(16) PDE_CASE(20) PDE_CASE(24) PDE_CASE(28) PDE_CASE(32)

template <typename IO>
int dispatch_fwd(int N, const SweepArgs& sa, int grid, size_t lds, hipStream_t st, bool timed = true) {
    switch (N) {
#define PDE_CASE(NN) case NN: return launch_sweep(adi_fwd_kernel<NN, kJFwd, IO>, sa, grid, lds, st, true, timed);
        PDE_N_LIST
#undef PDE_CASE
    }
    return PDE_E_UNSUPPORTED_N;
}
template <typename IO, bool MASKED>
int dispatch_bwd(int N, const SweepArgs& sa, int grid, size_t lds, hipStream_t st) {
    switch (N) {
#define PDE_CASE(NN) case NN: return launch_sweep(adi_bwd_kernel<NN, kJBwd, IO, MASKED>, sa, grid, lds, st, false, !MASKED);
        PDE_N_LIST
#undef PDE_CASE
    }
    return PDE_E_UNSUPPORTED_N;
}

int count_ckpt(const uint64_t m[2]) { return m ? __builtin_popcountll(m[0]) + __builtin_popcountll(m[1]) : 0; }

}  // namespace
}  // namespace pde

using namespace pde;

extern "C" {

size_t pde_adi_forward_workspace_bytes(const PdeAdiDesc* d) {
    if (check_desc(d) != PDE_OK) return 0;
    return coef_bytes(d) + tab_bytes();
}

size_t pde_adi_backward_workspace_bytes(const PdeAdiDesc* d, int32_t num_checkpoints) {
    if (check_desc(d) != PDE_OK || num_checkpoints < 0) return 0;
    const int G = groups_per_channel(d, kWaves * kJBwd, 1);
    size_t b = coef_bytes(d) + tab_bytes() + flag_bytes(d);
    b += align_up((size_t)G * d->C * 4 * kImage * sizeof(float), 256);
    b += align_up((size_t)num_checkpoints * d->B * d->C * d->N * d->N * sizeof(float), 256);
    return b;
}

int pde_adi_forward(const PdeAdiDesc* d, const void* u, void* y, const float* alpha_base, const float* beta_base,
                    const float* alpha_slope, const float* beta_slope, void* workspace, size_t workspace_bytes,
                    void* stream) {
    int rc = check_desc(d);
    if (rc != PDE_OK) return rc;
    if (!u || !y || !alpha_base || !beta_base || !alpha_slope || !beta_slope || !workspace) return PDE_E_BADARG;
    if (workspace_bytes < pde_adi_forward_workspace_bytes(d) || ((uintptr_t)workspace & 15)) return PDE_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* coef = static_cast<float*>(workspace);
    SweepTab* tab = reinterpret_cast<SweepTab*>(static_cast<char*>(workspace) + coef_bytes(d));
    rc = launch_factor(d, alpha_base, beta_base, alpha_slope, beta_slope, coef, tab, nullptr, st);
    if (rc != PDE_OK) return rc;
    SweepArgs sa{};
    sa.in0 = u; sa.out = y; sa.coef = coef; sa.tab = tab;
    sa.B = d->B; sa.C = d->C; sa.S = d->num_sweeps;
    sa.G = groups_per_channel(d, kWaves * kJFwd, 2);
    sa.one_eps = 1.0f + d->eps;
    const size_t lds = (size_t)(2 * kRecFwd + kWaves * kImage) * sizeof(float);
    const int grid = sa.G * d->C;
    return d->io_dtype == PDE_IO_F32 ? dispatch_fwd<float>(d->N, sa, grid, lds, st)
                                     : dispatch_fwd<bf16_t>(d->N, sa, grid, lds, st);
}

int pde_adi_backward(const PdeAdiDesc* d, const void* gy, const void* y, const void* u, const uint64_t ckpt_mask[2],
                     void* gu, const float* alpha_base, const float* beta_base, const float* alpha_slope,
                     const float* beta_slope, float* g_alpha_base, float* g_beta_base, float* g_alpha_slope,
                     float* g_beta_slope, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_desc(d);
    if (rc != PDE_OK) return rc;
    if (!gy || !y || !gu || !alpha_base || !beta_base || !alpha_slope || !beta_slope || !g_alpha_base ||
        !g_beta_base || !g_alpha_slope || !g_beta_slope || !workspace)
        return PDE_E_BADARG;
    const int nck = count_ckpt(ckpt_mask);
    int Sf = 0;                                            // forward sweeps to recompute
    if (nck) {
        if (!u) return PDE_E_BADARG;
        for (int s = 0; s < d->num_sweeps; ++s)
            if ((ckpt_mask[s >> 6] >> (s & 63)) & 1ull) Sf = s + 1;
        for (int s = d->num_sweeps; s < 128; ++s)
            if ((ckpt_mask[s >> 6] >> (s & 63)) & 1ull) return PDE_E_BADARG;    // bit beyond the schedule
        if (Sf >= d->num_sweeps) return PDE_E_BADARG;      // the last state is y itself
    }
    if (workspace_bytes < pde_adi_backward_workspace_bytes(d, nck) || ((uintptr_t)workspace & 15)) return PDE_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int G = groups_per_channel(d, kWaves * kJBwd, 1);
    char* ws = static_cast<char*>(workspace);
    float* coef = reinterpret_cast<float*>(ws);           ws += coef_bytes(d);
    SweepTab* tab = reinterpret_cast<SweepTab*>(ws);      ws += tab_bytes();
    int* varying = reinterpret_cast<int*>(ws);            ws += flag_bytes(d);
    float* part = reinterpret_cast<float*>(ws);           ws += align_up((size_t)G * d->C * 4 * kImage * sizeof(float), 256);
    float* ckpt = reinterpret_cast<float*>(ws);
    if (hipMemsetAsync(varying, 0, flag_bytes(d), st) != hipSuccess) return PDE_E_LAUNCH;
    rc = launch_factor(d, alpha_base, beta_base, alpha_slope, beta_slope, coef, tab, varying, st);
    if (rc != PDE_OK) return rc;

    SweepArgs sa{};
    sa.in0 = gy; sa.in1 = y; sa.in2 = u; sa.out = gu; sa.coef = coef; sa.part = part; sa.tab = tab;
    sa.varying = varying; sa.ckpt = ckpt;
    sa.ck[0] = nck ? ckpt_mask[0] : 0ull; sa.ck[1] = nck ? ckpt_mask[1] : 0ull;
    sa.Sf = Sf; sa.smooth3 = d->smooth3;
    sa.B = d->B; sa.C = d->C; sa.S = d->num_sweeps; sa.G = G;
    sa.one_eps = 1.0f + d->eps;
    float wgt[2] = {0.f, 0.f};
    float tfirst[2] = {0.f, 0.f};
    bool have[2] = {false, false};
    for (int s = 0; s < d->num_sweeps; ++s) {
        const int ax = d->sweep[s].axis;
        const float w = d->sweep[s].delta / d->sweep[s].h2;
        if (have[ax] && w != wgt[ax]) return PDE_E_BADARG;   // one weight per axis (true for every reference variant)
        if (!have[ax]) tfirst[ax] = d->sweep[s].t;
        wgt[ax] = w; have[ax] = true;
    }
    if (nck) {
        // pre-pass: run the forward from u up to the last checkpointed sweep and park those states
        SweepArgs fa = sa;
        fa.in0 = u; fa.in1 = nullptr; fa.out = nullptr; fa.part = nullptr;
        fa.S = Sf;
        fa.G = groups_per_channel(d, kWaves * kJFwd, 2);
        const size_t lds_f = (size_t)(2 * kRecFwd + kWaves * kImage) * sizeof(float);
        rc = d->io_dtype == PDE_IO_F32 ? dispatch_fwd<float>(d->N, fa, fa.G * d->C, lds_f, st, false)
                                       : dispatch_fwd<bf16_t>(d->N, fa, fa.G * d->C, lds_f, st, false);
        if (rc != PDE_OK) return rc;
    }
    const int grid = G * d->C;
    const size_t lds_fast = (size_t)(2 * kRecBwd + kWaves * kImage) * sizeof(float);
    const size_t lds_mask = (size_t)(2 * kRecStride + kWaves * kImage) * sizeof(float);
    // two launches over the same grid: every workgroup leaves at once unless its channel belongs
    // to the instantiation (decided on the device by the factor kernel, no host round trip)
    if (d->io_dtype == PDE_IO_F32) {
        rc = dispatch_bwd<float, false>(d->N, sa, grid, lds_fast, st);
        if (rc == PDE_OK) rc = dispatch_bwd<float, true>(d->N, sa, grid, lds_mask, st);
    } else {
        rc = dispatch_bwd<bf16_t, false>(d->N, sa, grid, lds_fast, st);
        if (rc == PDE_OK) rc = dispatch_bwd<bf16_t, true>(d->N, sa, grid, lds_mask, st);
    }
    if (rc != PDE_OK) return rc;

    PgradArgs pa{};
    pa.part = part; pa.ab = alpha_base; pa.bb = beta_base; pa.as = alpha_slope; pa.bs = beta_slope;
    pa.g_ab = g_alpha_base; pa.g_bb = g_beta_base; pa.g_as = g_alpha_slope; pa.g_bs = g_beta_slope;
    pa.varying = varying;
    pa.C = d->C; pa.N = d->N; pa.S = d->num_sweeps; pa.G = G;
    pa.smooth3 = d->smooth3; pa.has_max = d->has_clamp_max; pa.accumulate = 0;
    pa.cmax = d->clamp_max; pa.eps = d->eps;
    pa.wx = wgt[0]; pa.wy = wgt[1];
    pa.t_first[0] = tfirst[0]; pa.t_first[1] = tfirst[1];
    pa.have_axis[0] = have[0]; pa.have_axis[1] = have[1];
    hipLaunchKernelGGL(adi_pgrad_kernel, dim3(d->C), dim3(1024), 0, st, pa);
    return check_launch();
}

int pde_adi_kappa_max(const PdeAdiDesc* d, const float* alpha_base, const float* beta_base, const float* alpha_slope,
                      const float* beta_slope, float* kappa_max, void* stream) {
    int rc = check_desc(d);
    if (rc != PDE_OK) return rc;
    if (!alpha_base || !beta_base || !alpha_slope || !beta_slope || !kappa_max) return PDE_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(kappa_max, 0, (size_t)d->num_sweeps * sizeof(float), st) != hipSuccess) return PDE_E_LAUNCH;
    FactorArgs fa;
    fill_factor_args(fa, d, alpha_base, beta_base, alpha_slope, beta_slope);
    fa.kmax = kappa_max;
    hipLaunchKernelGGL(adi_kmax_kernel, dim3(d->num_sweeps * ((d->C + 1) / 2)), dim3(64), 0, st, fa);
    return check_launch();
}

int pde_timing_enable(int32_t on) {
    timing().on = on != 0;
    if (on) { timing().fwd_ms = timing().bwd_ms = 0; timing().fwd_n = timing().bwd_n = 0; }
    return PDE_OK;
}

int pde_timing_read(double* fwd_ms_sum, int64_t* fwd_launches, double* bwd_ms_sum, int64_t* bwd_launches) {
    for (auto& p : pending()) {
        float ms = 0.f;
        (void)hipEventSynchronize(p.e1);
        (void)hipEventElapsedTime(&ms, p.e0, p.e1);
        if (p.fwd) { timing().fwd_ms += ms; timing().fwd_n++; } else { timing().bwd_ms += ms; timing().bwd_n++; }
        (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1);
    }
    pending().clear();
    if (fwd_ms_sum) *fwd_ms_sum = timing().fwd_ms;
    if (fwd_launches) *fwd_launches = timing().fwd_n;
    if (bwd_ms_sum) *bwd_ms_sum = timing().bwd_ms;
    if (bwd_launches) *bwd_launches = timing().bwd_n;
    return PDE_OK;
}

const char* pde_version(void) { return "pdecnn-hip 0.2 (gfx950)"; }

}  // extern "C"
