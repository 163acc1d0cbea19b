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
//     a small kernel computes it once per call, the sweep kernels bring one record per
//     sweep into a small LDS ring (forward: through registers, 3 slots, one barrier per
//     sweep; backward: LDS-DMA, 2*steps+1 slots, one barrier per TIME STEP) and all
//     waves of a workgroup — which share a channel — read it from there;
//   * "skew": the upper half of the waves of a workgroup runs one sweep behind the lower
//     half (its window of the ring is shifted back by one record), so that the halves
//     are never both in the y sweep — whose re-layouts load the LDS pipe — at the same
//     time;
//   * x <-> y re-layout goes through a wave-private LDS image, 4 B written and read
//     per element, no workgroup barrier;
//   * backward: adjoint two-sided solves; the states needed by the coefficient
//     gradients are rebuilt backwards from the output, x_{s-1} = (A_s + eps I) x_s,
//     so nothing is stored per sweep; the state never leaves the row layout (the
//     y-direction second difference is taken across lanes with DPP), only the adjoint
//     is re-laid out.  Parameter gradients are accumulated over the batch in registers
//     and reduced deterministically (no float atomics).
//   * what bounds the kernels (profiles/README.md): the SIMD's register-file port.  A
//     wave64 fp32 VALU instruction holds its SIMD for 4 cycles (v_pk_*_f32: 8), and so does every dword
//     an LDS read returns or an LDS write takes; cycles per wave ~ 4 x (VALU + LDS dwords) predicts both
//     kernels within 10 %.  Hence: records arrive by LDS-DMA (no register round trip), the forward keeps
//     FOUR planes per lane so that each coefficient read serves four planes, and the planes of a lane are
//     written as one value (Pack<J>) so that the packed form stays one flag away (make PACK=1: 4 % slower).
#include "pde_adi_dev.h"
#include "pde_adi_small.h"
#include "pde_adi_wide.h"
#include "pde_adi_launch.h"
#include "pde_adi_gen.h"
#include "pde_adi_asm.h"

#include <cmath>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

namespace pde {

Timing& timing() {
    static Timing t;
    return t;
}

int ensure_dynamic_lds(const void* kernel, int bytes, unsigned long long& done) {
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PDE_E_LAUNCH;
    std::lock_guard<std::mutex> lk(mu);
    if (!((done >> dev) & 1ull)) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return PDE_E_LAUNCH;
        done |= 1ull << dev;
    }
    return PDE_OK;
}

namespace {

// ------------------------------------------------------------------------------------
// factorisation kernel: one thread per (sweep, channel, line)
// ------------------------------------------------------------------------------------
struct FactorArgs {
    SweepTab* tab;
    int* varying;           // [C] 1 when a clamp mask of the channel differs between sweeps; followed in memory by the
                            // per-(sweep, channel) partials this kernel writes: int dpart[S][C], float kpart[S][C]
    float t_first[2];       // time of the earliest sweep of each axis
    const float* ab;
    const float* bb;
    const float* as;
    const float* bs;
    float* coef;            // [S][C][kRecAll]
    float* wide;            // optional [S][C][kWideRec]: lane-major copy of the records (pde_common.h), may be null
    float* kmax;            // optional [S] (atomic max of coeff), may be null
    int C, N, S;
    int win;                // sweeps per launch window: tab[w] describes sweeps w*win .. w*win+win-1 as ONE launch
    int smooth3, has_max;
    float cmax, eps;
    PdeSweep sweep[PDE_MAX_SWEEPS];
};

// One thread per (sweep, channel, HALF line): the two-sided factorisation makes the halves of a line
// independent up to the junction factor (one lane exchange), so a wave is 32 lines x 2 halves of one
// channel — the sweep kernels' own layout — and each thread's serial chain (divisions!) is N/2 long.
// A thread writes its half of every image row as 16-byte stores.  No LDS.  The row-layout images
// (KAPX, MASKX) of a y sweep are NOT transposed through memory: thread (h, hf) recomputes the
// coefficients of its half of row h directly from the parameters (three rows for the smoothed variants).
template <int N>
__device__ __forceinline__ void store_half_row(float* row, int hf, const float (&v)[N / 2]) {
    // image row = [16 floats: half seen from the low end][16: half seen from the high end][4 pad]
    float h[kHalfPad];
#pragma unroll
    for (int k = 0; k < kHalfPad; ++k) h[k] = 0.f;
#pragma unroll
    for (int k = 0; k < N / 2; ++k) h[k] = v[k];
    float4* dst = reinterpret_cast<float4*>(row + hf * kHalfPad);
#pragma unroll
    for (int q = 0; q < kHalfPad / 4; ++q) dst[q] = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
    if (hf) *reinterpret_cast<float4*>(row + 2 * kHalfPad) = make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ float clamp_theta(float th, const FactorArgs& a, bool& pass) {
    pass = (th >= a.eps) && (!a.has_max || th <= a.cmax);                  // clamp passes the gradient
    th = fmaxf(th, a.eps);
    if (a.has_max) th = fminf(th, a.cmax);
    return th;
}

constexpr int kFacLd = PDE_MAX_N + 1;     // LDS row stride of the staged parameter planes
// ints reserved for the per-channel flags in front of the per-(sweep, channel) partials (64-int granules)
__host__ __device__ inline int flag_ints(int C) { return (C + 63) / 64 * 64; }

template <int N>
__global__ __launch_bounds__(128) void adi_factor_kernel(FactorArgs a) {
    constexpr int m = N / 2;
    __shared__ float fsm[2 * 2 * kFacLd * PDE_MAX_N];      // per wave: theta plane, pass-flag plane
    __shared__ __attribute__((aligned(16))) float rsm[2 * kRecStride];   // per wave: the record, assembled before it is stored
    const int pairs = (a.C + 1) / 2;
    const int s = blockIdx.x / pairs;
    const int c = 2 * (blockIdx.x % pairs) + (threadIdx.x >> 6);           // one wave per channel of the pair
    const int hf = (threadIdx.x >> 5) & 1, line = threadIdx.x & 31;
    if (blockIdx.x == 0 && threadIdx.x < a.S) {   // publish the sweep table(s): one per launch window
        const int idx = threadIdx.x;
        const int w = idx / a.win, loc = idx % a.win, lo = w * a.win;
        SweepTab* tab = a.tab + w;
        float tprev = 0.f, tlast = 0.f;
        int first = -1;
        const int ax = a.sweep[idx].axis;
        for (int q = lo; q < lo + a.win; ++q) {
            if (a.sweep[q].axis != ax) continue;
            if (first < 0) first = q - lo;
            if (q < idx) tprev = a.sweep[q].t;
            tlast = a.sweep[q].t;
        }
        tab->ysc[loc] = powf(1.0f + a.eps, -(float)(a.win - 1 - loc));
        tab->axis[loc] = ax;
        tab->dts[loc] = a.sweep[idx].t - tprev;
        tab->first_s[ax] = first;
        tab->t_last[ax] = tlast;
        if (loc == 0) {                              // an axis the window does not contain
            const int other = 1 - ax;
            bool seen = false;
            for (int q = lo; q < lo + a.win; ++q) seen |= (a.sweep[q].axis == other);
            if (!seen) { tab->first_s[other] = -1; tab->t_last[other] = 0.f; }
        }
    }
    float kmax_lane = 0.f;
    int any_differs_w = 0;
    const PdeSweep sw = a.sweep[s];
    const bool xax = sw.axis == PDE_AXIS_X;
    if (c < a.C) {                                   // wave-uniform
        // the record is assembled in LDS (every lane writes its half rows) and then stored with linear, fully
        // coalesced 16-byte stores: half-row stores straight to memory (16 B per lane at a 144-byte stride) took 29 us
        float* rec = rsm + (threadIdx.x >> 6) * kRecStride;
        float* grec = a.coef + ((size_t)s * a.C + c) * kRecStride;
        float zero[m];
#pragma unroll
        for (int k = 0; k < m; ++k) zero[k] = 0.f;
        const bool idle = line >= N;
        // idle lines get zero rows: lanes beyond the plane run the same instruction stream on zeros, so
        // nothing they compute can leak a NaN through a lane exchange.  They follow the code below on a
        // clamped line index (for the lane exchange) and store zeros at the end.
        const int ln = idle ? N - 1 : line;
        const float* base = xax ? a.ab : a.bb;
        const float* slope = xax ? a.as : a.bs;
        const size_t cbase = (size_t)c * N * N;
        const float third = 1.0f / 3.0f;
        const float one_eps = 1.0f + a.eps, r1e = 1.0f / one_eps;
        // The wave first brings its channel's plane of theta = clamp(base + slope*t, eps[, max]) (mnist_test.py:33-42)
        // and of the clamp pass-through flags into LDS with coalesced 16-byte loads (stride 33: a line is read along
        // either axis without bank conflicts); a lane then picks its half line — and, for a y sweep, its half ROW and
        // the rows above and below for the smoothing — from there.  (Per-lane 4-byte loads at a 128-byte lane stride
        // straight from memory made this kernel 31 us.)
        float* TH = fsm + (threadIdx.x >> 6) * (2 * kFacLd * PDE_MAX_N);
        float* PS = TH + kFacLd * PDE_MAX_N;
        bool differs = false;
        {
            const int wl = threadIdx.x & 63;
            const float4* b4 = reinterpret_cast<const float4*>(base + cbase);
            const float4* s4 = reinterpret_cast<const float4*>(slope + cbase);
            for (int f = wl; f < N * N / 4; f += 64) {
                const float4 bv = b4[f], sv = s4[f];
                const int hh = (4 * f) / N, ww = (4 * f) % N;
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w}, ss[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bool ps, ps0;
                    const float t = clamp_theta(bb[q] + ss[q] * sw.t, a, ps);
                    (void)clamp_theta(bb[q] + ss[q] * a.t_first[sw.axis], a, ps0);
                    differs |= (ps != ps0);
                    TH[hh * kFacLd + ww + q] = t;
                    PS[hh * kFacLd + ww + q] = ps ? 1.0f : 0.0f;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // my unknowns: k = 0..m-1 at global index i(k) = hf ? N-1-k : k; theta also one step beyond my
        // inner end (index m or m-1) for the 3-tap smoothing
        float th[m + 1], pass[m];
#pragma unroll
        for (int k = 0; k <= m; ++k) {
            const int i = hf ? N - 1 - k : k;
            const int at = xax ? ln * kFacLd + i : i * kFacLd + ln;
            th[k] = TH[at];
            if (k < m) pass[k] = PS[at];
        }
        // per-(sweep, channel) partial, combined by adi_flags_kernel: 1920 atomics on the ONE cache line that holds the
        // 30 per-sweep maxima were 20 of this kernel's 29 us
        any_differs_w = __any(differs ? 1 : 0) != 0 ? 1 : 0;
        float kap[m];
        if (a.smooth3) {                             // mnist_test.py:135-149 (replicate ends): prev/next along the line
#pragma unroll
            for (int k = 0; k < m; ++k) {
                const float outer = th[k > 0 ? k - 1 : 0];          // towards my end (replicated at k = 0)
                const float inner = th[k + 1];
                const float prev = hf ? inner : outer, nxt = hf ? outer : inner;   // in increasing global index
                kap[k] = (prev * third + th[k] * third) + nxt * third;
            }
        } else {
#pragma unroll
            for (int k = 0; k < m; ++k) kap[k] = th[k];
        }
#pragma unroll
        for (int k = 0; k < m; ++k) {
            kap[k] = (kap[k] * sw.delta) / sw.h2;         // coeff = theta*dt/dx**2  mnist_test.py:83
            if (!idle) kmax_lane = fmaxf(kmax_lane, kap[k]);
        }
        // two-sided elimination of (A + eps I) from my end inwards:
        // den_k = b_k - kap_k * (kap_{k-1}/den_{k-1}) + eps       mnist_test.py:169,177
        float inv[m], ee[m], invb[m];
        {
            float e = 0.f;                               // kap_{k-1}/den_{k-1} of the outer neighbour
#pragma unroll
            for (int k = 0; k < m; ++k) {
                const float kp = kap[k];
                const float b = (k == 0) ? 1.0f + kp : 1.0f + 2.0f * kp;    // Neumann ends, mnist_test.py:88-93
                const float den = (b - kp * e) + a.eps;
                const float iv = 1.0f / den;
                e = kp * iv;
                inv[k] = iv;
                invb[k] = iv * one_eps;
                ee[k] = e;
            }
        }
        const float e_other = __shfl_xor(ee[m - 1], 32, 64);
        const float e_lo = hf ? e_other : ee[m - 1], e_hi = hf ? ee[m - 1] : e_other;
        if (hf == 0) rec[kG_Jn + line] = idle ? 0.f : 1.0f / (1.0f - e_lo * e_hi);
        float* row = rec + line * kLineStride;
        store_half_row<N>(row + kG_Inv, hf, idle ? zero : inv);
        store_half_row<N>(row + kG_E, hf, idle ? zero : ee);
        store_half_row<N>(row + kG_InvB, hf, idle ? zero : invb);
        // coefficient and clamp mask in ROW layout: row h = line, element (h,w) at half_pos(w)
        float kx[m], mx[m];
        if (xax) {
#pragma unroll
            for (int k = 0; k < m; ++k) { kx[k] = kap[k] * r1e; mx[k] = pass[k]; }
        } else {
            // y sweep: coefficient of row h = line along w (smoothing runs along h)
            const int h = ln;
            const int hm = h > 0 ? h - 1 : 0, hp = h + 1 < N ? h + 1 : N - 1;
#pragma unroll
            for (int k = 0; k < m; ++k) {
                const int w = hf ? N - 1 - k : k;
                float t0 = TH[h * kFacLd + w];
                if (a.smooth3) t0 = (TH[hm * kFacLd + w] * third + t0 * third) + TH[hp * kFacLd + w] * third;
                kx[k] = ((t0 * sw.delta) / sw.h2) * r1e;
                mx[k] = PS[h * kFacLd + w];
            }
        }
        store_half_row<N>(row + kG_KapX, hf, idle ? zero : kx);
        store_half_row<N>(row + kG_MaskX, hf, idle ? zero : mx);
        if (a.wide != nullptr) {                     // lane-major copy, straight from my registers
            float* wr = a.wide + ((size_t)s * a.C + c) * kWideRec;
            const int wl = threadIdx.x & 63;
            if (hf == 0) wr[kW_Jn + line] = idle ? 0.f : 1.0f / (1.0f - e_lo * e_hi);
            auto put = [&](int off, const float (&v)[m]) __attribute__((always_inline)) {
                float4* dst = reinterpret_cast<float4*>(wr + off) + wl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 x;
                    x.x = (4 * q < m && !idle) ? v[4 * q < m ? 4 * q : 0] : 0.f;
                    x.y = (4 * q + 1 < m && !idle) ? v[4 * q + 1 < m ? 4 * q + 1 : 0] : 0.f;
                    x.z = (4 * q + 2 < m && !idle) ? v[4 * q + 2 < m ? 4 * q + 2 : 0] : 0.f;
                    x.w = (4 * q + 3 < m && !idle) ? v[4 * q + 3 < m ? 4 * q + 3 : 0] : 0.f;
                    dst[q * 64] = x;
                }
            };
            put(kW_Inv, inv); put(kW_E, ee); put(kW_InvB, invb); put(kW_KapX, kx);
        }
        __builtin_amdgcn_wave_barrier();
        static_assert(kRecStride % 4 == 0, "record is a whole number of 16-byte pieces");
        const float4* src = reinterpret_cast<const float4*>(rec);
        float4* dst = reinterpret_cast<float4*>(grec);
        for (int f = threadIdx.x & 63; f < kRecStride / 4; f += 64) dst[f] = src[f];
    }
    if (a.varying && c < a.C) {                      // my wave's (sweep, channel) partials: plain stores
        for (int o = 32; o > 0; o >>= 1) kmax_lane = fmaxf(kmax_lane, __shfl_xor(kmax_lane, o, 64));
        if ((threadIdx.x & 63) == 0) {
            int* dpart = a.varying + flag_ints(a.C);
            float* kpart = reinterpret_cast<float*>(dpart + (size_t)a.S * a.C);
            dpart[(size_t)s * a.C + c] = any_differs_w;
            kpart[(size_t)s * a.C + c] = kmax_lane;
        }
    }
}

// second stage of the flags / maxima: one wave per sweep (kmax[s] = max over the channels), then one per channel
// (varying[c] = OR over the sweeps): a few parallel loads and a wave reduction each — a single workgroup walking the
// partials serially took 14 us
__global__ __launch_bounds__(64) void adi_flags_kernel(int* varying, float* kmax, float* kmax_mapped, int S, int C) {
    const int* dpart = varying + flag_ints(C);
    const float* kpart = reinterpret_cast<const float*>(dpart + (size_t)S * C);
    const int lane = threadIdx.x, b = blockIdx.x;
    if (b < S) {
        if (kmax == nullptr) return;
        float m = 0.f;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, kpart[(size_t)b * C + c]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) {
            kmax[b] = m;
            if (kmax_mapped) kmax_mapped[b] = m;    // the caller's pinned host buffer, written from here: no copy launch
        }
    } else {
        const int c = b - S;
        int f = 0;
        for (int s = lane; s < S; s += 64) f |= dpart[(size_t)s * C + c];
        f = __any(f) ? 1 : 0;
        if (lane == 0) varying[c] = f;
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
    // whole-layer kernels for C <= 4 (pde_adi_small.h): block C adds up the matrix and skip-weight partials
    const float* gm_part;   // [gm_blocks][C][kGmStride] or null
    float* gM;              // [C][C]
    float* g_skip;          // scalar or null
    float* g_w;             // scalar or null: d/d(weight of this layer's output in the launch's weighted sum)
    const float* skip_w;
    int gm_blocks;
};

// One workgroup per (channel, sum): the four sums of a channel (A_x, T_x, A_y, T_y) each give exactly one output
// (g_alpha_base, g_alpha_slope, g_beta_base, g_beta_slope), so they reduce independently — 4 C workgroups instead
// of C (the whole-layer kernels for C <= 4 leave up to 1024 groups to add up: 3 workgroups took 44 us there).
__global__ __launch_bounds__(1024) void adi_pgrad_kernel(PgradArgs a) {
    __shared__ float sm[PDE_MAX_N][PDE_MAX_N + 1];
    const int N = a.N;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x == 4 * a.C) {                 // (only launched when gm_part is given)
        // C*C matrix entries + the skip-weight term + the output-weight term: one WAVE per output, its lanes stride
        // over the workgroups' partials and meet in a butterfly (fixed order).  One THREAD per output walking the
        // partials one dependent load after the other took 33 us for 128 workgroups.
        const int lane = tid & 63, wv = tid >> 6, nout = a.C * a.C + 2;
        // These few scalars are whole-tensor sums that cancel heavily (seen: 2880 terms of total size 220 adding up to
        // 0.0144): the partials are added in DOUBLE precision, so that the summation adds nothing to their fp32 rounding.
        for (int o = wv; o < nout; o += 16) {
            double sum = 0.0;
            if (o < a.C * a.C) {
                const int i = o / a.C, j = o % a.C;
                for (int g = lane; g < a.gm_blocks; g += 64) sum += a.gm_part[((size_t)g * a.C + i) * kGmStride + j];
            } else {
                const int col = (o == a.C * a.C) ? kSmallMaxC : kGmStride - 1;
                for (int g = lane; g < a.gm_blocks; g += 64)
                    for (int i = 0; i < a.C; ++i) sum += a.gm_part[((size_t)g * a.C + i) * kGmStride + col];
            }
            for (int x = 32; x > 0; x >>= 1) sum += __shfl_xor(sum, x, 64);
            if (lane == 0) {
                if (o < a.C * a.C) a.gM[o] = (float)sum;
                else if (o == a.C * a.C) {
                    if (a.g_skip != nullptr) {
                        const double sg = 1.0 / (1.0 + exp(-(double)*a.skip_w));
                        *a.g_skip = (float)((1.0 - sg) * sum);     // the kernel's partials already carry one factor sigmoid
                    }
                } else if (a.g_w != nullptr) *a.g_w = (float)sum;
            }
        }
        return;
    }
    const int c = blockIdx.x >> 2, arr = blockIdx.x & 3, ax = arr >> 1;      // arr: 0 A_x, 1 T_x, 2 A_y, 3 T_y
    const int h = tid / N, w = tid % N;
    const bool act = tid < N * N;
    if (act) {
        // fixed summation order (deterministic), eight groups at a time so that eight loads are in flight per thread
        const int e = h * kLineStride + half_pos(w, N);
        const size_t gs = (size_t)a.C * 4 * kImage;
        const float* p0 = a.part + ((size_t)c * 4 + arr) * kImage + e;
        float s[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] = 0.f;
        int g = 0;
        for (; g + 8 <= a.G; g += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) s[q] += p0[(size_t)(g + q) * gs];
        }
        for (; g < a.G; ++g) s[0] += p0[(size_t)g * gs];
        sm[h][w] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
    __syncthreads();
    if (!act) return;
    const size_t off = ((size_t)c * N + h) * N + w;
    const float wgt = -(ax == 0 ? a.wx : a.wy);
    float gv;
    if (a.varying[c]) {
        gv = sm[h][w] * (a.smooth3 ? wgt * (1.0f / 3.0f) : wgt);
    } else if (a.smooth3) {
        // transpose of the replicate-padded 3-tap average along the solve axis:
        // theta_bar_j = (1/3)(c_j q_j + q_{j-1} + q_{j+1}),  c_j = 2 at the two ends, else 1
        const int i = (ax == 0) ? w : h;
        auto at = [&](int ii) { return (ax == 0) ? sm[h][ii] : sm[ii][w]; };
        const float cj = (i == 0 || i == N - 1) ? 2.0f : 1.0f;
        float sa = at(i) * cj;
        if (i > 0) sa += at(i - 1);
        if (i < N - 1) sa += at(i + 1);
        gv = sa * (1.0f / 3.0f) * wgt;
    } else {
        gv = sm[h][w] * wgt;
    }
    // clamp pass-through mask, the same for every sweep of a channel that gets here unflagged
    if (!a.varying[c]) {
        const float base = (ax == 0 ? a.ab : a.bb)[off];
        const float slope = (ax == 0 ? a.as : a.bs)[off];
        const float th = base + slope * a.t_first[ax];
        const bool pass = (th >= a.eps) && (!a.has_max || th <= a.cmax);
        if (!pass || !a.have_axis[ax]) gv = 0.f;
    }
    float* o = (arr == 0) ? a.g_ab : (arr == 1) ? a.g_as : (arr == 2) ? a.g_bb : a.g_bs;
    if (a.accumulate) o[off] += gv;
    else o[off] = gv;
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// fused: the line length has register-resident sweep kernels (pde_adi_dev.h); otherwise pde_adi_gen.hip serves the
// whole-schedule entry points (forward, backward, kappa_max) and the per-step / one-launch families refuse
bool fused_n(int N) { return N >= 8 && N <= PDE_MAX_N && (N % 4) == 0; }
int check_desc(const PdeAdiDesc* d, bool allow_generic = false) {
    if (!d) return PDE_E_BADARG;
    if (d->B <= 0 || d->C <= 0 || d->num_sweeps <= 0) return PDE_E_BADARG;
    if (!fused_n(d->N) && !(allow_generic && gen_n_ok(d->N))) return PDE_E_UNSUPPORTED_N;
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
size_t flag_bytes(const PdeAdiDesc* d) {       // flags [C] | differs partials [S][C] | maxima partials [S][C]
    return align_up(((size_t)flag_ints(d->C) + 2 * (size_t)d->num_sweeps * d->C) * sizeof(int), 256);
}


// Workgroups per channel: just enough groups to fill the chip once (measured: more groups per channel
// cost more in per-workgroup prologue/epilogue and partial sums than they gain in L2 locality).
// With C a multiple of 8 the XCD-ordered block map keeps a channel's groups on one XCD.
int env_int(const char* name, int dflt) {           // developer tuning knobs (tools/, never needed in production)
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
bool use_xcd_map(const PdeAdiDesc* d) { return (d->C % 8) == 0 && env_int("PDE_XCD", 1) != 0; }
int groups_per_channel(const PdeAdiDesc* d, int planes_per_iter, int wg_per_cu) {
    if (wg_per_cu < 1) wg_per_cu = 1;
    const int nchunk = (d->B + planes_per_iter - 1) / planes_per_iter;
    int G = (256 * wg_per_cu + d->C - 1) / d->C;
    G = env_int(planes_per_iter == kWaves * kJBwd ? "PDE_G_BWD" : "PDE_G_FWD", G);
    if (G < 1) G = 1;
    if (G > nchunk) G = nchunk;
    return G;
}

void fill_factor_args(FactorArgs& fa, const PdeAdiDesc* d, const float* ab, const float* bb, const float* as,
                      const float* bs) {
    fa.tab = nullptr; fa.varying = nullptr; fa.coef = nullptr; fa.kmax = nullptr; fa.wide = nullptr;
    fa.ab = ab; fa.bb = bb; fa.as = as; fa.bs = bs;
    fa.C = d->C; fa.N = d->N; fa.S = d->num_sweeps; fa.win = d->num_sweeps;
    fa.smooth3 = d->smooth3; fa.has_max = d->has_clamp_max; fa.cmax = d->clamp_max; fa.eps = d->eps;
    fa.t_first[0] = fa.t_first[1] = 0.f;
    bool seen[2] = {false, false};
    for (int s = 0; s < d->num_sweeps; ++s) {
        fa.sweep[s] = d->sweep[s];
        const int ax = d->sweep[s].axis;
        if (!seen[ax]) { seen[ax] = true; fa.t_first[ax] = d->sweep[s].t; }
    }
}

// device address of a pinned host buffer (hipHostMalloc memory is mapped into the device's address space); null when
// the buffer is not mapped: the caller then copies
float* mapped_host(float* host) {
    // Used on single-GPU processes' hosts only (one visible device), where it is measured; with several devices visible
    // the copy path is kept unless PDE_KMAX_MAPPED=1 (a kernel store into memory another device's driver state mapped
    // is not something this build could test).
    static const int allow = [] {
        const int e = env_int("PDE_KMAX_MAPPED", -1);
        if (e >= 0) return e;
        int n = 0;
        return (hipGetDeviceCount(&n) == hipSuccess && n == 1) ? 1 : 0;
    }();
    void* dp = nullptr;
    if (!allow || !host || hipHostGetDevicePointer(&dp, host, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return static_cast<float*>(dp);
}

int launch_factor(const PdeAdiDesc* d, const float* ab, const float* bb, const float* as, const float* bs,
                  float* coef, SweepTab* tab, int* varying, float* kmax, hipStream_t st, int window = 0,
                  float* kmax_mapped = nullptr, float* wide = nullptr) {
    FactorArgs fa;
    fill_factor_args(fa, d, ab, bb, as, bs);
    fa.coef = coef; fa.tab = tab; fa.varying = varying; fa.kmax = kmax; fa.wide = wide;
    if (window > 0) fa.win = window;
    const dim3 grid(d->num_sweeps * ((d->C + 1) / 2));
    switch (d->N) {
#define PDE_CASE(NN) case NN: hipLaunchKernelGGL(adi_factor_kernel<NN>, grid, dim3(128), 0, st, fa); break;
        PDE_CASE(8) PDE_CASE(12) PDE_CASE(16) PDE_CASE(20) PDE_CASE(24) PDE_CASE(28) PDE_CASE(32)
#undef PDE_CASE
        default: return PDE_E_UNSUPPORTED_N;
    }
    if (varying) hipLaunchKernelGGL(adi_flags_kernel, dim3(d->num_sweeps + d->C), dim3(64), 0, st, varying, kmax, kmax_mapped, d->num_sweeps, d->C);
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

// The per-sweep coefficient maxima leave for the host right behind the factor kernel, BEFORE the sweep
// launches: whoever plans checkpoints from them waits for microseconds, not for the layer's forward.
int publish_kmax(const float* kmax_dev, float* kmax_host, void* event, int n, hipStream_t st, bool written = false) {
    if (kmax_host && !written) {
        if (!kmax_dev) return PDE_E_BADARG;
        if (hipMemcpyAsync(kmax_host, kmax_dev, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess)
            return PDE_E_LAUNCH;
    }
    if (event && hipEventRecord(static_cast<hipEvent_t>(event), st) != hipSuccess) return PDE_E_LAUNCH;
    return PDE_OK;
}

// Strang schedules as the reference builds them (mnist_test.py:55-63): the second x sweep of a step and the first of
// the next are at the same time with the same increment, i.e. have identical coefficient records
bool strang_pairs_identical(const PdeAdiDesc* d) {
    if (d->num_sweeps % 3) return false;
    for (int s = 2; s + 1 < d->num_sweeps; s += 3) {
        const PdeSweep &p = d->sweep[s], &q = d->sweep[s + 1];
        if (p.axis != PDE_AXIS_X || q.axis != PDE_AXIS_X || p.t != q.t || p.delta != q.delta || p.h2 != q.h2) return false;
    }
    return true;
}

// which compile-time step pattern the schedule follows
int split_of(const PdeAdiDesc* d) {
    const int S = d->num_sweeps;
    bool strang = (S % 3) == 0, lie = (S % 2) == 0;
    for (int s = 0; s < S; ++s) {
        const int ax = d->sweep[s].axis;
        if (ax != ((s % 3) == 1 ? PDE_AXIS_Y : PDE_AXIS_X)) strang = false;
        if (ax != ((s % 2) == 1 ? PDE_AXIS_Y : PDE_AXIS_X)) lie = false;
    }
    return strang ? kSplitStrang : (lie ? kSplitLie : kSplitAny);
}

template <typename F>
int timed_launch(F&& f, hipStream_t st, bool is_fwd, bool timed) {
    Timing& tm = timing();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool rec = tm.on && timed;
    if (rec) {
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, st);
    }
    const int rc = f();
    if (rec) {
        (void)hipEventRecord(e1, st);                     // resolved later, in pde_timing_read
        std::lock_guard<std::mutex> lk(launch_mutex());
        pending().push_back({e0, e1, is_fwd});
    }
    return rc;
}

#define PDE_N_LIST PDE_CASE(8) PDE_CASE(12) PDE_CASE(16) PDE_CASE(20) PDE_CASE(24) PDE_CASE(28) PDE_CASE(32)

int dispatch_fwd(const PdeAdiDesc* d, int split, const SweepArgs& sa, int grid, size_t lds, hipStream_t st,
                 bool timed = true) {
    return timed_launch([&]() -> int {
        switch (d->N) {
#define PDE_CASE(NN) case NN: return adi_launch_fwd_##NN(d->io_dtype, split, &sa, grid, lds, st);
            PDE_N_LIST
#undef PDE_CASE
        }
        return PDE_E_UNSUPPORTED_N;
    }, st, true, timed);
}

int dispatch_bwd(const PdeAdiDesc* d, int split, const SweepArgs& sa, int grid, hipStream_t st) {
    return timed_launch([&]() -> int {
        switch (d->N) {
#define PDE_CASE(NN) case NN: return adi_launch_bwd_##NN(d->io_dtype, split, &sa, grid, st);
            PDE_N_LIST
#undef PDE_CASE
        }
        return PDE_E_UNSUPPORTED_N;
    }, st, false, true);
}

int count_ckpt(const uint64_t m[2]) { return m ? __builtin_popcountll(m[0]) + __builtin_popcountll(m[1]) : 0; }

bool asm_fwd_eligible(const PdeAdiDesc* d) {
    return asm_fwd_enabled() && d->N == 32 && d->io_dtype == PDE_IO_F32 && split_of(d) == kSplitStrang && d->num_sweeps >= 6;
}

// ---- launch helpers shared by the whole-schedule entry points and the per-step ones -----------------
// forward sweeps of `d` (a whole schedule or one step of it) with records/table already in place
int launch_fwd_sweeps(const PdeAdiDesc* d, const void* u, void* y, const float* coef, const SweepTab* tab,
                      hipStream_t st) {
    SweepArgs sa{};
    sa.in0 = u; sa.out = y; sa.coef = coef; sa.tab = tab;
    sa.B = d->B; sa.C = d->C; sa.S = d->num_sweeps;
    sa.G = groups_per_channel(d, kWaves * kJFwd, 16 / kWaves);
    sa.one_eps = 1.0f + d->eps;
    sa.xcd_map = use_xcd_map(d);
    sa.pair_x = (split_of(d) == kSplitStrang && strang_pairs_identical(d)) ? 1 : 0;
    // N = 32, fp32 tensors, Strang schedules of two or more steps: the hand-scheduled assembly kernel (gen_adi_fwd_asm.py:
    // 16 waves, four planes per lane, rows held in registers, the record ring handed over by counters)
    if (asm_fwd_eligible(d)) {
        AsmBwdArgs aa{};
        aa.gy = u; aa.y = y; aa.coef = coef;
        aa.B = d->B; aa.C = d->C; aa.S = d->num_sweeps;
        aa.nchunk = (d->B + kAsmFwdPlanes - 1) / kAsmFwdPlanes;
        aa.G = groups_per_channel(d, kAsmFwdPlanes, 1);
        aa.K = d->num_sweeps / 3;
        aa.acc_part = sa.pair_x ? 2 : 0;
        const int xcd = sa.xcd_map;
        return timed_launch([&]() -> int { return asm_fwd_launch(aa, xcd, st); }, st, true, true);
    }
    const size_t lds = (size_t)(kRing * kRecFwdPad + kWaves * kImage) * sizeof(float);
    return dispatch_fwd(d, split_of(d), sa, sa.G * d->C, lds, st);
}

struct AxisWeights { float wgt[2], tfirst[2]; bool have[2]; };
int axis_weights(const PdeAdiDesc* d, AxisWeights& w) {
    w.wgt[0] = w.wgt[1] = w.tfirst[0] = w.tfirst[1] = 0.f;
    w.have[0] = w.have[1] = false;
    for (int s = 0; s < d->num_sweeps; ++s) {
        const int ax = d->sweep[s].axis;
        const float v = d->sweep[s].delta / d->sweep[s].h2;
        if (w.have[ax] && v != w.wgt[ax]) return PDE_E_BADARG;   // one weight per axis (true for every reference variant)
        if (!w.have[ax]) w.tfirst[ax] = d->sweep[s].t;
        w.wgt[ax] = v; w.have[ax] = true;
    }
    return PDE_OK;
}

// checkpoint mask -> (count, number of forward sweeps to recompute); PDE_E_BADARG when inconsistent
int ckpt_plan(const PdeAdiDesc* d, const uint64_t ckpt_mask[2], const void* u, int& nck, int& Sf) {
    nck = count_ckpt(ckpt_mask);
    Sf = 0;
    if (nck) {
        if (!u) return PDE_E_BADARG;
        for (int s = 0; s < d->num_sweeps; ++s)
            if ((ckpt_mask[s >> 6] >> (s & 63)) & 1ull) Sf = s + 1;
        for (int s = d->num_sweeps; s < 128; ++s)
            if ((ckpt_mask[s >> 6] >> (s & 63)) & 1ull) return PDE_E_BADARG;    // bit beyond the schedule
        if (Sf >= d->num_sweeps) return PDE_E_BADARG;      // the last state is y itself
    }
    return PDE_OK;
}

// the hand-scheduled assembly kernel serves N = 32, fp32 tensors, Strang schedules of two or more steps without checkpoints
// (one-step launches — the layers with a channel operator — stay with the HIP kernel, whose records are resident)
bool asm_bwd_eligible(const PdeAdiDesc* d, int nck) {
    return asm_bwd_waves() && d->N == 32 && d->io_dtype == PDE_IO_F32 && split_of(d) == kSplitStrang && !nck &&
           d->num_sweeps >= 6;
}

// backward sweeps of `d`: optional checkpoint pre-pass, then the adjoint launch.  Partial gradient sums
// go to `part` ([G][C][4][image]); with `accumulate` they are ADDED to what is there (per-step launches
// of one layer call: every workgroup owns its slots, so this is race-free and order-independent).
int launch_bwd_sweeps(const PdeAdiDesc* d, const void* gy, const void* y, const void* u, const uint64_t ckpt_mask[2],
                      int nck, int Sf, void* gu, const float* coef, const SweepTab* tab, const int* varying,
                      float* part, void* dbg, float* ckpt, int G, int accumulate, hipStream_t st) {
    SweepArgs sa{};
    sa.in0 = gy; sa.in1 = y; sa.in2 = u; sa.out = gu; sa.coef = coef; sa.part = part; sa.tab = tab;
    sa.varying = varying; sa.ckpt = ckpt;
    sa.ck[0] = nck ? ckpt_mask[0] : 0ull; sa.ck[1] = nck ? ckpt_mask[1] : 0ull;
    sa.Sf = Sf; sa.smooth3 = d->smooth3;
    sa.xcd_map = use_xcd_map(d);
    sa.B = d->B; sa.C = d->C; sa.S = d->num_sweeps; sa.G = G;
    sa.one_eps = 1.0f + d->eps;
    sa.gu_scale = (float)pow(1.0 + (double)d->eps, -(double)d->num_sweeps);
    sa.acc_part = accumulate;
    sa.pair_x = (split_of(d) == kSplitStrang && strang_pairs_identical(d)) ? 1 : 0;
    sa.dbg = dbg;
    int rc = PDE_OK;
    if (nck) {
        // pre-pass: run the forward from u up to the last checkpointed sweep and park those states
        SweepArgs fa = sa;
        fa.in0 = u; fa.in1 = nullptr; fa.out = nullptr; fa.part = nullptr;
        fa.S = Sf;
        fa.G = groups_per_channel(d, kWaves * kJFwd, 16 / kWaves);
        const size_t lds_f = (size_t)(kRing * kRecFwdPad + kWaves * kImage) * sizeof(float);
        // the pre-pass stops after sweep Sf-1, which need not be a step boundary: look the axes up
        rc = dispatch_fwd(d, kSplitAny, fa, fa.G * d->C, lds_f, st, false);
        if (rc != PDE_OK) return rc;
    }
    // N = 32, fp32 tensors, Strang steps, no checkpoints: the fast body runs as the hand-scheduled assembly kernel
    // (gen_adi_bwd_asm.py: 168 VGPRs, three waves per SIMD), the masked body as a launch of its own over the same groups
    const int split = split_of(d);
    const int nw = asm_bwd_waves();
    if (asm_bwd_eligible(d, nck)) {
        AsmBwdArgs aa{};
        aa.gy = gy; aa.y = y; aa.gu = gu; aa.coef = coef; aa.part = part; aa.tab = tab; aa.varying = varying;
        aa.B = d->B; aa.C = d->C; aa.S = d->num_sweeps; aa.G = G;
        aa.gu_scale = sa.gu_scale; aa.acc_part = (accumulate ? 1 : 0) | (sa.pair_x ? 2 : 0);
        aa.K = d->num_sweeps / 3;
        aa.nchunk = (d->B + asm_bwd_planes(nw) - 1) / asm_bwd_planes(nw);
        sa.only_masked = 1;
        // (the timing recorder brackets the assembly kernel alone: it is the launch bench.py's roofline line is about; the
        //  masked body behind it leaves at once on every channel without a moving mask)
        const int rc2 = timed_launch([&]() -> int { return asm_bwd_launch(nw, aa, sa.xcd_map, st); }, st, false, true);
        if (rc2 != PDE_OK) return rc2;
        if (env_int("PDE_ASM_NO_MASKED", 0)) return PDE_OK;              // diagnostics only
        return adi_launch_bwd_32(d->io_dtype, split, &sa, G * d->C, st);
    }
    // one launch, two halves of the grid: fast variant | masked variant; a workgroup leaves at once unless
    // its channel belongs to its variant (decided on the device by the factor kernel, no host round trip)
    return dispatch_bwd(d, split, sa, 2 * G * d->C, st);
}

// parameter gradients from the partial sums; `d` is the WHOLE schedule the sums were taken over
int launch_pgrad(const PdeAdiDesc* d, const AxisWeights& w, const float* alpha_base, const float* beta_base,
                 const float* alpha_slope, const float* beta_slope, float* g_alpha_base, float* g_beta_base,
                 float* g_alpha_slope, float* g_beta_slope, const int* varying, const float* part, int G,
                 hipStream_t st, const float* gm_part = nullptr, float* gM = nullptr, float* g_skip = nullptr,
                 const float* skip_w = nullptr, float* g_w = nullptr) {
    PgradArgs pa{};
    pa.gm_part = gm_part; pa.gM = gM; pa.g_skip = g_skip; pa.skip_w = skip_w; pa.g_w = g_w; pa.gm_blocks = G;
    pa.part = part; pa.ab = alpha_base; pa.bb = beta_base; pa.as = alpha_slope; pa.bs = beta_slope;
    pa.g_ab = g_alpha_base; pa.g_bb = g_beta_base; pa.g_as = g_alpha_slope; pa.g_bs = g_beta_slope;
    pa.varying = varying;
    pa.C = d->C; pa.N = d->N; pa.S = d->num_sweeps; pa.G = G;
    pa.smooth3 = d->smooth3; pa.has_max = d->has_clamp_max; pa.accumulate = 0;
    pa.cmax = d->clamp_max; pa.eps = d->eps;
    // the kernel's sums carry one factor (1+eps) (see pde_common.h, INVB)
    pa.wx = w.wgt[0] / (1.0f + d->eps); pa.wy = w.wgt[1] / (1.0f + d->eps);
    pa.t_first[0] = w.tfirst[0]; pa.t_first[1] = w.tfirst[1];
    pa.have_axis[0] = w.have[0]; pa.have_axis[1] = w.have[1];
    hipLaunchKernelGGL(adi_pgrad_kernel, dim3(4 * d->C + (gm_part ? 1 : 0)), dim3(1024), 0, st, pa);
    return check_launch();
}

// one step (sweeps k*sps .. k*sps+sps-1) of a whole schedule as a launch descriptor of its own
int step_desc(const PdeAdiDesc* d, int sps, int k, PdeAdiDesc& ds) {
    const int rc = check_desc(d);
    if (rc != PDE_OK) return rc;
    if (sps <= 0 || d->num_sweeps % sps != 0 || k < 0 || k >= d->num_sweeps / sps) return PDE_E_BADARG;
    ds = *d;
    ds.num_sweeps = sps;
    for (int s = 0; s < sps; ++s) ds.sweep[s] = d->sweep[k * sps + s];
    return PDE_OK;
}
// layout of the whole-schedule ("steps") workspace: records of all sweeps | channel flags | one table per step
size_t steps_tab_offset(const PdeAdiDesc* d) { return coef_bytes(d) + flag_bytes(d); }
// ... | lane-major records (only where the one-launch C = 32 / 64 path applies)
size_t steps_wide_offset(const PdeAdiDesc* d, int sps) {
    return steps_tab_offset(d) + align_up((size_t)(d->num_sweeps / sps) * sizeof(SweepTab), 256);
}
// which compile-time step pattern every step of the schedule follows (kSplitAny: not all the same)
int small_split(const PdeAdiDesc* d, int sps) {
    PdeAdiDesc ds;
    if (step_desc(d, sps, 0, ds) != PDE_OK) return kSplitAny;
    const int sp = split_of(&ds);
    for (int k = 1; k < d->num_sweeps / sps; ++k) {          // every step the same pattern
        if (step_desc(d, sps, k, ds) != PDE_OK || split_of(&ds) != sp) return kSplitAny;
    }
    return sp;
}

// ---- C = 32 / 64 fp32: the whole forward in one launch (pde_adi_wide.h) ----
bool wide_enabled() {
    static const bool on = [] { const char* e = getenv("PDE_WIDE"); return !(e && e[0] == '0'); }();
    return on;
}
bool wide_supported(const PdeAdiDesc* d, int sps) {
    if (!wide_enabled() || d->io_dtype != PDE_IO_F32 || (d->C != 32 && d->C != 64)) return false;
    bool n_ok = false;
#define PDE_WIDE_CASE(NN) n_ok |= (d->N == NN);
    PDE_WIDE_N_LIST
#undef PDE_WIDE_CASE
    if (!n_ok || (sps != 2 && sps != 3) || d->num_sweeps % sps) return false;
    const int sp = small_split(d, sps);
    return (sp == kSplitStrang && sps == 3) || (sp == kSplitLie && sps == 2);
}

}  // namespace
}  // namespace pde

using namespace pde;

extern "C" {

int pde_adi_line_length_path(int32_t N) { return fused_n(N) ? 1 : (gen_n_ok(N) ? 2 : 0); }

int pde_adi_backward_kernel(const PdeAdiDesc* d, int32_t num_checkpoints) {
    if (check_desc(d, true) != PDE_OK) return PDE_E_BADARG;
    if (!fused_n(d->N)) return 2;
    return asm_bwd_eligible(d, num_checkpoints) ? 1 : 0;
}

size_t pde_adi_forward_workspace_bytes(const PdeAdiDesc* d) {
    if (check_desc(d, true) != PDE_OK) return 0;
    if (!fused_n(d->N)) return gen_forward_workspace_bytes(d);
    return coef_bytes(d) + tab_bytes() + flag_bytes(d);
}

size_t pde_adi_backward_workspace_bytes(const PdeAdiDesc* d, int32_t num_checkpoints) {
    if (check_desc(d, true) != PDE_OK || num_checkpoints < 0) return 0;
    if (!fused_n(d->N)) return gen_backward_workspace_bytes(d, num_checkpoints);
    const int G = groups_per_channel(d, kWaves * kJBwd, 8 / kWaves);
    size_t b = coef_bytes(d) + tab_bytes() + flag_bytes(d);
    b += align_up((size_t)G * d->C * 4 * kImage * sizeof(float), 256) + 2048;    // + diagnostics scratch
    b += align_up((size_t)num_checkpoints * d->B * d->C * d->N * d->N * sizeof(float), 256);
    return b;
}

int pde_adi_forward(const PdeAdiDesc* d, const void* u, void* y, const float* alpha_base, const float* beta_base,
                    const float* alpha_slope, const float* beta_slope, float* kappa_max, float* kappa_max_host,
                    void* kappa_event, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_desc(d, true);
    if (rc != PDE_OK) return rc;
    if (!u || !y || !alpha_base || !beta_base || !alpha_slope || !beta_slope || !workspace) return PDE_E_BADARG;
    if (workspace_bytes < pde_adi_forward_workspace_bytes(d) || ((uintptr_t)workspace & 15)) return PDE_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!fused_n(d->N)) {                                 // any other line length: one thread per line (pde_adi_gen.hip)
        if (kappa_max_host && !kappa_max) return PDE_E_BADARG;
        rc = gen_factor(d, alpha_base, beta_base, alpha_slope, beta_slope, kappa_max, workspace, st);
        if (rc != PDE_OK) return rc;
        rc = publish_kmax(kappa_max, kappa_max_host, kappa_event, d->num_sweeps, st);
        if (rc != PDE_OK) return rc;
        return gen_forward_sweeps(d, u, y, workspace, st);
    }
    float* coef = static_cast<float*>(workspace);
    SweepTab* tab = reinterpret_cast<SweepTab*>(static_cast<char*>(workspace) + coef_bytes(d));
    int* varying = reinterpret_cast<int*>(static_cast<char*>(workspace) + coef_bytes(d) + tab_bytes());
    float* km = (kappa_max && kappa_max_host) ? mapped_host(kappa_max_host) : nullptr;
    rc = launch_factor(d, alpha_base, beta_base, alpha_slope, beta_slope, coef, tab, varying, kappa_max, st, 0, km);
    if (rc != PDE_OK) return rc;
    rc = publish_kmax(kappa_max, kappa_max_host, kappa_event, d->num_sweeps, st, km != nullptr);
    if (rc != PDE_OK) return rc;
    return launch_fwd_sweeps(d, u, y, coef, tab, st);
}

int pde_adi_backward(const PdeAdiDesc* d, const void* gy, const void* y, const void* u, const uint64_t ckpt_mask[2],
                     void* gu, const float* alpha_base, const float* beta_base, const float* alpha_slope,
                     const float* beta_slope, float* g_alpha_base, float* g_beta_base, float* g_alpha_slope,
                     float* g_beta_slope, const void* fwd_workspace, void* workspace, size_t workspace_bytes,
                     void* stream) {
    int rc = check_desc(d, true);
    if (rc != PDE_OK) return rc;
    if (!gy || !y || !gu || !alpha_base || !beta_base || !alpha_slope || !beta_slope || !g_alpha_base ||
        !g_beta_base || !g_alpha_slope || !g_beta_slope || !workspace)
        return PDE_E_BADARG;
    int nck, Sf;
    rc = ckpt_plan(d, ckpt_mask, u, nck, Sf);
    if (rc != PDE_OK) return rc;
    if (workspace_bytes < pde_adi_backward_workspace_bytes(d, nck) || ((uintptr_t)workspace & 15)) return PDE_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!fused_n(d->N))
        return gen_backward(d, gy, y, u, ckpt_mask, nck, Sf, gu, alpha_base, beta_base, alpha_slope, beta_slope, g_alpha_base,
                            g_beta_base, g_alpha_slope, g_beta_slope, fwd_workspace, workspace, st);
    const int G = groups_per_channel(d, kWaves * kJBwd, 8 / kWaves);
    char* ws = static_cast<char*>(workspace);
    float* coef = reinterpret_cast<float*>(ws);           ws += coef_bytes(d);
    SweepTab* tab = reinterpret_cast<SweepTab*>(ws);      ws += tab_bytes();
    int* varying = reinterpret_cast<int*>(ws);            ws += flag_bytes(d);
    float* part = reinterpret_cast<float*>(ws);           ws += align_up((size_t)G * d->C * 4 * kImage * sizeof(float), 256);
    void* dbg = ws;                                       ws += 2048;
    float* ckpt = reinterpret_cast<float*>(ws);
    if (fwd_workspace) {
        // the forward call of the same step left the factorisation, the sweep table and the
        // channel flags in its workspace: read them from there instead of recomputing
        const char* fw = static_cast<const char*>(fwd_workspace);
        coef = reinterpret_cast<float*>(const_cast<char*>(fw));
        tab = reinterpret_cast<SweepTab*>(const_cast<char*>(fw + coef_bytes(d)));
        varying = reinterpret_cast<int*>(const_cast<char*>(fw + coef_bytes(d) + tab_bytes()));
    } else {
        rc = launch_factor(d, alpha_base, beta_base, alpha_slope, beta_slope, coef, tab, varying, nullptr, st);
        if (rc != PDE_OK) return rc;
    }
    AxisWeights w;
    rc = axis_weights(d, w);
    if (rc != PDE_OK) return rc;
    rc = launch_bwd_sweeps(d, gy, y, u, ckpt_mask, nck, Sf, gu, coef, tab, varying, part, dbg, ckpt, G, 0, st);
    if (rc != PDE_OK) return rc;
    return launch_pgrad(d, w, alpha_base, beta_base, alpha_slope, beta_slope, g_alpha_base, g_beta_base, g_alpha_slope,
                        g_beta_slope, varying, part, G, st);
}

// ---- one layer call as a sequence of per-step launches (layers with a channel operator between the steps) ----
static size_t state_bytes(const PdeAdiDesc* d);
size_t pde_adi_steps_workspace_bytes(const PdeAdiDesc* d, int32_t sweeps_per_step) {
    if (check_desc(d) != PDE_OK || sweeps_per_step <= 0 || d->num_sweeps % sweeps_per_step) return 0;
    // ... | lane-major records (C = 32 / 64 one-launch forward)  or  this layer's term of a shared-input group's output (C <= 4)
    return steps_wide_offset(d, sweeps_per_step) +
           (wide_supported(d, sweeps_per_step) ? align_up((size_t)d->num_sweeps * d->C * kWideRec * sizeof(float), 256) : 0) +
           (pde_adi_small_supported(d, sweeps_per_step) ? align_up(state_bytes(d), 256) : 0);
}

// zero + factorise every sweep into a steps workspace; the per-sweep maxima optionally also straight into the caller's
// pinned host buffer (returns through `wrote_host` whether that happened)
static int factor_steps_impl(const PdeAdiDesc* d, int32_t sweeps_per_step, const float* alpha_base, const float* beta_base,
                             const float* alpha_slope, const float* beta_slope, float* kappa_max, float* kappa_max_host,
                             bool* wrote_host, void* steps_workspace, size_t workspace_bytes, void* stream) {
    if (wrote_host) *wrote_host = false;
    int rc = check_desc(d);
    if (rc != PDE_OK) return rc;
    if (!alpha_base || !beta_base || !alpha_slope || !beta_slope || !steps_workspace) return PDE_E_BADARG;
    const size_t need = pde_adi_steps_workspace_bytes(d, sweeps_per_step);
    if (need == 0) return PDE_E_BADARG;
    if (workspace_bytes < need || ((uintptr_t)steps_workspace & 15)) return PDE_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(steps_workspace);
    float* coef = reinterpret_cast<float*>(ws);
    int* varying = reinterpret_cast<int*>(ws + coef_bytes(d));
    SweepTab* tabs = reinterpret_cast<SweepTab*>(ws + steps_tab_offset(d));
    float* km = (kappa_max && kappa_max_host) ? mapped_host(kappa_max_host) : nullptr;
    if (wrote_host) *wrote_host = km != nullptr;
    float* wide = wide_supported(d, sweeps_per_step) ? reinterpret_cast<float*>(ws + steps_wide_offset(d, sweeps_per_step)) : nullptr;
    return launch_factor(d, alpha_base, beta_base, alpha_slope, beta_slope, coef, tabs, varying, kappa_max, st,
                         sweeps_per_step, km, wide);
}

int pde_adi_factor_steps(const PdeAdiDesc* d, int32_t sweeps_per_step, const float* alpha_base, const float* beta_base,
                         const float* alpha_slope, const float* beta_slope, float* kappa_max, void* steps_workspace,
                         size_t workspace_bytes, void* stream) {
    return factor_steps_impl(d, sweeps_per_step, alpha_base, beta_base, alpha_slope, beta_slope, kappa_max, nullptr, nullptr,
                             steps_workspace, workspace_bytes, stream);
}

int pde_adi_forward_step(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t step, const void* u, void* y,
                         const void* steps_workspace, void* stream) {
    PdeAdiDesc ds;
    int rc = step_desc(d, sweeps_per_step, step, ds);
    if (rc != PDE_OK) return rc;
    if (!u || !y || !steps_workspace) return PDE_E_BADARG;
    const char* ws = static_cast<const char*>(steps_workspace);
    const float* coef = reinterpret_cast<const float*>(ws) + (size_t)step * sweeps_per_step * d->C * kRecStride;
    const SweepTab* tab = reinterpret_cast<const SweepTab*>(ws + steps_tab_offset(d)) + step;
    return launch_fwd_sweeps(&ds, u, y, coef, tab, static_cast<hipStream_t>(stream));
}

size_t pde_adi_backward_step_workspace_bytes(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t num_checkpoints) {
    PdeAdiDesc ds;
    if (step_desc(d, sweeps_per_step, 0, ds) != PDE_OK || num_checkpoints < 0) return 0;
    const int G = groups_per_channel(&ds, kWaves * kJBwd, 8 / kWaves);
    return align_up((size_t)G * d->C * 4 * kImage * sizeof(float), 256) + 2048 +
           align_up((size_t)num_checkpoints * d->B * d->C * d->N * d->N * sizeof(float), 256);
}

int pde_adi_backward_step(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t step, const void* gy, const void* y,
                          const void* u, const uint64_t ckpt_mask[2], void* gu, const void* steps_workspace,
                          void* workspace, size_t workspace_bytes, int32_t accumulate, void* stream) {
    PdeAdiDesc ds;
    int rc = step_desc(d, sweeps_per_step, step, ds);
    if (rc != PDE_OK) return rc;
    if (!gy || !y || !gu || !steps_workspace || !workspace) return PDE_E_BADARG;
    int nck, Sf;
    rc = ckpt_plan(&ds, ckpt_mask, u, nck, Sf);
    if (rc != PDE_OK) return rc;
    if (workspace_bytes < pde_adi_backward_step_workspace_bytes(d, sweeps_per_step, nck) || ((uintptr_t)workspace & 15))
        return PDE_E_WORKSPACE;
    const int G = groups_per_channel(&ds, kWaves * kJBwd, 8 / kWaves);
    const char* fw = static_cast<const char*>(steps_workspace);
    const float* coef = reinterpret_cast<const float*>(fw) + (size_t)step * sweeps_per_step * d->C * kRecStride;
    const int* varying = reinterpret_cast<const int*>(fw + coef_bytes(d));
    const SweepTab* tab = reinterpret_cast<const SweepTab*>(fw + steps_tab_offset(d)) + step;
    char* ws = static_cast<char*>(workspace);
    float* part = reinterpret_cast<float*>(ws);           ws += align_up((size_t)G * d->C * 4 * kImage * sizeof(float), 256);
    void* dbg = ws;                                       ws += 2048;
    float* ckpt = reinterpret_cast<float*>(ws);
    return launch_bwd_sweeps(&ds, gy, y, u, ckpt_mask, nck, Sf, gu, coef, tab, varying, part, dbg, ckpt, G,
                             accumulate != 0, static_cast<hipStream_t>(stream));
}

int pde_adi_param_grads(const PdeAdiDesc* d, int32_t sweeps_per_step, const float* alpha_base, const float* beta_base,
                        const float* alpha_slope, const float* beta_slope, float* g_alpha_base, float* g_beta_base,
                        float* g_alpha_slope, float* g_beta_slope, const void* steps_workspace, const void* workspace,
                        void* stream) {
    PdeAdiDesc ds;
    int rc = step_desc(d, sweeps_per_step, 0, ds);
    if (rc != PDE_OK) return rc;
    if (!alpha_base || !beta_base || !alpha_slope || !beta_slope || !g_alpha_base || !g_beta_base || !g_alpha_slope ||
        !g_beta_slope || !steps_workspace || !workspace)
        return PDE_E_BADARG;
    AxisWeights w;
    rc = axis_weights(d, w);                               // over the whole schedule
    if (rc != PDE_OK) return rc;
    const int G = groups_per_channel(&ds, kWaves * kJBwd, 8 / kWaves);
    const int* varying = reinterpret_cast<const int*>(static_cast<const char*>(steps_workspace) + coef_bytes(d));
    return launch_pgrad(d, w, alpha_base, beta_base, alpha_slope, beta_slope, g_alpha_base, g_beta_base, g_alpha_slope,
                        g_beta_slope, varying, static_cast<const float*>(workspace), G, static_cast<hipStream_t>(stream));
}

// ---- the whole layer call with a channel operator between the steps, looped on the host in C++ ----------
// (one call from the binding instead of two per step: at the reference's own sizes, C = 3 and batch 128, the
// layer is bound by the host's launch rate, and every Python -> ctypes transition costs as much as a launch)
static size_t state_bytes(const PdeAdiDesc* d) {
    return (size_t)d->B * d->C * d->N * d->N * (d->io_dtype == PDE_IO_BF16 ? 2 : 4);
}

static int wide_forward(const PdeAdiDesc* d, int sps, int mode, const void* u, void* states, void* y, const float* M,
                        const void* steps_workspace, int keep, hipStream_t st) {
    WideArgs wa{};
    wa.u = u; wa.states = states; wa.M = M; wa.last = y;
    wa.coef = reinterpret_cast<const float*>(static_cast<const char*>(steps_workspace) + steps_wide_offset(d, sps));
    wa.B = d->B; wa.K = d->num_sweeps / sps; wa.mode = mode; wa.keep = keep;
    const int grid = d->B < 2048 ? d->B : 2048;
    switch (d->N) {
#define PDE_WIDE_CASE(NN) case NN: return adi_launch_wide_fwd_##NN(d->C, small_split(d, sps), &wa, grid, st);
        PDE_WIDE_N_LIST
#undef PDE_WIDE_CASE
    }
    return PDE_E_UNSUPPORTED_N;
}

int pde_adi_mixed_one_launch(const PdeAdiDesc* d, int32_t sweeps_per_step) {
    return (check_desc(d) == PDE_OK && wide_supported(d, sweeps_per_step)) ? 1 : 0;
}

int pde_adi_mixed_forward(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t mode, const void* u, void* states, void* y,
                          const float* M, const float* alpha_base, const float* beta_base, const float* alpha_slope,
                          const float* beta_slope, float* kappa_max, float* kappa_max_host, void* kappa_event,
                          void* steps_workspace, size_t workspace_bytes, void* stream) {
    if (!u || !states || !M || (mode != 1 && mode != 2)) return PDE_E_BADARG;
    bool wrote = false;
    int rc = factor_steps_impl(d, sweeps_per_step, alpha_base, beta_base, alpha_slope, beta_slope, kappa_max, kappa_max_host,
                               &wrote, steps_workspace, workspace_bytes, stream);
    if (rc != PDE_OK) return rc;
    rc = publish_kmax(kappa_max, kappa_max_host, kappa_event, d->num_sweeps, static_cast<hipStream_t>(stream), wrote);
    if (rc != PDE_OK) return rc;
    const int K = d->num_sweeps / sweeps_per_step, HW = d->N * d->N;
    if (wide_supported(d, sweeps_per_step))
        return wide_forward(d, sweeps_per_step, mode, u, states, y, M, steps_workspace, 1, static_cast<hipStream_t>(stream));
    const size_t sb = state_bytes(d);
    char* st = static_cast<char*>(states);
    const void* cur = u;
    for (int k = 0; k < K; ++k) {
        void* a_k = st + (size_t)(2 * k) * sb;
        void* b_k = (k == K - 1 && y) ? y : st + (size_t)(2 * k + 1) * sb;      // the layer output may live outside `states`
        if (mode == 1) {                                   // cifar10.py:91: u <- M u, then the step's sweeps
            rc = pde_channel_mix_forward(d->B, d->C, HW, d->io_dtype, cur, M, a_k, stream);
            if (rc == PDE_OK) rc = pde_adi_forward_step(d, sweeps_per_step, k, a_k, b_k, steps_workspace, stream);
        } else {                                           // SVHN.py:60-71: the sweeps, then u <- K u
            rc = pde_adi_forward_step(d, sweeps_per_step, k, cur, a_k, steps_workspace, stream);
            if (rc == PDE_OK) rc = pde_channel_mix_forward(d->B, d->C, HW, d->io_dtype, a_k, M, b_k, stream);
        }
        if (rc != PDE_OK) return rc;
        cur = b_k;
    }
    return PDE_OK;
}

size_t pde_adi_mixed_backward_workspace_bytes(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t num_checkpoints) {
    const size_t a = pde_adi_backward_step_workspace_bytes(d, sweeps_per_step, num_checkpoints);
    if (a == 0) return 0;
    return align_up(a, 256) + align_up(pde_channel_mix_backward_workspace_bytes(d->B, d->C, d->N * d->N), 256) +
           2 * align_up(state_bytes(d), 256);
}

int pde_adi_mixed_backward(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t mode, const void* gy, const void* u,
                           const void* states, const void* y, const float* M, const uint64_t ckpt_mask[2], void* gu,
                           const float* alpha_base, const float* beta_base, const float* alpha_slope,
                           const float* beta_slope, float* g_alpha_base, float* g_beta_base, float* g_alpha_slope,
                           float* g_beta_slope, float* gM, const void* steps_workspace, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (!gy || !u || !states || !M || !gu || !gM || !workspace || (mode != 1 && mode != 2)) return PDE_E_BADARG;
    const int nck = count_ckpt(ckpt_mask);
    const size_t need = pde_adi_mixed_backward_workspace_bytes(d, sweeps_per_step, nck);
    if (need == 0) return PDE_E_BADARG;
    if (workspace_bytes < need || ((uintptr_t)workspace & 15)) return PDE_E_WORKSPACE;
    const int K = d->num_sweeps / sweeps_per_step, HW = d->N * d->N;
    const size_t sb = state_bytes(d);
    const size_t ws_a = align_up(pde_adi_backward_step_workspace_bytes(d, sweeps_per_step, nck), 256);
    const size_t ws_m = align_up(pde_channel_mix_backward_workspace_bytes(d->B, d->C, HW), 256);
    char* w = static_cast<char*>(workspace);
    void* aws = w;
    void* mws = w + ws_a;
    void* buf[2] = {w + ws_a + ws_m, w + ws_a + ws_m + align_up(sb, 256)};
    const char* st = static_cast<const char*>(states);
    const void* g_in = gy;                                 // gradient entering the step (never written)
    int rc = PDE_OK;
    if (nck && wide_supported(d, sweeps_per_step)) {
        // the one-launch forward kept the sweep outputs only; a checkpointed backward also reads the operator's outputs
        char* sw = const_cast<char*>(st);
        for (int k = 0; k < K && rc == PDE_OK; ++k) {
            if (mode == 1)
                rc = pde_channel_mix_forward(d->B, d->C, HW, d->io_dtype, k ? sw + (size_t)(2 * k - 1) * sb : u, M,
                                             sw + (size_t)(2 * k) * sb, stream);
            else if (k + 1 < K)
                rc = pde_channel_mix_forward(d->B, d->C, HW, d->io_dtype, sw + (size_t)(2 * k) * sb, M,
                                             sw + (size_t)(2 * k + 1) * sb, stream);
        }
        if (rc != PDE_OK) return rc;
    }
    for (int k = K - 1; k >= 0; --k) {
        const int first = (k == K - 1), last = (k == 0);
        const void* a_k = st + (size_t)(2 * k) * sb;
        const void* b_k = (first && y) ? y : st + (size_t)(2 * k + 1) * sb;
        const void* prev = (k == 0) ? u : st + (size_t)(2 * k - 1) * sb;
        void* g_mid = (g_in == buf[0]) ? buf[1] : buf[0];
        void* g_out = last ? gu : ((g_mid == buf[0]) ? buf[1] : buf[0]);
        if (mode == 1) {                                   // step = mix (prev -> a_k), sweeps (a_k -> b_k)
            rc = pde_adi_backward_step(d, sweeps_per_step, k, g_in, b_k, nck ? a_k : nullptr, ckpt_mask, g_mid,
                                       steps_workspace, aws, ws_a, first ? 0 : 1, stream);
            if (rc == PDE_OK)
                rc = pde_channel_mix_backward_steps(d->B, d->C, HW, d->io_dtype, prev, g_mid, M, g_out, gM, mws, ws_m,
                                                    first ? 0 : 1, last ? 1 : 0, stream);
        } else {                                           // step = sweeps (prev -> a_k), mix (a_k -> b_k)
            rc = pde_channel_mix_backward_steps(d->B, d->C, HW, d->io_dtype, a_k, g_in, M, g_mid, gM, mws, ws_m,
                                                first ? 0 : 1, last ? 1 : 0, stream);
            if (rc == PDE_OK)
                rc = pde_adi_backward_step(d, sweeps_per_step, k, g_mid, a_k, nck ? prev : nullptr, ckpt_mask, g_out,
                                           steps_workspace, aws, ws_a, first ? 0 : 1, stream);
        }
        if (rc != PDE_OK) return rc;
        g_in = g_out;
    }
    return pde_adi_param_grads(d, sweeps_per_step, alpha_base, beta_base, alpha_slope, beta_slope, g_alpha_base,
                               g_beta_base, g_alpha_slope, g_beta_slope, steps_workspace, aws, stream);
}

// ---- the whole layer in ONE launch per pass: C <= 4 channels with a channel operator between the steps ----------
// (pde_adi_small.h; the reference's own models: cifar10.py:253-258 C = 3 mixing before every step,
// SVHN.py:238 C = 3 coupling after every step + skip blend)
static int small_grid(const PdeAdiDesc* d) { return d->B < 1024 ? d->B : 1024; }
static int dispatch_small(bool fwd, const PdeAdiDesc* d, int split, const SmallArgs& sa, size_t lds, hipStream_t st) {
    const int grid = small_grid(d) * (sa.par ? sa.L : 1);
    switch (d->N) {
#define PDE_SMALL_CASE(NN) case NN: return fwd ? adi_launch_small_fwd_##NN(d->io_dtype, split, &sa, grid, lds, st) \
                                               : adi_launch_small_bwd_##NN(d->io_dtype, split, &sa, grid, lds, st);
        PDE_SMALL_N_LIST
#undef PDE_SMALL_CASE
    }
    return PDE_E_UNSUPPORTED_N;
}
static size_t small_lds_fwd(int C) { return (size_t)C * (2 * kRecFwdPad + kImage) * sizeof(float); }
static size_t small_lds_bwd(int C) { return (size_t)C * (2 * kSmallRecB + 2 * kImage) * sizeof(float); }

int pde_adi_small_supported(const PdeAdiDesc* d, int32_t sweeps_per_step) {
    if (check_desc(d) != PDE_OK || d->C > kSmallMaxC) return 0;
    if (sweeps_per_step != 2 && sweeps_per_step != 3) return 0;
    if (d->num_sweeps % sweeps_per_step) return 0;
    bool n_ok = false;
#define PDE_SMALL_CASE(NN) n_ok |= (d->N == NN);
    PDE_SMALL_N_LIST
#undef PDE_SMALL_CASE
    if (!n_ok) return 0;
    const int sp = small_split(d, sweeps_per_step);
    if (!((sp == kSplitStrang && sweeps_per_step == 3) || (sp == kSplitLie && sweeps_per_step == 2))) return 0;
    AxisWeights w;
    return axis_weights(d, w) == PDE_OK ? 1 : 0;
}

static void small_fill(SmallLayer& sl, const PdeAdiDesc* d, int sps, int mode, const void* steps_workspace, const float* M,
                       const float* skip_weight, float weight, const float* weight_ptr = nullptr) {
    const char* ws = static_cast<const char*>(steps_workspace);
    sl.coef = reinterpret_cast<const float*>(ws);
    sl.varying = reinterpret_cast<const int*>(ws + coef_bytes(d));
    sl.tabs = reinterpret_cast<const SweepTab*>(ws + steps_tab_offset(d));
    sl.M = M; sl.skip_w = skip_weight;
    sl.K = d->num_sweeps / sps; sl.mode = mode; sl.smooth3 = d->smooth3;
    sl.step_scale = (float)pow(1.0 + (double)d->eps, -(double)sps);
    sl.w = weight; sl.wp = weight_ptr;
}

// Layers that share an input run side by side (one layer per workgroup) while the batch alone does not fill the chip:
// the three cifar10 layers on 128 samples are 128 workgroups of 3 waves each walking 51 sweeps one after the other
static int multi_parallel(int32_t L, const PdeAdiDesc* d) { return (L > 1 && d->B < 1024) ? 1 : 0; }
static int combine_slabs(const PdeAdiDesc* d, const SmallArgs& sa, void* out, hipStream_t st) {
    CombineArgs ca{};
    for (int i = 0; i < sa.L; ++i) ca.slab[i] = sa.layer[i].slab;
    ca.out = out; ca.L = sa.L;
    ca.n4 = (size_t)d->B * d->C * d->N * d->N / 4;
    return d->io_dtype == PDE_IO_F32 ? small_combine_io<float>(ca, st) : small_combine_io<bf16_t>(ca, st);
}

// layers of one launch must agree on everything the kernel instantiation and the grid depend on
static int multi_check(int32_t L, const PdeSmallLayer* layers) {
    if (L < 1 || L > kSmallMaxL || !layers) return PDE_E_BADARG;
    const PdeAdiDesc* d0 = layers[0].desc;
    for (int i = 0; i < L; ++i) {
        const PdeSmallLayer& y = layers[i];
        if (!y.desc || !pde_adi_small_supported(y.desc, y.sweeps_per_step)) return PDE_E_BADARG;
        if (!y.M || (y.mode != 1 && y.mode != 2) || (y.skip_weight && y.mode != 2)) return PDE_E_BADARG;
        if (y.desc->B != d0->B || y.desc->C != d0->C || y.desc->N != d0->N || y.desc->io_dtype != d0->io_dtype ||
            y.sweeps_per_step != layers[0].sweeps_per_step)
            return PDE_E_BADARG;
        if (L > 1 && y.mode != 1) return PDE_E_BADARG;     // shared-input groups exist for the mixing-first layers only
        if (!y.alpha_base || !y.beta_base || !y.alpha_slope || !y.beta_slope || !y.steps_workspace) return PDE_E_BADARG;
    }
    return PDE_OK;
}

int pde_adi_multi_forward(int32_t num_layers, const PdeSmallLayer* layers, const void* u, void* out, void* kappa_event,
                          void* stream) {
    int rc = multi_check(num_layers, layers);
    if (rc != PDE_OK) return rc;
    if (!u || !out) return PDE_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const PdeAdiDesc* d0 = layers[0].desc;
    SmallArgs sa{};
    sa.u = u; sa.out = out; sa.B = d0->B; sa.C = d0->C; sa.L = num_layers;
    sa.par = multi_parallel(num_layers, d0);
    for (int i = 0; i < num_layers; ++i) {
        const PdeSmallLayer& y = layers[i];
        bool wrote = false;
        rc = factor_steps_impl(y.desc, y.sweeps_per_step, y.alpha_base, y.beta_base, y.alpha_slope, y.beta_slope,
                               y.kappa_max, y.kappa_max_host, &wrote, y.steps_workspace, y.steps_workspace_bytes, stream);
        if (rc != PDE_OK) return rc;
        rc = publish_kmax(y.kappa_max, y.kappa_max_host, nullptr, y.desc->num_sweeps, st, wrote);
        if (rc != PDE_OK) return rc;
        small_fill(sa.layer[i], y.desc, y.sweeps_per_step, y.mode, y.steps_workspace, y.M, y.skip_weight, y.weight, y.weight_ptr);
        sa.layer[i].states = y.states;
        sa.layer[i].psum = y.plane_sums;
        sa.layer[i].slab = static_cast<char*>(y.steps_workspace) + steps_wide_offset(y.desc, y.sweeps_per_step);
    }
    if (kappa_event && hipEventRecord(static_cast<hipEvent_t>(kappa_event), st) != hipSuccess) return PDE_E_LAUNCH;
    rc = dispatch_small(true, d0, small_split(d0, layers[0].sweeps_per_step), sa, small_lds_fwd(d0->C), st);
    if (rc != PDE_OK || !sa.par) return rc;
    return combine_slabs(d0, sa, out, st);
}

size_t pde_adi_small_backward_workspace_bytes(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t num_checkpoints) {
    if (!pde_adi_small_supported(d, sweeps_per_step) || num_checkpoints < 0) return 0;
    const int G = small_grid(d), K = d->num_sweeps / sweeps_per_step;
    return align_up((size_t)G * d->C * 4 * kImage * sizeof(float), 256) +
           align_up((size_t)G * d->C * kGmStride * sizeof(float), 256) +
           align_up(state_bytes(d), 256) +                 // this layer's term of a shared-input group's input gradient
           align_up((size_t)K * num_checkpoints * d->B * d->C * d->N * d->N * sizeof(float), 256);
}

int pde_adi_multi_backward(int32_t num_layers, const PdeSmallLayer* layers, const void* gy, const void* u, void* gu,
                           void* stream) {
    int rc = multi_check(num_layers, layers);
    if (rc != PDE_OK) return rc;
    if (!u || !gu) return PDE_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const PdeAdiDesc* d0 = layers[0].desc;
    const int sps = layers[0].sweeps_per_step, G = small_grid(d0), split = small_split(d0, sps);
    SmallArgs sa{};
    sa.u = u; sa.gy = gy; sa.out = gu; sa.B = d0->B; sa.C = d0->C; sa.L = num_layers;
    sa.par = multi_parallel(num_layers, d0);
    bool any_ck = false;
    for (int i = 0; i < num_layers; ++i) {
        const PdeSmallLayer& y = layers[i];
        if (!y.states || !y.gM || !y.workspace || !y.g_alpha_base || !y.g_beta_base || !y.g_alpha_slope ||
            !y.g_beta_slope || (y.skip_weight && !y.g_skip_weight))
            return PDE_E_BADARG;
        PdeAdiDesc ds;
        rc = step_desc(y.desc, sps, 0, ds);
        if (rc != PDE_OK) return rc;
        int nck, Sf;
        rc = ckpt_plan(&ds, y.ckpt_mask, u, nck, Sf);      // the mask is relative to a step
        if (rc != PDE_OK) return rc;
        if (y.workspace_bytes < pde_adi_small_backward_workspace_bytes(y.desc, sps, nck) || ((uintptr_t)y.workspace & 15))
            return PDE_E_WORKSPACE;
        SmallLayer& sl = sa.layer[i];
        small_fill(sl, y.desc, sps, y.mode, y.steps_workspace, y.M, y.skip_weight, y.weight, y.weight_ptr);
        char* ws = static_cast<char*>(y.workspace);
        sl.part = reinterpret_cast<float*>(ws);            ws += align_up((size_t)G * d0->C * 4 * kImage * sizeof(float), 256);
        sl.gm_part = reinterpret_cast<float*>(ws);         ws += align_up((size_t)G * d0->C * kGmStride * sizeof(float), 256);
        sl.slab = ws;                                      ws += align_up(state_bytes(d0), 256);
        sl.ckpt = nck ? reinterpret_cast<float*>(ws) : nullptr;
        sl.nck = nck;
        sl.ck[0] = nck ? y.ckpt_mask[0] : 0ull; sl.ck[1] = nck ? y.ckpt_mask[1] : 0ull;
        any_ck |= nck != 0;
    }
    if (any_ck) {                                          // pre-pass: the forward again, parking the states inside the steps
        SmallArgs fa = sa;
        fa.out = nullptr; fa.gy = nullptr;
        rc = dispatch_small(true, d0, split, fa, small_lds_fwd(d0->C), st);
        if (rc != PDE_OK) return rc;
    }
    for (int i = 0; i < num_layers; ++i) {
        sa.layer[i].states = const_cast<void*>(layers[i].states);
        sa.layer[i].gys = layers[i].gys;
        sa.layer[i].roff = layers[i].g_plane_sums;
    }
    rc = dispatch_small(false, d0, split, sa, small_lds_bwd(d0->C), st);
    if (rc == PDE_OK && sa.par) rc = combine_slabs(d0, sa, gu, st);
    if (rc != PDE_OK) return rc;
    for (int i = 0; i < num_layers; ++i) {
        const PdeSmallLayer& y = layers[i];
        AxisWeights w;
        rc = axis_weights(y.desc, w);
        if (rc != PDE_OK) return rc;
        rc = launch_pgrad(y.desc, w, y.alpha_base, y.beta_base, y.alpha_slope, y.beta_slope, y.g_alpha_base, y.g_beta_base,
                          y.g_alpha_slope, y.g_beta_slope, sa.layer[i].varying, sa.layer[i].part, G, st, sa.layer[i].gm_part,
                          y.gM, y.skip_weight ? y.g_skip_weight : nullptr, y.skip_weight, y.g_weight);
        if (rc != PDE_OK) return rc;
    }
    return PDE_OK;
}

// one layer: thin wrappers
int pde_adi_small_forward(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t mode, const void* u, void* y, void* states,
                          const float* M, const float* skip_weight, const float* alpha_base, const float* beta_base,
                          const float* alpha_slope, const float* beta_slope, float* kappa_max, float* kappa_max_host,
                          void* kappa_event, void* steps_workspace, size_t workspace_bytes, void* stream) {
    PdeSmallLayer l{};
    l.desc = d; l.sweeps_per_step = sweeps_per_step; l.mode = mode; l.M = M; l.skip_weight = skip_weight;
    l.alpha_base = alpha_base; l.beta_base = beta_base; l.alpha_slope = alpha_slope; l.beta_slope = beta_slope;
    l.weight = 1.0f; l.states = states; l.steps_workspace = steps_workspace; l.steps_workspace_bytes = workspace_bytes;
    l.kappa_max = kappa_max; l.kappa_max_host = kappa_max_host;
    return pde_adi_multi_forward(1, &l, u, y, kappa_event, stream);
}

int pde_adi_small_backward(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t mode, const void* gy, const void* u,
                           const void* states, const float* M, const float* skip_weight, const uint64_t ckpt_mask[2],
                           void* gu, const float* alpha_base, const float* beta_base, const float* alpha_slope,
                           const float* beta_slope, float* g_alpha_base, float* g_beta_base, float* g_alpha_slope,
                           float* g_beta_slope, float* gM, float* g_skip_weight, const void* steps_workspace,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (!gy) return PDE_E_BADARG;
    PdeSmallLayer l{};
    l.desc = d; l.sweeps_per_step = sweeps_per_step; l.mode = mode; l.M = M; l.skip_weight = skip_weight;
    l.alpha_base = alpha_base; l.beta_base = beta_base; l.alpha_slope = alpha_slope; l.beta_slope = beta_slope;
    l.weight = 1.0f; l.states = const_cast<void*>(states); l.steps_workspace = const_cast<void*>(steps_workspace);
    l.ckpt_mask = ckpt_mask; l.g_alpha_base = g_alpha_base; l.g_beta_base = g_beta_base; l.g_alpha_slope = g_alpha_slope;
    l.g_beta_slope = g_beta_slope; l.gM = gM; l.g_skip_weight = g_skip_weight; l.workspace = workspace;
    l.workspace_bytes = workspace_bytes;
    return pde_adi_multi_backward(1, &l, gy, u, gu, stream);
}

int pde_adi_kappa_max(const PdeAdiDesc* d, const float* alpha_base, const float* beta_base, const float* alpha_slope,
                      const float* beta_slope, float* kappa_max, void* stream) {
    int rc = check_desc(d, true);
    if (rc != PDE_OK) return rc;
    if (!alpha_base || !beta_base || !alpha_slope || !beta_slope || !kappa_max) return PDE_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!fused_n(d->N)) return gen_kappa_max(d, alpha_base, beta_base, alpha_slope, beta_slope, kappa_max, st);
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
    std::lock_guard<std::mutex> lk(launch_mutex());
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

const char* pde_version(void) { return "pdecnn-hip 0.4 (gfx950)"; }

}  // extern "C"
