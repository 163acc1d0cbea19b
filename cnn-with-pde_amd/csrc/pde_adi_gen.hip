// K1 for ANY line length (2 <= N <= 128): the implicit sweeps of mnist_test.py:50-198 / cifar10.py:86-211 with one THREAD per
// line and the plane in LDS.  The fused kernels of pde_adi_dev.h hold a line in the registers of two lanes and exist for
// N = 8, 12, ..., 32 — the sizes the reference's own call sites use; its classes take any `size` (mnist_test.py:12,
// cifar10.py:25, SVHN.py:13), and this file is what serves the others: the reference's plain Thomas recurrences
// (mnist_test.py:151-198) per line, the adjoint as the transposed recurrences, the state rebuilt backwards or read from
// checkpoints exactly as the fused backward does, the clamp mask and the transposed 3-tap smoothing applied per sweep.
// Correct and deterministic, not tuned: a plane of 64 x 64 keeps 64 threads busy.
#include "pde_common.h"
#include "pde_adi_gen.h"

namespace pde {
namespace {

constexpr int kGenArr = 4;                  // per (sweep, channel): coeff | c* | 1/den | clamp pass-through, each [k][line]

struct GenSweep { int axis; float t, scale, pad; };     // device copy of the schedule: scale = delta/h2 as floats divide

struct GenFactorArgs {
    const float *ab, *bb, *as, *bs;
    float* fac;                              // [S][C][kGenArr][N*N]
    GenSweep* tab;                           // [S]
    float* kmax;                             // nullptr | [S], zeroed before the launch
    int C, N, S, smooth3, has_max;
    float cmax, eps;
    PdeSweep sweep[PDE_MAX_SWEEPS];
};

__device__ __forceinline__ float block_max(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float m = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
    return m;
}

// one workgroup per (sweep, channel), one thread per line
__global__ void gen_factor_kernel(GenFactorArgs a) {
    __shared__ float red[4];
    const int s = blockIdx.x / a.C, c = blockIdx.x % a.C, N = a.N, ln = threadIdx.x;
    const PdeSweep sw = a.sweep[s];
    const int ax = sw.axis;
    if (a.tab && c == 0 && ln == 0) a.tab[s] = GenSweep{ax, sw.t, sw.delta / sw.h2, 0.f};
    float kmx = 0.f;
    if (ln < N) {
        const float* base = (ax == PDE_AXIS_X ? a.ab : a.bb) + (size_t)c * N * N;
        const float* slope = (ax == PDE_AXIS_X ? a.as : a.bs) + (size_t)c * N * N;
        const int lstride = (ax == PDE_AXIS_X) ? N : 1, kstride = (ax == PDE_AXIS_X) ? 1 : N;
        float* f = a.fac ? a.fac + ((size_t)s * a.C + c) * kGenArr * N * N : nullptr;   // nullptr: the maxima alone
        auto raw = [&](int k) { const int i = ln * lstride + k * kstride; return base[i] + slope[i] * sw.t; };
        auto theta = [&](int k) {
            float th = fmaxf(raw(k), a.eps);
            if (a.has_max) th = fminf(th, a.cmax);
            return th;
        };
        const float third = 1.0f / 3.0f;
        float cs_prev = 0.f;
        for (int k = 0; k < N; ++k) {
            const float r = raw(k);
            const bool pass = (r >= a.eps) && (!a.has_max || r <= a.cmax);
            float th = theta(k);
            if (a.smooth3) th = (theta(k > 0 ? k - 1 : 0) * third + th * third) + theta(k + 1 < N ? k + 1 : N - 1) * third;
            const float co = th * sw.delta / sw.h2;
            const float b = (k == 0 || k == N - 1) ? 1.0f + co : 1.0f + 2.0f * co;
            const float den = (k ? b + co * cs_prev : b) + a.eps;
            const float cs = (k < N - 1) ? -co / den : 0.f;
            if (f) {
                const size_t o = (size_t)k * N + ln;
                f[o] = co;
                f[(size_t)N * N + o] = cs;
                f[(size_t)2 * N * N + o] = 1.0f / den;
                f[(size_t)3 * N * N + o] = pass ? 1.0f : 0.f;
            }
            cs_prev = cs;
            kmx = fmaxf(kmx, co);
        }
    }
    if (a.kmax) {
        const float m = block_max(kmx, red);
        if (threadIdx.x == 0) atomicMax(reinterpret_cast<int*>(a.kmax) + s, __float_as_int(m));   // coefficients are positive
    }
}

template <typename IO> struct GenIo;
template <> struct GenIo<float> {
    __device__ static float ld(const void* p, size_t i) { return static_cast<const float*>(p)[i]; }
    __device__ static void st(void* p, size_t i, float v) { static_cast<float*>(p)[i] = v; }
};
struct gen_bf16 { unsigned short v; };
template <> struct GenIo<gen_bf16> {
    __device__ static float ld(const void* p, size_t i) {
        return __uint_as_float((unsigned)static_cast<const unsigned short*>(p)[i] << 16);
    }
    __device__ static void st(void* p, size_t i, float v) { static_cast<unsigned short*>(p)[i] = f32_to_bf16_hw(v); }
};

struct GenSweepArgs {
    const void *in0, *in1;                   // forward: u, -; backward: gy, y
    void* out;                               // forward: y (nullptr: checkpoint pre-pass); backward: gu
    const float* fac;
    const GenSweep* tab;
    float* ckpt;                             // [nck][B][C][N*N] fp32 | nullptr
    float* part;                             // backward: [G][C][4][N*N]
    unsigned long long ck[2];
    int B, C, N, S, G;
    float eps;
};

__device__ __forceinline__ int gen_ck_bit(const unsigned long long (&ck)[2], int s) { return (int)((ck[s >> 6] >> (s & 63)) & 1ull); }
__device__ __forceinline__ int gen_ck_slot(const unsigned long long (&ck)[2], int s) {
    const unsigned long long below = (s & 63) ? (ck[s >> 6] & ((1ull << (s & 63)) - 1ull)) : 0ull;
    return __popcll(below) + ((s >> 6) ? __popcll(ck[0]) : 0);
}

// The per-line recurrences are serial, so what a thread waits for is latency: the loops below work in batches of kGenBatch
// elements — all LDS reads of the plane and all coefficient reads (global memory, L2-resident) of a batch are issued
// together, then the dependent arithmetic runs on registers, then the batch is written back.  (Written one element at a
// time the compiler must keep every LDS read behind the previous element's LDS write — it cannot know the stride is not
// zero — and a thread pays a full LDS or L2 round trip per element.)
constexpr int kGenBatch = 8;

// forward: one workgroup per plane; sweeps 0..S-1 on the plane in LDS ([row][N+1])
template <typename IO>
__global__ void gen_fwd_kernel(GenSweepArgs a) {
    extern __shared__ float X[];
    const int N = a.N, ld = N + 1, tid = threadIdx.x, T = blockDim.x, NN = N * N;
    const size_t plane = (size_t)NN, pb = (size_t)blockIdx.x * plane;         // blockIdx = b*C + c
    const int c = blockIdx.x % a.C;
    for (int e = tid; e < NN; e += T) X[(e / N) * ld + (e % N)] = GenIo<IO>::ld(a.in0, pb + e);
    __syncthreads();
    for (int s = 0; s < a.S; ++s) {
        const GenSweep sw = a.tab[s];
        if (tid < N) {
            const float* __restrict__ f = a.fac + ((size_t)s * a.C + c) * kGenArr * plane + tid;   // [arr][k][line = tid]
            float* v = X + (sw.axis == PDE_AXIS_X ? tid * ld : tid);
            const int st = (sw.axis == PDE_AXIS_X) ? 1 : ld;
            // d*_0 = d_0/den_0, d*_i = (d_i - a_i d*_{i-1})/den_i with a_i = -coeff_i   (mnist_test.py:167-185)
            float prev = v[0] * f[2 * plane];
            v[0] = prev;
            for (int k0 = 1; k0 < N; k0 += kGenBatch) {
                float t[kGenBatch], co[kGenBatch], iv[kGenBatch];
#pragma unroll
                for (int j = 0; j < kGenBatch; ++j) {
                    const int k = k0 + j < N ? k0 + j : N - 1;
                    t[j] = v[k * st];
                    co[j] = f[(size_t)k * N];
                    iv[j] = f[2 * plane + (size_t)k * N];
                }
#pragma unroll
                for (int j = 0; j < kGenBatch; ++j)
                    if (k0 + j < N) { prev = (t[j] + co[j] * prev) * iv[j]; t[j] = prev; }
#pragma unroll
                for (int j = 0; j < kGenBatch; ++j)
                    if (k0 + j < N) v[(k0 + j) * st] = t[j];
            }
            // x_{N-1} = d*_{N-1}, x_i = d*_i - c*_i x_{i+1}                                (mnist_test.py:187-196)
            for (int k0 = N - 2; k0 >= 0; k0 -= kGenBatch) {
                float t[kGenBatch], cs[kGenBatch];
#pragma unroll
                for (int j = 0; j < kGenBatch; ++j) {
                    const int k = k0 - j >= 0 ? k0 - j : 0;
                    t[j] = v[k * st];
                    cs[j] = f[plane + (size_t)k * N];
                }
#pragma unroll
                for (int j = 0; j < kGenBatch; ++j)
                    if (k0 - j >= 0) { prev = t[j] - cs[j] * prev; t[j] = prev; }
#pragma unroll
                for (int j = 0; j < kGenBatch; ++j)
                    if (k0 - j >= 0) v[(k0 - j) * st] = t[j];
            }
        }
        __syncthreads();
        if (a.ckpt && gen_ck_bit(a.ck, s)) {
            float* dst = a.ckpt + (size_t)gen_ck_slot(a.ck, s) * a.B * a.C * plane + pb;
            for (int e = tid; e < NN; e += T) dst[e] = X[(e / N) * ld + (e % N)];
        }
    }
    if (a.out)
        for (int e = tid; e < NN; e += T) GenIo<IO>::st(a.out, pb + e, X[(e / N) * ld + (e % N)]);
}

// backward: workgroup (c, g) walks the planes b = g, g+G, ... of channel c; adjoint in R, state in X (both LDS);
// parameter-gradient partial sums in part[g][c][arr][N*N], every entry owned by one thread of this workgroup.
// ALDS: the four partial-sum images live in LDS beside the two planes and go to `part` once, at the end (chosen while four
// workgroups still fit on a CU, see the launch); otherwise every update is a read-modify-write of global memory by the
// owning thread.
template <typename IO, bool ALDS>
__global__ void gen_bwd_kernel(GenSweepArgs a, int smooth3) {
    extern __shared__ float gen_smem[];
    const int N = a.N, ld = N + 1, tid = threadIdx.x, T = blockDim.x, NN = N * N;
    float* X = gen_smem;
    float* R = X + (size_t)N * ld;
    float* ACC = R + (size_t)N * ld;                      // ALDS: [4][N][ld], indexed like the planes
    const size_t plane = (size_t)NN;
    const int c = blockIdx.x % a.C, g = blockIdx.x / a.C;
    float* part = a.part + ((size_t)g * a.C + c) * 4 * plane;
    if constexpr (ALDS) {
        for (int e = tid; e < 4 * N * ld; e += T) ACC[e] = 0.f;
    } else {
        for (size_t e = tid; e < 4 * plane; e += T) part[e] = 0.f;
    }
    const float one_eps = 1.0f + a.eps, third = 1.0f / 3.0f;
    for (int b = g; b < a.B; b += a.G) {
        const size_t pb = ((size_t)b * a.C + c) * plane;
        __syncthreads();
        for (int e = tid; e < NN; e += T) {
            R[(e / N) * ld + (e % N)] = GenIo<IO>::ld(a.in0, pb + e);
            X[(e / N) * ld + (e % N)] = GenIo<IO>::ld(a.in1, pb + e);
        }
        __syncthreads();
        for (int s = a.S - 1; s >= 0; --s) {
            const GenSweep sw = a.tab[s];
            if (tid < N) {
                const float* __restrict__ f = a.fac + ((size_t)s * a.C + c) * kGenArr * plane + tid;   // [arr][k][line = tid]
                const bool xs = sw.axis == PDE_AXIS_X;
                float* r = R + (xs ? tid * ld : tid);
                float* x = X + (xs ? tid * ld : tid);
                const int st = xs ? 1 : ld;
                // transposed recurrences: U^T w = r (unit lower, sub-diagonal c*), L^T lam = w (diagonal den, super-diagonal a)
                float prev = r[0];
                for (int k0 = 1; k0 < N; k0 += kGenBatch) {
                    float t[kGenBatch], cs[kGenBatch];
#pragma unroll
                    for (int j = 0; j < kGenBatch; ++j) {
                        const int k = k0 + j < N ? k0 + j : N - 1;
                        t[j] = r[k * st];
                        cs[j] = f[plane + (size_t)(k - 1) * N];
                    }
#pragma unroll
                    for (int j = 0; j < kGenBatch; ++j)
                        if (k0 + j < N) { prev = t[j] - cs[j] * prev; t[j] = prev; }
#pragma unroll
                    for (int j = 0; j < kGenBatch; ++j)
                        if (k0 + j < N) r[(k0 + j) * st] = t[j];
                }
                prev = prev * f[2 * plane + (size_t)(N - 1) * N];
                r[(N - 1) * st] = prev;
                for (int k0 = N - 2; k0 >= 0; k0 -= kGenBatch) {
                    float t[kGenBatch], co[kGenBatch], iv[kGenBatch];
#pragma unroll
                    for (int j = 0; j < kGenBatch; ++j) {
                        const int k = k0 - j >= 0 ? k0 - j : 0;
                        t[j] = r[k * st];
                        co[j] = f[(size_t)(k + 1) * N];
                        iv[j] = f[2 * plane + (size_t)k * N];
                    }
#pragma unroll
                    for (int j = 0; j < kGenBatch; ++j)
                        if (k0 - j >= 0) { prev = (t[j] + co[j] * prev) * iv[j]; t[j] = prev; }
#pragma unroll
                    for (int j = 0; j < kGenBatch; ++j)
                        if (k0 - j >= 0) r[(k0 - j) * st] = t[j];
                }
                // coefficient gradient -lam.q with the sweep's OUTPUT state, q = (Neumann second difference, sign flipped);
                // x_old = (1+eps) x + coeff q; then the transposed smoothing (entry j is complete once k = j+1 is known),
                // the clamp mask, and the two parameters: d/d base, d/d slope = t * d/d base
                float* __restrict__ pbase = part + (xs ? 0 : 2) * plane;
                float* __restrict__ pslope = pbase + plane;
                // partial sums in global memory are kept [k][line] for BOTH axes (threads of a wave then touch consecutive
                // words; [line][k] made every x-sweep update a cache line of its own): the alpha images are stored
                // transposed and gen_reduce_kernel turns them back
                const int pl = tid, pk = N;
                float* lbase = ACC + (xs ? 0 : 2) * N * ld + (xs ? tid * ld : tid);
                auto add = [&](int j, float gv) __attribute__((always_inline)) {
                    if constexpr (ALDS) {
                        lbase[j * st] += gv;
                        lbase[N * ld + j * st] += sw.t * gv;
                    } else {
                        pbase[pl + j * pk] += gv;
                        pslope[pl + j * pk] += sw.t * gv;
                    }
                };
                float xm = 0.f, xc = x[0], g2 = 0.f, g1 = 0.f;            // x_{k-1}, x_k; gsm_{k-2}, gsm_{k-1}
                for (int k0 = 0; k0 < N; k0 += kGenBatch) {
                    float xn[kGenBatch], lam[kGenBatch], co[kGenBatch], ps[kGenBatch], gout[kGenBatch];
#pragma unroll
                    for (int j = 0; j < kGenBatch; ++j) {
                        const int k = k0 + j < N ? k0 + j : N - 1;
                        xn[j] = (k0 + j + 1 < N) ? x[(k0 + j + 1) * st] : 0.f;      // x_{k+1}
                        lam[j] = r[k * st];
                        co[j] = f[(size_t)k * N];
                        ps[j] = f[3 * plane + (size_t)k * N];
                    }
                    const float ps_before = k0 > 0 ? f[3 * plane + (size_t)(k0 - 1) * N] : 0.f;   // mask of entry k0-1 (smoothing)
#pragma unroll
                    for (int j = 0; j < kGenBatch; ++j) {
                        const int k = k0 + j;
                        gout[j] = 0.f;
                        if (k < N) {
                            const float xp = xn[j];
                            const float q = ((k == 0 || k == N - 1) ? xc : 2.0f * xc) - xm - xp;
                            const float g0 = -lam[j] * q * sw.scale;
                            xn[j] = one_eps * xc + co[j] * q;                       // becomes x_old[k]
                            xm = xc;
                            xc = xp;
                            if (!smooth3) {
                                gout[j] = g0 * ps[j];                               // entry k
                            } else if (k >= 1) {                                    // finishes entry k-1
                                float gv = (g2 * third + g1 * third) + g0 * third;
                                if (k == 1) gv += g1 * third;                       // replicate end: theta_0 is used twice by sm_0
                                gout[j] = gv * (j > 0 ? ps[j - 1] : ps_before);
                            }
                            g2 = g1;
                            g1 = g0;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < kGenBatch; ++j) {
                        const int k = k0 + j;
                        if (k < N) {
                            x[k * st] = xn[j];
                            if (!smooth3) add(k, gout[j]);
                            else if (k >= 1) add(k - 1, gout[j]);
                        }
                    }
                }
                if (smooth3) {                                             // entry N-1: (gsm_{N-2} + 2 gsm_{N-1}) / 3
                    const float gv = (g2 * third + g1 * third) + g1 * third;
                    add(N - 1, gv * f[3 * plane + (size_t)(N - 1) * N]);
                }
            }
            __syncthreads();
            if (s > 0 && a.ckpt && gen_ck_bit(a.ck, s - 1)) {             // the parked state instead of the rebuilt one
                const float* src = a.ckpt + (size_t)gen_ck_slot(a.ck, s - 1) * a.B * a.C * plane + pb;
                for (int e = tid; e < NN; e += T) X[(e / N) * ld + (e % N)] = src[e];
                __syncthreads();
            }
        }
        for (int e = tid; e < NN; e += T) GenIo<IO>::st(a.out, pb + e, R[(e / N) * ld + (e % N)]);
    }
    if constexpr (ALDS) {
        __syncthreads();
        for (int arr = 0; arr < 4; ++arr)
            for (int e = tid; e < NN; e += T)              // alpha images transposed, as above
                part[arr * plane + e] = ACC[arr * N * ld + (arr < 2 ? (e % N) * ld + (e / N) : (e / N) * ld + (e % N))];
    }
}

// the four parameter gradients: partial sums added over the groups in a fixed order
__global__ void gen_reduce_kernel(const float* part, int G, int C, int N, float* g_ab, float* g_as, float* g_bb, float* g_bs) {
    const int NN = N * N, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= C * NN) return;
    const int c = e / NN, p = e % NN, pt = (p % N) * N + p / N;        // the alpha images are stored transposed
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < G; ++g)
        for (int arr = 0; arr < 4; ++arr) s[arr] += part[(((size_t)g * C + c) * 4 + arr) * NN + (arr < 2 ? pt : p)];
    g_ab[e] = s[0]; g_as[e] = s[1]; g_bb[e] = s[2]; g_bs[e] = s[3];
}

size_t up256(size_t x) { return (x + 255) / 256 * 256; }
size_t fac_bytes(const PdeAdiDesc* d) { return up256((size_t)d->num_sweeps * d->C * kGenArr * d->N * d->N * sizeof(float)); }
size_t tab_bytes_gen() { return up256(sizeof(GenSweep) * PDE_MAX_SWEEPS); }
int gen_groups(const PdeAdiDesc* d) {
    int G = (1024 + d->C - 1) / d->C;
    return G > d->B ? d->B : (G < 1 ? 1 : G);
}
int gen_threads(int N) { return (N + 63) / 64 * 64; }

int launch_gen_factor(const PdeAdiDesc* d, const float* ab, const float* bb, const float* as, const float* bs, float* fac,
                      GenSweep* tab, float* kmax, hipStream_t st) {
    GenFactorArgs fa;
    fa.ab = ab; fa.bb = bb; fa.as = as; fa.bs = bs; fa.fac = fac; fa.tab = tab; fa.kmax = kmax;
    fa.C = d->C; fa.N = d->N; fa.S = d->num_sweeps; fa.smooth3 = d->smooth3; fa.has_max = d->has_clamp_max;
    fa.cmax = d->clamp_max; fa.eps = d->eps;
    for (int s = 0; s < d->num_sweeps; ++s) fa.sweep[s] = d->sweep[s];
    if (kmax && hipMemsetAsync(kmax, 0, sizeof(float) * d->num_sweeps, st) != hipSuccess) return PDE_E_LAUNCH;
    hipLaunchKernelGGL(gen_factor_kernel, dim3(d->num_sweeps * d->C), dim3(gen_threads(d->N)), 0, st, fa);
    return check_launch();
}

constexpr int kGenLdsMax = 160 * 1024;        // the CU's LDS (two 128 x 129 planes are 132 KB)
template <typename K>
int gen_lds(K kernel, unsigned long long& done) { return ensure_dynamic_lds((const void*)kernel, kGenLdsMax, done); }

int launch_gen_fwd(const PdeAdiDesc* d, const void* u, void* y, const float* fac, const GenSweep* tab, int S, float* ckpt,
                   const uint64_t ck[2], hipStream_t st) {
    GenSweepArgs sa{};
    sa.in0 = u; sa.out = y; sa.fac = fac; sa.tab = tab; sa.ckpt = ckpt;
    sa.ck[0] = ck ? ck[0] : 0ull; sa.ck[1] = ck ? ck[1] : 0ull;
    sa.B = d->B; sa.C = d->C; sa.N = d->N; sa.S = S; sa.eps = d->eps;
    const size_t lds = (size_t)d->N * (d->N + 1) * sizeof(float);
    static unsigned long long done_f = 0, done_b = 0;
    int rc;
    if (d->io_dtype == PDE_IO_F32) {
        if ((rc = gen_lds(gen_fwd_kernel<float>, done_f)) != PDE_OK) return rc;
        hipLaunchKernelGGL(gen_fwd_kernel<float>, dim3(d->B * d->C), dim3(gen_threads(d->N)), lds, st, sa);
    } else {
        if ((rc = gen_lds(gen_fwd_kernel<gen_bf16>, done_b)) != PDE_OK) return rc;
        hipLaunchKernelGGL(gen_fwd_kernel<gen_bf16>, dim3(d->B * d->C), dim3(gen_threads(d->N)), lds, st, sa);
    }
    return check_launch();
}

}  // namespace

bool gen_n_ok(int N) { return N >= 2 && N <= PDE_MAX_N_GENERIC; }

size_t gen_forward_workspace_bytes(const PdeAdiDesc* d) { return fac_bytes(d) + tab_bytes_gen(); }

size_t gen_backward_workspace_bytes(const PdeAdiDesc* d, int nck) {
    return fac_bytes(d) + tab_bytes_gen() + up256((size_t)gen_groups(d) * d->C * 4 * d->N * d->N * sizeof(float)) +
           up256((size_t)nck * d->B * d->C * d->N * d->N * sizeof(float));
}

int gen_kappa_max(const PdeAdiDesc* d, const float* ab, const float* bb, const float* as, const float* bs, float* kmax,
                  hipStream_t st) {
    return launch_gen_factor(d, ab, bb, as, bs, nullptr, nullptr, kmax, st);
}

int gen_factor(const PdeAdiDesc* d, const float* ab, const float* bb, const float* as, const float* bs, float* kmax,
               void* workspace, hipStream_t st) {
    char* ws = static_cast<char*>(workspace);
    return launch_gen_factor(d, ab, bb, as, bs, reinterpret_cast<float*>(ws), reinterpret_cast<GenSweep*>(ws + fac_bytes(d)),
                             kmax, st);
}

int gen_forward_sweeps(const PdeAdiDesc* d, const void* u, void* y, const void* workspace, hipStream_t st) {
    const char* ws = static_cast<const char*>(workspace);
    return launch_gen_fwd(d, u, y, reinterpret_cast<const float*>(ws), reinterpret_cast<const GenSweep*>(ws + fac_bytes(d)),
                          d->num_sweeps, nullptr, nullptr, st);
}

int gen_backward(const PdeAdiDesc* d, const void* gy, const void* y, const void* u, const uint64_t ckpt_mask[2], int nck,
                 int Sf, void* gu, const float* ab, const float* bb, const float* as, const float* bs, float* g_ab,
                 float* g_bb, float* g_as, float* g_bs, const void* fwd_workspace, void* workspace, hipStream_t st) {
    char* ws = static_cast<char*>(workspace);
    const float* fac = reinterpret_cast<const float*>(ws);
    const GenSweep* tab = reinterpret_cast<const GenSweep*>(ws + fac_bytes(d));
    ws += fac_bytes(d) + tab_bytes_gen();
    const int G = gen_groups(d);
    float* part = reinterpret_cast<float*>(ws);
    ws += up256((size_t)G * d->C * 4 * d->N * d->N * sizeof(float));
    float* ckpt = nck ? reinterpret_cast<float*>(ws) : nullptr;
    int rc;
    if (fwd_workspace) {
        const char* fw = static_cast<const char*>(fwd_workspace);
        fac = reinterpret_cast<const float*>(fw);
        tab = reinterpret_cast<const GenSweep*>(fw + fac_bytes(d));
    } else {
        rc = launch_gen_factor(d, ab, bb, as, bs, const_cast<float*>(fac), const_cast<GenSweep*>(tab), nullptr, st);
        if (rc != PDE_OK) return rc;
    }
    if (nck) {
        rc = launch_gen_fwd(d, u, nullptr, fac, tab, Sf, ckpt, ckpt_mask, st);
        if (rc != PDE_OK) return rc;
    }
    GenSweepArgs sa{};
    sa.in0 = gy; sa.in1 = y; sa.out = gu; sa.fac = fac; sa.tab = tab; sa.ckpt = ckpt; sa.part = part;
    sa.ck[0] = nck ? ckpt_mask[0] : 0ull; sa.ck[1] = nck ? ckpt_mask[1] : 0ull;
    sa.B = d->B; sa.C = d->C; sa.N = d->N; sa.S = d->num_sweeps; sa.G = G; sa.eps = d->eps;
    const size_t img = (size_t)d->N * (d->N + 1) * sizeof(float);
    const bool alds = 4 * 6 * img <= (size_t)kGenLdsMax;  // the partial sums beside the planes while four workgroups fit on a CU (N <= 40)
    const size_t lds = (alds ? 6 : 2) * img;
    static unsigned long long done[4] = {0, 0, 0, 0};
    const dim3 grid(G * d->C), block(gen_threads(d->N));
#define PDE_GEN_BWD(IO, AL, SLOT)                                                                        \
    do {                                                                                                 \
        if ((rc = gen_lds(gen_bwd_kernel<IO, AL>, done[SLOT])) != PDE_OK) return rc;                     \
        hipLaunchKernelGGL((gen_bwd_kernel<IO, AL>), grid, block, lds, st, sa, (int)d->smooth3);         \
    } while (0)
    if (d->io_dtype == PDE_IO_F32) {
        if (alds) PDE_GEN_BWD(float, true, 0); else PDE_GEN_BWD(float, false, 1);
    } else {
        if (alds) PDE_GEN_BWD(gen_bf16, true, 2); else PDE_GEN_BWD(gen_bf16, false, 3);
    }
#undef PDE_GEN_BWD
    if ((rc = check_launch()) != PDE_OK) return rc;
    const int NN = d->N * d->N, total = d->C * NN;
    hipLaunchKernelGGL(gen_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, part, G, d->C, d->N, g_ab, g_as, g_bb, g_bs);
    return check_launch();
}

}  // namespace pde
