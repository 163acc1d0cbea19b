// The Ruthotto-Haber "symmetric layer" of cifar_2version.py:190-220 on the fp32 matrix cores of gfx950:
//
//     F_sym(Y) = -act(BN(Y K^T)) K            Y: (B, D) flattened image batch, K: (D, D) dense, D = C*H*W (3072)
//
// and the residual steps built on it (ParabolicBlock :223-236, HamiltonianBlock :239-258), which all have the form
// out = base + scale * (act(BN(X K^T)) K).  Three kernels on the fp32-input MFMAs (v_mfma_f32_16x16x4_f32 / 32x32x2_f32:
// exact fp32 FMA chains at the fp32 vector rate, VALU left free for the epilogues):
//
//   strip kernel, NT  P = X K^T for ALL batch rows (up to 128: the reference's batch; larger batches go by row blocks with the
//                     statistics in a pass of their own) and a strip of 16 output features per workgroup, so the BatchNorm1d
//                     statistics of a feature (over the batch) are workgroup-local and normalisation + activation are
//                     the epilogue of the product (forward); in the backward the same product shape carries
//                     dH = dF K^T with the activation derivative and the BatchNorm backward (two per-feature sums)
//                     as epilogue;
//   strip kernel, NN  out = base + scale * (H K): the residual update as epilogue (forward), dX = dP K (backward);
//   outer kernel      dK = dP^T X + s * H^T dF: both uses of K in one pass over the (D, D) gradient (contraction over
//                     the batch: the output write, 37.7 MB for D = 3072, is what bounds it).
//
// Boundary: plain pointers, caller-owned buffers, asynchronous on the given stream (include/pdecnn.h).
#include "pde_common.h"

#include <map>
#include <mutex>

#include <cstdlib>

namespace pde {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRhThreads = 256;          // outer kernel: 4 waves
constexpr int kRhStripThreads = 512;     // strip kernels: 8 waves, two per SIMD
constexpr int kRhWaves = 8;
constexpr int kRhCols = 16;              // output features per strip workgroup: D/16 workgroups (192 for D = 3072)
// contraction slab of a strip workgroup with R row blocks per wave: 64 / R, so that the X slab (128 R rows) stays ~35 KB
// and two of them (double buffering: one barrier per slab) fit beside each other whatever the batch
template <int R> struct RhSlab { static constexpr int BK = 64 / R, LDA = BK + 4; };
constexpr int kRhLdN = 20;
constexpr int kRhRows32 = 128;             // batch rows of a 32-column strip workgroup (four waves x 32)               // LDS row stride of a [BK][16] slab (NN operand): the four k-quarters hit disjoint banks

enum { kActIdentity = 0, kActRelu = 1, kActTanh = 2 };

// ---- strip product: acc[b][j] = sum_k X[b][k] * Wop[k][j] for every batch row (padded) and 16 output features ----
//   NT: Wop[k][j] = W[(n0 + j) * ldw + k]        NN: Wop[k][j] = W[k * ldw + n0 + j]
// v_mfma_f32_16x16x4_f32: A lane l holds A[i = l & 15][k = l >> 4], B lane l holds B[k = l >> 4][j = l & 15], D lane l
// holds rows 4 * (l >> 4) + r (r = 0..3) of column l & 15.  Wave w owns the row blocks (16 rows) w, w + 8, ... (R of
// them; the kernels are instantiated with R = 1: 128 batch rows per workgroup, larger batches are cut into row blocks).  Contraction order inside a group of 16: step s (0..3) takes k = 4 * kq + s from quarter kq = l >> 4 — one
// 16-byte LDS read feeds four steps.  16 columns per workgroup instead of 32 (v_mfma_f32_32x32x2_f32) because a
// batch of 128 is only 96 strips of 32: with 192 the product runs on three quarters of the chip instead of three eighths.
template <int R, bool NT>
__device__ __forceinline__ void strip_gemm(const float* __restrict__ X, int B, int Kdim, int ldx, const float* __restrict__ W,
                                           int ldw, int n0, f32x4 (&acc)[R], float* As, float* Ws) {
    constexpr int BK = RhSlab<R>::BK, LDA = RhSlab<R>::LDA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jj = lane & 15, kq = lane >> 4;
    constexpr int ROWS = kRhWaves * R * 16;               // padded batch rows held by the workgroup
    constexpr int C4 = BK / 4;                            // float4 per slab row
    constexpr int APT = ROWS * C4 / kRhStripThreads;      // X-slab float4 per thread (4)
    constexpr int WF4 = kRhCols * BK / 4;                 // W-slab float4 (<= 256)
    constexpr int ASZ = ROWS * LDA, WSZ = NT ? kRhCols * LDA : BK * kRhLdN;      // floats per buffer
#pragma unroll
    for (int m = 0; m < R; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 pa[APT], pw;
    auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < APT; ++m) {
            const int f = tid + kRhStripThreads * m, row = f / C4, c4 = f % C4;
            pa[m] = (row < B) ? *reinterpret_cast<const float4*>(X + (size_t)row * ldx + k0 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < WF4) {
            if (NT) {
                const int row = tid / C4, c4 = tid % C4;  // feature n0 + row, k0 + 4 c4
                pw = *reinterpret_cast<const float4*>(W + (size_t)(n0 + row) * ldw + k0 + 4 * c4);
            } else {
                const int row = tid >> 2, c4 = tid & 3;   // k0 + row, features n0 + 4 c4
                pw = *reinterpret_cast<const float4*>(W + (size_t)(k0 + row) * ldw + n0 + 4 * c4);
            }
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        float* A = As + buf * ASZ;
        float* Wb = Ws + buf * WSZ;
#pragma unroll
        for (int m = 0; m < APT; ++m) {
            const int f = tid + kRhStripThreads * m, row = f / C4, c4 = f % C4;
            *reinterpret_cast<float4*>(A + row * LDA + 4 * c4) = pa[m];
        }
        if (tid < WF4) {
            if (NT) {
                const int row = tid / C4, c4 = tid % C4;
                *reinterpret_cast<float4*>(Wb + row * LDA + 4 * c4) = pw;
            } else {
                const int row = tid >> 2, c4 = tid & 3;
                *reinterpret_cast<float4*>(Wb + row * kRhLdN + 4 * c4) = pw;
            }
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < Kdim; k0 += BK) {
        const bool more = k0 + BK < Kdim;
        if (more) fetch(k0 + BK);                         // global -> registers while the matrix cores work on this slab
        const float* A = As + buf * ASZ;
        const float* Wb = Ws + buf * WSZ;
#pragma unroll
        for (int t = 0; t < BK / 16; ++t) {
            float b[4];
            if (NT) {
                const float4 bv = *reinterpret_cast<const float4*>(Wb + jj * LDA + 16 * t + 4 * kq);
                b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) b[s] = Wb[(16 * t + 4 * kq + s) * kRhLdN + jj];
            }
#pragma unroll
            for (int m = 0; m < R; ++m) {
                const int rb = wave + kRhWaves * m;
                const float4 av = *reinterpret_cast<const float4*>(A + (rb * 16 + jj) * LDA + 16 * t + 4 * kq);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[0], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[1], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[2], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[3], acc[m], 0, 0, 0);
            }
        }
        if (more) stage(buf ^ 1);                         // the other buffer: last read one barrier ago
        __syncthreads();
        buf ^= 1;
    }
}

// batch row of accumulator component r of this wave's m-th row block (column = n0 + (lane & 15))
__device__ __forceinline__ int acc_row(int wave, int m, int r, int kq) { return (wave + kRhWaves * m) * 16 + 4 * kq + r; }

// sum over the whole workgroup of a per-lane value that belongs to column (lane & 15): every lane gets its column's total
__device__ __forceinline__ float column_total(float v, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jj = lane & 15;
    __syncthreads();                                      // `red` may still be read from the previous call
    red[wave * 64 + lane] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kRhWaves; ++w)                    // fixed order: deterministic
        s += (red[w * 64 + jj] + red[w * 64 + 16 + jj]) + (red[w * 64 + 32 + jj] + red[w * 64 + 48 + jj]);
    return s;
}

__device__ __forceinline__ float act_fwd(float x, int act) {
    return act == kActRelu ? fmaxf(x, 0.f) : (act == kActTanh ? tanhf(x) : x);
}
// derivative written in terms of the activation's OUTPUT h
__device__ __forceinline__ float act_bwd(float h, int act) {
    return act == kActRelu ? (h > 0.f ? 1.f : 0.f) : (act == kActTanh ? 1.f - h * h : 1.f);
}

struct RhFwdArgs {
    const float* X; const float* K; const float* gamma; const float* beta;
    float* run_mean; float* run_var;
    float* P; float* H; float* mean; float* invstd;
    int B, D, act, training;
    float momentum, eps;
};

// P = X K^T, BatchNorm1d over the batch (cifar_2version.py:201, 214-215), activation (:216)
template <int R>
__global__ __launch_bounds__(kRhStripThreads) void rh_fwd_strip_kernel(RhFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float rh_smem[];
    float* As = rh_smem;                                                 // [2][128 R][LDA]
    float* Ws = As + 2 * kRhWaves * R * 16 * RhSlab<R>::LDA;             // [2][16][LDA]
    float* red = Ws + 2 * kRhCols * RhSlab<R>::LDA;                      // [8][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, jj = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * kRhCols, col = n0 + jj;
    f32x4 acc[R];
    strip_gemm<R, true>(a.X, a.B, a.D, a.D, a.K, a.D, n0, acc, As, Ws);

    float mu, istd;
    if (a.training) {
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[m][r];   // padded rows hold exact zeros
        mu = column_total(s, red) / (float)a.B;
        float q = 0.f;
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dlt = acc[m][r] - mu;
                if (acc_row(wave, m, r, kq) < a.B) q = fmaf(dlt, dlt, q);
            }
        const float var = column_total(q, red) / (float)a.B;             // biased, as BatchNorm normalises
        istd = 1.0f / sqrtf(var + a.eps);
        if (threadIdx.x < kRhCols && a.run_mean != nullptr) {             // running statistics (unbiased variance)
            const float unb = a.B > 1 ? var * (float)a.B / (float)(a.B - 1) : var;
            a.run_mean[col] = (1.f - a.momentum) * a.run_mean[col] + a.momentum * mu;
            a.run_var[col] = (1.f - a.momentum) * a.run_var[col] + a.momentum * unb;
        }
    } else {
        mu = a.run_mean[col];
        istd = 1.0f / sqrtf(a.run_var[col] + a.eps);
    }
    if (threadIdx.x < kRhCols) { a.mean[col] = mu; a.invstd[col] = istd; }
    const float g = a.gamma[col], bt = a.beta[col];
#pragma unroll
    for (int m = 0; m < R; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = acc_row(wave, m, r, kq);
            if (row < a.B) {
                const float p = acc[m][r];
                const float hn = fmaf((p - mu) * istd, g, bt);
                a.P[(size_t)row * a.D + col] = p;
                a.H[(size_t)row * a.D + col] = act_fwd(hn, a.act);
            }
        }
}

struct RhBwdArgs {
    const float* G; const float* K; const float* gamma;
    const float* P; const float* H; const float* mean; const float* invstd;
    float* dP; float* g_gamma; float* g_beta;
    int B, D, act, training;
    float scale;
};

struct RhAxpyArgs {
    const float* X; const float* K; const float* base; float* out;
    int B, D;
    float scale;
};

// out = base + scale * (X K)     (cifar_2version.py:217 and the residual updates :234, :254-255; backward: dX = dP K)
template <int R>
__global__ __launch_bounds__(kRhStripThreads) void rh_axpy_strip_kernel(RhAxpyArgs a) {
    extern __shared__ __attribute__((aligned(16))) float rh_smem[];
    float* As = rh_smem;
    float* Ws = As + 2 * kRhWaves * R * 16 * RhSlab<R>::LDA;             // [2][BK][20]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, jj = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * kRhCols, col = n0 + jj;
    const int row0 = blockIdx.y * (kRhWaves * R * 16);    // row block of this workgroup (rows are independent here)
    const int rows = a.B - row0 < kRhWaves * R * 16 ? a.B - row0 : kRhWaves * R * 16;
    f32x4 acc[R];
    strip_gemm<R, false>(a.X + (size_t)row0 * a.D, rows, a.D, a.D, a.K, a.D, n0, acc, As, Ws);
#pragma unroll
    for (int m = 0; m < R; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = acc_row(wave, m, r, kq);
            if (row < rows) {
                const size_t o = (size_t)(row0 + row) * a.D + col;
                a.out[o] = a.base != nullptr ? fmaf(a.scale, acc[m][r], a.base[o]) : a.scale * acc[m][r];
            }
        }
}

// out = X K^T, nothing else: the product of the batches above 128 rows, whose BatchNorm statistics span several row
// blocks (= workgroups) and are taken by rh_bn_fwd_kernel / rh_bn_bwd_kernel below
template <int R>
__global__ __launch_bounds__(kRhStripThreads) void rh_nt_strip_kernel(RhAxpyArgs a) {
    extern __shared__ __attribute__((aligned(16))) float rh_smem[];
    float* As = rh_smem;
    float* Ws = As + 2 * kRhWaves * R * 16 * RhSlab<R>::LDA;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, jj = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * kRhCols, col = n0 + jj;
    const int row0 = blockIdx.y * (kRhWaves * R * 16);
    const int rows = a.B - row0 < kRhWaves * R * 16 ? a.B - row0 : kRhWaves * R * 16;
    f32x4 acc[R];
    strip_gemm<R, true>(a.X + (size_t)row0 * a.D, rows, a.D, a.D, a.K, a.D, n0, acc, As, Ws);
#pragma unroll
    for (int m = 0; m < R; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = acc_row(wave, m, r, kq);
            if (row < rows) a.out[(size_t)(row0 + row) * a.D + col] = acc[m][r];
        }
}

// ---- BatchNorm1d over a batch of any size, beside the products (64 features per workgroup; thread = (feature, row
// quarter); the (B, 64) slice is read two or three times from L2) -------------------------------------------------
constexpr int kBnCols = 64;
__device__ __forceinline__ float bn_column_total(float v, float (*red)[kBnCols]) {
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    __syncthreads();
    red[q][c] = v;
    __syncthreads();
    return (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// P (B, D) -> statistics, H = act(BN(P)); running statistics as in rh_fwd_strip_kernel
__global__ __launch_bounds__(256) void rh_bn_fwd_kernel(RhFwdArgs a) {
    __shared__ float red[4][kBnCols];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6, col = blockIdx.x * kBnCols + c;
    float mu, istd;
    if (a.training) {
        float s = 0.f;
        for (int r = q; r < a.B; r += 4) s += a.P[(size_t)r * a.D + col];
        mu = bn_column_total(s, red) / (float)a.B;
        float v = 0.f;
        for (int r = q; r < a.B; r += 4) { const float d = a.P[(size_t)r * a.D + col] - mu; v = fmaf(d, d, v); }
        const float var = bn_column_total(v, red) / (float)a.B;
        istd = 1.0f / sqrtf(var + a.eps);
        if (q == 0 && a.run_mean != nullptr) {
            const float unb = a.B > 1 ? var * (float)a.B / (float)(a.B - 1) : var;
            a.run_mean[col] = (1.f - a.momentum) * a.run_mean[col] + a.momentum * mu;
            a.run_var[col] = (1.f - a.momentum) * a.run_var[col] + a.momentum * unb;
        }
    } else {
        mu = a.run_mean[col];
        istd = 1.0f / sqrtf(a.run_var[col] + a.eps);
    }
    if (q == 0) { a.mean[col] = mu; a.invstd[col] = istd; }
    const float g = a.gamma[col], bt = a.beta[col];
    for (int r = q; r < a.B; r += 4) {
        const size_t o = (size_t)r * a.D + col;
        a.H[o] = act_fwd(fmaf((a.P[o] - mu) * istd, g, bt), a.act);
    }
}

// dP holds G K^T on entry; on exit the gradient of P.  dgamma, dbeta.
__global__ __launch_bounds__(256) void rh_bn_bwd_kernel(RhBwdArgs a) {
    __shared__ float red[4][kBnCols];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6, col = blockIdx.x * kBnCols + c;
    const float mu = a.mean[col], istd = a.invstd[col], g = a.gamma[col];
    float sb = 0.f, sg = 0.f;
    for (int r = q; r < a.B; r += 4) {
        const size_t o = (size_t)r * a.D + col;
        const float dhn = a.scale * a.dP[o] * act_bwd(a.H[o], a.act);
        sb += dhn;
        sg = fmaf(dhn, (a.P[o] - mu) * istd, sg);
    }
    const float dbeta = bn_column_total(sb, red);
    const float dgamma = bn_column_total(sg, red);
    if (q == 0) { a.g_beta[col] = dbeta; a.g_gamma[col] = dgamma; }
    const float inv_b = 1.0f / (float)a.B;
    for (int r = q; r < a.B; r += 4) {
        const size_t o = (size_t)r * a.D + col;
        const float dhn = a.scale * a.dP[o] * act_bwd(a.H[o], a.act);
        const float xhat = (a.P[o] - mu) * istd;
        a.dP[o] = a.training ? g * istd * (dhn - (dbeta + xhat * dgamma) * inv_b) : g * istd * dhn;
    }
}


// dH = scale * (G K^T); through the activation and the BatchNorm (training: batch statistics take part) -> dP, dgamma, dbeta
template <int R>
__global__ __launch_bounds__(kRhStripThreads) void rh_bwd_strip_kernel(RhBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float rh_smem[];
    float* As = rh_smem;
    float* Ws = As + 2 * kRhWaves * R * 16 * RhSlab<R>::LDA;
    float* red = Ws + 2 * kRhCols * RhSlab<R>::LDA;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, jj = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * kRhCols, col = n0 + jj;
    f32x4 acc[R];
    strip_gemm<R, true>(a.G, a.B, a.D, a.D, a.K, a.D, n0, acc, As, Ws);
    const float mu = a.mean[col], istd = a.invstd[col], g = a.gamma[col];
    float sb = 0.f, sg = 0.f;
    f32x4 xh[R];
#pragma unroll
    for (int m = 0; m < R; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = acc_row(wave, m, r, kq);
            float dhn = 0.f, xhat = 0.f;
            if (row < a.B) {
                const size_t o = (size_t)row * a.D + col;
                dhn = a.scale * acc[m][r] * act_bwd(a.H[o], a.act);
                xhat = (a.P[o] - mu) * istd;
            }
            acc[m][r] = dhn;
            xh[m][r] = xhat;
            sb += dhn;
            sg = fmaf(dhn, xhat, sg);
        }
    const float dbeta = column_total(sb, red);
    const float dgamma = column_total(sg, red);
    if (threadIdx.x < kRhCols) { a.g_beta[col] = dbeta; a.g_gamma[col] = dgamma; }
    const float inv_b = 1.0f / (float)a.B;
#pragma unroll
    for (int m = 0; m < R; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = acc_row(wave, m, r, kq);
            if (row < a.B) {
                const float dhn = acc[m][r];
                const float dp = a.training ? g * istd * (dhn - (dbeta + xh[m][r] * dgamma) * inv_b) : g * istd * dhn;
                a.dP[(size_t)row * a.D + col] = dp;
            }
        }
}

// ---- batches up to 128 rows: 32-column strips, the contraction split over workgroups ----------------------------------
// One 16-column strip workgroup per CU (D / 16 = 192 of them at D = 3072) leaves the matrix cores waiting: for its K rows
// from HBM (a slab of MFMAs does not cover that latency and there is no second workgroup on the CU to fill in) and for
// LDS (a 16 x 16 x 4 MFMA consumes 512 B of operands per 32 cycles).  Here a workgroup of four waves owns 128 rows x 32
// columns on v_mfma_f32_32x32x2_f32 (the same 512 B feed 64 cycles), and the contraction of a strip is split over S
// workgroups (96 strips x 8 = 768 = three resident per CU at D = 3072): each leaves its partial tile in the workspace
// (30 us against 48 for the product, tools/ubench/rh_strip32), and a second, small launch per product adds the S tiles
// of a strip in split order and runs the epilogue (BatchNorm statistics, activation, residual update).  The kernel
// boundary is the hand-off: an in-kernel one (the workgroup whose ticket comes last finishes the strip) was built and
// measured — with the agent-scope release / acquire the non-coherent per-XCD L2s need, it gave back all but 3 % of the
// gain (254 us per forward + backward against 261; 117 us per PRODUCT with __threadfence() in every thread).
constexpr int kS32Cols = 32, kS32Bk = 32, kS32Lda = kS32Bk + 4, kS32LdN = kS32Cols + 4;
// WV waves per workgroup, 32 batch rows each: 4 (up to 128 rows) or 2 (up to 64, the reference's batch: half the rows of the
// four-wave tile would be padding there); one partial tile is [wave][4][lane] float4
constexpr int tile32_floats(int wv) { return wv * 64 * 16; }

template <bool NT, int WV>
__device__ __forceinline__ void strip32_gemm(const float* __restrict__ X, int B, int klen, int ldx, const float* __restrict__ W,
                                             int ldw, int n0, f32x16& acc, float* As, float* Ws) {
    constexpr int BK = kS32Bk, LDA = kS32Lda, C4 = BK / 4;
    constexpr int ROWS = WV * 32, THREADS = WV * 64;
    constexpr int ASZ = ROWS * LDA, WSZ = NT ? kS32Cols * LDA : BK * kS32LdN;
    constexpr int APT = ROWS * C4 / THREADS;             // X-slab float4 per thread (4)
    constexpr int WPT = kS32Cols * BK / 4 / THREADS;      // W-slab float4 per thread (1 or 2)
    static_assert(WPT * THREADS == kS32Cols * BK / 4, "whole operand float4s per thread");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jj = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 pa[APT], pw[WPT];
    auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < APT; ++m) {
            const int f = tid + THREADS * m, row = f / C4, c4 = f % C4;
            const int rc = row < B ? row : B - 1;         // rows beyond the batch: re-read the last one, zeroed when staged
            pa[m] = *reinterpret_cast<const float4*>(X + (size_t)rc * ldx + k0 + 4 * c4);
        }
#pragma unroll
        for (int m = 0; m < WPT; ++m) {
            const int f = tid + THREADS * m;
            if (NT) {
                const int row = f / C4, c4 = f % C4;      // feature n0 + row, k0 + 4 c4
                pw[m] = *reinterpret_cast<const float4*>(W + (size_t)(n0 + row) * ldw + k0 + 4 * c4);
            } else {
                const int row = f / (kS32Cols / 4), c4 = f % (kS32Cols / 4);   // k0 + row, features n0 + 4 c4
                pw[m] = *reinterpret_cast<const float4*>(W + (size_t)(k0 + row) * ldw + n0 + 4 * c4);
            }
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        float* A = As + buf * ASZ;
        float* Wb = Ws + buf * WSZ;
#pragma unroll
        for (int m = 0; m < APT; ++m) {
            const int f = tid + THREADS * m, row = f / C4, c4 = f % C4;
            *reinterpret_cast<float4*>(A + row * LDA + 4 * c4) = row < B ? pa[m] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int m = 0; m < WPT; ++m) {
            const int f = tid + THREADS * m;
            if (NT) {
                const int row = f / C4, c4 = f % C4;
                *reinterpret_cast<float4*>(Wb + row * LDA + 4 * c4) = pw[m];
            } else {
                const int row = f / (kS32Cols / 4), c4 = f % (kS32Cols / 4);
                *reinterpret_cast<float4*>(Wb + row * kS32LdN + 4 * c4) = pw[m];
            }
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < klen; k0 += BK) {
        const bool more = k0 + BK < klen;
        if (more) fetch(k0 + BK);
        const float* A = As + buf * ASZ;
        const float* Wb = Ws + buf * WSZ;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {                // 8 k's = 4 MFMAs: half-wave kh takes k = 8 g + 4 kh + s in step s
            const float4 av = *reinterpret_cast<const float4*>(A + (wave * 32 + jj) * LDA + 8 * g + 4 * kh);
            float b[4];
            if (NT) {
                const float4 bv = *reinterpret_cast<const float4*>(Wb + jj * LDA + 8 * g + 4 * kh);
                b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) b[s] = Wb[(8 * g + 4 * kh + s) * kS32LdN + jj];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b[3], acc, 0, 0, 0);
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
}

// batch row of accumulator component r (column = n0 + (lane & 31)): the 32 x 32 accumulator layout
__device__ __forceinline__ int acc32_row(int wave, int r, int kh) { return wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh; }

// partial tile of (strip, split): [wave][4][lane] float4, lane-major, so that stores and loads are whole 1 KB rows
__device__ __forceinline__ void strip32_store(const f32x16& acc, float* part, int strip, int split, int S, int tile_floats) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4* mine = reinterpret_cast<f32x4*>(part + ((size_t)strip * S + split) * tile_floats) + wave * 4 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) mine[q * 64] = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
}
// ---- epilogue launches: one workgroup per (strip, column quarter) = 128 (64) rows x 8 columns, thread t = ((w * 4 + q) * 2
// + kh) * 8 + j holds the float4 of tile wave w, register quad q, lane kh * 32 + 8 * quarter + j: rows
// 32 w + 8 q + 4 kh + (0..3) of column 8 * quarter + j.  (One workgroup per strip — 96 of them — took 8-11 us per
// launch: too few loads in flight.)
struct Epi32 {
    int w, q, kh, j, col, row0;                           // row0: first of my four consecutive batch rows
    __device__ __forceinline__ Epi32(int strip, int quarter) {
        const int t = threadIdx.x;
        j = t & 7; kh = (t >> 3) & 1; q = (t >> 4) & 3; w = t >> 6;
        col = strip * kS32Cols + 8 * quarter + j;
        row0 = 32 * w + 8 * q + 4 * kh;
    }
};
// the S partial tiles of a strip added in split order (the loads of four tiles in flight together)
__device__ __forceinline__ f32x4 strip32_sum(const float* part, int strip, int quarter, int S, const Epi32& e) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int tile_floats = (int)(blockDim.x >> 6) * 64 * 16;
    const size_t at = (size_t)(e.w * 4 + e.q) * 64 + e.kh * 32 + 8 * quarter + e.j;
    for (int s0 = 0; s0 < S; s0 += 4) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sc = s0 + i < S ? s0 + i : S - 1;
            v[i] = reinterpret_cast<const f32x4*>(part + ((size_t)strip * S + sc) * tile_floats)[at];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (s0 + i < S) acc += v[i];
    }
    return acc;
}
// sum over the workgroup of a per-thread value that belongs to column (t & 7): every thread gets its column's total
__device__ __forceinline__ float column_total8(float v, float* red) {
    v += __shfl_xor(v, 8, 64);                            // fixed order: deterministic
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    const int t = threadIdx.x;
    __syncthreads();
    if ((t & 63) < 8) red[(t >> 6) * 8 + (t & 7)] = v;
    __syncthreads();
    const float lo = red[t & 7] + red[8 + (t & 7)];
    return blockDim.x > 128 ? lo + (red[16 + (t & 7)] + red[24 + (t & 7)]) : lo;      // four waves or two
}

struct RhPartArgs { const float* X; const float* W; float* part; int B, D, S; };

// partial products of a strip: blockIdx.y = slice of the contraction
template <bool NT, int WV>
__global__ __launch_bounds__(WV * 64) void rh_part32_kernel(RhPartArgs a) {
    extern __shared__ __attribute__((aligned(16))) float rh_smem[];
    float* As = rh_smem;                                                 // [2][32 WV][LDA]
    float* Ws = As + 2 * WV * 32 * kS32Lda;                              // [2][32][LDA] / [2][BK][LdN]
    const int strip = blockIdx.x, split = blockIdx.y, n0 = strip * kS32Cols;
    const int klen = a.D / a.S, kb = split * klen;
    f32x16 acc;
    strip32_gemm<NT, WV>(a.X + kb, a.B, klen, a.D, NT ? a.W + kb : a.W + (size_t)kb * a.D, a.D, n0, acc, As, Ws);
    strip32_store(acc, a.part, strip, split, a.S, tile32_floats(WV));
}

// P = sum of the partial tiles; BatchNorm1d over the batch (cifar_2version.py:201, 214-215), activation (:216)
__global__ __launch_bounds__(256) void rh_fwd32_epi_kernel(RhFwdArgs a, const float* part, int S) {
    __shared__ float red[32];
    const Epi32 e(blockIdx.x, blockIdx.y);
    f32x4 acc = strip32_sum(part, blockIdx.x, blockIdx.y, S, e);
    const int col = e.col;
    float mu, istd;
    if (a.training) {
        mu = column_total8((acc[0] + acc[1]) + (acc[2] + acc[3]), red) / (float)a.B;      // padded rows hold exact zeros
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dlt = acc[r] - mu;
            if (e.row0 + r < a.B) q = fmaf(dlt, dlt, q);
        }
        const float var = column_total8(q, red) / (float)a.B;              // biased, as BatchNorm normalises
        istd = 1.0f / sqrtf(var + a.eps);
        if (threadIdx.x < 8 && a.run_mean != nullptr) {                   // running statistics (unbiased variance)
            const float unb = a.B > 1 ? var * (float)a.B / (float)(a.B - 1) : var;
            a.run_mean[col] = (1.f - a.momentum) * a.run_mean[col] + a.momentum * mu;
            a.run_var[col] = (1.f - a.momentum) * a.run_var[col] + a.momentum * unb;
        }
    } else {
        mu = a.run_mean[col];
        istd = 1.0f / sqrtf(a.run_var[col] + a.eps);
    }
    if (threadIdx.x < 8) { a.mean[col] = mu; a.invstd[col] = istd; }
    const float g = a.gamma[col], bt = a.beta[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = e.row0 + r;
        if (row < a.B) {
            const float p = acc[r];
            a.P[(size_t)row * a.D + col] = p;
            a.H[(size_t)row * a.D + col] = act_fwd(fmaf((p - mu) * istd, g, bt), a.act);
        }
    }
}

// out = base + scale * (sum of the partial tiles)
__global__ __launch_bounds__(256) void rh_axpy32_epi_kernel(RhAxpyArgs a, const float* part, int S) {
    const Epi32 e(blockIdx.x, blockIdx.y);
    const f32x4 acc = strip32_sum(part, blockIdx.x, blockIdx.y, S, e);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = e.row0 + r;
        if (row < a.B) {
            const size_t o = (size_t)row * a.D + e.col;
            a.out[o] = a.base != nullptr ? fmaf(a.scale, acc[r], a.base[o]) : a.scale * acc[r];
        }
    }
}

// dH = scale * (sum of the partial tiles of G K^T); through the activation and the BatchNorm -> dP, dgamma, dbeta
__global__ __launch_bounds__(256) void rh_bwd32_epi_kernel(RhBwdArgs a, const float* part, int S) {
    __shared__ float red[32];
    const Epi32 e(blockIdx.x, blockIdx.y);
    f32x4 acc = strip32_sum(part, blockIdx.x, blockIdx.y, S, e);
    const int col = e.col;
    const float mu = a.mean[col], istd = a.invstd[col], g = a.gamma[col];
    float sb = 0.f, sg = 0.f;
    f32x4 xh;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = e.row0 + r;
        float dhn = 0.f, xhat = 0.f;
        if (row < a.B) {
            const size_t o = (size_t)row * a.D + col;
            dhn = a.scale * acc[r] * act_bwd(a.H[o], a.act);
            xhat = (a.P[o] - mu) * istd;
        }
        acc[r] = dhn;
        xh[r] = xhat;
        sb += dhn;
        sg = fmaf(dhn, xhat, sg);
    }
    const float dbeta = column_total8(sb, red);
    const float dgamma = column_total8(sg, red);
    if (threadIdx.x < 8) { a.g_beta[col] = dbeta; a.g_gamma[col] = dgamma; }
    const float inv_b = 1.0f / (float)a.B;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = e.row0 + r;
        if (row < a.B) {
            const float dhn = acc[r];
            a.dP[(size_t)row * a.D + col] = a.training ? g * istd * (dhn - (dbeta + xh[r] * dgamma) * inv_b) : g * istd * dhn;
        }
    }
}

struct RhOuterArgs {
    const float* A1; const float* B1; const float* A2; const float* B2;
    float* out;
    int B, D;
    float s2;
};

// out[i][j] = sum_b A1[b][i] B1[b][j] + s2 * sum_b A2[b][i] B2[b][j]: the gradient of K from both of its uses
// (dP^T X from the first product, H^T dF from the second).  64 x 192 tile per workgroup, wave (wi, wj) a 32 x 96 part
// (three 32 x 32 accumulators): D = 3072 is 48 x 16 = 768 tiles, exactly three per CU — with 128 x 128 tiles (576) a
// quarter of the CUs carried three workgroups and the others two, and the launch took as long as the former (77 us).
constexpr int kOutTi = 64, kOutTj = 192;
__global__ __launch_bounds__(kRhThreads) void rh_outer_kernel(RhOuterArgs a) {
    constexpr int BKB = 16, LDA_ = kOutTi + 4, LDB_ = kOutTj + 4;
    __shared__ __attribute__((aligned(16))) float Sa[2][BKB * LDA_];      // double-buffered: one barrier per slab
    __shared__ __attribute__((aligned(16))) float Sb[2][BKB * LDB_];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jj = lane & 31, kh = lane >> 5;
    const int i0 = blockIdx.y * kOutTi, j0 = blockIdx.x * kOutTj;
    const int wi = wave >> 1, wj = wave & 1;
    f32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const int Bp = (a.B + BKB - 1) / BKB * BKB;         // (the two products are walked as 2 Bp virtual batch rows)
    // a slab is BKB batch rows of [64 | 192] floats = 64 float4 per row: four float4 per thread
    constexpr int PT = BKB * 64 / kRhThreads;
    float4 pv[PT];
    auto fetch = [&](int v) __attribute__((always_inline)) {          // v: virtual batch row of the slab's first row
        const bool second = v >= Bp;
        const float* A = second ? a.A2 : a.A1;
        const float* Bm = second ? a.B2 : a.B1;
#pragma unroll
        for (int m = 0; m < PT; ++m) {
            const int f = tid + kRhThreads * m, row = f >> 6, c4 = f & 63;
            const int b = (second ? v - Bp : v) + row;
            pv[m] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < a.B) {                                // (D is a multiple of 64: a float4 never straddles the edge)
                if (c4 < kOutTi / 4) {
                    if (i0 + 4 * c4 < a.D) pv[m] = *reinterpret_cast<const float4*>(A + (size_t)b * a.D + i0 + 4 * c4);
                } else if (j0 + 4 * (c4 - kOutTi / 4) < a.D) {
                    pv[m] = *reinterpret_cast<const float4*>(Bm + (size_t)b * a.D + j0 + 4 * (c4 - kOutTi / 4));
                    if (second) { pv[m].x *= a.s2; pv[m].y *= a.s2; pv[m].z *= a.s2; pv[m].w *= a.s2; }
                }
            }
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < PT; ++m) {
            const int f = tid + kRhThreads * m, row = f >> 6, c4 = f & 63;
            if (c4 < kOutTi / 4) *reinterpret_cast<float4*>(Sa[buf] + row * LDA_ + 4 * c4) = pv[m];
            else *reinterpret_cast<float4*>(Sb[buf] + row * LDB_ + 4 * (c4 - kOutTi / 4)) = pv[m];
        }
    };
    const int total = 2 * Bp;
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int v = 0; v < total; v += BKB) {
        const bool more = v + BKB < total;
        if (more) fetch(v + BKB);
#pragma unroll
        for (int s = 0; s < BKB / 2; ++s) {
            const float af = Sa[buf][(2 * s + kh) * LDA_ + wi * 32 + jj];
            float bf[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) bf[q] = Sb[buf][(2 * s + kh) * LDB_ + wj * 96 + q * 32 + jj];
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf[q], acc[q], 0, 0, 0);
        }
        if (more) stage(buf ^ 1);                         // the other buffer: last read one barrier ago
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const int j = j0 + wj * 96 + q * 32 + jj;
            if (i < a.D && j < a.D) a.out[(size_t)i * a.D + j] = acc[q][r];
        }
}

// ---- the same gradient with every fp32 operand as three bf16 pieces on the bf16 matrix cores -----------------------------
// (see pde_mix_bf16.hip: x = hi + mid + lo, the six piece products of order <= 2, each exact in the fp32 accumulator;
// v_mfma_f32_32x32x16_bf16 does 16 contraction steps in 32 cycles where v_mfma_f32_32x32x2_f32 does 2 in 64.)
// A slab is 16 batch rows — ONE contraction group — of the [64 | 192] features of the tile, split when it is deposited in
// LDS as bf16 images [piece][batch row][feature]; both operands have the contraction index (batch) as the image's ROW, so both
// come out through the transposing read ds_read_b64_tr_b16 (operand lane (h, r): feature r of its group of 32, batch rows
// 8h .. 8h+7).  Double-buffered: one barrier per slab.
typedef short rh_v4s __attribute__((ext_vector_type(4)));
typedef short rh_v8s __attribute__((ext_vector_type(8)));
typedef __bf16 rh_v8bf __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned short rh_f2bf(float f) { return f32_to_bf16_hw(f); }
__device__ __forceinline__ float rh_bf2f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }
// operand with k = the image's rows 0..15, the other index = features 32 fg .. 32 fg + 31 (PITCH bytes per row)
template <int PITCH>
__device__ __forceinline__ rh_v8bf rh_tr_operand(const unsigned char* img, int fg, int lane) {
    const int grp = lane >> 4, i = lane & 15, h = lane >> 5;
    const int q = i >> 2, p = i & 3;
    const unsigned char* a0 = img + (8 * h + q) * PITCH + (32 * fg + 16 * (grp & 1) + 4 * p) * 2;
    const rh_v4s lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rh_v4s __attribute__((address_space(3)))*)(a0));
    const rh_v4s hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rh_v4s __attribute__((address_space(3)))*)(a0 + 4 * PITCH));
    const rh_v8s all = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(rh_v8bf, all);
}

// TI x TJ tile per workgroup, wave (wi, wj) a (TI/2) x (TJ/2) part = NI x NJ accumulators of 32 x 32
template <int TI, int TJ>
__global__ __launch_bounds__(kRhThreads) void rh_outer_split_kernel(RhOuterArgs a) {
    constexpr int BKB = 16, NI = TI / 64, NJ = TJ / 64;
    constexpr int PA = TI * 2 + 16, PB = TJ * 2 + 16;      // bytes per image row (16 bytes of padding)
    constexpr int IA = 16 * PA, IB = 16 * PB;              // one piece, 16 batch rows
    constexpr int BUF = 3 * (IA + IB);                     // one slab
    constexpr int F4 = (TI + TJ) / 4;                      // float4 per batch row of the slab
    constexpr int PT = BKB * F4 / kRhThreads;
    static_assert(BKB * F4 % kRhThreads == 0 && TI % 64 == 0 && TJ % 64 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char rh_spl[];   // [2][BUF]
    unsigned char* S = rh_spl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jj = lane & 31, kh = lane >> 5;
    const int i0 = blockIdx.y * TI, j0 = blockIdx.x * TJ;
    const int wi = wave >> 1, wj = wave & 1;
    f32x16 acc[NI][NJ];
#pragma unroll
    for (int p = 0; p < NI; ++p)
#pragma unroll
        for (int q = 0; q < NJ; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.f;
    const int Bp = (a.B + BKB - 1) / BKB * BKB;
    // (two slabs ahead of the matrix cores instead of one was tried: 84 us against 57 at 64 x 192)
    float4 pv[PT];
    auto fetch = [&](int v) __attribute__((always_inline)) {
        const bool second = v >= Bp;
        const float* A = second ? a.A2 : a.A1;
        const float* Bm = second ? a.B2 : a.B1;
#pragma unroll
        for (int m = 0; m < PT; ++m) {
            const int f = tid + kRhThreads * m, row = f / F4, c4 = f % F4;
            const int b = (second ? v - Bp : v) + row;
            pv[m] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < a.B) {                                // (D is a multiple of 64: a float4 never straddles the edge)
                if (c4 < TI / 4) {
                    if (i0 + 4 * c4 < a.D) pv[m] = *reinterpret_cast<const float4*>(A + (size_t)b * a.D + i0 + 4 * c4);
                } else if (j0 + 4 * (c4 - TI / 4) < a.D) {
                    pv[m] = *reinterpret_cast<const float4*>(Bm + (size_t)b * a.D + j0 + 4 * (c4 - TI / 4));
                    if (second) { pv[m].x *= a.s2; pv[m].y *= a.s2; pv[m].z *= a.s2; pv[m].w *= a.s2; }
                }
            }
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        unsigned char* base = S + buf * BUF;
#pragma unroll
        for (int m = 0; m < PT; ++m) {
            const int f = tid + kRhThreads * m, row = f / F4, c4 = f % F4;
            const float v[4] = {pv[m].x, pv[m].y, pv[m].z, pv[m].w};
            unsigned short pc[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pc[0][j] = rh_f2bf(v[j]);
                const float r1 = v[j] - rh_bf2f(pc[0][j]);
                pc[1][j] = rh_f2bf(r1);
                pc[2][j] = rh_f2bf(r1 - rh_bf2f(pc[1][j]));
            }
            const bool isA = c4 < TI / 4;
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                unsigned char* dst = isA ? base + sp * IA + row * PA + 8 * c4
                                         : base + 3 * IA + sp * IB + row * PB + 8 * (c4 - TI / 4);
                *reinterpret_cast<uint2*>(dst) = make_uint2(pc[sp][0] | (pc[sp][1] << 16), pc[sp][2] | (pc[sp][3] << 16));
            }
        }
    };
    const int total = 2 * Bp;
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int v = 0; v < total; v += BKB) {
        const bool more = v + BKB < total;
        if (more) fetch(v + BKB);
        const unsigned char* ia = S + buf * BUF;
        const unsigned char* ib = ia + 3 * IA;
        rh_v8bf af[NI][3];
#pragma unroll
        for (int p = 0; p < NI; ++p)
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) af[p][sp] = rh_tr_operand<PA>(ia + sp * IA, wi * NI + p, lane);
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            rh_v8bf bf[3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) bf[sp] = rh_tr_operand<PB>(ib + sp * IB, wj * NJ + q, lane);
#pragma unroll
            for (int p = 0; p < NI; ++p) {
                acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[p][1], bf[1], acc[p][q], 0, 0, 0);   // smallest terms first
                acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[p][0], bf[2], acc[p][q], 0, 0, 0);
                acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[p][2], bf[0], acc[p][q], 0, 0, 0);
                acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[p][0], bf[1], acc[p][q], 0, 0, 0);
                acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[p][1], bf[0], acc[p][q], 0, 0, 0);
                acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[p][0], bf[0], acc[p][q], 0, 0, 0);
            }
        }
        if (more) stage(buf ^ 1);                         // the other buffer: last read one barrier ago
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int p = 0; p < NI; ++p)
#pragma unroll
        for (int q = 0; q < NJ; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + (wi * NI + p) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int j = j0 + (wj * NJ + q) * 32 + jj;
                if (i < a.D && j < a.D) a.out[(size_t)i * a.D + j] = acc[p][q][r];
            }
}

template <int TI, int TJ>
int launch_outer_split(const RhOuterArgs& o, int D, hipStream_t st) {
    constexpr int lds = 2 * 3 * 16 * ((TI * 2 + 16) + (TJ * 2 + 16));
    static unsigned long long configured = 0;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(rh_outer_split_kernel<TI, TJ>), lds, configured) != PDE_OK) return PDE_E_LAUNCH;
    hipLaunchKernelGGL((rh_outer_split_kernel<TI, TJ>), dim3((D + TJ - 1) / TJ, (D + TI - 1) / TI), dim3(kRhThreads), lds, st, o);
    return check_launch();
}

constexpr size_t strip_lds() {
    // X slabs + the larger of the two W-slab shapes, both double-buffered, + the reduction scratch
    return (size_t)(2 * kRhWaves * 16 * RhSlab<1>::LDA + 2 * (kRhCols * RhSlab<1>::LDA > RhSlab<1>::BK * kRhLdN
                    ? kRhCols * RhSlab<1>::LDA : RhSlab<1>::BK * kRhLdN) + kRhWaves * 64) * sizeof(float);
}
constexpr int kRhRows = kRhWaves * 16;                    // batch rows of one strip workgroup (128)
// The "dynamic LDS limit set on device d" bits are kept per KERNEL ADDRESS: two kernels with the same signature
// (rh_axpy_strip_kernel<1> and rh_nt_strip_kernel<1> both take RhAxpyArgs) are the same template instantiation of this
// function, so a function-local static would be shared and only the first of them would ever get its attribute.
template <typename ARGS, typename KERN>
int launch_strip(KERN kern, const ARGS& a, int D, int row_blocks, hipStream_t st) {
    static std::mutex mu;
    static std::map<const void*, unsigned long long> done;
    unsigned long long* bits;
    {
        std::lock_guard<std::mutex> lk(mu);
        bits = &done[reinterpret_cast<const void*>(kern)];    // (std::map nodes do not move)
    }
    unsigned long long& configured = *bits;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)strip_lds(), configured) != PDE_OK) return PDE_E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(D / kRhCols, row_blocks), dim3(kRhStripThreads), strip_lds(), st, a);
    return check_launch();
}

bool rh_dims_ok(int B, int D) { return B >= 1 && D >= 64 && (D % 64) == 0; }

// Waves of a 32-column strip workgroup for this batch, and the split of a strip's contraction: a power of two (2..8),
// slices of whole 32-wide slabs and at least two of them, three workgroups per CU at D = 3072;
// 0 = this batch / width keeps the 16-column kernels
int rh_waves32(int B) { return B <= 64 ? 2 : 4; }
int rh_split32(int B, int D) {
    static const bool off = getenv("PDE_RH_NO_STRIP32") != nullptr;
    if (off || B > kRhRows32) return 0;
    const int wv = rh_waves32(B);
    int S = 8;                                           // (two waves: 16 slices measured slower, 171 us against 158 per layer)
    static const int forced = getenv("PDE_RH_SPLIT") ? atoi(getenv("PDE_RH_SPLIT")) : 0;     // diagnostics: the starting split
    if (forced >= 2 && forced <= 16 && (forced & (forced - 1)) == 0) S = forced;
    while (S > 1 && (D % (S * 2 * kS32Bk) != 0 || (D / kS32Cols) * S > (wv == 2 ? 2048 : 1024))) S >>= 1;
    return S >= 2 ? S : 0;
}
size_t rh_split32_bytes(int B, int D, int S) {
    return S < 2 ? 0 : (size_t)(D / kS32Cols) * S * tile32_floats(rh_waves32(B)) * sizeof(float);
}
constexpr size_t strip32_lds(int wv) {
    return (size_t)(2 * wv * 32 * kS32Lda + 2 * (kS32Cols * kS32Lda > kS32Bk * kS32LdN ? kS32Cols * kS32Lda : kS32Bk * kS32LdN))
           * sizeof(float);
}
// partial products X W^T (NT) or X W (NN) of every strip into `ws`
template <bool NT, int WV>
int launch_part32_wv(const float* X, const float* W, int B, int D, int S, void* ws, hipStream_t st) {
    static unsigned long long configured = 0;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(rh_part32_kernel<NT, WV>), (int)strip32_lds(WV), configured) != PDE_OK)
        return PDE_E_LAUNCH;
    RhPartArgs a{X, W, static_cast<float*>(ws), B, D, S};
    hipLaunchKernelGGL((rh_part32_kernel<NT, WV>), dim3(D / kS32Cols, S), dim3(WV * 64), strip32_lds(WV), st, a);
    return check_launch();
}
template <bool NT>
int launch_part32(const float* X, const float* W, int B, int D, int S, void* ws, hipStream_t st) {
    return rh_waves32(B) == 2 ? launch_part32_wv<NT, 2>(X, W, B, D, S, ws, st) : launch_part32_wv<NT, 4>(X, W, B, D, S, ws, st);
}

}  // namespace
}  // namespace pde

using namespace pde;

extern "C" {

int pde_sym_layer_supported(int32_t B, int32_t D) { return rh_dims_ok(B, D) ? 1 : 0; }

size_t pde_sym_layer_workspace_bytes(int32_t B, int32_t D) {
    if (!rh_dims_ok(B, D)) return 0;
    return rh_split32_bytes(B, D, rh_split32(B, D));
}

int pde_sym_layer_forward(int32_t B, int32_t D, int32_t act, int32_t training, const float* X, const float* K,
                          const float* bn_weight, const float* bn_bias, float* running_mean, float* running_var,
                          float momentum, float eps, const float* base, float scale, float* P, float* H, float* mean,
                          float* invstd, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!rh_dims_ok(B, D)) return PDE_E_BADARG;
    if (!X || !K || !bn_weight || !bn_bias || !P || !H || !mean || !invstd || !out) return PDE_E_BADARG;
    if (act < kActIdentity || act > kActTanh) return PDE_E_BADARG;
    if (!training && (!running_mean || !running_var)) return PDE_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    RhFwdArgs f{X, K, bn_weight, bn_bias, running_mean, running_var, P, H, mean, invstd, B, D, act, training ? 1 : 0, momentum, eps};
    const int nblk = (B + kRhRows - 1) / kRhRows;
    int rc;
    const int S = workspace ? rh_split32(B, D) : 0;       // no workspace: the 16-column kernels (one workgroup per strip)
    if (S >= 2) {
        if (workspace_bytes < rh_split32_bytes(B, D, S) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return PDE_E_WORKSPACE;
        const float* part = static_cast<const float*>(workspace);
        rc = launch_part32<true>(X, K, B, D, S, workspace, st);
        if (rc != PDE_OK) return rc;
        hipLaunchKernelGGL(rh_fwd32_epi_kernel, dim3(D / kS32Cols, 4), dim3(64 * rh_waves32(B)), 0, st, f, part, S);
        rc = launch_part32<false>(H, K, B, D, S, workspace, st);
        if (rc != PDE_OK) return rc;
        RhAxpyArgs x{H, K, base, out, B, D, scale};
        hipLaunchKernelGGL(rh_axpy32_epi_kernel, dim3(D / kS32Cols, 4), dim3(64 * rh_waves32(B)), 0, st, x, part, S);
        return check_launch();
    }
    if (nblk == 1) {                                      // the whole batch in one strip workgroup: statistics as epilogue
        rc = launch_strip(rh_fwd_strip_kernel<1>, f, D, 1, st);
    } else {                                              // product by row blocks, then the statistics over all of them
        RhAxpyArgs p{X, K, nullptr, P, B, D, 1.0f};
        rc = launch_strip(rh_nt_strip_kernel<1>, p, D, nblk, st);
        if (rc != PDE_OK) return rc;
        hipLaunchKernelGGL(rh_bn_fwd_kernel, dim3(D / kBnCols), dim3(256), 0, st, f);
        rc = check_launch();
    }
    if (rc != PDE_OK) return rc;
    RhAxpyArgs x{H, K, base, out, B, D, scale};
    return launch_strip(rh_axpy_strip_kernel<1>, x, D, nblk, st);
}

int pde_sym_layer_backward(int32_t B, int32_t D, int32_t act, int32_t training, const float* g_out, float scale,
                           const float* X, const float* K, const float* bn_weight, const float* P, const float* H,
                           const float* mean, const float* invstd, float* dP, float* gX, float* gK,
                           float* g_bn_weight, float* g_bn_bias, void* workspace, size_t workspace_bytes, void* stream) {
    if (!rh_dims_ok(B, D)) return PDE_E_BADARG;
    if (!g_out || !X || !K || !bn_weight || !P || !H || !mean || !invstd || !dP || !gX || !gK || !g_bn_weight || !g_bn_bias)
        return PDE_E_BADARG;
    if (act < kActIdentity || act > kActTanh) return PDE_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    RhBwdArgs b{g_out, K, bn_weight, P, H, mean, invstd, dP, g_bn_weight, g_bn_bias, B, D, act, training ? 1 : 0, scale};
    const int nblk = (B + kRhRows - 1) / kRhRows;
    int rc;
    const int S = workspace ? rh_split32(B, D) : 0;
    if (S >= 2) {
        if (workspace_bytes < rh_split32_bytes(B, D, S) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return PDE_E_WORKSPACE;
        const float* part = static_cast<const float*>(workspace);
        rc = launch_part32<true>(g_out, K, B, D, S, workspace, st);
        if (rc != PDE_OK) return rc;
        hipLaunchKernelGGL(rh_bwd32_epi_kernel, dim3(D / kS32Cols, 4), dim3(64 * rh_waves32(B)), 0, st, b, part, S);
        rc = launch_part32<false>(dP, K, B, D, S, workspace, st);
        if (rc != PDE_OK) return rc;
        RhAxpyArgs x32{dP, K, nullptr, gX, B, D, 1.0f};
        hipLaunchKernelGGL(rh_axpy32_epi_kernel, dim3(D / kS32Cols, 4), dim3(64 * rh_waves32(B)), 0, st, x32, part, S);
        rc = check_launch();
    } else if (nblk == 1) {
        rc = launch_strip(rh_bwd_strip_kernel<1>, b, D, 1, st);
    } else {
        RhAxpyArgs p{g_out, K, nullptr, dP, B, D, 1.0f};
        rc = launch_strip(rh_nt_strip_kernel<1>, p, D, nblk, st);
        if (rc != PDE_OK) return rc;
        hipLaunchKernelGGL(rh_bn_bwd_kernel, dim3(D / kBnCols), dim3(256), 0, st, b);
        rc = check_launch();
    }
    if (rc != PDE_OK) return rc;
    if (S < 2) {
        RhAxpyArgs x{dP, K, nullptr, gX, B, D, 1.0f};
        rc = launch_strip(rh_axpy_strip_kernel<1>, x, D, nblk, st);
        if (rc != PDE_OK) return rc;
    }
    RhOuterArgs o{dP, X, H, g_out, gK, B, D, scale};
    // three-piece bf16 products on the same 64 x 192 tiles (57 us against 66 at B = 128, D = 3072; 128 x 192 and 128 x 128
    // tiles: 66 and 63 us); PDE_RH_NO_SPLIT=1 selects the fp32-MFMA kernel
    static const bool split = getenv("PDE_RH_NO_SPLIT") == nullptr;
    if (split) return launch_outer_split<kOutTi, kOutTj>(o, D, st);
    hipLaunchKernelGGL(rh_outer_kernel, dim3((D + kOutTj - 1) / kOutTj, (D + kOutTi - 1) / kOutTi), dim3(kRhThreads), 0, st, o);
    return check_launch();
}

}  // extern "C"
