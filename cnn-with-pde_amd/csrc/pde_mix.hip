// Channel operators of the PDE layers:
//   out[b,i,p] = sum_j M[i,j] u[b,j,p]
// cifar10.py:65-72 apply_channel_mixing (M @ u_flat) and SVHN.py:78-86
// apply_channel_coupling (u_perm @ K^T) are both this product.
//
// Forward / input-gradient: one thread per pixel (4 pixels for fp32 I/O would be
// the next step), 8 output channels per pass, the matrix row read with scalar loads
// (it is wave-uniform), the pixel's channel column streamed from L1/L2.
// Matrix gradient gM = G U^T (a C x C product with K = B*HW): LDS-tiled 32x32 output
// tiles, split over K across workgroups, partial tiles reduced in a fixed order.
#include "pde_common.h"

#include <cstdlib>

namespace pde {
namespace {

// dynamic-LDS limit: a per-device attribute of a kernel, set once per (kernel, device)
inline void ensure_lds(const void* kernel, int bytes, unsigned long long& done) { (void)ensure_dynamic_lds(kernel, bytes, done); }

template <typename IO> struct Io;
template <> struct Io<float> {
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
struct bf16s { unsigned short v; };
template <> struct Io<bf16s> {
    __device__ static __forceinline__ float ld(const bf16s* p) { return __uint_as_float((unsigned int)p->v << 16); }
    __device__ static __forceinline__ void st(bf16s* p, float f) { p->v = f32_to_bf16_hw(f); }
};

constexpr int kOC = 8;   // output channels per pass

// out[b,i,p] = sum_j W(i,j) u[b,j,p];  W(i,j) = TRANS ? M[j,i] : M[i,j]
template <typename IO, bool TRANS>
__global__ __launch_bounds__(256) void mix_apply_kernel(const IO* __restrict__ u, const float* __restrict__ M,
                                                        IO* __restrict__ out, int C, int HW) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const IO* ub = u + (size_t)b * C * HW + p;
    IO* ob = out + (size_t)b * C * HW + p;
    for (int i0 = 0; i0 < C; i0 += kOC) {
        float acc[kOC];
#pragma unroll
        for (int r = 0; r < kOC; ++r) acc[r] = 0.f;
        for (int j = 0; j < C; ++j) {
            const float uj = Io<IO>::ld(ub + (size_t)j * HW);
#pragma unroll
            for (int r = 0; r < kOC; ++r) {
                const int i = i0 + r;
                if (i < C) acc[r] = fmaf(TRANS ? M[j * C + i] : M[i * C + j], uj, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < kOC; ++r)
            if (i0 + r < C) Io<IO>::st(ob + (size_t)(i0 + r) * HW, acc[r]);
    }
}

// gM tile: part[split][i][j] = sum over this split's (b,p) of g[b,i,p] u[b,j,p]
constexpr int kT = 32;    // output tile edge
constexpr int kK = 64;    // pixels per LDS chunk
template <typename IO>
__global__ __launch_bounds__(256) void mix_gm_kernel(const IO* __restrict__ u, const IO* __restrict__ g,
                                                     float* __restrict__ part, int B, int C, int HW, int nsplit, int acc) {
    __shared__ float sg[kK][kT + 1];
    __shared__ float su[kK][kT + 1];
    const int tiles = (C + kT - 1) / kT;
    const int ti = blockIdx.x / tiles, tj = blockIdx.x % tiles;
    const int split = blockIdx.y;
    const int tid = threadIdx.x;
    const int oi = (tid / 16) * 2, oj = (tid % 16) * 2;      // 2x2 outputs per thread
    // double accumulators: a thread adds thousands of products here, and at small C the whole sum is a few numbers that
    // may cancel (the 1 x 1 coupling gradient of a C = 1 layer was 3.6e-5 from the fp64 value where torch's pairwise fp32
    // sum is at 4e-7; a seeded walk of round 4 drew it); the products of two floats are exact in double
    double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
    const int chunks_per_b = (HW + kK - 1) / kK;
    const long total = (long)B * chunks_per_b;
    for (long ch = split; ch < total; ch += nsplit) {
        const int b = (int)(ch / chunks_per_b);
        const int p0 = (int)(ch % chunks_per_b) * kK;
        // load [32 channels][64 pixels] of g (rows ti*32..) and u (rows tj*32..), transposed into [k][c]
        for (int e = tid; e < kT * kK; e += 256) {
            const int c = e / kK, k = e % kK;
            const int p = p0 + k;
            const int ci = ti * kT + c, cj = tj * kT + c;
            sg[k][c] = (ci < C && p < HW) ? Io<IO>::ld(g + ((size_t)b * C + ci) * HW + p) : 0.f;
            su[k][c] = (cj < C && p < HW) ? Io<IO>::ld(u + ((size_t)b * C + cj) * HW + p) : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < kK; ++k) {
            const float g0 = sg[k][oi], g1 = sg[k][oi + 1];
            const float u0 = su[k][oj], u1 = su[k][oj + 1];
            a00 += (double)g0 * (double)u0; a01 += (double)g0 * (double)u1;
            a10 += (double)g1 * (double)u0; a11 += (double)g1 * (double)u1;
        }
        __syncthreads();
    }
    float* dst = part + (size_t)split * C * C;
    const int i = ti * kT + oi, j = tj * kT + oj;
    auto put = [&](int ii, int jj2, float v) { if (ii < C && jj2 < C) dst[ii * C + jj2] = acc ? dst[ii * C + jj2] + v : v; };
    put(i, j, (float)a00); put(i, j + 1, (float)a01); put(i + 1, j, (float)a10); put(i + 1, j + 1, (float)a11);
}

__global__ __launch_bounds__(256) void mix_gm_reduce_kernel(const float* __restrict__ part, float* __restrict__ gM,
                                                            int CC, int nsplit) {
    // 32 outputs per workgroup; thread (q, e) adds splits q, q+8, ... of output e (16 loads in flight),
    // the eight partial sums are combined in a fixed order (bitwise reproducible)
    __shared__ float sh[8][32];
    const int el = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    float s = 0.f;
    if (e < CC) {
        int k = q;
        for (; k + 8 * 15 < nsplit; k += 8 * 16) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = part[(size_t)(k + 8 * i) * CC + e];
#pragma unroll
            for (int i = 0; i < 16; ++i) s += v[i];
        }
        for (; k < nsplit; k += 8) s += part[(size_t)k * CC + e];
    }
    sh[q][el] = s;
    __syncthreads();
    if (q == 0 && e < CC)
        gM[e] = ((sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el])) + ((sh[4][el] + sh[5][el]) + (sh[6][el] + sh[7][el]));
}

// ---- fp32 MFMA path (C a multiple of 32) -------------------------------------------------
// out = W u per pixel is a (C x C) x (C x pixels) product: v_mfma_f32_32x32x2_f32 runs it at the
// fp32 vector rate with exact fp32 FMA chains and leaves the VALU free, so the kernel is bound by
// streaming u in and out once (8 B/element).  Operand maps (cdna_hip_programming.md §3):
//   A: lane l holds A[i = l&31][k = l>>5]      B: lane l holds B[k = l>>5][j = l&31]
//   D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
// Each wave takes 128 consecutive pixels of one sample: lane (j, kh) loads 4 consecutive pixels
// (one 16-byte load) of channel 2*ks+kh — component q is the B operand of sub-strip q — so the four
// D values of a lane are 4 consecutive pixels again and go out as one 16-byte store.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename IO> struct Io4;
template <> struct Io4<float> {
    __device__ static __forceinline__ float4 ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static __forceinline__ void st(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct Io4<bf16s> {
    __device__ static __forceinline__ float4 ld(const bf16s* p) {
        const ushort4 q = *reinterpret_cast<const ushort4*>(p);
        return make_float4(__uint_as_float((unsigned)q.x << 16), __uint_as_float((unsigned)q.y << 16),
                           __uint_as_float((unsigned)q.z << 16), __uint_as_float((unsigned)q.w << 16));
    }
    __device__ static __forceinline__ void st(bf16s* p, float4 v) {
        bf16s t[4];
        Io<bf16s>::st(&t[0], v.x); Io<bf16s>::st(&t[1], v.y); Io<bf16s>::st(&t[2], v.z); Io<bf16s>::st(&t[3], v.w);
        *reinterpret_cast<ushort4*>(p) = make_ushort4(t[0].v, t[1].v, t[2].v, t[3].v);
    }
};

constexpr int kMixTG = 2;      // output-channel tiles (of 32) accumulated per pass over the input

template <typename IO>
__global__ __launch_bounds__(256, 2) void mix_apply_mfma_kernel(const IO* __restrict__ u, const float* __restrict__ M,
                                                             IO* __restrict__ out, int B, int C, int HW, int trans) {
    extern __shared__ float wfrag[];                    // [C/32 tiles][C/2 k-steps][64 lanes]: A fragments of W
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = C / 32, KS = C / 2;
    for (int e = tid; e < C * C; e += 256) {
        const int ln = e & 63, ks = (e >> 6) % KS, it = (e >> 6) / KS;
        const int i = 32 * it + (ln & 31), k = 2 * ks + (ln >> 5);
        wfrag[e] = trans ? M[k * C + i] : M[i * C + k];
    }
    __syncthreads();
    const int per_sample = (HW + 127) / 128;
    const long nblk = (long)B * per_sample;
    const int kh = lane >> 5, jj = lane & 31;
    for (long blk = (long)blockIdx.x * 4 + wave; blk < nblk; blk += (long)gridDim.x * 4) {
        const int b = (int)(blk / per_sample);
        const int px = (int)(blk % per_sample) * 128 + 4 * jj;
        const bool pv = px < HW;                        // HW is a multiple of 4: all four pixels or none
        const IO* ub = u + (size_t)b * C * HW + px;
        IO* ob = out + (size_t)b * C * HW + px;
        for (int t0 = 0; t0 < T; t0 += kMixTG) {
            f32x16 acc[kMixTG][4];
#pragma unroll
            for (int t = 0; t < kMixTG; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][q][r] = 0.f;
            // input loads run one batch of 4 k-steps ahead of the MFMAs (two register sets)
            auto fetch = [&](float4 (&x)[4], int ks0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    x[i] = (pv && ks0 + i < KS) ? Io4<IO>::ld(ub + (size_t)(2 * (ks0 + i) + kh) * HW)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
            };
            auto consume = [&](const float4 (&x)[4], int ks0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int t = 0; t < kMixTG; ++t) {
                        if (t0 + t < T) {
                            const float av = wfrag[((t0 + t) * KS + ks0 + i) * 64 + lane];
                            acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, x[i].x, acc[t][0], 0, 0, 0);
                            acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, x[i].y, acc[t][1], 0, 0, 0);
                            acc[t][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, x[i].z, acc[t][2], 0, 0, 0);
                            acc[t][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, x[i].w, acc[t][3], 0, 0, 0);
                        }
                    }
                }
            };
            float4 xa[4], xb[4];
            fetch(xa, 0);
            for (int ks0 = 0; ks0 < KS; ks0 += 8) {          // KS = C/2 is a multiple of 16
                fetch(xb, ks0 + 4);
                consume(xa, ks0);
                fetch(xa, ks0 + 8);
                consume(xb, ks0 + 4);
            }
            if (pv) {
#pragma unroll
                for (int t = 0; t < kMixTG; ++t) {
                    if (t0 + t < T) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ch = 32 * (t0 + t) + (r & 3) + 8 * (r >> 2) + 4 * kh;
                            Io4<IO>::st(ob + (size_t)ch * HW, make_float4(acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]));
                        }
                    }
                }
            }
        }
    }
}

// gM = G U^T with the pixel index as the contraction: A[i][k] = g[channel i][pixel k],
// B[k][j] = u[channel j][pixel k].  Both operands have the CHANNEL on the lane, so tiles
// [C channels][32 pixels] are staged through LDS (coalesced 16-byte loads, stride-33 rows for
// conflict-free fragment reads).  Wave w owns output tiles w, w+4, ... of the (C/32)^2 grid; the
// pixel range is split over workgroups and the partial matrices are added in a fixed order.
constexpr int kGmKP = 64;      // pixels per staged chunk
constexpr int kGmLd = kGmKP + 2; // row stride (floats): even for 8-byte stores, 2*ch+p banks stay distinct
template <typename IO>
__global__ __launch_bounds__(256) void mix_gm_mfma_kernel(const IO* __restrict__ u, const IO* __restrict__ g,
                                                          float* __restrict__ part, int B, int C, int HW, int nsplit,
                                                          int accp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [2][C][kGmLd]
    float* sg = sm;
    float* su = sm + C * kGmLd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = C / 32, ntile = T * T;
    const int kh = lane >> 5, jj = lane & 31;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int per_sample = (HW + kGmKP - 1) / kGmKP;
    const long total = (long)B * per_sample;
    for (long ch = blockIdx.x; ch < total; ch += nsplit) {
        const int b = (int)(ch / per_sample);
        const int p0 = (int)(ch % per_sample) * kGmKP;
        __syncthreads();                               // previous chunk fully consumed
        for (int e = tid; e < C * (kGmKP / 4); e += 256) {
            const int c = e / (kGmKP / 4), c4 = e % (kGmKP / 4);
            const int p = p0 + 4 * c4;
            const bool pv = p < HW;
            const size_t off = ((size_t)b * C + c) * HW + p;
            const float4 gv = pv ? Io4<IO>::ld(g + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 uv = pv ? Io4<IO>::ld(u + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            float* dg = sg + c * kGmLd + 4 * c4;
            float* du = su + c * kGmLd + 4 * c4;
            *reinterpret_cast<float2*>(dg) = make_float2(gv.x, gv.y); *reinterpret_cast<float2*>(dg + 2) = make_float2(gv.z, gv.w);
            *reinterpret_cast<float2*>(du) = make_float2(uv.x, uv.y); *reinterpret_cast<float2*>(du + 2) = make_float2(uv.z, uv.w);
        }
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < kGmKP / 2; ++ks) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int tile = wave + 4 * t;
                if (tile < ntile) {
                    const int it = tile / T, jt = tile % T;
                    const float av = sg[(32 * it + jj) * kGmLd + 2 * ks + kh];
                    const float bv = su[(32 * jt + jj) * kGmLd + 2 * ks + kh];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }
    float* dst = part + (size_t)blockIdx.x * C * C;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tile = wave + 4 * t;
        if (tile < ntile) {
            const int it = tile / T, jt = tile % T;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * kh;
                dst[i * C + 32 * jt + jj] = acc[t][r];
            }
        }
    }
}

// Backward of the mixing in ONE pass (C = 64: the cifar10 width, 4 waves; C = 128: the SVHN width,
// 8 waves): a workgroup stages a [C channels][64 pixels] tile of g and of u in LDS (as
// mix_gm_mfma_kernel does) and uses it twice,
//   gu tile  = M^T g     (A = M^T fragments from LDS, B = g rows;  tile = (channel tile, pixel half))
//   gM part += g u^T     (A = g, B = u, contraction over the 64 pixels;  tile = (i tile, j tile))
// so g is read once instead of twice and the two products share the staging: 12 B/element
// (read g, u; write gu) instead of 8 + 8.  Partial gM matrices are reduced by mix_gm_reduce_kernel.
// WLDS: the M^T fragment table sits in LDS (C = 64: 16 KB); for C = 128 it would take 64 KB and leave
// room for one workgroup per CU only, so there the fragments are read from the (L2-resident) table
// `Mfrag` that mix_frag_kernel lays out once per call, and two workgroups overlap staging with MFMAs.
template <typename IO, int C, int W, bool WLDS>
__global__ __launch_bounds__(64 * W) void mix_bwd_fused_kernel(const IO* __restrict__ u, const IO* __restrict__ g,
                                                               const float* __restrict__ M, const float* __restrict__ Mfrag,
                                                               IO* __restrict__ gu, float* __restrict__ part, int B, int HW,
                                                               int nsplit, int accp) {
    constexpr int KS = C / 2, T = C / 32, NT = 64 * W;
    constexpr int NTM = T * T / W;                       // gM tiles per wave
    constexpr int NTU = 2 * T / W;                       // gu tiles per wave
    static_assert(NTM >= 1 && NTU >= 1 && T * T % W == 0 && 2 * T % W == 0, "tile split");
    extern __shared__ __attribute__((aligned(16))) float sm[];   // wfrag [T][KS][64] | sg [C][kGmLd] | su [C][kGmLd]
    float* wfrag = sm;
    float* sg = sm + (WLDS ? C * C : 0);
    float* su = sg + C * kGmLd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, jj = lane & 31;
    if constexpr (WLDS) {
        for (int e = tid; e < C * C; e += NT) {                    // A fragments of M^T: A[i][k] = M[k][i]
            const int ln = e & 63, ks = (e >> 6) % KS, it = (e >> 6) / KS;
            const int i = 32 * it + (ln & 31), k = 2 * ks + (ln >> 5);
            wfrag[e] = M[k * C + i];
        }
    }
    f32x16 acc_m[NTM];
#pragma unroll
    for (int t = 0; t < NTM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_m[t][r] = 0.f;
    const int per_sample = (HW + kGmKP - 1) / kGmKP;
    const long total = (long)B * per_sample;
    // The next chunk's tile is fetched into registers while the MFMAs of the current one run.  All workgroups start
    // together and every one of them alternated between waiting for its loads and multiplying: 768 x 32 KB in flight
    // for a moment, then nothing (per-workgroup stamps: ~24 k cycles per chunk against 12 k of MFMA for the three
    // waves of a SIMD and ~16 k of traffic at copy speed).
    constexpr int ITER = C * (kGmKP / 4) / NT;
    static_assert(C * (kGmKP / 4) % NT == 0, "the tile is a whole number of 16-byte pieces per thread");
    float4 gq[ITER], uq[ITER];
    auto fetch = [&](long chn) __attribute__((always_inline)) {
        const int fb = (int)(chn / per_sample);
        const int fp0 = (int)(chn % per_sample) * kGmKP;
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int e = tid + i * NT;
            const int c = e / (kGmKP / 4), c4 = e % (kGmKP / 4);
            const int p = fp0 + 4 * c4;
            const bool pv = p < HW && chn < total;
            const size_t off = ((size_t)fb * C + c) * HW + p;
            gq[i] = pv ? Io4<IO>::ld(g + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            uq[i] = pv ? Io4<IO>::ld(u + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(blockIdx.x);
    for (long ch = blockIdx.x; ch < total; ch += nsplit) {
        const int b = (int)(ch / per_sample);
        const int p0 = (int)(ch % per_sample) * kGmKP;
        __syncthreads();                               // previous chunk fully consumed (and wfrag written)
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int e = tid + i * NT;
            const int c = e / (kGmKP / 4), c4 = e % (kGmKP / 4);
            float* dg = sg + c * kGmLd + 4 * c4;
            float* du = su + c * kGmLd + 4 * c4;
            *reinterpret_cast<float2*>(dg) = make_float2(gq[i].x, gq[i].y); *reinterpret_cast<float2*>(dg + 2) = make_float2(gq[i].z, gq[i].w);
            *reinterpret_cast<float2*>(du) = make_float2(uq[i].x, uq[i].y); *reinterpret_cast<float2*>(du + 2) = make_float2(uq[i].z, uq[i].w);
        }
        __syncthreads();
        fetch(ch + nsplit);                            // in flight under the MFMAs below
        // gM: contraction index = pixel pair ks of the tile
#pragma unroll 4
        for (int ks = 0; ks < kGmKP / 2; ++ks) {
#pragma unroll
            for (int t = 0; t < NTM; ++t) {
                const int tile = wave + W * t, it = tile / T, jt = tile % T;
                const float am = sg[(32 * it + jj) * kGmLd + 2 * ks + kh];
                const float bm = su[(32 * jt + jj) * kGmLd + 2 * ks + kh];
                acc_m[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(am, bm, acc_m[t], 0, 0, 0);
            }
        }
        // gu: contraction index = channel pair ks
        f32x16 acc_u[NTU];
#pragma unroll
        for (int t = 0; t < NTU; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_u[t][r] = 0.f;
#pragma unroll 4
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int t = 0; t < NTU; ++t) {
                const int tile = wave + W * t, ot = tile >> 1, pg = tile & 1;
                const float au = WLDS ? wfrag[(ot * KS + ks) * 64 + lane] : Mfrag[(ot * KS + ks) * 64 + lane];
                const float bu = sg[(2 * ks + kh) * kGmLd + 32 * pg + jj];
                acc_u[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(au, bu, acc_u[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < NTU; ++t) {
            const int tile = wave + W * t, ot = tile >> 1, pg = tile & 1;
            const int px = p0 + 32 * pg + jj;
            if (px < HW) {
                IO* ob = gu + (size_t)b * C * HW + px;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int chn = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    Io<IO>::st(ob + (size_t)chn * HW, acc_u[t][r]);
                }
            }
        }
    }
    float* dst = part + (size_t)blockIdx.x * C * C;
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
        const int tile = wave + W * t, it = tile / T, jt = tile % T;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * kh;
            float* o = dst + i * C + 32 * jt + jj;
            *o = accp ? *o + acc_m[t][r] : acc_m[t][r];
        }
    }
}
// fragment table of M^T in the order the MFMA A operand wants it: [C/32 tiles][C/2 k-steps][64 lanes]
__global__ __launch_bounds__(256) void mix_frag_kernel(const float* __restrict__ M, float* __restrict__ frag, int C) {
    const int KS = C / 2;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= C * C) return;
    const int ln = e & 63, ks = (e >> 6) % KS, it = (e >> 6) / KS;
    const int i = 32 * it + (ln & 31), k = 2 * ks + (ln >> 5);
    frag[e] = M[k * C + i];
}
bool mfma_fused_ok(int C, int HW) { return (C == 32 || C == 64 || C == 96 || C == 128) && (HW % 4) == 0; }
int fused_splits(int B, int C, int HW) {
    const long chunks = (long)B * ((HW + kGmKP - 1) / kGmKP);
    // workgroups resident at once: C=32 one-wave groups, 6 per CU (21 KB of LDS); 64: 3 (50 KB); 96: 1 (87 KB); 128: 2 (68 KB)
    const long want = C == 32 ? 1536 : C == 64 ? 768 : C == 96 ? 256 : 512;
    return (int)(chunks < want ? chunks : want);
}
template <typename IO, int C, int W, bool WLDS>
void launch_fused(const void* u, const void* g, const float* M, void* gu, float* part, int B, int HW, int nsplit,
                  int accp, hipStream_t st) {
    const size_t lds = (size_t)((WLDS ? C * C : 0) + 2 * C * kGmLd) * sizeof(float);
    float* frag = part + (size_t)nsplit * C * C;     // behind the partial matrices (workspace sized for it)
    if (!WLDS) hipLaunchKernelGGL(mix_frag_kernel, dim3((C * C + 255) / 256), dim3(256), 0, st, M, frag, C);
    static unsigned long long cfg = 0;
    ensure_lds((const void*)mix_bwd_fused_kernel<IO, C, W, WLDS>, (int)lds, cfg);
    hipLaunchKernelGGL((mix_bwd_fused_kernel<IO, C, W, WLDS>), dim3(nsplit), dim3(64 * W), lds, st, (const IO*)u, (const IO*)g, M, frag, (IO*)gu, part, B, HW, nsplit, accp);
}

bool mfma_apply_ok(int C, int HW) { return (C % 32) == 0 && C <= 128 && (HW % 4) == 0; }   // W fragments: C*C*4 B of LDS
bool mfma_gm_ok(int C, int HW) { return (C % 32) == 0 && C <= 128 && (HW % 4) == 0; }   // up to 16 tiles: 4 per wave
int gm_mfma_splits(int B, int HW) {
    const long chunks = (long)B * ((HW + kGmKP - 1) / kGmKP);
    return (int)(chunks < 512 ? chunks : 512);
}

int gm_splits(int B, int C, int HW) {
    const int tiles = (C + kT - 1) / kT;
    const long chunks = (long)B * ((HW + kK - 1) / kK);
    long n = 1024 / (tiles * tiles);
    if (n < 1) n = 1;
    if (n > chunks) n = chunks;
    return (int)n;
}

int launch_apply_mfma(int B, int C, int HW, int io, const void* u, const float* M, void* out, int trans,
                      hipStream_t st) {
    const size_t lds = (size_t)C * C * sizeof(float);
    const long nblk = (long)B * ((HW + 127) / 128);
    long grid = (nblk + 3) / 4;
    if (grid > 1024) grid = 1024;
    if (io == PDE_IO_F32) {
        static unsigned long long cfg = 0;
        ensure_lds((const void*)mix_apply_mfma_kernel<float>, 65536, cfg);
        hipLaunchKernelGGL((mix_apply_mfma_kernel<float>), dim3((unsigned)grid), dim3(256), lds, st, (const float*)u, M, (float*)out, B, C, HW, trans);
    } else {
        static unsigned long long cfg = 0;
        ensure_lds((const void*)mix_apply_mfma_kernel<bf16s>, 65536, cfg);
        hipLaunchKernelGGL((mix_apply_mfma_kernel<bf16s>), dim3((unsigned)grid), dim3(256), lds, st, (const bf16s*)u, M, (bf16s*)out, B, C, HW, trans);
    }
    return check_launch();
}

}  // namespace
}  // namespace pde

using namespace pde;

extern "C" {

int pde_channel_mix_forward(int32_t B, int32_t C, int32_t HW, int32_t io_dtype, const void* u, const float* M,
                            void* out, void* stream) {
    if (B <= 0 || C <= 0 || HW <= 0 || !u || !M || !out) return PDE_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype != PDE_IO_F32 && io_dtype != PDE_IO_BF16) return PDE_E_BADARG;
    if (io_dtype == PDE_IO_BF16 && mix_bf16_ok(C, HW)) return mix_bf16_apply(B, C, HW, u, M, out, 0, st);
    if (mfma_apply_ok(C, HW)) return launch_apply_mfma(B, C, HW, io_dtype, u, M, out, 0, st);
    dim3 grid((HW + 255) / 256, B);
    if (io_dtype == PDE_IO_F32)
        hipLaunchKernelGGL((mix_apply_kernel<float, false>), grid, dim3(256), 0, st, (const float*)u, M, (float*)out, C, HW);
    else
        hipLaunchKernelGGL((mix_apply_kernel<bf16s, false>), grid, dim3(256), 0, st, (const bf16s*)u, M, (bf16s*)out, C, HW);
    return check_launch();
}

size_t pde_channel_mix_backward_workspace_bytes(int32_t B, int32_t C, int32_t HW) {
    if (B <= 0 || C <= 0 || HW <= 0) return 0;
    // the maximum over every path a call with these dimensions can take (tensor type, PDE_MIX_* switches):
    // the caller sizes one workspace without saying which tensor type it will pass
    int n = gm_splits(B, C, HW);
    if (mfma_fused_ok(C, HW)) n = n > fused_splits(B, C, HW) ? n : fused_splits(B, C, HW);
    if (mfma_gm_ok(C, HW)) n = n > gm_mfma_splits(B, HW) ? n : gm_mfma_splits(B, HW);
    if (mix_bf16_ok(C, HW)) n = n > mix_bf16_splits(B, C, HW) ? n : mix_bf16_splits(B, C, HW);
    if (mix_split_ok(C, HW)) n = n > mix_split_splits(B, C, HW) ? n : mix_split_splits(B, C, HW);
    return (size_t)(n + 1) * C * C * sizeof(float);      // partial matrices + one fragment table
}

int pde_channel_mix_backward(int32_t B, int32_t C, int32_t HW, int32_t io_dtype, const void* u, const void* gout,
                             const float* M, void* gu, float* gM, void* workspace, size_t workspace_bytes,
                             void* stream) {
    return pde_channel_mix_backward_steps(B, C, HW, io_dtype, u, gout, M, gu, gM, workspace, workspace_bytes, 0, 1, stream);
}

// The same with the matrix gradient spread over several calls (one per time step of a layer):
// accumulate = 0 starts the partial sums in `workspace`, 1 adds to them; finalize = 1 reduces them into gM.
int pde_channel_mix_backward_steps(int32_t B, int32_t C, int32_t HW, int32_t io_dtype, const void* u, const void* gout,
                                   const float* M, void* gu, float* gM, void* workspace, size_t workspace_bytes,
                                   int32_t accumulate, int32_t finalize, void* stream) {
    if (B <= 0 || C <= 0 || HW <= 0 || !u || !gout || !M || !gu || !workspace || (finalize && !gM)) return PDE_E_BADARG;
    if (workspace_bytes < pde_channel_mix_backward_workspace_bytes(B, C, HW)) return PDE_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (io_dtype != PDE_IO_F32 && io_dtype != PDE_IO_BF16) return PDE_E_BADARG;
    dim3 grid((HW + 255) / 256, B);
    const int tiles = (C + kT - 1) / kT;
    float* part = static_cast<float*>(workspace);
    if (io_dtype == PDE_IO_BF16 && mix_bf16_ok(C, HW)) {   // exact bf16 products on the bf16 MFMA
        const int nsplit = mix_bf16_splits(B, C, HW);
        const int rc = mix_bf16_backward(B, C, HW, u, gout, M, gu, part, nsplit, accumulate, st);
        if (rc != PDE_OK) return rc;
        if (finalize) hipLaunchKernelGGL(mix_gm_reduce_kernel, dim3((C * C + 31) / 32), dim3(256), 0, st, part, gM, C * C, nsplit);
        return check_launch();
    }
    if (io_dtype == PDE_IO_F32 && mix_split_ok(C, HW)) {   // fp32 tensors, three bf16 pieces per operand on the bf16 MFMA
        const int nsplit = mix_split_splits(B, C, HW);
        const int rc = mix_split_backward(B, C, HW, u, gout, M, gu, part, nsplit, accumulate, st);
        if (rc != PDE_OK) return rc;
        if (finalize) hipLaunchKernelGGL(mix_gm_reduce_kernel, dim3((C * C + 31) / 32), dim3(256), 0, st, part, gM, C * C, nsplit);
        return check_launch();
    }
    if (mfma_fused_ok(C, HW) && getenv("PDE_MIX_UNFUSED") == nullptr) {
        const int nsplit = fused_splits(B, C, HW);
        if (io_dtype == PDE_IO_F32) {
            if (C == 32) launch_fused<float, 32, 1, true>(u, gout, M, gu, part, B, HW, nsplit, accumulate, st);
            else if (C == 64) launch_fused<float, 64, 4, true>(u, gout, M, gu, part, B, HW, nsplit, accumulate, st);
            else if (C == 96) launch_fused<float, 96, 3, true>(u, gout, M, gu, part, B, HW, nsplit, accumulate, st);
            else launch_fused<float, 128, 8, false>(u, gout, M, gu, part, B, HW, nsplit, accumulate, st);
        } else {
            if (C == 32) launch_fused<bf16s, 32, 1, true>(u, gout, M, gu, part, B, HW, nsplit, accumulate, st);
            else if (C == 64) launch_fused<bf16s, 64, 4, true>(u, gout, M, gu, part, B, HW, nsplit, accumulate, st);
            else if (C == 96) launch_fused<bf16s, 96, 3, true>(u, gout, M, gu, part, B, HW, nsplit, accumulate, st);
            else launch_fused<bf16s, 128, 8, false>(u, gout, M, gu, part, B, HW, nsplit, accumulate, st);
        }
        if (finalize) hipLaunchKernelGGL(mix_gm_reduce_kernel, dim3((C * C + 31) / 32), dim3(256), 0, st, part, gM, C * C, nsplit);
        return check_launch();
    }
    // gu = M^T gout
    if (mfma_apply_ok(C, HW)) {
        const int rc = launch_apply_mfma(B, C, HW, io_dtype, gout, M, gu, 1, st);
        if (rc != PDE_OK) return rc;
    } else if (io_dtype == PDE_IO_F32) {
        hipLaunchKernelGGL((mix_apply_kernel<float, true>), grid, dim3(256), 0, st, (const float*)gout, M, (float*)gu, C, HW);
    } else {
        hipLaunchKernelGGL((mix_apply_kernel<bf16s, true>), grid, dim3(256), 0, st, (const bf16s*)gout, M, (bf16s*)gu, C, HW);
    }
    // gM = G U^T
    int nsplit;
    if (mfma_gm_ok(C, HW)) {
        nsplit = gm_mfma_splits(B, HW);
        const size_t lds = (size_t)2 * C * kGmLd * sizeof(float);
        if (io_dtype == PDE_IO_F32) {
            static unsigned long long cfg = 0;
        ensure_lds((const void*)mix_gm_mfma_kernel<float>, 72 * 1024, cfg);
            hipLaunchKernelGGL((mix_gm_mfma_kernel<float>), dim3(nsplit), dim3(256), lds, st, (const float*)u, (const float*)gout, part, B, C, HW, nsplit, accumulate);
        } else {
            static unsigned long long cfg = 0;
        ensure_lds((const void*)mix_gm_mfma_kernel<bf16s>, 72 * 1024, cfg);
            hipLaunchKernelGGL((mix_gm_mfma_kernel<bf16s>), dim3(nsplit), dim3(256), lds, st, (const bf16s*)u, (const bf16s*)gout, part, B, C, HW, nsplit, accumulate);
        }
    } else {
        nsplit = gm_splits(B, C, HW);
        if (io_dtype == PDE_IO_F32)
            hipLaunchKernelGGL((mix_gm_kernel<float>), dim3(tiles * tiles, nsplit), dim3(256), 0, st, (const float*)u,
                               (const float*)gout, part, B, C, HW, nsplit, accumulate);
        else
            hipLaunchKernelGGL((mix_gm_kernel<bf16s>), dim3(tiles * tiles, nsplit), dim3(256), 0, st, (const bf16s*)u,
                               (const bf16s*)gout, part, B, C, HW, nsplit, accumulate);
    }
    if (finalize) hipLaunchKernelGGL(mix_gm_reduce_kernel, dim3((C * C + 31) / 32), dim3(256), 0, st, part, gM, C * C, nsplit);
    return check_launch();
}

}  // extern "C"
