// Channel operators on bf16 tensors (SVHN width C = 128, bf16 I/O).
//
// With bf16 inputs the products of the mixing are exact in a bf16 MFMA: v_mfma_f32_32x32x16_bf16 multiplies
// bf16 x bf16 into fp32 and accumulates in fp32, at 16x the rate of the fp32 MFMA the generic path uses
// (32 cycles for K = 16 against 64 cycles for K = 2).  Only the fp32 matrix M has to be split, M = hi + lo
// with both parts bf16 (relative error 2^-17, far below the bf16 resolution of the outputs):
//   forward      out^T tile = u^T (hi + lo)^T      2 MFMAs per K = 16
//   backward     gu^T  tile = g^T (hi + lo)        2 MFMAs per K = 16
//                gM   += g u^T over the pixels     1 MFMA  per K = 16, EXACT products
// Operand maps (cdna_hip_programming.md §3): lane l = (h = l>>5, r = l&31) holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7; D: col = r, row = (reg&3) + 8*(reg>>2) + 4*h.
// A workgroup stages a [C channels][64 pixels] tile in LDS in its natural layout (rows of 128 B + 16 B pad):
//   * k = pixel (gM): both operands are row reads, ds_read_b128;
//   * k = channel (out, gu): the A operand u^T / g^T comes from the same image through the transposing read
//     ds_read_b64_tr_b16 (4 channel rows x 16 pixels per 16 lanes, delivered pixel-major);
//   * the result tile is [pixel][channel] on the lanes, i.e. every lane holds 4 x 4 consecutive pixels of ONE
//     channel: it goes back through an LDS image and out as 16-byte stores.
#include "pde_common.h"

#include <cstdlib>

namespace pde {
namespace {

typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kPx = 64;                 // pixels per tile
constexpr int kRow = 144;               // bytes per tile row: 128 + 16 pad (16 consecutive rows hit 64 distinct banks)

__device__ __forceinline__ unsigned short f2bf(float f) { return f32_to_bf16_hw(f); }
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

// hi/lo bf16 fragments of M for the B operand, in LDS: [piece][tile][k-step][lane][8].
// TRANS = false (forward):  B[k = in j][col = out i]  = M[i][j]:  lane (h,r) of (ot,ks): M[32ot + r][16ks + 8h + jj]
// TRANS = true  (backward): B[k = out i][col = in j]  = M[i][j]:  lane (h,r) of (jt,ks): M[16ks + 8h + jj][32jt + r]
template <int C, bool TRANS>
__device__ __forceinline__ void build_frags(const float* __restrict__ M, unsigned short* frag, int tid, int nthreads) {
    constexpr int T = C / 32, KS = C / 16;
    for (int e = tid; e < T * KS * 64; e += nthreads) {
        const int lane = e & 63, ks = (e >> 6) % KS, t = (e >> 6) / KS;
        const int h = lane >> 5, r = lane & 31;
        unsigned short hi[8], lo[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float w = TRANS ? M[(16 * ks + 8 * h + jj) * C + 32 * t + r] : M[(32 * t + r) * C + 16 * ks + 8 * h + jj];
            hi[jj] = f2bf(w);
            lo[jj] = f2bf(w - bf2f(hi[jj]));
        }
        uint4* dh = reinterpret_cast<uint4*>(frag + (size_t)e * 8);
        uint4* dl = reinterpret_cast<uint4*>(frag + (size_t)(T * KS * 64 + e) * 8);
        *dh = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
        *dl = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
    }
}

// global [C][HW] bf16 (one sample, pixels p0..p0+63) -> LDS image, 16-byte pieces
template <int C>
__device__ __forceinline__ void load_tile(const unsigned short* __restrict__ src, int HW, int p0, unsigned char* img,
                                          int tid, int nthreads) {
    for (int e = tid; e < C * 8; e += nthreads) {
        const int row = e >> 3, ch = e & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)row * HW + p0 + 8 * ch);
        *reinterpret_cast<uint4*>(img + row * kRow + 16 * ch) = v;
    }
}
// The same in two halves, so that the NEXT tile's global loads are in flight while the current one is
// being multiplied: fetch -> (compute of the previous tile) -> deposit.
// (every thread owns exactly two 16-byte pieces of a tile: C*8 pieces, C*4 threads; plain variables, not an
// array: an array passed down by reference ended up in scratch here)
template <int C>
__device__ __forceinline__ void fetch_tile(const unsigned short* __restrict__ src, int HW, int p0, uint4& t0, uint4& t1, int tid) {
    constexpr int NT = C * 4;
    const int e0 = tid, e1 = tid + NT;
    t0 = *reinterpret_cast<const uint4*>(src + (size_t)(e0 >> 3) * HW + p0 + 8 * (e0 & 7));
    t1 = *reinterpret_cast<const uint4*>(src + (size_t)(e1 >> 3) * HW + p0 + 8 * (e1 & 7));
}
template <int C>
__device__ __forceinline__ void deposit_tile(const uint4& t0, const uint4& t1, unsigned char* img, int tid) {
    constexpr int NT = C * 4;
    const int e0 = tid, e1 = tid + NT;
    *reinterpret_cast<uint4*>(img + (e0 >> 3) * kRow + 16 * (e0 & 7)) = t0;
    *reinterpret_cast<uint4*>(img + (e1 >> 3) * kRow + 16 * (e1 & 7)) = t1;
}

template <int C>
__device__ __forceinline__ void store_tile(unsigned short* __restrict__ dst, int HW, int p0, const unsigned char* img,
                                           int tid, int nthreads) {
    for (int e = tid; e < C * 8; e += nthreads) {
        const int row = e >> 3, ch = e & 7;
        *reinterpret_cast<uint4*>(dst + (size_t)row * HW + p0 + 8 * ch) = *reinterpret_cast<const uint4*>(img + row * kRow + 16 * ch);
    }
}

// A operand X^T from the natural image of X ([channel][pixel]): A[row = pixel 32pg + r][k = channel 16ks + 8h + jj]
__device__ __forceinline__ v8bf tr_operand(const unsigned char* img, int pg, int ks, int lane) {
    const int grp = lane >> 4, i = lane & 15, h = lane >> 5;
    const int q = i >> 2, p = i & 3;
    // lane 4q+p of a 16-lane group supplies row q (of 4 channel rows), pixel columns 4p..4p+3 of the group's 16
    const unsigned char* a0 = img + (16 * ks + 8 * h + q) * kRow + (32 * pg + 16 * (grp & 1) + 4 * p) * 2;
    const v4s lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(a0));
    const v4s hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(a0 + 4 * kRow));
    const v8s all = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(v8bf, all);
}
__device__ __forceinline__ v8bf frag_operand(const unsigned short* frag, int idx) {
    return __builtin_bit_cast(v8bf, *reinterpret_cast<const v8s*>(frag + (size_t)idx * 8));
}
// row operand: 8 consecutive pixels of one channel row of the image
__device__ __forceinline__ v8bf row_operand(const unsigned char* img, int row, int px) {
    return __builtin_bit_cast(v8bf, *reinterpret_cast<const v8s*>(img + row * kRow + px * 2));
}
// D tile [pixel][channel]: lane (h,r) holds channel 32t + r, pixels 32pg + 8q + 4h + 0..3 in regs 4q..4q+3
__device__ __forceinline__ void put_result(unsigned char* img, const f32x16& acc, int t, int pg, int lane) {
    const int h = lane >> 5, r = lane & 31;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned int a = f2bf(acc[4 * q]) | ((unsigned int)f2bf(acc[4 * q + 1]) << 16);
        const unsigned int b = f2bf(acc[4 * q + 2]) | ((unsigned int)f2bf(acc[4 * q + 3]) << 16);
        *reinterpret_cast<uint2*>(img + (32 * t + r) * kRow + (32 * pg + 8 * q + 4 * h) * 2) = make_uint2(a, b);
    }
}

// ---- forward: out = M u -------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(C * 4) void mix_apply_bf16_kernel(const unsigned short* __restrict__ u, const float* __restrict__ M,
                                                               unsigned short* __restrict__ out, int B, int HW, int trans) {
    constexpr int T = C / 32, KS = C / 16, NT = C * 4, FR = T * KS * 64;     // waves = 2 pixel groups x T channel tiles
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* frag = reinterpret_cast<unsigned short*>(smem);          // [2][FR][8] bf16 = 4*C*C bytes
    unsigned char* img_in = smem + (size_t)4 * C * C;
    unsigned char* img_out = img_in + C * kRow;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (trans) build_frags<C, true>(M, frag, tid, NT);
    else build_frags<C, false>(M, frag, tid, NT);
    const int pg = wave & 1, ot = wave >> 1;
    const int per_sample = HW / kPx;
    const long total = (long)B * per_sample;
    uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;
    if ((long)blockIdx.x < total)
        fetch_tile<C>(u + (size_t)(blockIdx.x / per_sample) * C * HW, HW, (int)(blockIdx.x % per_sample) * kPx, n0, n1, tid);
    for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int b = (int)(tile / per_sample), p0 = (int)(tile % per_sample) * kPx;
        __syncthreads();                                   // fragments built / previous tile's images consumed
        deposit_tile<C>(n0, n1, img_in, tid);
        __syncthreads();
        const long tn = tile + gridDim.x;                  // the next tile's loads fly during this tile's MFMAs
        if (tn < total) fetch_tile<C>(u + (size_t)(tn / per_sample) * C * HW, HW, (int)(tn % per_sample) * kPx, n0, n1, tid);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 2
        for (int ks = 0; ks < KS; ++ks) {
            const v8bf a = tr_operand(img_in, pg, ks, lane);
            const int fi = (ot * KS + ks) * 64 + lane;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag_operand(frag, fi), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag_operand(frag, FR + fi), acc, 0, 0, 0);
        }
        put_result(img_out, acc, ot, pg, lane);
        __syncthreads();
        store_tile<C>(out + (size_t)b * C * HW, HW, p0, img_out, tid, NT);
    }
}

// ---- backward: gu = M^T g and the partial sums of gM = g u^T ---------------------------------------------
template <int C>
__global__ __launch_bounds__(C * 4) void mix_bwd_bf16_kernel(const unsigned short* __restrict__ u, const unsigned short* __restrict__ g,
                                                             const float* __restrict__ M, unsigned short* __restrict__ gu,
                                                             float* __restrict__ part, int B, int HW, int accp) {
    constexpr int T = C / 32, KS = C / 16, NT = C * 4, W = C / 16, FR = T * KS * 64;
    constexpr int NTM = T * T / W;                         // gM tiles per wave (C = 64: 1, C = 128: 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* frag = reinterpret_cast<unsigned short*>(smem);
    unsigned char* img_g = smem + (size_t)4 * C * C;
    unsigned char* img_u = img_g + C * kRow;
    unsigned char* img_o = img_u + C * kRow;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, r = lane & 31;
    build_frags<C, true>(M, frag, tid, NT);
    const int pg = wave & 1, jt = wave >> 1;               // my gu tile: pixel group, input-channel tile
    f32x16 acc_m[NTM];
#pragma unroll
    for (int t = 0; t < NTM; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_m[t][i] = 0.f;
    const int per_sample = HW / kPx;
    const long total = (long)B * per_sample;
    uint4 g0 = make_uint4(0, 0, 0, 0), g1 = g0, u0 = g0, u1 = g0;
    if ((long)blockIdx.x < total) {
        const size_t o = (size_t)(blockIdx.x / per_sample) * C * HW;
        const int q0 = (int)(blockIdx.x % per_sample) * kPx;
        fetch_tile<C>(g + o, HW, q0, g0, g1, tid);
        fetch_tile<C>(u + o, HW, q0, u0, u1, tid);
    }
    for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int b = (int)(tile / per_sample), p0 = (int)(tile % per_sample) * kPx;
        __syncthreads();
        deposit_tile<C>(g0, g1, img_g, tid);
        deposit_tile<C>(u0, u1, img_u, tid);
        __syncthreads();
        const long tn = tile + gridDim.x;
        if (tn < total) {
            const size_t o = (size_t)(tn / per_sample) * C * HW;
            const int q0 = (int)(tn % per_sample) * kPx;
            fetch_tile<C>(g + o, HW, q0, g0, g1, tid);
            fetch_tile<C>(u + o, HW, q0, u0, u1, tid);
        }
        // gM: contraction over the 64 pixels of the tile, 16 per MFMA
#pragma unroll
        for (int kp = 0; kp < kPx / 16; ++kp) {
#pragma unroll
            for (int t = 0; t < NTM; ++t) {
                const int tl = wave + W * t, it = tl / T, jt2 = tl % T;
                const v8bf a = row_operand(img_g, 32 * it + r, 16 * kp + 8 * h);
                const v8bf bb = row_operand(img_u, 32 * jt2 + r, 16 * kp + 8 * h);
                acc_m[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc_m[t], 0, 0, 0);
            }
        }
        // gu^T tile = g^T (hi + lo)
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 2
        for (int ks = 0; ks < KS; ++ks) {
            const v8bf a = tr_operand(img_g, pg, ks, lane);
            const int fi = (jt * KS + ks) * 64 + lane;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag_operand(frag, fi), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag_operand(frag, FR + fi), acc, 0, 0, 0);
        }
        put_result(img_o, acc, jt, pg, lane);
        __syncthreads();
        store_tile<C>(gu + (size_t)b * C * HW, HW, p0, img_o, tid, NT);
    }
    float* dst = part + (size_t)blockIdx.x * C * C;
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
        const int tl = wave + W * t, it = tl / T, jt2 = tl % T;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = 32 * it + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            float* o = dst + i * C + 32 * jt2 + r;
            *o = accp ? *o + acc_m[t][reg] : acc_m[t][reg];
        }
    }
}

// ---- fp32 tensors on the bf16 matrix cores: every operand as THREE bf16 pieces -------------------------------------
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 3 x 8 mantissa bits, the remainder is
// below 2^-24 |x|.  A product a b is taken as the six piece products of order <= 2 (hi hi, hi mid, mid hi, hi lo, lo hi,
// mid mid; the dropped ones are below 2^-24 |a b|), each EXACT in the MFMA's fp32 accumulation: fp32-level accuracy at
// 6/16 of the fp32 MFMA's time (v_mfma_f32_32x32x16_bf16 does 16 k-steps in 32 cycles, v_mfma_f32_32x32x2_f32 2 in 64).
// Same structure as mix_bwd_bf16_kernel: a [C][64 px] tile of g and u, split once when it is deposited in LDS (three
// natural-layout images per tensor), both products from there; the result tile goes out through an fp32 image that reuses
// the space of two u images.  C = 64: 24 KB of M fragments + 54 KB of images: two workgroups per CU.
constexpr int kRowF = 272;              // bytes per row of the fp32 result image: 256 + 16 pad

__device__ __forceinline__ void split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    hi = f2bf(x);
    const float r1 = x - bf2f(hi);
    mid = f2bf(r1);
    lo = f2bf(r1 - bf2f(mid));
}

template <int C>
__device__ __forceinline__ void build_frags3(const float* __restrict__ M, unsigned short* frag, int tid, int nthreads) {
    // B[k = out i][col = in j] = M[i][j] (backward): lane (h,r) of (jt,ks): M[16ks + 8h + jj][32jt + r]; [piece][tile][ks][lane][8]
    constexpr int T = C / 32, KS = C / 16, FR = T * KS * 64;
    for (int e = tid; e < FR; e += nthreads) {
        const int lane = e & 63, ks = (e >> 6) % KS, t = (e >> 6) / KS;
        const int h = lane >> 5, r = lane & 31;
        unsigned short p[3][8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) split3(M[(16 * ks + 8 * h + jj) * C + 32 * t + r], p[0][jj], p[1][jj], p[2][jj]);
#pragma unroll
        for (int q = 0; q < 3; ++q)
            *reinterpret_cast<uint4*>(frag + ((size_t)q * FR + e) * 8) =
                make_uint4(p[q][0] | (p[q][1] << 16), p[q][2] | (p[q][3] << 16), p[q][4] | (p[q][5] << 16), p[q][6] | (p[q][7] << 16));
    }
}

template <int C>
__global__ __launch_bounds__(C * 4) void mix_bwd_split_kernel(const float* __restrict__ u, const float* __restrict__ g,
                                                              const float* __restrict__ M, float* __restrict__ gu,
                                                              float* __restrict__ part, int B, int HW, int accp) {
    constexpr int T = C / 32, KS = C / 16, NT = C * 4, W = C / 16, FR = T * KS * 64, TT = T * T;
    // gM tiles (T x T of 32 x 32) over the W waves: whole tiles per wave where there are at least as many tiles as waves
    // (C = 64: one each; C = 96: nine tiles on six waves), else (C = 32: one tile, two waves) the pixel contraction of a
    // tile is cut into W / TT parts and the partial tiles meet in LDS at the end
    constexpr int KPARTS = TT >= W ? 1 : W / TT;
    constexpr int NTM = TT >= W ? (TT + W - 1) / W : 1;
    constexpr int IMG = C * kRow;                          // bytes of one bf16 image
    constexpr int PCS = C * 16 / NT;                       // 16-byte pieces of an fp32 tile per thread (4)
    static_assert(2 * IMG >= C * kRowF, "the result image fits in two piece images");
    static_assert(KPARTS == 1 || (kPx / 16) % KPARTS == 0, "the pixel contraction splits evenly");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* frag = reinterpret_cast<unsigned short*>(smem);          // [3][FR][8] bf16 = 6*C*C bytes
    unsigned char* img_g = smem + (size_t)6 * C * C;                          // [3][C][kRow]
    unsigned char* img_u = img_g + 3 * IMG;                                   // [3][C][kRow]
    unsigned char* img_o = img_u + IMG;                                       // fp32 result over u's mid and lo images
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, r = lane & 31;
    build_frags3<C>(M, frag, tid, NT);
    const int pg = wave & 1, jt = wave >> 1;               // my gu tile: pixel group, input-channel tile
    f32x16 acc_m[NTM];
#pragma unroll
    for (int t = 0; t < NTM; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_m[t][i] = 0.f;
    const int per_sample = (HW + kPx - 1) / kPx;           // the last tile of a sample may be ragged (HW a multiple of 4)
    const long total = (long)B * per_sample;
    float4 gq[PCS], uq[PCS];
    auto fetch = [&](long tile) __attribute__((always_inline)) {
        const size_t o = (size_t)(tile / per_sample) * C * HW;
        const int q0 = (int)(tile % per_sample) * kPx;
        const int valid = HW - q0 < kPx ? HW - q0 : kPx;
#pragma unroll
        for (int i = 0; i < PCS; ++i) {
            const int e = tid + i * NT, row = e >> 4, c4 = e & 15;
            const int cc = 4 * c4 < valid ? 4 * c4 : 0;    // beyond the plane: re-read the tile's first piece, zeroed when deposited
            gq[i] = *reinterpret_cast<const float4*>(g + o + (size_t)row * HW + q0 + cc);
            uq[i] = *reinterpret_cast<const float4*>(u + o + (size_t)row * HW + q0 + cc);
        }
    };
    auto deposit = [&](const float4 (&q)[PCS], unsigned char* img, int valid) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PCS; ++i) {
            const int e = tid + i * NT, row = e >> 4, c4 = e & 15;
            const bool in = 4 * c4 < valid;
            const float v[4] = {in ? q[i].x : 0.f, in ? q[i].y : 0.f, in ? q[i].z : 0.f, in ? q[i].w : 0.f};
            unsigned short p[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split3(v[j], p[0][j], p[1][j], p[2][j]);
#pragma unroll
            for (int s = 0; s < 3; ++s)
                *reinterpret_cast<uint2*>(img + s * IMG + row * kRow + 8 * c4) = make_uint2(p[s][0] | (p[s][1] << 16), p[s][2] | (p[s][3] << 16));
        }
    };
    if ((long)blockIdx.x < total) fetch(blockIdx.x);
    for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int b = (int)(tile / per_sample), p0 = (int)(tile % per_sample) * kPx;
        const int valid = HW - p0 < kPx ? HW - p0 : kPx;
        __syncthreads();                                   // fragments built / previous tile's images consumed and stored
        deposit(gq, img_g, valid);
        deposit(uq, img_u, valid);
        __syncthreads();
        const long tn = tile + gridDim.x;
        if (tn < total) fetch(tn);                         // the next tile's loads fly during this tile's MFMAs
        // gM: contraction over the 64 pixels of the tile (zeros beyond a ragged one), 16 per MFMA, six piece products
#pragma unroll
        for (int kq = 0; kq < kPx / 16 / KPARTS; ++kq) {
            const int kp = KPARTS == 1 ? kq : kq * KPARTS + wave / TT;
#pragma unroll
            for (int t = 0; t < NTM; ++t) {
                const int tl = KPARTS == 1 ? wave + W * t : wave % TT;
                if (tl < TT) {
                    const int it = tl / T, jt2 = tl % T;
                    v8bf a[3], bb[3];
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        a[s] = row_operand(img_g + s * IMG, 32 * it + r, 16 * kp + 8 * h);
                        bb[s] = row_operand(img_u + s * IMG, 32 * jt2 + r, 16 * kp + 8 * h);
                    }
                    acc_m[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bb[1], acc_m[t], 0, 0, 0);   // smallest terms first
                    acc_m[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[2], acc_m[t], 0, 0, 0);
                    acc_m[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bb[0], acc_m[t], 0, 0, 0);
                    acc_m[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[1], acc_m[t], 0, 0, 0);
                    acc_m[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bb[0], acc_m[t], 0, 0, 0);
                    acc_m[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[0], acc_m[t], 0, 0, 0);
                }
            }
        }
        // gu^T tile = g^T M, pieces of g against pieces of M
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 2
        for (int ks = 0; ks < KS; ++ks) {
            const int fi = (jt * KS + ks) * 64 + lane;
            v8bf a[3], f[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                a[s] = tr_operand(img_g + s * IMG, pg, ks, lane);
                f[s] = frag_operand(frag, s * FR + fi);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], f[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], f[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], f[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], f[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], f[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], f[0], acc, 0, 0, 0);
        }
        __syncthreads();                                   // every wave is done with the u images: the result goes over two of them
        // D tile [pixel][channel]: lane (h,r) holds channel 32 jt + r, pixels 32pg + 8q + 4h + 0..3 in regs 4q..4q+3
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(img_o + (32 * jt + r) * kRowF + (32 * pg + 8 * q + 4 * h) * 4) =
                make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PCS; ++i) {
            const int e = tid + i * NT, row = e >> 4, c4 = e & 15;
            if (4 * c4 < valid)
                *reinterpret_cast<float4*>(gu + (size_t)b * C * HW + (size_t)row * HW + p0 + 4 * c4) =
                    *reinterpret_cast<const float4*>(img_o + row * kRowF + 16 * c4);
        }
    }
    float* dst = part + (size_t)blockIdx.x * C * C;
    if (KPARTS > 1) {                                      // partial tiles of the same gM tile: added in wave order through LDS
        __syncthreads();
        float* red = reinterpret_cast<float*>(img_g);      // [W][16][64]
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) red[(wave * 16 + reg) * 64 + lane] = acc_m[0][reg];
        __syncthreads();
        if (wave < TT) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                float v = 0.f;
#pragma unroll
                for (int kpart = 0; kpart < KPARTS; ++kpart) v += red[((wave + TT * kpart) * 16 + reg) * 64 + lane];
                acc_m[0][reg] = v;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
        const int tl = KPARTS == 1 ? wave + W * t : wave;
        if (tl < TT && (KPARTS == 1 || wave < TT)) {
            const int it = tl / T, jt2 = tl % T;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = 32 * it + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                float* o = dst + i * C + 32 * jt2 + r;
                *o = accp ? *o + acc_m[t][reg] : acc_m[t][reg];
            }
        }
    }
}

inline void ensure_lds(const void* kernel, int bytes, unsigned long long& done) { (void)ensure_dynamic_lds(kernel, bytes, done); }

template <int C>
void launch_apply(const void* u, const float* M, void* out, int B, int HW, int trans, hipStream_t st) {
    const size_t lds = (size_t)4 * C * C + 2 * (size_t)C * kRow;
    static unsigned long long cfg = 0;
    ensure_lds((const void*)mix_apply_bf16_kernel<C>, (int)lds, cfg);
    const long tiles = (long)B * (HW / kPx);
    const int per_cu = C == 64 ? 4 : 1;                    // workgroups an LDS footprint of 34 / 100 KB allows
    const long grid = tiles < 256L * per_cu ? tiles : 256L * per_cu;
    hipLaunchKernelGGL((mix_apply_bf16_kernel<C>), dim3((unsigned)grid), dim3(C * 4), lds, st, (const unsigned short*)u, M,
                       (unsigned short*)out, B, HW, trans);
}
template <int C>
void launch_bwd(const void* u, const void* g, const float* M, void* gu, float* part, int B, int HW, int nsplit, int accp,
                hipStream_t st) {
    const size_t lds = (size_t)4 * C * C + 3 * (size_t)C * kRow;
    static unsigned long long cfg = 0;
    ensure_lds((const void*)mix_bwd_bf16_kernel<C>, (int)lds, cfg);
    hipLaunchKernelGGL((mix_bwd_bf16_kernel<C>), dim3((unsigned)nsplit), dim3(C * 4), lds, st, (const unsigned short*)u,
                       (const unsigned short*)g, M, (unsigned short*)gu, part, B, HW, accp);
}

template <int C>
void launch_split(const void* u, const void* g, const float* M, void* gu, float* part, int B, int HW, int nsplit, int accp,
                  hipStream_t st) {
    const size_t lds = (size_t)6 * C * C + 6 * (size_t)C * kRow;
    static unsigned long long cfg = 0;
    ensure_lds((const void*)mix_bwd_split_kernel<C>, (int)lds, cfg);
    hipLaunchKernelGGL((mix_bwd_split_kernel<C>), dim3((unsigned)nsplit), dim3(C * 4), lds, st, (const float*)u, (const float*)g, M,
                       (float*)gu, part, B, HW, accp);
}

}  // namespace

// fp32 tensors through the bf16 matrix cores, three pieces per operand: C = 32, 64 or 96 (at C = 128 the three fragment
// tables of M and the six piece images do not fit in LDS together), planes of any multiple of 4 pixels (the last 64-pixel
// tile of a sample may be ragged)
bool mix_split_ok(int C, int HW) {
    return (C == 32 || C == 64 || C == 96) && (HW % 4) == 0 && HW >= 4 && getenv("PDE_MIX_NO_SPLIT") == nullptr;
}
int mix_split_splits(int B, int C, int HW) {
    const long tiles = (long)B * ((HW + kPx - 1) / kPx);
    const long want = C == 32 ? 1024 : (C == 64 ? 512 : 256);   // resident workgroups: 33 / 78 / 138 KB of LDS each
    return (int)(tiles < want ? tiles : want);
}
int mix_split_backward(int B, int C, int HW, const void* u, const void* g, const float* M, void* gu, float* part, int nsplit,
                       int accp, hipStream_t st) {
    if (C == 32) launch_split<32>(u, g, M, gu, part, B, HW, nsplit, accp, st);
    else if (C == 64) launch_split<64>(u, g, M, gu, part, B, HW, nsplit, accp, st);
    else launch_split<96>(u, g, M, gu, part, B, HW, nsplit, accp, st);
    return check_launch();
}

// entry points used by pde_mix.hip (same shared library)
bool mix_bf16_ok(int C, int HW) {
    return (C == 64 || C == 128) && (HW % kPx) == 0 && getenv("PDE_MIX_NO_BF16_MFMA") == nullptr;
}
int mix_bf16_splits(int B, int C, int HW) {
    const long tiles = (long)B * (HW / kPx);
    const long want = C == 64 ? 768 : 256;                 // resident workgroups (LDS 43 / 118 KB each)
    return (int)(tiles < want ? tiles : want);
}
int mix_bf16_apply(int B, int C, int HW, const void* u, const float* M, void* out, int trans, hipStream_t st) {
    if (C == 64) launch_apply<64>(u, M, out, B, HW, trans, st);
    else launch_apply<128>(u, M, out, B, HW, trans, st);
    return check_launch();
}
int mix_bf16_backward(int B, int C, int HW, const void* u, const void* g, const float* M, void* gu, float* part, int nsplit,
                      int accp, hipStream_t st) {
    if (C == 64) launch_bwd<64>(u, g, M, gu, part, B, HW, nsplit, accp, st);
    else launch_bwd<128>(u, g, M, gu, part, B, HW, nsplit, accp, st);
    return check_launch();
}

}  // namespace pde
