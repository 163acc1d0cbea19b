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

namespace pde {
namespace {

template <typename IO> struct Io;
template <> struct Io<float> {
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
struct bf16s { unsigned short v; };
template <> struct Io<bf16s> {
    __device__ static __forceinline__ float ld(const bf16s* p) { return __uint_as_float((unsigned int)p->v << 16); }
    __device__ static __forceinline__ void st(bf16s* p, float f) {
        unsigned int u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) { p->v = (unsigned short)((u >> 16) | 0x40); return; }
        u += 0x7fffu + ((u >> 16) & 1u);
        p->v = (unsigned short)(u >> 16);
    }
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
                                                     float* __restrict__ part, int B, int C, int HW, int nsplit) {
    __shared__ float sg[kK][kT + 1];
    __shared__ float su[kK][kT + 1];
    const int tiles = (C + kT - 1) / kT;
    const int ti = blockIdx.x / tiles, tj = blockIdx.x % tiles;
    const int split = blockIdx.y;
    const int tid = threadIdx.x;
    const int oi = (tid / 16) * 2, oj = (tid % 16) * 2;      // 2x2 outputs per thread
    float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
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
            a00 = fmaf(g0, u0, a00); a01 = fmaf(g0, u1, a01);
            a10 = fmaf(g1, u0, a10); a11 = fmaf(g1, u1, a11);
        }
        __syncthreads();
    }
    float* dst = part + (size_t)split * C * C;
    const int i = ti * kT + oi, j = tj * kT + oj;
    if (i < C && j < C) dst[i * C + j] = a00;
    if (i < C && j + 1 < C) dst[i * C + j + 1] = a01;
    if (i + 1 < C && j < C) dst[(i + 1) * C + j] = a10;
    if (i + 1 < C && j + 1 < C) dst[(i + 1) * C + j + 1] = a11;
}

__global__ void mix_gm_reduce_kernel(const float* __restrict__ part, float* __restrict__ gM, int CC, int nsplit) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= CC) return;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * CC + e];
    gM[e] = s;
}

int gm_splits(int B, int C, int HW) {
    const int tiles = (C + kT - 1) / kT;
    const long chunks = (long)B * ((HW + kK - 1) / kK);
    long n = 1024 / (tiles * tiles);
    if (n < 1) n = 1;
    if (n > chunks) n = chunks;
    return (int)n;
}

}  // namespace
}  // namespace pde

using namespace pde;

extern "C" {

int pde_channel_mix_forward(int32_t B, int32_t C, int32_t HW, int32_t io_dtype, const void* u, const float* M,
                            void* out, void* stream) {
    if (B <= 0 || C <= 0 || HW <= 0 || !u || !M || !out) return PDE_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((HW + 255) / 256, B);
    if (io_dtype == PDE_IO_F32)
        hipLaunchKernelGGL((mix_apply_kernel<float, false>), grid, dim3(256), 0, st, (const float*)u, M, (float*)out, C, HW);
    else if (io_dtype == PDE_IO_BF16)
        hipLaunchKernelGGL((mix_apply_kernel<bf16s, false>), grid, dim3(256), 0, st, (const bf16s*)u, M, (bf16s*)out, C, HW);
    else
        return PDE_E_BADARG;
    return check_launch();
}

size_t pde_channel_mix_backward_workspace_bytes(int32_t B, int32_t C, int32_t HW) {
    if (B <= 0 || C <= 0 || HW <= 0) return 0;
    return (size_t)gm_splits(B, C, HW) * C * C * sizeof(float);
}

int pde_channel_mix_backward(int32_t B, int32_t C, int32_t HW, int32_t io_dtype, const void* u, const void* gout,
                             const float* M, void* gu, float* gM, void* workspace, size_t workspace_bytes,
                             void* stream) {
    if (B <= 0 || C <= 0 || HW <= 0 || !u || !gout || !M || !gu || !gM || !workspace) return PDE_E_BADARG;
    if (workspace_bytes < pde_channel_mix_backward_workspace_bytes(B, C, HW)) return PDE_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((HW + 255) / 256, B);
    const int tiles = (C + kT - 1) / kT;
    const int nsplit = gm_splits(B, C, HW);
    float* part = static_cast<float*>(workspace);
    if (io_dtype == PDE_IO_F32) {
        hipLaunchKernelGGL((mix_apply_kernel<float, true>), grid, dim3(256), 0, st, (const float*)gout, M, (float*)gu, C, HW);
        hipLaunchKernelGGL((mix_gm_kernel<float>), dim3(tiles * tiles, nsplit), dim3(256), 0, st, (const float*)u,
                           (const float*)gout, part, B, C, HW, nsplit);
    } else if (io_dtype == PDE_IO_BF16) {
        hipLaunchKernelGGL((mix_apply_kernel<bf16s, true>), grid, dim3(256), 0, st, (const bf16s*)gout, M, (bf16s*)gu, C, HW);
        hipLaunchKernelGGL((mix_gm_kernel<bf16s>), dim3(tiles * tiles, nsplit), dim3(256), 0, st, (const bf16s*)u,
                           (const bf16s*)gout, part, B, C, HW, nsplit);
    } else {
        return PDE_E_BADARG;
    }
    hipLaunchKernelGGL(mix_gm_reduce_kernel, dim3((C * C + 255) / 256), dim3(256), 0, st, part, gM, C * C, nsplit);
    return check_launch();
}

}  // extern "C"
