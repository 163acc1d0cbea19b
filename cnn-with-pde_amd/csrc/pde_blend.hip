// SVHN skip connection (SVHN.py:73-74): out = s*u0 + (1-s)*u,  s = sigmoid(skip_weight), in one pass
// over the tensors (torch needs four elementwise kernels forward and six backward for it).
//   backward:  g_u0 = s*g,  g_u = (1-s)*g,  g_skip_weight = s*(1-s) * sum g*(u0 - u)
// The sum is taken per workgroup and added up in a fixed order by a second tiny kernel (no float atomics).  It is a sum of
// differences of nearly equal numbers that cancels heavily (u is u0 after a few small diffusion steps), so every stage of
// it — per thread, per wave, per workgroup, over the workgroups — is carried in DOUBLE precision: what is left of the
// scalar's error is the fp32 rounding of the layer's own output, not the summation (round 4; the kernel is bandwidth-bound,
// the fp64 FMAs are free).
#include "pde_common.h"

namespace pde {
namespace {

struct bf16v { unsigned short v; };
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) { return f32_to_bf16_hw(f); }
template <typename IO> struct Vec;                      // 8 elements per thread and access
template <> struct Vec<float> {
    static constexpr int kBytes = 32;
    __device__ static __forceinline__ void ld(const float* p, float (&x)[8]) {
        const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
    __device__ static __forceinline__ void st(float* p, const float (&x)[8]) {
        reinterpret_cast<float4*>(p)[0] = make_float4(x[0], x[1], x[2], x[3]);
        reinterpret_cast<float4*>(p)[1] = make_float4(x[4], x[5], x[6], x[7]);
    }
    __device__ static __forceinline__ float ld1(const float* p) { return *p; }
    __device__ static __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct Vec<bf16v> {
    __device__ static __forceinline__ void ld(const bf16v* p, float (&x)[8]) {
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        const unsigned int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[2 * i] = bf2f((unsigned short)(w[i] & 0xffffu)); x[2 * i + 1] = bf2f((unsigned short)(w[i] >> 16)); }
    }
    __device__ static __forceinline__ void st(bf16v* p, const float (&x)[8]) {
        unsigned int w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = f2bf(x[2 * i]) | ((unsigned int)f2bf(x[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ static __forceinline__ float ld1(const bf16v* p) { return bf2f(p->v); }
    __device__ static __forceinline__ void st1(bf16v* p, float v) { p->v = f2bf(v); }
};

__device__ __forceinline__ float sigmoidf(float w) { return 1.0f / (1.0f + expf(-w)); }

template <typename IO>
__global__ __launch_bounds__(256) void blend_fwd_kernel(const IO* __restrict__ u0, const IO* __restrict__ u,
                                                        const float* __restrict__ skip_weight, IO* __restrict__ out, size_t n) {
    const float s = sigmoidf(*skip_weight), t = 1.0f - s;
    const size_t stride = (size_t)gridDim.x * 256 * 8;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            float a[8], b[8], o[8];
            Vec<IO>::ld(u0 + i, a); Vec<IO>::ld(u + i, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = s * a[k] + t * b[k];
            Vec<IO>::st(out + i, o);
        } else {
            for (size_t j = i; j < n; ++j) Vec<IO>::st1(out + j, s * Vec<IO>::ld1(u0 + j) + t * Vec<IO>::ld1(u + j));
        }
    }
}

template <typename IO>
__global__ __launch_bounds__(256) void blend_bwd_kernel(const IO* __restrict__ g, const IO* __restrict__ u0, const IO* __restrict__ u,
                                                        const float* __restrict__ skip_weight, IO* __restrict__ g_u0,
                                                        IO* __restrict__ g_u, double* __restrict__ part, size_t n) {
    __shared__ double sh[4];
    const float s = sigmoidf(*skip_weight), t = 1.0f - s;
    const size_t stride = (size_t)gridDim.x * 256 * 8;
    double acc = 0.0;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            float gg[8], a[8], b[8], o0[8], o1[8];
            Vec<IO>::ld(g + i, gg); Vec<IO>::ld(u0 + i, a); Vec<IO>::ld(u + i, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) { o0[k] = s * gg[k]; o1[k] = t * gg[k]; acc = fma((double)gg[k], (double)a[k] - (double)b[k], acc); }
            Vec<IO>::st(g_u0 + i, o0); Vec<IO>::st(g_u + i, o1);
        } else {
            for (size_t j = i; j < n; ++j) {
                const float gj = Vec<IO>::ld1(g + j);
                Vec<IO>::st1(g_u0 + j, s * gj); Vec<IO>::st1(g_u + j, t * gj);
                acc = fma((double)gj, (double)Vec<IO>::ld1(u0 + j) - (double)Vec<IO>::ld1(u + j), acc);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void blend_reduce_kernel(const double* __restrict__ part, int nparts,
                                                           const float* __restrict__ skip_weight, float* __restrict__ g_w) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = 1.0 / (1.0 + exp(-(double)*skip_weight));
        *g_w = (float)(s * (1.0 - s) * ((sh[0] + sh[1]) + (sh[2] + sh[3])));
    }
}

int blend_grid(size_t n) {
    const size_t blocks = (n + 2047) / 2048;
    return (int)(blocks < 2048 ? (blocks ? blocks : 1) : 2048);
}

}  // namespace
}  // namespace pde

using namespace pde;

extern "C" {

int pde_skip_blend_forward(int64_t n, int32_t io_dtype, const void* u0, const void* u, const float* skip_weight, void* out,
                           void* stream) {
    if (n <= 0 || !u0 || !u || !skip_weight || !out) return PDE_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = blend_grid((size_t)n);
    if (io_dtype == PDE_IO_F32)
        hipLaunchKernelGGL(blend_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)u0, (const float*)u, skip_weight, (float*)out, (size_t)n);
    else if (io_dtype == PDE_IO_BF16)
        hipLaunchKernelGGL(blend_fwd_kernel<bf16v>, dim3(grid), dim3(256), 0, st, (const bf16v*)u0, (const bf16v*)u, skip_weight, (bf16v*)out, (size_t)n);
    else
        return PDE_E_BADARG;
    return check_launch();
}

size_t pde_skip_blend_backward_workspace_bytes(int64_t n) { return n > 0 ? (size_t)blend_grid((size_t)n) * sizeof(double) : 0; }

int pde_skip_blend_backward(int64_t n, int32_t io_dtype, const void* g, const void* u0, const void* u, const float* skip_weight,
                            void* g_u0, void* g_u, float* g_skip_weight, void* workspace, size_t workspace_bytes, void* stream) {
    if (n <= 0 || !g || !u0 || !u || !skip_weight || !g_u0 || !g_u || !g_skip_weight || !workspace) return PDE_E_BADARG;
    if (workspace_bytes < pde_skip_blend_backward_workspace_bytes(n)) return PDE_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = blend_grid((size_t)n);
    if ((uintptr_t)workspace & 7) return PDE_E_WORKSPACE;
    double* part = static_cast<double*>(workspace);
    if (io_dtype == PDE_IO_F32)
        hipLaunchKernelGGL(blend_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)g, (const float*)u0, (const float*)u, skip_weight, (float*)g_u0, (float*)g_u, part, (size_t)n);
    else if (io_dtype == PDE_IO_BF16)
        hipLaunchKernelGGL(blend_bwd_kernel<bf16v>, dim3(grid), dim3(256), 0, st, (const bf16v*)g, (const bf16v*)u0, (const bf16v*)u, skip_weight, (bf16v*)g_u0, (bf16v*)g_u, part, (size_t)n);
    else
        return PDE_E_BADARG;
    hipLaunchKernelGGL(blend_reduce_kernel, dim3(1), dim3(256), 0, st, part, grid, skip_weight, g_skip_weight);
    return check_launch();
}

}  // extern "C"
