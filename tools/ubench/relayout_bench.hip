// Micro-benchmark of the wave-private 32x32 row<->column re-layout through LDS (gfx950): CU cycles per
// plane for the variants considered for the sweep kernels.  One workgroup per CU, W waves.
//   V0  16 x ds_write_b32 (column scatter)  + 4 x ds_read_b128 (row)      <- shipped
//   V1   4 x ds_write_b128 (row)            + 16 x ds_read_b32 (column gather)
//   V2  two planes as float2: 16 x ds_write_b64 + 8 x ds_read_b128
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kStride = 36, kImage = 1152, kHalfPad = 16, M = 16, N = 32;

template <int V>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l = lane & 31;
    float* T = sm + wave * kImage * (V == 2 ? 2 : 1);
    float v[M], w[M];
    for (int i = 0; i < M; ++i) { v[i] = tid + i; w[i] = tid - i; }
    const int mypos = (l < M) ? l : kHalfPad + (N - 1 - l);
    const int myrow = (l < M) ? l : M + (N - 1 - l);
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {
            float* dst = T + hf * M * kStride + mypos;
#pragma unroll
            for (int k = 0; k < M; ++k) dst[k * kStride] = v[k];
            __builtin_amdgcn_wave_barrier();
            const float* src = T + myrow * kStride + hf * kHalfPad;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 x = *reinterpret_cast<const float4*>(src + 4 * i);
                v[4 * i] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
            }
            __builtin_amdgcn_wave_barrier();
        } else if (V == 1) {
            float* dst = T + myrow * kStride + hf * kHalfPad;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(dst + 4 * i) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
            __builtin_amdgcn_wave_barrier();
            const float* src = T + hf * M * kStride + mypos;
#pragma unroll
            for (int k = 0; k < M; ++k) v[k] = src[k * kStride];
            __builtin_amdgcn_wave_barrier();
        } else {
            float2* dst = reinterpret_cast<float2*>(T) + hf * M * kStride + mypos;
#pragma unroll
            for (int k = 0; k < M; ++k) dst[k * kStride] = make_float2(v[k], w[k]);
            __builtin_amdgcn_wave_barrier();
            const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<float2*>(T) + myrow * kStride + hf * kHalfPad);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 x = src[i];
                v[2 * i] = x.x; w[2 * i] = x.y; v[2 * i + 1] = x.z; w[2 * i + 1] = x.w;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    float s = 0.f;
    for (int i = 0; i < M; ++i) s += v[i] + w[i];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int V>
void run(float* out, int waves) {
    const int iters = 4096;
    const size_t lds = (size_t)waves * kImage * 4 * (V == 2 ? 2 : 1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(64 * waves), lds, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(64 * waves), lds, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9;
    const double planes = (double)iters * waves * (V == 2 ? 2 : 1);
    printf("V%d waves/CU=%2d: %.1f CU-cycles per plane re-layout (%.3f ms)\n", V, waves, cyc / planes, ms);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    for (int w : {1, 4, 8, 16}) { run<0>(out, w); run<1>(out, w); if (w <= 8) run<2>(out, w); }
    return 0;
}
