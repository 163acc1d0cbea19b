// Micro-benchmark: what does v_mfma_f32_32x32x2_f32 really sustain, and what does feeding it from LDS cost?
// Grid 256 x WG workgroups of 256 threads (one wave per SIMD per workgroup), every wave runs `reps` groups of 8 MFMAs
// on 2 accumulators; operands (a) constant registers, (b) one ds_read_b32 per operand and MFMA, (c) one ds_read_b128
// per 4 MFMAs and operand.  Wall time from HIP events -> TFLOP/s and the clock that would explain it at 100 % issue
// (4096 flops per MFMA, 64 cycles per MFMA and SIMD); s_memtime gives wave cycles (100 MHz counter) for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k_mfma(float* out, int reps) {
    __shared__ __attribute__((aligned(16))) float sm[64 * 68 * 2];
    for (int e = threadIdx.x; e < 64 * 68 * 2; e += 256) sm[e] = 0.001f * (e & 63);
    __syncthreads();
    const int lane = threadIdx.x & 63, jj = lane & 31, kh = lane >> 5;
    f32x16 acc[2];
    for (int t = 0; t < 2; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = 1.0f + lane * 1e-3f, b = 0.5f;
    const float* pa = sm + jj * 68 + 4 * kh;
    const float* pb = sm + 64 * 68 + jj * 68 + 4 * kh;
    for (int it = 0; it < reps; ++it) {
        const int m = it & 7;
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a0 = pa[8 * m + q], b0 = pb[8 * m + q], b1 = pb[32 * 68 + 8 * m + q];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            }
        } else {
            const float4 a4 = *reinterpret_cast<const float4*>(pa + 8 * m);
            const float4 b4 = *reinterpret_cast<const float4*>(pb + 8 * m);
            const float4 c4 = *reinterpret_cast<const float4*>(pb + 32 * 68 + 8 * m);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, c4.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, c4.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, c4.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, c4.w, acc[1], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 2; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int wg_per_cu, float* out) {
    const int reps = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, reps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_mfma<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double mfma = 256.0 * wg_per_cu * 4 * reps * 8;            // wave-level MFMAs
    const double tf = mfma * 4096 / (ms * 1e-3) / 1e12;
    const double clk = mfma * 64 / 1024 / (ms * 1e-3) / 1e9;         // GHz if the matrix unit never idled
    printf("%-34s %d waves/SIMD: %.3f ms, %.1f TFLOP/s, = %.2f GHz x 100%% MFMA issue\n", name, wg_per_cu, ms, tf, clk);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 3; ++w) {
        run<0>("operands in registers", w, out);
        run<1>("ds_read_b32 per operand and MFMA", w, out);
        run<2>("ds_read_b128 per 4 MFMAs", w, out);
    }
    return 0;
}
