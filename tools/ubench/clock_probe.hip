// What does the shader clock do under (a) fp32 MFMA chains alone, (b) HBM streaming alone, (c) both together?
// Every wave counts its own core cycles (s_memtime) over the launch; the host divides by the wall time of the launch
// (HIP events).  768 workgroups x 256 threads (3 per CU), like the fused mixing backward.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: MFMA only, 1: streaming only, 2: both
__global__ __launch_bounds__(256) void k_probe(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4, float* out,
                                               unsigned long long* cyc, int reps) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    const int lane = threadIdx.x & 63;
    f32x16 acc[2];
    for (int t = 0; t < 2; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = 1.0f + lane * 1e-3f, b = 0.5f;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (int it = 0; it < reps; ++it) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE != 0) { v = src[i % n4]; }
        if (MODE != 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
            }
        }
        if (MODE != 0) { s.x += v.x; dst[i % n4] = v; i += stride; }
    }
    float r = s.x;
    for (int t = 0; t < 2; ++t)
        for (int q = 0; q < 16; ++q) r += acc[t][q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, const float4* src, float4* dst, size_t n4, float* out, unsigned long long* cyc, int reps) {
    const int G = 768;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(G), dim3(256), 0, 0, src, dst, n4, out, cyc, reps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(G), dim3(256), 0, 0, src, dst, n4, out, cyc, reps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(G * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2], mx = (double)h.back();
    const double bytes = (MODE != 0) ? 2.0 * G * 256 * 16.0 * reps : 0.0;
    const double mf = (MODE != 1) ? (double)G * 4 * reps * 16 : 0.0;
    printf("%-28s %.3f ms: counter %.0f (median) / %.0f (max) ticks per wave -> %.2f GHz if ticks are shader cycles; "
           "%.2f TB/s, %.1f TFLOP/s\n", name, ms, med, mx, mx / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12,
           mf * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t n4 = (size_t)64 << 20;               // 1 GiB source, 1 GiB destination
    float4 *src, *dst; float* out; unsigned long long* cyc;
    (void)hipMalloc(&src, n4 * 16); (void)hipMalloc(&dst, n4 * 16); (void)hipMalloc(&out, 768 * 256 * 4); (void)hipMalloc(&cyc, 768 * 4 * 8);
    (void)hipMemset(src, 0, n4 * 16);
    run<0>("MFMA chains only", src, dst, n4, out, cyc, 4096);
    run<1>("streaming only", src, dst, n4, out, cyc, 4096);
    run<2>("MFMA chains + streaming", src, dst, n4, out, cyc, 4096);
    run<0>("MFMA chains only (again)", src, dst, n4, out, cyc, 4096);
    return 0;
}
