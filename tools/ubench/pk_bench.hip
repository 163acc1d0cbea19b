// Micro-benchmark: issue cost of v_fmac_f32 vs v_pk_fma_f32 in the sweep kernels' dependency pattern
// (a recurrence along k on two planes: two independent scalar chains vs ONE packed chain), 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int M = 16;
__global__ __launch_bounds__(512) void k_scalar(float* out, int reps, float e) {
    float v[2][M];
    for (int i = 0; i < M; ++i) { v[0][i] = threadIdx.x + i; v[1][i] = threadIdx.x - i; }
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int k = 1; k < M; ++k) { v[0][k] = fmaf(e, v[0][k - 1], v[0][k]); v[1][k] = fmaf(e, v[1][k - 1], v[1][k]); }
#pragma unroll
        for (int k = M - 2; k >= 0; --k) { v[0][k] = fmaf(e, v[0][k + 1], v[0][k]); v[1][k] = fmaf(e, v[1][k + 1], v[1][k]); }
    }
    float s = 0.f;
    for (int i = 0; i < M; ++i) s += v[0][i] + v[1][i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k_packed(float* out, int reps, float e, const float* ev) {
    v2f v[M];
    float ee[M];
    for (int i = 0; i < M; ++i) { v[i] = v2f{(float)(threadIdx.x + i), (float)(threadIdx.x - i)}; ee[i] = ev[i] * e; }
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int k = 1; k < M; ++k) v[k] = __builtin_elementwise_fma(v2f{ee[k], ee[k]}, v[k - 1], v[k]);
#pragma unroll
        for (int k = M - 2; k >= 0; --k) v[k] = __builtin_elementwise_fma(v2f{ee[k], ee[k]}, v[k + 1], v[k]);
    }
    float s = 0.f;
    for (int i = 0; i < M; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    float* ev; hipMalloc(&ev, 64); float h[16]; for (int i = 0; i < 16; ++i) h[i] = 1.0f - i * 1e-3f; hipMemcpy(ev, h, 64, hipMemcpyHostToDevice);
    const int reps = 4096;
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto go = [&]() { if (mode == 0) hipLaunchKernelGGL(k_scalar, dim3(256), dim3(512), 0, 0, out, reps, 0.999f);
                          else hipLaunchKernelGGL(k_packed, dim3(256), dim3(512), 0, 0, out, reps, 0.999f, ev); };
        go(); hipDeviceSynchronize();
        hipEventRecord(e0); go(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms = %.0f kcycles for %d element-updates per lane pair\n", mode ? "packed (v_pk_fma_f32)" : "scalar (v_fmac_f32 x2)", ms, ms * 2.4e3, reps * 30);
    }
    return 0;
}
