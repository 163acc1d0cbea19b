// Micro-benchmark: the forward x-sweep recurrences (elimination + substitution on J = 4 planes per lane, shared
// coefficients) as 4 scalar chains vs 2 packed chains (v_pk_fma_f32 with the coefficient broadcast by op_sel),
// W waves per SIMD, no LDS.  Cycles per sweep per wave from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int M = 16;

template <int MODE>   // 0: 4 scalar chains; 1: 2 packed chains; 2: 1 packed chain (J = 2); 3: 2 scalar chains (J = 2)
__global__ void k(unsigned long long* cyc, float* out, const float* coef, int reps) {
    extern __shared__ float pad[];
    float e[M], inv[M];
#pragma unroll
    for (int i = 0; i < M; ++i) { e[i] = coef[i] * 0.01f; inv[i] = coef[M + i]; }
    float s[4][M];
    v2f p[2][M];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < M; ++i) s[j][i] = threadIdx.x * 0.001f + i + j;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < M; ++i) p[j][i] = v2f{s[2 * j][i], s[2 * j + 1][i]};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < reps; ++it) {
        if (MODE == 0 || MODE == 3) {
            constexpr int J = MODE == 0 ? 4 : 2;
#pragma unroll
            for (int kk = 0; kk < M; ++kk)
#pragma unroll
                for (int j = 0; j < J; ++j) { const float t = s[j][kk] * inv[kk]; s[j][kk] = kk == 0 ? t : fmaf(e[kk], s[j][kk - 1], t); }
#pragma unroll
            for (int kk = M - 2; kk >= 0; --kk)
#pragma unroll
                for (int j = 0; j < J; ++j) s[j][kk] = fmaf(e[kk], s[j][kk + 1], s[j][kk]);
        } else {
            constexpr int J = MODE == 1 ? 2 : 1;
#pragma unroll
            for (int kk = 0; kk < M; ++kk)
#pragma unroll
                for (int j = 0; j < J; ++j) { const v2f t = p[j][kk] * v2f{inv[kk], inv[kk]}; p[j][kk] = kk == 0 ? t : __builtin_elementwise_fma(v2f{e[kk], e[kk]}, p[j][kk - 1], t); }
#pragma unroll
            for (int kk = M - 2; kk >= 0; --kk)
#pragma unroll
                for (int j = 0; j < J; ++j) p[j][kk] = __builtin_elementwise_fma(v2f{e[kk], e[kk]}, p[j][kk + 1], p[j][kk]);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < M; ++i) acc += s[0][i] + s[1][i] + s[2][i] + s[3][i] + p[0][i].x + p[0][i].y + p[1][i].x + p[1][i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (threadIdx.x == 99999) pad[0] = acc;
}

template <int MODE>
void run(int W, unsigned long long* cyc, float* out, const float* coef) {
    const int reps = 1024;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256 * W), 100 * 1024, 0, cyc, out, coef, reps); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256 * 4 * W);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / reps;
    const char* nm[] = {"4 scalar chains (J=4)", "2 packed chains (J=4)", "1 packed chain  (J=2)", "2 scalar chains (J=2)"};
    const int planes = (MODE == 0 || MODE == 1) ? 4 : 2;
    // SIMD time per plane-sweep: a wave's cycles are shared by the W waves resident on its SIMD
    printf("%s, %d waves/SIMD: %.0f cycles per sweep per wave = %.1f SIMD cycles per plane-sweep\n", nm[MODE], W, med,
           med / (planes * W));
}
int main() {
    unsigned long long* cyc; float* out; float* coef;
    hipMalloc(&cyc, 256 * 16 * 8); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&coef, 128);
    float h[32]; for (int i = 0; i < 32; ++i) h[i] = 0.9f + 0.001f * i; hipMemcpy(coef, h, 128, hipMemcpyHostToDevice);
    for (int W = 1; W <= 4; ++W) { run<0>(W, cyc, out, coef); run<1>(W, cyc, out, coef); run<2>(W, cyc, out, coef); run<3>(W, cyc, out, coef); }
    return 0;
}
