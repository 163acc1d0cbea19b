// Micro-benchmark: what stops TWO waves of a SIMD from overlapping their VALU issue?  The backward sweep kernel
// runs 2 waves/SIMD and gets the throughput of one (profiles/README.md), while pure VOP2 chains overlap perfectly
// (sweep_pk).  Variants of an x-sweep-like body (J = 2 planes, adjoint-solve + state-update arithmetic):
//   0  registers only (e/inv/kap loaded once)
//   1  + the three coefficient arrays re-read from LDS every sweep (12 ds_read_b128 + waits)
//   2  + 64 accumulator registers live (acc += g*q) -> ~200 VGPRs
//   3  1 + 2
//   4  3 + s_nop/s_waitcnt-style scalar noise (8 s_nop per sweep)
// W waves per SIMD (1..2 with the big footprint), one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int M = 16;

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* cyc, float* out, const float* coef, int reps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    constexpr bool kLds = (MODE == 1 || MODE >= 3), kAcc = MODE >= 2, kNoise = MODE == 4;
    for (int i = threadIdx.x; i < 3 * 64 * M; i += blockDim.x) lds[i] = coef[i % 32] * (i < 64 * M ? 0.01f : 1.0f);
    float e[M], inv[M], kap[M];
#pragma unroll
    for (int i = 0; i < M; ++i) { e[i] = coef[i] * 0.01f; inv[i] = coef[M + i]; kap[i] = coef[i] * 0.02f; }
    float r[2][M], x[2][M], acc[4][M];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < M; ++i) { r[j][i] = threadIdx.x * 0.001f + i + j; x[j][i] = r[j][i] * 0.5f; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < M; ++i) acc[a][i] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < reps; ++it) {
        asm volatile("" ::: "memory");
        if (kLds) {
            const float4* p = reinterpret_cast<const float4*>(lds) + lane * 4;       // 64 B per lane per array
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = p[q]; e[4*q] = v.x; e[4*q+1] = v.y; e[4*q+2] = v.z; e[4*q+3] = v.w; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = p[256 + q]; inv[4*q] = v.x; inv[4*q+1] = v.y; inv[4*q+2] = v.z; inv[4*q+3] = v.w; }
        }
        // adjoint solve: H pass, G pass, scaling (3 ops per element)
#pragma unroll
        for (int kk = 1; kk < M; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) r[j][kk] = fmaf(e[kk - 1], r[j][kk - 1], r[j][kk]);
#pragma unroll
        for (int kk = M - 2; kk >= 0; --kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) r[j][kk] = fmaf(e[kk + 1], r[j][kk + 1], r[j][kk]);
#pragma unroll
        for (int kk = 0; kk < M; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) r[j][kk] *= inv[kk];
        if (kLds) {
            const float4* p = reinterpret_cast<const float4*>(lds) + lane * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = p[512 + q]; kap[4*q] = v.x; kap[4*q+1] = v.y; kap[4*q+2] = v.z; kap[4*q+3] = v.w; }
        }
        if (kNoise) { asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0"); }
        // state update (4 ops per element): q = 2x - x[k-1] - x[k+1]; acc += g q; x += kap q
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float xn = 0.f;
#pragma unroll
            for (int kk = M - 1; kk >= 0; --kk) {
                const float xo = x[j][kk];
                const float q = (kk == 0) ? xo - xn : fmaf(2.0f, xo, -x[j][kk - 1]) - xn;
                if (kAcc) acc[0][kk] = fmaf(r[j][kk], q, acc[0][kk]);
                else r[j][kk] = fmaf(r[j][kk], q, 1e-3f);
                x[j][kk] = fmaf(kap[kk], q, xo);
                xn = xo;
            }
        }
        if (kAcc) {
#pragma unroll
            for (int kk = 0; kk < M; ++kk) { acc[1][kk] = fmaf(0.5f, acc[0][kk], acc[1][kk]); acc[3][kk] = fmaf(0.5f, acc[2][kk], acc[3][kk]); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < M; ++i) s += r[0][i] + r[1][i] + x[0][i] + x[1][i] + acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
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
    printf("mode %d, %d waves/SIMD: %.0f cycles per sweep per wave = %.0f SIMD cycles per wave-sweep\n", MODE, W, med, med / W);
}
int main() {
    unsigned long long* cyc; float* out; float* coef;
    hipMalloc(&cyc, 256 * 16 * 8); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&coef, 128);
    float h[32]; for (int i = 0; i < 32; ++i) h[i] = 0.9f + 0.001f * i; hipMemcpy(coef, h, 128, hipMemcpyHostToDevice);
    for (int W = 1; W <= 2; ++W) { run<0>(W, cyc, out, coef); run<1>(W, cyc, out, coef); run<2>(W, cyc, out, coef); run<3>(W, cyc, out, coef); run<4>(W, cyc, out, coef); }
    return 0;
}
