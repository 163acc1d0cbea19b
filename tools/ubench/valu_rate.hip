// Micro-benchmark: VALU issue rate of one SIMD as a function of (waves per SIMD, independent dependency chains
// per wave).  Answers: is a wave64 v_fma_f32 2 or 4 cycles of SIMD time, and how many waves / chains does it take
// to reach the SIMD's rate?  One workgroup per CU (grid 256, LDS padding keeps a second one out), W waves per
// SIMD, each wave runs K dependent FMA chains round-robin.  Cycles from s_memtime inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int K, bool PK>
__global__ void k_chain(unsigned long long* cyc, float* out, int reps, float e, float c) {
    extern __shared__ float pad[];
    typedef float v2f __attribute__((ext_vector_type(2)));
    float v[K];
    v2f w[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { v[i] = threadIdx.x * 0.001f + i; w[i] = v2f{v[i], v[i] + 1.f}; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (PK) w[i] = __builtin_elementwise_fma(v2f{e, e}, w[i], v2f{c, c});
                else v[i] = fmaf(e, v[i], c);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < K; ++i) s += PK ? (w[i].x + w[i].y) : v[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (threadIdx.x == 99999) pad[0] = s;
}

template <int K, bool PK>
void run(int W, unsigned long long* cyc, float* out) {
    const int reps = 2048;
    const int threads = 256 * W;
    hipFuncSetAttribute((const void*)k_chain<K, PK>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_chain<K, PK>), dim3(256), dim3(threads), 100 * 1024, 0, cyc, out, reps, 0.999f, 0.001f);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(256 * 4 * W);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double n = (double)reps * 8 * K;                   // VALU instructions per wave
    printf("%s K=%d chains, %d waves/SIMD: %.0f cycles/wave median -> %.2f cyc per instr per wave, %.2f cyc of SIMD per wave-instr\n",
           PK ? "v_pk_fma_f32" : "v_fma_f32   ", K, W, med, med / n, med / (n * W));
}

int main() {
    unsigned long long* cyc; float* out;
    hipMalloc(&cyc, 256 * 16 * 8); hipMalloc(&out, 256 * 1024 * 4);
    for (int W = 1; W <= 4; ++W) {
        run<1, false>(W, cyc, out); run<2, false>(W, cyc, out); run<4, false>(W, cyc, out); run<8, false>(W, cyc, out);
    }
    for (int W = 1; W <= 4; W *= 2) { run<2, true>(W, cyc, out); run<8, true>(W, cyc, out); }
    return 0;
}
