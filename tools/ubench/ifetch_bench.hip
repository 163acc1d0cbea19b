// Micro-benchmark: is straight-line code bound by instruction fetch?  A loop whose body is K
// independent-chain VALU instructions (8 chains), K = 512 .. 16384, with 4-byte (v_fmac_f32 e32) or
// 8-byte (v_fma_f32 e64 / VOP3) encodings, 2 waves per SIMD on every CU.  Prints cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K, bool WIDE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < K / 8; ++i) {
            if (WIDE) {      // VOP3 encoding (8 bytes): clamp modifier forces e64
                asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x3) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x4) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x5) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x6) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x7) : "v"(a), "v"(b));
            } else {         // VOP2 encoding (4 bytes)
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int K, bool WIDE>
void run(float* out, int waves) {
    const int iters = (1 << 22) / K;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<K, WIDE>), dim3(256), dim3(64 * waves), 0, 0, out, iters, 0.5f, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<K, WIDE>), dim3(256), dim3(64 * waves), 0, 0, out, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("body %5d instr x %d B = %6.1f KB, waves/CU=%d: %.2f SIMD-cycles per instruction (%.3f ms)\n", K, WIDE ? 8 : 4,
           K * (WIDE ? 8 : 4) / 1024.0, waves, cyc / ((double)iters * K * waves / 4), ms);
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    for (int waves : {4, 8}) {
        run<512, false>(out, waves); run<2048, false>(out, waves); run<4096, false>(out, waves); run<8192, false>(out, waves); run<16384, false>(out, waves);
        run<512, true>(out, waves); run<2048, true>(out, waves); run<4096, true>(out, waves); run<8192, true>(out, waves);
    }
    return 0;
}
