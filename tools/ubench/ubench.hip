// Micro-benchmarks that size design decisions of the sweep kernels (gfx950).
//   ./ubench  -> prints cycles per wave-instruction for LDS float atomics vs stores vs reads and
//                VALU issue cost at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITER = 2048;

template <int MODE>
__global__ void lds_kernel(float* out, int iters) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x;
    float* p = sm + tid;                      // lane-contiguous, conflict-free
    const int stride = blockDim.x;
    float v = tid * 0.5f, acc = 0.f;
    for (int e = tid; e < 16 * stride; e += stride) sm[e] = 0.f;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (MODE == 0) __hip_atomic_fetch_add(p + k * stride, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) p[k * stride] = v;
            if (MODE == 2) acc += p[k * stride];
            if (MODE == 3) { float4 q = *reinterpret_cast<float4*>(sm + (tid * 4 + k * 4 * stride) % (16 * stride)); acc += q.x + q.w; }
        }
        v += 1.0f;
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + tid] = acc + sm[tid];
}

__global__ void valu_kernel(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
            x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void valu_dep_kernel(float* out, int iters, float a, float b) {     // one dependent chain per lane
    float x0 = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) x0 = fmaf(x0, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0;
}

template <typename F>
float time_ms(F&& f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float) * 8);
    const double ghz = 2.4;
    const char* names[] = {"ds_add_f32 (atomic, no return)", "ds_write_b32", "ds_read_b32", "ds_read_b128"};
    for (int waves : {4, 8, 16}) {
        const int threads = 64 * (waves > 16 ? 16 : waves);
        for (int mode = 0; mode < 4; ++mode) {
            const size_t lds = 16 * threads * sizeof(float);
            float ms = 0;
            auto run = [&]() {
                if (mode == 0) hipLaunchKernelGGL(lds_kernel<0>, dim3(256), dim3(threads), lds, 0, out, ITER);
                if (mode == 1) hipLaunchKernelGGL(lds_kernel<1>, dim3(256), dim3(threads), lds, 0, out, ITER);
                if (mode == 2) hipLaunchKernelGGL(lds_kernel<2>, dim3(256), dim3(threads), lds, 0, out, ITER);
                if (mode == 3) hipLaunchKernelGGL(lds_kernel<3>, dim3(256), dim3(threads), lds, 0, out, ITER);
            };
            ms = time_ms(run);
            const double cyc = ms * 1e-3 * ghz * 1e9;                    // per CU (one block per CU)
            const double per_inst = cyc / ((double)ITER * 16 * waves);  // CU cycles per wave-instruction
            printf("%-32s waves/CU=%2d  %.2f CU-cycles per wave-instruction (%.3f ms)\n", names[mode], waves, per_inst, ms);
        }
    }
    for (int wps : {1, 2, 4, 8}) {
        const int threads = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps;
        const int blocks = 256 * ((64 * 4 * wps) / threads);
        float ms = time_ms([&]() { hipLaunchKernelGGL(valu_kernel, dim3(blocks), dim3(threads), 0, 0, out, ITER, 1.0001f, 0.5f); });
        const double cyc = ms * 1e-3 * ghz * 1e9;
        printf("v_fma independent x8: waves/SIMD=%d  %.2f SIMD-cycles per wave-instruction (%.3f ms)\n", wps,
               cyc / ((double)ITER * 64 * wps), ms);
        ms = time_ms([&]() { hipLaunchKernelGGL(valu_dep_kernel, dim3(blocks), dim3(threads), 0, 0, out, ITER, 1.0001f, 0.5f); });
        const double cyc2 = ms * 1e-3 * ghz * 1e9;
        printf("v_fma dependent chain: waves/SIMD=%d  %.2f cycles per instruction per wave, %.2f SIMD-cycles per wave-instruction\n",
               wps, cyc2 / ((double)ITER * 64), cyc2 / ((double)ITER * 64 * wps));
    }
    return 0;
}
