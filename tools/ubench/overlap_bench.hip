// Micro-benchmark: do VALU work and LDS re-layout traffic of DIFFERENT waves of a CU overlap?
// 8 waves per CU (2 per SIMD).  mode 0: all waves VALU chains; mode 1: all waves re-layout;
// mode 2: waves 0-3 VALU, waves 4-7 re-layout (same per-wave work as in modes 0/1).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kStride = 36, kImage = 1152, kHalfPad = 16, M = 16, N = 32;

__device__ __forceinline__ void valu_work(float (&v)[2][M], float e, int reps) {
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int k = 1; k < M; ++k) { v[0][k] = fmaf(e, v[0][k - 1], v[0][k]); v[1][k] = fmaf(e, v[1][k - 1], v[1][k]); }
#pragma unroll
        for (int k = M - 2; k >= 0; --k) { v[0][k] = fmaf(e, v[0][k + 1], v[0][k]); v[1][k] = fmaf(e, v[1][k + 1], v[1][k]); }
    }
}
__device__ __forceinline__ void lds_work(float (&v)[2][M], float* T, int l, int hf, int reps) {
    const int mypos = (l < M) ? l : kHalfPad + (N - 1 - l);
    const int myrow = (l < M) ? l : M + (N - 1 - l);
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* dst = T + hf * M * kStride + mypos;
#pragma unroll
            for (int k = 0; k < M; ++k) dst[k * kStride] = v[j][k];
            __builtin_amdgcn_wave_barrier();
            const float* src = T + myrow * kStride + hf * kHalfPad;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 x = *reinterpret_cast<const float4*>(src + 4 * i);
                v[j][4 * i] = x.x; v[j][4 * i + 1] = x.y; v[j][4 * i + 2] = x.z; v[j][4 * i + 3] = x.w;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}
__global__ __launch_bounds__(512) void k(float* out, int mode, int vreps, int lreps, float e) {
    __shared__ __attribute__((aligned(16))) float sm[8 * kImage];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hf = lane >> 5, l = lane & 31;
    float v[2][M];
    for (int i = 0; i < M; ++i) { v[0][i] = tid + i; v[1][i] = tid - i; }
    float* T = sm + wave * kImage;
    const bool do_valu = mode == 0 || (mode == 2 && wave < 4) || mode == 3;
    const bool do_lds = mode == 1 || (mode == 2 && wave >= 4) || mode == 3;
    if (mode == 3) {                    // every wave alternates (in phase): the kernels' lock-step
        for (int it = 0; it < 64; ++it) { valu_work(v, e, vreps / 64); lds_work(v, T, l, hf, lreps / 64); }
    } else {
        if (do_valu) valu_work(v, e, vreps);
        if (do_lds) lds_work(v, T, l, hf, lreps);
    }
    float s = 0.f;
    for (int i = 0; i < M; ++i) s += v[0][i] + v[1][i];
    out[blockIdx.x * 512 + tid] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int vreps = 4096, lreps = 2048;
    const char* names[] = {"all 8 waves VALU chains (60 fmac x 4096)", "all 8 waves re-layout (2 planes x 2048)",
                           "4 waves VALU + 4 waves re-layout", "all 8 waves alternate VALU / re-layout in phase (full work of both)"};
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, vreps, lreps, 0.999f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, vreps, lreps, 0.999f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-70s %.3f ms = %.0f kcycles\n", names[mode], ms, ms * 2.4e3);
    }
    return 0;
}
