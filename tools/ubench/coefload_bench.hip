// Micro-benchmark: every wave of a CU streams the same "coefficient records" straight from global
// memory (L1/L2 hits) with coalesced 16-byte loads, as a barrier-free alternative to staging them in
// LDS.  Prints CU cycles per KB per wave and the implied cycles per 12-KB record.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(512) void k(const float4* __restrict__ coef, float* out, int iters, int nrec, int rec_f4,
                                         int loads) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chan = blockIdx.x & 63;
    const float4* base = coef + (size_t)chan * nrec * rec_f4;
    float4 acc = make_float4(0, 0, 0, 0);
    int r = wave % nrec;                       // waves start at different records when skewed
    for (int it = 0; it < iters; ++it) {
        const float4* p = base + (size_t)r * rec_f4 + lane;
        float4 q[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) if (i < loads) q[i] = p[i * 64];
#pragma unroll
        for (int i = 0; i < 12; ++i) if (i < loads) { acc.x += q[i].x; acc.y += q[i].y; acc.z += q[i].z; acc.w += q[i].w; }
        r = (r + 1 == nrec) ? 0 : r + 1;
    }
    out[blockIdx.x * blockDim.x + tid] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    const int nrec = 30, rec_f4 = 12 * 64, C = 64;
    float4* coef; float* out;
    hipMalloc(&coef, (size_t)C * nrec * rec_f4 * sizeof(float4));
    hipMemset(coef, 0, (size_t)C * nrec * rec_f4 * sizeof(float4));
    hipMalloc(&out, 256 * 1024 * 4);
    for (int waves : {4, 8}) for (int loads : {8, 12}) {
        const int iters = 4096;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, coef, out, iters, nrec, rec_f4, loads);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, coef, out, iters, nrec, rec_f4, loads);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("waves/CU=%d loads=%2d KB/wave-iter: %.1f CU-cycles per wave-record, %.2f cycles/KB, chip %.1f TB/s (%.3f ms)\n", waves, loads,
               cyc / ((double)iters * waves), cyc / ((double)iters * waves * loads), 256.0 * iters * waves * loads * 1024 / (ms * 1e-3) / 1e12, ms);
    }
    return 0;
}
