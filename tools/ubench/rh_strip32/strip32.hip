// Micro-benchmark (not part of the product): the strip product of the symmetric layer, P = X K^T for 128 batch rows, as
// 32-column strips on v_mfma_f32_32x32x2_f32 (four waves x 32 rows), the contraction split over SPLIT workgroups per strip
// that leave partial tiles in a workspace.  Question: how far below the 48 us of the 16-column / 16x16x4 kernel
// (pde_rh.hip) does the product itself get when more workgroups are resident and every LDS byte feeds twice the flops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef BK
#define BK 32
#endif
constexpr int LDA = BK + 4, ROWS = 128, COLS = 32, NTH = 256;

template <bool NT>
__global__ __launch_bounds__(NTH) void strip32(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ part,
                                               int B, int D, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ASZ = ROWS * LDA, WSZ = NT ? COLS * LDA : BK * (COLS + 4);
    float* As = smem;
    float* Ws = smem + 2 * ASZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jj = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.x * COLS, split = blockIdx.y;
    const int kb = split * ksplit;
    constexpr int C4 = BK / 4, APT = ROWS * C4 / NTH, WPT = (COLS * BK / 4 + NTH - 1) / NTH;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 pa[APT], pw[WPT];
    auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < APT; ++m) {
            const int f = tid + NTH * m, row = f / C4, c4 = f % C4;
            const int rc = row < B ? row : B - 1;
            pa[m] = *reinterpret_cast<const float4*>(X + (size_t)rc * D + k0 + 4 * c4);
        }
#pragma unroll
        for (int m = 0; m < WPT; ++m) {
            const int f = tid + NTH * m;
            if (NT) {
                const int row = f / C4, c4 = f % C4;
                pw[m] = *reinterpret_cast<const float4*>(W + (size_t)(n0 + row) * D + k0 + 4 * c4);
            } else {
                const int row = f / (COLS / 4), c4 = f % (COLS / 4);
                pw[m] = *reinterpret_cast<const float4*>(W + (size_t)(k0 + row) * D + n0 + 4 * c4);
            }
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        float* A = As + buf * ASZ;
        float* Wb = Ws + buf * WSZ;
#pragma unroll
        for (int m = 0; m < APT; ++m) {
            const int f = tid + NTH * m, row = f / C4, c4 = f % C4;
            *reinterpret_cast<float4*>(A + row * LDA + 4 * c4) = row < B ? pa[m] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int m = 0; m < WPT; ++m) {
            const int f = tid + NTH * m;
            if (NT) {
                const int row = f / C4, c4 = f % C4;
                *reinterpret_cast<float4*>(Wb + row * LDA + 4 * c4) = pw[m];
            } else {
                const int row = f / (COLS / 4), c4 = f % (COLS / 4);
                *reinterpret_cast<float4*>(Wb + row * (COLS + 4) + 4 * c4) = pw[m];
            }
        }
    };
    fetch(kb);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < ksplit; k0 += BK) {
        const bool more = k0 + BK < ksplit;
        if (more) fetch(kb + k0 + BK);
        const float* A = As + buf * ASZ;
        const float* Wb = Ws + buf * WSZ;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {                // 8 k's = 4 MFMAs: half-wave kh takes k = 8g + 4kh + s in step s
            const float4 av = *reinterpret_cast<const float4*>(A + (wave * 32 + jj) * LDA + 8 * g + 4 * kh);
            float b[4];
            if (NT) {
                const float4 bv = *reinterpret_cast<const float4*>(Wb + jj * LDA + 8 * g + 4 * kh);
                b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) b[s] = Wb[(8 * g + 4 * kh + s) * (COLS + 4) + jj];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b[3], acc, 0, 0, 0);
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // partial tile, lane-major: [strip][split][wave][r/4][lane] float4
    float4* dst = reinterpret_cast<float4*>(part) + (((size_t)blockIdx.x * gridDim.y + split) * 4 + wave) * 4 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q * 64] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = 128, D = 3072;
    std::vector<float> hx((size_t)B * D), hk((size_t)D * D);
    srand(1);
    for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hk) v = (rand() % 2001 - 1000) / 1000.f;
    float *X, *K, *part;
    CK(hipMalloc(&X, hx.size() * 4)); CK(hipMalloc(&K, hk.size() * 4));
    CK(hipMalloc(&part, (size_t)(D / COLS) * 32 * 4 * 64 * 16 * 4));
    CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(K, hk.data(), hk.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nt = 1; nt >= 0; --nt)
        for (int S : {4, 8, 12, 16, 24}) {
            const int ks = D / S;
            const size_t lds = (size_t)(2 * ROWS * LDA + 2 * (nt ? COLS * LDA : BK * (COLS + 4))) * 4;
            auto kern = nt ? strip32<true> : strip32<false>;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(D / COLS, S), dim3(NTH), lds, 0, X, K, part, B, D, ks);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(D / COLS, S), dim3(NTH), lds, 0, X, K, part, B, D, ks);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // check one element of the combined tile against the host (NT only)
            std::vector<float> hp((size_t)S * 4 * 4 * 64 * 4);
            CK(hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost));   // strip 0
            double got = 0, ref = 0;
            // wave 0, lane 0, r = 0: row 0, col 0
            for (int s = 0; s < S; ++s) got += hp[(((size_t)s * 4 + 0) * 4 * 64 + 0) * 4 + 0];
            for (int k = 0; k < D; ++k) ref += (double)hx[k] * (nt ? hk[k] : hk[(size_t)k * D]);
            printf("%s BK=%d split %d: %.1f us per launch (%.1f TFLOP/s)  P[0][0] %.4f ref %.4f  lds %zu\n", nt ? "NT" : "NN", BK, S,
                   ms / 20 * 1e3, 2.0 * B * D * D / (ms / 20 * 1e-3) / 1e12, got, ref, lds);
        }
    return 0;
}
