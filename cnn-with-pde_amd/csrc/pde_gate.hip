// Epilogue of cifar10.MultiScaleExtractor (cifar10.py:270-280) behind the PDE layers, SURVEY.md §8f-3:
//     features_i = y_i * gate_i[b,c]            (SpatialAttention, cifar10.py:232-244; gate = MLP(avgpool(y_i + pos)))
//     combined   = sum_i w_i features_i         (softmax weights, cifar10.py:277-280)
// The average pool comes out of the PDE kernel itself (PdeSmallLayer.plane_sums); the small (B,C) MLP stays in torch;
// this file is the two passes over the full tensors that remain:
//   forward : combined[b,c,p] = sum_i w_i g_i[b,c] y_i[b,c,p]                           (one pass instead of 3 muls + 2 adds)
//   backward: gy_i[b,c,p] = w_i g_i[b,c] g[b,c,p],   dot_i[b,c] = sum_p g[b,c,p] y_i[b,c,p]   (one pass; the dots feed the
//             gradients of the gates and of the weights: dL/dg_i = w_i dot_i, dL/dw_i = sum_bc g_i dot_i)
// One wave per plane; 16-byte accesses; sums reduced in a fixed order (no atomics).
#include "pde_common.h"

namespace pde {
namespace {

constexpr int kGateMaxL = 4;
struct GateArgs {
    const void* y[kGateMaxL];        // L tensors (B,C,HW)
    const float* gate[kGateMaxL];    // L arrays (B,C)
    void* gy[kGateMaxL];             // bwd: L outputs (B,C,HW)
    float* dot[kGateMaxL];           // bwd: L outputs (B,C)
    const float* w;                  // (L) device
    const void* g;                   // bwd: dL/dcombined
    void* out;                       // fwd
    int L, planes, HW;
};

struct bf16g { unsigned short v; };
template <typename IO> struct G4;
template <> struct G4<float> {
    __device__ static __forceinline__ float4 ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static __forceinline__ void st(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct G4<bf16g> {
    __device__ static __forceinline__ float4 ld(const bf16g* p) {
        const ushort4 q = *reinterpret_cast<const ushort4*>(p);
        return make_float4(__uint_as_float((unsigned)q.x << 16), __uint_as_float((unsigned)q.y << 16),
                           __uint_as_float((unsigned)q.z << 16), __uint_as_float((unsigned)q.w << 16));
    }
    __device__ static __forceinline__ void st(bf16g* p, float4 v) {
        ushort4 q;
        q.x = f32_to_bf16_hw(v.x); q.y = f32_to_bf16_hw(v.y); q.z = f32_to_bf16_hw(v.z); q.w = f32_to_bf16_hw(v.w);
        *reinterpret_cast<ushort4*>(p) = q;
    }
};

template <typename IO>
__global__ __launch_bounds__(256) void gate_combine_fwd_kernel(GateArgs a) {
    const int lane = threadIdx.x & 63;
    const int pc = blockIdx.x * 4 + (threadIdx.x >> 6);          // one wave per plane
    if (pc >= a.planes) return;
    float f[kGateMaxL];
#pragma unroll
    for (int i = 0; i < kGateMaxL; ++i) f[i] = (i < a.L) ? a.w[i] * a.gate[i][pc] : 0.f;
    const size_t base = (size_t)pc * a.HW;
    IO* out = static_cast<IO*>(a.out) + base;
    for (int e = 4 * lane; e < a.HW; e += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < kGateMaxL; ++i) {
            if (i < a.L) {
                const float4 v = G4<IO>::ld(static_cast<const IO*>(a.y[i]) + base + e);
                acc.x = fmaf(f[i], v.x, acc.x); acc.y = fmaf(f[i], v.y, acc.y);
                acc.z = fmaf(f[i], v.z, acc.z); acc.w = fmaf(f[i], v.w, acc.w);
            }
        }
        G4<IO>::st(out + e, acc);
    }
}

template <typename IO>
__global__ __launch_bounds__(256) void gate_combine_bwd_kernel(GateArgs a) {
    const int lane = threadIdx.x & 63;
    const int pc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pc >= a.planes) return;
    float f[kGateMaxL], d[kGateMaxL];
#pragma unroll
    for (int i = 0; i < kGateMaxL; ++i) { f[i] = (i < a.L) ? a.w[i] * a.gate[i][pc] : 0.f; d[i] = 0.f; }
    const size_t base = (size_t)pc * a.HW;
    const IO* g = static_cast<const IO*>(a.g) + base;
    for (int e = 4 * lane; e < a.HW; e += 256) {
        const float4 gv = G4<IO>::ld(g + e);
#pragma unroll
        for (int i = 0; i < kGateMaxL; ++i) {
            if (i < a.L) {
                const float4 v = G4<IO>::ld(static_cast<const IO*>(a.y[i]) + base + e);
                d[i] += gv.x * v.x + gv.y * v.y + gv.z * v.z + gv.w * v.w;
                G4<IO>::st(static_cast<IO*>(a.gy[i]) + base + e, make_float4(f[i] * gv.x, f[i] * gv.y, f[i] * gv.z, f[i] * gv.w));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kGateMaxL; ++i) {
        if (i < a.L) {
            float v = d[i];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) a.dot[i][pc] = v;
        }
    }
}

int gate_check(int L, int B, int C, int HW, int io) {
    if (L < 1 || L > kGateMaxL || B <= 0 || C <= 0 || HW <= 0 || (HW % 4) != 0) return PDE_E_BADARG;
    if (io != PDE_IO_F32 && io != PDE_IO_BF16) return PDE_E_BADARG;
    return PDE_OK;
}

}  // namespace
}  // namespace pde

using namespace pde;

extern "C" {

int pde_gate_combine_forward(int32_t L, int32_t B, int32_t C, int32_t HW, int32_t io_dtype, const void* const* ys,
                             const float* const* gates, const float* weights, void* out, void* stream) {
    int rc = gate_check(L, B, C, HW, io_dtype);
    if (rc != PDE_OK) return rc;
    if (!ys || !gates || !weights || !out) return PDE_E_BADARG;
    GateArgs a{};
    for (int i = 0; i < L; ++i) {
        if (!ys[i] || !gates[i]) return PDE_E_BADARG;
        a.y[i] = ys[i]; a.gate[i] = gates[i];
    }
    a.w = weights; a.out = out; a.L = L; a.planes = B * C; a.HW = HW;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((a.planes + 3) / 4), block(256);
    if (io_dtype == PDE_IO_F32) hipLaunchKernelGGL((gate_combine_fwd_kernel<float>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gate_combine_fwd_kernel<bf16g>), grid, block, 0, st, a);
    return check_launch();
}

int pde_gate_combine_backward(int32_t L, int32_t B, int32_t C, int32_t HW, int32_t io_dtype, const void* g,
                              const void* const* ys, const float* const* gates, const float* weights, void* const* gys,
                              float* const* dots, void* stream) {
    int rc = gate_check(L, B, C, HW, io_dtype);
    if (rc != PDE_OK) return rc;
    if (!g || !ys || !gates || !weights || !gys || !dots) return PDE_E_BADARG;
    GateArgs a{};
    for (int i = 0; i < L; ++i) {
        if (!ys[i] || !gates[i] || !gys[i] || !dots[i]) return PDE_E_BADARG;
        a.y[i] = ys[i]; a.gate[i] = gates[i]; a.gy[i] = gys[i]; a.dot[i] = dots[i];
    }
    a.w = weights; a.g = g; a.L = L; a.planes = B * C; a.HW = HW;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((a.planes + 3) / 4), block(256);
    if (io_dtype == PDE_IO_F32) hipLaunchKernelGGL((gate_combine_bwd_kernel<float>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gate_combine_bwd_kernel<bf16g>), grid, block, 0, st, a);
    return check_launch();
}

}  // extern "C"
