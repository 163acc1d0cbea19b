// What follows the PDE feature extractor in cifar10.CIFAR10PDENoConv (cifar10.py:346-353), SURVEY.md §8f-3:
//     features = BatchNorm2d(combined)                      (:346, training mode: statistics of the batch)
//     pooled   = cat([AdaptiveAvgPool2d(4x4)(features), AdaptiveMaxPool2d(4x4)(features)], dim=1)      (:349-353)
// In torch: a statistics pass, a normalisation pass that writes `features`, two pooling passes that read it and a
// concatenation.  Here `features` is never written: one pass for the per-plane statistics (mean and centred second
// moment, combined per channel in a fixed order in double precision), one pass that normalises in registers and writes
// the (B, 2C, 4, 4) result with the arg-max positions; the backward is one pass for the two per-channel sums of the
// BatchNorm gradient and one pass for dL/dcombined.  fp32, planes N x N with N a multiple of 4 (uniform pooling windows).
// One wave per plane; reductions in a fixed order (no atomics).
#include "pde_common.h"

namespace pde {
namespace {

constexpr int kTailMaxN = 64;

struct TailArgs {
    const float* x;         // (B,C,N,N)
    const float* gamma;     // (C) or null (= 1)
    const float* beta;      // (C) or null (= 0)
    float* mean;            // (C): training: written by the finalize kernel; eval: the running mean
    float* invstd;          // (C)
    float* run_mean;        // (C) or null: updated in training
    float* run_var;         // (C) or null
    float* part;            // (B*C, 2): per-plane (mean, M2) forward; (sum dz, sum dz xhat) backward
    float* out;             // fwd (B,2C,4,4)
    int* argmax;            // (B,C,4,4): flattened position inside the plane
    const float* gout;      // bwd (B,2C,4,4)
    float* gx;              // bwd (B,C,N,N)
    float* ggamma;          // bwd (C)
    float* gbeta;           // bwd (C)
    int B, C, N, training;
    float eps, momentum;
};

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// per-plane mean and centred second moment (two passes over registers: no cancellation)
__global__ __launch_bounds__(256) void tail_plane_stats_kernel(TailArgs a) {
    const int lane = threadIdx.x & 63, pc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pc >= a.B * a.C) return;
    const int HW = a.N * a.N;
    const float* p = a.x + (size_t)pc * HW;
    float s = 0.f;
    for (int e = 4 * lane; e < HW; e += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + e);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float m = wave_sum(s) / (float)HW;
    float q = 0.f;
    for (int e = 4 * lane; e < HW; e += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + e);
        const float d0 = v.x - m, d1 = v.y - m, d2 = v.z - m, d3 = v.w - m;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    q = wave_sum(q);
    if (lane == 0) { a.part[2 * pc] = m; a.part[2 * pc + 1] = q; }
}

// per channel: one wave; lane i combines the planes b = i, i+64, ... in order (Chan's update, double precision), then the
// 64 partial (count, mean, M2) triples meet in a butterfly — a fixed order; running statistics by lane 0
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, 64); hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(64) void tail_channel_stats_kernel(TailArgs a) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const double n1 = (double)a.N * a.N;
    double cnt = 0.0, mean = 0.0, M2 = 0.0;
    for (int b = lane; b < a.B; b += 64) {
        const double mb = a.part[2 * (b * a.C + c)], qb = a.part[2 * (b * a.C + c) + 1];
        const double delta = mb - mean, tot = cnt + n1;
        mean += delta * n1 / tot;
        M2 += qb + delta * delta * cnt * n1 / tot;
        cnt = tot;
    }
    for (int m = 1; m < 64; m <<= 1) {
        const double c2 = shfl_xor_f64(cnt, m), m2 = shfl_xor_f64(mean, m), q2 = shfl_xor_f64(M2, m);
        const double tot = cnt + c2;
        if (tot > 0.0) {
            // symmetric in the two partners: both lanes of a pair end up with the same triple
            const double delta = m2 - mean;
            const double nm = (cnt * mean + c2 * m2) / tot;
            M2 = M2 + q2 + delta * delta * cnt * c2 / tot;
            mean = nm;
            cnt = tot;
        }
    }
    if (lane != 0) return;
    const double var = M2 / cnt;
    a.mean[c] = (float)mean;
    a.invstd[c] = (float)(1.0 / sqrt(var + (double)a.eps));
    if (a.run_mean != nullptr) a.run_mean[c] = (1.0f - a.momentum) * a.run_mean[c] + a.momentum * (float)mean;
    if (a.run_var != nullptr) a.run_var[c] = (1.0f - a.momentum) * a.run_var[c] + a.momentum * (float)(M2 / (cnt > 1.0 ? cnt - 1.0 : 1.0));
}

// eval mode: invstd from the running variance
__global__ __launch_bounds__(64) void tail_eval_stats_kernel(TailArgs a) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= a.C) return;
    a.mean[c] = a.run_mean[c];
    a.invstd[c] = 1.0f / sqrtf(a.run_var[c] + a.eps);
}

// one wave per plane: normalise, 4x4 average and max pooling (window (N/4)^2), arg-max positions
__global__ __launch_bounds__(256) void tail_pool_fwd_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float img[];      // [4 waves][N*N]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, pc = blockIdx.x * 4 + wv;
    if (pc >= a.B * a.C) return;
    const int N = a.N, HW = N * N, c = pc % a.C, b = pc / a.C, win = N / 4;
    const float sc = a.invstd[c] * (a.gamma ? a.gamma[c] : 1.0f);
    const float sh = (a.beta ? a.beta[c] : 0.0f) - a.mean[c] * sc;
    const float* p = a.x + (size_t)pc * HW;
    float* z = img + (size_t)wv * HW;
    for (int e = 4 * lane; e < HW; e += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + e);
        *reinterpret_cast<float4*>(z + e) = make_float4(fmaf(v.x, sc, sh), fmaf(v.y, sc, sh), fmaf(v.z, sc, sh), fmaf(v.w, sc, sh));
    }
    __builtin_amdgcn_wave_barrier();
    // lane = 4 * window + quarter: a quarter of the window's rows each, then the four lanes meet
    const int wd = lane >> 2, qt = lane & 3, wi = wd >> 2, wj = wd & 3;
    float s = 0.f, mx = -INFINITY;
    int am = 0;
    for (int r = qt; r < win; r += 4) {
        const int h = wi * win + r;
        for (int k = 0; k < win; ++k) {
            const int pos = h * N + wj * win + k;
            const float v = z[pos];
            s += v;
            if (v > mx) { mx = v; am = pos; }          // first maximum in row-major order (as torch)
        }
    }
    for (int o = 1; o < 4; o <<= 1) {
        s += __shfl_xor(s, o, 64);
        const float m2 = __shfl_xor(mx, o, 64);
        const int a2 = __shfl_xor(am, o, 64);
        if (m2 > mx || (m2 == mx && a2 < am)) { mx = m2; am = a2; }
    }
    if (qt == 0) {
        a.out[((size_t)b * 2 * a.C + c) * 16 + wd] = s / (float)(win * win);
        a.out[((size_t)b * 2 * a.C + a.C + c) * 16 + wd] = mx;
        a.argmax[(size_t)pc * 16 + wd] = am;
    }
}

// dz of an element: its window's average share plus the max share when it is the arg-max
__device__ __forceinline__ float tail_dz(const float* ga, const float* gm, const int* am, int pos, int N, int win, float inv_area) {
    const int h = pos / N, w = pos % N, wd = (h / win) * 4 + (w / win);
    return ga[wd] * inv_area + (am[wd] == pos ? gm[wd] : 0.f);
}

// backward pass 1: per plane sum dz and sum dz * xhat
__global__ __launch_bounds__(256) void tail_bwd_sums_kernel(TailArgs a) {
    __shared__ float ga[4][16], gm[4][16];
    __shared__ int am[4][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, pc = blockIdx.x * 4 + wv;
    if (pc >= a.B * a.C) return;
    const int N = a.N, HW = N * N, c = pc % a.C, b = pc / a.C, win = N / 4;
    if (lane < 16) {
        ga[wv][lane] = a.gout[((size_t)b * 2 * a.C + c) * 16 + lane];
        gm[wv][lane] = a.gout[((size_t)b * 2 * a.C + a.C + c) * 16 + lane];
        am[wv][lane] = a.argmax[(size_t)pc * 16 + lane];
    }
    __builtin_amdgcn_wave_barrier();
    const float mean = a.mean[c], is = a.invstd[c], inv_area = 1.0f / (float)(win * win);
    const float* p = a.x + (size_t)pc * HW;
    float s0 = 0.f, s1 = 0.f;
    for (int e = 4 * lane; e < HW; e += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + e);
        const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dz = tail_dz(ga[wv], gm[wv], am[wv], e + k, N, win, inv_area);
            s0 += dz;
            s1 = fmaf(dz, (xs[k] - mean) * is, s1);
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (lane == 0) { a.part[2 * pc] = s0; a.part[2 * pc + 1] = s1; }
}
__global__ __launch_bounds__(64) void tail_bwd_channel_kernel(TailArgs a) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int b = lane; b < a.B; b += 64) { s0 += a.part[2 * (b * a.C + c)]; s1 += a.part[2 * (b * a.C + c) + 1]; }
    for (int m = 1; m < 64; m <<= 1) { s0 += shfl_xor_f64(s0, m); s1 += shfl_xor_f64(s1, m); }
    if (lane == 0) { a.gbeta[c] = (float)s0; a.ggamma[c] = (float)s1; }
}
// backward pass 2: dL/dx
__global__ __launch_bounds__(256) void tail_bwd_dx_kernel(TailArgs a) {
    __shared__ float ga[4][16], gm[4][16];
    __shared__ int am[4][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, pc = blockIdx.x * 4 + wv;
    if (pc >= a.B * a.C) return;
    const int N = a.N, HW = N * N, c = pc % a.C, b = pc / a.C, win = N / 4;
    if (lane < 16) {
        ga[wv][lane] = a.gout[((size_t)b * 2 * a.C + c) * 16 + lane];
        gm[wv][lane] = a.gout[((size_t)b * 2 * a.C + a.C + c) * 16 + lane];
        am[wv][lane] = a.argmax[(size_t)pc * 16 + lane];
    }
    __builtin_amdgcn_wave_barrier();
    const float mean = a.mean[c], is = a.invstd[c], inv_area = 1.0f / (float)(win * win);
    const float gsc = (a.gamma ? a.gamma[c] : 1.0f) * is;
    const float n = (float)a.B * (float)HW;
    const float k0 = a.training ? a.gbeta[c] / n : 0.f, k1 = a.training ? a.ggamma[c] / n : 0.f;
    const float* p = a.x + (size_t)pc * HW;
    float* o = a.gx + (size_t)pc * HW;
    for (int e = 4 * lane; e < HW; e += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + e);
        const float xs[4] = {v.x, v.y, v.z, v.w};
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dz = tail_dz(ga[wv], gm[wv], am[wv], e + k, N, win, inv_area);
            r[k] = gsc * (dz - k0 - (xs[k] - mean) * is * k1);
        }
        *reinterpret_cast<float4*>(o + e) = make_float4(r[0], r[1], r[2], r[3]);
    }
}

int tail_check(int B, int C, int N) {
    if (B <= 0 || C <= 0 || N < 4 || N > kTailMaxN || (N % 4) != 0) return PDE_E_BADARG;
    return PDE_OK;
}

}  // namespace
}  // namespace pde

using namespace pde;

extern "C" {

size_t pde_bn_pool_workspace_bytes(int32_t B, int32_t C) {
    return (B > 0 && C > 0) ? (size_t)B * C * 2 * sizeof(float) : 0;
}

int pde_bn_pool_forward(int32_t B, int32_t C, int32_t N, const float* x, const float* gamma, const float* beta, float eps,
                        int32_t training, float momentum, float* running_mean, float* running_var, float* mean,
                        float* invstd, float* out, int32_t* argmax, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = tail_check(B, C, N);
    if (rc != PDE_OK) return rc;
    if (!x || !mean || !invstd || !out || !argmax || !workspace) return PDE_E_BADARG;
    if (!training && (!running_mean || !running_var)) return PDE_E_BADARG;
    if (workspace_bytes < pde_bn_pool_workspace_bytes(B, C)) return PDE_E_WORKSPACE;
    TailArgs a{};
    a.x = x; a.gamma = gamma; a.beta = beta; a.mean = mean; a.invstd = invstd; a.run_mean = running_mean; a.run_var = running_var;
    a.part = static_cast<float*>(workspace); a.out = out; a.argmax = argmax;
    a.B = B; a.C = C; a.N = N; a.training = training; a.eps = eps; a.momentum = momentum;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 planes((B * C + 3) / 4), chans((C + 63) / 64);
    if (training) {
        hipLaunchKernelGGL(tail_plane_stats_kernel, planes, dim3(256), 0, st, a);
        hipLaunchKernelGGL(tail_channel_stats_kernel, dim3(C), dim3(64), 0, st, a);
    } else {
        hipLaunchKernelGGL(tail_eval_stats_kernel, chans, dim3(64), 0, st, a);
    }
    hipLaunchKernelGGL(tail_pool_fwd_kernel, planes, dim3(256), (size_t)4 * N * N * sizeof(float), st, a);
    return check_launch();
}

int pde_bn_pool_backward(int32_t B, int32_t C, int32_t N, const float* x, const float* gamma, const float* mean,
                         const float* invstd, const int32_t* argmax, const float* gout, int32_t training, float* gx,
                         float* ggamma, float* gbeta, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = tail_check(B, C, N);
    if (rc != PDE_OK) return rc;
    if (!x || !mean || !invstd || !argmax || !gout || !gx || !ggamma || !gbeta || !workspace) return PDE_E_BADARG;
    if (workspace_bytes < pde_bn_pool_workspace_bytes(B, C)) return PDE_E_WORKSPACE;
    TailArgs a{};
    a.x = x; a.gamma = gamma; a.mean = const_cast<float*>(mean); a.invstd = const_cast<float*>(invstd);
    a.argmax = const_cast<int*>(argmax); a.gout = gout; a.gx = gx; a.ggamma = ggamma; a.gbeta = gbeta;
    a.part = static_cast<float*>(workspace);
    a.B = B; a.C = C; a.N = N; a.training = training;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 planes((B * C + 3) / 4);
    hipLaunchKernelGGL(tail_bwd_sums_kernel, planes, dim3(256), 0, st, a);
    hipLaunchKernelGGL(tail_bwd_channel_kernel, dim3(C), dim3(64), 0, st, a);
    hipLaunchKernelGGL(tail_bwd_dx_kernel, planes, dim3(256), 0, st, a);
    return check_launch();
}

}  // extern "C"
