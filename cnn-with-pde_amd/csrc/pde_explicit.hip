// K2 — explicit 5-point layers.
//
// explicit5: tiny_imagenet.py:34-72, one relaxed step
//     a_c = clamp(alpha_base_c, eps, max_coeff);  v = s_c u
//     out = u + relax*(v + a_c*dt*Lap0(v) - u)            Lap0: zero ghost cells (conv2d padding=1)
//   HBM-bound: 4 B read + 4 B written per element.  Planes of 64x64, 32x32 and 16x16 take the wave-per-plane
//   kernels below: a wave keeps its whole plane in registers (lane = (row group, 4-column group), R consecutive
//   rows of one float4 each), so every element is loaded from memory exactly once; east/west neighbours come
//   from the neighbouring lane by DPP wave shifts, north/south from the lane's own registers, the two halo rows
//   of a row group from the lanes W/4 away (__shfl); num_steps > 1 stay in registers between the steps.  Other
//   plane sizes take the generic kernel (neighbours through L1/L2), one launch per step.
//   Backward (Lap0 is self-adjoint):
//     gu   = (1-relax) g + relax*s_c*(g + a_c dt Lap0 g)
//     gs_c = relax * sum (g + a_c dt Lap0 g) u
//     ga_c = [eps <= alpha_base_c <= max_coeff] * relax*dt*s_c * sum (Lap0 g) u
//   one workgroup per (b,c) plane writes its two partial sums; a second kernel adds
//   them over b in a fixed order (no float atomics).
//
// jacobi: emotion_recognition.py:82-97, P = reflect_pad(u); nt times
//     P_int += A_i (P[i+1,j]-2P[i,j]+P[i-1,j]) + B_j (P[i,j+1]-2P[i,j]+P[i,j-1]); ring frozen.
//   One workgroup per sample, the padded plane lives in LDS for the whole time loop.
//   Backward recomputes the forward, parking each state in the workspace, then walks the
//   adjoint back, accumulating dA_i, dB_j per sample; a second kernel sums over samples.
#include "pde_common.h"

namespace pde {
namespace {

struct bf16e { unsigned short v; };
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) { return f32_to_bf16_hw(f); }
template <typename IO> struct V4;
template <> struct V4<float> {
    __device__ static __forceinline__ float4 ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static __forceinline__ void st(float* p, float4 v) {       // written once, never re-read here
        typedef float f4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(p));
    }
    __device__ static __forceinline__ float ld1(const float* p) { return *p; }
    __device__ static __forceinline__ float4 ld_once(const float* p) {    // no neighbour re-reads this one
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    __device__ static __forceinline__ float4 round(float4 v) { return v; }
};
template <> struct V4<bf16e> {
    __device__ static __forceinline__ float4 ld(const bf16e* p) {
        const ushort4 q = *reinterpret_cast<const ushort4*>(p);
        return make_float4(bf2f(q.x), bf2f(q.y), bf2f(q.z), bf2f(q.w));
    }
    __device__ static __forceinline__ void st(bf16e* p, float4 v) {
        ushort4 q; q.x = f2bf(v.x); q.y = f2bf(v.y); q.z = f2bf(v.z); q.w = f2bf(v.w);
        *reinterpret_cast<ushort4*>(p) = q;
    }
    __device__ static __forceinline__ float ld1(const bf16e* p) { return bf2f(p->v); }
    __device__ static __forceinline__ float4 ld_once(const bf16e* p) { return ld(p); }
    __device__ static __forceinline__ float4 round(float4 v) {
        return make_float4(bf2f(f2bf(v.x)), bf2f(f2bf(v.y)), bf2f(f2bf(v.z)), bf2f(f2bf(v.w)));
    }
};

// 5-point Laplacian with zero ghost cells of 4 consecutive columns (h, w0..w0+3) of one plane
template <typename IO>
__device__ __forceinline__ float4 lap0_4(const IO* plane, int H, int W, int h, int w0, const float4& c) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 n = (h > 0) ? V4<IO>::ld(plane + (size_t)(h - 1) * W + w0) : z;
    const float4 s = (h + 1 < H) ? V4<IO>::ld(plane + (size_t)(h + 1) * W + w0) : z;
    const float l = (w0 > 0) ? V4<IO>::ld1(plane + (size_t)h * W + w0 - 1) : 0.f;
    const float r = (w0 + 4 < W) ? V4<IO>::ld1(plane + (size_t)h * W + w0 + 4) : 0.f;
    float4 o;
    o.x = n.x + s.x + l + c.y - 4.f * c.x;
    o.y = n.y + s.y + c.x + c.z - 4.f * c.y;
    o.z = n.z + s.z + c.y + c.w - 4.f * c.z;
    o.w = n.w + s.w + c.z + r - 4.f * c.w;
    return o;
}

// (TI / TO: the step's input and output types — a call with bf16 tensors reads bf16 in its first step and writes bf16 in
//  its last; what passes between the steps is fp32)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void explicit5_fwd_kernel(const TI* __restrict__ u, const float* __restrict__ alpha,
                                                            const float* __restrict__ scale, TO* __restrict__ out,
                                                            int C, int H, int W, float dt, float eps, float maxc,
                                                            float relax) {
    const int pc = blockIdx.x;                         // plane index b*C + c
    const int c = pc % C;
    const float a = fminf(fmaxf(alpha[c], eps), maxc) * dt;
    const float s = scale[c];
    const TI* plane = u + (size_t)pc * H * W;
    TO* oplane = out + (size_t)pc * H * W;
    const int W4 = W / 4;
    for (int f = threadIdx.x; f < H * W4; f += 256) {
        const int h = f / W4, w0 = 4 * (f % W4);
        const float4 cu = V4<TI>::ld(plane + (size_t)h * W + w0);
        const float4 lp = lap0_4<TI>(plane, H, W, h, w0, cu);          // Lap0(u); Lap0(v) = s*Lap0(u)
        float4 o;
        // v = s u; new = v + a*(s*lap); out = u + relax*(new - u)
        { const float v = s * cu.x; const float nw = v + a * (s * lp.x); o.x = cu.x + relax * (nw - cu.x); }
        { const float v = s * cu.y; const float nw = v + a * (s * lp.y); o.y = cu.y + relax * (nw - cu.y); }
        { const float v = s * cu.z; const float nw = v + a * (s * lp.z); o.z = cu.z + relax * (nw - cu.z); }
        { const float v = s * cu.w; const float nw = v + a * (s * lp.w); o.w = cu.w + relax * (nw - cu.w); }
        V4<TO>::st(oplane + (size_t)h * W + w0, o);
    }
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

template <typename TU, typename TG, typename TO>
__global__ __launch_bounds__(256) void explicit5_bwd_kernel(const TU* __restrict__ u, const TG* __restrict__ g,
                                                            const float* __restrict__ alpha,
                                                            const float* __restrict__ scale, TO* __restrict__ gu,
                                                            float* __restrict__ part, int C, int H, int W, float dt,
                                                            float eps, float maxc, float relax, int accumulate) {
    __shared__ float sh[4];
    const int pc = blockIdx.x;
    const int c = pc % C;
    const float a = fminf(fmaxf(alpha[c], eps), maxc) * dt;
    const float s = scale[c];
    const TU* up = u + (size_t)pc * H * W;
    const TG* gp = g + (size_t)pc * H * W;
    TO* op = gu + (size_t)pc * H * W;
    const int W4 = W / 4;
    float p1 = 0.f, p2 = 0.f;                          // sum g*u, sum Lap0(g)*u
    for (int f = threadIdx.x; f < H * W4; f += 256) {
        const int h = f / W4, w0 = 4 * (f % W4);
        const float4 cg = V4<TG>::ld(gp + (size_t)h * W + w0);
        const float4 cu = V4<TU>::ld_once(up + (size_t)h * W + w0);
        const float4 lg = lap0_4<TG>(gp, H, W, h, w0, cg);
        float4 o;
        o.x = (1.f - relax) * cg.x + relax * s * (cg.x + a * lg.x);
        o.y = (1.f - relax) * cg.y + relax * s * (cg.y + a * lg.y);
        o.z = (1.f - relax) * cg.z + relax * s * (cg.z + a * lg.z);
        o.w = (1.f - relax) * cg.w + relax * s * (cg.w + a * lg.w);
        V4<TO>::st(op + (size_t)h * W + w0, o);
        p1 += cg.x * cu.x + cg.y * cu.y + cg.z * cu.z + cg.w * cu.w;
        p2 += lg.x * cu.x + lg.y * cu.y + lg.z * cu.z + lg.w * cu.w;
    }
    const float s1 = block_sum_256(p1, sh);
    const float s2 = block_sum_256(p2, sh);
    if (threadIdx.x == 0) {
        part[2 * (size_t)pc] = accumulate ? part[2 * (size_t)pc] + s1 : s1;
        part[2 * (size_t)pc + 1] = accumulate ? part[2 * (size_t)pc + 1] + s2 : s2;
    }
}

// one workgroup per channel: 256 threads add the per-plane partial sums of their samples, then a
// fixed-order block reduction (a single 64-thread block looping over all samples took 84 us at B=256)
__global__ __launch_bounds__(256) void explicit5_pgrad_kernel(const float* __restrict__ part,
                                                              const float* __restrict__ alpha,
                                                              const float* __restrict__ scale,
                                                              float* __restrict__ ga, float* __restrict__ gs, int B,
                                                              int C, float dt, float eps, float maxc, float relax) {
    __shared__ float sh[4];
    const int c = blockIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        s1 += part[2 * ((size_t)b * C + c)];
        s2 += part[2 * ((size_t)b * C + c) + 1];
    }
    s1 = block_sum_256(s1, sh);
    s2 = block_sum_256(s2, sh);
    if (threadIdx.x == 0) {
        const float ab = alpha[c];
        const float a = fminf(fmaxf(ab, eps), maxc) * dt;
        gs[c] = relax * (s1 + a * s2);
        ga[c] = (ab >= eps && ab <= maxc) ? relax * dt * scale[c] * s2 : 0.f;
    }
}


// ---------------------------------------------------------------------------------------
// wave-per-plane kernels: the plane lives in registers for the whole call
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_shift(float v) {     // 0 is shifted in at the ends of the wave
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int kWaveShl1 = 0x130;    // lane i <- lane i+1
constexpr int kWaveShr1 = 0x138;    // lane i <- lane i-1
__device__ __forceinline__ float4 shfl_up4(float4 v, int d) {
    return make_float4(__shfl_up(v.x, d, 64), __shfl_up(v.y, d, 64), __shfl_up(v.z, d, 64), __shfl_up(v.w, d, 64));
}
__device__ __forceinline__ float4 shfl_down4(float4 v, int d) {
    return make_float4(__shfl_down(v.x, d, 64), __shfl_down(v.y, d, 64), __shfl_down(v.z, d, 64), __shfl_down(v.w, d, 64));
}

// Lap0 of the lane's R rows (zero ghost cells): same operation order as lap0_4
template <int R, int W4>
__device__ __forceinline__ void lap0_rows(const float4 (&c)[R], float4 (&lp)[R], int cg, int rg) {
    constexpr int RG = 64 / W4;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 top = shfl_up4(c[R - 1], W4), bot = shfl_down4(c[0], W4);
    if (rg == 0) top = z;
    if (rg == RG - 1) bot = z;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const float4 n = (i > 0) ? c[i - 1] : top;
        const float4 s = (i < R - 1) ? c[i + 1] : bot;
        float l = dpp_shift<kWaveShr1>(c[i].w), r = dpp_shift<kWaveShl1>(c[i].x);
        if (cg == 0) l = 0.f;
        if (cg == W4 - 1) r = 0.f;
        lp[i].x = n.x + s.x + l + c[i].y - 4.f * c[i].x;
        lp[i].y = n.y + s.y + c[i].x + c[i].z - 4.f * c[i].y;
        lp[i].z = n.z + s.z + c[i].y + c[i].w - 4.f * c[i].z;
        lp[i].w = n.w + s.w + c[i].z + r - 4.f * c[i].w;
    }
}

template <typename IO, int R, int W4>
__global__ __launch_bounds__(256) void explicit5_fwd_wave(const IO* __restrict__ u, const float* __restrict__ alpha,
                                                          const float* __restrict__ scale, IO* __restrict__ out,
                                                          float* __restrict__ states, int nplanes, int C, int num_steps,
                                                          float dt, float eps, float maxc, float relax) {
    constexpr int W = 4 * W4, H = R * (64 / W4);
    const int lane = threadIdx.x & 63;
    const int pc = blockIdx.x * 4 + (threadIdx.x >> 6);          // one wave per plane
    if (pc >= nplanes) return;
    const int c = pc % C;
    const float a = fminf(fmaxf(alpha[c], eps), maxc) * dt;
    const float s = scale[c];
    const int cg = lane % W4, rg = lane / W4;
    const size_t off = (size_t)pc * H * W + (size_t)(rg * R) * W + 4 * cg;
    float4 cur[R], lp[R];
#pragma unroll
    for (int i = 0; i < R; ++i) cur[i] = V4<IO>::ld_once(u + off + (size_t)i * W);
    for (int step = 0; step < num_steps; ++step) {
        lap0_rows<R, W4>(cur, lp, cg, rg);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            // v = s u; new = v + a*(s*lap); out = u + relax*(new - u)      tiny_imagenet.py:43-49
            { const float v = s * cur[i].x; const float nw = v + a * (s * lp[i].x); cur[i].x = cur[i].x + relax * (nw - cur[i].x); }
            { const float v = s * cur[i].y; const float nw = v + a * (s * lp[i].y); cur[i].y = cur[i].y + relax * (nw - cur[i].y); }
            { const float v = s * cur[i].z; const float nw = v + a * (s * lp[i].z); cur[i].z = cur[i].z + relax * (nw - cur[i].z); }
            { const float v = s * cur[i].w; const float nw = v + a * (s * lp[i].w); cur[i].w = cur[i].w + relax * (nw - cur[i].w); }
        }
        // inputs of the later steps, for the backward: kept in fp32 whatever the tensors' type (with bf16 states the
        // alpha-gradient — a sum over the whole batch that cancels to a few units — was a quarter off on a drawn case)
        if (step + 1 < num_steps && states != nullptr) {
            float* sp = states + (size_t)step * nplanes * H * W;
#pragma unroll
            for (int i = 0; i < R; ++i) V4<float>::st(sp + off + (size_t)i * W, cur[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) V4<IO>::st(out + off + (size_t)i * W, cur[i]);
}

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// backward of num_steps steps: g walks back through the steps in registers; u_{k-1} (the input of step k) is the
// layer input for k = 1 and states[k-2] otherwise
template <typename IO, int R, int W4>
__global__ __launch_bounds__(256) void explicit5_bwd_wave(const IO* __restrict__ u, const float* __restrict__ states,
                                                          const IO* __restrict__ g, const float* __restrict__ alpha,
                                                          const float* __restrict__ scale, IO* __restrict__ gu,
                                                          float* __restrict__ part, int nplanes, int C, int num_steps,
                                                          float dt, float eps, float maxc, float relax) {
    constexpr int W = 4 * W4, H = R * (64 / W4);
    const int lane = threadIdx.x & 63;
    const int pc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pc >= nplanes) return;
    const int c = pc % C;
    const float a = fminf(fmaxf(alpha[c], eps), maxc) * dt;
    const float s = scale[c];
    const int cg = lane % W4, rg = lane / W4;
    const size_t off = (size_t)pc * H * W + (size_t)(rg * R) * W + 4 * cg;
    float4 cg4[R], lg[R];
#pragma unroll
    for (int i = 0; i < R; ++i) cg4[i] = V4<IO>::ld_once(g + off + (size_t)i * W);
    float p1 = 0.f, p2 = 0.f;                                     // sum g*u, sum Lap0(g)*u over all steps
    for (int step = num_steps; step >= 1; --step) {
        const float* sp = states + (size_t)(step >= 2 ? step - 2 : 0) * nplanes * H * W;
        lap0_rows<R, W4>(cg4, lg, cg, rg);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const float4 cu = (step == 1) ? V4<IO>::ld_once(u + off + (size_t)i * W) : V4<float>::ld_once(sp + off + (size_t)i * W);
            p1 += cg4[i].x * cu.x + cg4[i].y * cu.y + cg4[i].z * cu.z + cg4[i].w * cu.w;
            p2 += lg[i].x * cu.x + lg[i].y * cu.y + lg[i].z * cu.z + lg[i].w * cu.w;
            cg4[i].x = (1.f - relax) * cg4[i].x + relax * s * (cg4[i].x + a * lg[i].x);
            cg4[i].y = (1.f - relax) * cg4[i].y + relax * s * (cg4[i].y + a * lg[i].y);
            cg4[i].z = (1.f - relax) * cg4[i].z + relax * s * (cg4[i].z + a * lg[i].z);
            cg4[i].w = (1.f - relax) * cg4[i].w + relax * s * (cg4[i].w + a * lg[i].w);
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) V4<IO>::st(gu + off + (size_t)i * W, cg4[i]);
    p1 = wave_sum(p1);
    p2 = wave_sum(p2);
    if (lane == 0) { part[2 * (size_t)pc] = p1; part[2 * (size_t)pc + 1] = p2; }
}

// which plane sizes the wave-per-plane kernels cover
bool wave_plane_ok(int H, int W) { return (H == 64 && W == 64) || (H == 32 && W == 32) || (H == 16 && W == 16); }

template <typename IO>
void launch_fwd_wave(int H, const IO* u, const float* alpha, const float* scale, IO* out, float* states, int nplanes, int C,
                     int num_steps, float dt, float eps, float maxc, float relax, hipStream_t st) {
    const dim3 grid((nplanes + 3) / 4), block(256);
    if (H == 64) hipLaunchKernelGGL((explicit5_fwd_wave<IO, 16, 16>), grid, block, 0, st, u, alpha, scale, out, states, nplanes, C, num_steps, dt, eps, maxc, relax);
    else if (H == 32) hipLaunchKernelGGL((explicit5_fwd_wave<IO, 4, 8>), grid, block, 0, st, u, alpha, scale, out, states, nplanes, C, num_steps, dt, eps, maxc, relax);
    else hipLaunchKernelGGL((explicit5_fwd_wave<IO, 1, 4>), grid, block, 0, st, u, alpha, scale, out, states, nplanes, C, num_steps, dt, eps, maxc, relax);
}
template <typename IO>
void launch_bwd_wave(int H, const IO* u, const float* states, const IO* g, const float* alpha, const float* scale, IO* gu,
                     float* part, int nplanes, int C, int num_steps, float dt, float eps, float maxc, float relax,
                     hipStream_t st) {
    const dim3 grid((nplanes + 3) / 4), block(256);
    if (H == 64) hipLaunchKernelGGL((explicit5_bwd_wave<IO, 16, 16>), grid, block, 0, st, u, states, g, alpha, scale, gu, part, nplanes, C, num_steps, dt, eps, maxc, relax);
    else if (H == 32) hipLaunchKernelGGL((explicit5_bwd_wave<IO, 4, 8>), grid, block, 0, st, u, states, g, alpha, scale, gu, part, nplanes, C, num_steps, dt, eps, maxc, relax);
    else hipLaunchKernelGGL((explicit5_bwd_wave<IO, 1, 4>), grid, block, 0, st, u, states, g, alpha, scale, gu, part, nplanes, C, num_steps, dt, eps, maxc, relax);
}

// ---------------------------------------------------------------------------------------
// jacobi (emotion_recognition.PDELayer)
// ---------------------------------------------------------------------------------------
constexpr int kJMax = 64;                  // max H, W
constexpr int kJP = kJMax + 2;

__device__ __forceinline__ int reflect_src(int m, int n) {   // padded index m in [0,n+1] -> source index in [0,n)
    return m == 0 ? 1 : (m == n + 1 ? n - 2 : m - 1);
}

// one explicit update P -> Q of the interior; ring copied
__device__ __forceinline__ void jacobi_step(const float* P, float* Q, const float* a, const float* b, int H, int W) {
    const int Wp = W + 2;
    for (int e = threadIdx.x; e < (H + 2) * Wp; e += blockDim.x) {
        const int i = e / Wp, j = e % Wp;
        float v = P[e];
        if (i >= 1 && i <= H && j >= 1 && j <= W) {
            const float d1 = P[e + Wp] - 2.f * v + P[e - Wp];
            const float d2 = P[e + 1] - 2.f * v + P[e - 1];
            v = v + a[i - 1] * d1 + b[j - 1] * d2;
        }
        Q[e] = v;
    }
}

__global__ __launch_bounds__(256) void jacobi_fwd_kernel(const float* __restrict__ u, const float* __restrict__ a_row,
                                                         const float* __restrict__ b_col, float* __restrict__ out,
                                                         int H, int W, int nt) {
    extern __shared__ float sm[];
    const int Wp = W + 2, PN = (H + 2) * Wp;
    float* P = sm;
    float* Q = sm + PN;
    float* a = sm + 2 * PN;
    float* b = a + H;
    const float* ub = u + (size_t)blockIdx.x * H * W;
    for (int e = threadIdx.x; e < H; e += blockDim.x) a[e] = a_row[e];
    for (int e = threadIdx.x; e < W; e += blockDim.x) b[e] = b_col[e];
    for (int e = threadIdx.x; e < PN; e += blockDim.x) {
        const int i = e / Wp, j = e % Wp;
        P[e] = ub[reflect_src(i, H) * W + reflect_src(j, W)];          // F.pad(..., mode='reflect')
    }
    __syncthreads();
    for (int n = 0; n < nt; ++n) {
        jacobi_step(P, Q, a, b, H, W);
        __syncthreads();
        float* t = P; P = Q; Q = t;
    }
    float* ob = out + (size_t)blockIdx.x * H * W;
    for (int e = threadIdx.x; e < H * W; e += blockDim.x) ob[e] = P[(e / W + 1) * Wp + (e % W) + 1];
}

// workspace per sample: nt states of PN floats, then H+W partial sums
__global__ __launch_bounds__(256) void jacobi_bwd_kernel(const float* __restrict__ u, const float* __restrict__ gout,
                                                         const float* __restrict__ a_row,
                                                         const float* __restrict__ b_col, float* __restrict__ gu,
                                                         float* __restrict__ states, float* __restrict__ part,
                                                         int H, int W, int nt) {
    extern __shared__ float sm[];
    const int Wp = W + 2, PN = (H + 2) * Wp;
    float* P = sm;
    float* Q = sm + PN;
    float* a = sm + 2 * PN;
    float* b = a + H;
    float* ga = b + W;
    float* gb = ga + H;
    const int s = blockIdx.x;
    const float* ub = u + (size_t)s * H * W;
    float* st = states + (size_t)s * nt * PN;
    for (int e = threadIdx.x; e < H; e += blockDim.x) { a[e] = a_row[e]; ga[e] = 0.f; }
    for (int e = threadIdx.x; e < W; e += blockDim.x) { b[e] = b_col[e]; gb[e] = 0.f; }
    for (int e = threadIdx.x; e < PN; e += blockDim.x) {
        const int i = e / Wp, j = e % Wp;
        P[e] = ub[reflect_src(i, H) * W + reflect_src(j, W)];
    }
    __syncthreads();
    // forward recompute: park the state BEFORE each step
    for (int n = 0; n < nt; ++n) {
        for (int e = threadIdx.x; e < PN; e += blockDim.x) st[(size_t)n * PN + e] = P[e];
        jacobi_step(P, Q, a, b, H, W);
        __syncthreads();
        float* t = P; P = Q; Q = t;
    }
    // adjoint: G = dL/dP_nt (zero ring, gout inside)
    float* G = P;
    float* Gn = Q;
    const float* gob = gout + (size_t)s * H * W;
    for (int e = threadIdx.x; e < PN; e += blockDim.x) {
        const int i = e / Wp, j = e % Wp;
        G[e] = (i >= 1 && i <= H && j >= 1 && j <= W) ? gob[(i - 1) * W + (j - 1)] : 0.f;
    }
    __threadfence_block();
    __syncthreads();
    for (int n = nt - 1; n >= 0; --n) {
        const float* Pn = st + (size_t)n * PN;                 // state before step n (this block wrote it)
        // coefficient gradients: dA_i += sum_j G[i,j] d1(Pn)[i,j];  dB_j += sum_i G[i,j] d2(Pn)[i,j]
        for (int r = threadIdx.x; r < H + W; r += blockDim.x) {
            float acc = 0.f;
            if (r < H) {
                const int i = r + 1;
                for (int j = 1; j <= W; ++j) {
                    const int e = i * Wp + j;
                    acc += G[e] * (Pn[e + Wp] - 2.f * Pn[e] + Pn[e - Wp]);
                }
                ga[r] += acc;
            } else {
                const int j = r - H + 1;
                for (int i = 1; i <= H; ++i) {
                    const int e = i * Wp + j;
                    acc += G[e] * (Pn[e + 1] - 2.f * Pn[e] + Pn[e - 1]);
                }
                gb[r - H] += acc;
            }
        }
        // dL/dP_n from dL/dP_{n+1}
        for (int e = threadIdx.x; e < PN; e += blockDim.x) {
            const int i = e / Wp, j = e % Wp;
            const bool in = (i >= 1 && i <= H && j >= 1 && j <= W);
            float v = in ? (1.f - 2.f * a[i - 1] - 2.f * b[j - 1]) * G[e] : G[e];
            if (j >= 1 && j <= W) {                                       // vertical neighbours are interior columns
                if (i - 1 >= 1 && i - 1 <= H) v += a[i - 2] * G[e - Wp];
                if (i + 1 >= 1 && i + 1 <= H) v += a[i] * G[e + Wp];
            }
            if (i >= 1 && i <= H) {
                if (j - 1 >= 1 && j - 1 <= W) v += b[j - 2] * G[e - 1];
                if (j + 1 >= 1 && j + 1 <= W) v += b[j] * G[e + 1];
            }
            Gn[e] = v;
        }
        __syncthreads();
        float* t = G; G = Gn; Gn = t;
    }
    // adjoint of the reflect padding: fold the ring back onto rows/cols 1 and H-2 / W-2
    float* gub = gu + (size_t)s * H * W;
    for (int e = threadIdx.x; e < H * W; e += blockDim.x) {
        const int i = e / W, j = e % W;
        float v = 0.f;
        for (int di = 0; di < 2; ++di) {
            int m;
            if (di == 0) m = i + 1;
            else if (i == 1) m = 0;
            else if (i == H - 2) m = H + 1;
            else continue;
            for (int dj = 0; dj < 2; ++dj) {
                int nn;
                if (dj == 0) nn = j + 1;
                else if (j == 1) nn = 0;
                else if (j == W - 2) nn = W + 1;
                else continue;
                v += G[m * Wp + nn];
            }
            // rows 1 and H-2 coincide when H == 3; not supported (H,W >= 4 checked on the host)
        }
        gub[e] = v;
    }
    float* pp = part + (size_t)s * (H + W);
    for (int e = threadIdx.x; e < H; e += blockDim.x) pp[e] = ga[e];
    for (int e = threadIdx.x; e < W; e += blockDim.x) pp[H + e] = gb[e];
}

__global__ void jacobi_pgrad_kernel(const float* __restrict__ part, float* __restrict__ ga, float* __restrict__ gb,
                                    int B, int H, int W) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= H + W) return;
    float s = 0.f;
    for (int k = 0; k < B; ++k) s += part[(size_t)k * (H + W) + r];
    if (r < H) ga[r] = s; else gb[r - H] = s;
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace
}  // namespace pde

using namespace pde;

extern "C" {

int pde_explicit5_forward(int32_t B, int32_t C, int32_t H, int32_t W, int32_t io_dtype, const void* u,
                          const float* alpha_base, const float* channel_scaling, float dt, float eps, float max_coeff,
                          float relax, int32_t num_steps, void* states, void* out, void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (W % 4) != 0 || num_steps < 1 || !u || !alpha_base || !channel_scaling || !out)
        return PDE_E_BADARG;
    if (io_dtype != PDE_IO_F32 && io_dtype != PDE_IO_BF16) return PDE_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nplanes = B * C;
    if (wave_plane_ok(H, W)) {                             // the plane stays in registers over all steps
        if (io_dtype == PDE_IO_F32)
            launch_fwd_wave<float>(H, (const float*)u, alpha_base, channel_scaling, (float*)out, (float*)states, nplanes, C,
                                   num_steps, dt, eps, max_coeff, relax, st);
        else
            launch_fwd_wave<bf16e>(H, (const bf16e*)u, alpha_base, channel_scaling, (bf16e*)out, (float*)states, nplanes, C,
                                   num_steps, dt, eps, max_coeff, relax, st);
        return check_launch();
    }
    if (num_steps > 1 && !states) return PDE_E_BADARG;     // generic sizes: one launch per step through `states` (fp32)
    const size_t tf = (size_t)nplanes * H * W;
    float* sf = static_cast<float*>(states);
    const bool bf = io_dtype == PDE_IO_BF16;
#define PDE_EX_FWD(TI, TO, SRC, DST)                                                                                       \
    hipLaunchKernelGGL((explicit5_fwd_kernel<TI, TO>), dim3(nplanes), dim3(256), 0, st, (const TI*)(SRC), alpha_base,        \
                       channel_scaling, (TO*)(DST), C, H, W, dt, eps, max_coeff, relax)
    for (int k = 0; k < num_steps; ++k) {
        const bool first = k == 0, last = k == num_steps - 1;
        const void* src = first ? u : static_cast<const void*>(sf + (size_t)(k - 1) * tf);
        void* dst = last ? out : static_cast<void*>(sf + (size_t)k * tf);
        if (!bf || (!first && !last)) PDE_EX_FWD(float, float, src, dst);
        else if (first && last) PDE_EX_FWD(bf16e, bf16e, src, dst);
        else if (first) PDE_EX_FWD(bf16e, float, src, dst);
        else PDE_EX_FWD(float, bf16e, src, dst);
    }
#undef PDE_EX_FWD
    return check_launch();
}

size_t pde_explicit5_backward_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W, int32_t io_dtype,
                                              int32_t num_steps) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || num_steps < 1) return 0;
    size_t b = align256((size_t)B * C * 2 * sizeof(float));
    if (num_steps > 1 && !wave_plane_ok(H, W))             // generic sizes: two gradient buffers to ping-pong through
        b += 2 * align256((size_t)B * C * H * W * sizeof(float));
    return b;
}

int pde_explicit5_backward(int32_t B, int32_t C, int32_t H, int32_t W, int32_t io_dtype, const void* u,
                           const void* states, const void* gout, const float* alpha_base, const float* channel_scaling,
                           float dt, float eps, float max_coeff, float relax, int32_t num_steps, void* gu,
                           float* g_alpha_base, float* g_channel_scaling, void* workspace, size_t workspace_bytes,
                           void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (W % 4) != 0 || num_steps < 1 || !u || !gout || !alpha_base ||
        !channel_scaling || !gu || !g_alpha_base || !g_channel_scaling || !workspace || (num_steps > 1 && !states))
        return PDE_E_BADARG;
    if (io_dtype != PDE_IO_F32 && io_dtype != PDE_IO_BF16) return PDE_E_BADARG;
    if (workspace_bytes < pde_explicit5_backward_workspace_bytes(B, C, H, W, io_dtype, num_steps)) return PDE_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    const int nplanes = B * C;
    if (wave_plane_ok(H, W)) {
        if (io_dtype == PDE_IO_F32)
            launch_bwd_wave<float>(H, (const float*)u, (const float*)states, (const float*)gout, alpha_base, channel_scaling,
                                   (float*)gu, part, nplanes, C, num_steps, dt, eps, max_coeff, relax, st);
        else
            launch_bwd_wave<bf16e>(H, (const bf16e*)u, (const float*)states, (const bf16e*)gout, alpha_base, channel_scaling,
                                   (bf16e*)gu, part, nplanes, C, num_steps, dt, eps, max_coeff, relax, st);
    } else {
        // what passes between the steps is fp32 (the states the forward left, the gradient buffers here), whatever io_dtype
        const size_t tf = (size_t)nplanes * H * W;
        float* buf0 = reinterpret_cast<float*>(static_cast<char*>(workspace) + align256((size_t)nplanes * 2 * sizeof(float)));
        float* buf1 = reinterpret_cast<float*>(reinterpret_cast<char*>(buf0) + align256(tf * sizeof(float)));
        const float* sf = static_cast<const float*>(states);
        const bool bf = io_dtype == PDE_IO_BF16;
        const void* gin = gout;
#define PDE_EX_BWD(TU, TG, TO, UP, GIN, GDST, ACC)                                                                          \
    hipLaunchKernelGGL((explicit5_bwd_kernel<TU, TG, TO>), dim3(nplanes), dim3(256), 0, st, (const TU*)(UP), (const TG*)(GIN), \
                       alpha_base, channel_scaling, (TO*)(GDST), part, C, H, W, dt, eps, max_coeff, relax, ACC)
        for (int k = num_steps; k >= 1; --k) {
            const bool ufirst = k == 1, gfirst = k == num_steps;       // u / gu are the caller's tensors, gout too
            const void* up = ufirst ? u : static_cast<const void*>(sf + (size_t)(k - 2) * tf);
            void* gdst = ufirst ? gu : static_cast<void*>(gin == buf0 ? buf1 : buf0);
            const int acc = gfirst ? 0 : 1;
            if (!bf) PDE_EX_BWD(float, float, float, up, gin, gdst, acc);
            else if (ufirst && gfirst) PDE_EX_BWD(bf16e, bf16e, bf16e, up, gin, gdst, acc);
            else if (ufirst) PDE_EX_BWD(bf16e, float, bf16e, up, gin, gdst, acc);
            else if (gfirst) PDE_EX_BWD(float, bf16e, float, up, gin, gdst, acc);
            else PDE_EX_BWD(float, float, float, up, gin, gdst, acc);
            gin = gdst;
        }
#undef PDE_EX_BWD
    }
    hipLaunchKernelGGL(explicit5_pgrad_kernel, dim3(C), dim3(256), 0, st, part, alpha_base, channel_scaling,
                       g_alpha_base, g_channel_scaling, B, C, dt, eps, max_coeff, relax);
    return check_launch();
}

static size_t jacobi_lds(int H, int W, bool bwd) {
    const size_t PN = (size_t)(H + 2) * (W + 2);
    return (2 * PN + (bwd ? 2 : 1) * (H + W)) * sizeof(float);
}

int pde_jacobi_forward(int32_t B, int32_t H, int32_t W, int32_t nt, const float* u, const float* a_row,
                       const float* b_col, float* out, void* stream) {
    if (B <= 0 || H < 4 || W < 4 || H > kJMax || W > kJMax || nt < 0 || !u || !a_row || !b_col || !out)
        return PDE_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(jacobi_fwd_kernel, dim3(B), dim3(256), jacobi_lds(H, W, false), st, u, a_row, b_col, out, H, W, nt);
    return check_launch();
}

size_t pde_jacobi_backward_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t nt) {
    if (B <= 0 || H <= 0 || W <= 0 || nt < 0) return 0;
    const size_t PN = (size_t)(H + 2) * (W + 2);
    return align256((size_t)B * nt * PN * sizeof(float)) + align256((size_t)B * (H + W) * sizeof(float));
}

int pde_jacobi_backward(int32_t B, int32_t H, int32_t W, int32_t nt, const float* u, const float* gout,
                        const float* a_row, const float* b_col, float* gu, float* g_a_row, float* g_b_col,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (B <= 0 || H < 4 || W < 4 || H > kJMax || W > kJMax || nt < 0 || !u || !gout || !a_row || !b_col || !gu ||
        !g_a_row || !g_b_col || !workspace)
        return PDE_E_BADARG;
    if (workspace_bytes < pde_jacobi_backward_workspace_bytes(B, H, W, nt)) return PDE_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t PN = (size_t)(H + 2) * (W + 2);
    float* states = static_cast<float*>(workspace);
    float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + align256((size_t)B * nt * PN * sizeof(float)));
    hipLaunchKernelGGL(jacobi_bwd_kernel, dim3(B), dim3(256), jacobi_lds(H, W, true), st, u, gout, a_row, b_col, gu,
                       states, part, H, W, nt);
    hipLaunchKernelGGL(jacobi_pgrad_kernel, dim3((H + W + 63) / 64), dim3(64), 0, st, part, g_a_row, g_b_col, B, H, W);
    return check_launch();
}

}  // extern "C"
