// Device code of the K1 sweep kernels (included by pde_adi_inst.hip once per line length N,
// and by pde_adi.hip for the argument structures).  See pde_adi.hip for the design notes.
#pragma once
#include "pde_common.h"

#include <type_traits>

namespace pde {
namespace {

#ifndef PDE_WAVES
#define PDE_WAVES 8
#endif
constexpr int kWaves = PDE_WAVES;                 // waves per workgroup
constexpr int kThreads = kWaves * 64;
#ifndef PDE_RING
#define PDE_RING 3
#endif
constexpr int kRing = PDE_RING;                   // coefficient-record ring of the skewed kernels (2 is enough without the skew)
#ifndef PDE_SKEW
#define PDE_SKEW 1
#endif
// waves kWaves/2.. run one sweep behind waves 0..kWaves/2-1 (wave w and w + kWaves/2 share a SIMD)
__device__ __forceinline__ int wave_lag(int wave) { return (PDE_SKEW && wave >= kWaves / 2) ? 1 : 0; }
// The second-dispatched half of a workgroup (waves kWaves/2..) loses the SIMD's issue arbitration against its
// older partner on every instruction (MI355X_MICROARCH.md, "Two waves per SIMD", item 4): one static s_setprio
// for that half before the main loop evens the two out, so nobody waits at the step barrier for a starved wave.
#ifndef PDE_PRIO
#define PDE_PRIO 1
#endif
__device__ __forceinline__ void young_half_priority(int wave) {
#if PDE_PRIO
    if (wave >= kWaves / 2) __builtin_amdgcn_s_setprio(PDE_PRIO);
#endif
}

// Per-launch sweep table read by the sweep kernels with scalar loads (keeping it in the
// kernel arguments makes hipcc hold all of it in SGPRs and spill them).
struct SweepTab {
    int axis[PDE_MAX_SWEEPS];
    float dts[PDE_MAX_SWEEPS];      // bwd: t_s minus t of the previous (earlier) sweep of the same axis
    int first_s[2];                 // bwd: earliest sweep of each axis (-1: none)
    float t_last[2];                // bwd: time of the latest sweep of each axis
    float ysc[PDE_MAX_SWEEPS];      // bwd: (1+eps)^-(S-1-s): true state after sweep s -> rescaled state
};

// order of the axes inside one time step, known at compile time for the two splits the
// reference uses; kSplitAny looks the axis up per sweep
constexpr int kSplitAny = 0;
constexpr int kSplitStrang = 1;       // x, y, x   (mnist_test.py:55-63)
constexpr int kSplitLie = 2;          // x, y      (cifar_2version.py:93-99)

struct SweepArgs {
    const void* in0;        // fwd: u        bwd: gy
    const void* in1;        // fwd: -        bwd: y
    void* out;              // fwd: y        bwd: gu
    const float* coef;      // [S][C][kRecAll]
    float* part;            // bwd: [G][C][4][kImage] partial parameter-gradient sums
    const SweepTab* tab;
    const int* varying;     // bwd: [C] per-channel "clamp mask changes with time" flag
    const void* in2;        // bwd with checkpoints: u
    float* ckpt;            // bwd with checkpoints: [slots][B][C][N][N] fp32
    unsigned long long ck[2];   // bit s: the state after sweep s is checkpointed
    int Sf;                 // bwd: sweeps 0..Sf-1 are recomputed forward first (0: no checkpoints)
    int smooth3;
    int xcd_map;            // 1: XCD-ordered block -> (channel, group) map (needs C % 8 == 0)
    int B, C, S, G;
    float one_eps;          // 1 + eps
    float gu_scale;         // bwd: (1+eps)^-S
    int acc_part;           // bwd: add the partial gradient sums to what `part` holds (per-step launches)
    int pair_x;             // Strang schedules: the last x sweep of a step and the first of the next have the SAME record
                            // (same time, same increment): its rows are loaded from LDS once for the two sweeps
    void* dbg;              // diagnostic builds only
    int only_masked;        // bwd: the grid is C*G blocks of the MASKED body alone (the fast body ran as the assembly kernel)
};

// The sweep table and the channel flags are written by an earlier kernel and only read here.
// Reading them through the constant address space makes hipcc use scalar loads; through a plain
// global pointer it issues VECTOR loads followed by s_waitcnt vmcnt(0), which also drains the
// coefficient prefetch that was just issued (measured: a full memory round trip per sweep).
typedef const __attribute__((address_space(4))) SweepTab* ConstTab;
typedef const __attribute__((address_space(4))) int* ConstInt;
__device__ __forceinline__ ConstTab as_const(const SweepTab* p) { return (ConstTab)(unsigned long long)p; }
__device__ __forceinline__ ConstInt as_const(const int* p) { return (ConstInt)(unsigned long long)p; }

// blockIdx -> (channel, group).  Workgroups are observed to be dealt round-robin over the 8 XCDs
// (MI355X_MICROARCH.md, "Workgroup dispatch"), each with a private 4 MB L2.  With xcd_map the G
// groups of one channel are consecutive on ONE XCD, so an XCD walks through its channels one (or
// two) at a time and the channel's coefficient records (S x 14 KB) stay L2-resident instead of
// eight channels' worth competing with the streamed tensors.  Placement affects speed only.
__device__ __forceinline__ void block_to_work(int b, int C, int G, int xcd_map, int& c, int& g) {
    if (xcd_map) {
        const int x = b & 7, j = b >> 3;
        c = x + 8 * (j / G);
        g = j % G;
    } else {
        c = b % C;
        g = b / C;
    }
}

// 16 bytes per lane, global -> LDS without passing through registers (LDS address = wave-uniform
// base + 16*lane).  Issued from inline asm ON PURPOSE: with the builtin, hipcc treats the DMA as a
// pending LDS store and puts s_waitcnt vmcnt(0) in front of the next ds_read of ANY address, i.e. it
// waits out the full memory latency right after issuing the prefetch.  The asm form is invisible to
// that pass, so the kernel waits itself: dma_wait_all() before the barrier that publishes the data.
// (m0 is not used by anything else in these kernels; gfx9 DS instructions do not need it.)
__device__ __forceinline__ void lds_dma16(const float* gsrc, float* lds_wave_base) {
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(gsrc) : "memory");
}
// The same with a wave-uniform base address (SGPR pair) and a per-lane byte offset that never changes
// (16*lane): no vector address arithmetic per piece.
__device__ __forceinline__ void lds_dma16_s(const float* sbase, unsigned voff, float* lds_wave_base) {
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory");
}
// The same with the LDS destination given as a byte address (wave-uniform): for callers whose LDS pointers reach the
// call through enough control flow that hipcc no longer proves their address space (the generic -> LDS cast then
// fails in the backend: "V_CMP_NE_U32_e32 0, $src_shared_base").
__device__ __forceinline__ unsigned lds_byte_address(const float* lds_base) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const float*)lds_base;
}
// a pointer the compiler cannot prove wave-uniform, made so (the "s" operand of the DMA must be an SGPR pair)
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void lds_dma16_a(const float* sbase, unsigned voff, unsigned lds_bytes) {
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_bytes);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Diagnostic builds only (tools/ablate.sh): PDE_ABL bit 0 = coefficient rows are read from LDS for the first item only,
// bit 1 = no re-layouts, bit 2 = no plane loads/stores after the first chunk.  Results are WRONG; only the launch time
// is read.  No shipped kernel is built with PDE_ABL != 0.
#ifndef PDE_ABL
#define PDE_ABL 0
#endif

#ifdef PDE_STAMP
// Diagnostic build only (tools/stamp.sh): cycle stamps of one wave's phases inside a sweep.  The
// values go to a debug buffer nothing else reads; no shipped kernel executes a stamp.
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define PDE_STAMP_AT(i) do { if (stamp_on) stamps[i] = stamp(); } while (0)
#else
#define PDE_STAMP_AT(i) do { } while (0)
#endif

// value held by lane ^ 32 (the other half of my line).  v_permlane32_swap (new on gfx950) does it in
// the VALU; __shfl_xor becomes ds_bpermute_b32, which queues behind the re-layout traffic of the
// other waves in the LDS pipe, right on the critical path of every solve (the junction).
__device__ __forceinline__ float xchg_half(float v, int hf) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[0] = lo|lo, r[1] = hi|hi
    return __builtin_bit_cast(float, hf ? r[0] : r[1]);
}

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int kDppWaveShl1 = 0x130;   // lane i <- lane i+1
constexpr int kDppWaveShr1 = 0x138;   // lane i <- lane i-1

// ---- the J planes of a lane as one value ------------------------------------------------------
// The planes a lane works on (backward 2, forward 4) go through identical arithmetic with shared
// coefficients, so they are written as ONE value of type Pack<J>::P: with PDE_PACK=0 (default) a struct
// of scalars, i.e. J independent instruction chains; with PDE_PACK=1 an ext-vector, i.e. v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 with the coefficient broadcast by op_sel.  On gfx950 a wave64 fp32 VALU
// instruction holds its SIMD for 4 cycles and a packed one for 8 (tools/ubench/pk_bench.hip), so the
// packed build issues 22 % fewer instructions and is 4 % slower; it is kept as a build option only.
#ifndef PDE_PACK
#define PDE_PACK 0             // the Makefile's default; 1: v_pk_*_f32 (measured slower)
#endif
#if PDE_PACK
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2f v2_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v4f v4_fma(v4f a, v4f b, v4f c) { return __builtin_elementwise_fma(a, b, c); }
#else
// comparison build (make PACK=0): the same code on two scalar registers per pair
struct v2f { float x, y; };
__device__ __forceinline__ v2f operator+(v2f a, v2f b) { return v2f{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ v2f operator-(v2f a, v2f b) { return v2f{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ v2f operator*(v2f a, v2f b) { return v2f{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ v2f operator-(v2f a) { return v2f{-a.x, -a.y}; }
__device__ __forceinline__ v2f v2_fma(v2f a, v2f b, v2f c) { return v2f{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
struct v4f { float x, y, z, w; };
__device__ __forceinline__ v4f operator+(v4f a, v4f b) { return v4f{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ v4f operator-(v4f a, v4f b) { return v4f{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
__device__ __forceinline__ v4f operator*(v4f a, v4f b) { return v4f{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
__device__ __forceinline__ v4f operator-(v4f a) { return v4f{-a.x, -a.y, -a.z, -a.w}; }
__device__ __forceinline__ v4f v4_fma(v4f a, v4f b, v4f c) {
    return v4f{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)};
}
#endif
struct v3f { float x, y, z; };
__device__ __forceinline__ v3f operator+(v3f a, v3f b) { return v3f{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ v3f operator-(v3f a, v3f b) { return v3f{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ v3f operator*(v3f a, v3f b) { return v3f{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ v3f operator-(v3f a) { return v3f{-a.x, -a.y, -a.z}; }
template <int J> struct Pack;
template <> struct Pack<1> { using P = float; };
template <> struct Pack<3> { using P = v3f; };
template <> struct Pack<2> { using P = v2f; };
template <> struct Pack<4> { using P = v4f; };        // forward only: four planes share every coefficient read
template <int C> __device__ __forceinline__ float pk_get(float p) { return p; }
template <int C> __device__ __forceinline__ float pk_get(v2f p) { return C ? p.y : p.x; }
template <int C> __device__ __forceinline__ float pk_get(v3f p) { return C == 0 ? p.x : C == 1 ? p.y : p.z; }
template <int C> __device__ __forceinline__ void pk_set(v3f& p, float v) { if (C == 0) p.x = v; else if (C == 1) p.y = v; else p.z = v; }
__device__ __forceinline__ v3f pk_fma(v3f a, v3f b, v3f c) { return v3f{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z)}; }
template <int C> __device__ __forceinline__ float pk_get(v4f p) { return C == 0 ? p.x : C == 1 ? p.y : C == 2 ? p.z : p.w; }
template <int C> __device__ __forceinline__ void pk_set(float& p, float v) { p = v; }
template <int C> __device__ __forceinline__ void pk_set(v2f& p, float v) { if (C) p.y = v; else p.x = v; }
template <int C> __device__ __forceinline__ void pk_set(v4f& p, float v) {
    if (C == 0) p.x = v; else if (C == 1) p.y = v; else if (C == 2) p.z = v; else p.w = v;
}
__device__ __forceinline__ float pk_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return v2_fma(a, b, c); }
__device__ __forceinline__ v4f pk_fma(v4f a, v4f b, v4f c) { return v4_fma(a, b, c); }
template <class P> __device__ __forceinline__ P pk_bc(float s);
template <> __device__ __forceinline__ float pk_bc<float>(float s) { return s; }
template <> __device__ __forceinline__ v2f pk_bc<v2f>(float s) { return v2f{s, s}; }
template <> __device__ __forceinline__ v4f pk_bc<v4f>(float s) { return v4f{s, s, s, s}; }
template <> __device__ __forceinline__ v3f pk_bc<v3f>(float s) { return v3f{s, s, s}; }
__device__ __forceinline__ float pk_hsum(float p) { return p; }
__device__ __forceinline__ float pk_hsum(v2f p) { return p.x + p.y; }
__device__ __forceinline__ float pk_hsum(v3f p) { return p.x + p.y + p.z; }
__device__ __forceinline__ float pk_hsum(v4f p) { return (p.x + p.y) + (p.z + p.w); }
// per-component map (cross-lane moves have no packed form)
template <class F> __device__ __forceinline__ float pk_map(float p, F&& f) { return f(p); }
template <class F> __device__ __forceinline__ v2f pk_map(v2f p, F&& f) { return v2f{f(p.x), f(p.y)}; }
template <class F> __device__ __forceinline__ v3f pk_map(v3f p, F&& f) { return v3f{f(p.x), f(p.y), f(p.z)}; }
template <class F> __device__ __forceinline__ v4f pk_map(v4f p, F&& f) { return v4f{f(p.x), f(p.y), f(p.z), f(p.w)}; }

// Stage COUNT floats of a coefficient record global -> registers -> LDS: up to three 16-byte
// pieces per thread, held in plain locals of the kernel (a struct here ends up in scratch).
struct Staged { float4 r0, r1, r2, r3, r4; };
template <int COUNT>
__device__ __forceinline__ void stage_load(const float* src, int tid, Staged& st) {
    constexpr int F4 = COUNT / 4;
    static_assert(F4 <= 5 * kThreads, "record too large for five pieces per thread");
    const float4* p = reinterpret_cast<const float4*>(src);
    st.r0 = p[tid];
    if constexpr (F4 > kThreads) { if (F4 >= 2 * kThreads || tid + kThreads < F4) st.r1 = p[tid + kThreads]; }
    if constexpr (F4 > 2 * kThreads) { if (F4 >= 3 * kThreads || tid + 2 * kThreads < F4) st.r2 = p[tid + 2 * kThreads]; }
    if constexpr (F4 > 3 * kThreads) { if (F4 >= 4 * kThreads || tid + 3 * kThreads < F4) st.r3 = p[tid + 3 * kThreads]; }
    if constexpr (F4 > 4 * kThreads) { if (tid + 4 * kThreads < F4) st.r4 = p[tid + 4 * kThreads]; }
}
template <int COUNT>
__device__ __forceinline__ void stage_store(float* dst, int tid, const Staged& st) {
    constexpr int F4 = COUNT / 4;
    float4* p = reinterpret_cast<float4*>(dst);
    p[tid] = st.r0;
    if constexpr (F4 > kThreads) { if (F4 >= 2 * kThreads || tid + kThreads < F4) p[tid + kThreads] = st.r1; }
    if constexpr (F4 > 2 * kThreads) { if (F4 >= 3 * kThreads || tid + 2 * kThreads < F4) p[tid + 2 * kThreads] = st.r2; }
    if constexpr (F4 > 3 * kThreads) { if (F4 >= 4 * kThreads || tid + 3 * kThreads < F4) p[tid + 3 * kThreads] = st.r3; }
    if constexpr (F4 > 4 * kThreads) { if (tid + 4 * kThreads < F4) p[tid + 4 * kThreads] = st.r4; }
}

// ---- plane I/O through the wave's LDS image (natural [h][w] rows, stride 36) --------
template <typename IO> struct IoTraits;
template <> struct IoTraits<float> {
    static constexpr int kVec = 4;                      // elements per 16-byte access
    // planes are touched exactly once per launch: non-temporal, so that they stream past the L2 lines
    // that hold the coefficient records (re-read by every chunk of every workgroup of the channel)
    typedef float f4 __attribute__((ext_vector_type(4)));
    __device__ static __forceinline__ float4 load4(const float* p) {
        const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    __device__ static __forceinline__ void store4(float* p, float4 v) {
        __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(p));
    }
};
struct bf16_t { unsigned short v; };
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) { return f32_to_bf16_hw(f); }
template <> struct IoTraits<bf16_t> {
    __device__ static __forceinline__ float4 load4(const bf16_t* p) {
        typedef unsigned short u4 __attribute__((ext_vector_type(4)));
        const u4 q = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
        return make_float4(bf16_to_f32(q.x), bf16_to_f32(q.y), bf16_to_f32(q.z), bf16_to_f32(q.w));
    }
    __device__ static __forceinline__ void store4(bf16_t* p, float4 v) {
        typedef unsigned short u4 __attribute__((ext_vector_type(4)));
        const u4 q = {f32_to_bf16(v.x), f32_to_bf16(v.y), f32_to_bf16(v.z), f32_to_bf16(v.w)};
        __builtin_nontemporal_store(q, reinterpret_cast<u4*>(p));
    }
};

template <int N>
struct Geo {
    static constexpr int M = N / 2;
    static constexpr int NN4 = N * N / 4;               // float4 per plane
    static constexpr int R4 = N / 4;                    // float4 per row
    static constexpr int kLoads = (NN4 + 63) / 64;      // per-lane 16-byte accesses per plane
};

// global -> registers (issue only).  `valid` must be wave-uniform.
template <int N, typename IO>
__device__ __forceinline__ void plane_fetch(const IO* gp, bool valid, int lane, float4 (&q)[Geo<N>::kLoads]) {
#pragma unroll
    for (int i = 0; i < Geo<N>::kLoads; ++i) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
#pragma unroll
        for (int i = 0; i < Geo<N>::kLoads; ++i) {
            const int f = i * 64 + lane;
            if ((i + 1) * 64 <= Geo<N>::NN4 || f < Geo<N>::NN4) q[i] = IoTraits<IO>::load4(gp + 4 * f);
        }
    }
}

// compile-time loop: every index below is a constant in the AST, so register arrays are
// scalarised before any select-of-loads folding can turn them into dynamic indexing
template <int I, int E, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, E>(f);
    }
}

// Plane I/O goes through a wave-private LDS image whose rows are in natural order and whose
// columns are in half order (half_pos): the coalesced side (float4 = two pairs of columns)
// puts each pair where its half expects it, swapping the two floats of a pair that lands in
// the mirrored high half; the row side then moves whole half rows with ds_*_b128.
template <int N>
__device__ __forceinline__ int pair_slot(int row, int w, bool& swapped) {
    swapped = w >= N / 2;                       // w is even and N/2 is even-or-odd*2: pairs never straddle
    return row * kLineStride + (swapped ? kHalfPad + (N - 2 - w) : w);
}

template <int N, int C, class P>
__device__ __forceinline__ void plane_to_rows(const float4 (&q)[Geo<N>::kLoads], float* T, int lane, int l, int hf,
                                              P (&v)[Geo<N>::M]) {
    constexpr int M = Geo<N>::M;
#pragma unroll
    for (int i = 0; i < Geo<N>::kLoads; ++i) {
        const int f = i * 64 + lane;
        if ((i + 1) * 64 <= Geo<N>::NN4 || f < Geo<N>::NN4) {
            const int row = f / Geo<N>::R4, w0 = 4 * (f % Geo<N>::R4);
            bool s0, s1;
            const int a0 = pair_slot<N>(row, w0, s0), a1 = pair_slot<N>(row, w0 + 2, s1);
            *reinterpret_cast<float2*>(&T[a0]) = s0 ? make_float2(q[i].y, q[i].x) : make_float2(q[i].x, q[i].y);
            *reinterpret_cast<float2*>(&T[a1]) = s1 ? make_float2(q[i].w, q[i].z) : make_float2(q[i].z, q[i].w);
        }
    }
    __builtin_amdgcn_wave_barrier();
    const float* src = T + l * kLineStride + hf * kHalfPad;
#pragma unroll
    for (int i = 0; i < (M + 3) / 4; ++i) {
        const float4 x = *reinterpret_cast<const float4*>(src + 4 * i);
        pk_set<C>(v[4 * i], x.x);
        if (4 * i + 1 < M) pk_set<C>(v[4 * i + 1], x.y);
        if (4 * i + 2 < M) pk_set<C>(v[4 * i + 2], x.z);
        if (4 * i + 3 < M) pk_set<C>(v[4 * i + 3], x.w);
    }
    __builtin_amdgcn_wave_barrier();
}

template <int N, int C, typename IO, class P>
__device__ __forceinline__ void rows_to_plane(const P (&v)[Geo<N>::M], float* T, int lane, int l, int hf,
                                              IO* gp, bool valid) {
    constexpr int M = Geo<N>::M;
    if (l < N) {
        float* dst = T + l * kLineStride + hf * kHalfPad;
#pragma unroll
        for (int i = 0; i < (M + 3) / 4; ++i) {
            float4 x;
            x.x = pk_get<C>(v[4 * i]);
            x.y = (4 * i + 1 < M) ? pk_get<C>(v[4 * i + 1]) : 0.f;
            x.z = (4 * i + 2 < M) ? pk_get<C>(v[4 * i + 2]) : 0.f;
            x.w = (4 * i + 3 < M) ? pk_get<C>(v[4 * i + 3]) : 0.f;
            *reinterpret_cast<float4*>(dst + 4 * i) = x;
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < Geo<N>::kLoads; ++i) {
        const int f = i * 64 + lane;
        if ((i + 1) * 64 <= Geo<N>::NN4 || f < Geo<N>::NN4) {
            const int row = f / Geo<N>::R4, w0 = 4 * (f % Geo<N>::R4);
            bool s0, s1;
            const int a0 = pair_slot<N>(row, w0, s0), a1 = pair_slot<N>(row, w0 + 2, s1);
            const float2 p0 = *reinterpret_cast<const float2*>(&T[a0]);
            const float2 p1 = *reinterpret_cast<const float2*>(&T[a1]);
            float4 x;
            x.x = s0 ? p0.y : p0.x; x.y = s0 ? p0.x : p0.y;
            x.z = s1 ? p1.y : p1.x; x.w = s1 ? p1.x : p1.y;
            if (valid) IoTraits<IO>::store4(gp + 4 * f, x);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// row layout <-> column layout of one plane (an involution; same code both ways).
// The image used here is private to this exchange, so its rows and its columns are both kept
// in half order (row index hf*M + k, column index half_pos): every access is then
// lane base + compile-time offset.
template <int N, int C, class P>
__device__ __forceinline__ void relayout(P (&v)[Geo<N>::M], float* T, int l, int hf) {
    constexpr int M = Geo<N>::M;
    if (l < N) {
        const int mypos = (l < M) ? l : kHalfPad + (N - 1 - l);
        float* dst = T + hf * M * kLineStride + mypos;
#pragma unroll
        for (int k = 0; k < M; ++k) dst[k * kLineStride] = pk_get<C>(v[k]);
    }
    __builtin_amdgcn_wave_barrier();
    // (idle lanes, l >= N, read their own never-written row: the formula below would send them to rows of live data or,
    // for N < 22, in front of the image — into whatever another wave or an unwritten ring tail holds, NaN included)
    const int myrow = (l < M) ? l : (l < N ? M + (N - 1 - l) : l);
    const float* src = T + myrow * kLineStride + hf * kHalfPad;
#pragma unroll
    for (int i = 0; i < (M + 3) / 4; ++i) {
        const float4 x = *reinterpret_cast<const float4*>(src + 4 * i);
        pk_set<C>(v[4 * i], x.x);
        if (4 * i + 1 < M) pk_set<C>(v[4 * i + 1], x.y);
        if (4 * i + 2 < M) pk_set<C>(v[4 * i + 2], x.z);
        if (4 * i + 3 < M) pk_set<C>(v[4 * i + 3], x.w);
    }
    __builtin_amdgcn_wave_barrier();
}

// (A two-planes-at-once variant with ds_write_b64 was measured and dropped: 3-6 % slower in both
// kernels and 82 vs 67 CU cycles per plane in tools/ubench/relayout_bench.hip.)
template <int N, int J>
__device__ __forceinline__ void relayout_all(typename Pack<J>::P (&v)[Geo<N>::M], float* T, int l, int hf) {
    sfor<0, J>([&](auto CC) __attribute__((always_inline)) { relayout<N, decltype(CC)::value>(v, T, l, hf); });
}

template <int M>
__device__ __forceinline__ void load_half(const float* src, float (&dst)[M]) {
#pragma unroll
    for (int i = 0; i < (M + 3) / 4; ++i) {
        const float4 x = *reinterpret_cast<const float4*>(src + 4 * i);
        dst[4 * i] = x.x;
        if (4 * i + 1 < M) dst[4 * i + 1] = x.y;
        if (4 * i + 2 < M) dst[4 * i + 2] = x.z;
        if (4 * i + 3 < M) dst[4 * i + 3] = x.w;
    }
}

// ---- forward: (A + eps I) x = d on J planes, two-sided ---------------------------------
template <int M, int J>
__device__ __forceinline__ void solve_fwd_rows(typename Pack<J>::P (&v)[M], const float (&e)[M], const float (&inv)[M], float jn,
                                               int hf) {
    using P = typename Pack<J>::P;
    // elimination from my end inwards: D_k = d_k*inv_k + e_k*D_{k-1}
#pragma unroll
    for (int k = 0; k < M; ++k) {
        const P t = v[k] * pk_bc<P>(inv[k]);
        v[k] = (k == 0) ? t : pk_fma(pk_bc<P>(e[k]), v[k - 1], t);
    }
    // junction: x_in = (D_in + e_in * D_in(partner)) / (1 - e_t e_b)
    {
        const P other = pk_map(v[M - 1], [&](float z) { return xchg_half(z, hf); });
        v[M - 1] = pk_fma(pk_bc<P>(e[M - 1]), other, v[M - 1]) * pk_bc<P>(jn);
    }
    // substitution outwards: x_k = D_k + e_k*x_{k+1}
#pragma unroll
    for (int k = M - 2; k >= 0; --k) v[k] = pk_fma(pk_bc<P>(e[k]), v[k + 1], v[k]);
}
template <int M>
__device__ __forceinline__ void load_fwd_rows(const float* rec, int l, int hf, float (&e)[M], float (&inv)[M], float& jn) {
    load_half<M>(rec + kF_E + l * kLineStride + hf * kHalfPad, e);
    load_half<M>(rec + kF_Inv + l * kLineStride + hf * kHalfPad, inv);
    jn = rec[kF_Jn + l];
}
template <int M, int J>
__device__ __forceinline__ void solve_fwd(typename Pack<J>::P (&v)[M], const float* rec, int l, int hf) {
    float e[M], inv[M], jn;
    load_fwd_rows<M>(rec, l, hf, e, inv, jn);
    solve_fwd_rows<M, J>(v, e, inv, jn, hf);
}

__device__ __forceinline__ int ck_slot(const unsigned long long (&ck)[2], int s) {
    return s < 64 ? __builtin_popcountll(ck[0] & ((1ull << s) - 1ull))
                  : __builtin_popcountll(ck[0]) + __builtin_popcountll(ck[1] & ((1ull << (s - 64)) - 1ull));
}
__device__ __forceinline__ int ck_bit(const unsigned long long (&ck)[2], int s) {
    const unsigned long long w = (s < 64) ? ck[0] : ck[1];      // (a run-time index would put the pair in scratch)
    return (int)((w >> (s & 63)) & 1ull);
}

template <int N, int J, typename IO>
__device__ __forceinline__ void load_planes(const IO* base, int q, int wave, int lane, int l, int hf, int B, int C,
                                            int c, float* T, typename Pack<J>::P (&v)[N / 2]) {
    constexpr int PPI = kWaves * J;
    float4 raw[J][Geo<N>::kLoads];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int b = q * PPI + wave * J + j;
        plane_fetch<N, IO>(base + ((size_t)b * C + c) * (size_t)(N * N), b < B, lane, raw[j]);
    }
    sfor<0, J>([&](auto CC) __attribute__((always_inline)) {
        plane_to_rows<N, decltype(CC)::value>(raw[decltype(CC)::value], T, lane, l, hf, v);
    });
}

// two planes at a time: half the memory round trips of load_planes_seq with 32 registers in flight
template <int N, int J, typename IO>
__device__ __forceinline__ void load_planes_pairs(const IO* base, int q, int wave, int lane, int l, int hf, int B, int C,
                                                  int c, float* T, typename Pack<J>::P (&v)[N / 2]) {
    static_assert(J % 2 == 0, "pairs");
    constexpr int PPI = kWaves * J;
    sfor<0, J / 2>([&](auto PC) __attribute__((always_inline)) {
        constexpr int j0 = 2 * decltype(PC)::value;
        float4 raw0[Geo<N>::kLoads], raw1[Geo<N>::kLoads];
        const int b = q * PPI + wave * J + j0;
        plane_fetch<N, IO>(base + ((size_t)b * C + c) * (size_t)(N * N), b < B, lane, raw0);
        plane_fetch<N, IO>(base + ((size_t)(b + 1) * C + c) * (size_t)(N * N), b + 1 < B, lane, raw1);
        plane_to_rows<N, j0>(raw0, T, lane, l, hf, v);
        plane_to_rows<N, j0 + 1>(raw1, T, lane, l, hf, v);
    });
}

// one plane at a time (fewer registers in flight; used where the caller's own state is large)
template <int N, int J, typename IO>
__device__ __forceinline__ void load_planes_seq(const IO* base, int q, int wave, int lane, int l, int hf, int B, int C,
                                                int c, float* T, typename Pack<J>::P (&v)[N / 2]) {
    constexpr int PPI = kWaves * J;
    sfor<0, J>([&](auto CC) __attribute__((always_inline)) {
        constexpr int j = decltype(CC)::value;
        float4 raw[Geo<N>::kLoads];
        const int b = q * PPI + wave * J + j;
        plane_fetch<N, IO>(base + ((size_t)b * C + c) * (size_t)(N * N), b < B, lane, raw);
        plane_to_rows<N, j>(raw, T, lane, l, hf, v);
    });
}

template <int N, int J, typename IO>
__device__ __forceinline__ void store_planes(IO* base, int q, int wave, int lane, int l, int hf, int B, int C, int c,
                                             float* T, const typename Pack<J>::P (&v)[N / 2]) {
    constexpr int PPI = kWaves * J;
    sfor<0, J>([&](auto CC) __attribute__((always_inline)) {
        constexpr int j = decltype(CC)::value;
        const int b = q * PPI + wave * J + j;
        rows_to_plane<N, j, IO>(v, T, lane, l, hf, base + ((size_t)b * C + c) * (size_t)(N * N), b < B);
    });
}

// ---- forward kernel ---------------------------------------------------------------------
// SPLIT fixes the order of the axes inside a time step at compile time (straight-line code per
// step, no per-sweep axis branch); kSplitAny reads the axis of every sweep from the table.
// (second bound: waves per SIMD.  4 = two workgroups per CU; the bf16 instantiation otherwise takes 138 VGPRs and runs alone on its CU)
template <int N, int J, typename IO, int SPLIT>
__global__ __launch_bounds__(kThreads, (kWaves == 8 ? 4 : 1)) void adi_fwd_kernel(SweepArgs a) {
    constexpr int M = Geo<N>::M;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cbuf = smem;                                   // [kRing][kRecFwdPad]
    float* tbuf = smem + kRing * kRecFwdPad;              // [kWaves][kImage]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = lane >> 5, l = lane & 31;
    int c, g;
    block_to_work(blockIdx.x, a.C, a.G, a.xcd_map, c, g);
    float* T = tbuf + wave * kImage;
    const IO* u = static_cast<const IO*>(a.in0);
    IO* y = static_cast<IO*>(a.out);
    const ConstTab tab = as_const(a.tab);
    constexpr int PPI = kWaves * J;                       // planes per workgroup iteration
    const int nchunk = (a.B + PPI - 1) / PPI;
    const size_t plane = (size_t)N * N;
    // Phase skew (see "skew" in pde_adi.hip): the upper half of the waves runs ONE sweep behind the
    // lower half, so while one half is in a y sweep (re-layouts: LDS pipe) the other is in an x sweep
    // (VALU); in lock-step every wave wants the same pipe at the same time.
    // A launch of at most kRing sweeps (the per-step launches of the layers with a channel operator: 2 or 3) keeps ALL
    // its records in the ring for the whole launch: no staging per sweep, no barrier per sweep, no skew — the waves of
    // a workgroup run free of each other, so one wave's plane loads and stores overlap the others' sweeps (with a
    // barrier per sweep every workgroup of the launch loaded, swept and stored in step with all the others).
    const bool resident = a.S <= kRing;
    const int lag = resident ? 0 : wave_lag(wave);
    young_half_priority(wave);

    // rows >= N of the wave images are never written: zero them once so idle lanes read zeros
    for (int e = tid; e < kWaves * kImage; e += kThreads) tbuf[e] = 0.f;
    // Sweeps are numbered along the whole job of this workgroup (item i = sweep i % S of chunk i / S).
    // In barrier interval t the lower waves run item t from ring slot t % 3, the upper waves item
    // t-1 from slot (t-1) % 3, and every wave brings its pieces of item t+1 into slot (t+1) % 3 by
    // LDS-DMA (no registers: a record staged through VGPRs costs a VMEM return and a ds_write per dword,
    // 15 % of everything this kernel moves through the register file).
    int cur = 0;                                          // ring slot of my current item
    // Wave w brings pieces w, w + kWaves, ... of a record (whole 1-KB pieces: the padded window lies inside the global
    // record, so no lane is masked).  Everything that does not depend on the sweep is computed once: a piece is one
    // scalar multiply-add for the source, one add for m0 and the DMA itself (the address arithmetic used to be a fifth
    // of this kernel's scalar instructions, and every instruction of any kind costs its wave an issue slot).
    constexpr int PPRF = kRecFwdPad / 256;                // 1-KB pieces per record
    static_assert(kG_Inv + kRecFwdPad <= kRecStride, "the padded forward window must stay inside the record");
    const float* dma_src0 = uniform_ptr(a.coef + (size_t)c * kRecStride + kG_Inv + wave * 256);
    const unsigned dma_step = (unsigned)a.C * kRecStride;             // floats between the records of two sweeps
    const unsigned dma_lds0 = lds_byte_address(cbuf) + wave * 1024u;
    const unsigned dma_voff = 16u * lane;
    auto dma_rec = [&](int slot, int s) __attribute__((always_inline)) {
        const float* src = dma_src0 + (size_t)((unsigned)s * dma_step);
        const unsigned dst = dma_lds0 + (unsigned)slot * (kRecFwdPad * 4u);
#pragma unroll
        for (int i = 0; i < (PPRF + kWaves - 1) / kWaves; ++i) {
            if ((i + 1) * kWaves <= PPRF || wave + i * kWaves < PPRF)
                lds_dma16_a(src + i * kWaves * 256, dma_voff, dst + i * kWaves * 1024u);
        }
    };
    if (resident) {
        for (int s = 0; s < a.S; ++s) dma_rec(s, s);
    } else {
        dma_rec(0, 0);
    }
    dma_wait_all();
    __syncthreads();
    if (lag) {                                            // interval 0 of the upper waves: staging only
        dma_rec(1, a.S > 1 ? 1 : 0);
        dma_wait_all();
        __syncthreads();
    }

    // coefficient rows of the current sweep: the first x sweep of a Strang step keeps those of the previous step's last
    // one (same record, see SweepArgs::pair_x) instead of reading them from LDS again
    float ce[M], cinv[M], cjn = 0.f;
    bool abl_loaded = false;
    for (int q = g; q < nchunk; q += a.G) {
        typename Pack<J>::P v[M];
        if (!((PDE_ABL & 4) && abl_loaded)) {
#ifndef PDE_FWD_PAR_LOAD
#define PDE_FWD_PAR_LOAD 0
#endif
        if constexpr (J > 2 && PDE_FWD_PAR_LOAD == 2) load_planes_pairs<N, J, IO>(u, q, wave, lane, l, hf, a.B, a.C, c, T, v);
        else if constexpr (J > 2 && !PDE_FWD_PAR_LOAD) load_planes_seq<N, J, IO>(u, q, wave, lane, l, hf, a.B, a.C, c, T, v);   // 16, not 16*J, registers in flight
        else load_planes<N, J, IO>(u, q, wave, lane, l, hf, a.B, a.C, c, T, v);
        }
        auto sweep = [&](auto AXC, int s, auto TWINC) {
            constexpr int AX = decltype(AXC)::value;
            int sp = s + lag + 1;                         // sweep of the item staged in this interval
            if (sp >= a.S) sp -= a.S;
            if (sp >= a.S) sp -= a.S;
            int ps = cur + lag + 1;
            if (ps >= kRing) ps -= kRing;
            if (!resident && !((PDE_ABL & 8) && abl_loaded)) dma_rec(ps, sp);
            const float* rec = cbuf + (resident ? s : cur) * kRecFwdPad;
            const int axs = (AX >= 0) ? AX : tab->axis[s];
            if (axs == PDE_AXIS_Y && !(PDE_ABL & 2)) relayout_all<N, J>(v, T, l, hf);
            if (!(decltype(TWINC)::value && a.pair_x != 0 && s != 0) && !((PDE_ABL & 1) && abl_loaded)) load_fwd_rows<M>(rec, l, hf, ce, cinv, cjn);
            abl_loaded = true;
            solve_fwd_rows<M, J>(v, ce, cinv, cjn, hf);
            if (axs == PDE_AXIS_Y && !(PDE_ABL & 2)) relayout_all<N, J>(v, T, l, hf);
            if (a.ckpt != nullptr && ck_bit(a.ck, s)) {   // backward pre-pass: park this state (fp32)
                float* slot = a.ckpt + (size_t)ck_slot(a.ck, s) * a.B * a.C * plane;
                store_planes<N, J, float>(slot, q, wave, lane, l, hf, a.B, a.C, c, T, v);
            }
            if (!resident && !((PDE_ABL & 8) && s > 0)) {
                dma_wait_all();                           // my pieces of the next record have landed
                __syncthreads();
                cur = (cur == kRing - 1) ? 0 : cur + 1;
            }
        };
        if constexpr (SPLIT == kSplitStrang) {
            for (int s = 0; s < a.S; s += 3) {
                sweep(std::integral_constant<int, PDE_AXIS_X>{}, s, std::true_type{});       // twin of the sweep before it
                sweep(std::integral_constant<int, PDE_AXIS_Y>{}, s + 1, std::false_type{});
                sweep(std::integral_constant<int, PDE_AXIS_X>{}, s + 2, std::false_type{});
            }
        } else if constexpr (SPLIT == kSplitLie) {
            for (int s = 0; s < a.S; s += 2) {
                sweep(std::integral_constant<int, PDE_AXIS_X>{}, s, std::false_type{});
                sweep(std::integral_constant<int, PDE_AXIS_Y>{}, s + 1, std::false_type{});
            }
        } else {
            for (int s = 0; s < a.S; ++s) sweep(std::integral_constant<int, -1>{}, s, std::false_type{});
        }
        if (y != nullptr && !((PDE_ABL & 4) && q + a.G < nchunk)) store_planes<N, J, IO>(y, q, wave, lane, l, hf, a.B, a.C, c, T, v);
    }
    if (!lag) __syncthreads();                            // the interval in which the upper waves finish
}

// ---- backward -------------------------------------------------------------------------
// The coefficient rows (e, inv, jn) are fetched by the caller.
// adjoint two-sided solve (A + eps I)^T g = r on the J planes of a lane, in place.
template <int M, int J>
__device__ __forceinline__ void solve_adj(typename Pack<J>::P (&r)[M], const float (&e)[M], const float (&inv)[M],
                                          float jn, int hf) {
    using P = typename Pack<J>::P;
    // H_k = r_k + e_{k-1} H_{k-1}
#pragma unroll
    for (int k = 1; k < M; ++k) r[k] = pk_fma(pk_bc<P>(e[k - 1]), r[k - 1], r[k]);
    // junction: G_in = (H_in + e_in(partner) H_in(partner)) / (1 - e_t e_b)
    {
        const P mine = r[M - 1] * pk_bc<P>(e[M - 1]);
        const P pv = pk_map(mine, [&](float z) { return xchg_half(z, hf); });
        r[M - 1] = (r[M - 1] + pv) * pk_bc<P>(jn);
    }
    // G_k = H_k + e_{k+1} G_{k+1};   g_k = inv_k G_k
#pragma unroll
    for (int k = M - 2; k >= 0; --k) r[k] = pk_fma(pk_bc<P>(e[k + 1]), r[k + 1], r[k]);
#pragma unroll
    for (int k = 0; k < M; ++k) r[k] = r[k] * pk_bc<P>(inv[k]);             // inv = (1+eps)/den
}

// acc += sum over the planes of a lane of g*q (the accumulators are shared by the planes)
__device__ __forceinline__ float acc_gq(float acc, float g, float q) { return fmaf(g, q, acc); }
__device__ __forceinline__ float acc_gq(float acc, v2f g, v2f q) { return fmaf(g.y, q.y, fmaf(g.x, q.x, acc)); }
__device__ __forceinline__ float acc_gq(float acc, v3f g, v3f q) { return fmaf(g.z, q.z, fmaf(g.y, q.y, fmaf(g.x, q.x, acc))); }
__device__ __forceinline__ float acc_gq(float acc, v4f g, v4f q) { return fmaf(g.w, q.w, fmaf(g.z, q.z, fmaf(g.y, q.y, fmaf(g.x, q.x, acc)))); }

// After an x sweep has been undone on the adjoint (g in r[]), use the sweep's OUTPUT state
// x to (1) add g.(Lx) to the coefficient-gradient sums, (2) rebuild the sweep's input
// x_prev = (1+eps) x + kap.(Lx).  L = Neumann second difference along the row.
// MASKED: the clamp mask of this channel changes with time, so the sum over sweeps cannot be
// masked (and un-smoothed) once at the end: do both here, per sweep.
template <int M, int J, bool MASKED>
__device__ __forceinline__ void state_x(const typename Pack<J>::P (&g)[M], typename Pack<J>::P (&x)[M],
                                        float (&acc)[M], typename Pack<J>::P xin, const float (&kap)[M],
                                        const float* rec, int l, int hf, int smooth) {
    using P = typename Pack<J>::P;
    float msk[MASKED ? M : 1];
    if constexpr (MASKED) load_half<M>(rec + kB_MaskX + l * kLineStride + hf * kHalfPad, msk);
    P gq[MASKED ? M : 1];
    P xo_next = xin;                                     // inner neighbour of k = M-1 (fetched before the solve)
#pragma unroll
    for (int k = M - 1; k >= 0; --k) {
        const P xo = x[k];
        P q;
        if (k == 0) q = xo - xo_next;
        else q = pk_fma(pk_bc<P>(2.0f), xo, -x[k - 1]) - xo_next;
        if constexpr (MASKED) gq[k] = g[k] * q;
        else acc[k] = acc_gq(acc[k], g[k], q);
        x[k] = pk_fma(pk_bc<P>(kap[k]), q, xo);
        xo_next = xo;
    }
    if constexpr (MASKED) {
        if (smooth) {                                    // transpose of the replicate 3-tap average (x1/3 later)
            const P gin = pk_map(gq[M - 1], [&](float z) { return xchg_half(z, hf); });
#pragma unroll
            for (int k = 0; k < M; ++k) {
                P z = (k == 0) ? pk_bc<P>(2.0f) * gq[0] : gq[k] + gq[k - 1];
                z = z + ((k < M - 1) ? gq[k + 1] : gin);
                acc[k] = fmaf(msk[k], pk_hsum(z), acc[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < M; ++k) acc[k] = fmaf(msk[k], pk_hsum(gq[k]), acc[k]);
        }
    }
}

// q -= [h>0] x(row h-1) + [h<N-1] x(row h+1): the neighbour rows are the neighbour lanes; the DPP
// shift rides on the fmac (src0 = shifted x, 0 shifted in at the wave's ends).  No packed form.
__device__ __forceinline__ float lap_rows(float q, float xo, float nmu, float nmd) {
    asm("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(q) : "v"(xo), "v"(nmu));
    asm("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(q) : "v"(xo), "v"(nmd));
    return q;
}
__device__ __forceinline__ v2f lap_rows(v2f q, v2f xo, float nmu, float nmd) {
    return v2f{lap_rows(q.x, xo.x, nmu, nmd), lap_rows(q.y, xo.y, nmu, nmd)};
}
__device__ __forceinline__ v3f lap_rows(v3f q, v3f xo, float nmu, float nmd) {
    return v3f{lap_rows(q.x, xo.x, nmu, nmd), lap_rows(q.y, xo.y, nmu, nmd), lap_rows(q.z, xo.z, nmu, nmd)};
}
__device__ __forceinline__ v4f lap_rows(v4f q, v4f xo, float nmu, float nmd) {
    return v4f{lap_rows(q.x, xo.x, nmu, nmd), lap_rows(q.y, xo.y, nmu, nmd), lap_rows(q.z, xo.z, nmu, nmd), lap_rows(q.w, xo.w, nmu, nmd)};
}

// Same for a y sweep, with the state (and g) in ROW layout: the second difference (and, when
// MASKED, the transposed smoothing) runs across lanes (rows h-1, h+1 = lanes l-1, l+1 of the
// same half).
template <int N, int J, bool MASKED>
__device__ __forceinline__ void state_y(const typename Pack<J>::P (&g)[N / 2], typename Pack<J>::P (&x)[N / 2],
                                        float (&acc)[N / 2], const float (&kap)[N / 2], const float* rec, int l,
                                        int hf, int smooth) {
    using P = typename Pack<J>::P;
    constexpr int M = N / 2;
    float msk[MASKED ? M : 1];
    if constexpr (MASKED) load_half<M>(rec + kB_MaskX + l * kLineStride + hf * kHalfPad, msk);
    const bool edge = (l == 0 || l == N - 1);
    const float kk = edge ? 1.0f : 2.0f;
    const float kz = edge ? 2.0f : 1.0f;
    const float mu = (l > 0) ? 1.0f : 0.0f;
    const float md = (l < N - 1) ? 1.0f : 0.0f;
    const float nmu = -mu, nmd = -md;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        const P xo = x[k];
        const P q = lap_rows(pk_bc<P>(kk) * xo, xo, nmu, nmd);
        if constexpr (MASKED) {
            const P gq = g[k] * q;
            P z = gq;
            if (smooth) {
                const P zu = pk_map(gq, [&](float w) { return dpp_move<kDppWaveShr1>(w); });
                const P zd = pk_map(gq, [&](float w) { return dpp_move<kDppWaveShl1>(w); });
                z = pk_bc<P>(kz) * gq;
                z = pk_fma(pk_bc<P>(mu), zu, z);
                z = pk_fma(pk_bc<P>(md), zd, z);
            }
            acc[k] = fmaf(msk[k], pk_hsum(z), acc[k]);
        } else {
            acc[k] = acc_gq(acc[k], g[k], q);
        }
        x[k] = pk_fma(pk_bc<P>(kap[k]), q, xo);
    }
}

// The same as state_y (fast body only: clamp masks applied after the sum over sweeps), with the neighbour ROWS fetched
// through the wave's private LDS image instead of DPP.  On gfx950 a DPP instruction holds the VALU for two issue slots
// and costs its wave ~18 cycles beside a partner wave (tools/ubench/valu_issue, real_stream: the y sweep's VALU stream
// with its 64 v_fmac_f32_dpp takes 1957 cycles per wave at two waves per SIMD, the x sweep's 180 plain instructions
// 800): a lane writes its half row (4 x ds_write_b128) and reads the half rows above and below (8 x ds_read_b128),
// then the second difference is three plain VOP2 instructions per element.
#ifndef PDE_YLDS
#define PDE_YLDS 0          // measured: 363 us against 327 with DPP (and 14 spilled registers)
#endif
template <int N, int J>
__device__ __forceinline__ void state_y_lds(const typename Pack<J>::P (&g)[N / 2], typename Pack<J>::P (&x)[N / 2],
                                            float (&acc)[N / 2], const float (&kap)[N / 2], float* T, int l, int hf) {
    constexpr int M = N / 2;
    const bool edge = (l == 0 || l == N - 1);
    const float kk = edge ? 1.0f : 2.0f;
    const float nmu = (l > 0) ? -1.0f : 0.0f;            // row above exists
    const float nmd = (l < N - 1) ? -1.0f : 0.0f;        // row below exists
    const int lu = (l > 0 && l < N) ? l - 1 : l, ld = (l < N - 1) ? l + 1 : l;   // (a missing neighbour reads my own row, times 0)
    float* mine = T + l * kLineStride + hf * kHalfPad;
    const float* up = T + lu * kLineStride + hf * kHalfPad;
    const float* dn = T + ld * kLineStride + hf * kHalfPad;
    sfor<0, J>([&](auto CC) __attribute__((always_inline)) {
        constexpr int C = decltype(CC)::value;
        if (l < N) {
#pragma unroll
            for (int i = 0; i < (M + 3) / 4; ++i) {
                float4 v;
                v.x = pk_get<C>(x[4 * i]);
                v.y = (4 * i + 1 < M) ? pk_get<C>(x[4 * i + 1]) : 0.f;
                v.z = (4 * i + 2 < M) ? pk_get<C>(x[4 * i + 2]) : 0.f;
                v.w = (4 * i + 3 < M) ? pk_get<C>(x[4 * i + 3]) : 0.f;
                *reinterpret_cast<float4*>(mine + 4 * i) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // four elements at a time: 8 registers of neighbour values in flight, not 32
        sfor<0, (M + 3) / 4>([&](auto IC) __attribute__((always_inline)) {
            constexpr int i = decltype(IC)::value;
            const float4 u4 = *reinterpret_cast<const float4*>(up + 4 * i);
            const float4 d4 = *reinterpret_cast<const float4*>(dn + 4 * i);
            const float xu[4] = {u4.x, u4.y, u4.z, u4.w}, xd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                constexpr int dummy = 0; (void)dummy;
                const int k = 4 * i + j;
                if (k < M) {
                    const float xo = pk_get<C>(x[k]);
                    const float q = fmaf(nmd, xd[j], fmaf(nmu, xu[j], kk * xo));
                    acc[k] = fmaf(pk_get<C>(g[k]), q, acc[k]);
                    pk_set<C>(x[k], fmaf(kap[k], q, xo));
                }
            }
        });
        __builtin_amdgcn_wave_barrier();
    });
}

// LDS footprint (floats) of the coefficient buffers of the backward kernel.  With a compile-time
// step pattern the records are staged by LDS-DMA (global_load_lds: no registers, no ds_write) into
// a ring of kRing slots, one record per barrier interval, the upper waves running one sweep behind
// the lower ones (phase skew, as in the forward kernel); records are padded to whole 1-KB DMA pieces.
#ifndef PDE_BWD_ITEMB
#define PDE_BWD_ITEMB 0     // 1: one barrier per sweep, two-slot ring, no phase skew (small LDS footprint: two workgroups per CU)
#endif
template <bool MASKED, int SPLIT>
struct BwdStage {
    static constexpr int kSps = SPLIT == kSplitStrang ? 3 : (SPLIT == kSplitLie ? 2 : 1);
    static constexpr bool kStep = SPLIT != kSplitAny;
    static constexpr int kRec = MASKED ? kRecBwdMasked : kRecBwd;
    static constexpr int kRecPad = kStep ? ((kRec / 4 + 63) / 64) * 256 : kRec;
    static constexpr int kSlots = PDE_BWD_ITEMB ? 2 : 2 * kSps + (PDE_SKEW ? 1 : 0);   // record ring of the pattern kernels
    static constexpr int kFloats = kStep ? kSlots * kRecPad : 2 * kRecPad;
};

template <int N, int J, typename IO, bool MASKED, int SPLIT>
__device__ __forceinline__ void adi_bwd_body(const SweepArgs& a, int blk) {
    constexpr int M = Geo<N>::M;
    using ST = BwdStage<MASKED, SPLIT>;
    constexpr int REC = ST::kRec, RECP = ST::kRecPad, SPS = ST::kSps;
    int c, g;
    block_to_work(blk, a.C, a.G, a.xcd_map, c, g);
    if ((as_const(a.varying)[c] != 0) != MASKED) return;  // the other instantiation owns this channel
    const ConstTab tab = as_const(a.tab);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cbuf = smem;                                   // pattern: [kSlots][RECP]; any: [2][REC]
    float* tbuf = smem + ST::kFloats;                     // [kWaves][kImage]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = lane >> 5, l = lane & 31;
    float* T = tbuf + wave * kImage;
    const IO* gy = static_cast<const IO*>(a.in0);
    const IO* yy = static_cast<const IO*>(a.in1);
    IO* gu = static_cast<IO*>(a.out);
    constexpr int PPI = kWaves * J;
    const int nchunk = (a.B + PPI - 1) / PPI;
    const size_t plane = (size_t)N * N;

    float Ax[M], Tx[M], Ay[M], Ty[M];
#pragma unroll
    for (int k = 0; k < M; ++k) Ax[k] = Tx[k] = Ay[k] = Ty[k] = 0.f;

    for (int e = tid; e < kWaves * kImage; e += kThreads) tbuf[e] = 0.f;
    // per-axis table entries: scalar loads once, not a dependent chain of them in every sweep
    const int first_x = tab->first_s[0], first_y = tab->first_s[1];
    const float tlast_x = tab->t_last[0], tlast_y = tab->t_last[1];
    // The per-sweep time increments live in the lanes of ONE register (lane s = sweep s) and are picked with v_readlane:
    // a scalar load per item costs the wave its full memory latency (~550 cycles measured, a quarter of the run time of
    // a build with everything else removed) because the compiler must wait for lgkmcnt(0) before the next LDS read.
    const float dts_lanes = a.tab->dts[lane < PDE_MAX_SWEEPS ? lane : 0];
    auto dts_of = [&](int s) __attribute__((always_inline)) -> float {
        if (s < 64) return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dts_lanes), s));
        return tab->dts[s];
    };

    // one time step per launch (the per-step launches of the layers with a channel operator): its SPS records stay in
    // the ring for the whole launch and the waves run free of each other — see adi_fwd_kernel
    const bool resident = ST::kStep && a.S == SPS;
    const int lag = (ST::kStep && !resident && !PDE_BWD_ITEMB) ? wave_lag(wave) : 0;
    // (no young_half_priority here: measured 6 % slower in the backward, 4.5 % faster in the forward)
    // record of sweep s -> ring slot `slot`
    // (whole 1-KB pieces, sweep-independent address parts computed once: see adi_fwd_kernel)
    constexpr int PPR = RECP / 256;                       // 1-KB pieces per record
    static_assert(!ST::kStep || kBwdOff + RECP <= kRecStride, "the padded backward window must stay inside the record");
    const float* dma_src0 = uniform_ptr(a.coef + (size_t)c * kRecStride + kBwdOff + wave * 256);
    const unsigned dma_step = (unsigned)a.C * kRecStride;
    const unsigned dma_lds0 = lds_byte_address(cbuf) + wave * 1024u;
    const unsigned dma_voff = 16u * lane;
    auto dma_rec = [&](int slot, int s) __attribute__((always_inline)) {
        const float* src = dma_src0 + (size_t)((unsigned)s * dma_step);
        const unsigned dst = dma_lds0 + (unsigned)slot * (RECP * 4u);
#pragma unroll
        for (int i = 0; i < (PPR + kWaves - 1) / kWaves; ++i) {
            if ((i + 1) * kWaves <= PPR || wave + i * kWaves < PPR)
                lds_dma16_a(src + i * kWaves * 256, dma_voff, dst + i * kWaves * 1024u);
        }
    };

    using P = typename Pack<J>::P;
    P r[M], x[M];
    int q = g;                                            // chunk of my current item
    bool more = q + a.G < nchunk;
    // The time-weighted sums use summation by parts over the whole processing sequence
    // (all chunks, sweeps in decreasing time):  sum_i tau_i G_i = sum_i (tau_i - tau_{i+1}) R_i
    // with R_i the running sum of g.q and tau_{i+1} the time of the next processed sweep of
    // the same axis (0 after the very last one).  So Ax/Ay double as R and are never reset.
    // coefficient rows of the current sweep; an x sweep that follows its twin (see pair_x) keeps them: every dword an
    // LDS read returns costs the SIMD ~6 cycles (tools/ubench/simd_share.hip), the three rows of a sweep are 37 % of an
    // x sweep's time
    float ce[M], cinv[M], ckap[M], cjn = 0.f;
    bool abl_loaded = false;
#ifdef PDE_STAMP
    // phase stamps of waves 0 and 4 of one workgroup during one barrier interval: [wave/4][item in step][phase] at dbg + 1024
    bool stamp_on = false;
    unsigned long long* stamps = nullptr;
#endif
    auto body = [&](auto AXC, int axr, int s, const float* rec, float dts, auto TWINC) __attribute__((always_inline)) {
        constexpr int AX = decltype(AXC)::value;
        const int axs = (AX >= 0) ? AX : axr;
        if (more && s == (axs == PDE_AXIS_Y ? first_y : first_x)) dts -= (axs == PDE_AXIS_Y ? tlast_y : tlast_x);
        const float* crow = rec + l * kLineStride + hf * kHalfPad;
        // the newest sweep of my chunk has no twin before it (the previous item belongs to another chunk)
#ifndef PDE_BWD_NO_TWIN
#define PDE_BWD_NO_TWIN 0   // 1: every sweep reads its own coefficient rows (no rows kept across the twin x sweeps)
#endif
        const bool twin = !PDE_BWD_NO_TWIN && decltype(TWINC)::value && a.pair_x != 0 && s != a.S - 1;
        const bool abl_skip = (PDE_ABL & 1) && abl_loaded;
        abl_loaded = true;
        if (!twin && !abl_skip) cjn = rec[kB_Jn + l];
        PDE_STAMP_AT(0);
        if (axs == PDE_AXIS_Y) {
            if (!(PDE_ABL & 2)) relayout_all<N, J>(r, T, l, hf);
            PDE_STAMP_AT(1);
            if (!abl_skip) {
                load_half<M>(crow + kB_E, ce);
                load_half<M>(crow + kB_Inv, cinv);
            }
            solve_adj<M, J>(r, ce, cinv, cjn, hf);
            PDE_STAMP_AT(2);
            if (!(PDE_ABL & 2)) relayout_all<N, J>(r, T, l, hf);
            PDE_STAMP_AT(3);
            if (!abl_skip) load_half<M>(crow + kB_KapX, ckap);
            if constexpr (!MASKED && PDE_YLDS) state_y_lds<N, J>(r, x, Ay, ckap, T, l, hf);
            else state_y<N, J, MASKED>(r, x, Ay, ckap, rec, l, hf, a.smooth3);
            PDE_STAMP_AT(4);
            if (dts != 0.f && !(PDE_ABL & 16)) {
#pragma unroll
                for (int k = 0; k < M; ++k) Ty[k] = fmaf(dts, Ay[k], Ty[k]);
            }
        } else {
            // partner half's innermost state: issue the exchange now, use it after the solve
            const P xin = pk_map(x[M - 1], [&](float z) { return xchg_half(z, hf); });
            if (!twin && !abl_skip) {
                load_half<M>(crow + kB_E, ce);
                load_half<M>(crow + kB_Inv, cinv);
            }
            PDE_STAMP_AT(1);
            solve_adj<M, J>(r, ce, cinv, cjn, hf);
            PDE_STAMP_AT(2);
            if (!twin && !abl_skip) load_half<M>(crow + kB_KapX, ckap);
            PDE_STAMP_AT(3);
            state_x<M, J, MASKED>(r, x, Ax, xin, ckap, rec, l, hf, a.smooth3);
            PDE_STAMP_AT(4);
            if (dts != 0.f && !(PDE_ABL & 16)) {
#pragma unroll
                for (int k = 0; k < M; ++k) Tx[k] = fmaf(dts, Ax[k], Tx[k]);
            }
        }
        PDE_STAMP_AT(5);
        // x now holds the rebuilt state after sweep s-1; take the checkpoint instead if there is one
#ifndef PDE_NO_CKPT
#define PDE_NO_CKPT 0
#endif
        if (!PDE_NO_CKPT && s > 0 && a.ckpt != nullptr && ck_bit(a.ck, s - 1)) {
            const float* slot = a.ckpt + (size_t)ck_slot(a.ck, s - 1) * a.B * a.C * plane;
            load_planes<N, J, float>(slot, q, wave, lane, l, hf, a.B, a.C, c, T, x);
            const float sc = tab->ysc[s - 1];             // true state -> rescaled state of sweep s-1
#pragma unroll
            for (int k = 0; k < M; ++k) x[k] = x[k] * pk_bc<P>(sc);
        }
    };
    // Plane traffic of a chunk.  All 2J planes of a chunk are fetched together (their destination registers are dead at that
    // point): one memory round trip instead of 2J (-21 us of 345 on the 512x64x32x32 launch).
    constexpr int PPI_ = kWaves * J;
    auto fetch_J = [&](const IO* base, int qq, float4 (&raw)[J][Geo<N>::kLoads]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int b = qq * PPI_ + wave * J + j;
            plane_fetch<N, IO>(base + ((size_t)b * a.C + c) * (size_t)(N * N), b < a.B, lane, raw[j]);
        }
    };
    auto place_J = [&](const float4 (&raw)[J][Geo<N>::kLoads], P (&dst)[M]) __attribute__((always_inline)) {
        sfor<0, J>([&](auto CC) __attribute__((always_inline)) {
            plane_to_rows<N, decltype(CC)::value>(raw[decltype(CC)::value], T, lane, l, hf, dst);
        });
    };
    auto chunk_in = [&]() __attribute__((always_inline)) {
        if ((PDE_ABL & 4) && abl_loaded) return;
        float4 rg[J][Geo<N>::kLoads], ry[J][Geo<N>::kLoads];
        fetch_J(gy, q, rg);
        fetch_J(yy, q, ry);
        place_J(rg, r);
        place_J(ry, x);
    };
    auto chunk_out = [&]() __attribute__((always_inline)) {
        if ((PDE_ABL & 4) && more) return;
#pragma unroll
        for (int k = 0; k < M; ++k) r[k] = r[k] * pk_bc<P>(a.gu_scale);      // undo the (1+eps) carried per sweep
        store_planes<N, J, IO>(gu, q, wave, lane, l, hf, a.B, a.C, c, T, r);
    };
    // end of chunk q: store its result; the next chunk comes in at the top of its first item.
    // (Requesting the next chunk's planes BEFORE the stores — into the registers of the dead state — or loading them right
    // here was tried: hipcc then spills 110-190 registers of this kernel and the launch takes twice as long.)
    auto chunk_turn = [&]() __attribute__((always_inline)) {
        chunk_out();
        q += a.G;
        more = q + a.G < nchunk;
    };

    if constexpr (ST::kStep) {
        // Sweeps are numbered along the whole job of this workgroup in processing order (item i = sweep
        // S-1 - i % S of my chunk i / S).  A barrier interval covers one time step (SPS items): the lower
        // waves run items SPS*t .. SPS*t+SPS-1, the upper waves the same window shifted back by one item
        // (phase skew: they are in an x sweep, VALU, while the lower ones are in the y sweep, LDS
        // re-layouts, and the other way round), and everybody's DMA brings in the records of the next
        // interval.  Live at any time: 2*SPS+1 consecutive items = the ring (slot = item % kSlots).
        constexpr int kSlots = ST::kSlots;
        const int nmine = (nchunk - g + a.G - 1) / a.G;                  // my chunks: g, g+G, ...
        const int ntot = nmine * a.S;
        const int nint = ntot / SPS + 1;
        int dslot = 0, dsw = a.S - 1, ditem = 0;                         // next record to fetch
        auto dma_next = [&]() __attribute__((always_inline)) {
#pragma unroll 1
            for (int i = 0; i < SPS; ++i) {
                if (ditem < ntot) dma_rec(dslot, dsw);
                ++ditem;
                dslot = (dslot + 1 == kSlots) ? 0 : dslot + 1;
                dsw = (dsw == 0) ? a.S - 1 : dsw - 1;
            }
        };
        auto dma_one = [&]() __attribute__((always_inline)) {
            if (ditem < ntot) dma_rec(dslot, dsw);
            ++ditem;
            dslot = (dslot + 1 == kSlots) ? 0 : dslot + 1;
            dsw = (dsw == 0) ? a.S - 1 : dsw - 1;
        };
        if (PDE_BWD_ITEMB) dma_one(); else dma_next();
        dma_wait_all();
        __syncthreads();
            // The two halves run the same loop with their own compile-time pattern (position of an item
        // inside its time step -> axis, chunk boundaries): x y x | x y x for the lower waves, the same
        // sequence cut one item earlier for the upper ones.
        auto run = [&](auto LAGC) __attribute__((always_inline)) {
            constexpr int LAG = decltype(LAGC)::value;
            int item = -LAG, s = a.S - 1, slot = 0;       // my next item, its sweep, its ring slot
#pragma unroll 1
            for (int t = 0; t < nint; ++t) {
                if (!resident && !PDE_BWD_ITEMB && !((PDE_ABL & (8 | 64)) && t > 1)) dma_next();
                sfor<0, SPS>([&](auto IC) __attribute__((always_inline)) {
                    constexpr int pos = (decltype(IC)::value - LAG + SPS) % SPS;     // 0: newest sweep of a step
                    constexpr int AX = ((SPS - 1 - pos) == 1) ? PDE_AXIS_Y : PDE_AXIS_X;
                    if (item >= 0 && item < ntot) {
#ifdef PDE_STAMP
                        stamp_on = (blk == 5 && t == 13 && (wave == 0 || wave == 4) && lane == 0);
                        stamps = reinterpret_cast<unsigned long long*>(a.dbg) + 128 + ((wave >> 2) * SPS + decltype(IC)::value) * 6;
#endif
                        if (PDE_BWD_ITEMB && !resident) dma_one();
                        if constexpr (pos == 0) {
                            if (s == a.S - 1) chunk_in();
                        }
                        // Strang, newest sweep of a step (an x sweep): its record equals that of the sweep processed just
                        // before it, the first x sweep of the next step
                        body(std::integral_constant<int, AX>{}, AX, s, cbuf + (size_t)slot * RECP, dts_of(s),
                             std::bool_constant<(SPLIT == kSplitStrang && pos == 0 && !MASKED)>{});
                        if constexpr (pos == SPS - 1) {
                            if (s == 0) {
                                chunk_turn();
                                s = a.S;
                            }
                        }
                        --s;
                        slot = (slot + 1 == (resident ? SPS : kSlots)) ? 0 : slot + 1;
                        if (PDE_BWD_ITEMB && !resident) {
                            dma_wait_all();
                            __syncthreads();
                        }
                    }
                    ++item;
                });
                if (resident || PDE_BWD_ITEMB || ((PDE_ABL & 8) && t > 1)) continue;   // nothing staged, nothing shared: no wait, no barrier
#ifdef PDE_STAMP
                // every wave of one workgroup, intervals 2 and 3: work done, DMA landed, barrier passed
                const bool tl_on = (blk == 5 && (t == 2 || t == 3) && lane == 0);
                unsigned long long* tl = reinterpret_cast<unsigned long long*>(a.dbg) + (wave * 2 + (t - 2)) * 3;
                if (tl_on) tl[0] = stamp();
#endif
                dma_wait_all();                           // my DMA pieces (issued a whole step ago) have landed
#ifdef PDE_STAMP
                if (tl_on) tl[1] = stamp();
#endif
                if (!((PDE_ABL & 32) && t > 1))           // (ablation 32: DMA kept, barrier dropped; 64: the reverse)
                __syncthreads();                          // ... everyone's have, and everyone is done reading
#ifdef PDE_STAMP
                if (tl_on) tl[2] = stamp();
#endif
            }
        };
        if (lag) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 0>{});
    } else {
        Staged stg;
        stg.r0 = stg.r1 = stg.r2 = stg.r3 = stg.r4 = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned n = 0;                                   // buffer parity
        stage_load<REC>(a.coef + ((size_t)(a.S - 1) * a.C + c) * kRecStride + kBwdOff, tid, stg);
        stage_store<REC>(cbuf, tid, stg);
        __syncthreads();
        for (; q < nchunk; q += a.G) {
            more = q + a.G < nchunk;
            chunk_in();
            for (int s = a.S - 1; s >= 0; --s) {
                const int snext = (s > 0) ? s - 1 : a.S - 1;
                const bool pre = (s > 0) || more;
                if (pre) stage_load<REC>(a.coef + ((size_t)snext * a.C + c) * kRecStride + kBwdOff, tid, stg);
                body(std::integral_constant<int, -1>{}, tab->axis[s], s, cbuf + (n & 1) * RECP, tab->dts[s], std::false_type{});
                if (pre) stage_store<REC>(cbuf + ((n + 1) & 1) * RECP, tid, stg);
                __syncthreads();
                ++n;
            }
            chunk_out();
        }
    }

    // deterministic reduction of the four sums over the waves of this workgroup
    __syncthreads();
    float* dst = a.part + ((size_t)g * a.C + c) * 4 * kImage;
#pragma unroll
    for (int arr = 0; arr < 4; ++arr) {
        float* row = T + l * kLineStride + hf * kHalfPad;
#pragma unroll
        for (int k = 0; k < M; ++k) row[k] = (arr == 0) ? Ax[k] : (arr == 1) ? Tx[k] : (arr == 2) ? Ay[k] : Ty[k];
        __syncthreads();
        for (int e = tid; e < kImage; e += kThreads) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) sum += tbuf[w * kImage + e];
            dst[arr * kImage + e] = a.acc_part ? dst[arr * kImage + e] + sum : sum;
        }
        __syncthreads();
    }
}

// One launch for both variants: blocks [0, C*G) run the fast body (J planes per lane, compile-time step
// pattern), blocks [C*G, 2*C*G) the MASKED body (per-sweep clamp masks, table-driven); a block whose
// channel belongs to the other variant — decided on the device by the factor kernel — leaves at once.
// (Two launches over the same grid cost ~6 us more: the second one's blocks all start, look, and exit.)
#ifndef PDE_BWD_MINW
#define PDE_BWD_MINW 1
#endif
template <int N, int J, typename IO, int SPLIT>
__global__ __launch_bounds__(kThreads, PDE_BWD_MINW) void adi_bwd_kernel(SweepArgs a) {
    const int nb = a.C * a.G;
    if (a.only_masked) {
#ifndef PDE_BWD_NO_MASKED
        adi_bwd_body<N, 1, IO, true, kSplitAny>(a, (int)blockIdx.x);
#endif
        return;
    }
    if ((int)blockIdx.x < nb) adi_bwd_body<N, J, IO, false, SPLIT>(a, (int)blockIdx.x);
#ifndef PDE_BWD_NO_MASKED        // diagnostic builds: the fast body alone (its own register allocation)
    else adi_bwd_body<N, 1, IO, true, kSplitAny>(a, (int)blockIdx.x - nb);
#endif
}

}  // namespace
}  // namespace pde
