/*
 * pdecnn.h — C ABI of libpdecnn_hip.so: the MI355X (gfx950) implementation of the
 * PDE diffusion-layer hot path of MariMamgo/CNN-with-PDE.
 *
 * Boundary.  The reference has no FFI; its boundary is the nn.Module surface of the
 * layer classes.  This library is what a Python autograd.Function
 * binds with ctypes; every entry point cites the reference code
 * it replaces.  Conventions:
 *   - plain pointers and sizes only; all tensor pointers are DEVICE pointers
 *     (the caller owns every buffer), NCHW contiguous;
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and
 *     returns immediately; no allocation, no synchronisation.  Process-wide state is limited to
 *     bookkeeping that never changes a result: one "dynamic-LDS attribute set" bit per (kernel,
 *     device) behind a mutex, and the optional timing recorder of pde_timing_*;
 *   - return value: 0 on success, a negative PDE_E_* code otherwise (never throws).
 */
#ifndef PDECNN_H
#define PDECNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDE_MAX_SWEEPS 96          /* e.g. 32 Strang steps */
#define PDE_MAX_N 32               /* longest line of the fused implicit kernels (N = 8, 12, ..., 32; the reference uses 28, 32) */
#define PDE_MAX_N_GENERIC 128      /* longest line of the any-size path (one thread per line, plane in LDS)              */

#define PDE_OK 0
#define PDE_E_BADARG (-1)          /* null pointer, non-positive dim, bad enum */
#define PDE_E_UNSUPPORTED_N (-2)   /* whole-schedule calls: N < 2 or > PDE_MAX_N_GENERIC; per-step / one-launch
                                      families: N not one of the fused line lengths */
#define PDE_E_TOO_MANY_SWEEPS (-3)
#define PDE_E_LAUNCH (-4)          /* hipLaunch failed; hipPeekAtLastError/hipGetLastError has details */
#define PDE_E_WORKSPACE (-5)       /* workspace too small / misaligned */

#define PDE_IO_F32 0
#define PDE_IO_BF16 1

#define PDE_AXIS_X 0               /* solve along W with alpha (mnist_test.py:67-98)  */
#define PDE_AXIS_Y 1               /* solve along H with beta  (mnist_test.py:100-133) */

/* One implicit (backward-Euler) sweep of the split scheme, in execution order.
 * Mirrors one diffuse_x/diffuse_y call of the reference time loop
 * (mnist_test.py:50-63, cifar10.py:86-110, cifar_2version.py:81-101). */
typedef struct PdeSweep {
    int32_t axis;       /* PDE_AXIS_X / PDE_AXIS_Y */
    float   delta;      /* (float) time increment of this sweep: dt/2 or dt           */
    float   h2;         /* (float) dx**2 or dy**2: coeff = theta*delta/h2              */
    float   t;          /* (float) current_time at which alpha/beta are evaluated      */
} PdeSweep;

/* Static description of one fused run of sweeps over a (B,C,N,N) tensor. */
typedef struct PdeAdiDesc {
    int32_t B, C, N;            /* H = W = N                                             */
    int32_t io_dtype;           /* PDE_IO_F32 | PDE_IO_BF16 (tensor I/O; math is fp32)   */
    int32_t num_sweeps;
    int32_t smooth3;            /* 1: 3-tap replicate average of the coefficient along the
                                   solve axis (mnist_test.py:135-149)                     */
    int32_t has_clamp_max;      /* 1: clamp(theta, eps, clamp_max) (cifar10.py:60-61)     */
    float   clamp_max;
    float   eps;                /* stability_eps: clamp floor AND the Thomas +eps          */
    PdeSweep sweep[PDE_MAX_SWEEPS];
} PdeAdiDesc;

/* ---- K1: implicit ADI time-stepper -------------------- */

/* Which kernels serve line length N: 1 = fused register-resident sweeps (N = 8, 12, ..., 32: every entry point of
 * this header), 2 = the any-size path (2 <= N <= PDE_MAX_N_GENERIC, any other N: the reference's classes take any `size`,
 * mnist_test.py:12, cifar10.py:25, SVHN.py:13) — pde_adi_forward / pde_adi_backward / pde_adi_kappa_max and their
 * workspace queries only, one thread per line with the reference's own Thomas recurrences (mnist_test.py:151-198); a
 * layer with a channel operator is then composed per step by the caller (pde_channel_mix_* + one-step schedules),
 * 0 = unsupported. */
int pde_adi_line_length_path(int32_t N);

/* Which kernel pde_adi_backward runs for this schedule's unmasked channels: 0 = the HIP kernel (adi_bwd_kernel), 1 = the
 * hand-scheduled gfx950 assembly kernel (csrc/gen_adi_bwd_asm.py: N = 32, fp32 tensors, Strang steps — mnist_test.py:55-63,
 * cifar10.py:84-110 — two or more of them, no checkpoints), 2 = the any-size kernels.  Same arithmetic either way: the
 * adjoint of the time loop and the batch sums of the coefficient gradients; PDE_ASM_BWD=0 in the
 * environment keeps the library on 0. */
int pde_adi_backward_kernel(const PdeAdiDesc* d, int32_t num_checkpoints);

/* Bytes of scratch the forward/backward calls need (256-byte aligned base expected). */
size_t pde_adi_forward_workspace_bytes(const PdeAdiDesc* d);
size_t pde_adi_backward_workspace_bytes(const PdeAdiDesc* d, int32_t num_checkpoints);

/* y = (prod_s (A_s + eps I)^-1) u.  Replaces DiffusionLayer.forward's time loop with its
 * diffuse_x/diffuse_y/thomas_solver_batch calls (mnist_test.py:44-198; cifar10.py:74-211
 * without apply_channel_mixing, which is pde_channel_mix_*).
 * alpha_xxx / beta_xxx: (C,N,N) fp32.  u, y: (B,C,N,N) of io_dtype; y must not alias u.
 * kappa_max: NULL, or a device buffer of num_sweeps floats that receives the maximum
 * coefficient of every sweep (same values as pde_adi_kappa_max, at no extra launch).
 * kappa_max_host: NULL, or PINNED host memory of num_sweeps floats: the maxima are written there by the
 * factorisation's own second kernel when the buffer is mapped into the device's address space and the
 * process sees one device, else copied asynchronously — either way BEFORE the sweep kernel is launched
 * (needs kappa_max).  kappa_event: NULL, or a hipEvent_t the call records on `stream` behind that copy:
 * the host can plan the backward's checkpoints from this call's own coefficients after a wait of
 * microseconds, long before the forward has finished.
 * After the call the workspace holds the factorisation of every sweep; while it stays intact it
 * may be handed to pde_adi_backward as fwd_workspace to skip refactorising. */
int pde_adi_forward(const PdeAdiDesc* d, const void* u, void* y,
                    const float* alpha_base, const float* beta_base,
                    const float* alpha_slope, const float* beta_slope,
                    float* kappa_max, float* kappa_max_host, void* kappa_event,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Exact reverse-mode derivative of pde_adi_forward (the reference gets it from autograd,
 * SURVEY.md §3d).  Inputs: gy = dL/dy, y = forward output.  Outputs: gu = dL/du and the
 * four parameter gradients (C,N,N) fp32 (overwritten, not accumulated).
 * States needed for the coefficient gradients are rebuilt backwards from y
 * (x_{s-1} = (A_s+eps I) x_s).  `ckpt_mask` bit s set means: do NOT rebuild the state
 * after sweep s, read it from a checkpoint instead; the kernel then first recomputes the
 * forward from `u` to write those checkpoints into the workspace.  ckpt_mask == 0 needs
 * neither u nor checkpoint space (u may be NULL).  Bits are given low word first:
 * sweep s is bit (s%64) of ckpt_mask[s/64].
 * Restriction: all sweeps of one axis must share delta/h2 (true for every reference variant:
 * Strang x(dt/2) y(dt) x(dt/2), Lie x(dt/2) y(dt/2)); otherwise PDE_E_BADARG. */
int pde_adi_backward(const PdeAdiDesc* d, const void* gy, const void* y, const void* u,
                     const uint64_t ckpt_mask[2], void* gu,
                     const float* alpha_base, const float* beta_base,
                     const float* alpha_slope, const float* beta_slope,
                     float* g_alpha_base, float* g_beta_base,
                     float* g_alpha_slope, float* g_beta_slope,
                     const void* fwd_workspace /* NULL, or the intact workspace of the matching
                                                  pde_adi_forward call (same desc, same parameters) */,
                     void* workspace, size_t workspace_bytes, void* stream);

/* max over the tensor of coeff_s = theta_s*delta_s/h2_s for every sweep, written to
 * kappa_max[num_sweeps] (device, fp32).  Host code uses it to choose ckpt_mask.  */
int pde_adi_kappa_max(const PdeAdiDesc* d,
                      const float* alpha_base, const float* beta_base,
                      const float* alpha_slope, const float* beta_slope,
                      float* kappa_max, void* stream);

/* ---- K1 as a sequence of per-step launches --------------------------------------------------
 * The variants with a channel operator between the time steps (cifar10.py:91 mixing before every
 * step, SVHN.py:71 coupling after every step) cannot run their whole time loop in one launch.
 * These entry points serve ONE layer call as: one factorisation of the whole schedule, then one
 * sweep launch per step (`sweeps_per_step` consecutive sweeps of `d`, 3 for Strang, 2 for Lie), the
 * backward accumulating the parameter-gradient partial sums across its per-step launches so that
 * pde_adi_param_grads runs once.  `d` is always the descriptor of the WHOLE schedule. */
size_t pde_adi_steps_workspace_bytes(const PdeAdiDesc* d, int32_t sweeps_per_step);
/* zero + factorise every sweep, one sweep table per step; kappa_max as in pde_adi_forward */
int pde_adi_factor_steps(const PdeAdiDesc* d, int32_t sweeps_per_step,
                         const float* alpha_base, const float* beta_base,
                         const float* alpha_slope, const float* beta_slope,
                         float* kappa_max, void* steps_workspace, size_t workspace_bytes, void* stream);
/* y = sweeps of step `step` applied to u */
int pde_adi_forward_step(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t step,
                         const void* u, void* y, const void* steps_workspace, void* stream);
size_t pde_adi_backward_step_workspace_bytes(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t num_checkpoints);
/* gu = adjoint of step `step`; ckpt_mask bits are relative to the step (bit 0 = state after its first
 * sweep).  accumulate = 0 starts the partial sums held in `workspace`, 1 adds to them: use the same
 * workspace for every step of a call. */
int pde_adi_backward_step(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t step,
                          const void* gy, const void* y, const void* u, const uint64_t ckpt_mask[2], void* gu,
                          const void* steps_workspace, void* workspace, size_t workspace_bytes,
                          int32_t accumulate, void* stream);
/* the four parameter gradients from the partial sums accumulated in `workspace` */
int pde_adi_param_grads(const PdeAdiDesc* d, int32_t sweeps_per_step,
                        const float* alpha_base, const float* beta_base,
                        const float* alpha_slope, const float* beta_slope,
                        float* g_alpha_base, float* g_beta_base, float* g_alpha_slope, float* g_beta_slope,
                        const void* steps_workspace, const void* workspace, void* stream);

/* The same sequence looped on the host inside the library: ONE call per layer forward / backward.
 * mode 1: u <- M u before every step (cifar10.py:91, cifar_2version.py:86); 2: after every step
 * (SVHN.py:71).  `states`: K*2 tensors of u's shape and type, K = num_sweeps / sweeps_per_step —
 * states[2k] the output of step k's first operator, states[2k+1] of its second; the layer output is
 * states[2K-1].  The backward needs them intact, and `u`.  ckpt_mask is relative to a step.
 * pde_adi_mixed_one_launch: 1 when the forward of (d, sweeps_per_step) runs as the factorisation plus ONE launch
 * (fp32 tensors, C = 32 or 64, N = 28 or 32, every step x,y,x or x,y: a workgroup owns all channels of a sample for
 * the whole time loop and mixes them with the fp32 MFMA through LDS; the environment variable PDE_WIDE=0 turns it
 * off), else 0: one mixing launch and one sweep launch per step.  In the one-launch case the forward writes only
 * the sweep output of every step (states[2k+1] for mode 1, states[2k] for mode 2) and states[2K-1]; the backward
 * recomputes the other slots itself when its checkpoints need them, so the contract above is unchanged. */
int pde_adi_mixed_one_launch(const PdeAdiDesc* d, int32_t sweeps_per_step);
int pde_adi_mixed_forward(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t mode,
                          const void* u, void* states,
                          void* y /* NULL, or where the layer output goes INSTEAD of states[2K-1] (`states` then needs 2K-1
                                     tensors only; hand the same `y` to pde_adi_mixed_backward) */,
                          const float* M,
                          const float* alpha_base, const float* beta_base,
                          const float* alpha_slope, const float* beta_slope,
                          float* kappa_max, float* kappa_max_host, void* kappa_event /* as in pde_adi_forward */,
                          void* steps_workspace, size_t workspace_bytes, void* stream);
size_t pde_adi_mixed_backward_workspace_bytes(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t num_checkpoints);
int pde_adi_mixed_backward(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t mode,
                           const void* gy, const void* u, const void* states, const void* y /* as in the forward */,
                           const float* M,
                           const uint64_t ckpt_mask[2], void* gu,
                           const float* alpha_base, const float* beta_base,
                           const float* alpha_slope, const float* beta_slope,
                           float* g_alpha_base, float* g_beta_base, float* g_alpha_slope, float* g_beta_slope,
                           float* gM, const void* steps_workspace,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- K1, the whole layer in ONE launch per pass (C <= 4) -------------------------------------------
 * The reference's own models run these layers at C = 3 (cifar10.py:253-258: mixing before every step;
 * SVHN.py:238: coupling after every step, then the skip blend :73-74); per-step launches are bound by the
 * host's launch rate there.  Here a workgroup owns all C channels of its samples (one wave per channel) and
 * applies the C x C operator in registers at the step boundaries, so the forward is the factorisation kernel
 * plus one launch, the backward one launch plus the gradient epilogue.
 * pde_adi_small_supported: 1 when (d, sweeps_per_step) can take this path (C <= 4; N = 16, 28 or 32; every
 * step x,y,x or x,y), else 0 — callers then use pde_adi_mixed_*.
 * mode as in pde_adi_mixed_forward.  skip_weight: NULL, or (mode 2 only) the device scalar of SVHN.py:36: the
 * output is sigmoid(w) u + (1 - sigmoid(w)) u_K.  states: NULL (inference), or K tensors of u's shape and
 * type that receive the sweep output of every step (the backward needs them).  steps_workspace:
 * pde_adi_steps_workspace_bytes(); kappa_*: as in pde_adi_forward. */
int pde_adi_small_supported(const PdeAdiDesc* d, int32_t sweeps_per_step);
int pde_adi_small_forward(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t mode,
                          const void* u, void* y, void* states, const float* M, const float* skip_weight,
                          const float* alpha_base, const float* beta_base,
                          const float* alpha_slope, const float* beta_slope,
                          float* kappa_max, float* kappa_max_host, void* kappa_event,
                          void* steps_workspace, size_t workspace_bytes, void* stream);
size_t pde_adi_small_backward_workspace_bytes(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t num_checkpoints);
/* Exact reverse-mode derivative of pde_adi_small_forward: gu, the four coefficient gradients, gM (C,C) and,
 * with a skip blend, *g_skip_weight; all overwritten.  ckpt_mask is relative to a step (bit i: keep the state
 * after sweep i of every step instead of rebuilding it; the call then first re-runs the forward to park them). */
int pde_adi_small_backward(const PdeAdiDesc* d, int32_t sweeps_per_step, int32_t mode,
                           const void* gy, const void* u, const void* states, const float* M, const float* skip_weight,
                           const uint64_t ckpt_mask[2], void* gu,
                           const float* alpha_base, const float* beta_base,
                           const float* alpha_slope, const float* beta_slope,
                           float* g_alpha_base, float* g_beta_base, float* g_alpha_slope, float* g_beta_slope,
                           float* gM, float* g_skip_weight, const void* steps_workspace,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- K1, layers that share an input, in ONE launch per pass ------------------------
 * cifar10.py:272-274 runs three EnhancedDiffusionLayers (5, 8 and 4 steps with their own dt/dx and their own
 * parameters) on the same x and combines them with softmax weights (:277-280); cifar_2version.py:287-288 two.
 * pde_adi_multi_forward runs up to 4 such layers (C <= 4, mode 1) inside one launch — every workgroup walks its
 * samples through layer after layer, or, while the batch alone does not fill the chip (B < 1024), the layers run side
 * by side (one layer per workgroup; the terms of `out` / `gu` meet in a small second launch, in a fixed order) — and
 * writes out = sum_i weight_i * y_i; the y_i themselves are the last of each layer's `states`.  One layer with
 * weight 1 is pde_adi_small_forward.
 * The backward takes gy = dL/dout and/or per layer gys = dL/dy_i, and gives gu = sum_i gu_i, every layer's
 * parameter gradients, and g_weight = <gy, y_i>. */
typedef struct PdeSmallLayer {
    const PdeAdiDesc* desc;          /* the layer's own schedule; B, C, N, io_dtype equal across the layers  */
    int32_t sweeps_per_step;         /* equal across the layers (all Strang or all Lie)                       */
    int32_t mode;                    /* 1 | 2 as in pde_adi_mixed_forward; several layers: 1 only             */
    const float* M;                  /* (C,C)                                                                  */
    const float* skip_weight;        /* NULL | device scalar (mode 2)                                          */
    const float* alpha_base; const float* beta_base; const float* alpha_slope; const float* beta_slope;
    float weight;                    /* weight_i ...                                                           */
    const float* weight_ptr;         /* ... or, when not NULL, a device scalar holding it (no host round trip)  */
    void* states;                    /* K_i tensors (forward: NULL = inference)                                */
    float* plane_sums;               /* NULL | (B,C): forward writes sum over the plane of y_i (the adaptive average
                                        pool of cifar10.py:239 times H*W, at no extra pass)                      */
    void* steps_workspace; size_t steps_workspace_bytes;     /* pde_adi_steps_workspace_bytes(desc, sps)      */
    float* kappa_max; float* kappa_max_host;                 /* optional, as in pde_adi_forward               */
    /* backward only */
    const void* gys;                 /* NULL | dL/dy_i                                                         */
    const float* g_plane_sums;       /* NULL | (B,C) dL/d(plane_sums_i): added to every element of the plane    */
    const uint64_t* ckpt_mask;       /* relative to a step                                                     */
    float* g_alpha_base; float* g_beta_base; float* g_alpha_slope; float* g_beta_slope;
    float* gM; float* g_skip_weight; float* g_weight;        /* g_weight optional                              */
    void* workspace; size_t workspace_bytes;                 /* pde_adi_small_backward_workspace_bytes()      */
} PdeSmallLayer;
int pde_adi_multi_forward(int32_t num_layers, const PdeSmallLayer* layers, const void* u, void* out,
                          void* kappa_event, void* stream);
int pde_adi_multi_backward(int32_t num_layers, const PdeSmallLayer* layers, const void* gy, const void* u,
                           void* gu, void* stream);

/* ---- channel operators ------------------------------------------ */

/* out[b,i,p] = sum_j M[i,j] u[b,j,p]  — cifar10.py:65-72 apply_channel_mixing and
 * SVHN.py:78-86 apply_channel_coupling (both reduce to this).  M: (C,C) fp32 row-major.
 * u,out: (B,C,HW) of io_dtype; out must not alias u. */
int pde_channel_mix_forward(int32_t B, int32_t C, int32_t HW, int32_t io_dtype,
                            const void* u, const float* M, void* out, void* stream);
/* gu[b,j,p] = sum_i M[i,j] gout[b,i,p];  gM[i,j] = sum_{b,p} gout[b,i,p] u[b,j,p].
 * workspace: pde_channel_mix_backward_workspace_bytes(). */
size_t pde_channel_mix_backward_workspace_bytes(int32_t B, int32_t C, int32_t HW);
int pde_channel_mix_backward(int32_t B, int32_t C, int32_t HW, int32_t io_dtype,
                             const void* u, const void* gout, const float* M,
                             void* gu, float* gM,
                             void* workspace, size_t workspace_bytes, void* stream);
/* The same with gM spread over several calls that share `workspace` (one per time step of a layer):
 * accumulate = 0 starts the partial sums, 1 adds to them; finalize = 1 reduces them into gM
 * (gM may be NULL otherwise). */
int pde_channel_mix_backward_steps(int32_t B, int32_t C, int32_t HW, int32_t io_dtype,
                                   const void* u, const void* gout, const float* M,
                                   void* gu, float* gM,
                                   void* workspace, size_t workspace_bytes,
                                   int32_t accumulate, int32_t finalize, void* stream);

/* SVHN.py:73-74 skip connection: out = s*u0 + (1-s)*u with s = sigmoid(*skip_weight) (device scalar), n
 * elements of io_dtype, one pass.  Backward: g_u0 = s*g, g_u = (1-s)*g,
 * *g_skip_weight = s(1-s) * sum g*(u0-u) (deterministic two-stage sum). */
int pde_skip_blend_forward(int64_t n, int32_t io_dtype, const void* u0, const void* u,
                           const float* skip_weight, void* out, void* stream);
size_t pde_skip_blend_backward_workspace_bytes(int64_t n);
int pde_skip_blend_backward(int64_t n, int32_t io_dtype, const void* g, const void* u0, const void* u,
                            const float* skip_weight, void* g_u0, void* g_u, float* g_skip_weight,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ---- epilogue behind the shared-input layers -------------------------------------------
 * cifar10.py:270-280: features_i = y_i * gate_i[b,c] (SpatialAttention :232-244), combined = sum_i w_i features_i.
 * The gates are a small MLP of the average pool of y_i + pos_embed: the pool comes out of pde_adi_multi_forward
 * (PdeSmallLayer.plane_sums), the (B,C) MLP stays with the caller; these two calls are the passes over the full
 * tensors that remain.  ys: L <= 4 tensors (B,C,HW) of io_dtype; gates: L arrays (B,C) fp32; weights: (L) fp32 on
 * the device; HW a multiple of 4.
 *   forward : out[b,c,p]  = sum_i weights[i] gates[i][b,c] ys[i][b,c,p]
 *   backward: gys[i][b,c,p] = weights[i] gates[i][b,c] g[b,c,p];  dots[i][b,c] = sum_p g[b,c,p] ys[i][b,c,p]
 *             (dL/dgate_i = weights[i] dots[i], dL/dweights[i] = sum_bc gates[i] dots[i]). */
int pde_gate_combine_forward(int32_t L, int32_t B, int32_t C, int32_t HW, int32_t io_dtype,
                             const void* const* ys, const float* const* gates, const float* weights,
                             void* out, void* stream);
int pde_gate_combine_backward(int32_t L, int32_t B, int32_t C, int32_t HW, int32_t io_dtype, const void* g,
                              const void* const* ys, const float* const* gates, const float* weights,
                              void* const* gys, float* const* dots, void* stream);

/* ---- what follows the feature extractor in cifar10.CIFAR10PDENoConv ----------------------
 * cifar10.py:346-353: features = BatchNorm2d(combined); pooled = cat([AdaptiveAvgPool2d(4,4)(features),
 * AdaptiveMaxPool2d(4,4)(features)], dim=1).  `features` is never written: out (B,2C,4,4) fp32 comes straight from x
 * (B,C,N,N) fp32, N a multiple of 4 and <= 64.  training != 0: statistics of the batch (biased variance for the
 * normalisation; running_mean / running_var, when given, updated with `momentum` and the unbiased variance, as
 * torch.nn.BatchNorm2d does); else the running statistics.  gamma / beta may be NULL (1 / 0).  mean, invstd: (C)
 * outputs the backward needs, argmax: (B,C,4,4) int32 positions of the maxima inside their planes.
 * backward: gout (B,2C,4,4) -> gx, ggamma, gbeta (all overwritten). */
size_t pde_bn_pool_workspace_bytes(int32_t B, int32_t C);
int pde_bn_pool_forward(int32_t B, int32_t C, int32_t N, const float* x, const float* gamma, const float* beta,
                        float eps, int32_t training, float momentum, float* running_mean, float* running_var,
                        float* mean, float* invstd, float* out, int32_t* argmax,
                        void* workspace, size_t workspace_bytes, void* stream);
int pde_bn_pool_backward(int32_t B, int32_t C, int32_t N, const float* x, const float* gamma, const float* mean,
                         const float* invstd, const int32_t* argmax, const float* gout, int32_t training,
                         float* gx, float* ggamma, float* gbeta,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- K2: explicit 5-point layers --------------------------- */

/* tiny_imagenet.py:34-72: `num_steps` relaxed explicit steps (the reference's loop :44-49), each
 *   a_c = clamp(alpha_base_c, eps, max_coeff);  v = s_c u;
 *   u <- u + relax*(v + a_c*dt*Lap0(v) - u)      (Lap0: zero ghost cells, padding=1)
 * u,out: (B,C,H,W) of io_dtype.  states: NULL, or room for (num_steps-1) FP32 tensors of u's shape (whatever io_dtype:
 * with bf16 tensors only the layer's own input, output and gradients are bf16) that receive the inputs of steps
 * 2..num_steps (what pde_explicit5_backward needs); required for num_steps > 1 unless the plane is 64x64, 32x32 or 16x16
 * (those stay in registers over all steps, one launch). */
int pde_explicit5_forward(int32_t B, int32_t C, int32_t H, int32_t W, int32_t io_dtype,
                          const void* u, const float* alpha_base, const float* channel_scaling,
                          float dt, float eps, float max_coeff, float relax,
                          int32_t num_steps, void* states, void* out, void* stream);
size_t pde_explicit5_backward_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W, int32_t io_dtype,
                                              int32_t num_steps);
/* gu, g_alpha_base (C), g_channel_scaling (C): overwritten.  states: as written by the forward call
 * (may be NULL when num_steps == 1). */
int pde_explicit5_backward(int32_t B, int32_t C, int32_t H, int32_t W, int32_t io_dtype,
                           const void* u, const void* states, const void* gout,
                           const float* alpha_base, const float* channel_scaling,
                           float dt, float eps, float max_coeff, float relax, int32_t num_steps,
                           void* gu, float* g_alpha_base, float* g_channel_scaling,
                           void* workspace, size_t workspace_bytes, void* stream);

/* emotion_recognition.py:82-97: reflect-pad once, nt Jacobi updates of the interior with
 * row coefficients a_row[H] (multiplying the second difference along H) and column
 * coefficients b_col[W] (along W); the padded ring keeps its initial values.
 * u,out: (B,H,W) fp32 (the layer is single-channel).  H,W <= 64. */
int pde_jacobi_forward(int32_t B, int32_t H, int32_t W, int32_t nt,
                       const float* u, const float* a_row, const float* b_col,
                       float* out, void* stream);
size_t pde_jacobi_backward_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t nt);
/* gu (B,H,W), g_a_row[H], g_b_col[W]: overwritten. */
int pde_jacobi_backward(int32_t B, int32_t H, int32_t W, int32_t nt,
                        const float* u, const float* gout,
                        const float* a_row, const float* b_col,
                        float* gu, float* g_a_row, float* g_b_col,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- Ruthotto-Haber symmetric layer on the fp32 matrix cores ------------- */

/* cifar_2version.py:190-220 SymmetricLayer.forward and the residual steps built on it (ParabolicBlock :223-236,
 * HamiltonianBlock :239-258), fused:
 *     P = X K^T;  H = act(BatchNorm1d(P));  out = base + scale * (H K)
 * X, base, out, P, H: (B, D) fp32 row-major (the image batch flattened, D = C*H*W); K: (D, D) = nn.Linear weight;
 * bn_weight / bn_bias / running_mean / running_var: (D).  act: 0 identity, 1 relu, 2 tanh (cifar_2version.py:203-208).
 * training != 0: batch statistics (biased variance for normalisation; running statistics updated with `momentum`
 * and the unbiased variance, as torch.nn.BatchNorm1d does; running_* may be NULL = not tracked); training == 0:
 * running statistics.  P, H, mean[D], invstd[D] are written for the backward; base may be NULL (then out = scale*(H K)).
 * F_sym(Y) itself is base = NULL, scale = -1.  D a multiple of 64 (pde_sym_layer_supported); up to 128 batch rows the
 * statistics are the epilogue of the first product, larger batches run it by row blocks with a statistics pass beside it.
 * workspace: NULL, or pde_sym_layer_workspace_bytes(B, D) bytes of scratch (16-byte aligned; contents do not matter, no
 * two calls that share it in flight at once): with it, batches up to 128 rows run each product as 32-column strips whose
 * contraction is split over workgroups, the partial tiles added in a fixed order by a small second launch that also
 * carries the epilogue.  Without it (or where pde_sym_layer_workspace_bytes is 0): one workgroup per 16-column strip. */
int pde_sym_layer_supported(int32_t B, int32_t D);
size_t pde_sym_layer_workspace_bytes(int32_t B, int32_t D);
int pde_sym_layer_forward(int32_t B, int32_t D, int32_t act, int32_t training,
                          const float* X, const float* K, const float* bn_weight, const float* bn_bias,
                          float* running_mean, float* running_var, float momentum, float eps,
                          const float* base, float scale,
                          float* P, float* H, float* mean, float* invstd, float* out,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Backward of the above for an upstream gradient g_out (B, D) of `out` (the gradient of `base` is g_out itself and is
 * left to the caller).  dP: (B, D) scratch.  Overwritten: gX (B, D), gK (D, D) — both uses of K —, g_bn_weight[D],
 * g_bn_bias[D].  workspace: as in pde_sym_layer_forward (the same one may serve both). */
int pde_sym_layer_backward(int32_t B, int32_t D, int32_t act, int32_t training,
                           const float* g_out, float scale, const float* X, const float* K, const float* bn_weight,
                           const float* P, const float* H, const float* mean, const float* invstd,
                           float* dP, float* gX, float* gK, float* g_bn_weight, float* g_bn_bias,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- utilities ------------------------------------------------------------------------- */

/* Average device time (ms) per launch of the dominant kernel of the most recent
 * pde_adi_forward / pde_adi_backward issued with timing enabled; measured with HIP
 * events recorded on the stream the kernel was launched on.  bench.py's roofline leg. */
int pde_timing_enable(int32_t on);
int pde_timing_read(double* fwd_ms_sum, int64_t* fwd_launches, double* bwd_ms_sum, int64_t* bwd_launches);

const char* pde_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PDECNN_H */
