// Native host path of the launch-bound layer calls: the autograd node of one implicit diffusion layer
// (functional.adi_diffuse: mnist_test.py:44-198, fashion_mnist.py:18-196, the C = 1 shapes the reference trains on) written
// against torch's C++ autograd and the C ABI of include/pdecnn.h.  Same calls, same arguments and the same order of
// launches as functional._AdiFn (which stays the general path: coefficient-maxima sinks, lagged plans); what goes is the
// interpreter between them — at (64,1,28,28) the kernels of a forward + backward take 0.1 ms on the device and the Python
// around them 0.17 ms on the host.
//
// Nothing here computes: tensors are allocated through torch's caching allocator, every result comes from the HIP kernels
// of libpdecnn_hip.so on torch's current stream.
#include <torch/extension.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <map>
#include <tuple>
#include <cstring>
#include <stdexcept>
#include <string>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/pdecnn.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

const char* err_text(int rc) {
    switch (rc) {
        case -1: return "PDE_E_BADARG (null pointer, bad dimension or enum)";
        case -2:
            return "PDE_E_UNSUPPORTED_N (line length outside [2, 128], or a per-step / one-launch entry point at a line length "
                   "without fused kernels: those exist for multiples of 4 in [8, 32])";
        case -3: return "PDE_E_TOO_MANY_SWEEPS";
        case -4: return "PDE_E_LAUNCH (HIP launch failed)";
        case -5: return "PDE_E_WORKSPACE (workspace too small or misaligned)";
    }
    return "unknown error";
}

// A failed call of the C ABI: reaches Python as _lib.PdeError (a RuntimeError), like the ctypes path's
struct PdeFailure : public std::runtime_error {
    using std::runtime_error::runtime_error;
};
PyObject* g_error_class = nullptr;

#define PDE_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) throw PdeFailure(c10::str(__VA_ARGS__));   \
    } while (0)

void check(int rc, const char* what) {
    if (rc != 0) throw PdeFailure(std::string(what) + " failed: " + err_text(rc));
}

// ---- the way the per-sweep coefficient maxima reach the host: pinned slots, one event each -------------------------------
// pde_adi_forward writes the maxima into the slot and records the event right behind its factorisation kernel; the
// backward plans its checkpoints from them (functional.plan_checkpoints).  A slot is held from a forward until its node
// dies (a tensor over the slot's memory, kept in the node, hands it back from its deleter).
struct Slot {
    float* host = nullptr;
    hipEvent_t ev = nullptr;
    std::atomic<bool> busy{false};
    bool pooled = true;
};

constexpr int kRing = 256;
constexpr int kSlotFloats = PDE_MAX_SWEEPS * 4;

struct Ring {
    std::mutex mu;
    std::vector<std::unique_ptr<Slot>> slots;
    float* pool = nullptr;
    int next = 0;
};

Ring* ring_of(int dev) {
    // (never destroyed: a slot's ticket may be released — its deleter run — while the process is exiting, after static
    //  destructors of this library would have run)
    static std::mutex& mu = *new std::mutex();
    static std::vector<std::unique_ptr<Ring>>& rings = *new std::vector<std::unique_ptr<Ring>>();
    std::lock_guard<std::mutex> g(mu);
    if ((int)rings.size() <= dev) rings.resize(dev + 1);
    if (!rings[dev]) {
        auto r = std::make_unique<Ring>();
        TORCH_CHECK(hipHostMalloc((void**)&r->pool, sizeof(float) * kSlotFloats * kRing, hipHostMallocDefault) == hipSuccess,
                    "hipHostMalloc of the coefficient-maxima ring failed");
        for (int i = 0; i < kRing; ++i) {
            auto s = std::make_unique<Slot>();
            s->host = r->pool + (size_t)i * kSlotFloats;
            TORCH_CHECK(hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
            r->slots.push_back(std::move(s));
        }
        rings[dev] = std::move(r);
    }
    return rings[dev].get();
}

Slot* acquire_slot(int dev) {
    Ring* r = ring_of(dev);
    {
        std::lock_guard<std::mutex> g(r->mu);
        for (int off = 0; off < kRing; ++off) {
            int i = (r->next + off) % kRing;
            Slot* s = r->slots[i].get();
            if (!s->busy.load(std::memory_order_acquire)) {
                s->busy.store(true, std::memory_order_release);
                r->next = (i + 1) % kRing;
                return s;
            }
        }
    }
    // every slot is held by a node whose backward is outstanding: a slot of its own, freed with the node
    Slot* s = new Slot();
    s->pooled = false;
    s->busy.store(true);
    TORCH_CHECK(hipHostMalloc((void**)&s->host, sizeof(float) * kSlotFloats, hipHostMallocDefault) == hipSuccess,
                "hipHostMalloc failed");
    TORCH_CHECK(hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
    return s;
}

void release_slot(Slot* s) {
    if (s->pooled) {
        s->busy.store(false, std::memory_order_release);
        return;
    }
    (void)hipEventDestroy(s->ev);
    (void)hipHostFree(s->host);
    delete s;
}

// A CPU tensor over the slot's pinned floats whose deleter returns the slot: stored in the node, it ties the slot's
// life to the node's without a class registration.
Tensor slot_ticket(Slot* s, int n) {
    return at::from_blob(s->host, {n}, [s](void*) { release_slot(s); }, at::TensorOptions().dtype(at::kFloat));
}

void wait_event(hipEvent_t ev) {
    // Expected within microseconds of the device reaching the factorisation kernel — but the device may still be busy with
    // the previous step's backward when the host gets here (a device-bound loop: the host runs one pass ahead), so the poll
    // is bounded by TIME, not by a number of queries: a query of a pending event costs 15 ns on some hosts and 1 us on
    // others, and a blocking hipEventSynchronize costs 100-150 us to wake up from (0.60 ms per headline step instead of
    // 0.49 where 20,000 queries ran out first).
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int i = 0; i < 256; ++i)
            if (hipEventQuery(ev) == hipSuccess) return;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
    TORCH_CHECK(hipEventSynchronize(ev) == hipSuccess, "hipEventSynchronize failed");
}

// functional.plan_checkpoints: bit s set = keep the state after sweep s instead of rebuilding it
void plan_checkpoints(const float* kmax, int n, double amax, uint64_t mask[2]) {
    mask[0] = mask[1] = 0;
    double amp = 1.0;
    for (int s = n - 1; s >= 1; --s) {
        amp *= 1.0 + 4.0 * (double)kmax[s];
        if (amp > amax) {
            int b = s - 1;
            mask[b >> 6] |= 1ull << (b & 63);
            amp = 1.0;
        }
    }
}

int popcount2(const uint64_t m[2]) { return __builtin_popcountll(m[0]) + __builtin_popcountll(m[1]); }

// functional._as_chw: (C,N,N) fp32 contiguous view of a coefficient tensor given as (N,N), (1,N,N) or (C,N,N)
Tensor as_chw(const Tensor& p, int64_t C, int64_t N) {
    Tensor q = p.detach();
    if (q.dim() == 2) q = q.unsqueeze(0);
    PDE_REQUIRE(q.dim() == 3 && q.size(0) == C && q.size(1) == N && q.size(2) == N, "coefficient of shape ", p.sizes(),
                " does not match (", C, ",", N, ",", N, ")");
    if (q.scalar_type() != at::kFloat) q = q.to(at::kFloat);
    return q.contiguous();
}

Tensor bytes(size_t n, const Tensor& like) {
    return at::empty({(int64_t)std::max<size_t>(n, 256)}, like.options().dtype(at::kByte));
}

struct AdiFn : public torch::autograd::Function<AdiFn> {
    // ckpt_mode: 0 = explicit mask (ckpt_lo/hi), 1 = "auto" (planned in the backward from this call's coefficients),
    // 2 = explicit mask, and this call's coefficient maxima go to a slot the CALLER owns (slot_out: where to leave it) —
    // the "lagged" policy of layers.py: the next call plans from them, nobody waits
    static Tensor forward(AutogradContext* ctx, const Tensor& u_in, const Tensor& ab, const Tensor& bb, const Tensor& asl,
                          const Tensor& bsl, int64_t desc_addr, int64_t ckpt_mode, int64_t ckpt_lo, int64_t ckpt_hi,
                          double amax, bool need_grad, int64_t slot_out) {
        PDE_REQUIRE(u_in.is_cuda() && ab.is_cuda() && bb.is_cuda() && asl.is_cuda() && bsl.is_cuda(),
                    "libpdecnn_hip operators need CUDA/HIP tensors (there is no CPU fallback)");
        PDE_REQUIRE(u_in.dim() == 4 && u_in.size(2) == u_in.size(3), "expected (B,C,N,N), got ", u_in.sizes());
        PdeAdiDesc d;
        std::memcpy(&d, reinterpret_cast<const void*>(desc_addr), sizeof(d));
        const int64_t B = u_in.size(0), C = u_in.size(1), N = u_in.size(2);
        Tensor u = u_in.detach();
        if (u.scalar_type() != at::kFloat && u.scalar_type() != at::kBFloat16) u = u.to(at::kFloat);
        u = u.contiguous();
        PDE_REQUIRE(d.B == B && d.C == C && d.N == N && d.io_dtype == (u.scalar_type() == at::kBFloat16 ? PDE_IO_BF16 : PDE_IO_F32),
                    "descriptor does not match the tensor");
        Tensor p[4] = {as_chw(ab, C, N), as_chw(bb, C, N), as_chw(asl, C, N), as_chw(bsl, C, N)};
        const bool want_kmax = need_grad && (ckpt_mode == 1 || ckpt_mode == 2);
        c10::hip::HIPGuardMasqueradingAsCUDA guard(u.device());
        hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(u.device().index()).stream();
        Tensor y = at::empty_like(u);
        Tensor ws = bytes(pde_adi_forward_workspace_bytes(&d), u);
        Tensor kdev, ticket;
        Slot* slot = nullptr;
        if (want_kmax) {
            kdev = at::empty({(int64_t)d.num_sweeps}, u.options().dtype(at::kFloat));
            slot = acquire_slot(u.device().index());
            if (ckpt_mode == 1) ticket = slot_ticket(slot, d.num_sweeps);
            else *reinterpret_cast<Slot**>(slot_out) = slot;           // the caller's ticket holds it
        }
        check(pde_adi_forward(&d, u.data_ptr(), y.data_ptr(), p[0].data_ptr<float>(), p[1].data_ptr<float>(),
                              p[2].data_ptr<float>(), p[3].data_ptr<float>(), want_kmax ? kdev.data_ptr<float>() : nullptr,
                              slot ? slot->host : nullptr, slot ? (void*)slot->ev : nullptr, ws.data_ptr(), (size_t)ws.numel(),
                              (void*)st),
              "pde_adi_forward");
        if (need_grad) {
            const bool explicit_none = ckpt_mode == 0 && ckpt_lo == 0 && ckpt_hi == 0;
            ctx->save_for_backward({y, explicit_none ? Tensor() : u, p[0], p[1], p[2], p[3]});
            ctx->saved_data["ws"] = ws;                      // the factorisation, reused by the backward
            if (slot && ckpt_mode == 1) {
                ctx->saved_data["ticket"] = ticket;
                ctx->saved_data["slot"] = (int64_t) reinterpret_cast<intptr_t>(slot);
            }
            if (kdev.defined()) ctx->saved_data["kdev"] = kdev;       // (the device copy lives as long as the node)
            ctx->saved_data["desc"] = std::string(reinterpret_cast<const char*>(&d), sizeof(d));
            ctx->saved_data["ckpt"] = std::vector<int64_t>{ckpt_mode, ckpt_lo, ckpt_hi};
            ctx->saved_data["amax"] = amax;
            ctx->saved_data["sh0"] = ab.sizes().vec();
            ctx->saved_data["sh1"] = bb.sizes().vec();
            ctx->saved_data["sh2"] = asl.sizes().vec();
            ctx->saved_data["sh3"] = bsl.sizes().vec();
        }
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto saved = ctx->get_saved_variables();
        const Tensor &y = saved[0], &u = saved[1];
        PdeAdiDesc d;
        std::memcpy(&d, ctx->saved_data["desc"].toStringRef().data(), sizeof(d));
        auto ck = ctx->saved_data["ckpt"].toIntVector();
        uint64_t mask[2] = {(uint64_t)ck[1], (uint64_t)ck[2]};
        if (ck[0] == 1) {
            Slot* slot = reinterpret_cast<Slot*>((intptr_t)ctx->saved_data["slot"].toInt());
            wait_event(slot->ev);
            plan_checkpoints(slot->host, d.num_sweeps, ctx->saved_data["amax"].toDouble(), mask);
        }
        const bool any = (mask[0] | mask[1]) != 0;
        PDE_REQUIRE(!any || u.defined(), "a checkpoint plan needs the layer input, which was not kept");
        c10::hip::HIPGuardMasqueradingAsCUDA guard(y.device());                 // autograd's worker thread: set the device, fetch the stream here
        hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(y.device().index()).stream();
        Tensor gy = grads[0];
        if (gy.scalar_type() != y.scalar_type()) gy = gy.to(y.scalar_type());
        gy = gy.contiguous();
        Tensor gu = at::empty_like(y);
        static const char* const kShape[4] = {"sh0", "sh1", "sh2", "sh3"};
        Tensor gp[4];
        for (int i = 0; i < 4; ++i) gp[i] = at::empty(ctx->saved_data[kShape[i]].toIntVector(), saved[2 + i].options());
        Tensor ws = bytes(pde_adi_backward_workspace_bytes(&d, popcount2(mask)), y);
        Tensor fws = ctx->saved_data["ws"].toTensor();
        check(pde_adi_backward(&d, gy.data_ptr(), y.data_ptr(), any ? u.data_ptr() : nullptr, mask, gu.data_ptr(),
                               saved[2].data_ptr<float>(), saved[3].data_ptr<float>(), saved[4].data_ptr<float>(),
                               saved[5].data_ptr<float>(), gp[0].data_ptr<float>(), gp[1].data_ptr<float>(),
                               gp[2].data_ptr<float>(), gp[3].data_ptr<float>(), fws.data_ptr(), ws.data_ptr(),
                               (size_t)ws.numel(), (void*)st),
              "pde_adi_backward");
        return {gu, gp[0], gp[1], gp[2], gp[3], Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor adi(const Tensor& u, const Tensor& ab, const Tensor& bb, const Tensor& asl, const Tensor& bsl, int64_t desc_addr,
           int64_t ckpt_mode, int64_t ckpt_lo, int64_t ckpt_hi, double amax) {
    // forward() runs with grad mode off: whether anything is kept for a backward is decided out here
    const bool need_grad = at::GradMode::is_enabled() && (u.requires_grad() || ab.requires_grad() || bb.requires_grad() ||
                                                          asl.requires_grad() || bsl.requires_grad());
    return AdiFn::apply(u, ab, bb, asl, bsl, desc_addr, ckpt_mode, ckpt_lo, ckpt_hi, amax, need_grad, (int64_t)0);
}

// The coefficient maxima of one call, for whoever plans a LATER call from them (layers.py, "lagged" policy): `event`
// / `query()` / `host` as the ctypes path's tickets have them.  Holds its pinned slot until it is dropped.
struct Ticket {
    Slot* slot;
    int n;
    Ticket(Slot* s, int n_) : slot(s), n(n_) {}
    Ticket(const Ticket&) = delete;
    ~Ticket() { release_slot(slot); }
    bool query() const { return hipEventQuery(slot->ev) == hipSuccess; }
    void synchronize() const { wait_event(slot->ev); }
    Tensor host() const { return at::from_blob(slot->host, {n}, at::TensorOptions().dtype(at::kFloat)).clone(); }
};

std::pair<Tensor, std::shared_ptr<Ticket>> adi_lagged(const Tensor& u, const Tensor& ab, const Tensor& bb, const Tensor& asl,
                                                      const Tensor& bsl, int64_t desc_addr, int64_t ckpt_lo, int64_t ckpt_hi) {
    const bool need_grad = at::GradMode::is_enabled() && (u.requires_grad() || ab.requires_grad() || bb.requires_grad() ||
                                                          asl.requires_grad() || bsl.requires_grad());
    Slot* slot = nullptr;
    Tensor y = AdiFn::apply(u, ab, bb, asl, bsl, desc_addr, (int64_t)2, ckpt_lo, ckpt_hi, 0.0, need_grad,
                            (int64_t) reinterpret_cast<intptr_t>(&slot));
    PdeAdiDesc d;
    std::memcpy(&d, reinterpret_cast<const void*>(desc_addr), sizeof(d));
    return {y, slot ? std::make_shared<Ticket>(slot, d.num_sweeps) : std::shared_ptr<Ticket>()};
}


// the union over the steps of the step-local plans (functional._AdiSmallFn.backward)
void plan_steps(const float* kmax, int K, int sps, double amax, uint64_t mask[2]) {
    mask[0] = mask[1] = 0;
    for (int k = 0; k < K; ++k) {
        uint64_t m[2];
        plan_checkpoints(kmax + (size_t)k * sps, sps, amax, m);
        mask[0] |= m[0];
        mask[1] |= m[1];
    }
}

Tensor as_f32(const Tensor& t) {
    Tensor q = t.detach();
    if (q.scalar_type() != at::kFloat) q = q.to(at::kFloat);
    return q.contiguous();
}

std::vector<int64_t> states_shape(int64_t K, const Tensor& u) {
    std::vector<int64_t> sh{K};
    for (auto v : u.sizes()) sh.push_back(v);
    return sh;
}

// ---- one layer with a channel operator between its steps, C <= 4, one launch per pass (functional._AdiSmallFn:
// cifar10.py:84-112 "pre", SVHN.py:55-76 "post" with the skip blend) -----------------------------------------------
struct SmallFn : public torch::autograd::Function<SmallFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& u_in, const Tensor& ab, const Tensor& bb, const Tensor& asl,
                          const Tensor& bsl, const Tensor& M, const std::optional<Tensor>& skip_opt, int64_t desc_addr, int64_t sps,
                          int64_t mode, int64_t ckpt_mode, int64_t ckpt_lo, double amax, bool need_grad) {
        const Tensor skip = skip_opt.has_value() ? *skip_opt : Tensor();
        PDE_REQUIRE(u_in.is_cuda() && ab.is_cuda() && bb.is_cuda() && asl.is_cuda() && bsl.is_cuda() && M.is_cuda() &&
                        (!skip.defined() || skip.is_cuda()),
                    "libpdecnn_hip operators need CUDA/HIP tensors (there is no CPU fallback)");
        PdeAdiDesc d;
        std::memcpy(&d, reinterpret_cast<const void*>(desc_addr), sizeof(d));
        const int64_t B = u_in.size(0), C = u_in.size(1), N = u_in.size(2);
        Tensor u = u_in.detach();
        if (u.scalar_type() != at::kFloat && u.scalar_type() != at::kBFloat16) u = u.to(at::kFloat);
        u = u.contiguous();
        PDE_REQUIRE(d.B == B && d.C == C && d.N == N && d.io_dtype == (u.scalar_type() == at::kBFloat16 ? PDE_IO_BF16 : PDE_IO_F32),
                    "descriptor does not match the tensor");
        const int64_t K = d.num_sweeps / sps;
        Tensor p[4] = {as_chw(ab, C, N), as_chw(bb, C, N), as_chw(asl, C, N), as_chw(bsl, C, N)};
        Tensor Mf = as_f32(M);
        Tensor sw = skip.defined() ? as_f32(skip).reshape({1}) : Tensor();
        const bool want_kmax = need_grad && ckpt_mode == 1;
        c10::hip::HIPGuardMasqueradingAsCUDA guard(u.device());
        hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(u.device().index()).stream();
        Tensor sws = bytes(pde_adi_steps_workspace_bytes(&d, (int32_t)sps), u);
        Tensor kdev, ticket, states;
        Slot* slot = nullptr;
        if (want_kmax) {
            kdev = at::empty({(int64_t)d.num_sweeps}, u.options().dtype(at::kFloat));
            slot = acquire_slot(u.device().index());
            ticket = slot_ticket(slot, d.num_sweeps);
        }
        if (need_grad) states = at::empty(states_shape(K, u), u.options());
        Tensor y = at::empty_like(u);
        check(pde_adi_small_forward(&d, (int32_t)sps, (int32_t)mode, u.data_ptr(), y.data_ptr(),
                                    need_grad ? states.data_ptr() : nullptr, Mf.data_ptr<float>(),
                                    sw.defined() ? sw.data_ptr<float>() : nullptr, p[0].data_ptr<float>(), p[1].data_ptr<float>(),
                                    p[2].data_ptr<float>(), p[3].data_ptr<float>(), want_kmax ? kdev.data_ptr<float>() : nullptr,
                                    slot ? slot->host : nullptr, slot ? (void*)slot->ev : nullptr, sws.data_ptr(),
                                    (size_t)sws.numel(), (void*)st),
              "pde_adi_small_forward");
        if (need_grad) {
            ctx->save_for_backward({u, Mf, sw, p[0], p[1], p[2], p[3]});
            ctx->saved_data["states"] = states;
            ctx->saved_data["sws"] = sws;
            if (slot) {
                ctx->saved_data["ticket"] = ticket;
                ctx->saved_data["slot"] = (int64_t) reinterpret_cast<intptr_t>(slot);
            }
            ctx->saved_data["desc"] = std::string(reinterpret_cast<const char*>(&d), sizeof(d));
            ctx->saved_data["cfg"] = std::vector<int64_t>{sps, mode, ckpt_mode, ckpt_lo, (int64_t)M.scalar_type(),
                                                          skip.defined() ? (int64_t)skip.scalar_type() : -1};
            ctx->saved_data["amax"] = amax;
            ctx->saved_data["sh0"] = ab.sizes().vec();
            ctx->saved_data["sh1"] = bb.sizes().vec();
            ctx->saved_data["sh2"] = asl.sizes().vec();
            ctx->saved_data["sh3"] = bsl.sizes().vec();
            if (skip.defined()) ctx->saved_data["shs"] = skip.sizes().vec();
        }
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto saved = ctx->get_saved_variables();
        const Tensor &u = saved[0], &Mf = saved[1], &sw = saved[2];
        PdeAdiDesc d;
        std::memcpy(&d, ctx->saved_data["desc"].toStringRef().data(), sizeof(d));
        auto cfg = ctx->saved_data["cfg"].toIntVector();
        const int sps = (int)cfg[0], mode = (int)cfg[1], K = d.num_sweeps / sps;
        uint64_t mask[2] = {(uint64_t)cfg[3], 0};
        if (cfg[2] == 1) {
            Slot* slot = reinterpret_cast<Slot*>((intptr_t)ctx->saved_data["slot"].toInt());
            wait_event(slot->ev);
            plan_steps(slot->host, K, sps, ctx->saved_data["amax"].toDouble(), mask);
        }
        c10::hip::HIPGuardMasqueradingAsCUDA guard(u.device());
        hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(u.device().index()).stream();
        Tensor gy = grads[0];
        if (gy.scalar_type() != u.scalar_type()) gy = gy.to(u.scalar_type());
        gy = gy.contiguous();
        Tensor gu = at::empty_like(gy);
        static const char* const kShape[4] = {"sh0", "sh1", "sh2", "sh3"};
        Tensor gp[4];
        for (int i = 0; i < 4; ++i) gp[i] = at::empty(ctx->saved_data[kShape[i]].toIntVector(), saved[3 + i].options());
        Tensor gM = at::empty_like(Mf);
        Tensor gsw = sw.defined() ? at::empty({1}, Mf.options()) : Tensor();
        Tensor ws = bytes(pde_adi_small_backward_workspace_bytes(&d, sps, popcount2(mask)), u);
        Tensor states = ctx->saved_data["states"].toTensor(), sws = ctx->saved_data["sws"].toTensor();
        check(pde_adi_small_backward(&d, sps, mode, gy.data_ptr(), u.data_ptr(), states.data_ptr(), Mf.data_ptr<float>(),
                                     sw.defined() ? sw.data_ptr<float>() : nullptr, mask, gu.data_ptr(),
                                     saved[3].data_ptr<float>(), saved[4].data_ptr<float>(), saved[5].data_ptr<float>(),
                                     saved[6].data_ptr<float>(), gp[0].data_ptr<float>(), gp[1].data_ptr<float>(),
                                     gp[2].data_ptr<float>(), gp[3].data_ptr<float>(), gM.data_ptr<float>(),
                                     gsw.defined() ? gsw.data_ptr<float>() : nullptr, sws.data_ptr(), ws.data_ptr(),
                                     (size_t)ws.numel(), (void*)st),
              "pde_adi_small_backward");
        if ((int64_t)gM.scalar_type() != cfg[4]) gM = gM.to((at::ScalarType)cfg[4]);
        if (gsw.defined()) {
            if ((int64_t)gsw.scalar_type() != cfg[5]) gsw = gsw.to((at::ScalarType)cfg[5]);
            gsw = gsw.reshape(ctx->saved_data["shs"].toIntVector());
        }
        return {gu, gp[0], gp[1], gp[2], gp[3], gM, gsw, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor small(const Tensor& u, const Tensor& ab, const Tensor& bb, const Tensor& asl, const Tensor& bsl, const Tensor& M,
             const std::optional<Tensor>& skip, int64_t desc_addr, int64_t sps, int64_t mode, int64_t ckpt_mode, int64_t ckpt_lo,
             double amax) {
    const Tensor sk = (skip.has_value() && skip->defined()) ? *skip : Tensor();
    const std::optional<Tensor> sko = sk.defined() ? std::optional<Tensor>(sk) : std::nullopt;
    const bool need_grad = at::GradMode::is_enabled() &&
                           (u.requires_grad() || ab.requires_grad() || bb.requires_grad() || asl.requires_grad() ||
                            bsl.requires_grad() || M.requires_grad() || (sk.defined() && sk.requires_grad()));
    return SmallFn::apply(u, ab, bb, asl, bsl, M, sko, desc_addr, sps, mode, ckpt_mode, ckpt_lo, amax, need_grad);
}

// ---- one layer with a channel operator between its steps at any width: one factorisation, then per step one mixing and
// one sweep launch (or the whole forward in one launch at C = 32 / 64), looped inside the library
// (functional._AdiMixedFn: cifar10.py:84-112 "pre", SVHN.py:55-72 "post") -------------------------------------------------
struct MixedFn : public torch::autograd::Function<MixedFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& u_in, const Tensor& ab, const Tensor& bb, const Tensor& asl,
                          const Tensor& bsl, const Tensor& M, int64_t desc_addr, int64_t sps, int64_t mode, int64_t ckpt_mode,
                          int64_t ckpt_lo, double amax, bool need_grad) {
        PDE_REQUIRE(u_in.is_cuda() && ab.is_cuda() && bb.is_cuda() && asl.is_cuda() && bsl.is_cuda() && M.is_cuda(),
                    "libpdecnn_hip operators need CUDA/HIP tensors (there is no CPU fallback)");
        PDE_REQUIRE(u_in.dim() == 4 && u_in.size(2) == u_in.size(3), "expected (B,C,N,N), got ", u_in.sizes());
        PdeAdiDesc d;
        std::memcpy(&d, reinterpret_cast<const void*>(desc_addr), sizeof(d));
        const int64_t B = u_in.size(0), C = u_in.size(1), N = u_in.size(2);
        Tensor u = u_in.detach();
        if (u.scalar_type() != at::kFloat && u.scalar_type() != at::kBFloat16) u = u.to(at::kFloat);
        u = u.contiguous();
        PDE_REQUIRE(d.B == B && d.C == C && d.N == N && d.io_dtype == (u.scalar_type() == at::kBFloat16 ? PDE_IO_BF16 : PDE_IO_F32),
                    "descriptor does not match the tensor");
        const int64_t K = d.num_sweeps / sps;
        Tensor p[4] = {as_chw(ab, C, N), as_chw(bb, C, N), as_chw(asl, C, N), as_chw(bsl, C, N)};
        Tensor Mf = as_f32(M);
        const bool want_kmax = need_grad && ckpt_mode == 1;
        c10::hip::HIPGuardMasqueradingAsCUDA guard(u.device());
        hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(u.device().index()).stream();
        Tensor sws = bytes(pde_adi_steps_workspace_bytes(&d, (int32_t)sps), u);
        Tensor kdev, ticket;
        Slot* slot = nullptr;
        if (want_kmax) {
            kdev = at::empty({(int64_t)d.num_sweeps}, u.options().dtype(at::kFloat));
            slot = acquire_slot(u.device().index());
            ticket = slot_ticket(slot, d.num_sweeps);
        }
        // states[2k]: output of step k's first operator, states[2k+1]: of its second (= input of step k + 1); the last of
        // them is the layer output and lives in a tensor of its own
        Tensor states = at::empty(states_shape(2 * K - 1, u), u.options());
        Tensor y = at::empty_like(u);
        check(pde_adi_mixed_forward(&d, (int32_t)sps, (int32_t)mode, u.data_ptr(), states.data_ptr(), y.data_ptr(), Mf.data_ptr<float>(),
                                    p[0].data_ptr<float>(), p[1].data_ptr<float>(), p[2].data_ptr<float>(), p[3].data_ptr<float>(),
                                    want_kmax ? kdev.data_ptr<float>() : nullptr, slot ? slot->host : nullptr,
                                    slot ? (void*)slot->ev : nullptr, sws.data_ptr(), (size_t)sws.numel(), (void*)st),
              "pde_adi_mixed_forward");
        if (need_grad) {
            ctx->save_for_backward({u, states, Mf, p[0], p[1], p[2], p[3], y});
            ctx->saved_data["sws"] = sws;
            if (slot) {
                ctx->saved_data["ticket"] = ticket;
                ctx->saved_data["slot"] = (int64_t) reinterpret_cast<intptr_t>(slot);
            }
            ctx->saved_data["desc"] = std::string(reinterpret_cast<const char*>(&d), sizeof(d));
            ctx->saved_data["cfg"] = std::vector<int64_t>{sps, mode, ckpt_mode, ckpt_lo, (int64_t)M.scalar_type()};
            ctx->saved_data["amax"] = amax;
            ctx->saved_data["sh0"] = ab.sizes().vec();
            ctx->saved_data["sh1"] = bb.sizes().vec();
            ctx->saved_data["sh2"] = asl.sizes().vec();
            ctx->saved_data["sh3"] = bsl.sizes().vec();
        }
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto saved = ctx->get_saved_variables();
        const Tensor &u = saved[0], &states = saved[1], &Mf = saved[2];
        PdeAdiDesc d;
        std::memcpy(&d, ctx->saved_data["desc"].toStringRef().data(), sizeof(d));
        auto cfg = ctx->saved_data["cfg"].toIntVector();
        const int sps = (int)cfg[0], mode = (int)cfg[1], K = d.num_sweeps / sps;
        uint64_t mask[2] = {(uint64_t)cfg[3], 0};
        if (cfg[2] == 1) {
            Slot* slot = reinterpret_cast<Slot*>((intptr_t)ctx->saved_data["slot"].toInt());
            wait_event(slot->ev);
            plan_steps(slot->host, K, sps, ctx->saved_data["amax"].toDouble(), mask);
        }
        c10::hip::HIPGuardMasqueradingAsCUDA guard(u.device());
        hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(u.device().index()).stream();
        Tensor gy = grads[0];
        if (gy.scalar_type() != u.scalar_type()) gy = gy.to(u.scalar_type());
        gy = gy.contiguous();
        Tensor gu = at::empty_like(gy);
        static const char* const kShape[4] = {"sh0", "sh1", "sh2", "sh3"};
        Tensor gp[4];
        for (int i = 0; i < 4; ++i) gp[i] = at::empty(ctx->saved_data[kShape[i]].toIntVector(), saved[3 + i].options());
        Tensor gM = at::empty_like(Mf);
        Tensor ws = bytes(pde_adi_mixed_backward_workspace_bytes(&d, sps, popcount2(mask)), u);
        Tensor sws = ctx->saved_data["sws"].toTensor();
        check(pde_adi_mixed_backward(&d, sps, mode, gy.data_ptr(), u.data_ptr(), states.data_ptr(), saved[7].data_ptr(),
                                     Mf.data_ptr<float>(), mask,
                                     gu.data_ptr(), saved[3].data_ptr<float>(), saved[4].data_ptr<float>(),
                                     saved[5].data_ptr<float>(), saved[6].data_ptr<float>(), gp[0].data_ptr<float>(),
                                     gp[1].data_ptr<float>(), gp[2].data_ptr<float>(), gp[3].data_ptr<float>(),
                                     gM.data_ptr<float>(), sws.data_ptr(), ws.data_ptr(), (size_t)ws.numel(), (void*)st),
              "pde_adi_mixed_backward");
        if ((int64_t)gM.scalar_type() != cfg[4]) gM = gM.to((at::ScalarType)cfg[4]);
        return {gu, gp[0], gp[1], gp[2], gp[3], gM, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor mixed(const Tensor& u, const Tensor& ab, const Tensor& bb, const Tensor& asl, const Tensor& bsl, const Tensor& M,
             int64_t desc_addr, int64_t sps, int64_t mode, int64_t ckpt_mode, int64_t ckpt_lo, double amax) {
    const bool need_grad = at::GradMode::is_enabled() && (u.requires_grad() || ab.requires_grad() || bb.requires_grad() ||
                                                          asl.requires_grad() || bsl.requires_grad() || M.requires_grad());
    return MixedFn::apply(u, ab, bb, asl, bsl, M, desc_addr, sps, mode, ckpt_mode, ckpt_lo, amax, need_grad);
}

// ---- several mixing-first layers on the SAME input, one launch per pass (functional._AdiMultiFn: cifar10.py:272-280,
// cifar_2version.py:287-288).  Returns (sum_i w_i y_i, y_1 .. y_L[, s_1 .. s_L]) ----------------------------------------
struct MultiFn : public torch::autograd::Function<MultiFn> {
    static variable_list forward(AutogradContext* ctx, const Tensor& u_in, const std::optional<Tensor>& w_opt, at::TensorList flat,
                                 std::vector<int64_t> desc_addrs, int64_t sps, bool want_sums, int64_t ckpt_mode,
                                 std::vector<int64_t> masks, double amax, bool need_grad) {
        const int nl = (int)desc_addrs.size();
        const Tensor weights = w_opt.has_value() ? *w_opt : Tensor();
        PDE_REQUIRE(nl >= 1 && (int)flat.size() == 5 * nl, "adi_diffuse_multi: five tensors per layer");
        PDE_REQUIRE(u_in.is_cuda() && (!weights.defined() || weights.is_cuda()),
                    "libpdecnn_hip operators need CUDA/HIP tensors (there is no CPU fallback)");
        for (const auto& t : flat) PDE_REQUIRE(t.is_cuda(), "libpdecnn_hip operators need CUDA/HIP tensors (there is no CPU fallback)");
        const int64_t B = u_in.size(0), C = u_in.size(1), N = u_in.size(2);
        Tensor u = u_in.detach();
        if (u.scalar_type() != at::kFloat && u.scalar_type() != at::kBFloat16) u = u.to(at::kFloat);
        u = u.contiguous();
        const bool want_kmax = need_grad && ckpt_mode == 1;
        ctx->set_materialize_grads(false);
        Tensor wdev = weights.defined() ? as_f32(weights) : Tensor();
        c10::hip::HIPGuardMasqueradingAsCUDA guard(u.device());
        hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(u.device().index()).stream();

        std::vector<PdeAdiDesc> descs(nl);
        std::vector<PdeSmallLayer> arr(nl);
        std::vector<Tensor> save{u, wdev}, hold, states_v, sums_v, tickets;
        std::vector<int64_t> slots, meta;
        std::string shapes;                                   // the parameter shapes, for the gradients
        for (int i = 0; i < nl; ++i) {
            PdeAdiDesc& d = descs[i];
            std::memcpy(&d, reinterpret_cast<const void*>(desc_addrs[i]), sizeof(d));
            PDE_REQUIRE(d.B == B && d.C == C && d.N == N && d.io_dtype == (u.scalar_type() == at::kBFloat16 ? PDE_IO_BF16 : PDE_IO_F32),
                        "descriptor does not match the tensor");
            const int64_t K = d.num_sweeps / sps;
            Tensor p[4];
            for (int j = 0; j < 4; ++j) p[j] = as_chw(flat[5 * i + j], C, N);
            Tensor Mf = as_f32(flat[5 * i + 4]);
            Tensor sws = bytes(pde_adi_steps_workspace_bytes(&d, (int32_t)sps), u);
            Tensor states = at::empty(states_shape(K, u), u.options());
            Tensor kdev;
            Slot* slot = nullptr;
            if (want_kmax) {
                kdev = at::empty({(int64_t)d.num_sweeps}, u.options().dtype(at::kFloat));
                slot = acquire_slot(u.device().index());
                tickets.push_back(slot_ticket(slot, d.num_sweeps));
                hold.push_back(kdev);
            }
            slots.push_back((int64_t) reinterpret_cast<intptr_t>(slot));
            Tensor psum = want_sums ? at::empty({B, C}, u.options().dtype(at::kFloat)) : Tensor();
            PdeSmallLayer& a = arr[i];
            std::memset(&a, 0, sizeof(a));
            a.desc = &d;
            a.sweeps_per_step = (int32_t)sps;
            a.mode = 1;
            a.M = Mf.data_ptr<float>();
            a.alpha_base = p[0].data_ptr<float>();
            a.beta_base = p[1].data_ptr<float>();
            a.alpha_slope = p[2].data_ptr<float>();
            a.beta_slope = p[3].data_ptr<float>();
            a.weight = 0.f;
            a.weight_ptr = wdev.defined() ? wdev.data_ptr<float>() + i : nullptr;
            a.states = states.data_ptr();
            a.steps_workspace = sws.data_ptr();
            a.steps_workspace_bytes = (size_t)sws.numel();
            a.kappa_max = want_kmax ? kdev.data_ptr<float>() : nullptr;
            a.kappa_max_host = slot ? slot->host : nullptr;
            a.plane_sums = psum.defined() ? psum.data_ptr<float>() : nullptr;
            for (int j = 0; j < 4; ++j) save.push_back(p[j]);
            save.push_back(Mf);
            hold.push_back(sws);
            states_v.push_back(states);
            sums_v.push_back(psum);
            meta.push_back((int64_t)flat[5 * i + 4].scalar_type());
        }
        Tensor out = at::empty_like(u);
        Slot* last = want_kmax ? reinterpret_cast<Slot*>((intptr_t)slots.back()) : nullptr;   // recorded behind the last layer's copy
        check(pde_adi_multi_forward(nl, arr.data(), u.data_ptr(), out.data_ptr(), last ? (void*)last->ev : nullptr, (void*)st),
              "pde_adi_multi_forward");
        if (need_grad) {
            ctx->save_for_backward(save);
            ctx->saved_data["hold"] = hold;
            ctx->saved_data["states"] = states_v;
            if (want_kmax) ctx->saved_data["tickets"] = tickets;
            ctx->saved_data["slots"] = slots;
            ctx->saved_data["descs"] = std::string(reinterpret_cast<const char*>(descs.data()), sizeof(PdeAdiDesc) * nl);
            ctx->saved_data["cfg"] = std::vector<int64_t>{nl, sps, want_sums ? 1 : 0, ckpt_mode,
                                                          weights.defined() ? 1 : 0};
            ctx->saved_data["masks"] = masks;
            ctx->saved_data["mdt"] = meta;
            ctx->saved_data["amax"] = amax;
            for (int i = 0; i < 4 * nl; ++i)
                ctx->saved_data["sh" + std::to_string(i)] = flat[5 * (i / 4) + (i % 4)].sizes().vec();
        }
        variable_list res{out};
        for (int i = 0; i < nl; ++i) res.push_back(states_v[i].select(0, states_v[i].size(0) - 1));   // the last sweep output of every layer
        if (want_sums)
            for (int i = 0; i < nl; ++i) res.push_back(sums_v[i]);
        return res;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto saved = ctx->get_saved_variables();
        auto cfg = ctx->saved_data["cfg"].toIntVector();
        const int nl = (int)cfg[0], sps = (int)cfg[1];
        const bool want_sums = cfg[2] != 0, has_w = cfg[4] != 0;
        const Tensor &u = saved[0], &wdev = saved[1];
        std::vector<PdeAdiDesc> descs(nl);
        std::memcpy(descs.data(), ctx->saved_data["descs"].toStringRef().data(), sizeof(PdeAdiDesc) * nl);
        auto slots = ctx->saved_data["slots"].toIntVector();
        auto masks = ctx->saved_data["masks"].toIntVector();
        auto mdt = ctx->saved_data["mdt"].toIntVector();
        auto states_v = ctx->saved_data["states"].toTensorVector();
        auto hold = ctx->saved_data["hold"].toTensorVector();
        const double amax = ctx->saved_data["amax"].toDouble();
        if (cfg[3] == 1) wait_event(reinterpret_cast<Slot*>((intptr_t)slots.back())->ev);
        c10::hip::HIPGuardMasqueradingAsCUDA guard(u.device());
        hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(u.device().index()).stream();

        Tensor gout = grads[0];
        if (gout.defined()) {
            if (gout.scalar_type() != u.scalar_type()) gout = gout.to(u.scalar_type());
            gout = gout.contiguous();
        }
        std::vector<PdeSmallLayer> arr(nl);
        std::vector<uint64_t> mk(2 * nl);
        std::vector<Tensor> keep;
        variable_list res(2 + 5 * nl + 7);                    // u, weights, five per layer, the seven non-tensor arguments
        std::vector<Tensor> gws;
        bool any_in = gout.defined();
        const int kmax_stride = cfg[3] == 1 ? 2 : 1;          // hold: [kdev, sws] per layer with coefficient maxima, else [sws]
        for (int i = 0; i < nl; ++i) {
            PdeAdiDesc& d = descs[i];
            const int K = d.num_sweeps / sps;
            uint64_t* m = &mk[2 * i];
            m[0] = m[1] = 0;
            if (cfg[3] == 1) plan_steps(reinterpret_cast<Slot*>((intptr_t)slots[i])->host, K, sps, amax, m);
            else m[0] = (uint64_t)masks[i];
            const Tensor* p = &saved[2 + 5 * i];
            const Tensor& Mf = saved[2 + 5 * i + 4];
            Tensor ws = bytes(pde_adi_small_backward_workspace_bytes(&d, sps, popcount2(m)), u);
            Tensor gp[4];
            for (int j = 0; j < 4; ++j) gp[j] = at::empty(ctx->saved_data["sh" + std::to_string(4 * i + j)].toIntVector(), p[j].options());
            Tensor gM = at::empty_like(Mf);
            Tensor gw = at::empty({1}, Mf.options());
            Tensor gyi = grads[1 + i], gsi = want_sums ? grads[1 + nl + i] : Tensor();
            if (gyi.defined()) {
                if (gyi.scalar_type() != u.scalar_type()) gyi = gyi.to(u.scalar_type());
                gyi = gyi.contiguous();
            }
            if (gsi.defined()) {
                if (gsi.scalar_type() != at::kFloat) gsi = gsi.to(at::kFloat);
                gsi = gsi.contiguous();
            }
            any_in = any_in || gyi.defined() || gsi.defined();
            const Tensor& sws = hold[kmax_stride * i + (kmax_stride - 1)];
            PdeSmallLayer& a = arr[i];
            std::memset(&a, 0, sizeof(a));
            a.desc = &d;
            a.sweeps_per_step = sps;
            a.mode = 1;
            a.M = Mf.data_ptr<float>();
            a.alpha_base = p[0].data_ptr<float>();
            a.beta_base = p[1].data_ptr<float>();
            a.alpha_slope = p[2].data_ptr<float>();
            a.beta_slope = p[3].data_ptr<float>();
            a.weight = 0.f;
            a.weight_ptr = wdev.defined() ? wdev.data_ptr<float>() + i : nullptr;
            a.states = states_v[i].data_ptr();
            a.steps_workspace = sws.data_ptr();
            a.steps_workspace_bytes = (size_t)sws.numel();
            a.gys = gyi.defined() ? gyi.data_ptr() : nullptr;
            a.g_plane_sums = gsi.defined() ? gsi.data_ptr<float>() : nullptr;
            a.ckpt_mask = m;
            a.g_alpha_base = gp[0].data_ptr<float>();
            a.g_beta_base = gp[1].data_ptr<float>();
            a.g_alpha_slope = gp[2].data_ptr<float>();
            a.g_beta_slope = gp[3].data_ptr<float>();
            a.gM = gM.data_ptr<float>();
            a.g_weight = gw.data_ptr<float>();
            a.workspace = ws.data_ptr();
            a.workspace_bytes = (size_t)ws.numel();
            keep.push_back(ws);
            keep.push_back(gyi);
            keep.push_back(gsi);
            for (int j = 0; j < 4; ++j) res[2 + 5 * i + j] = gp[j];
            res[2 + 5 * i + 4] = gM;                                        // converted to the operator's type behind the launch
            gws.push_back(gw);
        }
        PDE_REQUIRE(any_in, "adi_diffuse_multi: no incoming gradient");
        Tensor gu = at::empty_like(u);
        check(pde_adi_multi_backward(nl, arr.data(), gout.defined() ? gout.data_ptr() : nullptr, u.data_ptr(), gu.data_ptr(),
                                     (void*)st),
              "pde_adi_multi_backward");
        for (int i = 0; i < nl; ++i)
            if ((int64_t)res[2 + 5 * i + 4].scalar_type() != mdt[i]) res[2 + 5 * i + 4] = res[2 + 5 * i + 4].to((at::ScalarType)mdt[i]);
        res[0] = gu;
        if (has_w) res[1] = at::cat(gws);
        return res;
    }
};

variable_list multi(const Tensor& u, const std::optional<Tensor>& weights, std::vector<Tensor> flat,
                    std::vector<int64_t> desc_addrs, int64_t sps, bool want_sums, int64_t ckpt_mode, std::vector<int64_t> masks,
                    double amax) {
    const Tensor w = (weights.has_value() && weights->defined()) ? *weights : Tensor();
    const std::optional<Tensor> wo = w.defined() ? std::optional<Tensor>(w) : std::nullopt;
    bool need_grad = u.requires_grad() || (w.defined() && w.requires_grad());
    for (const auto& t : flat) need_grad = need_grad || t.requires_grad();
    need_grad = need_grad && at::GradMode::is_enabled();
    return MultiFn::apply(u, wo, at::TensorList(flat), desc_addrs, sps, want_sums, ckpt_mode, masks, amax, need_grad);
}


// ---- the Ruthotto-Haber symmetric layer (functional._SymLayerFn: cifar_2version.py:190-258) ----------------------------
// out = base + scale * (act(BatchNorm1d(X K^T)) K).  Scratch for the split strip products is kept per (device, stream,
// width): calls on one stream are ordered, and a captured graph keeps pointing at memory that stays allocated.
Tensor sym_workspace(int64_t B, int64_t D, const Tensor& like, hipStream_t st) {
    const size_t n = pde_sym_layer_workspace_bytes((int32_t)B, (int32_t)D);
    if (n == 0) return Tensor();
    // (never destroyed: device tensors must not be freed from a static destructor, after the allocator may be gone)
    static std::mutex& mu = *new std::mutex();
    static std::map<std::tuple<int, void*, int64_t, size_t>, Tensor>& cache = *new std::map<std::tuple<int, void*, int64_t, size_t>, Tensor>();
    std::lock_guard<std::mutex> g(mu);
    auto key = std::make_tuple((int)like.device().index(), (void*)st, D, n);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    if (cache.size() > 64) cache.clear();
    Tensor ws = at::empty({(int64_t)n}, like.options().dtype(at::kByte));
    cache.emplace(key, ws);
    return ws;
}

struct SymFn : public torch::autograd::Function<SymFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& X, const Tensor& K, const Tensor& gamma, const Tensor& beta,
                          const std::optional<Tensor>& base, const std::optional<Tensor>& run_mean,
                          const std::optional<Tensor>& run_var, bool training, double momentum, double eps, double scale,
                          int64_t act, bool need_grad) {
        PDE_REQUIRE(X.is_cuda() && K.is_cuda() && gamma.is_cuda() && beta.is_cuda() && (!base.has_value() || base->is_cuda()),
                    "libpdecnn_hip operators need CUDA/HIP tensors (there is no CPU fallback)");
        PDE_REQUIRE(X.dim() == 2 && K.dim() == 2 && K.size(0) == X.size(1) && K.size(1) == X.size(1), "expected X (B, D), K (D, D)");
        const int64_t B = X.size(0), D = X.size(1);
        Tensor Xf = as_f32(X), Kf = as_f32(K), gm = as_f32(gamma), bt = as_f32(beta);
        Tensor bs = base.has_value() ? as_f32(*base) : Tensor();
        c10::hip::HIPGuardMasqueradingAsCUDA guard(X.device());
        hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(X.device().index()).stream();
        Tensor P = at::empty({B, D}, Xf.options()), H = at::empty({B, D}, Xf.options()), out = at::empty({B, D}, Xf.options());
        Tensor mean = at::empty({D}, Xf.options()), invstd = at::empty({D}, Xf.options());
        Tensor ws = sym_workspace(B, D, Xf, st);
        float* rm = run_mean.has_value() ? run_mean->data_ptr<float>() : nullptr;
        float* rv = run_var.has_value() ? run_var->data_ptr<float>() : nullptr;
        check(pde_sym_layer_forward((int32_t)B, (int32_t)D, (int32_t)act, training ? 1 : 0, Xf.data_ptr<float>(),
                                    Kf.data_ptr<float>(), gm.data_ptr<float>(), bt.data_ptr<float>(), rm, rv, (float)momentum,
                                    (float)eps, bs.defined() ? bs.data_ptr<float>() : nullptr, (float)scale, P.data_ptr<float>(),
                                    H.data_ptr<float>(), mean.data_ptr<float>(), invstd.data_ptr<float>(), out.data_ptr<float>(),
                                    ws.defined() ? ws.data_ptr() : nullptr, ws.defined() ? (size_t)ws.numel() : 0, (void*)st),
              "pde_sym_layer_forward");
        if (need_grad) {
            ctx->save_for_backward({Xf, Kf, gm, P, H, mean, invstd});
            ctx->saved_data["cfg"] = std::vector<int64_t>{training ? 1 : 0, act, base.has_value() ? 1 : 0};
            ctx->saved_data["scale"] = scale;
        }
        return out;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto sv = ctx->get_saved_variables();
        const Tensor &Xf = sv[0], &Kf = sv[1], &gm = sv[2], &P = sv[3], &H = sv[4], &mean = sv[5], &invstd = sv[6];
        auto cfg = ctx->saved_data["cfg"].toIntVector();
        const double scale = ctx->saved_data["scale"].toDouble();
        const int64_t B = Xf.size(0), D = Xf.size(1);
        c10::hip::HIPGuardMasqueradingAsCUDA guard(Xf.device());
        hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(Xf.device().index()).stream();
        Tensor g = grads[0];
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        g = g.contiguous();
        Tensor dP = at::empty_like(Xf), gX = at::empty_like(Xf), gK = at::empty_like(Kf);
        Tensor gg = at::empty({D}, Xf.options()), gb = at::empty({D}, Xf.options());
        Tensor ws = sym_workspace(B, D, Xf, st);
        check(pde_sym_layer_backward((int32_t)B, (int32_t)D, (int32_t)cfg[1], (int32_t)cfg[0], g.data_ptr<float>(), (float)scale,
                                     Xf.data_ptr<float>(), Kf.data_ptr<float>(), gm.data_ptr<float>(), P.data_ptr<float>(),
                                     H.data_ptr<float>(), mean.data_ptr<float>(), invstd.data_ptr<float>(), dP.data_ptr<float>(),
                                     gX.data_ptr<float>(), gK.data_ptr<float>(), gg.data_ptr<float>(), gb.data_ptr<float>(),
                                     ws.defined() ? ws.data_ptr() : nullptr, ws.defined() ? (size_t)ws.numel() : 0, (void*)st),
              "pde_sym_layer_backward");
        return {gX, gK, gg, gb, cfg[2] ? g : Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor sym(const Tensor& X, const Tensor& K, const Tensor& gamma, const Tensor& beta, const std::optional<Tensor>& base,
           const std::optional<Tensor>& run_mean, const std::optional<Tensor>& run_var, bool training, double momentum, double eps,
           double scale, int64_t act) {
    const bool need_grad = at::GradMode::is_enabled() && (X.requires_grad() || K.requires_grad() || gamma.requires_grad() ||
                                                          beta.requires_grad() || (base.has_value() && base->requires_grad()));
    auto opt = [](const std::optional<Tensor>& t) { return (t.has_value() && t->defined()) ? t : std::optional<Tensor>(); };
    return SymFn::apply(X, K, gamma, beta, opt(base), opt(run_mean), opt(run_var), training, momentum, eps, scale, act, need_grad);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "native host path of the launch-bound PDE-layer calls (torch C++ autograd over the C ABI of pdecnn.h)";
    m.def("adi", &adi, "functional.adi_diffuse without the interpreter: one implicit diffusion layer call",
          py::arg("u"), py::arg("alpha_base"), py::arg("beta_base"), py::arg("alpha_time_coeff"), py::arg("beta_time_coeff"),
          py::arg("desc_addr"), py::arg("ckpt_mode"), py::arg("ckpt_lo"), py::arg("ckpt_hi"), py::arg("amax"));
    py::class_<Ticket, std::shared_ptr<Ticket>>(m, "Ticket", "the per-sweep coefficient maxima of one call (pinned slot + event)")
        .def("query", &Ticket::query)
        .def("synchronize", &Ticket::synchronize)
        .def("wait", [](std::shared_ptr<Ticket> t) {
            t->synchronize();
            std::vector<float> v(t->slot->host, t->slot->host + t->n);
            return v;
        })
        .def_property_readonly("host", &Ticket::host)
        .def_property_readonly("event", [](std::shared_ptr<Ticket> t) { return t; });
    m.def("adi_lagged", &adi_lagged, "adi with an explicit checkpoint mask that also delivers this call's coefficient maxima",
          py::arg("u"), py::arg("alpha_base"), py::arg("beta_base"), py::arg("alpha_time_coeff"), py::arg("beta_time_coeff"),
          py::arg("desc_addr"), py::arg("ckpt_lo"), py::arg("ckpt_hi"));
    m.def("small", &small, "functional.adi_diffuse_small: one layer with a channel operator, C <= 4, one launch per pass",
          py::arg("u"), py::arg("alpha_base"), py::arg("beta_base"), py::arg("alpha_time_coeff"), py::arg("beta_time_coeff"),
          py::arg("M"), py::arg("skip_weight"), py::arg("desc_addr"), py::arg("sweeps_per_step"), py::arg("mode"),
          py::arg("ckpt_mode"), py::arg("ckpt_lo"), py::arg("amax"));
    m.def("mixed", &mixed, "functional.adi_diffuse_mixed: one layer with a channel operator between its steps, any width",
          py::arg("u"), py::arg("alpha_base"), py::arg("beta_base"), py::arg("alpha_time_coeff"), py::arg("beta_time_coeff"),
          py::arg("M"), py::arg("desc_addr"), py::arg("sweeps_per_step"), py::arg("mode"), py::arg("ckpt_mode"),
          py::arg("ckpt_lo"), py::arg("amax"));
    m.def("multi", &multi, "functional.adi_diffuse_multi: layers that share an input, one launch per pass",
          py::arg("u"), py::arg("weights"), py::arg("flat"), py::arg("desc_addrs"), py::arg("sweeps_per_step"),
          py::arg("want_sums"), py::arg("ckpt_mode"), py::arg("masks"), py::arg("amax"));
    m.def("sym", &sym, "functional.sym_layer's autograd node: out = base + scale * (act(BatchNorm1d(X K^T)) K)",
          py::arg("X"), py::arg("K"), py::arg("gamma"), py::arg("beta"), py::arg("base"), py::arg("running_mean"),
          py::arg("running_var"), py::arg("training"), py::arg("momentum"), py::arg("eps"), py::arg("scale"), py::arg("act"));
    m.def("set_error_class", [](py::object cls) {
        Py_XDECREF(g_error_class);
        g_error_class = cls.release().ptr();
    }, "the Python exception class failed calls of the C ABI raise (_lib.PdeError)");
    py::register_exception_translator([](std::exception_ptr p) {
        try {
            if (p) std::rethrow_exception(p);
        } catch (const PdeFailure& e) {
            PyErr_SetString(g_error_class ? g_error_class : PyExc_RuntimeError, e.what());
        }
    });
    m.def("abi_version", []() { return std::string(pde_version()); });
}
