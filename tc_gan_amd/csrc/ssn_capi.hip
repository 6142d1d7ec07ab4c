// extern "C" surface of libssnode.so (see include/ssnode_mi355x.h).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <condition_variable>
#include <mutex>
#include <utility>
#include <string>
#include <vector>
#include "ssn_host.h"

namespace {

thread_local std::string g_last_error;

int fail(hipError_t e, const char* where) {
    g_last_error = std::string(where) + ": " + hipGetErrorString(e);
    return SSN_ERR_BASE + (int)e;
}
#define SSN_TRY(expr)                                     \
    do {                                                  \
        hipError_t e__ = (expr);                          \
        if (e__ != hipSuccess) return fail(e__, #expr);   \
    } while (0)

// Operand precision of the AUTOMATIC kernel choice (kernel = 0, variant < 0): process-wide state behind
// ssn_set_operand_precision / ssn_get_operand_precision (include/ssnode_mi355x.h).  1 = the fp16-split matrix-core kernels
// where they apply (W and state as two fp16 parts each), 0 = fp32 operands only.  The environment variable SSN_FWD_SPLIT=0
// only sets the INITIAL value; callers change it at run time through the API and explicit kernel codes ignore it.
static std::atomic<int>& operand_precision_state() {
    static std::atomic<int> st{[] { const char* e = getenv("SSN_FWD_SPLIT"); return (e && e[0] == '0') ? 0 : 1; }()};
    return st;
}
static bool forward_split_default() { return operand_precision_state().load(std::memory_order_relaxed) != 0; }
template <typename T>
int solve_batch_impl(int variant, const T* W, const T* ext, int ext_per_draw, T* r, T* r_prev, int* codes,
                     int* steps, int B, int NB, int M, const ssn_solver_params* p, void* stream, bool dry_run = false) {
    // dry_run: no pointers, nothing is launched; returns the variant the call would run (ssn_solve_batch_variant_for)
    if ((B == 0 || NB == 0) && !dry_run) return 0;   // empty batch: nothing to do (pointers may be null)
    if (!p || (!dry_run && (!W || !ext || !r || !codes)) || B < 0 || NB < 0 || M <= 0 || (M & 1) || p->io_type < 0 ||
        p->io_type > 2 || p->max_iter < 0) {
        g_last_error = "ssn_solve_batch: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    ssn::SolveArgs<T> a;
    a.W = W; a.ext = ext; a.r = r; a.r_prev = r_prev; a.codes = codes; a.steps = steps;
    a.ext_per_draw = ext_per_draw; a.B = B; a.NB = NB; a.M = M; a.N = M / 2;
    a.io = ssn::make_io_consts<T>(*p);
    a.st = ssn::make_step_consts<T>(*p);
    hipStream_t st = (hipStream_t)stream;
    // variant: -1 auto (MFMA for large fp32 NB >= 4 batches, else tile > regw > stream), 0 streaming, 1 register-stationary DPP, 2 tile (shape chosen by
    // the library), 3 tile with split VGPR/LDS residency where instantiated, 4 tile with the whole W tile in VGPRs,
    // 5 fp32 MFMA kernel (NB >= 4), 6 fp16-split MFMA kernel (NB >= 4, asym_tanh; wide form), 7 the same in the alternating form,
    // 8 fp16-split MFMA kernel with two draws per workgroup (ssn_duo.hip)
    const bool tile_ok = ssn::tile_supported<T>(M, NB), regw_ok = ssn::regw_supported<T>(M, NB);
    bool mfma_ok = false;
    if constexpr (sizeof(T) == 4) mfma_ok = ssn::gen_mfma_supported(M, NB);
    bool split_ok = false;
    if constexpr (sizeof(T) == 4) split_ok = mfma_ok && ssn::solve_split_supported(a);
    if ((variant == 1 && !regw_ok) || (variant >= 2 && variant <= 4 && !tile_ok) || (variant == 5 && !mfma_ok) ||
        ((variant == 6 || variant == 7 || variant == 8) && !split_ok) || variant > 8) {
        g_last_error = "ssn_solve_batch: requested kernel variant has no instantiation for this size";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    // auto: the MFMA solver (variant 5) for fp32 with NB >= 4 stimuli per draw, 2N above the smallest ladder size and
    // enough (draw, 8 stimuli) workgroups to fill the chip -- 71 ms at the C2/NB=8 shape against 85-90 ms for the
    // split tile kernel (which runs one workgroup per (draw, stimulus) and so keeps small batches busier)
    if (variant < 0) {
        const bool big = (long)B * ((NB + 7) / 8) >= 192 && M > 104;
        // fp16-split forms: two draws per workgroup (8) once that still gives every CU a workgroup, else one (6)
        const int split_variant = (long)B * ((NB + 7) / 8) > 256 ? 8 : 6;
        variant = (mfma_ok && big) ? ((split_ok && forward_split_default()) ? split_variant : 5) : (tile_ok ? 2 : (regw_ok ? 1 : 0));
    }
    if (dry_run) return variant;
    switch (variant) {
        case 2: SSN_TRY(ssn::launch_tile<T>(a, st, 0)); break;
        case 3: SSN_TRY(ssn::launch_tile<T>(a, st, 1)); break;
        case 4: SSN_TRY(ssn::launch_tile<T>(a, st, 2)); break;
        case 5: if constexpr (sizeof(T) == 4) { SSN_TRY(ssn::launch_solve_mfma(a, st)); } break;
        case 6: if constexpr (sizeof(T) == 4) { SSN_TRY(ssn::launch_solve_split(a, st)); } break;
        case 7: if constexpr (sizeof(T) == 4) { a.split_narrow = 1; SSN_TRY(ssn::launch_solve_split(a, st)); } break;
        case 8: if constexpr (sizeof(T) == 4) { SSN_TRY(ssn::launch_solve_duo(a, st)); } break;
        case 1: SSN_TRY(ssn::launch_regw<T>(a, st)); break;
        default: SSN_TRY(ssn::launch_stream<T>(a, st)); break;
    }
    return 0;
}

// ---- host-buffer entry points: per-call arenas --------------------------------------------------
// The reference calls its solver from up to cpu_count() Python threads at once (ssnode.py:436-459), one
// small solve per call.  Each call borrows an Arena -- a non-blocking stream, a device buffer and a pinned
// host staging buffer, all grown on demand and kept -- from a process-wide pool and gives it back on
// return: no hipMalloc / hipFree / hipDeviceSynchronize per call, calls of different threads overlap on the
// device (one stream each), and no state is shared between concurrent calls.  Arenas are never freed (the
// pool outlives every caller; tearing HIP objects down from thread or process exit hooks is not safe).
struct Arena {
    int device = -1;
    hipStream_t stream = nullptr;
    char* dev = nullptr;   size_t dev_bytes = 0;
    char* host = nullptr;  size_t host_bytes = 0;

    hipError_t reserve(size_t need_dev, size_t need_host) {
        if (!stream) {
            hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
            if (e != hipSuccess) return e;
        }
        if (need_dev > dev_bytes) {
            if (dev) { hipStreamSynchronize(stream); hipFree(dev); dev = nullptr; dev_bytes = 0; }
            const size_t n = need_dev < (1u << 20) ? (1u << 20) : need_dev + need_dev / 4;
            hipError_t e = hipMalloc((void**)&dev, n);
            if (e != hipSuccess) return e;
            dev_bytes = n;
        }
        if (need_host > host_bytes) {
            if (host) { hipStreamSynchronize(stream); hipHostFree(host); host = nullptr; host_bytes = 0; }
            const size_t n = need_host < (1u << 20) ? (1u << 20) : need_host + need_host / 4;
            hipError_t e = hipHostMalloc((void**)&host, n, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            host_bytes = n;
        }
        return hipSuccess;
    }
};

class ArenaPool {
    std::mutex mu_;
    std::vector<Arena*> idle_;
public:
    Arena* acquire(int device) {
        {
            std::lock_guard<std::mutex> g(mu_);
            for (size_t i = idle_.size(); i-- > 0;)
                if (idle_[i]->device == device) {
                    Arena* a = idle_[i];
                    idle_.erase(idle_.begin() + (long)i);
                    return a;
                }
        }
        Arena* a = new Arena();
        a->device = device;
        return a;
    }
    void release(Arena* a) {
        std::lock_guard<std::mutex> g(mu_);
        idle_.push_back(a);
    }
};
ArenaPool& arena_pool() {
    static ArenaPool* pool = new ArenaPool();      // leaked on purpose, see above
    return *pool;
}
struct ArenaLease {
    Arena* a = nullptr;
    hipError_t open() {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        a = arena_pool().acquire(dev);
        return hipSuccess;
    }
    // An error return between an asynchronous copy and its wait must not hand staging memory with DMA in flight to the
    // next borrower: drain the arena's stream before it goes back to the pool (a no-op on the normal path, which has
    // already waited for its download).
    ~ArenaLease() {
        if (!a) return;
        if (a->stream) (void)hipStreamSynchronize(a->stream);
        arena_pool().release(a);
    }
};
inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// Host buffers below this size go through the pinned staging buffer (one memcpy, then truly asynchronous
// copies); larger ones are handed to hipMemcpyAsync as they are (the runtime stages pageable memory itself).
constexpr size_t kStageLimit = 8u << 20;

template <typename T>
int solve_batch_host_impl(const T* W, const T* ext, int ext_per_draw, T* r, T* r_prev, int* codes, int* steps,
                          int B, int NB, int M, const ssn_solver_params* p) {
    if (B == 0 || NB == 0) return 0;
    if (!p || !W || !ext || !r || !codes || B < 0 || NB < 0 || M <= 0 || (M & 1)) {
        g_last_error = "ssn_solve_batch_host: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    const size_t nW = (size_t)B * M * M * sizeof(T);
    const size_t nE = (size_t)(ext_per_draw ? B : 1) * NB * M * sizeof(T);
    const size_t nR = (size_t)B * NB * M * sizeof(T);
    const size_t nC = (size_t)B * NB * sizeof(int);
    // device image: [W | ext | r | r_prev | codes | steps]; the host staging image has the same offsets
    const size_t oW = 0, oE = oW + align256(nW), oR = oE + align256(nE), oP = oR + align256(nR),
                 oC = oP + align256(nR), oS = oC + align256(nC), total = oS + align256(nC);
    const bool staged = total <= kStageLimit;
    ArenaLease lease;
    SSN_TRY(lease.open());
    Arena& A = *lease.a;
    SSN_TRY(A.reserve(total, staged ? total : 0));
    hipStream_t st = A.stream;
    T *dW = (T*)(A.dev + oW), *dE = (T*)(A.dev + oE), *dR = (T*)(A.dev + oR), *dP = (T*)(A.dev + oP);
    int *dC = (int*)(A.dev + oC), *dS = (int*)(A.dev + oS);
    if (staged) {
        std::memcpy(A.host + oW, W, nW);
        std::memcpy(A.host + oE, ext, nE);
        std::memcpy(A.host + oR, r, nR);
        // one upload for the three inputs (they are contiguous in both images up to the end of r)
        SSN_TRY(hipMemcpyAsync(A.dev, A.host, oR + nR, hipMemcpyHostToDevice, st));
    } else {
        SSN_TRY(hipMemcpyAsync(dW, W, nW, hipMemcpyHostToDevice, st));
        SSN_TRY(hipMemcpyAsync(dE, ext, nE, hipMemcpyHostToDevice, st));
        SSN_TRY(hipMemcpyAsync(dR, r, nR, hipMemcpyHostToDevice, st));
    }
    int rc = solve_batch_impl<T>(-1, dW, dE, ext_per_draw, dR, dP, dC, dS, B, NB, M, p, st);
    if (rc) return rc;
    if (staged) {
        // one download: [r | r_prev | codes | steps]
        SSN_TRY(hipMemcpyAsync(A.host + oR, A.dev + oR, total - oR, hipMemcpyDeviceToHost, st));
        SSN_TRY(hipStreamSynchronize(st));
        std::memcpy(r, A.host + oR, nR);
        if (r_prev) std::memcpy(r_prev, A.host + oP, nR);
        std::memcpy(codes, A.host + oC, nC);
        if (steps) std::memcpy(steps, A.host + oS, nC);
    } else {
        SSN_TRY(hipMemcpyAsync(r, dR, nR, hipMemcpyDeviceToHost, st));
        if (r_prev) SSN_TRY(hipMemcpyAsync(r_prev, dP, nR, hipMemcpyDeviceToHost, st));
        SSN_TRY(hipMemcpyAsync(codes, dC, nC, hipMemcpyDeviceToHost, st));
        if (steps) SSN_TRY(hipMemcpyAsync(steps, dS, nC, hipMemcpyDeviceToHost, st));
        SSN_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

// ---- the reference's single-solve entry point ----------------------------------------------------
// One call = one (W, ext) pair = one workgroup of the fp64 kernel: latency-bound (~1 us per Euler step), and the
// runtime multiplexes streams onto a handful of hardware queues, so 16 caller threads with a stream each still run
// only ~4 solves at a time.  Calls that are in flight together are therefore COMBINED: a caller enqueues its request;
// whoever finds no leader active becomes the leader, takes every queued request with the same size and parameters,
// runs them as ONE batched launch (B = number of callers, one workgroup each) and hands the results back.  While a
// batch runs, the other threads' next calls queue up and form the next batch.  A lone caller is a batch of one.
// Results do not depend on the batching: every (draw, stimulus) pair is solved by its own workgroup with its own stop
// logic (tests/test_solver_gpu.py: batch-independence is bitwise).
struct SolveReq {
    int device, N;
    const double *W, *ext;
    double *r0, *r1;
    ssn_solver_params p;
    int code = -1, steps = 0, rc = 0;
    std::string error;                 // the batch leader's message when rc != 0 (thread-local on ITS thread)
    bool done = false;
    bool same_shape(const SolveReq& o) const {
        return device == o.device && N == o.N && p.io_type == o.p.io_type && p.max_iter == o.p.max_iter && p.k == o.p.k &&
               p.n == o.p.n && p.tau_E == o.p.tau_E && p.tau_I == o.p.tau_I && p.dt == o.p.dt && p.atol == o.p.atol &&
               p.rate_soft_bound == o.p.rate_soft_bound && p.rate_hard_bound == o.p.rate_hard_bound;
    }
};

// Solve a group of same-shaped requests in one launch.  Returns 0 or an SSN_ERR code (applies to the whole group).
int solve_group(const std::vector<SolveReq*>& grp) {
    const int B = (int)grp.size(), M = 2 * grp[0]->N;
    const size_t nW1 = (size_t)M * M * sizeof(double), nV1 = (size_t)M * sizeof(double);
    const size_t oW = 0, oE = oW + align256(B * nW1), oR = oE + align256(B * nV1), oP = oR + align256(B * nV1),
                 oC = oP + align256(B * nV1), oS = oC + align256(B * sizeof(int)), total = oS + align256(B * sizeof(int));
    const bool staged = total <= kStageLimit;
    // the group runs on ITS callers' device; the leader's own current device is put back on return
    struct DeviceScope {
        int prev = -1; bool changed = false;
        explicit DeviceScope(int want) { if (hipGetDevice(&prev) == hipSuccess && prev != want) changed = hipSetDevice(want) == hipSuccess; }
        ~DeviceScope() { if (changed) (void)hipSetDevice(prev); }
    } scope(grp[0]->device);
    ArenaLease lease;
    SSN_TRY(lease.open());
    Arena& A = *lease.a;
    SSN_TRY(A.reserve(total, staged ? total : 0));
    hipStream_t st = A.stream;
    double *dW = (double*)(A.dev + oW), *dE = (double*)(A.dev + oE), *dR = (double*)(A.dev + oR), *dP = (double*)(A.dev + oP);
    int *dC = (int*)(A.dev + oC), *dS = (int*)(A.dev + oS);
    if (staged) {
        for (int b = 0; b < B; ++b) {
            std::memcpy(A.host + oW + b * nW1, grp[b]->W, nW1);
            std::memcpy(A.host + oE + b * nV1, grp[b]->ext, nV1);
            std::memcpy(A.host + oR + b * nV1, grp[b]->r0, nV1);
        }
        SSN_TRY(hipMemcpyAsync(A.dev, A.host, oR + B * nV1, hipMemcpyHostToDevice, st));
    } else {
        for (int b = 0; b < B; ++b) {
            SSN_TRY(hipMemcpyAsync((char*)dW + b * nW1, grp[b]->W, nW1, hipMemcpyHostToDevice, st));
            SSN_TRY(hipMemcpyAsync((char*)dE + b * nV1, grp[b]->ext, nV1, hipMemcpyHostToDevice, st));
            SSN_TRY(hipMemcpyAsync((char*)dR + b * nV1, grp[b]->r0, nV1, hipMemcpyHostToDevice, st));
        }
    }
    int rc = solve_batch_impl<double>(-1, dW, dE, /*ext_per_draw=*/1, dR, dP, dC, dS, B, 1, M, &grp[0]->p, st);
    if (rc) return rc;
    std::vector<int> cs(2 * (size_t)B);
    if (staged) {
        SSN_TRY(hipMemcpyAsync(A.host + oR, A.dev + oR, total - oR, hipMemcpyDeviceToHost, st));
        SSN_TRY(hipStreamSynchronize(st));
        for (int b = 0; b < B; ++b) {
            std::memcpy(grp[b]->r0, A.host + oR + b * nV1, nV1);     // newest state
            std::memcpy(grp[b]->r1, A.host + oP + b * nV1, nV1);     // state one step before
            cs[b] = ((const int*)(A.host + oC))[b];
            cs[B + b] = ((const int*)(A.host + oS))[b];
        }
    } else {
        for (int b = 0; b < B; ++b) {
            SSN_TRY(hipMemcpyAsync(grp[b]->r0, (char*)dR + b * nV1, nV1, hipMemcpyDeviceToHost, st));
            SSN_TRY(hipMemcpyAsync(grp[b]->r1, (char*)dP + b * nV1, nV1, hipMemcpyDeviceToHost, st));
        }
        SSN_TRY(hipMemcpyAsync(cs.data(), dC, B * sizeof(int), hipMemcpyDeviceToHost, st));
        SSN_TRY(hipMemcpyAsync(cs.data() + B, dS, B * sizeof(int), hipMemcpyDeviceToHost, st));
        SSN_TRY(hipStreamSynchronize(st));
    }
    for (int b = 0; b < B; ++b) {
        SolveReq& q = *grp[b];
        q.code = cs[b]; q.steps = cs[B + b];
        // Now r0 = newest, r1 = previous.  Which caller buffer plays "r0" after `swaps` role exchanges
        // (ssnode.c:104-106) decides where the reference would have left them.
        if (q.code == 0) {
            // converged: the newest state is copied into the current "r0" role, so BOTH buffers hold it
            std::memcpy(q.r1, q.r0, nV1);
        } else {
            // code 2: return before the exchange -> swaps = steps-1, newest is in the "r1" role.
            // code 1: all max_iter exchanges done -> newest is in the "r0" role after `steps` swaps.
            const int swaps = (q.code == 2) ? q.steps - 1 : q.steps;
            const bool r0_role_is_caller_r0 = (swaps % 2 == 0);
            const bool newest_in_r0_role = (q.code == 1);
            if (newest_in_r0_role != r0_role_is_caller_r0)
                for (int i = 0; i < M; ++i) std::swap(q.r0[i], q.r1[i]);
        }
    }
    return 0;
}

class SolveCombiner {
    std::mutex mu_;
    std::condition_variable cv_;
    bool leader_active_ = false;
    std::vector<SolveReq*> pending_;
    static constexpr size_t kMaxGroup = 256;
public:
    int submit(SolveReq& q) {
        std::unique_lock<std::mutex> lk(mu_);
        pending_.push_back(&q);
        while (!q.done) {
            if (leader_active_) { cv_.wait(lk); continue; }
            // become the leader for one batch: the oldest request and everything queued that matches it
            leader_active_ = true;
            std::vector<SolveReq*> grp, rest;
            for (SolveReq* r : pending_)
                (grp.size() < kMaxGroup && (grp.empty() || r->same_shape(*grp[0])) ? grp : rest).push_back(r);
            pending_.swap(rest);
            lk.unlock();
            int rc = solve_group(grp);
            const std::string err = g_last_error;      // the leader's thread-local message, for every member
            lk.lock();
            for (SolveReq* r : grp) { r->rc = rc; if (rc) r->error = err; r->done = true; }
            leader_active_ = false;
            cv_.notify_all();
        }
        if (q.rc) g_last_error = q.error;              // every member of a failed batch reports the batch's error
        return q.rc;
    }
};
SolveCombiner& combiner() {
    static SolveCombiner* c = new SolveCombiner();     // leaked on purpose, like the arena pool
    return *c;
}

// Leaves the caller's two buffers as ssnode.c's pointer-swapping loop would (DESIGN.md "buffer parity").
int legacy_solve(int io_type, int N, double* W, double* ext, double k, double n, double* r0, double* r1,
                 double tau_E, double tau_I, double dt, int max_iter, double atol, double soft, double hard) {
    if (N <= 0 || !W || !ext || !r0 || !r1) {
        g_last_error = "solve_dynamics: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    if (max_iter <= 0) return 1;   // ssnode.c: loop body never runs, buffers untouched
    SolveReq q;
    q.N = N; q.W = W; q.ext = ext; q.r0 = r0; q.r1 = r1;
    q.p.io_type = io_type; q.p.max_iter = max_iter; q.p.k = k; q.p.n = n; q.p.tau_E = tau_E;
    q.p.tau_I = tau_I; q.p.dt = dt; q.p.atol = atol; q.p.rate_soft_bound = soft; q.p.rate_hard_bound = hard;
    SSN_TRY(hipGetDevice(&q.device));
    const int rc = combiner().submit(q);
    if (rc) { if (g_last_error.empty()) g_last_error = "solve_dynamics: batched launch failed"; return rc; }
    return q.code;
}

// Scalar helpers of the drop-in surface: `count` doubles in, `count_out` doubles out, one small kernel
// between them, on a borrowed arena (no allocation per call).
template <typename Launch>
bool scalar_roundtrip(const double* in, size_t n_in, double* out, size_t n_out, Launch&& launch) {
    ArenaLease lease;
    if (lease.open() != hipSuccess) return false;
    Arena& A = *lease.a;
    const size_t bytes = (n_in + n_out) * sizeof(double);
    if (A.reserve(bytes, bytes) != hipSuccess) return false;
    double* h = (double*)A.host;
    double* d = (double*)A.dev;
    std::memcpy(h, in, n_in * sizeof(double));
    if (hipMemcpyAsync(d, h, n_in * sizeof(double), hipMemcpyHostToDevice, A.stream) != hipSuccess) return false;
    if (launch(d, d + n_in, A.stream) != hipSuccess) return false;
    if (hipMemcpyAsync(h + n_in, d + n_in, n_out * sizeof(double), hipMemcpyDeviceToHost, A.stream) != hipSuccess) return false;
    if (hipStreamSynchronize(A.stream) != hipSuccess) return false;
    std::memcpy(out, h + n_in, n_out * sizeof(double));
    return true;
}

double legacy_io(int io_type, double v, double r0, double r1, double v0, double k, double n) {
    // The reference's scalar helpers take v0 explicitly (ssnode.c:25-53), so the constants are
    // filled in directly instead of being derived from the soft bound.
    ssn::IoConsts<double> c;
    c.io_type = io_type; c.k = k; c.n = n; c.v0 = v0; c.soft = r0; c.hard = r1;
    c.lin_slope = k * std::pow(v0, n - 1.0) * n;
    c.tanh_gain = n * r0 / ((r1 - r0) * v0);
    c.span = r1 - r0;
    c.span_gain = c.span * c.tanh_gain;
    c.log2k = std::log2(k);
    double out = std::numeric_limits<double>::quiet_NaN();
    scalar_roundtrip(&v, 1, &out, 1, [&](double* dv, double* dout, hipStream_t st) {
        return ssn::launch_io_eval<double>(dv, dout, 1, c, st);
    });
    return out;
}

template <typename T>
ssn::IoConsts<T> gen_io_consts(const ssn_gen_params& g) {
    ssn_solver_params p;
    std::memset(&p, 0, sizeof(p));
    p.io_type = g.io_type; p.k = g.k; p.n = g.n; p.rate_soft_bound = g.rate_soft_bound; p.rate_hard_bound = g.rate_hard_bound;
    return ssn::make_io_consts<T>(p);
}

// MFMA generator kernels: 0 = not used, else the number of 4-stimulus groups per workgroup.  kernel: 0 automatic
// (two groups when that already gives >= 192 workgroups, one group when only that fills the chip -- 128 draws x 8
// stimuli of the paper's runs -- else the tile kernels, which run one workgroup per (draw, stimulus)), 1 tile kernels,
// 2 MFMA with two groups per workgroup, 3 MFMA with one group per workgroup.
static int mfma_groups_for(int kernel, bool mfma_ok, int B, int NB) {
    if (kernel == 2 || kernel == 4 || kernel == 6 || kernel == 8) return 2;
    if (kernel == 3 || kernel == 5) return 1;
    if (kernel != 0 || !mfma_ok) return 0;
    if ((long)B * ((NB + 7) / 8) >= 192) return 2;
    if ((long)B * ((NB + 3) / 4) >= 192) return 1;
    return 0;
}

// automatic choice among the fp16-split forward kernels with two groups per unit: two draws per workgroup (ssn_duo.hip,
// 3.4 ms at C3) when that still gives every CU a workgroup, else one draw per workgroup (wide form, 5.3 ms at C3 but
// twice the workgroups)
static bool duo_default(int groups, int B, int NB) { return groups == 2 && (long)B * ((NB + 7) / 8) > 256; }

template <typename T>
int gen_forward_impl(const T* W, const T* ext, T* time_avg, T* dyn_row, T* rate_row, T* traj, T* df, int B, int NB,
                     int M, const ssn_gen_params* g, void* stream) {
    if (B == 0 || NB == 0) return 0;
    if (!g || !W || !ext || !time_avg || !dyn_row || !rate_row || (traj == nullptr) != (df == nullptr) || M <= 0 ||
        (M & 1) || g->seqlen < 1 || g->skip_steps < 0 || g->skip_steps >= g->seqlen || g->io_type < 0 || g->io_type > 2 ||
        !ssn::gen_supported<T>(M)) {
        g_last_error = "ssn_gen_forward: invalid argument or unsupported size";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    ssn::GenFwdArgs<T> a;
    a.W = W; a.ext = ext; a.time_avg = time_avg; a.dyn_row = dyn_row; a.rate_row = rate_row; a.traj = traj; a.df = df;
    a.B = B; a.NB = NB; a.M = M; a.seqlen = g->seqlen; a.skip = g->skip_steps;
    a.eps_E = (T)(g->dt / g->tau_E); a.eps_I = (T)(g->dt / g->tau_I); a.theta = (T)g->rate_penalty_threshold;
    a.io = gen_io_consts<T>(*g);
    if constexpr (sizeof(T) == 4) {
        // (trajectory stores address one draw's block with 32-bit byte offsets)
        const bool mfma_ok = ssn::gen_mfma_supported(M, NB) && (!traj || (long)NB * g->seqlen * M < (1L << 29));
        if (g->kernel >= 2 && g->kernel <= 8 && (!mfma_ok || g->kernel == 7)) {
            g_last_error = "ssn_gen_forward: the MFMA kernels need fp32, NB >= 4 and 2N <= 208";
            return SSN_ERR_BASE + (int)hipErrorInvalidValue;
        }
        const bool split_ok = mfma_ok && ssn::gen_split_rshift(a) >= 0;
        if (g->kernel >= 4 && g->kernel <= 8 && !split_ok) {
            g_last_error = "ssn_gen_forward: the fp16-split MFMA kernel needs the saturating I/O function (asym_tanh), "
                           "rate_hard_bound < 3e4 and dt <= tau";
            return SSN_ERR_BASE + (int)hipErrorInvalidValue;
        }
        if (const int groups = mfma_groups_for(g->kernel, mfma_ok, B, NB)) {
            a.mfma_groups = groups;
            a.split_narrow = g->kernel == 6;
            const bool split = g->kernel >= 4 || (g->kernel == 0 && split_ok && forward_split_default());
            if (g->kernel == 8 || (g->kernel == 0 && split && duo_default(groups, B, NB)))
                SSN_TRY(ssn::launch_gen_forward_duo(a, (hipStream_t)stream));
            else if (split) SSN_TRY(ssn::launch_gen_forward_split(a, (hipStream_t)stream));
            else SSN_TRY(ssn::launch_gen_forward_mfma(a, (hipStream_t)stream));
            return 0;
        }
    }
    SSN_TRY(ssn::launch_gen_forward<T>(a, (hipStream_t)stream));
    return 0;
}

template <typename T>
int gen_backward_impl(const T* W, const T* traj, T* delta, const T* gta, T* g_ext, double c_dyn, double c_rate, int B,
                      int NB, int M, const ssn_gen_params* g, void* stream, float* dmax = nullptr, int* tracked = nullptr) {
    if (tracked) *tracked = 0;
    if (B == 0 || NB == 0) return 0;
    if (!g || !W || !traj || !delta || !gta || M <= 0 || (M & 1) || g->seqlen < 1 || g->skip_steps < 0 ||
        g->skip_steps >= g->seqlen || !ssn::gen_supported<T>(M)) {
        g_last_error = "ssn_gen_backward: invalid argument or unsupported size";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    ssn::GenBwdArgs<T> a;
    a.W = W; a.traj = traj; a.delta = delta; a.g_time_avg = gta; a.g_ext = g_ext;
    a.B = B; a.NB = NB; a.M = M; a.seqlen = g->seqlen; a.skip = g->skip_steps;
    a.eps_E = (T)(g->dt / g->tau_E); a.eps_I = (T)(g->dt / g->tau_I); a.theta = (T)g->rate_penalty_threshold;
    a.c_dyn = (T)c_dyn; a.c_rate = (T)c_rate;
    if constexpr (sizeof(T) == 4) {
        const bool mfma_ok = ssn::gen_mfma_supported(M, NB) && (long)NB * g->seqlen * M < (1L << 29);
        if (g->kernel >= 2 && g->kernel <= 8 && !mfma_ok) {
            g_last_error = "ssn_gen_backward: the MFMA kernel needs fp32, NB >= 4 and 2N <= 208";
            return SSN_ERR_BASE + (int)hipErrorInvalidValue;
        }
        if (const int groups = mfma_groups_for(g->kernel, mfma_ok, B, NB)) {
            a.mfma_groups = groups;
            const bool split_ok = ssn::gen_split_backward_supported(M, NB);
            a.split_narrow = g->kernel == 6;
            const bool split = split_ok && (g->kernel >= 4 || (g->kernel == 0 && forward_split_default()));
            if (split_ok && (g->kernel == 8 || (g->kernel == 0 && forward_split_default() && duo_default(groups, B, NB)))) {
                if (dmax && tracked) {                  // max |delta| per draw for ssn_weight_grad_scaled_f32 (atomic max of bit patterns)
                    SSN_TRY(hipMemsetAsync(dmax, 0, sizeof(float) * (size_t)B, (hipStream_t)stream));
                    a.dmax = reinterpret_cast<unsigned*>(dmax);
                    *tracked = 1;
                }
                SSN_TRY(ssn::launch_gen_backward_duo(a, (hipStream_t)stream));
            } else if (split) {
                if (dmax && tracked) {                  // (the alternating split sweep keeps the same per-step maxima)
                    SSN_TRY(hipMemsetAsync(dmax, 0, sizeof(float) * (size_t)B, (hipStream_t)stream));
                    a.dmax = reinterpret_cast<unsigned*>(dmax);
                    *tracked = 1;
                }
                SSN_TRY(ssn::launch_gen_backward_split(a, (hipStream_t)stream));
            }
            else SSN_TRY(ssn::launch_gen_backward_mfma(a, (hipStream_t)stream));
            return 0;
        }
    }
    SSN_TRY(ssn::launch_gen_backward<T>(a, (hipStream_t)stream));
    return 0;
}

}  // namespace

template <typename T>
static int build_dw_impl(const T* z, const T* J, const T* D, const T* S, int which, T* dW, int B, int N, void* stream) {
    if (B < 0 || N < 1 || which < 0 || which > 2 || (B > 0 && (!dW || (which > 0 && !z) || !J || !D || !S))) {
        g_last_error = "ssn_build_dw: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    T jds[12];
    for (int q = 0; q < 4; ++q) { jds[q] = J[q]; jds[4 + q] = D[q]; jds[8 + q] = S[q]; }
    SSN_TRY(ssn::launch_build_dw<T>(z, jds, which, dW, B, N, (hipStream_t)stream));
    return 0;
}
template <typename T>
static int ss_system_impl(const T* R, const T* W, const T* dW, int dw_per_draw, const T* I, int i_per_draw, int nz,
                          int nb, int M, const ssn_solver_params* p, T* A, T* rhs, void* stream) {
    if (nz == 0 || nb == 0) return 0;
    if (!p || !R || !W || !dW || !I || !A || !rhs || nz < 0 || nb < 0 || M <= 0 || (M & 1) || p->io_type < 0 ||
        p->io_type > 2) {
        g_last_error = "ssn_ss_grad_system: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    SSN_TRY(ssn::launch_ss_system<T>(R, W, dW, dw_per_draw, I, i_per_draw, ssn::make_io_consts<T>(*p), nz, nb, M, A, rhs,
                                     (hipStream_t)stream));
    return 0;
}
extern "C" {

long ssn_critic_num_params(const int* dims, int nlayers) {
    long n = 0;
    for (int l = 0; l < nlayers; ++l) n += (long)dims[l] * dims[l + 1] + dims[l + 1];
    return n + dims[nlayers];
}
// Critics whose layers are all <= 128 wide run through the fused row-block kernels (ssn_critic_fused.hip: 3 launches per
// update, fp32 arithmetic whatever `precision` says); wider ones through the layer-by-layer MFMA GEMM chain.  The
// workspace is sized for whichever path the sizes select.
static size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }
// SSN_CRITIC_FUSED=0 in the environment keeps small critics on the layer-by-layer path too (A/B timing, tests of that path)
// Above ~2048 stacked rows the layer-by-layer chain (fixed ~250 us of launch latency, MFMA arithmetic) is as fast as the
// fused kernels (plain FMAs, time proportional to the rows).
static bool fused_ok(const int* dims, int nlayers, long rows) {
    static const bool enabled = [] { const char* v = std::getenv("SSN_CRITIC_FUSED"); return !(v && v[0] == '0'); }();
    return enabled && rows <= 2048 && ssn::critic_fused_supported(dims, nlayers);
}
size_t ssn_critic_workspace_floats(const int* dims, int nlayers, int batch_gd, int batch_p) {
    const size_t base = ssn::critic_workspace_floats(dims, nlayers, batch_gd, batch_p);
    return ssn::critic_fused_supported(dims, nlayers) ? max_sz(base, ssn::critic_fused_workspace_floats(dims, nlayers, batch_gd, batch_p)) : base;
}
int ssn_critic_forward(const float* params, const int* dims, int nlayers, const float* x, const float* cond, int batch,
                       int hide_cell_type, float* out, float* workspace, int precision, void* stream) {
    if (batch == 0) return 0;
    if (cond && fused_ok(dims, nlayers, batch)) {
        SSN_TRY(ssn::critic_fused_forward(params, dims, nullptr, nlayers, x, cond, batch, hide_cell_type, out, workspace, (hipStream_t)stream));
        return 0;
    }
    SSN_TRY(ssn::critic_forward(params, dims, nlayers, x, cond, batch, hide_cell_type, out, workspace, precision == 0,
                                (hipStream_t)stream));
    return 0;
}
int ssn_critic_loss_grad(const float* params, const int* dims, int nlayers, const float* xg, const float* cg,
                         const float* xd, const float* cd, const float* xp, const float* cp, int ng, int nd, int np,
                         float lmd, int hide_cell_type, float* grads, float* stats, float* dvals, float* workspace,
                         int precision, void* stream) {
    if (cg && cd && cp && fused_ok(dims, nlayers, (long)ng + nd + np)) {
        SSN_TRY(ssn::critic_fused_loss_grad(params, dims, nullptr, nlayers, xg, cg, xd, cd, xp, cp, ng, nd, np, lmd, hide_cell_type,
                                            grads, stats, dvals, workspace, (hipStream_t)stream));
        return 0;
    }
    SSN_TRY(ssn::critic_loss_grad(params, dims, nlayers, xg, cg, xd, cd, xp, cp, ng, nd, np, lmd, hide_cell_type, grads,
                                  stats, dvals, workspace, precision == 0, (hipStream_t)stream));
    return 0;
}
int ssn_critic_input_grad(const float* params, const int* dims, int nlayers, const float* x, const float* cond, int batch,
                          int hide_cell_type, float scale, float* gx, float* stats, float* workspace, int precision,
                          void* stream) {
    if (batch == 0) return 0;
    if (cond && fused_ok(dims, nlayers, batch)) {
        SSN_TRY(ssn::critic_fused_input_grad(params, dims, nullptr, nlayers, x, cond, batch, hide_cell_type, scale, gx, stats,
                                             workspace, (hipStream_t)stream));
        return 0;
    }
    SSN_TRY(ssn::critic_input_grad(params, dims, nlayers, x, cond, batch, hide_cell_type, scale, gx, stats, workspace,
                                   precision == 0, (hipStream_t)stream));
    return 0;
}
size_t ssn_critic_norm_workspace_floats(const int* dims, int nlayers, int batch_gd, int batch_p) {
    const size_t base = ssn::critic_norm_workspace_floats(dims, nlayers, batch_gd, batch_p);
    return ssn::critic_fused_supported(dims, nlayers) ? max_sz(base, ssn::critic_fused_workspace_floats(dims, nlayers, batch_gd, batch_p)) : base;
}
// The same three calls for a hidden nonlinearity x > 0 ? x : leak * x (lasagne's leaky_rectify = 0.01, very_leaky_rectify = 1/3,
// linear = 1; plain layers only).  Always the layer-by-layer path.
int ssn_critic_forward_leaky(const float* params, const int* dims, int nlayers, const float* x, const float* cond, int batch,
                             int hide_cell_type, float leak, float* out, float* workspace, int precision, void* stream) {
    if (batch == 0) return 0;
    SSN_TRY(ssn::critic_forward(params, dims, nlayers, x, cond, batch, hide_cell_type, out, workspace, precision == 0,
                                (hipStream_t)stream, leak));
    return 0;
}
int ssn_critic_loss_grad_leaky(const float* params, const int* dims, int nlayers, const float* xg, const float* cg,
                               const float* xd, const float* cd, const float* xp, const float* cp, int ng, int nd, int np,
                               float lmd, int hide_cell_type, float leak, float* grads, float* stats, float* dvals,
                               float* workspace, int precision, void* stream) {
    SSN_TRY(ssn::critic_loss_grad(params, dims, nlayers, xg, cg, xd, cd, xp, cp, ng, nd, np, lmd, hide_cell_type, grads,
                                  stats, dvals, workspace, precision == 0, (hipStream_t)stream, leak));
    return 0;
}
int ssn_critic_input_grad_leaky(const float* params, const int* dims, int nlayers, const float* x, const float* cond, int batch,
                                int hide_cell_type, float leak, float scale, float* gx, float* stats, float* workspace,
                                int precision, void* stream) {
    if (batch == 0) return 0;
    SSN_TRY(ssn::critic_input_grad(params, dims, nlayers, x, cond, batch, hide_cell_type, scale, gx, stats, workspace,
                                   precision == 0, (hipStream_t)stream, leak));
    return 0;
}
// ---- the general layer-by-layer critic: any lasagne nonlinearity, learnable scale after a layer normalisation ----------------
static bool act_flags_ok(const int* flags, int nlayers, int act) {
    if (act < 0 || act > 7 || nlayers < 0 || nlayers > 8) return false;
    for (int l = 0; flags && l < nlayers; ++l) if (flags[l] != 0 && flags[l] != 1 && flags[l] != 3) return false;
    return true;
}
long ssn_critic_num_params_act(const int* dims, const int* layer_flags, int nlayers) {
    if (!dims || !act_flags_ok(layer_flags, nlayers, 0)) return -1;
    return ssn::critic_act_num_params(dims, layer_flags, nlayers);
}
int ssn_critic_forward_act(const float* params, const int* dims, const int* layer_flags, int nlayers, int act, const float* x,
                           const float* cond, int batch, int hide_cell_type, float* out, float* workspace, int precision,
                           void* stream) {
    if (!act_flags_ok(layer_flags, nlayers, act)) { g_last_error = "ssn_critic_forward_act: invalid layer flags / activation"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    if (batch == 0) return 0;
    SSN_TRY(ssn::critic_norm_forward(params, dims, layer_flags, nlayers, x, cond, batch, hide_cell_type, out, workspace,
                                     precision == 0, (hipStream_t)stream, act));
    return 0;
}
int ssn_critic_loss_grad_act(const float* params, const int* dims, const int* layer_flags, int nlayers, int act, const float* xg,
                             const float* cg, const float* xd, const float* cd, const float* xp, const float* cp, int ng,
                             int nd, int np, float lmd, int hide_cell_type, float* grads, float* stats, float* dvals,
                             float* workspace, int precision, void* stream) {
    if (!act_flags_ok(layer_flags, nlayers, act)) { g_last_error = "ssn_critic_loss_grad_act: invalid layer flags / activation"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::critic_norm_loss_grad(params, dims, layer_flags, nlayers, xg, cg, xd, cd, xp, cp, ng, nd, np, lmd,
                                       hide_cell_type, grads, stats, dvals, workspace, precision == 0, (hipStream_t)stream, act));
    return 0;
}
int ssn_critic_input_grad_act(const float* params, const int* dims, const int* layer_flags, int nlayers, int act, const float* x,
                              const float* cond, int batch, int hide_cell_type, float scale, float* gx, float* stats,
                              float* workspace, int precision, void* stream) {
    if (!act_flags_ok(layer_flags, nlayers, act)) { g_last_error = "ssn_critic_input_grad_act: invalid layer flags / activation"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    if (batch == 0) return 0;
    SSN_TRY(ssn::critic_norm_input_grad(params, dims, layer_flags, nlayers, x, cond, batch, hide_cell_type, scale, gx, stats,
                                        workspace, precision == 0, (hipStream_t)stream, act));
    return 0;
}
int ssn_critic_accuracy_act(const float* params, const int* dims, const int* layer_flags, int nlayers, int act, const float* xg,
                            const float* cg, const float* xd, const float* cd, int ng, int nd, int hide_cell_type, float* acc,
                            float* dvals, float* workspace, int precision, void* stream) {
    if (int rc = ssn_critic_forward_act(params, dims, layer_flags, nlayers, act, xg, cg, ng, hide_cell_type, dvals, workspace, precision, stream)) return rc;
    if (int rc = ssn_critic_forward_act(params, dims, layer_flags, nlayers, act, xd, cd, nd, hide_cell_type, dvals + ng, workspace, precision, stream)) return rc;
    SSN_TRY(ssn::launch_mean_diff(dvals, ng, nd, acc, (hipStream_t)stream));
    return 0;
}
static bool norm_flags_plain(const int* layer_norm, int nlayers) {
    for (int l = 0; layer_norm && l < nlayers; ++l) if (layer_norm[l] != 0 && layer_norm[l] != 1) return false;
    return true;
}
int ssn_critic_forward_norm(const float* params, const int* dims, const int* layer_norm, int nlayers, const float* x,
                            const float* cond, int batch, int hide_cell_type, float* out, float* workspace, int precision,
                            void* stream) {
    if (!norm_flags_plain(layer_norm, nlayers)) { g_last_error = "ssn_critic_forward_norm: layer_norm flags are 0 / 1 (scaled layers: ssn_critic_forward_act)"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    if (batch == 0) return 0;
    if (cond && fused_ok(dims, nlayers, batch)) {
        SSN_TRY(ssn::critic_fused_forward(params, dims, layer_norm, nlayers, x, cond, batch, hide_cell_type, out, workspace, (hipStream_t)stream));
        return 0;
    }
    SSN_TRY(ssn::critic_norm_forward(params, dims, layer_norm, nlayers, x, cond, batch, hide_cell_type, out, workspace,
                                     precision == 0, (hipStream_t)stream));
    return 0;
}
int ssn_critic_loss_grad_norm(const float* params, const int* dims, const int* layer_norm, int nlayers, const float* xg,
                              const float* cg, const float* xd, const float* cd, const float* xp, const float* cp, int ng,
                              int nd, int np, float lmd, int hide_cell_type, float* grads, float* stats, float* dvals,
                              float* workspace, int precision, void* stream) {
    if (!norm_flags_plain(layer_norm, nlayers)) { g_last_error = "ssn_critic_loss_grad_norm: layer_norm flags are 0 / 1 (scaled layers: ssn_critic_loss_grad_act)"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    if (cg && cd && cp && fused_ok(dims, nlayers, (long)ng + nd + np)) {
        SSN_TRY(ssn::critic_fused_loss_grad(params, dims, layer_norm, nlayers, xg, cg, xd, cd, xp, cp, ng, nd, np, lmd, hide_cell_type,
                                            grads, stats, dvals, workspace, (hipStream_t)stream));
        return 0;
    }
    SSN_TRY(ssn::critic_norm_loss_grad(params, dims, layer_norm, nlayers, xg, cg, xd, cd, xp, cp, ng, nd, np, lmd,
                                       hide_cell_type, grads, stats, dvals, workspace, precision == 0, (hipStream_t)stream));
    return 0;
}
int ssn_critic_input_grad_norm(const float* params, const int* dims, const int* layer_norm, int nlayers, const float* x,
                               const float* cond, int batch, int hide_cell_type, float scale, float* gx, float* stats,
                               float* workspace, int precision, void* stream) {
    if (!norm_flags_plain(layer_norm, nlayers)) { g_last_error = "ssn_critic_input_grad_norm: layer_norm flags are 0 / 1 (scaled layers: ssn_critic_input_grad_act)"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    if (batch == 0) return 0;
    if (cond && fused_ok(dims, nlayers, batch)) {
        SSN_TRY(ssn::critic_fused_input_grad(params, dims, layer_norm, nlayers, x, cond, batch, hide_cell_type, scale, gx, stats,
                                             workspace, (hipStream_t)stream));
        return 0;
    }
    SSN_TRY(ssn::critic_norm_input_grad(params, dims, layer_norm, nlayers, x, cond, batch, hide_cell_type, scale, gx, stats,
                                        workspace, precision == 0, (hipStream_t)stream));
    return 0;
}
// mean D(xg) - mean D(xd) in one call (two critic forwards into `dvals`, one reduction in a fixed order): layer_norm NULL or all
// zero = plain layers, leak as in the _leaky entry points.
static int critic_accuracy_forwards(const float* params, const int* dims, const int* layer_norm, int nlayers, float leak, const float* xg,
                                    const float* cg, const float* xd, const float* cd, int ng, int nd, int hide_cell_type,
                                    float* dvals, float* workspace, int precision, void* stream, bool inputs_ready = false);
int ssn_critic_accuracy(const float* params, const int* dims, const int* layer_norm, int nlayers, float leak, const float* xg,
                        const float* cg, const float* xd, const float* cd, int ng, int nd, int hide_cell_type, float* acc,
                        float* dvals, float* workspace, int precision, void* stream) {
    if (int rc = critic_accuracy_forwards(params, dims, layer_norm, nlayers, leak, xg, cg, xd, cd, ng, nd, hide_cell_type, dvals,
                                          workspace, precision, stream)) return rc;
    SSN_TRY(ssn::launch_mean_diff(dvals, ng, nd, acc, (hipStream_t)stream));
    return 0;
}
static int critic_accuracy_forwards(const float* params, const int* dims, const int* layer_norm, int nlayers, float leak, const float* xg,
                                    const float* cg, const float* xd, const float* cd, int ng, int nd, int hide_cell_type,
                                    float* dvals, float* workspace, int precision, void* stream, bool inputs_ready) {
    bool norm = false;
    for (int l = 0; layer_norm && l < nlayers; ++l) norm = norm || layer_norm[l] != 0;
    if (!norm && !(cg && cd && (fused_ok(dims, nlayers, ng) || fused_ok(dims, nlayers, nd)))) {
        // plain layers on the layer-by-layer path: ONE pass over the stacked rows [xg; xd] (the values of two separate
        // forwards, bit for bit; narrow critics keep their single-launch forwards)
        SSN_TRY(ssn::critic_forward2(params, dims, nlayers, xg, cg, ng, xd, cd, nd, hide_cell_type, dvals, workspace, precision == 0,
                                     (hipStream_t)stream, leak, inputs_ready));
        return 0;
    }
    const struct { const float* x; const float* c; int n; float* out; } part[2] = {{xg, cg, ng, dvals}, {xd, cd, nd, dvals + ng}};
    for (int i = 0; i < 2; ++i) {
        int rc;
        if (norm) rc = ssn_critic_forward_norm(params, dims, layer_norm, nlayers, part[i].x, part[i].c, part[i].n, hide_cell_type,
                                               part[i].out, workspace, precision, stream);
        else if (leak != 0.f) rc = ssn_critic_forward_leaky(params, dims, nlayers, part[i].x, part[i].c, part[i].n, hide_cell_type,
                                                            leak, part[i].out, workspace, precision, stream);
        else rc = ssn_critic_forward(params, dims, nlayers, part[i].x, part[i].c, part[i].n, hide_cell_type, part[i].out,
                                     workspace, precision, stream);
        if (rc) return rc;
    }
    return 0;
}
static int optimizer_step_full(float* p, const float* g, float* s1, float* s2, long n, const ssn_opt_params* o, const double* gate,
                               double gate_bound, const float* clip_lo_v, const float* clip_hi_v, float* record, const float* record_tail,
                               void* stream);
static int optimizer_step_gated(float* p, const float* g, float* s1, float* s2, long n, const ssn_opt_params* o, const double* gate,
                                double gate_bound, void* stream);
int ssn_optimizer_step(float* p, const float* g, float* s1, float* s2, long n, const ssn_opt_params* o, void* stream) {
    return optimizer_step_gated(p, g, s1, s2, n, o, nullptr, 0.0, stream);
}
static int optimizer_step_gated(float* p, const float* g, float* s1, float* s2, long n, const ssn_opt_params* o, const double* gate,
                                double gate_bound, void* stream) {
    return optimizer_step_full(p, g, s1, s2, n, o, gate, gate_bound, nullptr, nullptr, nullptr, nullptr, stream);
}
static int optimizer_step_full(float* p, const float* g, float* s1, float* s2, long n, const ssn_opt_params* o, const double* gate,
                               double gate_bound, const float* clip_lo_v, const float* clip_hi_v, float* record, const float* record_tail,
                               void* stream) {
    if (!o || n < 0 || o->kind < 0 || o->kind > 2) {
        g_last_error = "ssn_optimizer_step: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    ssn::OptArgs a;
    a.gate = gate; a.gate_bound = gate_bound;
    a.clip_lo_v = clip_lo_v; a.clip_hi_v = clip_hi_v; a.record = record; a.record_tail = record_tail;
    a.p = p; a.g = g; a.s1 = s1; a.s2 = s2; a.n = n; a.kind = o->kind;
    a.lr = (float)o->learning_rate; a.beta1 = (float)o->beta1; a.beta2 = (float)o->beta2; a.eps = (float)o->epsilon;
    a.rho = (float)o->rho;
    // lasagne.updates.adam: a_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
    a.a_t = (float)(o->learning_rate * std::sqrt(1.0 - std::pow(o->beta2, o->step)) / (1.0 - std::pow(o->beta1, o->step)));
    a.l2_penalty = (float)o->reg_l2_penalty; a.l1_penalty = (float)o->reg_l1_penalty;
    a.l2_decay = (float)o->reg_l2_decay; a.l1_decay = (float)o->reg_l1_decay;
    a.clip = o->clip; a.clip_lo = (float)o->clip_lo; a.clip_hi = (float)o->clip_hi;
    a.skip_nonfinite = (o->reserved & 1) && n <= 64 && record;        // ssn_opt_params.reserved bit 0 (ssn_gen_apply_f32)
    SSN_TRY(ssn::optimizer_step(a, (hipStream_t)stream));
    return 0;
}

long ssn_gen_grads_ws_doubles(void) { return 2 * 128 + 1; }
int ssn_gen_grads_f32(const ssn_gen_grads* a, void* stream) {
    if (!a || !a->jds_part || a->B < 0 || a->nv < 0 || a->nv > 2 || !a->dmean || !a->ws || !a->out ||
        (a->nv > 0 && (!a->g_ext || !a->ext_base || !a->zin || a->NB <= 0 || a->M <= 0 || (a->M & 1)))) {
        g_last_error = "ssn_gen_grads: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    ssn::GenGradsArgs g{};
    g.part = a->jds_part; g.B = a->B; g.nv = a->nv; g.g_ext = a->g_ext; g.ext_base = a->ext_base; g.zin = a->zin;
    g.NB = a->NB; g.M = a->M; g.dmean = a->dmean; g.pens = a->pens64; g.dynamics_cost = a->dynamics_cost; g.rate_cost = a->rate_cost;
    g.ws = a->ws; g.out = a->out;
    SSN_TRY(ssn::launch_gen_grads(g, (hipStream_t)stream));
    return 0;
}
int ssn_gen_apply_f32(float* params, const float* grads, float* s1, float* s2, int n, const ssn_opt_params* opt, const float* clip_lo,
                      const float* clip_hi, float* record, void* stream) {
    if (!params || !grads || n <= 0 || (clip_lo == nullptr) != (clip_hi == nullptr)) {
        g_last_error = "ssn_gen_apply: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    return optimizer_step_full(params, grads, s1, s2, n, opt, nullptr, 0.0, clip_lo, clip_hi, record, record ? grads + n : nullptr, stream);
}
static int critic_step_impl(const ssn_critic_step* a, const double* gate, double gate_bound, void* stream);
int ssn_critic_step_run(const ssn_critic_step* a, void* stream) { return critic_step_impl(a, nullptr, 0.0, stream); }
int ssn_critic_step_gated_run(const ssn_critic_step* a, double rate_penalty_bound, void* stream) {
    if (!a || !a->pens64 || !(rate_penalty_bound > 0.0)) {
        g_last_error = "ssn_critic_step_gated_run: needs pens64 and a positive bound";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    return critic_step_impl(a, a->pens64 + 1, rate_penalty_bound, stream);
}
static int critic_step_impl(const ssn_critic_step* a, const double* gate, double gate_bound, void* stream) {
    if (!a || !a->params || !a->dims || !a->xg || !a->xd || !a->eps || !a->xp || !a->grads || !a->stats || !a->dvals ||
        !a->workspace || !a->opt || !a->acc_dvals || !a->tail || a->n <= 0 || a->nlayers < 0 || a->nseg < 0) {
        g_last_error = "ssn_critic_step_run: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    bool norm = false;
    for (int l = 0; a->layer_norm && l < a->nlayers; ++l) norm = norm || a->layer_norm[l] != 0;
    const int n = a->n, nx = a->dims[0] - (a->cond ? 3 : 0);        // cond NULL: the unconditional critic (dims[0] = nx)
    int rc;
    // plain layers on the layer-by-layer path with the one condition array of the loop: the penalty points and the three input
    // blocks come from ONE launch inside the loss pass (the bits of ssn_interpolate_f32 + the input kernels)
    const bool one_launch_inputs = !norm && a->cond && !fused_ok(a->dims, a->nlayers, 3L * n);
    if (!one_launch_inputs && (rc = ssn_interpolate_f32(a->eps, a->xd, a->xg, a->xp, n, nx, stream))) return rc;
    if (norm) rc = ssn_critic_loss_grad_norm(a->params, a->dims, a->layer_norm, a->nlayers, a->xg, a->cond, a->xd, a->cond, a->xp, a->cond,
                                             n, n, n, a->lmd, a->hide_cell_type, a->grads, a->stats, a->dvals, a->workspace, a->precision, stream);
    else if (one_launch_inputs) {
        SSN_TRY(ssn::critic_loss_grad(a->params, a->dims, a->nlayers, a->xg, a->cond, a->xd, a->cond, a->xp, a->cond, n, n, n, a->lmd,
                                      a->hide_cell_type, a->grads, a->stats, a->dvals, a->workspace, a->precision == 0,
                                      (hipStream_t)stream, a->leak, a->eps, a->xp));
        rc = 0;
    }
    else if (a->leak != 0.f) rc = ssn_critic_loss_grad_leaky(a->params, a->dims, a->nlayers, a->xg, a->cond, a->xd, a->cond, a->xp, a->cond,
                                                             n, n, n, a->lmd, a->hide_cell_type, a->leak, a->grads, a->stats, a->dvals,
                                                             a->workspace, a->precision, stream);
    else rc = ssn_critic_loss_grad(a->params, a->dims, a->nlayers, a->xg, a->cond, a->xd, a->cond, a->xp, a->cond, n, n, n, a->lmd,
                                   a->hide_cell_type, a->grads, a->stats, a->dvals, a->workspace, a->precision, stream);
    if (rc) return rc;
    const long nparams = ssn_critic_num_params(a->dims, a->nlayers);
    if ((rc = optimizer_step_gated(a->params, a->grads, a->opt_s1, a->opt_s2, nparams, a->opt, gate, gate_bound, stream))) return rc;
    // (the loss pass of the one-launch-inputs form left the input block of [xg; xd] at the head of the workspace, where the
    // stacked forward of the accuracy builds it: same rows, same conditions -- not built again)
    if ((rc = critic_accuracy_forwards(a->params, a->dims, a->layer_norm, a->nlayers, a->leak, a->xg, a->cond, a->xd, a->cond, n, n,
                                       a->hide_cell_type, a->acc_dvals, a->workspace, a->precision, stream, one_launch_inputs))) return rc;
    if (a->nseg > 0 && (!a->seg_bounds || !a->seg_ws)) { g_last_error = "ssn_critic_step_run: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    // accuracy, sums of squares and the head of the record: the chunk sums, then ONE finishing launch (same bits as
    // ssn_critic_accuracy + ssn_segment_sqnorms2_f32 + the head kernel)
    SSN_TRY(ssn::launch_step_finish(a->params, a->seg_bounds, a->nseg, a->seg_ws, a->acc_dvals, n, n, a->pens64, a->stats, a->tail,
                                    (hipStream_t)stream));
    return 0;
}

static ssn::FFArgs ff_args(const ssn_ff_params& p) {
    ssn::FFArgs a{};
    a.nsam = p.nsam; a.nhid = p.nhid; a.ni = p.ni; a.box = p.box;
    a.RF_l = (float)p.RF_l; a.RF_d = (float)p.RF_d; a.TH = (float)p.TH; a.TH_d = (float)p.TH_d;
    a.J = (float)p.J; a.a = (float)p.a;
    return a;
}
// Is the stimulus set a 3 x 3 x 3 product lattice in the model script's order?  (27 x 3 floats: read back once per call.)
static bool ff_lattice(const float* stim, int ni, ssn::FFLattice& lat, void* stream) {
    if (ni != 27) return false;
    float hs[27][3];
    if (hipMemcpyAsync(hs, stim, sizeof(hs), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return false;
    for (int k = 0; k < 3; ++k) { lat.x[k] = hs[9 * k][0]; lat.y[k] = hs[3 * k][1]; lat.z[k] = hs[k][2]; }
    bool lattice = true;
    for (int i = 0; i < 27 && lattice; ++i)
        lattice = hs[i][0] == lat.x[i / 9] && hs[i][1] == lat.y[(i / 3) % 3] && hs[i][2] == lat.z[i % 3];
    return lattice;
}
int ssn_ff_forward_sparse_f32(const float* RF_w, const int* conn_idx, const float* conn_str, int ncon, const float* TH_sam,
                              const float* stim, float* out, float* q, float* den, const ssn_ff_params* p, void* stream) {
    if (!p || p->nsam < 0 || p->nhid < 1 || p->ni < 1 || p->ni > 32 || p->box < 1 || ncon < 0 || (q == nullptr) != (den == nullptr) ||
        (p->nsam > 0 && (!RF_w || !TH_sam || !stim || !out || (ncon > 0 && (!conn_idx || !conn_str))))) {
        g_last_error = "ssn_ff_forward_sparse: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    ssn::FFArgs a = ff_args(*p);
    a.RF_w = RF_w; a.TH_sam = TH_sam; a.stim = stim; a.out = out; a.q = q; a.den = den;
    ssn::FFLattice lat;
    const bool lattice = ff_lattice(stim, p->ni, lat, stream);
    SSN_TRY(ssn::launch_ff_forward_sparse(a, lattice ? &lat : nullptr, conn_idx, conn_str, ncon, (hipStream_t)stream));
    return 0;
}
int ssn_ff_backward_sparse_f32(const float* RF_w, const int* conn_idx, const float* conn_str, int ncon, const float* stim, const float* q,
                               const float* den, const float* gq, float* dsig, const ssn_ff_params* p, void* stream) {
    if (!p || p->nsam < 0 || p->nhid < 1 || p->ni < 1 || p->ni > 32 || p->box < 1 || ncon < 0 || !q || !den || !gq || !dsig ||
        (p->nsam > 0 && (!RF_w || !stim || (ncon > 0 && (!conn_idx || !conn_str))))) {
        g_last_error = "ssn_ff_backward_sparse: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    ssn::FFArgs a = ff_args(*p);
    a.RF_w = RF_w; a.stim = stim; a.q = const_cast<float*>(q); a.den = const_cast<float*>(den);
    SSN_TRY(ssn::launch_ff_backward_sparse(a, conn_idx, conn_str, ncon, gq, dsig, (hipStream_t)stream));
    return 0;
}
int ssn_ff_forward_f32(const float* RF_w, const float* FF_con, const float* FF_str, const float* TH_sam,
                       const float* stim, float* out, float* q, float* den, const ssn_ff_params* p, void* stream) {
    if (!p || p->nsam < 0 || p->nhid < 1 || p->ni < 1 || p->ni > 32 || p->box < 1 || (q == nullptr) != (den == nullptr)) {
        g_last_error = "ssn_ff_forward: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    ssn::FFArgs a = ff_args(*p);
    a.RF_w = RF_w; a.FF_con = FF_con; a.FF_str = FF_str; a.TH_sam = TH_sam; a.stim = stim; a.out = out; a.q = q; a.den = den;
    ssn::FFLattice lat;
    const bool lattice = ff_lattice(stim, p->ni, lat, stream);
    SSN_TRY(ssn::launch_ff_forward(a, lattice ? &lat : nullptr, (hipStream_t)stream));
    return 0;
}
int ssn_ff_backward_f32(const float* RF_w, const float* FF_con, const float* FF_str, const float* stim, const float* q,
                        const float* den, const float* gq, float* dsig, const ssn_ff_params* p, void* stream) {
    if (!p || p->nsam < 0 || p->nhid < 1 || p->ni < 1 || p->ni > 32 || p->box < 1 || !q || !den || !gq || !dsig) {
        g_last_error = "ssn_ff_backward: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    ssn::FFArgs a = ff_args(*p);
    a.RF_w = RF_w; a.FF_con = FF_con; a.FF_str = FF_str; a.stim = stim;
    a.q = const_cast<float*>(q); a.den = const_cast<float*>(den);
    SSN_TRY(ssn::launch_ff_backward(a, gq, dsig, (hipStream_t)stream));
    return 0;
}

int ssn_build_dw_f32(const float* z, const float* J, const float* D, const float* S, int which, float* dW, int B, int N,
                     void* stream) { return build_dw_impl<float>(z, J, D, S, which, dW, B, N, stream); }
int ssn_build_dw_f64(const double* z, const double* J, const double* D, const double* S, int which, double* dW, int B,
                     int N, void* stream) { return build_dw_impl<double>(z, J, D, S, which, dW, B, N, stream); }

int ssn_ss_grad_system_f32(const float* R, const float* W, const float* dW, int dw_per_draw, const float* I,
                           int i_per_draw, int nz, int nb, int M, const ssn_solver_params* p, float* A, float* rhs,
                           void* stream) {
    return ss_system_impl<float>(R, W, dW, dw_per_draw, I, i_per_draw, nz, nb, M, p, A, rhs, stream);
}
int ssn_ss_grad_system_f64(const double* R, const double* W, const double* dW, int dw_per_draw, const double* I,
                           int i_per_draw, int nz, int nb, int M, const ssn_solver_params* p, double* A, double* rhs,
                           void* stream) {
    return ss_system_impl<double>(R, W, dW, dw_per_draw, I, i_per_draw, nz, nb, M, p, A, rhs, stream);
}

int ssn_lu_solve_f32(float* A, float* rhs, int* info, int nsys, int M, int nrhs, void* stream) {
    if (nsys < 0 || M < 0 || (nsys > 0 && M > 0 && (!A || !rhs))) { g_last_error = "ssn_lu_solve: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_lu_solve<float>(A, rhs, info, nsys, M, nrhs, (hipStream_t)stream));
    return 0;
}
int ssn_lu_solve_f64(double* A, double* rhs, int* info, int nsys, int M, int nrhs, void* stream) {
    if (nsys < 0 || M < 0 || (nsys > 0 && M > 0 && (!A || !rhs))) { g_last_error = "ssn_lu_solve: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_lu_solve<double>(A, rhs, info, nsys, M, nrhs, (hipStream_t)stream));
    return 0;
}
int ssn_moment_sums_f32(const float* x, int B, int D, double* sums, void* stream) {
    if (B < 0 || D < 0 || (D > 0 && (!x || !sums))) {
        g_last_error = "ssn_moment_sums: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    SSN_TRY(ssn::launch_moment_sums(x, B, D, sums, (hipStream_t)stream));
    return 0;
}
int ssn_moment_loss_grad_f32(const float* x, const double* sums, double global_batch, const double* data_moments,
                             const double* weights, int B, int D, float* gx, double* out, void* stream) {
    if (B < 0 || D < 0 || !(global_batch > 0) || !out || (D > 0 && (!x || !sums || !data_moments || !weights || !gx))) {
        g_last_error = "ssn_moment_loss_grad: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    SSN_TRY(ssn::launch_moment_loss_grad(x, sums, global_batch, data_moments, weights, B, D, gx, out, (hipStream_t)stream));
    return 0;
}

int ssn_set_operand_precision(int mode) {
    return operand_precision_state().exchange(mode ? 1 : 0);
}
int ssn_get_operand_precision(void) { return operand_precision_state().load(); }
int ssn_solve_batch_variant_for(int B, int NB, int M, int dtype_bytes, const ssn_solver_params* p) {
    if (B <= 0 || NB <= 0) return -1;
    const int v = dtype_bytes == 8
        ? solve_batch_impl<double>(-1, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, B, NB, M, p, nullptr, true)
        : solve_batch_impl<float>(-1, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, B, NB, M, p, nullptr, true);
    return v >= SSN_ERR_BASE ? -1 : v;
}

int ssn_gen_forward_variant(int B, int NB, int M, int seqlen, int save, const ssn_gen_params* g) {
    if (!g || B <= 0 || NB <= 0 || M <= 0 || (M & 1) || !ssn::gen_supported<float>(M)) return -1;
    ssn::GenFwdArgs<float> a{};
    a.B = B; a.NB = NB; a.M = M; a.seqlen = seqlen;
    a.eps_E = (float)(g->dt / g->tau_E); a.eps_I = (float)(g->dt / g->tau_I);
    a.io = gen_io_consts<float>(*g);
    const bool mfma_ok = ssn::gen_mfma_supported(M, NB) && (!save || (long)NB * seqlen * M < (1L << 29));
    const bool split_ok = mfma_ok && ssn::gen_split_rshift(a) >= 0;
    if ((g->kernel >= 2 && g->kernel <= 8 && !mfma_ok) || (g->kernel >= 4 && !split_ok) || g->kernel == 7) return -1;
    const int groups = mfma_groups_for(g->kernel, mfma_ok, B, NB);
    if (!groups) return 1;
    if (g->kernel == 8) return 8;
    const bool split = g->kernel >= 4 || (g->kernel == 0 && split_ok && forward_split_default());
    if (g->kernel == 0 && split && duo_default(groups, B, NB)) return 8;
    if (split && groups == 2 && (g->kernel == 6 || ssn::gen_split_wide_parts() == 0)) return 6;
    if (split && groups == 2 && ssn::gen_split_wide_parts() == 3) return 7;
    return (split ? 4 : 2) + (groups == 1 ? 1 : 0);
}

int ssn_gen_supported(int M, int dtype_bytes) {
    if (M <= 0 || (M & 1)) return 0;
    return dtype_bytes == 8 ? ssn::gen_supported<double>(M) : ssn::gen_supported<float>(M);
}
int ssn_gen_forward_f32(const float* W, const float* ext, float* time_avg, float* dyn_row, float* rate_row, float* traj,
                        float* df, int B, int NB, int M, const ssn_gen_params* p, void* stream) {
    return gen_forward_impl<float>(W, ext, time_avg, dyn_row, rate_row, traj, df, B, NB, M, p, stream);
}
int ssn_gen_forward_f64(const double* W, const double* ext, double* time_avg, double* dyn_row, double* rate_row,
                        double* traj, double* df, int B, int NB, int M, const ssn_gen_params* p, void* stream) {
    return gen_forward_impl<double>(W, ext, time_avg, dyn_row, rate_row, traj, df, B, NB, M, p, stream);
}
int ssn_gen_backward_f32(const float* W, const float* traj, float* df_delta, const float* g_time_avg, double c_dyn,
                         double c_rate, int B, int NB, int M, const ssn_gen_params* p, void* stream) {
    return gen_backward_impl<float>(W, traj, df_delta, g_time_avg, nullptr, c_dyn, c_rate, B, NB, M, p, stream);
}
int ssn_gen_backward_f64(const double* W, const double* traj, double* df_delta, const double* g_time_avg, double c_dyn,
                         double c_rate, int B, int NB, int M, const ssn_gen_params* p, void* stream) {
    return gen_backward_impl<double>(W, traj, df_delta, g_time_avg, nullptr, c_dyn, c_rate, B, NB, M, p, stream);
}
int ssn_gen_backward_ext_f32(const float* W, const float* traj, float* df_delta, const float* g_time_avg, float* g_ext,
                             double c_dyn, double c_rate, int B, int NB, int M, const ssn_gen_params* p, void* stream) {
    return gen_backward_impl<float>(W, traj, df_delta, g_time_avg, g_ext, c_dyn, c_rate, B, NB, M, p, stream);
}
int ssn_gen_backward_ext_f64(const double* W, const double* traj, double* df_delta, const double* g_time_avg,
                             double* g_ext, double c_dyn, double c_rate, int B, int NB, int M, const ssn_gen_params* p,
                             void* stream) {
    return gen_backward_impl<double>(W, traj, df_delta, g_time_avg, g_ext, c_dyn, c_rate, B, NB, M, p, stream);
}
int ssn_gen_backward_max_f32(const float* W, const float* traj, float* df_delta, const float* g_time_avg, float* g_ext,
                             float* dmax, int* tracked, double c_dyn, double c_rate, int B, int NB, int M,
                             const ssn_gen_params* p, void* stream) {
    return gen_backward_impl<float>(W, traj, df_delta, g_time_avg, g_ext, c_dyn, c_rate, B, NB, M, p, stream, dmax, tracked);
}
int ssn_gen_backward_fused_supported(int B, int NB, int M, const ssn_gen_params* p, float xmax) {
    if (!p) return 0;
    ssn::GenBwdArgs<float> a;
    a.B = B; a.NB = NB; a.M = M; a.seqlen = p->seqlen; a.skip = p->skip_steps;
    return p->skip_steps >= 0 && p->skip_steps < p->seqlen && (long)NB * p->seqlen * M < (1L << 29) &&
           ssn::gen_backward_fused_supported(a, xmax);
}
int ssn_gen_backward_fused_f32(const float* W, const float* traj, const float* df, const float* g_time_avg, float* g_ext,
                               float* gW, float* dmax, float xmax, double c_dyn, double c_rate, int B, int NB, int M,
                               const ssn_gen_params* p, void* stream) {
    if (B == 0 || NB == 0) return 0;
    if (!p || !W || !traj || !df || !g_time_avg || !gW || !ssn_gen_backward_fused_supported(B, NB, M, p, xmax)) {
        g_last_error = "ssn_gen_backward_fused: invalid argument or unsupported size (fp32, NB <= 8, even 2N <= 208, "
                       "NB T 2N < 2^29, 0 < xmax < inf)";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    ssn::GenBwdArgs<float> a;
    a.W = W; a.traj = traj; a.delta = const_cast<float*>(df); a.g_time_avg = g_time_avg; a.g_ext = g_ext;
    a.B = B; a.NB = NB; a.M = M; a.seqlen = p->seqlen; a.skip = p->skip_steps;
    a.eps_E = (float)(p->dt / p->tau_E); a.eps_I = (float)(p->dt / p->tau_I); a.theta = (float)p->rate_penalty_threshold;
    a.c_dyn = (float)c_dyn; a.c_rate = (float)c_rate;
    if (dmax) {
        SSN_TRY(hipMemsetAsync(dmax, 0, sizeof(float) * (size_t)B, (hipStream_t)stream));
        a.dmax = reinterpret_cast<unsigned*>(dmax);
    }
    SSN_TRY(ssn::launch_gen_backward_fused(a, gW, xmax, (hipStream_t)stream));
    return 0;
}
int ssn_weight_grad_scaled_f32(const float* delta, const float* traj, float* gW, int B, long K, int M, const float* dmax,
                               float xmax, void* stream) {
    if (!delta || !traj || !gW || !dmax || B < 0 || K < 0 || M <= 0 || M > 224 || !(xmax > 0.f) || !(xmax < __builtin_inff()) ||
        K * (long)M * 4 >= (1L << 31)) {
        g_last_error = "ssn_weight_grad_scaled: invalid argument (fp32, M <= 224, K M < 2^29, dmax on the device, 0 < xmax < inf)";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    SSN_TRY(ssn::launch_weight_grad_scaled(delta, traj, gW, B, K, M, reinterpret_cast<const unsigned*>(dmax), xmax,
                                           (hipStream_t)stream));
    return 0;
}
int ssn_weight_grad_f32(const float* delta, const float* traj, float* gW, int B, long K, int M, int kernel, void* stream) {
    if (B < 0 || K < 0 || M < 0 || (B > 0 && M > 0 && (!gW || (K > 0 && (!delta || !traj))))) {
        g_last_error = "ssn_weight_grad: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    SSN_TRY(ssn::launch_weight_grad<float>(delta, traj, gW, B, K, M, kernel, (hipStream_t)stream));
    return 0;
}
int ssn_weight_grad_f64(const double* delta, const double* traj, double* gW, int B, long K, int M, int kernel, void* stream) {
    if (B < 0 || K < 0 || M < 0 || (B > 0 && M > 0 && (!gW || (K > 0 && (!delta || !traj))))) {
        g_last_error = "ssn_weight_grad: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    SSN_TRY(ssn::launch_weight_grad<double>(delta, traj, gW, B, K, M, kernel, (hipStream_t)stream));
    return 0;
}
int ssn_jds_grad_f32(const float* gW, const float* z, const float* J, const float* D, const float* S, double* out,
                     int B, int N, void* stream) {
    float jds[12];
    for (int q = 0; q < 4; ++q) { jds[q] = J[q]; jds[4 + q] = D[q]; jds[8 + q] = S[q]; }
    SSN_TRY(ssn::launch_jds_grad<float>(gW, z, jds, out, B, N, (hipStream_t)stream));
    return 0;
}
int ssn_jds_grad_f64(const double* gW, const double* z, const double* J, const double* D, const double* S, double* out,
                     int B, int N, void* stream) {
    double jds[12];
    for (int q = 0; q < 4; ++q) { jds[q] = J[q]; jds[4 + q] = D[q]; jds[8 + q] = S[q]; }
    SSN_TRY(ssn::launch_jds_grad<double>(gW, z, jds, out, B, N, (hipStream_t)stream));
    return 0;
}

int ssn_abi_version(void) { return 1; }

int ssn_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { fail(e, "hipGetDeviceCount"); return -(int)e; }
    return n;
}

const char* ssn_last_error(void) { return g_last_error.c_str(); }

int ssn_solver_fast_path(int M, int NB, int dtype_bytes) {
    if (M <= 0 || (M & 1)) return 0;
    const bool tile = dtype_bytes == 8 ? ssn::tile_supported<double>(M, NB) : ssn::tile_supported<float>(M, NB);
    const bool regw = dtype_bytes == 8 ? ssn::regw_supported<double>(M, NB) : ssn::regw_supported<float>(M, NB);
    return tile ? 2 : (regw ? 1 : 0);
}

int ssn_solve_batch_f32(const float* W, const float* ext, int ext_per_draw, float* r, float* r_prev, int* codes,
                        int* steps, int B, int NB, int M, const ssn_solver_params* p, void* stream) {
    return solve_batch_impl<float>(-1, W, ext, ext_per_draw, r, r_prev, codes, steps, B, NB, M, p, stream);
}
int ssn_solve_batch_f64(const double* W, const double* ext, int ext_per_draw, double* r, double* r_prev,
                        int* codes, int* steps, int B, int NB, int M, const ssn_solver_params* p, void* stream) {
    return solve_batch_impl<double>(-1, W, ext, ext_per_draw, r, r_prev, codes, steps, B, NB, M, p, stream);
}
int ssn_solve_batch_f32_variant(int variant, const float* W, const float* ext, int ext_per_draw, float* r,
                                float* r_prev, int* codes, int* steps, int B, int NB, int M,
                                const ssn_solver_params* p, void* stream) {
    return solve_batch_impl<float>(variant, W, ext, ext_per_draw, r, r_prev, codes, steps, B, NB, M, p, stream);
}
int ssn_solve_batch_f64_variant(int variant, const double* W, const double* ext, int ext_per_draw, double* r,
                                double* r_prev, int* codes, int* steps, int B, int NB, int M,
                                const ssn_solver_params* p, void* stream) {
    return solve_batch_impl<double>(variant, W, ext, ext_per_draw, r, r_prev, codes, steps, B, NB, M, p, stream);
}
int ssn_solve_batch_host_f32(const float* W, const float* ext, int ext_per_draw, float* r, float* r_prev,
                             int* codes, int* steps, int B, int NB, int M, const ssn_solver_params* p) {
    return solve_batch_host_impl<float>(W, ext, ext_per_draw, r, r_prev, codes, steps, B, NB, M, p);
}
int ssn_solve_batch_host_f64(const double* W, const double* ext, int ext_per_draw, double* r, double* r_prev,
                             int* codes, int* steps, int B, int NB, int M, const ssn_solver_params* p) {
    return solve_batch_host_impl<double>(W, ext, ext_per_draw, r, r_prev, codes, steps, B, NB, M, p);
}

int ssn_build_w_f32(const float* z, const float* J, const float* D, const float* S, float* W, int B, int N, void* stream) {
    float jds[12];
    for (int q = 0; q < 4; ++q) { jds[q] = J[q]; jds[4 + q] = D[q]; jds[8 + q] = S[q]; }
    SSN_TRY(ssn::launch_build_w<float>(z, jds, W, B, N, (hipStream_t)stream));
    return 0;
}
int ssn_build_w_devparams_f32(const float* z, const float* jds_dev, float* W, int B, int N, void* stream) {
    if (!jds_dev || (B > 0 && (!z || !W)) || N < 1 || B < 0) { g_last_error = "ssn_build_w_devparams_f32: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_build_w<float>(z, nullptr, W, B, N, (hipStream_t)stream, jds_dev));
    return 0;
}
int ssn_build_w_f64(const double* z, const double* J, const double* D, const double* S, double* W, int B, int N, void* stream) {
    double jds[12];
    for (int q = 0; q < 4; ++q) { jds[q] = J[q]; jds[4 + q] = D[q]; jds[8 + q] = S[q]; }
    SSN_TRY(ssn::launch_build_w<double>(z, jds, W, B, N, (hipStream_t)stream));
    return 0;
}
int ssn_build_w_philox_f32(unsigned long long seed, unsigned long long offset, const float* J, const float* D, const float* S,
                           float* W, float* z, int B, int N, void* stream) {
    float jds[12];
    for (int q = 0; q < 4; ++q) { jds[q] = J[q]; jds[4 + q] = D[q]; jds[8 + q] = S[q]; }
    SSN_TRY(ssn::launch_build_w_philox<float>(seed, offset, jds, W, z, B, N, (hipStream_t)stream));
    return 0;
}
int ssn_build_w_philox_f64(unsigned long long seed, unsigned long long offset, const double* J, const double* D, const double* S,
                           double* W, double* z, int B, int N, void* stream) {
    double jds[12];
    for (int q = 0; q < 4; ++q) { jds[q] = J[q]; jds[4 + q] = D[q]; jds[8 + q] = S[q]; }
    SSN_TRY(ssn::launch_build_w_philox<double>(seed, offset, jds, W, z, B, N, (hipStream_t)stream));
    return 0;
}
int ssn_mt19937_jump_poly(unsigned long long nblocks, unsigned long long* bits) {
    if (!bits) { g_last_error = "ssn_mt19937_jump_poly: null output"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    if (ssn::mt19937_jump_poly(nblocks, bits) != 0) {
        g_last_error = "ssn_mt19937_jump_poly: the characteristic polynomial did not come out with degree 19937";
        return SSN_ERR_BASE + (int)hipErrorUnknown;
    }
    return 0;
}
int ssn_mt19937_random_sample_f32(unsigned int* key, int* pos, unsigned long long total, unsigned long long skip,
                                  unsigned long long count, float* out, void* stream) {
    if (!key || !pos) { g_last_error = "ssn_mt19937_random_sample: null state"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::mt19937_draw(key, pos, total, skip, count, out, 4, (hipStream_t)stream));
    return 0;
}
int ssn_mt19937_random_sample_f64(unsigned int* key, int* pos, unsigned long long total, unsigned long long skip,
                                  unsigned long long count, double* out, void* stream) {
    if (!key || !pos) { g_last_error = "ssn_mt19937_random_sample: null state"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::mt19937_draw(key, pos, total, skip, count, out, 8, (hipStream_t)stream));
    return 0;
}
int ssn_mt19937_random_sample_begin_f32(const unsigned int* key, int pos, unsigned long long total, unsigned long long skip,
                                        unsigned long long count, float* out, void* stream, int* ticket) {
    if (!key || !ticket) { g_last_error = "ssn_mt19937_random_sample_begin: null state / ticket"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    *ticket = -1;
    SSN_TRY(ssn::mt19937_begin(key, pos, total, skip, count, out, 4, (hipStream_t)stream, ticket));
    return 0;
}
int ssn_mt19937_random_sample_begin_f64(const unsigned int* key, int pos, unsigned long long total, unsigned long long skip,
                                        unsigned long long count, double* out, void* stream, int* ticket) {
    if (!key || !ticket) { g_last_error = "ssn_mt19937_random_sample_begin: null state / ticket"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    *ticket = -1;
    SSN_TRY(ssn::mt19937_begin(key, pos, total, skip, count, out, 8, (hipStream_t)stream, ticket));
    return 0;
}
int ssn_build_w_mt19937_begin_f32(const unsigned int* key, int pos, int B_total, int b0, int nb, const float* J, const float* D,
                                  const float* S, float* W, float* z, int N, void* stream, int* ticket) {
    if (!key || !ticket || !J || !D || !S || !W || N < 1 || B_total < 0 || b0 < 0 || nb < 0 || b0 + nb > B_total) {
        g_last_error = "ssn_build_w_mt19937_begin: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    *ticket = -1;
    float jds[12];
    for (int q = 0; q < 4; ++q) { jds[q] = J[q]; jds[4 + q] = D[q]; jds[8 + q] = S[q]; }
    const unsigned long long mm = 4ull * N * N;
    SSN_TRY(ssn::mt19937_begin(key, pos, mm * B_total, mm * b0, mm * nb, z, 4, (hipStream_t)stream, ticket, nb ? W : nullptr, jds, N));
    return 0;
}
int ssn_mt19937_random_sample_tail_begin_f32(const unsigned int* key, int pos, unsigned long long total, unsigned long long skip,
                                             unsigned long long count, float* out, int tail_kind, unsigned long long tail_total,
                                             unsigned long long tail_skip, unsigned long long tail_count, float* tail_out, void* stream,
                                             int* ticket) {
    if (!key || !ticket) { g_last_error = "ssn_mt19937_random_sample_tail_begin: null state / ticket"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    *ticket = -1;
    if (tail_kind != 1 && tail_kind != 2) { g_last_error = "ssn_mt19937_random_sample_tail_begin: tail_kind must be 1 (choice(2, n) * 2 - 1) or 2 (rand(n) * 2 - 1)"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    ssn::MtTail tail;
    tail.kind = tail_kind; tail.total = tail_total; tail.skip = tail_skip; tail.count = tail_count; tail.out = tail_out;
    SSN_TRY(ssn::mt19937_begin(key, pos, total, skip, count, out, 4, (hipStream_t)stream, ticket, nullptr, nullptr, 0, &tail));
    return 0;
}
int ssn_build_w_mt19937_tail_begin_f32(const unsigned int* key, int pos, int B_total, int b0, int nb, const float* J, const float* D,
                                       const float* S, float* W, float* z, int N, int tail_kind, float* zin, void* stream, int* ticket) {
    if (!key || !ticket || !J || !D || !S || (nb && (!W || !zin)) || N < 1 || B_total < 0 || b0 < 0 || nb < 0 || b0 + nb > B_total
        || (tail_kind != 1 && tail_kind != 2)) {
        g_last_error = "ssn_build_w_mt19937_tail_begin: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    *ticket = -1;
    float jds[12];
    for (int q = 0; q < 4; ++q) { jds[q] = J[q]; jds[4 + q] = D[q]; jds[8 + q] = S[q]; }
    const unsigned long long mm = 4ull * N * N, m = 2ull * N;
    ssn::MtTail tail;
    tail.kind = tail_kind; tail.total = m * B_total; tail.skip = m * b0; tail.count = m * nb; tail.out = zin;
    SSN_TRY(ssn::mt19937_begin(key, pos, mm * B_total, mm * b0, mm * nb, z, 4, (hipStream_t)stream, ticket, nb ? W : nullptr, jds, N, &tail));
    return 0;
}
int ssn_mt19937_plan(int pos, unsigned long long total, unsigned long long skip, unsigned long long count, long* out) {
    if (!out || !ssn::mt19937_plan(pos, total, skip, count, out)) { g_last_error = "ssn_mt19937_plan: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    return 0;
}
int ssn_mt19937_plan_tail(int pos, unsigned long long total, unsigned long long skip, unsigned long long count, int tail_kind,
                          unsigned long long tail_total, unsigned long long tail_skip, unsigned long long tail_count, long* out) {
    ssn::MtTail tail;
    tail.kind = tail_kind; tail.total = tail_total; tail.skip = tail_skip; tail.count = tail_count;
    if (!out || !ssn::mt19937_plan(pos, total, skip, count, out, &tail)) { g_last_error = "ssn_mt19937_plan_tail: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    return 0;
}
int ssn_mt19937_random_sample_finish(int ticket, unsigned int* key, int* pos) {
    if (!key || !pos) { g_last_error = "ssn_mt19937_random_sample_finish: null state"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::mt19937_finish(ticket, key, pos));
    return 0;
}
int ssn_gen_inputs_philox_f32(const ssn_gen_inputs* a, void* stream) {
    if (!a || !a->J || !a->D || !a->S || !a->bw || !a->con || !a->W || !a->ext || a->B < 0 || a->NB < 0 || a->N <= 0 ||
        (a->v && (!a->zin || !a->amp))) {
        g_last_error = "ssn_gen_inputs_philox: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    // one launch (ssn_aux.hip: gen_inputs_kernel): the numbers of ssn_philox_amp_f32, ssn_stimulus_amp_f32 and
    // ssn_build_w_philox_f32 called in that order
    ssn::GenInputsArgs g{};
    g.seed = a->seed; g.off_z = a->off_z; g.off_zin = a->off_zin;
    g.bw = a->bw; g.con = a->con; g.v = a->v; g.inv_l = 1.f / a->smoothness; g.bernoulli = a->bernoulli;
    g.W = a->W; g.z = a->z; g.zin = a->zin; g.amp = a->amp; g.ext = a->ext;
    g.B = a->B; g.NB = a->NB; g.N = a->N;
    const float jds12[12] = {a->J[0], a->J[1], a->J[2], a->J[3], a->D[0], a->D[1], a->D[2], a->D[3], a->S[0], a->S[1], a->S[2], a->S[3]};
    SSN_TRY(ssn::launch_gen_inputs(g, jds12, (hipStream_t)stream));
    return 0;
}
int ssn_stimulus_f32(const float* bw, const float* con, float smoothness, float* ext, int B, int NB, int N, void* stream) {
    SSN_TRY(ssn::launch_stimulus<float>(bw, con, smoothness, nullptr, ext, B, NB, N, (hipStream_t)stream));
    return 0;
}
int ssn_stimulus_f64(const double* bw, const double* con, double smoothness, double* ext, int B, int NB, int N, void* stream) {
    SSN_TRY(ssn::launch_stimulus<double>(bw, con, smoothness, nullptr, ext, B, NB, N, (hipStream_t)stream));
    return 0;
}
int ssn_stimulus_amp_f32(const float* bw, const float* con, float smoothness, const float* amp, float* ext, int B, int NB,
                         int N, void* stream) {
    SSN_TRY(ssn::launch_stimulus<float>(bw, con, smoothness, amp, ext, B, NB, N, (hipStream_t)stream));
    return 0;
}
int ssn_stimulus_hetero_f32(const float* bw, const float* con, float smoothness, const float* zin, const float* v, int nv, float* ext,
                            int B, int NB, int N, void* stream) {
    if (!zin || !v || (nv != 1 && nv != 2 && nv != 2 * N) || B < 0 || NB < 0 || N < 1) {
        g_last_error = "ssn_stimulus_hetero: invalid argument";
        return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    SSN_TRY(ssn::launch_stimulus_hetero(bw, con, smoothness, zin, v, nv, ext, B, NB, N, (hipStream_t)stream));
    return 0;
}
int ssn_stimulus_amp_f64(const double* bw, const double* con, double smoothness, const double* amp, double* ext, int B,
                         int NB, int N, void* stream) {
    SSN_TRY(ssn::launch_stimulus<double>(bw, con, smoothness, amp, ext, B, NB, N, (hipStream_t)stream));
    return 0;
}
int ssn_probe_scatter_f32(const float* g, const long* ids, const long* probes, float* g_ta, int n, int B, int NB, int M, void* stream) {
    if (n < 0 || B < 0 || NB < 0 || M < 0 || ((long)B * NB * M > 0 && !g_ta) || (n > 0 && (!g || !ids || !probes))) { g_last_error = "ssn_probe_scatter: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_probe_scatter<float>(g, ids, probes, g_ta, n, B, NB, M, (hipStream_t)stream));
    return 0;
}
int ssn_probe_scatter_f64(const double* g, const long* ids, const long* probes, double* g_ta, int n, int B, int NB, int M, void* stream) {
    if (n < 0 || B < 0 || NB < 0 || M < 0 || ((long)B * NB * M > 0 && !g_ta) || (n > 0 && (!g || !ids || !probes))) { g_last_error = "ssn_probe_scatter: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_probe_scatter<double>(g, ids, probes, g_ta, n, B, NB, M, (hipStream_t)stream));
    return 0;
}
long ssn_segment_sqnorms_ws_doubles(int n) { return ssn::segment_sqnorms_ws_doubles(n); }
int ssn_segment_sqnorms2_f32(const float* x, const long* bounds, int n, float* out, double* ws, void* stream) {
    if (n < 0 || (n > 0 && (!x || !bounds || !out || !ws))) { g_last_error = "ssn_segment_sqnorms2: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_segment_sqnorms(x, bounds, n, out, ws, (hipStream_t)stream));
    return 0;
}
// The round-2 signature (no scratch argument), kept under its name so that a caller built against the older header still
// gets what it asked for: the scratch comes from the stream-ordered allocator and goes back behind the two launches.
int ssn_segment_sqnorms_f32(const float* x, const long* bounds, int n, float* out, void* stream) {
    if (n < 0 || (n > 0 && (!x || !bounds || !out))) { g_last_error = "ssn_segment_sqnorms: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    if (n == 0) return 0;
    double* ws = nullptr;
    SSN_TRY(hipMallocAsync((void**)&ws, sizeof(double) * (size_t)ssn::segment_sqnorms_ws_doubles(n), (hipStream_t)stream));
    const hipError_t e = ssn::launch_segment_sqnorms(x, bounds, n, out, ws, (hipStream_t)stream);
    const hipError_t f = hipFreeAsync(ws, (hipStream_t)stream);
    SSN_TRY(e);
    SSN_TRY(f);
    return 0;
}
int ssn_interpolate_f32(const float* eps, const float* xd, const float* xg, float* xp, int rows, int cols, void* stream) {
    if (rows < 0 || cols < 0 || ((long)rows * cols > 0 && (!eps || !xd || !xg || !xp))) { g_last_error = "ssn_interpolate: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_interpolate(eps, xd, xg, xp, rows, cols, (hipStream_t)stream));
    return 0;
}
int ssn_penalty_means_f32(const float* dyn, const float* rate, long n, double scale_dyn, double scale_rate, double* ws, double* out, void* stream) {
    if (n < 0 || !ws || !out || (n > 0 && (!dyn || !rate))) { g_last_error = "ssn_penalty_means: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_penalty_means<float>(dyn, rate, n, scale_dyn, scale_rate, ws, out, (hipStream_t)stream));
    return 0;
}
int ssn_penalty_means_f64(const double* dyn, const double* rate, long n, double scale_dyn, double scale_rate, double* ws, double* out, void* stream) {
    if (n < 0 || !ws || !out || (n > 0 && (!dyn || !rate))) { g_last_error = "ssn_penalty_means: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_penalty_means<double>(dyn, rate, n, scale_dyn, scale_rate, ws, out, (hipStream_t)stream));
    return 0;
}
int ssn_penalty_means_probe_f32(const float* dyn, const float* rate, long n, double scale_dyn, double scale_rate, double* ws, double* out,
                                const float* time_avg, const long* ids, const long* probes, float* tc, int nsamp, int NB, int M, void* stream) {
    if (n < 0 || !ws || !out || (n > 0 && (!dyn || !rate)) || nsamp < 0 || NB < 0 || M < 0 || (nsamp > 0 && (!time_avg || !ids || !probes || !tc))) {
        g_last_error = "ssn_penalty_means_probe: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    SSN_TRY(ssn::launch_penalty_means<float>(dyn, rate, n, scale_dyn, scale_rate, ws, out, (hipStream_t)stream, time_avg, ids, probes,
                                             nsamp > 0 ? tc : nullptr, nsamp, NB, M));
    return 0;
}
int ssn_penalty_means_probe_f64(const double* dyn, const double* rate, long n, double scale_dyn, double scale_rate, double* ws, double* out,
                                const double* time_avg, const long* ids, const long* probes, double* tc, int nsamp, int NB, int M, void* stream) {
    if (n < 0 || !ws || !out || (n > 0 && (!dyn || !rate)) || nsamp < 0 || NB < 0 || M < 0 || (nsamp > 0 && (!time_avg || !ids || !probes || !tc))) {
        g_last_error = "ssn_penalty_means_probe: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue;
    }
    SSN_TRY(ssn::launch_penalty_means<double>(dyn, rate, n, scale_dyn, scale_rate, ws, out, (hipStream_t)stream, time_avg, ids, probes,
                                              nsamp > 0 ? tc : nullptr, nsamp, NB, M));
    return 0;
}
int ssn_philox_amp_f32(unsigned long long seed, unsigned long long offset, const float* v, float* zin, float* amp, unsigned long long n, int M, int bernoulli, void* stream) {
    if (n > 0 && (!v || !zin || !amp || M <= 0)) { g_last_error = "ssn_philox_amp: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_philox_amp<float>(seed, offset, v, zin, amp, n, M, bernoulli, (hipStream_t)stream));
    return 0;
}
int ssn_philox_amp_f64(unsigned long long seed, unsigned long long offset, const double* v, double* zin, double* amp, unsigned long long n, int M, int bernoulli, void* stream) {
    if (n > 0 && (!v || !zin || !amp || M <= 0)) { g_last_error = "ssn_philox_amp: invalid argument"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_philox_amp<double>(seed, offset, v, zin, amp, n, M, bernoulli, (hipStream_t)stream));
    return 0;
}
int ssn_philox_uniform_f32(unsigned long long seed, unsigned long long offset, float* out, unsigned long long n, void* stream) {
    if (n > 0 && !out) { g_last_error = "ssn_philox_uniform: null output"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_philox_uniform<float>(seed, offset, out, n, (hipStream_t)stream));
    return 0;
}
int ssn_philox_uniform_f64(unsigned long long seed, unsigned long long offset, double* out, unsigned long long n, void* stream) {
    if (n > 0 && !out) { g_last_error = "ssn_philox_uniform: null output"; return SSN_ERR_BASE + (int)hipErrorInvalidValue; }
    SSN_TRY(ssn::launch_philox_uniform<double>(seed, offset, out, n, (hipStream_t)stream));
    return 0;
}
int ssn_io_eval_f32(const float* v, float* out, long count, const ssn_solver_params* p, void* stream) {
    SSN_TRY(ssn::launch_io_eval<float>(v, out, count, ssn::make_io_consts<float>(*p), (hipStream_t)stream));
    return 0;
}
int ssn_io_eval_f64(const double* v, double* out, long count, const ssn_solver_params* p, void* stream) {
    SSN_TRY(ssn::launch_io_eval<double>(v, out, count, ssn::make_io_consts<double>(*p), (hipStream_t)stream));
    return 0;
}

// ---- drop-in symbols (tc_gan/ext/ssnode.c exports) ------------------------------
int solve_dynamics_asym_power_euler(int N, double* W, double* ext, double k, double n, double* r0, double* r1,
                                    double tau_E, double tau_I, double dt, int max_iter, double atol,
                                    double rate_soft_bound, double rate_hard_bound) {
    return legacy_solve(SSN_IO_POWER, N, W, ext, k, n, r0, r1, tau_E, tau_I, dt, max_iter, atol, rate_soft_bound, rate_hard_bound);
}
int solve_dynamics_asym_linear_euler(int N, double* W, double* ext, double k, double n, double* r0, double* r1,
                                     double tau_E, double tau_I, double dt, int max_iter, double atol,
                                     double rate_soft_bound, double rate_hard_bound) {
    return legacy_solve(SSN_IO_LINEAR, N, W, ext, k, n, r0, r1, tau_E, tau_I, dt, max_iter, atol, rate_soft_bound, rate_hard_bound);
}
int solve_dynamics_asym_tanh_euler(int N, double* W, double* ext, double k, double n, double* r0, double* r1,
                                   double tau_E, double tau_I, double dt, int max_iter, double atol,
                                   double rate_soft_bound, double rate_hard_bound) {
    return legacy_solve(SSN_IO_TANH, N, W, ext, k, n, r0, r1, tau_E, tau_I, dt, max_iter, atol, rate_soft_bound, rate_hard_bound);
}

double io_pow(double v, double r0, double r1, double v0, double k, double n) { return legacy_io(SSN_IO_POWER, v, r0, r1, v0, k, n); }
double io_alin(double v, double r0, double r1, double v0, double k, double n) { return legacy_io(SSN_IO_LINEAR, v, r0, r1, v0, k, n); }
double io_atanh(double v, double r0, double r1, double v0, double k, double n) { return legacy_io(SSN_IO_TANH, v, r0, r1, v0, k, n); }

double rate_to_volt(double rate, double k, double n) {
    // (rate/k)^(1/n) == the v at which io_pow(v) = rate: evaluate on the device through the same
    // pow routine the kernels use:  k' * x^n' with k' = 1, x = rate/k, n' = 1/n.
    ssn_solver_params p;
    std::memset(&p, 0, sizeof(p));
    p.io_type = SSN_IO_POWER; p.k = 1.0; p.n = 1.0 / n; p.rate_soft_bound = 1.0; p.rate_hard_bound = 2.0;
    const double x = rate / k;
    double out = std::numeric_limits<double>::quiet_NaN();
    const ssn::IoConsts<double> c = ssn::make_io_consts<double>(p);
    scalar_roundtrip(&x, 1, &out, 1, [&](double* dv, double* dout, hipStream_t st) {
        return ssn::launch_io_eval<double>(dv, dout, 1, c, st);
    });
    return out;
}

double dot(int dim, const double* x, const double* y) {
    if (dim <= 0) return 0.0;
    double out = std::numeric_limits<double>::quiet_NaN();
    std::vector<double> xy(2 * (size_t)dim);
    std::memcpy(xy.data(), x, dim * sizeof(double));
    std::memcpy(xy.data() + dim, y, dim * sizeof(double));
    scalar_roundtrip(xy.data(), xy.size(), &out, 1, [&](double* d, double* dout, hipStream_t st) {
        return ssn::launch_dot<double>(d, d + dim, dout, dim, st);
    });
    return out;
}

}  // extern "C"
