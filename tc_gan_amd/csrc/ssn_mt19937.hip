// numpy's RandomState.random_sample on the device, bit for bit, from the caller's (key, pos) -- the reference's noise stream
// `zs = rng.rand(batchsize, 2N, 2N)` (tc_gan/networks/ssn.py:434-439; order of the stream networks/cwgan.py:438-481; fp64 ->
// floatX downcast utils/theanoutils.py:9-16) without the host drawing it (0.25 s per 1024 x 200 x 200 draw against a 2.8 ms
// forward).  Integer work: the bar is bit equality with numpy (tests/test_mt19937_gpu.py), and the state handed back to the
// host RandomState is numpy's own after the same draw.
//
// MT19937 is one sequential recurrence (x[k+624] = x[k+397] ^ twist(x[k], x[k+1])).  The stream is cut into SEGMENTS of
// kSegBlocks = 256 blocks of 624 words; segment s starts from the state 256 s blocks ahead of the caller's, which is reached by
// jump-ahead polynomials (ssn_mt19937_poly.h): a ladder of levels with strides 4, 256, 16384, 2^20 blocks, 63 polynomials per
// level (digit d = 1..63 times the stride), so that any state is at most one jump per level away and all states of a level
// are computed by ONE launch from the states of the level above.
//
//   mt_jump_kernel   one workgroup per (state, share of the polynomial's taps): regenerates the 20560 words the taps can
//                    reach into LDS, accumulates XOR of x[tap + j] for its share; consumers XOR the F shares when they load
//                    the state.  LDS-read bound (624 words per tap).
//   mt_gen_kernel    one wave per segment: in-place regeneration in LDS (2.5 KB per wave), 104 words per pass (52 lanes x 2
//                    words: a pass depends on no pass nearer than two back), tempering, the (a >> 5, b >> 6) -> double of
//                    randomkit's rk_double, rounding to fp32 where asked (round to nearest even = numpy's astype), coalesced
//                    stores of the [skip, skip + count) window only (the rows of one rank of a data-parallel job).
//   mt_final_kernel  the state after the whole draw (every rank needs it, whatever rows it generates): a chain of at most
//                    four single-state jumps + at most four regenerations on a side stream of the library's own; the host
//                    waits for this chain only, never for the caller's stream.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>
#include "ssn_host.h"
#include "ssn_mt19937_poly.h"

namespace ssn {
namespace mt {

constexpr int kLevels = 4;                 // strides 4 * 64^l blocks
constexpr int kStride0Log2 = 2;
constexpr int kRadixLog2 = 6;
constexpr int kSegLevel = 1;               // segments are the states of level 1
constexpr int kSegBlocks = 1 << (kStride0Log2 + kRadixLog2 * kSegLevel);   // 256
constexpr int kTapCap = 19968;             // taps of one polynomial (uint16), padded
constexpr int kXLen = 20704;               // words of the regenerated sequence a jump workgroup keeps (19936 + 255 + 512 + 1)
constexpr int kMaxShare = 16;

struct Key { uint32_t w[kN]; };

__device__ __forceinline__ uint32_t twist_d(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t temper_d(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// ---- jump ------------------------------------------------------------------------------------------------------------
struct JumpArgs {
    Key root;                   // the caller's key: the parent of the top level
    const uint32_t* parent;     // [..][pF][624] shares of the states one level up, or nullptr: root
    long parent_lo;             // index (in units of the parent level's stride) of parent[0]
    int pF;
    uint32_t* dst;              // [count][F][624]
    long lo;                    // index (in units of this level's stride) of dst[0]
    int F;
    const uint32_t* taps;       // this level: [63][kTapCap / 2] dwords, two taps each, ascending
    const int* ntaps;           // [63]
};

__global__ __launch_bounds__(256) void mt_jump_kernel(const JumpArgs a) {
    extern __shared__ __align__(16) uint32_t x[];
    const int tid = threadIdx.x;
    const long n = a.lo + (long)blockIdx.x;
    const int f = blockIdx.y;
    const int d = (int)(n & 63);
    uint32_t* out = a.dst + ((size_t)blockIdx.x * a.F + f) * kN;
    const uint32_t* src = a.parent ? a.parent + (size_t)((n >> kRadixLog2) - a.parent_lo) * a.pF * kN : nullptr;
    for (int j = tid; j < kN; j += 256) {
        uint32_t v = 0;
        if (src) { for (int s = 0; s < a.pF; ++s) v ^= src[(size_t)s * kN + j]; }
        else v = a.root.w[j];
        x[j] = v;
    }
    if (d == 0) {            // the parent's own state: share 0 carries it (exact, low bits of word 0 included), the others nothing
        __syncthreads();
        for (int j = tid; j < kN; j += 256) out[j] = f == 0 ? x[j] : 0u;
        return;
    }
    const uint32_t* tp = a.taps + (size_t)(d - 1) * (kTapCap / 2);
    const int nt = a.ntaps[d - 1];
    const int e0 = (int)((long)nt * f / a.F), e1 = (int)((long)nt * (f + 1) / a.F);
    const int last = e1 > e0 ? (int)((tp[(e1 - 1) >> 1] >> (((e1 - 1) & 1) * 16)) & 0xffffu) : -1;
    // x[624 .. last + 624): 227 independent words per step
    const int need = last + kN;                                  // words [0, need) are read by the valid outputs
    __syncthreads();
    for (int k0 = 0; k0 + kN < need; k0 += kN - kM) {
        const int k = k0 + tid;
        if (tid < kN - kM && k + kN < need) x[k + kN] = x[k + kM] ^ twist_d(x[k], x[k + 1]);
        __syncthreads();
    }
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0;
    const uint32_t* xt = x + tid;
    for (int e = e0; e < e1; ++e) {
        const int t = (int)((tp[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        acc0 ^= xt[t];
        acc1 ^= xt[t + 256];
        acc2 ^= xt[t + 512];            // (tid >= 112: beyond the state, never stored; inside the allocation)
    }
    out[tid] = acc0;
    out[tid + 256] = acc1;
    if (tid + 512 < kN) out[tid + 512] = acc2;
}

// ---- the state after the draw ----------------------------------------------------------------------------------------
// One wave: loads the state (XOR of shares) `steps` - 1 blocks before the wanted one, regenerates `steps` >= 1 times (the last
// regeneration makes every bit of the key numpy's, the low bits of word 0 included), writes the key.
__global__ __launch_bounds__(64) void mt_final_kernel(const uint32_t* state, int F, int steps, uint32_t* out) {
    __shared__ __align__(16) uint32_t key[kN + 8];
    volatile uint32_t* k = key;
    const int lane = threadIdx.x;
    for (int j = lane; j < kN; j += 64) {
        uint32_t v = 0;
        for (int s = 0; s < F; ++s) v ^= state[(size_t)s * kN + j];
        k[j] = v;
    }
    for (int b = 0; b < steps; ++b) {
        for (int c = 0; c < 6; ++c) {
            if (lane < 52) {
                const int i = 104 * c + 2 * lane;
                const uint32_t a0 = k[i], a1 = k[i + 1], a2 = k[i + 2 == kN ? 0 : i + 2];
                int i1 = i + kM; if (i1 >= kN) i1 -= kN;
                int i2 = i1 + 1; if (i2 == kN) i2 = 0;
                const uint32_t m0 = k[i1], m1 = k[i2];
                k[i] = m0 ^ twist_d(a0, a1);
                k[i + 1] = m1 ^ twist_d(a1, a2);
            }
        }
    }
    for (int j = lane; j < kN; j += 64) out[j] = k[j];
}

// ---- generation ------------------------------------------------------------------------------------------------------
template <typename T>
struct GenArgs {
    const uint32_t* states;     // [s_hi - s_lo + 1][F][624]
    int F;
    long s_lo, s_hi;            // segments to run
    long b_hi;                  // last block whose words are wanted
    int pos;                    // position in block 0 of stream word 0
    long skip, count;           // doubles [skip, skip + count) of the draw go to out[0 .. count)
    T* out;
};

template <typename T>
__device__ __forceinline__ void emit_pair(const GenArgs<T>& a, uint32_t wa, uint32_t wb, long q) {
    const long qq = q - a.skip;
    if (qq >= 0 && qq < a.count) {
        // randomkit rk_double: (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53, exact in double
        const double v = ((double)(wa >> 5) * 67108864.0 + (double)(wb >> 6)) * (1.0 / 9007199254740992.0);
        a.out[qq] = (T)v;          // T = float: round to nearest even, as numpy's astype(float32)
    }
}

template <typename T>
__global__ __launch_bounds__(256) void mt_gen_kernel(const GenArgs<T> a) {
    __shared__ __align__(16) uint32_t keys[4][kN + 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long seg = a.s_lo + (long)blockIdx.x * 4 + wave;
    if (seg > a.s_hi) return;                       // (no workgroup barrier in this kernel: waves are independent)
    volatile uint32_t* k = keys[wave];
    const uint32_t* src = a.states + (size_t)(seg - a.s_lo) * a.F * kN;
    for (int j = lane; j < kN; j += 64) {
        uint32_t v = 0;
        for (int s = 0; s < a.F; ++s) v ^= src[(size_t)s * kN + j];
        k[j] = v;
    }
    const bool odd = a.pos & 1;
    const bool act = lane < 52;
    // doubles pair stream words (2q, 2q + 1); with an odd position a pair is (word i - 1, word i) of a block, i even, and the
    // first pair of a block takes its first word from the block before: `carry`
    uint32_t carry = temper_d(k[kN - 1]);
    const long b0 = seg * kSegBlocks;               // the block the loaded state holds
    if (seg == 0 && a.pos < kN) {
        // the rest of the caller's current block (exact copy of its key)
        const long qb = -(long)((a.pos + (odd ? 1 : 0)) >> 1);
        for (int c = 0; c < 6; ++c) {
            const int i = 104 * c + 2 * (act ? lane : 51);
            const uint32_t t0 = temper_d(k[i]), t1 = temper_d(k[i + 1]);
            if (odd) {
                uint32_t prev = __shfl_up(t1, 1);
                if (lane == 0) prev = carry;
                if (act) emit_pair(a, prev, t0, qb + (i >> 1));
                carry = __builtin_amdgcn_readlane(t1, 51);
            } else if (act) emit_pair(a, t0, t1, qb + (i >> 1));
        }
    }
    long nb = a.b_hi - b0;
    if (nb > kSegBlocks) nb = kSegBlocks;
    for (long bb = 1; bb <= nb; ++bb) {
        const long b = b0 + bb;
        const long qb = (b * kN - a.pos - (odd ? 1 : 0)) >> 1;     // (exact: the numerator is even)
        for (int c = 0; c < 6; ++c) {
            const int i = 104 * c + 2 * (act ? lane : 51);
            const uint32_t a0 = k[i], a1 = k[i + 1], a2 = k[i + 2 == kN ? 0 : i + 2];
            int i1 = i + kM; if (i1 >= kN) i1 -= kN;
            int i2 = i1 + 1; if (i2 == kN) i2 = 0;
            const uint32_t m0 = k[i1], m1 = k[i2];
            const uint32_t n0 = m0 ^ twist_d(a0, a1), n1 = m1 ^ twist_d(a1, a2);
            if (act) { k[i] = n0; k[i + 1] = n1; }
            const uint32_t t0 = temper_d(n0), t1 = temper_d(n1);
            if (odd) {
                uint32_t prev = __shfl_up(t1, 1);
                if (lane == 0) prev = carry;
                if (act) emit_pair(a, prev, t0, qb + (i >> 1));
                carry = __builtin_amdgcn_readlane(t1, 51);
            } else if (act) emit_pair(a, t0, t1, qb + (i >> 1));
        }
    }
}

// ---- host: polynomial tables -------------------------------------------------------------------------------------------
struct HostTables {
    std::mutex mu;
    Field field;
    bool field_ok = false, field_tried = false;
    Poly base[kLevels];                                  // t^(624 * stride_l)
    bool have[kLevels] = {};
    std::vector<uint32_t> taps[kLevels];                 // [63][kTapCap / 2]
    std::vector<int> ntaps[kLevels];                     // [63]
    bool ensure_field() {
        if (!field_tried) { field_tried = true; field_ok = field.init(); }
        return field_ok;
    }
    // caller holds mu
    bool ensure_level(int l) {
        if (have[l]) return true;
        if (!ensure_field()) return false;
        for (int q = 0; q <= l; ++q) {
            if (have[q]) continue;
            if (q == 0) {
                base[0] = Field::monomial(kN);
                for (int s = 0; s < kStride0Log2; ++s) base[0] = field.sqr(base[0]);
            } else {
                base[q] = base[q - 1];
                for (int s = 0; s < kRadixLog2; ++s) base[q] = field.sqr(base[q]);
            }
            taps[q].assign((size_t)63 * (kTapCap / 2), 0u);
            ntaps[q].assign(63, 0);
            Poly p = base[q];
            for (int d = 1; d <= 63; ++d) {
                if (d > 1) p = field.mul(p, base[q]);
                uint32_t* tp = taps[q].data() + (size_t)(d - 1) * (kTapCap / 2);
                int nt = 0;
                for (int i = 0; i < kDeg; ++i)
                    if (get_bit(p.w, i)) { tp[nt >> 1] |= (uint32_t)i << ((nt & 1) * 16); ++nt; }
                ntaps[q][d - 1] = nt;
            }
            have[q] = true;
        }
        return true;
    }
};
static HostTables& host_tables() { static HostTables* t = new HostTables; return *t; }

struct DeviceTables {
    uint32_t* taps[kLevels] = {};
    int* ntaps[kLevels] = {};
    bool lds_attr = false;
    // side stream, pinned state buffer and workspace of the chain that computes the state after the draw
    hipStream_t side = nullptr;
    hipEvent_t done = nullptr;
    uint32_t* pinned = nullptr;      // [624]
    uint32_t* chain = nullptr;       // [kLevels][kMaxShare][624]
    std::mutex chain_mu;             // the chain's buffers are one set per device: calls on a device take turns in it
};
static std::mutex g_dev_mu;
static DeviceTables* g_dev[64] = {};

static hipError_t device_tables(int need_level, DeviceTables** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(g_dev_mu);
    DeviceTables*& t = g_dev[dev];
    if (!t) t = new DeviceTables;
    if (!t->lds_attr) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(mt_jump_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(kXLen * sizeof(uint32_t)));
        if (e != hipSuccess) return e;
        t->lds_attr = true;
    }
    if (!t->side) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if ((e = hipStreamCreateWithPriority(&t->side, hipStreamNonBlocking, hi)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&t->done, hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipHostMalloc((void**)&t->pinned, sizeof(uint32_t) * kN, hipHostMallocDefault)) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&t->chain, sizeof(uint32_t) * (size_t)kLevels * kMaxShare * kN)) != hipSuccess) return e;
    }
    HostTables& h = host_tables();
    std::lock_guard<std::mutex> gh(h.mu);
    if (!h.ensure_level(need_level)) return hipErrorUnknown;
    for (int l = 0; l <= need_level; ++l) {
        if (t->taps[l]) continue;
        uint32_t* dt = nullptr; int* dn = nullptr;
        if ((e = hipMalloc((void**)&dt, h.taps[l].size() * sizeof(uint32_t))) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&dn, 63 * sizeof(int))) != hipSuccess) return e;
        if ((e = hipMemcpy(dt, h.taps[l].data(), h.taps[l].size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) return e;
        if ((e = hipMemcpy(dn, h.ntaps[l].data(), 63 * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return e;
        t->taps[l] = dt; t->ntaps[l] = dn;
    }
    *out = t;
    return hipSuccess;
}

static int share_for(long count) {
    int F = 1;
    while (F < kMaxShare && count * (F * 2) <= 512) F *= 2;
    return F;
}
static inline int level_shift(int l) { return kStride0Log2 + kRadixLog2 * l; }

}  // namespace mt

// t^(624 nblocks) mod phi as 313 64-bit words (host only; tests check it against stepping the generator)
int mt19937_jump_poly(unsigned long long nblocks, unsigned long long* bits) {
    mt::HostTables& h = mt::host_tables();
    std::lock_guard<std::mutex> g(h.mu);
    if (!h.ensure_field()) return 1;
    const mt::Poly p = h.field.block_jump(nblocks);
    std::memcpy(bits, p.w, sizeof p.w);
    return 0;
}

// The draw.  key / pos: numpy's RandomState state (host, in/out).  The next `total` doubles of the stream are consumed; doubles
// [skip, skip + count) of them are written to out (device; elem = 4: float, 8: double) on `st`.
hipError_t mt19937_draw(uint32_t* key, int* pos_io, unsigned long long total, unsigned long long skip, unsigned long long count,
                        void* out, int elem, hipStream_t st) {
    using namespace mt;
    const int pos = *pos_io;
    if (pos < 0 || pos > kN || skip + count > total || (count && !out) || (elem != 4 && elem != 8)) return hipErrorInvalidValue;
    if (total == 0) return hipSuccess;
    if (total > (1ull << 40)) return hipErrorInvalidValue;
    const long p_end = (long)pos + 2 * (long)total;              // position of the first unconsumed word, from block 0
    const long b_f = p_end <= kN ? 0 : (p_end - 1) / kN;         // block of the state after the draw
    // levels needed: the top digit of the last block / segment must be < 64
    long b_hi = 0, s_lo = 0, s_hi = 0;
    if (count) {
        const long b_lo = ((long)pos + 2 * (long)skip) / kN;
        b_hi = ((long)pos + 2 * (long)(skip + count) - 1) / kN;
        s_lo = b_lo >= 1 ? (b_lo - 1) / kSegBlocks : 0;
        s_hi = b_hi >= 1 ? (b_hi - 1) / kSegBlocks : 0;
    }
    int top_e = 0, top_b = kSegLevel;
    const long tgt = b_f >= 1 ? b_f - 1 : 0;                    // the chain reaches block b_f - 1, then regenerates once
    while ((tgt >> level_shift(top_e)) >= 64) if (++top_e >= kLevels) return hipErrorInvalidValue;
    while ((s_hi >> (level_shift(top_b) - level_shift(kSegLevel))) >= 64) if (++top_b >= kLevels) return hipErrorInvalidValue;
    DeviceTables* t = nullptr;
    hipError_t e = device_tables(top_e > top_b ? top_e : top_b, &t);
    if (e != hipSuccess) return e;

    JumpArgs ja;
    std::memcpy(ja.root.w, key, sizeof ja.root.w);
    const size_t lds = kXLen * sizeof(uint32_t);

    // (1) the state after the draw, on the side stream
    std::unique_lock<std::mutex> chain_lock(t->chain_mu, std::defer_lock);
    if (b_f >= 1) {
        chain_lock.lock();
        const uint32_t* parent = nullptr; long parent_lo = 0; int pF = 1;
        for (int l = top_e; l >= 0; --l) {
            const long idx = tgt >> level_shift(l);
            ja.parent = parent; ja.parent_lo = parent_lo; ja.pF = pF;
            ja.dst = t->chain + (size_t)l * kMaxShare * kN; ja.lo = idx; ja.F = kMaxShare;
            ja.taps = t->taps[l]; ja.ntaps = t->ntaps[l];
            hipLaunchKernelGGL(mt_jump_kernel, dim3(1, kMaxShare), dim3(256), lds, t->side, ja);
            parent = ja.dst; parent_lo = idx; pF = kMaxShare;
        }
        const int steps = (int)(tgt & ((1 << kStride0Log2) - 1)) + 1;
        hipLaunchKernelGGL(mt_final_kernel, dim3(1), dim3(64), 0, t->side, parent, pF, steps, t->pinned);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = hipEventRecord(t->done, t->side)) != hipSuccess) return e;
    }

    // (2) the wanted doubles, on the caller's stream
    if (count) {
        const uint32_t* parent = nullptr; long parent_lo = 0; int pF = 1;
        uint32_t* bufs[kLevels] = {};
        for (int l = top_b; l >= kSegLevel; --l) {
            const int sh = level_shift(l) - level_shift(kSegLevel);
            const long lo = s_lo >> sh, hi = s_hi >> sh, cnt = hi - lo + 1;
            const int F = share_for(cnt);
            if ((e = hipMallocAsync((void**)&bufs[l], sizeof(uint32_t) * (size_t)cnt * F * kN, st)) != hipSuccess) break;
            ja.parent = parent; ja.parent_lo = parent_lo; ja.pF = pF;
            ja.dst = bufs[l]; ja.lo = lo; ja.F = F;
            ja.taps = t->taps[l]; ja.ntaps = t->ntaps[l];
            hipLaunchKernelGGL(mt_jump_kernel, dim3((unsigned)cnt, F), dim3(256), lds, st, ja);
            parent = bufs[l]; parent_lo = lo; pF = F;
        }
        if (e == hipSuccess) {
            const long nseg = s_hi - s_lo + 1;
            if (elem == 4) {
                GenArgs<float> ga{parent, pF, s_lo, s_hi, b_hi, pos, (long)skip, (long)count, (float*)out};
                hipLaunchKernelGGL(mt_gen_kernel<float>, dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, st, ga);
            } else {
                GenArgs<double> ga{parent, pF, s_lo, s_hi, b_hi, pos, (long)skip, (long)count, (double*)out};
                hipLaunchKernelGGL(mt_gen_kernel<double>, dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, st, ga);
            }
            e = hipGetLastError();
        }
        for (int l = 0; l < kLevels; ++l)
            if (bufs[l]) { const hipError_t fe = hipFreeAsync(bufs[l], st); if (e == hipSuccess) e = fe; }
    }

    // (3) the new state: wait for the side chain only
    if (b_f >= 1) {
        const hipError_t we = hipEventSynchronize(t->done);
        if (we != hipSuccess) return we;
        if (e != hipSuccess) return e;
        std::memcpy(key, t->pinned, sizeof(uint32_t) * kN);
        *pos_io = (int)(p_end - b_f * kN);
    } else {
        if (e != hipSuccess) return e;
        *pos_io = (int)p_end;
    }
    return hipSuccess;
}

}  // namespace ssn
