// numpy's RandomState.random_sample on the device, bit for bit, from the caller's (key, pos) -- the reference's noise stream
// `zs = rng.rand(batchsize, 2N, 2N)` (tc_gan/networks/ssn.py:434-439; order of the stream networks/cwgan.py:438-481; fp64 ->
// floatX downcast utils/theanoutils.py:9-16) without the host drawing it (0.25 s per 1024 x 200 x 200 draw against a 2.8 ms
// forward).  Integer work: the bar is bit equality with numpy (tests/test_mt19937_gpu.py), and the state handed back to the
// host RandomState is numpy's own after the same draw.
//
// MT19937 is one sequential recurrence (x[k+624] = x[k+397] ^ twist(x[k], x[k+1])).  The stream is cut into SEGMENTS of
// 128 * step blocks of 624 words (step = 1, 2, 4, 8 by the size of the draw); a segment starts from a state so many blocks ahead of
// the caller's, which is reached by jump-ahead polynomials (ssn_mt19937_poly.h): a ladder of levels with strides 128, 2^15, 2^23
// blocks, 255 polynomials per level (digit d = 1..255 times the stride; 0.15 s of host arithmetic and 10.7 MB of device memory per
// level, at first use), so that any segment state is at most one jump per level away and all states of a level are computed by
// ONE launch from the states of the level above: one round for draws up to 10 M doubles per unit of `step`, two beyond.
//
// The states (expand + jump) are computed on a stream of the library's own into the library's buffers, beside whatever the
// caller's stream is still running; the caller's stream waits for them and runs the generation kernel alone.
//
//   mt_expand_kernel one workgroup per parent state: the 20608 words x[] its children's taps can reach, to memory.
//   mt_jump_kernel   one workgroup per (state, quarter of the polynomial's tap range): its window of the parent's x[] in LDS,
//                    twice (the second copy shifted by one word, so that every tap is ONE aligned 8-byte read per lane: words
//                    tap + 2 t, tap + 2 t + 1 for lane t); taps arrive as LDS byte offsets, 128 per vector load a group ahead,
//                    and reach the scalar side by v_readlane; XOR into two registers per lane.  Consumers XOR the four shares
//                    when they load the state.
//   mt_gen_kernel    one workgroup per segment: wave 0 regenerates block after block (104 words per pass, 52 lanes x 2 words:
//                    a pass depends on no pass nearer than two back), the other waves temper the block before, form randomkit's
//                    rk_double (a >> 5, b >> 6) -> double, round to fp32 where asked (round to nearest even = numpy's astype)
//                    and store the [skip, skip + count) window only (the rows of one rank of a data-parallel job).  BUILDW
//                    form: W = make_W_with_x of the number in the same pass (nine waves, six of them emitting); TAIL form: a
//                    second window behind the doubles -- zs_in of the heterogeneous-input models, `choice(2, n) * 2 - 1` (one
//                    32-bit output per element, its low bit) or `rand(n) * 2 - 1`.
//   mt_solo_kernel   the state after the whole draw (every rank needs it, whatever rows it generates): ONE launch on a side
//                    stream of the library's own, by the exact polynomial of the draw's length (kept per length); the host
//                    waits for this launch only, never for the caller's stream.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include "ssn_host.h"
#include "ssn_mt19937_poly.h"

namespace ssn {
namespace mt {

#ifndef SSN_MT_STRIDE0_LOG2
#define SSN_MT_STRIDE0_LOG2 7
#endif
#ifndef SSN_MT_RADIX_LOG2
#define SSN_MT_RADIX_LOG2 8
#endif
#ifndef SSN_MT_MAX_STEP
#define SSN_MT_MAX_STEP 8
#endif
#ifndef SSN_MT_FRACTIONAL_ROUNDS
#define SSN_MT_FRACTIONAL_ROUNDS 0
#endif
constexpr int kLevels = 3;                 // strides 128 * 256^l blocks
constexpr int kStride0Log2 = SSN_MT_STRIDE0_LOG2;
constexpr int kRadixLog2 = SSN_MT_RADIX_LOG2;   // 255 polynomials per level: up to 255 * 128 * step blocks (10 M doubles x step) in ONE round of jumps
constexpr int kDigits = (1 << kRadixLog2) - 1;
constexpr int kSegLevel = 0;               // segments start at states of level 0: every `step`-th one, 128 * step blocks long
constexpr int kSoloParts = 16;             // workgroups of the kernel that computes the state after the draw
// a polynomial's taps are cut into kShares index ranges of kShareSpan; one workgroup accumulates one share of one state
constexpr int kShares = 4;
constexpr int kShareSpan = 4992;                       // 4 * 4992 = 19968 >= 19937
constexpr int kWinWords = kShareSpan + kN + 16;        // words of the sequence a share can reach (5616) + pad
constexpr int kCopy1Off = kWinWords * 4;               // byte offset of the copy shifted by one word (8-byte reads of odd taps)
constexpr int kZeroOff = 2 * kCopy1Off;                // 320 x 8 bytes of zeros: the target of the padding codes
constexpr int kJumpLds = kZeroOff + 320 * 8;           // 47616 bytes: three workgroups per CU
constexpr int kCodeCap = kShareSpan + 256;             // codes of one share (uint16 byte offsets): groups of 128, one group of pad
constexpr int kXSeq = 20608;                           // words of an expanded sequence (3 * 4992 + 5616 = 20592 are read)

struct Key { uint32_t w[kN]; };

__device__ __forceinline__ uint32_t twist_d(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
// the same in six instructions for the generator wave, whose instruction count is the kernel's time (the compiler turns the
// C form into and / and / or / shift / and / compare / select: ten)
__device__ __forceinline__ uint32_t twist_xor(uint32_t u, uint32_t v, uint32_t far) {
    uint32_t y, m;
    const uint32_t lowmask = 0x7fffffffu;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(y) : "s"(lowmask), "v"(v), "v"(u));     // (mask & v) | (~mask & u)
    asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(m) : "v"(v));                               // bit 0 of v, sign-extended: 0 or ~0
    return far ^ (y >> 1) ^ (m & 0x9908b0dfu);
}
__device__ __forceinline__ uint32_t temper_d(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// ---- expand: the word sequence x[0 .. kXSeq) of a state ---------------------------------------------------------------------
struct ExpandArgs {
    Key root;                   // the caller's key (states == nullptr)
    const uint32_t* states;     // [count][nparts][624] partial sums of states, or nullptr: root
    int nparts;
    uint32_t* xseq;             // [count][kXSeq]
};

__device__ __forceinline__ uint32_t xor_parts(const uint32_t* src, int nparts, int j) {
    uint32_t v = 0;
    for (int q = 0; q < nparts; ++q) v ^= src[(size_t)q * kN + j];
    return v;
}

__global__ __launch_bounds__(256) void mt_expand_kernel(const ExpandArgs a) {
    extern __shared__ __align__(16) uint32_t x[];
    const int tid = threadIdx.x;
    const uint32_t* src = a.states ? a.states + (size_t)blockIdx.x * a.nparts * kN : nullptr;
    for (int j = tid; j < kN; j += 256) x[j] = src ? xor_parts(src, a.nparts, j) : a.root.w[j];
    __syncthreads();
    for (int k0 = 0; k0 + kN < kXSeq; k0 += kN - kM) {          // 227 independent words per step
        const int k = k0 + tid;
        if (tid < kN - kM && k + kN < kXSeq) x[k + kN] = x[k + kM] ^ twist_d(x[k], x[k + 1]);
        __syncthreads();
    }
    uint32_t* o = a.xseq + (size_t)blockIdx.x * kXSeq;
    for (int k = tid; k < kXSeq; k += 256) o[k] = x[k];
}

// ---- jump ------------------------------------------------------------------------------------------------------------
struct JumpArgs {
    const uint32_t* xseq;       // [..][kXSeq] expanded sequences of the states one level up (or of the root)
    long parent_lo;             // index (in units of the parent level's stride) of xseq[0]
    uint32_t* dst;              // [count][kShares * sub][624]
    long lo;                    // index (in units of this level's stride) of dst[0]; dst[k] is state lo + k * step
    int step;
    int sub;                    // workgroups per share (1, 2 or 4: few states -> more workgroups each)
    const uint16_t* codes;      // this level: [kDigits][kShares][kCodeCap] LDS byte offsets of the taps, padded with kZeroOff
    const int* counts;          // [kDigits][kShares], multiples of 128
};

// Two taps: the codes of a group of 128 sit in the wave's lanes (one dword = two uint16 LDS byte offsets per lane, fetched by a
// vector load a group ahead: the vector-memory counter is not the LDS counter, so waiting for codes never drains the reads in
// flight) and come to the scalar side by v_readlane.
#define SSN_MT_TAP2(k)                                                                          \
    {                                                                                           \
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cw, k);                     \
        const uint2 u = *(const uint2*)(lds + lane8 + (w & 0xffffu));                           \
        const uint2 v = *(const uint2*)(lds + lane8 + (w >> 16));                               \
        acc0 ^= u.x; acc1 ^= u.y; acc0 ^= v.x; acc1 ^= v.y;                                     \
    }
#define SSN_MT_TAP8(k) SSN_MT_TAP2(k) SSN_MT_TAP2(k + 1) SSN_MT_TAP2(k + 2) SSN_MT_TAP2(k + 3)

__global__ __launch_bounds__(320) void mt_jump_kernel(const JumpArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x;
    const long n = a.lo + (long)blockIdx.x * a.step;
    const int f = blockIdx.y / a.sub, sb = blockIdx.y % a.sub;
    const int d = (int)(n & kDigits);
    const uint32_t* xs = a.xseq + (size_t)((n >> kRadixLog2) - a.parent_lo) * kXSeq;
    uint32_t* out = a.dst + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * kN;
    if (d == 0) {            // the parent's own state: part 0 carries it (exact, low bits of word 0 included), the others nothing
        for (int j = tid; j < kN; j += 320) out[j] = blockIdx.y == 0 ? xs[j] : 0u;
        return;
    }
    const int base = f * kShareSpan;
    uint32_t* c0 = (uint32_t*)lds;
    uint32_t* c1 = (uint32_t*)(lds + kCopy1Off);
    uint32_t* zz = (uint32_t*)(lds + kZeroOff);
    for (int k = tid; k <= kShareSpan + kN; k += 320) {
        const uint32_t v = xs[base + k];
        if (k < kShareSpan + kN) c0[k] = v;
        if (k) c1[k - 1] = v;
    }
    for (int k = tid; k < 640; k += 320) zz[k] = 0u;
    __syncthreads();
    const int groups = a.counts[(d - 1) * kShares + f] >> 7;                 // groups of 128 taps
    const int g0 = groups * sb / a.sub, g1 = groups * (sb + 1) / a.sub;
    const uint32_t* cp = (const uint32_t*)(a.codes + ((size_t)(d - 1) * kShares + f) * kCodeCap) + (tid & 63);
    const int lane8 = (tid < kN / 2 ? tid : kN / 2 - 1) * 8;
    uint32_t acc0 = 0, acc1 = 0;
    uint32_t cw = cp[(size_t)g0 * 64];
    for (int g = g0; g < g1; ++g) {
        const uint32_t cnext = cp[(size_t)(g + 1) * 64];                   // (the table is padded: one group ahead is in bounds)
        SSN_MT_TAP8(0) SSN_MT_TAP8(4) SSN_MT_TAP8(8) SSN_MT_TAP8(12) SSN_MT_TAP8(16) SSN_MT_TAP8(20) SSN_MT_TAP8(24) SSN_MT_TAP8(28)
        SSN_MT_TAP8(32) SSN_MT_TAP8(36) SSN_MT_TAP8(40) SSN_MT_TAP8(44) SSN_MT_TAP8(48) SSN_MT_TAP8(52) SSN_MT_TAP8(56) SSN_MT_TAP8(60)
        cw = cnext;
    }
    if (tid < kN / 2) { out[2 * tid] = acc0; out[2 * tid + 1] = acc1; }
}

// ---- the state after the draw ----------------------------------------------------------------------------------------
// Every rank needs it, whatever rows it generates, and the HOST waits for it (its next `choice` continues from there): ONE
// launch on a stream of the library's own.  The jump is by the exact polynomial of this draw's length (t^(624 T) mod phi, T =
// blocks to the one before the last: computed on the host at the first draw of that length, ~15 ms, and kept), so there is one
// level and no chain of dependent launches, each of which would queue for a CU beside the caller's kernels.
// grid (1, kSoloParts): workgroup y = (share, quarter of the share's taps).  It regenerates its own window of the sequence from
// the caller's key (run-up through a ring of 1248 words, then the window in place), accumulates its taps, writes its partial
// state; the workgroup that finishes last XORs the partials, regenerates once (every bit of the key becomes numpy's, the low
// bits of word 0 included) and writes the key to pinned host memory.
struct SoloArgs {
    Key root;
    const uint16_t* codes;      // [kShares][kCodeCap]
    const int* counts;          // [kShares]
    uint32_t* partials;         // [kSoloParts][624]
    unsigned* counter;          // zero between launches
    uint32_t* out;              // [624], host-visible
};

__global__ __launch_bounds__(320) void mt_solo_kernel(const SoloArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    __shared__ uint32_t ring[2 * kN];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    constexpr int sub = kSoloParts / kShares;
    const int f = blockIdx.y / sub, sb = blockIdx.y % sub;
    const int base = f * kShareSpan;
    uint32_t* c0 = (uint32_t*)lds;
    uint32_t* c1 = (uint32_t*)(lds + kCopy1Off);
    uint32_t* zz = (uint32_t*)(lds + kZeroOff);
    for (int j = tid; j < kN; j += 320) ring[j] = a.root.w[j];
    for (int k = tid; k < 640; k += 320) zz[k] = 0u;
    __syncthreads();
    // run-up: x[624 .. ) in the ring (x[k] at k mod 1248) until x[base .. base + 624) is there
    int k0 = 0;
    for (; k0 < base; k0 += kN - kM) {
        const int k = k0 + tid;
        if (tid < kN - kM)
            ring[(k + kN) % (2 * kN)] = ring[(k + kM) % (2 * kN)] ^ twist_d(ring[k % (2 * kN)], ring[(k + 1) % (2 * kN)]);
        __syncthreads();
    }
    for (int j = tid; j < kN; j += 320) c0[j] = ring[(base + j) % (2 * kN)];
    __syncthreads();
    for (int q0 = 0; q0 + kN <= kShareSpan + kN; q0 += kN - kM) {          // the window, in place
        const int q = q0 + tid;
        if (tid < kN - kM && q + kN <= kShareSpan + kN) c0[q + kN] = c0[q + kM] ^ twist_d(c0[q], c0[q + 1]);
        __syncthreads();
    }
    for (int k = tid; k < kShareSpan + kN; k += 320) c1[k] = c0[k + 1];
    __syncthreads();
    const int groups = a.counts[f] >> 7;
    const int g0 = groups * sb / sub, g1 = groups * (sb + 1) / sub;
    const uint32_t* cp = (const uint32_t*)(a.codes + (size_t)f * kCodeCap) + (tid & 63);
    const int lane8 = (tid < kN / 2 ? tid : kN / 2 - 1) * 8;
    uint32_t acc0 = 0, acc1 = 0;
    uint32_t cw = cp[(size_t)g0 * 64];
    for (int g = g0; g < g1; ++g) {
        const uint32_t cnext = cp[(size_t)(g + 1) * 64];
        SSN_MT_TAP8(0) SSN_MT_TAP8(4) SSN_MT_TAP8(8) SSN_MT_TAP8(12) SSN_MT_TAP8(16) SSN_MT_TAP8(20) SSN_MT_TAP8(24) SSN_MT_TAP8(28)
        SSN_MT_TAP8(32) SSN_MT_TAP8(36) SSN_MT_TAP8(40) SSN_MT_TAP8(44) SSN_MT_TAP8(48) SSN_MT_TAP8(52) SSN_MT_TAP8(56) SSN_MT_TAP8(60)
        cw = cnext;
    }
    uint32_t* mine = a.partials + (size_t)blockIdx.y * kN;
    if (tid < kN / 2) { mine[2 * tid] = acc0; mine[2 * tid + 1] = acc1; }
    // the last workgroup to get here finishes the job (nobody waits for anybody)
    __threadfence();
    __syncthreads();
    if (tid == 0) is_last = atomicAdd(a.counter, 1u) == (unsigned)kSoloParts - 1u;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    for (int j = tid; j < kN; j += 320) {
        uint32_t v = 0;
        for (int q = 0; q < kSoloParts; ++q) v ^= __builtin_nontemporal_load(a.partials + (size_t)q * kN + j);
        ring[j] = v;
    }
    __syncthreads();
    // one regeneration, not in place: new block into ring[624 ..)
    for (int q0 = 0; q0 < kN; q0 += kN - kM) {
        const int q = q0 + tid;
        if (tid < kN - kM && q < kN) {
            const uint32_t nxt = q + 1 < kN ? ring[q + 1] : ring[kN];                 // x[624] = the new word 0
            const uint32_t far = q + kM < kN ? ring[q + kM] : ring[kN + q + kM - kN];
            ring[kN + q] = far ^ twist_d(ring[q], nxt);
        }
        __syncthreads();
    }
    for (int j = tid; j < kN; j += 320) a.out[j] = ring[kN + j];
    if (tid == 0) *a.counter = 0u;
}

// ---- generation ------------------------------------------------------------------------------------------------------
// One workgroup per segment, four waves: wave 0 regenerates block after block (old block in one LDS buffer, new block into the
// other: 104 words per pass, 52 lanes x 2 words; a pass reads new words no nearer than 124 back, so the next pass's operands are
// fetched before this pass's store), waves 1-3 temper, convert and store the block made in the round before, two passes each.
// One workgroup barrier per block.
template <typename T>
struct GenArgs {
    const uint32_t* states;     // [s_hi - s_lo + 1][nparts][624]
    int nparts;
    long s_lo, s_hi;            // segments to run
    int seg_blocks;             // blocks per segment
    long b_hi;                  // last block whose words are wanted
    int pos;                    // position in block 0 of stream word 0
    long skip, count;           // doubles [skip, skip + count) of the draw go to out[0 .. count)
    T* out;                     // the numbers themselves (nullptr: not wanted -- only with W below)
    // W of make_W_with_x straight from the numbers (T = float): out[e] is element e of z[B][M][M], M = 2 N
    T* W = nullptr;
    JDS<float> p = {};
    int N = 0;
    unsigned long long magic_m = 0;   // ceil(2^40 / M): e / M = (e * magic_m) >> 40 for e < 2^28 (count is checked)
    // The tail (TAIL kernels): what the reference draws right behind zs -- zs_in of the heterogeneous-input models
    // (ssn.py:710-720), from stream pair tail_q0 = `total` on: kind 1 = rng.choice(2, n) * 2 - 1 (one 32-bit word per element:
    // numpy's masked rejection with mask 1 takes the low bit of every output and rejects nothing), kind 2 = rng.rand(n) * 2 - 1
    // (one double per element).  Elements [tail_skip, tail_skip + tail_count) go to tail_out.
    long tail_q0 = 0, tail_skip = 0, tail_count = 0;
    int tail_kind = 0;
    float* tail_out = nullptr;
};

constexpr int kBufStride = kN + 32;        // (words 624.. of a buffer: where the twelve idle lanes of a pass put their stores)

struct GenOperands { uint32_t a0, a1, a2, m0, m1; };

// operands of pass c of a regeneration: old block at word offset `oldw`, new block at `neww` of `flat`
template <int C>
__device__ __forceinline__ GenOperands gen_load(const uint32_t* flat, int oldw, int neww, int l) {
    const int i = 104 * C + 2 * l;
    GenOperands r;
    r.a0 = flat[oldw + i];
    r.a1 = flat[oldw + i + 1];
    r.a2 = flat[(i + 2 < kN) ? oldw + i + 2 : neww];
    r.m0 = flat[(i + kM < kN) ? oldw + i + kM : neww + i + kM - kN];
    r.m1 = flat[(i + kM + 1 < kN) ? oldw + i + kM + 1 : neww + i + kM + 1 - kN];
    asm volatile("" ::: "memory");      // (the wave's LDS accesses stay in program order: other lanes' stores are read)
    return r;
}
template <int C>
__device__ __forceinline__ void gen_store(uint32_t* flat, int neww, int l, bool act, const GenOperands& r) {
    const uint32_t n0 = twist_xor(r.a0, r.a1, r.m0), n1 = twist_xor(r.a1, r.a2, r.m1);
    *(uint2*)(flat + neww + (act ? 104 * C + 2 * l : kN + 2 * (l - 52))) = make_uint2(n0, n1);     // (no branch: lanes 52-63 store aside)
    asm volatile("" ::: "memory");
}
template <int P>
__device__ __forceinline__ void gen_block(uint32_t* flat, int lane) {
    constexpr int oldw = P * kBufStride, neww = (1 - P) * kBufStride;
    const bool act = lane < 52;
    const int l = act ? lane : 51;
    GenOperands r0 = gen_load<0>(flat, oldw, neww, l);
    GenOperands r1 = gen_load<1>(flat, oldw, neww, l);
    gen_store<0>(flat, neww, lane, act, r0);
    r0 = gen_load<2>(flat, oldw, neww, l);
    gen_store<1>(flat, neww, lane, act, r1);
    r1 = gen_load<3>(flat, oldw, neww, l);
    gen_store<2>(flat, neww, lane, act, r0);
    r0 = gen_load<4>(flat, oldw, neww, l);
    gen_store<3>(flat, neww, lane, act, r1);
    r1 = gen_load<5>(flat, oldw, neww, l);         // (pass 5 reads new[0] and new[396]: passes 0 and 3, stored above)
    gen_store<4>(flat, neww, lane, act, r0);
    gen_store<5>(flat, neww, lane, act, r1);
}

template <typename T, bool ODD, bool BUILDW, bool TAIL>
__device__ __forceinline__ void emit_pass(const GenArgs<T>& a, const uint32_t* blk, const uint32_t* carry_in,
                                          int c, int lane, long qb, const float* gtab) {
    const bool act = lane < 52;
    const int i = 104 * c + 2 * (act ? lane : 51);
    uint32_t w0, w1;
    if (ODD) {                      // pair (word i - 1, word i); the first pair of a block starts in the block before
        w0 = blk[i ? i - 1 : 0];
        w1 = blk[i];
        if (i == 0) w0 = *carry_in;
    } else {
        w0 = blk[i];
        w1 = blk[i + 1];
    }
    w0 = temper_d(w0);
    w1 = temper_d(w1);
    const long qq = qb + (i >> 1) - a.skip;
    if (act && (unsigned long)qq < (unsigned long)a.count) {
        // randomkit rk_double: (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53, exact in double
        const double v = ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) * (1.0 / 9007199254740992.0);
        const T z = (T)v;          // T = float: round to nearest even, as numpy's astype(float32)
        if constexpr (BUILDW) {
            // the element's place in its draw: e = (b M + row) M + col
            const unsigned e = (unsigned)qq, M = 2u * (unsigned)a.N;
            const unsigned er = (unsigned)(((unsigned long long)e * a.magic_m) >> 40);       // e / M
            const unsigned col = e - er * M;
            const unsigned row = er - (unsigned)(((unsigned long long)er * a.magic_m) >> 40) * M;
            // the Gaussian factor from the launch's table (w_gauss of (pq, |i - j|), the bits w_from_z computes), the rest as there
            const int pp = (int)row >= a.N, i = (int)row - pp * a.N;
            const int cq = (int)col >= a.N, j = (int)col - cq * a.N;
            const int pq = pp * 2 + cq, dij = i > j ? i - j : j - i;
            // (J, D of the block by selects on uniform values: indexing the argument struct by a per-lane pq is a vector load
            // from the argument segment for every number, with its full latency in a wave that has one pass to hide it behind)
            const float Jpq = cq ? (pp ? a.p.J[3] : a.p.J[1]) : (pp ? a.p.J[2] : a.p.J[0]);
            const float Dpq = cq ? (pp ? a.p.D[3] : a.p.D[1]) : (pp ? a.p.D[2] : a.p.D[0]);
            a.W[qq] = w_combine_vals<float>(Jpq, Dpq, cq, gtab[pq * a.N + dij], (float)z);
            if (a.out) a.out[qq] = z;
        } else {
            a.out[qq] = z;
        }
    } else if constexpr (TAIL) {
        const long tq = qq + a.skip - a.tail_q0;          // pairs behind the end of the doubles
        if (act && tq >= 0) {
            if (a.tail_kind == 1) {
                const long e0 = 2 * tq - a.tail_skip;
                if ((unsigned long)e0 < (unsigned long)a.tail_count) a.tail_out[e0] = (w0 & 1u) ? 1.f : -1.f;
                if ((unsigned long)(e0 + 1) < (unsigned long)a.tail_count) a.tail_out[e0 + 1] = (w1 & 1u) ? 1.f : -1.f;
            } else {
                const long e = tq - a.tail_skip;
                if ((unsigned long)e < (unsigned long)a.tail_count) {
                    const double v = ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) * (1.0 / 9007199254740992.0);
                    a.tail_out[e] = (float)(2.0 * v - 1.0);       // (exact in double, then round to nearest: the host's astype)
                }
            }
        }
    }
}

template <typename T, bool ODD, bool BUILDW, bool TAIL>
__global__ __launch_bounds__(BUILDW ? 576 : 256) void mt_gen_kernel(const GenArgs<T> a) {
    // waves: 0 regenerates; three emit two passes of a block each -- or, with W formed here (twice the work per number), six emit
    // one pass each.  Waves w and w + 4 of a workgroup share a SIMD: in the nine-wave form waves 4 and 8 do nothing but keep the
    // barrier, so that the regenerating wave -- the launch's critical path -- has its SIMD to itself, as in the four-wave form.
    constexpr int EW = BUILDW ? 6 : 3, PASSES = 6 / EW, NTHR = BUILDW ? 576 : 256;
    __shared__ __align__(16) uint32_t buf[2 * kBufStride];
    __shared__ uint32_t carry[2];          // raw last word of the block emitted in the round before (ODD)
    extern __shared__ __align__(16) float gtab_dyn[];     // BUILDW: [4][N] Gaussian factors of make_W_with_x (w_gauss)
    const float* gtab = nullptr;
    if constexpr (BUILDW) {
        const float inv_nm1 = a.N > 1 ? 1.f / (float)(a.N - 1) : 0.f;
        for (int t = threadIdx.x; t < 4 * a.N; t += NTHR) gtab_dyn[t] = w_gauss<float>(a.p, inv_nm1, t / a.N, t % a.N);
        gtab = gtab_dyn;            // (the barriers below order it before the first emission)
    }
    uint32_t* flat = buf;
    uint32_t* vcarry = carry;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long seg = a.s_lo + (long)blockIdx.x;
    const uint32_t* src = a.states + (size_t)blockIdx.x * a.nparts * kN;
    for (int j = tid; j < kN; j += NTHR) flat[j] = xor_parts(src, a.nparts, j);
    __syncthreads();
    if (tid == 0) { vcarry[0] = flat[kN - 1]; vcarry[1] = 0u; }
    __syncthreads();
    const long b0 = seg * a.seg_blocks;             // the block the loaded state holds
    long nbl = a.b_hi - b0;
    if (nbl > a.seg_blocks) nbl = a.seg_blocks;
    if (nbl < 0) nbl = 0;
    const int nb = (int)nbl;
    const int first = (seg == 0 && a.pos < kN) ? 0 : 1;     // segment 0 also emits the rest of the caller's current block
    if (wave == 0) __builtin_amdgcn_s_setprio(3);           // (the regenerating wave is the launch's critical path: ahead of an emitter on its SIMD)
    for (int bb = 0; bb <= nb; ++bb) {
        const int p = bb & 1;
        if (wave == 0) {
            if (bb < nb) { if (p) gen_block<1>(flat, lane); else gen_block<0>(flat, lane); }
        } else if (bb >= first) {
            const long b = b0 + bb;
            const long qb = (b * kN - a.pos - (ODD ? 1 : 0)) >> 1;         // (exact: the numerator is even)
            const uint32_t* blk = flat + p * kBufStride;
            const int ei = BUILDW ? (wave < 4 ? wave - 1 : wave - 2) : wave - 1;        // emitter 0 .. EW - 1 (waves 4, 8: none)
            if (!BUILDW || (wave & 3) != 0) {
                const int c = PASSES * ei;
                emit_pass<T, ODD, BUILDW, TAIL>(a, blk, vcarry + (p ^ 1), c, lane, qb, gtab);
                if constexpr (PASSES == 2) emit_pass<T, ODD, BUILDW, TAIL>(a, blk, vcarry + (p ^ 1), c + 1, lane, qb, gtab);
                if (ODD && ei == EW - 1 && lane == 0) vcarry[p] = blk[kN - 1];
            }
        }
        __syncthreads();
    }
}

// ---- host: polynomial tables -------------------------------------------------------------------------------------------
// LDS byte offsets of a polynomial's taps, share by share (even tap: 8 bytes at word `rel` of the share's window; odd: at
// rel - 1 of the copy shifted by one word); counts padded to groups of 128 with the offset of the zero area
static void pack_codes(const Poly& p, uint16_t* codes, int* counts) {
    for (int f = 0; f < kShares; ++f) {
        uint16_t* cp = codes + (size_t)f * kCodeCap;
        for (int q = 0; q < kCodeCap; ++q) cp[q] = (uint16_t)kZeroOff;
        int nt = 0;
        for (int rel = 0; rel < kShareSpan; ++rel) {
            const int i = f * kShareSpan + rel;
            if (i < kDeg && get_bit(p.w, i)) cp[nt++] = (uint16_t)((rel & ~1) * 4 + (rel & 1) * kCopy1Off);
        }
        counts[f] = (nt + 127) & ~127;
    }
}

struct HostTables {
    std::mutex mu;
    Field field;
    bool field_ok = false, field_tried = false;
    Poly base[kLevels];                                  // t^(624 * stride_l)
    bool have[kLevels] = {};
    std::vector<uint16_t> codes[kLevels];                // [kDigits][kShares][kCodeCap]
    std::vector<int> counts[kLevels];                    // [kDigits][kShares]
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
            codes[q].assign((size_t)kDigits * kShares * kCodeCap, (uint16_t)kZeroOff);
            counts[q].assign(kDigits * kShares, 0);
            Poly p = base[q];
            for (int d = 1; d <= kDigits; ++d) {
                if (d > 1) p = field.mul(p, base[q]);
                pack_codes(p, codes[q].data() + (size_t)(d - 1) * kShares * kCodeCap, counts[q].data() + (d - 1) * kShares);
            }
            have[q] = true;
        }
        return true;
    }
};
static HostTables& host_tables() { static HostTables* t = new HostTables; return *t; }

struct DeviceTables {
    uint16_t* codes[kLevels] = {};
    int* counts[kLevels] = {};
    bool lds_attr = false;
    // side stream, pinned state buffer and workspace of the kernel that computes the state after the draw
    hipStream_t side = nullptr;
    // the state after a draw is handed over through a ticket (a draw may return before that state is known: the host does
    // the waiting when it next needs its generator): pinned key buffer + event per ticket
    static constexpr int kTickets = 8;
    struct Ticket { hipEvent_t done = nullptr; uint32_t* pinned = nullptr; bool busy = false, computed = false; int pos = 0; unsigned gen = 0; uint32_t key[kN]; };
    Ticket tickets[kTickets];
    unsigned next_ticket = 0;
    // states of draws whose slot was needed again before their owner asked (a generator that sat idle while others drew):
    // fetched at takeover and kept here by ticket
    struct Parked { int pos; uint32_t key[kN]; };
    std::map<int, Parked> parked;
    uint32_t* chain = nullptr;       // [kSoloParts][624] partial states + the arrival counter
    std::mutex chain_mu;             // one set per device: calls on a device take turns in it
    // Segment states are computed on the side stream too, into buffers of the library's own: that work needs nothing of the
    // caller's (no output buffer, none of its stream's results), so it starts when the call is made and runs beside whatever
    // the caller's stream is still busy with; the caller's stream waits for `ready` and runs the generation kernel only.  A
    // ring of sets: a set is reused once the generation kernel that read it has finished (`consumed`).
    struct BulkSet {
        uint32_t* states[kLevels] = {}; size_t states_cap[kLevels] = {};
        uint32_t* xseq[kLevels] = {}; size_t xseq_cap[kLevels] = {};
        hipEvent_t ready = nullptr, consumed = nullptr;
        bool used = false;
    };
    static constexpr int kBulkSets = 3;
    BulkSet bulk[kBulkSets];
    unsigned next_bulk = 0;
    // exact polynomials by jump length in blocks (a run draws the same shapes over and over: two lengths per shape)
    struct Exact { uint16_t* codes; int* counts; };
    std::map<long, Exact> exact;
};
static std::mutex g_dev_mu;
static DeviceTables* g_dev[64] = {};

// the device copy of t^(624 nblocks) mod phi (caller holds t->chain_mu)
static hipError_t exact_poly(DeviceTables* t, long nblocks, DeviceTables::Exact* out) {
    auto it = t->exact.find(nblocks);
    if (it != t->exact.end()) { *out = it->second; return hipSuccess; }
    if (t->exact.size() >= 256) {          // (a run that keeps changing its draw length: start over rather than grow for ever)
        for (auto& kv : t->exact) { (void)hipFree(kv.second.codes); (void)hipFree(kv.second.counts); }
        t->exact.clear();
    }
    std::vector<uint16_t> codes((size_t)kShares * kCodeCap);
    int counts[kShares];
    {
        HostTables& h = host_tables();
        std::lock_guard<std::mutex> g(h.mu);
        if (!h.ensure_field()) return hipErrorUnknown;
        pack_codes(h.field.block_jump((unsigned long long)nblocks), codes.data(), counts);
    }
    DeviceTables::Exact x{};
    hipError_t e;
    if ((e = hipMalloc((void**)&x.codes, codes.size() * sizeof(uint16_t))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&x.counts, sizeof counts)) == hipSuccess
        && (e = hipMemcpy(x.codes, codes.data(), codes.size() * sizeof(uint16_t), hipMemcpyHostToDevice)) == hipSuccess)
        e = hipMemcpy(x.counts, counts, sizeof counts, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(x.codes); if (x.counts) (void)hipFree(x.counts); return e; }
    t->exact[nblocks] = x;
    *out = x;
    return hipSuccess;
}

static hipError_t device_tables(int need_level, DeviceTables** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(g_dev_mu);
    DeviceTables*& t = g_dev[dev];
    if (!t) t = new DeviceTables;
    if (!t->lds_attr) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(mt_expand_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(kXSeq * sizeof(uint32_t)));
        if (e != hipSuccess) return e;
        t->lds_attr = true;
    }
    if (!t->side) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        // (SSN_MT_PRIORITY=low|mid in the environment: A/B runs -- the default is the greatest priority the device offers)
        const char* pe = getenv("SSN_MT_PRIORITY");
        const int prio = (pe && pe[0] == 'l') ? lo : ((pe && pe[0] == 'm') ? (lo + hi) / 2 : hi);
        if ((e = hipStreamCreateWithPriority(&t->side, hipStreamNonBlocking, prio)) != hipSuccess) return e;
        for (auto& tk : t->tickets) {
            if ((e = hipEventCreateWithFlags(&tk.done, hipEventDisableTiming)) != hipSuccess) return e;
            if ((e = hipHostMalloc((void**)&tk.pinned, sizeof(uint32_t) * kN, hipHostMallocDefault)) != hipSuccess) return e;
        }
        if ((e = hipMalloc((void**)&t->chain, sizeof(uint32_t) * ((size_t)kSoloParts * kN + 16))) != hipSuccess) return e;
        if ((e = hipMemset(t->chain, 0, sizeof(uint32_t) * ((size_t)kSoloParts * kN + 16))) != hipSuccess) return e;
    }
    HostTables& h = host_tables();
    std::lock_guard<std::mutex> gh(h.mu);
    if (need_level >= 0 && !h.ensure_level(need_level)) return hipErrorUnknown;
    for (int l = 0; l <= need_level; ++l) {
        if (t->codes[l]) continue;
        uint16_t* dc = nullptr; int* dn = nullptr;
        if ((e = hipMalloc((void**)&dc, h.codes[l].size() * sizeof(uint16_t))) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&dn, h.counts[l].size() * sizeof(int))) != hipSuccess) return e;
        if ((e = hipMemcpy(dc, h.codes[l].data(), h.codes[l].size() * sizeof(uint16_t), hipMemcpyHostToDevice)) != hipSuccess) return e;
        if ((e = hipMemcpy(dn, h.counts[l].data(), h.counts[l].size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return e;
        t->codes[l] = dc; t->counts[l] = dn;
    }
    *out = t;
    return hipSuccess;
}

static inline int level_shift(int l) { return kStride0Log2 + kRadixLog2 * l; }

// Block 0 is the caller's key (words pos.. of it are still unused), block b its b-th regeneration; segment s starts from the
// state of block s * seg_blocks and regenerates blocks s * seg_blocks + 1 .. (s + 1) * seg_blocks (segment 0 also emits the
// rest of block 0).
struct Plan { long p_end, b_f, b_lo, b_hi, s_lo, s_hi; int step; };
static bool make_plan(int pos, unsigned long long total, unsigned long long skip, unsigned long long count, Plan& pl,
                      const MtTail* tail = nullptr) {
    if (pos < 0 || pos > kN || skip + count > total || total > (1ull << 40)) return false;
    // the tail in 32-bit outputs: all of it, and the wanted window (from the end of the doubles)
    unsigned long long tw_total = 0, tw_first = 0, tw_n = 0;
    if (tail && tail->kind) {
        if ((tail->kind != 1 && tail->kind != 2) || tail->skip + tail->count > tail->total || tail->total > (1ull << 40)) return false;
        const unsigned long long per = tail->kind == 1 ? 1 : 2;
        tw_total = per * tail->total; tw_first = per * tail->skip; tw_n = per * tail->count;
    }
    pl.p_end = (long)pos + 2 * (long)total + (long)tw_total;     // position of the first unconsumed word, from block 0
    pl.b_f = pl.p_end <= kN ? 0 : (pl.p_end - 1) / kN;           // block of the state after the draw
    pl.b_lo = pl.b_hi = pl.s_lo = pl.s_hi = 0;
    pl.step = 1;
    if (count || tw_n) {
        const long first = count ? 2 * (long)skip : 2 * (long)total + (long)tw_first;
        const long last = tw_n ? 2 * (long)total + (long)(tw_first + tw_n) - 1 : 2 * (long)(skip + count) - 1;
        pl.b_lo = ((long)pos + first) / kN;
        pl.b_hi = ((long)pos + last) / kN;
        // segment length 128 * step blocks, by a cost model of the two launches that matter (measured, MI355X): the jump runs 4
        // workgroups per state, 768 at a time, ~70 us a round; a segment's workgroup generates a block in ~0.5 us
        double best = 1e30;
        for (int c = 1; c <= SSN_MT_MAX_STEP; c *= 2) {
            const long len = (long)c << kStride0Log2;
            const long nseg = (pl.b_hi - pl.b_lo + len) / len;
            const double rounds = SSN_MT_FRACTIONAL_ROUNDS ? (4.0 * (double)nseg / 768.0 < 1.0 ? 1.0 : 4.0 * (double)nseg / 768.0)
                                                           : (double)((4 * nseg + 767) / 768);
            const double cost = 70.0 * rounds + 0.5 * (double)len;
            if (cost < best) { best = cost; pl.step = c; }
        }
        const long seg_blocks = (long)pl.step << kStride0Log2;
        pl.s_lo = pl.b_lo >= 1 ? (pl.b_lo - 1) / seg_blocks : 0;
        pl.s_hi = pl.b_hi >= 1 ? (pl.b_hi - 1) / seg_blocks : 0;
    }
    return true;
}
// workgroups per share of a round with `count` states: enough to put a workgroup on most CUs when the states are few
static inline int sub_for(long count) { return count * kShares * 4 <= 512 ? 4 : (count * kShares * 2 <= 512 ? 2 : 1); }

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

// Where a draw lands in the block / segment grid (host arithmetic only: tests check it against numpy's positions)
bool mt19937_plan(int pos, unsigned long long total, unsigned long long skip, unsigned long long count, long* out, const MtTail* tail) {
    mt::Plan pl;
    if (!mt::make_plan(pos, total, skip, count, pl, tail)) return false;
    out[0] = pl.p_end - pl.b_f * mt::kN;   // pos after the draw
    out[1] = pl.b_f;                       // regenerations between the caller's key and the key after the draw
    out[2] = (long)pl.step << mt::kStride0Log2;    // blocks per segment
    out[3] = pl.s_lo; out[4] = pl.s_hi;    // segments this call generates
    out[5] = pl.b_lo; out[6] = pl.b_hi;    // first / last block holding a wanted word
    return true;
}

// The state after a draw begun with mt19937_begin: waits for that draw's one launch on the library's stream, nothing else.
hipError_t mt19937_finish(int ticket, uint32_t* key, int* pos) {
    using namespace mt;
    const int dev = (ticket >> 8) & 255, slot = ticket & 255;
    if (ticket < 0 || dev >= 64 || slot >= DeviceTables::kTickets || !g_dev[dev]) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lock(g_dev[dev]->chain_mu);
    DeviceTables::Ticket& tk = g_dev[dev]->tickets[slot];
    if (!tk.busy || (int)(tk.gen & 0x7fffu) != (ticket >> 16)) {
        auto it = g_dev[dev]->parked.find(ticket);           // its slot was taken over: the state was fetched then
        if (it == g_dev[dev]->parked.end()) return hipErrorInvalidValue;      // (finished before, or never begun)
        std::memcpy(key, it->second.key, sizeof(uint32_t) * kN);
        *pos = it->second.pos;
        g_dev[dev]->parked.erase(it);
        return hipSuccess;
    }
    if (tk.computed) {
        const hipError_t e = hipEventSynchronize(tk.done);
        if (e != hipSuccess) return e;
        std::memcpy(key, tk.pinned, sizeof(uint32_t) * kN);
    } else {
        std::memcpy(key, tk.key, sizeof(uint32_t) * kN);
    }
    *pos = tk.pos;
    tk.busy = false;
    return hipSuccess;
}

// The draw.  key / pos: numpy's RandomState state (host).  The next `total` doubles of the stream are consumed; doubles
// [skip, skip + count) of them are written to out (device; elem = 4: float, 8: double) on `st`.  Returns a ticket for the state
// after the draw (mt19937_finish).
hipError_t mt19937_begin(const uint32_t* key, int pos, unsigned long long total, unsigned long long skip, unsigned long long count,
                         void* out, int elem, hipStream_t st, int* ticket, float* W, const float* jds12, int N, const MtTail* tail) {
    using namespace mt;
    // W != nullptr (elem 4): W of make_W_with_x from the numbers, in the generation kernel itself; `out` (the z) may then be null
    if ((count && !out && !W) || (elem != 4 && elem != 8)) return hipErrorInvalidValue;
    if (W && (elem != 4 || !jds12 || N < 1 || N > 2048 || count >= (1ull << 28) || (skip % (4ull * N * N)) != 0)) return hipErrorInvalidValue;
    if (tail && !tail->kind) tail = nullptr;
    if (tail && (elem != 4 || (tail->count && !tail->out))) return hipErrorInvalidValue;
    const bool want_tail = tail && tail->count;
    if (want_tail && count) {
        // two windows far apart (a rank's rows of z and its rows of the tail): two launches over what each needs, not one over
        // everything in between
        const unsigned long long gap = 2 * (total - skip - count) + (tail->kind == 1 ? 1 : 2) * tail->skip;
        if (gap > 256ull * kN) {
            MtTail none = *tail;
            none.count = 0; none.out = nullptr;
            const hipError_t e1 = mt19937_begin(key, pos, total, skip, count, out, elem, st, ticket, W, jds12, N, &none);
            if (e1 != hipSuccess) return e1;
            return mt19937_begin(key, pos, total, 0, 0, nullptr, elem, st, nullptr, nullptr, nullptr, 0, tail);
        }
    }
    Plan pl;
    if (!make_plan(pos, total, skip, count, pl, tail)) return hipErrorInvalidValue;
    const long p_end = pl.p_end, b_f = pl.b_f, b_hi = pl.b_hi, s_lo = pl.s_lo, s_hi = pl.s_hi;
    const int step = pl.step;
    // levels needed: the top digit of the last segment must be <= kDigits
    const bool any = count || want_tail;
    int top_b = any ? kSegLevel : -1;
    if (any)
        while (((s_hi * step) >> (level_shift(top_b) - level_shift(kSegLevel))) > kDigits) if (++top_b >= kLevels) return hipErrorInvalidValue;
    DeviceTables* t = nullptr;
    hipError_t e = device_tables(top_b, &t);
    if (e != hipSuccess) return e;

    ExpandArgs xa;
    std::memcpy(xa.root.w, key, sizeof xa.root.w);
    const size_t xlds = kXSeq * sizeof(uint32_t);
    JumpArgs ja;

    // (1) the state after the draw, on the side stream: block b_f - 1 by the exact polynomial, then one regeneration
    std::unique_lock<std::mutex> chain_lock(t->chain_mu);
    if (ticket) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const int slot = (int)(t->next_ticket++ % DeviceTables::kTickets);
        DeviceTables::Ticket& tk = t->tickets[slot];
        // a slot still marked busy after the ring has come round: its owner has not asked for the state yet (a generator that sat
        // idle while others drew, or one that was dropped).  The state is fetched now (its launch is long over) and parked.
        if (tk.busy) {
            DeviceTables::Parked pk;
            pk.pos = tk.pos;
            if (tk.computed) {
                if ((e = hipEventSynchronize(tk.done)) != hipSuccess) return e;
                std::memcpy(pk.key, tk.pinned, sizeof pk.key);
            } else {
                std::memcpy(pk.key, tk.key, sizeof pk.key);
            }
            if (t->parked.size() >= 4096) t->parked.erase(t->parked.begin());       // (dropped generators: do not grow for ever)
            t->parked[(int)((tk.gen & 0x7fffu) << 16) | (dev << 8) | slot] = pk;
            tk.busy = false;
        }
        ++tk.gen;
        tk.busy = true; tk.computed = b_f >= 1;
        tk.pos = b_f >= 1 ? (int)(p_end - b_f * kN) : (int)p_end;
        if (b_f < 1) std::memcpy(tk.key, key, sizeof tk.key);
        *ticket = (int)((tk.gen & 0x7fffu) << 16) | (dev << 8) | slot;
        if (b_f >= 1) {
            DeviceTables::Exact ex{};
            if ((e = exact_poly(t, b_f - 1, &ex)) != hipSuccess) return e;
            SoloArgs sa;
            std::memcpy(sa.root.w, key, sizeof sa.root.w);
            sa.codes = ex.codes; sa.counts = ex.counts;
            sa.partials = t->chain; sa.counter = (unsigned*)(t->chain + (size_t)kSoloParts * kN); sa.out = tk.pinned;
            hipLaunchKernelGGL(mt_solo_kernel, dim3(1, kSoloParts), dim3(320), kJumpLds, t->side, sa);
            if ((e = hipGetLastError()) != hipSuccess) { tk.busy = false; return e; }
            if ((e = hipEventRecord(tk.done, t->side)) != hipSuccess) { tk.busy = false; return e; }
        }
    }

    // (2) the wanted doubles: the states of every level between the root and the segments, top down, on the side stream (behind
    // the launch above) into a set of the library's buffers; then, on the caller's stream, the generation kernel
    if (any) {
        DeviceTables::BulkSet& bs = t->bulk[t->next_bulk++ % DeviceTables::kBulkSets];
        if (!bs.ready) {
            if ((e = hipEventCreateWithFlags(&bs.ready, hipEventDisableTiming)) != hipSuccess) return e;
            if ((e = hipEventCreateWithFlags(&bs.consumed, hipEventDisableTiming)) != hipSuccess) return e;
        }
        if (bs.used && (e = hipStreamWaitEvent(t->side, bs.consumed, 0)) != hipSuccess) return e;
        const uint32_t* parent = nullptr; long parent_cnt = 1; int parent_parts = 1;
        for (int l = top_b; l >= kSegLevel && e == hipSuccess; --l) {
            const int sh = level_shift(l) - level_shift(kSegLevel);
            // level 0: the segment starts (every step-th state); above: every state between the first and the last one's ancestors
            const int lstep = l == kSegLevel ? step : 1;
            const long lo = (s_lo * step) >> sh, hi = (s_hi * step) >> sh, cnt = (hi - lo) / lstep + 1;
            const int sub = sub_for(cnt);
            const size_t need_x = sizeof(uint32_t) * (size_t)parent_cnt * kXSeq, need_s = sizeof(uint32_t) * (size_t)cnt * kShares * sub * kN;
            if (need_x > bs.xseq_cap[l] || need_s > bs.states_cap[l]) {
                // (grow: rare -- a larger draw than any before.  The set may still be read by the kernels of an earlier draw)
                if (bs.used && (e = hipEventSynchronize(bs.consumed)) != hipSuccess) break;
                if ((e = hipStreamSynchronize(t->side)) != hipSuccess) break;
                if (need_x > bs.xseq_cap[l]) {
                    if (bs.xseq[l]) (void)hipFree(bs.xseq[l]);
                    bs.xseq[l] = nullptr; bs.xseq_cap[l] = 0;
                    if ((e = hipMalloc((void**)&bs.xseq[l], need_x + need_x / 4)) != hipSuccess) break;
                    bs.xseq_cap[l] = need_x + need_x / 4;
                }
                if (need_s > bs.states_cap[l]) {
                    if (bs.states[l]) (void)hipFree(bs.states[l]);
                    bs.states[l] = nullptr; bs.states_cap[l] = 0;
                    if ((e = hipMalloc((void**)&bs.states[l], need_s + need_s / 4)) != hipSuccess) break;
                    bs.states_cap[l] = need_s + need_s / 4;
                }
            }
            xa.states = parent; xa.nparts = parent_parts; xa.xseq = bs.xseq[l];
            hipLaunchKernelGGL(mt_expand_kernel, dim3((unsigned)parent_cnt), dim3(256), xlds, t->side, xa);
            ja.xseq = bs.xseq[l]; ja.parent_lo = lo >> kRadixLog2;
            ja.dst = bs.states[l]; ja.lo = lo; ja.step = lstep; ja.sub = sub;
            ja.codes = t->codes[l]; ja.counts = t->counts[l];
            hipLaunchKernelGGL(mt_jump_kernel, dim3((unsigned)cnt, kShares * sub), dim3(320), kJumpLds, t->side, ja);
            parent = bs.states[l]; parent_cnt = cnt; parent_parts = kShares * sub;
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(bs.ready, t->side);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, bs.ready, 0);
        if (e == hipSuccess) {
            const unsigned nseg = (unsigned)(s_hi - s_lo + 1);
            const bool odd = pos & 1;
            if (elem == 4) {
                GenArgs<float> ga{parent, parent_parts, s_lo, s_hi, step << kStride0Log2, b_hi, pos, (long)skip, (long)count, (float*)out};
                if (want_tail) {
                    ga.tail_q0 = (long)total; ga.tail_skip = (long)tail->skip; ga.tail_count = (long)tail->count;
                    ga.tail_kind = tail->kind; ga.tail_out = tail->out;
                }
                size_t glds = 0;
                if (W) {
                    ga.W = W; ga.N = N;
                    ga.magic_m = ((1ull << 40) + 2ull * N - 1) / (2ull * N);
                    for (int q = 0; q < 4; ++q) {
                        ga.p.J[q] = jds12[q]; ga.p.D[q] = jds12[4 + q];
                        ga.p.inv2s2[q] = 1.f / (2.f * jds12[8 + q] * jds12[8 + q]);
                    }
                    glds = sizeof(float) * 4 * (size_t)N;
                }
                // <float, ODD, BUILDW, TAIL>
                const int form = (odd ? 4 : 0) | (W ? 2 : 0) | (want_tail ? 1 : 0);
                const dim3 grid(nseg), blk(W ? 576 : 256);
                switch (form) {
                    case 0: hipLaunchKernelGGL((mt_gen_kernel<float, false, false, false>), grid, blk, glds, st, ga); break;
                    case 1: hipLaunchKernelGGL((mt_gen_kernel<float, false, false, true>), grid, blk, glds, st, ga); break;
                    case 2: hipLaunchKernelGGL((mt_gen_kernel<float, false, true, false>), grid, blk, glds, st, ga); break;
                    case 3: hipLaunchKernelGGL((mt_gen_kernel<float, false, true, true>), grid, blk, glds, st, ga); break;
                    case 4: hipLaunchKernelGGL((mt_gen_kernel<float, true, false, false>), grid, blk, glds, st, ga); break;
                    case 5: hipLaunchKernelGGL((mt_gen_kernel<float, true, false, true>), grid, blk, glds, st, ga); break;
                    case 6: hipLaunchKernelGGL((mt_gen_kernel<float, true, true, false>), grid, blk, glds, st, ga); break;
                    default: hipLaunchKernelGGL((mt_gen_kernel<float, true, true, true>), grid, blk, glds, st, ga); break;
                }
            } else {
                GenArgs<double> ga{parent, parent_parts, s_lo, s_hi, step << kStride0Log2, b_hi, pos, (long)skip, (long)count, (double*)out};
                if (odd) hipLaunchKernelGGL((mt_gen_kernel<double, true, false, false>), dim3(nseg), dim3(256), 0, st, ga);
                else hipLaunchKernelGGL((mt_gen_kernel<double, false, false, false>), dim3(nseg), dim3(256), 0, st, ga);
            }
            e = hipGetLastError();
            if (e == hipSuccess) { e = hipEventRecord(bs.consumed, st); bs.used = true; }
        }
    }

    return e;
}

// begin + finish: returns when the state after the draw is known
hipError_t mt19937_draw(uint32_t* key, int* pos_io, unsigned long long total, unsigned long long skip, unsigned long long count,
                        void* out, int elem, hipStream_t st) {
    int ticket = -1;
    const hipError_t e = mt19937_begin(key, *pos_io, total, skip, count, out, elem, st, &ticket, nullptr, nullptr, 0);
    if (ticket < 0) return e;
    const hipError_t f = mt19937_finish(ticket, key, pos_io);
    return e != hipSuccess ? e : f;
}

}  // namespace ssn
