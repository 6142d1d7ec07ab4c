// Shared pieces of the two-draw kernels (ssn_duo.hip) and of the fused adjoint + dL/dW sweep (ssn_fuse.hip): the deal of
// W's 16 x 32 units to the waves of a draw, a wave's operands and MFMA chain, the B image's helpers (join of the two
// sum parts, round-to-nearest split, wave maximum), the phase barrier.  See ssn_duo.hip for the design.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "ssn_device.h"
#include "ssn_host.h"
#include "ssn_mfma_io.h"


#ifndef SSN_DUO_EARLY
#define SSN_DUO_EARLY 1         // row tiles a wave finishes right behind its own chain (forward kernels; 0 = none)
#endif
#ifndef SSN_DUO_EARLY_MASK
#define SSN_DUO_EARLY_MASK 0xf  // ... for the waves w of a draw with bit w set (the others finish every tile in the serial phase)
#endif
#ifndef SSN_DUO_PREB
#define SSN_DUO_PREB 0          // forward: 1 = the B operand of the wave's own k tile is read in front of the phase barrier and the
                                // chain starts with that tile (measured at C3, same box, three alternating runs: 3.14 ms against
                                // 3.16 -- the first read behind the barrier is not what stretches a chain; off); 2 = the MFMAs of
                                // that tile as well, at the end of the serial phase (3.17-3.22 against 3.13-3.14: they take
                                // from the partner wave's chain what they save the own one)
#endif
#ifndef SSN_DUO_TEST_EVERY
#define SSN_DUO_TEST_EVERY 1   // > 1 (power of two): TIMING BUILD, wrong stop steps -- see solve serial part in ssn_duo.hip
#endif
#ifndef SSN_DUO_PREV_ALWAYS
#define SSN_DUO_PREV_ALWAYS 0
#endif
#ifndef SSN_DUO_TEST_RING
#define SSN_DUO_TEST_RING 0
#endif
#ifndef SSN_DUO_EARLY_SOLVE
#define SSN_DUO_EARLY_SOLVE 2   // solver: the candidate state of a wave's last row tile behind its own chain: 2 = the four-tile wave only
                                // (27.4 ms at C2 with 8 stimuli against 28.3 with 1 = every wave and 29.1 with 0 = none, same box)
#endif
#ifndef SSN_DUO_EARLY_BWD
#define SSN_DUO_EARLY_BWD 1     // adjoint sweep: the last row tile of a step after the window behind the previous step's chain (0 = off)
#endif
#ifndef SSN_DUO_SOLVE_NL
#define SSN_DUO_SOLVE_NL 4      // solver, 2N > 152: units of every wave whose low part W_m lives in LDS instead of registers
#endif
#ifndef SSN_DUO_BPF
#define SSN_DUO_BPF 1           // chains: the B operand of k tile kk + 1 is requested before the MFMAs of tile kk (0 = compiler's order)
#endif
#ifndef SSN_DUO_ABLATE
#define SSN_DUO_ABLATE 0        // diagnostic builds (timing only, wrong results): 1 = no nonlinearity, 2 = one FMA per MFMA,
                                // 4 / 8 = serial part / chain at s_setprio 1
#endif

#ifndef SSN_DUO_STAMP
#define SSN_DUO_STAMP 0         // diagnostic build: s_memtime stamps of workgroup 0 (waves 0 and 4) summed per segment into
                                // duo_stamps[] (read back by ssn_debug_duo_stamps; tools/time_fwd.py prints them)
#endif


#ifndef SSN_DUO_NL_SAVE
#define SSN_DUO_NL_SAVE 0           // W parts per wave the forward with stores keeps in LDS (2N > 152; 4 until round 4: 5.1-5.3 against 5.02 ms)
#endif
#ifndef SSN_DUO_STORE_LAYOUT
#define SSN_DUO_STORE_LAYOUT 0    // 1: timing experiment, WRONG results -- every wave store 512 contiguous bytes
#endif
#ifndef SSN_DUO_ONEPASS
#define SSN_DUO_ONEPASS 1         // W prologue: 1 = the units fetched for max |W| stay in registers and are split from there (W read once);
                                  // 0 = round 4's two passes (max |W|, barrier, fetch again and split)
#endif
#ifndef SSN_DUO_ONEPASS_SOLVE
#define SSN_DUO_ONEPASS_SOLVE 0   // ... in solve_duo_kernel (2N > 152: the one kernel whose time loop has no register to spare -- with the
                                  // fp32 units alive across barrier (A) the allocator reloads 11 values from scratch in every step)
#endif
#ifndef SSN_DUO_STORE_AUX
#define SSN_DUO_STORE_AUX 0       // cache policy of the trajectory / f' / delta stores (raw buffer store aux: 1 sc0, 2 nt, 16 sc1)
#endif

namespace ssn {

typedef _Float16 hv8 __attribute__((ext_vector_type(8)));
typedef _Float16 hv2 __attribute__((ext_vector_type(2)));
typedef float fv2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int duo_w_exp(unsigned maxbits) {          // a = 14 - floor(log2 max |W|)
    const int biased = (int)((maxbits >> 23) & 0xffu);
    const int a = 14 - ((biased ? biased : 1) - 127);
    return a > 100 ? 100 : (a < -100 ? -100 : a);
}
__device__ __forceinline__ float duo_pow2(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }

template <int MK>
struct Duo16 {
    static constexpr int NRT = (MK + 15) / 16, NKT = (MK + 31) / 32;     // row tiles (16 rows), k tiles (32 columns)
    static constexpr int UNITS = NRT * NKT;
    static constexpr int WM = 4;                                         // waves per draw
    // first unit (rt * NKT + kt) of wave w, row-major deal.  The wave that finishes the odd row tile out (13 row tiles over
    // 4 waves at MK = 208: the last wave finishes four) gets the short share of the matrix work: 23 / 23 / 23 / 22 units.
    static constexpr int start(int w) { return w == WM ? UNITS : (UNITS + WM - 1) / WM * w; }
    static constexpr int BROW = 256;                                     // B image row: (k tile, k octet) x 16 columns x 8 fp16
    static constexpr int BB = NKT * 4 * BROW;
    static constexpr int XS = 1024;                                      // one partial-sum slot: 64 lanes x 16 B
    static constexpr int SYNC = 2 * BB + (WM - 1) * XS;                  // free-running form: [0] finished (wave, step) pairs, [1 + w] steps whose partial sum wave w has stored
    static constexpr int DRAW = SYNC + 32;                               // per draw: two B images (step parity), slots of waves 0 .. WM - 2, sync words
    // Kernels whose serial part needs more registers than the plain forward's (trajectory stores, the solver's stop
    // protocol) keep the low parts (W_m) of the last NL units of every wave in LDS instead of registers (one 16-byte operand
    // per lane and unit, re-read every step) -- otherwise they spill into the time loop.  Measured at C3 / C2 with 8 stimuli:
    // plain forward 3.38 ms with NL = 0 against 3.67 with 4 (the extra reads sit on the chain's critical path, and it did not
    // spill); forward with stores 6.1 -> 5.6 ms, solver 40.2 -> 34.3 ms with NL = 4.  (Round 4: with the chain's B operand one tile
    // ahead and the serial part as it is now, the forward with stores no longer spills in the time loop at NL = 0: 5.02 ms
    // against 5.1-5.3 with 4, SSN_DUO_NL_SAVE.)
    static constexpr int nl(bool heavy) { return (heavy && MK > 152) ? SSN_DUO_NL_SAVE : 0; }
    static constexpr int nl_solve() { return MK > 152 ? SSN_DUO_SOLVE_NL : 0; }
    // adjoint sweep: 10 state values per row and stimulus -- by the number of row tiles a wave finishes (6 or 8 values per lane)
    // (window steps / the steps after the window: the window keeps four trajectory rows per value in registers)
    // (wave 0 of the window loop is the one short of registers: it takes three units from its two three-tile neighbours)
    static constexpr int nl_bwd_win(int wv, int ntf) { return MK > 152 ? (ntf >= 4 ? 20 : (wv == 0 ? 19 : 15)) : (MK > 104 ? 4 : 0); }
    static constexpr int nl_bwd(int ntf, bool gext) { return MK > 152 ? (ntf >= 4 ? 12 : 8) + (gext ? 4 : 0) : (MK > 104 ? 4 : 0); }
    static constexpr int LDS = 2 * DRAW + 16;
    static_assert(start(1) >= NKT, "a row tile is shared by at most two waves");
};
template <int MK, int WV>
struct DuoWave {
    using S = Duo16<MK>;
    static constexpr int U0 = S::start(WV), U1 = S::start(WV + 1), NU = U1 - U0;
    static constexpr int RT0 = U0 / S::NKT, RT1 = (U1 - 1) / S::NKT, NT = RT1 - RT0 + 1;
    static constexpr bool HEAD_SHARED = (U0 % S::NKT) != 0;              // wave WV - 1 holds the head of my first row tile
    static constexpr bool TAIL_SHARED = (U1 % S::NKT) != 0;              // wave WV + 1 holds the tail of my last one and finishes it
    static constexpr int NTF = NT - (TAIL_SHARED ? 1 : 0);               // row tiles RT0 .. RT0 + NTF - 1 are finished here
    static_assert(NTF >= 1, "every wave finishes at least one row tile");
};

// the lane's 8 elements of unit u: W[16 rt + li][32 kt + 8 lg .. + 7], zero outside M x M
__device__ __forceinline__ void duo_fetch(const __amdgpu_buffer_rsrc_t& rsrc, int M, int row, int k0, float (&w)[8]) {
    const int voff = ((row < M ? row : M - 1) * M + k0) * 4;
    const mf4 lo = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
    const mf4 hi = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16, 0, 0));
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = (row < M && k0 + e < M) ? v[e] : 0.f;
}

// the same elements of W^T (adjoint sweep): A[row][k] = W[k][row]; 64-byte runs over the 16 lanes of a row tile
__device__ __forceinline__ void duo_fetch_t(const __amdgpu_buffer_rsrc_t& rsrc, int M, int row, int k0, float (&w)[8]) {
    const int rowc = row < M ? row : M - 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e < M ? k0 + e : M - 1;
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (k * M + rowc) * 4, 0, 0));
        w[e] = (row < M && k0 + e < M) ? v : 0.f;
    }
}

__device__ __forceinline__ float dpp_ror8(float x) {                  // lane li of a 16-lane row <- lane (li + 8) % 16
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
}

// Join of the two parts of a row tile's sums.  After the chain lanes s and 8 + s of a 16-lane row hold, for rows 0 .. 3 of the
// tile (registers x, y, z, w), the products with the two fp16 parts of stimulus s; lane s (DPP banks 0, 1 of its row)
// finishes rows 0, 1 and lane 8 + s (banks 2, 3) rows 2, 3: value e of a lane is lo_e + ror8(lo_e) in banks 0, 1 and
// hi_e + ror8(hi_e) in banks 2, 3.  Two DPP adds with complementary bank masks write each half of the row from its own
// register -- no selects (the builtin form, `hi ? z : x` twice and one DPP add, is three instructions per value).
// Inline asm hides the instructions from the compiler's hazard recognizer, so the wait states are spelled out: FIRST = the
// first join after a chain or after compiler-generated VALU writes of the operands: 11 wait states cover a finished
// 8-pass MFMA -> VALU read and the 2 a VALU write -> DPP read needs.  The joins are volatile: they keep their order.
// MEASURED (C3 forward, same box, both builds loaded in alternation): 3.20 ms with the asm form, 3.21 ms with the builtin
// form -- 7 instructions fewer per serial part (of ~130) buy nothing, the wait states cost what the selects did.  The
// builtin form is the default; -DSSN_DUO_JOIN_ASM=1 builds the other one (parity-green on the same tests).
#ifndef SSN_DUO_JOIN_ASM
#define SSN_DUO_JOIN_ASM 0
#endif
// (`also0/1`: the operands of the tile's SECOND join, named as inputs of the first so that compiler-generated writes
// of them -- the partial sum of a neighbour wave added to a shared tile -- sit in front of the wait states as well.)
template <bool FIRST>
__device__ __forceinline__ float duo_join(float lo, float hi, int is_hi, float also0 = 0.f, float also1 = 0.f) {
#if SSN_DUO_JOIN_ASM
    float r;
    if (FIRST) asm volatile("s_nop 10\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3"
                            : "=v"(r) : "v"(lo), "v"(hi), "v"(also0), "v"(also1));
    else asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "=v"(r) : "v"(lo));
    asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(r) : "v"(hi));
    return r;
#else
    const float k = is_hi ? hi : lo, o = is_hi ? lo : hi;
    return k + dpp_ror8(o);
#endif
}

// max over the wave of a non-negative float (bit patterns order like the values), the same value in every lane's SGPR copy
__device__ __forceinline__ unsigned duo_wave_max_bits(float x) {
    unsigned v = __builtin_bit_cast(unsigned, x);
#define SSN_DUO_DPP_MAX(CTRL, ROWMASK)                                                                                   \
    {                                                                                                                     \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);                   \
        v = o > v ? o : v;                                                                                                \
    }
    SSN_DUO_DPP_MAX(0xB1, 0xf)     // quad_perm [1,0,3,2]
    SSN_DUO_DPP_MAX(0x4E, 0xf)     // quad_perm [2,3,0,1]
    SSN_DUO_DPP_MAX(0x141, 0xf)    // row_half_mirror
    SSN_DUO_DPP_MAX(0x140, 0xf)    // row_mirror: every lane = row max
    SSN_DUO_DPP_MAX(0x142, 0xa)    // row_bcast15 -> rows 1, 3
    SSN_DUO_DPP_MAX(0x143, 0xc)    // row_bcast31 -> rows 2, 3
#undef SSN_DUO_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// x 2^rshift = h + m by round to nearest, two values per call; returns the packed fp16 pairs
// (plain fp32 instructions on purpose: beside a partner wave's MFMA stream a packed v_pk_*_f32 costs several times its
// two scalar halves -- MI355X_MICROARCH.md, cycle constants; this file is built with -fno-slp-vectorize for the same reason)
__device__ __forceinline__ void duo_split2(float x0, float x1, float rs, unsigned& h, unsigned& m) {
    const float s0 = x0 * rs, s1 = x1 * rs;
    const hv2 hh = __builtin_convertvector((fv2){s0, s1}, hv2);
    const float d0 = s0 - (float)hh[0], d1 = s1 - (float)hh[1];
    h = __builtin_bit_cast(unsigned, hh);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector((fv2){d0, d1}, hv2));
}

// A wave's share of W (two fp16 parts of W 2^a) and its MFMA chain.
template <int MK, int WV, int NL, bool TR = false>
struct DuoOperands {
    using S = Duo16<MK>;
    using WS = DuoWave<MK, WV>;
    static constexpr int NU = WS::NU, NT = WS::NT, RT0 = WS::RT0, U0 = WS::U0, U1 = WS::U1, NR = NU - NL;
    using LdsH8 = __attribute__((address_space(3))) hv8*;
    hv8 Ah[NU], Am[NR > 0 ? NR : 1];
    unsigned wl;                                  // LDS byte address of this lane's slot of the first LDS-resident unit

    // pass 1 over the wave's units: max |W|
    static __device__ __forceinline__ float max_abs(const __amdgpu_buffer_rsrc_t& rsrc, int M, int li, int lg) {
        float mx = 0.f;
        for (int u = U0; u < U1; ++u) {
            float w[8];
            if (TR) duo_fetch_t(rsrc, M, 16 * (u / S::NKT) + li, 32 * (u % S::NKT) + 8 * lg, w);
            else duo_fetch(rsrc, M, 16 * (u / S::NKT) + li, 32 * (u % S::NKT) + 8 * lg, w);
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, __builtin_fabsf(w[e]));
        }
        return mx;
    }
    // ONE pass over W (SSN_DUO_ONEPASS, default): the wave's units stay in registers as fp32 from the fetch that finds max |W|
    // (NU x 8 = 176-184 registers, the size of the two fp16 parts they become: each unit's 8 registers turn into its
    // 4 + 4), so the draw's W crosses HBM once per launch instead of twice.  Same loads, same arithmetic, same bits.
    struct Raw { float w[NU][8]; };
    static __device__ __forceinline__ float fetch(const __amdgpu_buffer_rsrc_t& rsrc, int M, int li, int lg, Raw& raw) {
        float mx = 0.f;
#pragma unroll
        for (int ui = 0; ui < NU; ++ui) {
            if (TR) duo_fetch_t(rsrc, M, 16 * ((U0 + ui) / S::NKT) + li, 32 * ((U0 + ui) % S::NKT) + 8 * lg, raw.w[ui]);
            else duo_fetch(rsrc, M, 16 * ((U0 + ui) / S::NKT) + li, 32 * ((U0 + ui) % S::NKT) + 8 * lg, raw.w[ui]);
        }
#pragma unroll
        for (int ui = 0; ui < NU; ++ui)
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, __builtin_fabsf(raw.w[ui][e]));
        return mx;
    }
    __device__ __forceinline__ void split(const Raw& raw, float sa, char* wlds, int lane) {
        wl = (unsigned)(size_t)(LdsH8)wlds + (unsigned)(lane * 16);
#pragma unroll
        for (int ui = 0; ui < NU; ++ui) {
            hv8 m;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sc = raw.w[ui][e] * sa;
                const _Float16 h = (_Float16)sc;
                Ah[ui][e] = h;
                m[e] = (_Float16)(sc - (float)h);
            }
            if (ui < NR) Am[ui < NR ? ui : 0] = m;
            else *(LdsH8)(size_t)(wl + (unsigned)((ui - NR) * 1024)) = m;
        }
    }
    // pass 2: W 2^a = W_h + W_m by round to nearest
    __device__ __forceinline__ void load(const __amdgpu_buffer_rsrc_t& rsrc, int M, int li, int lg, float sa, char* wlds, int lane) {
        wl = (unsigned)(size_t)(LdsH8)wlds + (unsigned)(lane * 16);
#pragma unroll
        for (int ui = 0; ui < NU; ++ui) {
            float w[8];
            if (TR) duo_fetch_t(rsrc, M, 16 * ((U0 + ui) / S::NKT) + li, 32 * ((U0 + ui) % S::NKT) + 8 * lg, w);
            else duo_fetch(rsrc, M, 16 * ((U0 + ui) / S::NKT) + li, 32 * ((U0 + ui) % S::NKT) + 8 * lg, w);
            hv8 m;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sc = w[e] * sa;
                const _Float16 h = (_Float16)sc;
                Ah[ui][e] = h;
                m[e] = (_Float16)(sc - (float)h);
            }
            if (ui < NR) Am[ui < NR ? ui : 0] = m;
            else *(LdsH8)(size_t)(wl + (unsigned)((ui - NR) * 1024)) = m;
        }
    }
    // Take over the operands of `o`, which keeps MORE units in LDS (NL2 >= NL), and pull the difference into registers
    // (adjoint sweep: the penalty-window steps need the registers for trajectory rows, the steps after them do not).
    template <int NL2>
    __device__ __forceinline__ void promote_from(const DuoOperands<MK, WV, NL2, TR>& o) {
        static_assert(NL2 >= NL, "promotion only moves units from LDS into registers");
        constexpr int NR2 = NU - NL2;
#pragma unroll
        for (int ui = 0; ui < NU; ++ui) Ah[ui] = o.Ah[ui];
#pragma unroll
        for (int ui = 0; ui < NR; ++ui) {
            if (ui < NR2) Am[ui] = o.Am[ui < NR2 ? ui : 0];
            else Am[ui] = *(LdsH8)(size_t)(o.wl + (unsigned)((ui - NR2) * 1024));
        }
        wl = o.wl + (unsigned)((NR - NR2) * 1024);
    }
    // The k tile whose B rows (row tiles 2 kt and 2 kt + 1) this wave finishes and publishes ITSELF, or -1: its B operand can
    // be read back right after the wave's own stores (the DS instructions of one wave execute in order), in front of the
    // barrier that the other k tiles have to wait for.
    static constexpr int own_kt() {
        for (int kt = 0; kt < S::NKT; ++kt) {
            const int r0 = 2 * kt, r1 = 2 * kt + 1;
            const bool in0 = r0 >= RT0 && r0 < RT0 + WS::NTF;
            const bool in1 = r1 >= S::NRT || (r1 >= RT0 && r1 < RT0 + WS::NTF);
            if (in0 && in1) return kt;
        }
        return -1;
    }
    static __device__ __forceinline__ hv8 read_b(unsigned rd, int kt) {
        using LdsB = const __attribute__((address_space(3))) hv8*;
        return *(LdsB)(size_t)(rd + (unsigned)(kt * 4 * S::BROW));
    }
    // acc[t] = (W_h + W_m)[row tile RT0 + t, my k range] . B, B read from the image at LDS byte address rd (per lane).
    // ROT: the k tiles in the order own_kt(), own_kt() + 1, ... (mod NKT); PRE: the first of them comes in `bpre`.
    // KA .. KB: the part of that order to run (the whole chain by default; SSN_DUO_PREB = 2 runs [0, 1) -- the wave's own k
    // tile -- in front of the barrier, at the end of the serial phase, and [1, NKT) behind it); the sums start at zero with KA = 0.
    template <bool ROT = false, bool PRE = false, int KA = 0, int KB = S::NKT>
    __device__ __forceinline__ void chain(unsigned rd, mf4 (&acc)[NT], const hv8& bpre = hv8{}) const {
        using LdsB = const __attribute__((address_space(3))) hv8*;
        constexpr int K0 = (ROT && own_kt() >= 0) ? own_kt() : 0;
        if constexpr (KA == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = (mf4){0.f, 0.f, 0.f, 0.f};
        }
        // B operands one k tile AHEAD of the MFMAs that take them (SSN_DUO_BPF): left to itself the compiler issues a tile's
        // ds_read_b128 right in front of its first MFMA and waits (seen in the assembly: read, s_waitcnt lgkmcnt(0), six MFMAs,
        // seven times per chain in the register-bound kernels) -- ~100 cycles of LDS latency per k tile with the matrix pipe idle.
        // The scheduling fence keeps the prefetch in front of the tile's MFMAs; the wait it needs is a counted one, one tile later.
        hv8 bnext = hv8{};
        if constexpr (SSN_DUO_BPF && KA < KB) {
            if (!(PRE && ROT && own_kt() >= 0 && KA == 0)) bnext = read_b(rd, (K0 + KA) % S::NKT);
        }
#pragma unroll
        for (int kk = KA; kk < KB; ++kk) {
            const int kt = (K0 + kk) % S::NKT;
            hv8 b1;
            if constexpr (SSN_DUO_BPF) {
                b1 = (PRE && ROT && own_kt() >= 0 && kk == 0) ? bpre : bnext;
                if (kk + 1 < KB) bnext = read_b(rd, (K0 + kk + 1) % S::NKT);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                b1 = (PRE && ROT && own_kt() >= 0 && kk == 0) ? bpre : read_b(rd, kt);
            }
#pragma unroll
            for (int part = 0; part < 2; ++part) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int u = (RT0 + t) * S::NKT + kt;
                    if (u >= U0 && u < U1) {
                        const int ui = u - U0;
                        hv8 aop;
                        if (part == 0) aop = Ah[ui];
                        else if (ui < NR) aop = Am[ui < NR ? ui : 0];
                        else aop = *(LdsB)(size_t)(wl + (unsigned)((ui - NR) * 1024));
                        if (SSN_DUO_ABLATE & 2) acc[t].x += (float)aop[0] * (float)b1[0];   // (one FMA per MFMA)
                        else acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aop, b1, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
};

// Hand-over words in LDS (free-running form).  DS instructions of one wave execute in program order, so a counter bumped
// after the data stores of the same wave is seen only after them; the waiting side reads the word, then the data.
__device__ __forceinline__ void duo_signal_add(unsigned addr, int lane) {
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(1u) : "memory");
}
__device__ __forceinline__ void duo_signal_set(unsigned addr, unsigned value, int lane) {
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(value) : "memory");
}
// waits until the word at `addr` has reached `target` (monotonic counters); bounded: after ~2^16 polls the wave stops waiting
// for good (`dead`), runs to the end without further waits and poisons its outputs -- a lost wake-up must not hang the chip
__device__ __forceinline__ void duo_wait_ge(unsigned addr, int target, bool& dead) {
    if (dead) return;
    for (int spin = 0;; ++spin) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)v) - (unsigned)target) >= 0) return;
        if (spin >= (1 << 16)) { dead = true; return; }
        __builtin_amdgcn_s_sleep(1);
    }
}

// f(u), f'(u) for the NE values of a lane; eight values go through in two halves (fewer values in flight at once: the
// wave that finishes four row tiles is the one short of registers)
// The barrier between two phases.  MFMAs touch registers only, so the scheduler is free to sink the tail of a chain below
// an s_barrier and interleave it with the serial part -- where it waits for a matrix pipe the partner wave's chain is
// filling, with the wave's own vector stream stuck in order behind it (seen in the assembly: up to 21 of 46 MFMAs moved).
// Nothing crosses this one.
__device__ __forceinline__ void duo_phase_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
}

template <bool WANT_DF, int NE>
__device__ __forceinline__ void duo_eval(const IoSelect& io, const float (&uu)[NE], float (&ff)[NE], float (&dfn)[NE]) {
    if constexpr (NE > 6) {                  // 7 or 8 values: 4 first, then the rest (fewer registers pinned at once)
        float u4[4], f4[4], d4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { u4[i] = uu[i]; d4[i] = 0.f; }
        io.template evaln<WANT_DF, 4>(u4, f4, d4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { ff[i] = f4[i]; dfn[i] = d4[i]; }
        constexpr int NR = NE - 4;
        float ur[NR], fr[NR], dr[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) { ur[i] = uu[4 + i]; dr[i] = 0.f; }
        io.template evaln<WANT_DF, NR>(ur, fr, dr);
#pragma unroll
        for (int i = 0; i < NR; ++i) { ff[4 + i] = fr[i]; dfn[4 + i] = dr[i]; }
    } else {
        io.template evaln<WANT_DF, NE>(uu, ff, dfn);
    }
}

// Half-real tail tile (2N = 194 ... 200 at MK = 208: the 13th row tile holds at most 8 real rows, all in lane groups 0, 1).
// The wave that finishes it would carry 8 values per lane, 2 of them padding in EVERY lane, and its serial part is the
// length of a phase.  In the `HT` forms it finishes that tile one value per lane instead: after the DPP join lanes 0-31
// hold the real row pairs (a, b); v_permlane32_swap hands b of lane l to lane 32 + l, so lane (lg, hi, st) finishes row
// 4 (lg & 1) + 2 hi + (lg >> 1) of the tile -- 7 values per lane instead of 8 (6 / 6 / 6 / 7 over the four waves = 1600 / 256).
__device__ __forceinline__ float duo_tail_take(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, r[0]);                 // lanes 0-31: a, lanes 32-63: b of lane - 32
}
constexpr bool duo_half_tail(int MK, int M) { return MK == 208 && M > 192 && M <= 200; }

}  // namespace ssn
