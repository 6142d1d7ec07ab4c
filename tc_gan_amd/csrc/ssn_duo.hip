// Two weight draws per workgroup, every wave both roles: the SSN recurrence for NB >= 4 stimuli per draw on the fp16
// matrix cores (exact-product split, see ssn_mfma16.hip) with the chain of one draw hidden behind the serial part of
// the other.
//
// ssn_mfma16.hip's wide form puts all 8 stimuli of a draw into the 16 operand columns of v_mfma_f32_16x16x32_f16 (column s
// = r_h, column 8 + s = r_m of stimulus s), which halves the matrix work of the alternating form but leaves nothing to
// alternate with: a step there is [chain] barrier [serial part] barrier, matrix waves and serial waves take turns and the
// matrix pipe idles two thirds of the time.  Here the second unit of work is a second DRAW:
//
//   workgroup = 8 waves = 2 draws x 4 waves; waves w and w + 4 share a SIMD and belong to different draws;
//   every wave holds its quarter of its draw's W (two fp16 parts, 22-23 tiles of 16 x 32, 184 registers) for the whole
//   launch AND finishes the rows of the row tiles it completes: in even phases the waves of draw 0 run their chains
//   while the waves of draw 1 run their serial parts (nonlinearity, Euler update, reductions, split of the new state),
//   in odd phases the other way round, ONE s_barrier per phase.  Each SIMD therefore always has one wave on the matrix
//   pipe and one on the vector pipe.
//
//   No round trip of the sums through LDS: the wave that ran a row tile's chain keeps the accumulators in registers
//   across the barrier and finishes them itself.  In the accumulator layout lane (lg, li) holds rows 4 lg .. 4 lg + 3 of
//   column li; columns s and 8 + s are the two parts of stimulus s, so one DPP row rotate by 8 adds them and the two
//   lanes share the four rows: lane s finishes rows 4 lg, 4 lg + 1, lane 8 + s rows 4 lg + 2, 4 lg + 3 -- every lane two
//   rows per row tile, 6 or 8 values per step.  Only a row tile whose k range is split between two neighbouring waves
//   (3 of 13 at 2N = 200, the price of an even deal of the 91 tiles) sends one partial sum through LDS, written at the end
//   of the chain phase and read behind the barrier in the serial phase of the same draw.
//   The new state goes to LDS as two fp16 parts by ROUND TO NEAREST (h = rn(s), m = rn(s - h): |s - h - m| <= 2^-23 |s|;
//   v_cvt_pk_f16_f32), where the four waves of the draw read it as their B operand in the next chain phase.
//
// Same arithmetic per step as the other forms (accumulation fp32, W = W_h + W_m by round to nearest, one power-of-two scale
// per draw from max |W|, state scale from the rate bound of the saturating I/O function: asym_tanh only).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "ssn_device.h"
#include "ssn_host.h"
#include "ssn_mfma_io.h"

#ifndef SSN_DUO_ABLATE
#define SSN_DUO_ABLATE 0        // diagnostic builds (timing only, wrong results): 1 = no nonlinearity, 2 = one FMA per MFMA,
                                // 4 / 8 = serial part / chain at s_setprio 1
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

__device__ __forceinline__ float dpp_ror8(float x) {                  // lane li of a 16-lane row <- lane (li + 8) % 16
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
}

// x 2^rshift = h + m by round to nearest, two values per call; returns the packed fp16 pairs
__device__ __forceinline__ void duo_split2(float x0, float x1, float rs, unsigned& h, unsigned& m) {
    const fv2 s = (fv2){x0, x1} * (fv2){rs, rs};
    const hv2 hh = __builtin_convertvector(s, hv2);
    const fv2 d = s - __builtin_convertvector(hh, fv2);
    h = __builtin_bit_cast(unsigned, hh);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(d, hv2));
}

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

template <int MK, int WV, bool SAVE, bool FREE>
__device__ __forceinline__ void duo_forward_wave(const GenFwdArgs<float>& a, int rshift, int d, int b, int s0, bool valid,
                                                 int lane, char* dlds, unsigned* wmax) {
    using S = Duo16<MK>;
    using WS = DuoWave<MK, WV>;
    constexpr int NU = WS::NU, NT = WS::NT, NTF = WS::NTF, RT0 = WS::RT0, U0 = WS::U0, U1 = WS::U1;
    constexpr int NE = 2 * NTF;
    const int M = a.M, N = a.M / 2, T_ = a.seqlen;
    const int li = lane & 15, lg = lane >> 4, hi = li >> 3, st = li & 7;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W + (size_t)b * M * M), 0, M * M * 4, 0x00020000);
    // ---- W: pass 1 = max |W| of the draw, pass 2 = the two fp16 parts of W 2^a
    float mx = 0.f;
    for (int u = U0; u < U1; ++u) {
        float w[8];
        duo_fetch(rsrc, M, 16 * (u / S::NKT) + li, 32 * (u % S::NKT) + 8 * lg, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, __builtin_fabsf(w[e]));
    }
    atomicMax(wmax, __builtin_bit_cast(unsigned, mx));
    __syncthreads();                                                          // (A)
    const int wexp = duo_w_exp(*wmax);
    const float sa = duo_pow2(wexp), usc = duo_pow2(-wexp - rshift), rs = duo_pow2(rshift);
    hv8 Ah[NU], Am[NU];
#pragma unroll
    for (int ui = 0; ui < NU; ++ui) {
        float w[8];
        duo_fetch(rsrc, M, 16 * ((U0 + ui) / S::NKT) + li, 32 * ((U0 + ui) % S::NKT) + 8 * lg, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sc = w[e] * sa;
            const _Float16 h = (_Float16)sc;
            Ah[ui][e] = h;
            Am[ui][e] = (_Float16)(sc - (float)h);
        }
    }
    // ---- the values this lane finishes: row tile RT0 + tf, rows 4 lg + 2 hi + e, stimulus s0 + st
    const IoSelect io(a.io);
    const int s = s0 + st;
    const bool live = valid && s < a.NB;
    float rc[NE], ex[NE], eps[NE], ta[NE];
    float dps = 0.f, rps = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int row = 16 * (RT0 + i / 2) + 4 * lg + 2 * hi + (i & 1);
        rc[i] = ta[i] = 0.f;
        ex[i] = (s < a.NB && row < M) ? a.ext[((size_t)b * a.NB + s) * M + row] : 0.f;
        eps[i] = row < N ? a.eps_E : a.eps_I;
    }
    const size_t blk_elems = (size_t)a.NB * T_ * M;
    __amdgpu_buffer_rsrc_t rs_traj, rs_df;
    int toff = -1;                               // byte offset of (my stimulus, step 0, row 4 lg + 2 hi) within this draw's block
    if constexpr (SAVE) {
        rs_traj = __builtin_amdgcn_make_buffer_rsrc(a.traj + (size_t)b * blk_elems, 0, (int)(blk_elems * 4), 0x00020000);
        rs_df = __builtin_amdgcn_make_buffer_rsrc(a.df + (size_t)b * blk_elems, 0, (int)(blk_elems * 4), 0x00020000);
        toff = live ? (int)(((size_t)s * T_ * M + 4 * lg + 2 * hi) * 4) : -1;
    }
    using LdsH8 = const __attribute__((address_space(3))) hv8*;
    using LdsF4 = __attribute__((address_space(3))) mf4*;
    using LdsU = __attribute__((address_space(3))) unsigned*;
    const unsigned bimg = (unsigned)(size_t)(LdsH8)dlds;
    const unsigned xs = bimg + (unsigned)(2 * S::BB) + (unsigned)(lane * 16);             // + slot * XS
    const unsigned b_rd = bimg + (unsigned)(lg * S::BROW + li * 16);                      // + kt * 4 * BROW
    // new state of rows 16 rt + 4 lg + 2 hi + {0, 1}: k tile rt / 2, k octet 2 (rt & 1) + lg / 2, element 4 (lg & 1) + 2 hi
    const unsigned b_wr = bimg + (unsigned)((lg >> 1) * S::BROW + st * 16 + (lg & 1) * 8 + hi * 4);

    const unsigned sync = bimg + (unsigned)S::SYNC;
    bool dead = false;
    mf4 acc[NT];
    auto chain = [&](int it) {
        // free-running form: all four waves of the draw must have stored the state of step it - 1 (image it & 1)
        if (FREE && it > 0) duo_wait_ge(sync, S::WM * it, dead);
        const unsigned rd = b_rd + (FREE ? (unsigned)((it & 1) * S::BB) : 0u);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (mf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < S::NKT; ++kt) {
            const hv8 b1 = *(LdsH8)(size_t)(rd + (unsigned)(kt * 4 * S::BROW));
#pragma unroll
            for (int part = 0; part < 2; ++part) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int u = (RT0 + t) * S::NKT + kt;
                    if (u >= U0 && u < U1) {
                        if (SSN_DUO_ABLATE & 2) acc[t].x += (float)(part ? Am[u - U0] : Ah[u - U0])[0] * (float)b1[0];   // (one FMA per MFMA)
                        else acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(part ? Am[u - U0] : Ah[u - U0], b1, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (WS::TAIL_SHARED) {
            *(LdsF4)(size_t)(xs + (unsigned)(WV * S::XS)) = acc[NT - 1];
            if (FREE) duo_signal_set(sync + 4u * (1 + WV), (unsigned)(it + 1), lane);
        }
    };
    auto serial = [&](auto WIN, int it) {
        constexpr bool win_on = decltype(WIN)::value;
        float uu[NE], ff[NE], dfn[NE];
#pragma unroll
        for (int tf = 0; tf < NTF; ++tf) {
            mf4 sm = acc[tf];
            if constexpr (WS::HEAD_SHARED) {
                if (tf == 0) {
                    if (FREE) duo_wait_ge(sync + 4u * WV, it + 1, dead);          // wave WV - 1 has stored its partial sum of step it
                    sm += *(LdsF4)(size_t)(xs + (unsigned)((WV - 1) * S::XS));
                }
            }
            // lane s (hi = 0) keeps rows 0, 1 and offers rows 2, 3 of its part; lane 8 + s keeps 2, 3 and offers 0, 1
            const float k0 = hi ? sm.z : sm.x, k1 = hi ? sm.w : sm.y;
            const float o0 = hi ? sm.x : sm.z, o1 = hi ? sm.y : sm.w;
            uu[2 * tf] = fmaf(k0 + dpp_ror8(o0), usc, ex[2 * tf]);
            uu[2 * tf + 1] = fmaf(k1 + dpp_ror8(o1), usc, ex[2 * tf + 1]);
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) dfn[i] = 0.f;
        if (SSN_DUO_ABLATE & 1) {                    // (no nonlinearity: the rest of the serial part stays)
#pragma unroll
            for (int i = 0; i < NE; ++i) ff[i] = uu[i];
        } else {
            io.template evaln<SAVE, NE>(uu, ff, dfn);
        }
        const float win2 = (it > a.skip) ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const float r1 = fmaf(eps[i], ff[i] - rc[i], rc[i]);                 // (1 - eps) r + eps f(u)
            const float dd = r1 - rc[i];
            if constexpr (win_on) {
                ta[i] += r1;
                rps += fmaxf(r1 - a.theta, 0.f);
                dps = fmaf(win2 * dd, dd, dps);
            }
            rc[i] = r1;
        }
#pragma unroll
        for (int tf = 0; tf < NTF; ++tf) {
            constexpr int dummy = 0; (void)dummy;
            const int rt = RT0 + tf;
            if constexpr (SAVE) {
                const int off = (toff < 0 || 16 * rt + 4 * lg + 2 * hi >= M) ? -1 : toff + (it * M + 16 * rt) * 4;
                const fv2 rv = {rc[2 * tf], rc[2 * tf + 1]}, dv = {dfn[2 * tf], dfn[2 * tf + 1]};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(unsigned __attribute__((ext_vector_type(2))), rv), rs_traj, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(unsigned __attribute__((ext_vector_type(2))), dv), rs_df, off, 0, 0);
            }
            unsigned h, m;
            duo_split2(rc[2 * tf], rc[2 * tf + 1], rs, h, m);
            const unsigned wr = b_wr + (unsigned)(((rt >> 1) * 4 + 2 * (rt & 1)) * S::BROW) + (FREE ? (unsigned)(((it + 1) & 1) * S::BB) : 0u);
            *(LdsU)(size_t)wr = h;
            *(LdsU)(size_t)(wr + 128u) = m;
        }
        if (FREE) duo_signal_add(sync, lane);
    };
    constexpr std::integral_constant<bool, false> W0{};
    constexpr std::integral_constant<bool, true> W1{};
    const int nskip = a.skip < T_ ? (a.skip > 0 ? a.skip : 0) : T_;
    __syncthreads();                                                          // (B)
    if constexpr (FREE) {
        // no workgroup barrier from here on: the four waves of a draw meet at their own counters, the two draws drift
        for (int it = 0; it < nskip; ++it) { chain(it); serial(W0, it); }
        for (int it = nskip; it < T_; ++it) { chain(it); serial(W1, it); }
    } else {
        if (d) __syncthreads();                        // draw 1 runs one phase behind draw 0
        for (int it = 0; it < nskip; ++it) {
            chain(it);
            __syncthreads();
            serial(W0, it);
            __syncthreads();
        }
        for (int it = nskip; it < T_; ++it) {
            chain(it);
            __syncthreads();
            serial(W1, it);
            __syncthreads();
        }
        if (!d) __syncthreads();
    }

    if (!live) return;
    const float inv = dead ? __builtin_nanf("") : 1.f / (float)(T_ - a.skip);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int row = 16 * (RT0 + i / 2) + 4 * lg + 2 * hi + (i & 1);
        if (row >= M) continue;
        const size_t o = ((size_t)b * a.NB + s) * M + row;
        a.time_avg[o] = ta[i] * inv;
        // window sums of (x_{t+1} - x_t)^2 and relu(x_t - theta): this lane's total over its rows, booked on its first row
        a.dyn_row[o] = i == 0 ? dps : 0.f;
        a.rate_row[o] = i == 0 ? rps : 0.f;
    }
}

// grid: ceil(units / 2) workgroups, unit = (draw, group of 8 stimuli); 512 threads
template <int MK, bool SAVE, bool FREE>
__global__ void __launch_bounds__(512, 2) gen_forward_duo_kernel(GenFwdArgs<float> a, int rshift) {
    using S = Duo16<MK>;
    __shared__ __align__(16) char lds[S::LDS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (uniform: buffer descriptors stay in SGPRs)
    const int d = wave >> 2;
    const int ngroups = (a.NB + 7) / 8;
    const long nunits = (long)a.B * ngroups;
    long unit = 2L * blockIdx.x + d;
    const bool valid = unit < nunits;
    if (!valid) unit = nunits - 1;                  // an odd unit count: the idle half repeats the last unit and stores nothing
    const int b = (int)(unit / ngroups), s0 = (int)(unit % ngroups) * 8;
    for (int c = threadIdx.x; c < S::LDS / 4; c += blockDim.x) reinterpret_cast<unsigned*>(lds)[c] = 0u;
    __syncthreads();                                // (zeroed before any wave records max |W|; state 0 = the B images)
    char* const dlds = lds + d * S::DRAW;
    unsigned* const wmax = reinterpret_cast<unsigned*>(lds + 2 * S::DRAW) + d;
    switch (wave & 3) {
        case 0: duo_forward_wave<MK, 0, SAVE, FREE>(a, rshift, d, b, s0, valid, lane, dlds, wmax); break;
        case 1: duo_forward_wave<MK, 1, SAVE, FREE>(a, rshift, d, b, s0, valid, lane, dlds, wmax); break;
        case 2: duo_forward_wave<MK, 2, SAVE, FREE>(a, rshift, d, b, s0, valid, lane, dlds, wmax); break;
        default: duo_forward_wave<MK, 3, SAVE, FREE>(a, rshift, d, b, s0, valid, lane, dlds, wmax); break;
    }
}

// SSN_DUO_FREE=0: the lock-step form (one workgroup barrier per phase) instead of the free-running one (A/B timing)
static bool duo_free_running() {
    static const bool on = [] { const char* e = getenv("SSN_DUO_FREE"); return !(e && e[0] == '0'); }();
    return on;
}
static int duo_pick_mk(int M) {
    const int ladder[] = {104, 152, 208};
    for (int mk : ladder) if (M <= mk) return mk;
    return 0;
}

template <int MK>
static hipError_t launch_duo_fwd_mk(const GenFwdArgs<float>& a, int rshift, hipStream_t st) {
    const long nunits = (long)a.B * ((a.NB + 7) / 8);
    const dim3 grid((unsigned)((nunits + 1) / 2));
    if (duo_free_running()) {
        if (a.traj) hipLaunchKernelGGL((gen_forward_duo_kernel<MK, true, true>), grid, dim3(512), 0, st, a, rshift);
        else hipLaunchKernelGGL((gen_forward_duo_kernel<MK, false, true>), grid, dim3(512), 0, st, a, rshift);
    } else {
        if (a.traj) hipLaunchKernelGGL((gen_forward_duo_kernel<MK, true, false>), grid, dim3(512), 0, st, a, rshift);
        else hipLaunchKernelGGL((gen_forward_duo_kernel<MK, false, false>), grid, dim3(512), 0, st, a, rshift);
    }
    return hipGetLastError();
}
// rshift from gen_split_rshift (ssn_mfma16.hip): the same applicability rules as the other fp16-split forms
hipError_t launch_gen_forward_duo(const GenFwdArgs<float>& a, hipStream_t st) {
    const int rshift = gen_split_rshift(a);
    if (rshift < 0) return hipErrorInvalidValue;
    switch (duo_pick_mk(a.M)) {
        case 104: return launch_duo_fwd_mk<104>(a, rshift, st);
        case 152: return launch_duo_fwd_mk<152>(a, rshift, st);
        case 208: return launch_duo_fwd_mk<208>(a, rshift, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ssn
