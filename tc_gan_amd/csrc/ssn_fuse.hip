// The BPTT adjoint sweep with dL/dW accumulated ON CHIP (round 4): one launch in place of gen_backward_duo_kernel (ssn_duo.hip) +
// gw_split_kernel (ssn_gw.hip); the delta stream between them (7.9 GB at the C3 shape, written by one, read back with the
// trajectory by the other) never exists.  Opt-in (ssn_gen_backward_fused_f32, --gen-kernel duo-fused): measured it TIES the
// two launches at small draw counts and is 7 % slower at 1024 draws (8.93 against 8.3-8.4 ms), DESIGN 3.7d says why.
//
//   workgroup = 4 waves = ONE draw, one wave per SIMD, so a wave may use the whole 512-entry register file: its quarter of
//   W^T as fp16 parts (SSN_FUSE_NPREG 16 x 32 parts in registers, the rest in LDS) AND a strip of the 13 x 13 grid of 16 x 16
//   accumulator tiles of dL/dW (FuseTiles: 43 / 42 / 42 / 42 tiles = 172 accumulator registers), which stay there for the
//   whole sweep.  Per step tau, two phases, one s_barrier each:
//     A: serial part of step tau (as gen_backward_duo: join of the chain's sums, delta_tau = eps f'(u_tau) a_tau, carry,
//        lagged power-of-two scale, delta_tau as two fp16 parts into the B image of this step's chain) + x_{tau-1} as two
//        fp16 parts into a second image of the same layout + the rank-8 update of step tau + 1:
//            gW[i][j] += sum_s delta_{tau+1}[s][i] x_tau[s][j]
//        = one v_mfma_f32_16x16x32_f16 per tile: K = 32 = 8 stimuli x {d_h x_h, d_h x_m, d_m x_h, d_m x_m}, issued one at a
//        time at fenced sites spread over the serial part's own instruction stream;
//     B: chain W^T delta_tau (46 MFMAs per wave).
//   Both operands of the update are per-neuron vectors over (part, stimulus), the transpose of what the images hold
//   ([column = 8 part + stimulus][8 neurons]): ds_read_b64_tr_b16 delivers them transposed, two reads per operand, no
//   shuffles and no second copy of delta (tools/microbench/tr_read_check.hip pins the address map on the hardware).
//
// Scales.  The chain's image of delta_tau carries the lagged scale 2^bexp(tau) of the two-draw sweep (max |delta_{tau+1}|
// at 2^7).  The accumulators need ONE scale per draw for the whole sum, so they follow the running minimum sexp of bexp
// (|delta| growing backwards in time lowers it); the factor 2^(sexp - bexp) <= 1 goes into the x image of the same step
// (x 2^(xexp + sexp - bexp), xexp from the rate bound of the saturating I/O function) -- a step whose delta is far below
// the largest one seen so far loses low bits of a product that is negligible against the sum -- and when sexp drops, the
// accumulators are multiplied by the power of two once (a handful of times per sweep).  At the end
// gW = acc 2^-(sexp + xexp).  A delta that outgrows the lagged scale poisons its draw with NaN as before.
//
// What the compiler needs to be told (each of these cost a failed build):
//   * the update's MFMAs are inline assembly with the tile as a "+a" operand: as builtins the allocator moves the 172
//     accumulator registers between the two halves of the file around every step (120,000 v_accvgpr moves, 4000 spills);
//   * ONE copy of the step per loop: the two-draw sweep's three statically rotated copies are three register assignments
//     the back edges have to permute; the f'(u) / x prefetch is a register queue moved by v_mov instead;
//   * the rare rescaling multiply goes through LDS (ds_write_b128 / ds_read_b128 on accumulator registers): written on the
//     tiles it drags them into the vector half (170 spills, and every spill reload waits behind the HBM prefetch);
//   * inline-assembly MFMAs are invisible to the hazard recognizer: nothing the compiler generates may read a tile soon
//     after its MFMA.  The steady-state loops hold no accumulator moves (checked in the ISA; tests/test_fused_gpu.py
//     would see a stale read as missing contributions), and the rescale / the final read-out wait explicitly.
#include "ssn_duo_core.h"

#ifndef SSN_FUSE_STAMP
#define SSN_FUSE_STAMP 0        // diagnostic build: s_memtime ticks of workgroup 0 per segment (ssn_debug_fuse_stamps)
#endif
#ifndef SSN_FUSE_NPREG
#define SSN_FUSE_NPREG 28       // 2N > 152: 16 x 32 parts of W^T (of a wave's 44-46: two fp16 parts of 22-23 units) kept in registers; the rest
                                // lives in LDS and is read by the chain.  18 ... 36 build without spills; chain 1300 -> 880 cycles
                                // from 18 to 32, 28 leaves the dL/d ext build its registers
#endif
#ifndef SSN_FUSE_NV
#define SSN_FUSE_NV 0
#endif
#if SSN_FUSE_NV
#define SSN_FUSE_ASM asm
#else
#define SSN_FUSE_ASM asm volatile
#endif
#ifndef SSN_FUSE_ABLATE
#define SSN_FUSE_ABLATE 0       // timing only (wrong results): 1 = no rank-8 update, 2 = no chain
#endif

namespace ssn {

template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<N, I + 1>(f); }
}

#if SSN_FUSE_STAMP
__device__ unsigned long long fuse_stamps[32];
#endif

typedef short sv4 __attribute__((ext_vector_type(4)));
typedef short sv8 __attribute__((ext_vector_type(8)));

// the (part, stimulus) vector of one neuron per lane from an image in the chain's layout: two transposed reads (stimuli 0-3, 4-7)
__device__ __forceinline__ hv8 fuse_read_tr(unsigned addr) {
    using LdsV = __attribute__((address_space(3))) sv4*;
    const sv4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsV)(size_t)addr);
    const sv4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsV)(size_t)(addr + 64u));
    const sv8 v = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    return __builtin_bit_cast(hv8, v);
}

// The NRT x NRT grid of 16 x 16 accumulator tiles of dL/dW in strips: wave w holds the full tile rows r = w (mod 4) below
// NF = 4 floor(NRT / 4), and of each of the NRT - NF rows left over a contiguous run of columns (the widest run goes to wave 0,
// whose serial part is the lightest): 43 / 42 / 42 / 42 tiles at 2N = 208.  A strip needs few operands per step: <= 4 transposed
// reads of delta rows, held for the step, and the 13 x columns streamed one after the other.
template <int MK, int WV>
struct FuseTiles {
    using S = Duo16<MK>;
    static constexpr int NRT = S::NRT, NF = 4 * (NRT / 4), NFR = NRT / 4, NLR = NRT - NF;
    static constexpr int C0 = (3 - WV) * NRT / 4, C1 = (4 - WV) * NRT / 4, CW = C1 - C0;       // column run of the left-over rows
    static constexpr int NA = NFR + NLR;                                                       // delta operands per step
};

template <int MK>
struct FuseLds {
    using S = Duo16<MK>;
    static constexpr int NPMAX = 2 * ((S::UNITS + S::WM - 1) / S::WM);                          // parts of the largest share
    static constexpr int NPREG = MK > 152 ? SSN_FUSE_NPREG : NPMAX;
    static constexpr int NPL = NPMAX - NPREG;                                                  // parts per wave in LDS
    static constexpr int DIMG = 0, XIMG = 2 * S::BB, XSL = 4 * S::BB, SLOTS = XSL + (S::WM - 1) * S::XS, WMAX = SLOTS + 16,
                         WLDS = WMAX + 16, RSC = WLDS + S::WM * NPL * 1024, TOTAL = RSC + S::WM * 1024;    // RSC: 1 KB per wave for rescaling
};

// A wave's share of W^T as fp16 parts (part p = 2 unit + {0: high, 1: low}): the first NPREG parts in registers, the rest in
// LDS (one 16-byte operand per lane and part, 1 KB per part), and the chain over them.
template <int MK, int WV, int NPREG>
struct FuseOperands {
    using S = Duo16<MK>;
    using WS = DuoWave<MK, WV>;
    static constexpr int NU = WS::NU, NT = WS::NT, RT0 = WS::RT0, U0 = WS::U0, U1 = WS::U1, NP = 2 * NU;
    static constexpr int NREG = NPREG < NP ? NPREG : NP;
    using LdsH8 = __attribute__((address_space(3))) hv8*;
    hv8 R[NREG > 0 ? NREG : 1];
    unsigned wl;

    __device__ __forceinline__ void load(const __amdgpu_buffer_rsrc_t& rsrc, int M, int li, int lg, float sa, char* wlds, int lane) {
        wl = (unsigned)(size_t)(LdsH8)wlds + (unsigned)(lane * 16);
#pragma unroll
        for (int ui = 0; ui < NU; ++ui) {
            float w[8];
            duo_fetch_t(rsrc, M, 16 * ((U0 + ui) / S::NKT) + li, 32 * ((U0 + ui) % S::NKT) + 8 * lg, w);
            hv8 h, m;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sc = w[e] * sa;
                h[e] = (_Float16)sc;
                m[e] = (_Float16)(sc - (float)h[e]);
            }
            if (2 * ui < NREG) R[2 * ui < NREG ? 2 * ui : 0] = h; else *(LdsH8)(size_t)(wl + (unsigned)((2 * ui - NREG) * 1024)) = h;
            if (2 * ui + 1 < NREG) R[2 * ui + 1 < NREG ? 2 * ui + 1 : 0] = m; else *(LdsH8)(size_t)(wl + (unsigned)((2 * ui + 1 - NREG) * 1024)) = m;
        }
    }
    __device__ __forceinline__ hv8 part(int p) const {
        using LdsB = const __attribute__((address_space(3))) hv8*;
        if (p < NREG) return R[p < NREG ? p : 0];
        return *(LdsB)(size_t)(wl + (unsigned)((p - NREG) * 1024));
    }
    __device__ __forceinline__ void chain(unsigned rd, mf4 (&acc)[NT]) const {
        using LdsB = const __attribute__((address_space(3))) hv8*;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (mf4){0.f, 0.f, 0.f, 0.f};
        hv8 bnext = *(LdsB)(size_t)rd;
#pragma unroll
        for (int kt = 0; kt < S::NKT; ++kt) {
            const hv8 b1 = bnext;
            if (kt + 1 < S::NKT) bnext = *(LdsB)(size_t)(rd + (unsigned)((kt + 1) * 4 * S::BROW));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int u = (RT0 + t) * S::NKT + kt;
                    if (u >= U0 && u < U1) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(part(2 * (u - U0) + pt), b1, acc[t], 0, 0, 0);
                }
        }
    }
};

template <int MK, int WV, bool GEXT>
__device__ __forceinline__ void fuse_backward_wave(const GenBwdArgs<float>& a, float* __restrict__ gW, int xexp, int b, int lane,
                                                   char* lds) {
    using S = Duo16<MK>;
    using WS = DuoWave<MK, WV>;
    using FT = FuseTiles<MK, WV>;
    using FL = FuseLds<MK>;
    constexpr int NT = WS::NT, NTF = WS::NTF, RT0 = WS::RT0;
    constexpr int NE = 2 * NTF;
    constexpr int NRT = FT::NRT, NFR = FT::NFR, NLR = FT::NLR, CW = FT::CW;
    const int M = a.M, N = a.M / 2, T_ = a.seqlen;
    const int li = lane & 15, lg = lane >> 4, hi = li >> 3, st = li & 7;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W + (size_t)b * M * M), 0, M * M * 4, 0x00020000);
    using Ops = FuseOperands<MK, WV, FuseLds<MK>::NPREG>;
    using OpsMax = DuoOperands<MK, WV, 0, true>;
    using LdsH8 = const __attribute__((address_space(3))) hv8*;
    using LdsF4 = __attribute__((address_space(3))) mf4*;
    using LdsU = __attribute__((address_space(3))) unsigned*;
    using LdsUC = volatile const __attribute__((address_space(3))) unsigned*;
    const unsigned base = (unsigned)(size_t)(LdsH8)lds;
    unsigned* const wmax = reinterpret_cast<unsigned*>(lds + FL::WMAX);
    atomicMax(wmax, __builtin_bit_cast(unsigned, OpsMax::max_abs(rsrc, M, li, lg)));
    const unsigned xs = base + (unsigned)FL::XSL + (unsigned)(lane * 16);
    const unsigned b_rd = base + (unsigned)(lg * S::BROW + li * 16);                                  // + image, + kt * 4 * BROW
    const unsigned b_wr = base + (unsigned)((lg >> 1) * S::BROW + st * 16 + (lg & 1) * 8 + hi * 4);   // + image, + row tile part
    const unsigned slots = base + (unsigned)FL::SLOTS;
    const unsigned rsc = base + (unsigned)FL::RSC + (unsigned)(WV * 1024 + lane * 16);
    auto slot = [&](int tau) { return slots + 4u * (unsigned)((tau + 3) % 3); };
    // transposed operand reads: lane (li = 4 q + p) of a 16-lane group supplies row q (stimulus), columns 4 p .. 4 p + 3 (neurons)
    const unsigned tr_lane = (unsigned)(((li & 3) >> 1) * S::BROW + (li >> 2) * 16 + (li & 1) * 8);
    const unsigned tr_a = base + tr_lane + (unsigned)(8 * (lg >> 1) * 16);       // delta operand: parts h h m m over the lane groups
    const unsigned tr_b = base + tr_lane + (unsigned)(8 * (lg & 1) * 16);        // x operand:     parts h m h m
    // ---- the values this lane finishes
    const int s = st;
    const bool live = s < a.NB;
    const float inv = 1.f / (float)(T_ - a.skip);
    const size_t blk_elems = (size_t)a.NB * T_ * M;
    const __amdgpu_buffer_rsrc_t rs_traj =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.traj) + (size_t)b * blk_elems, 0, (int)(blk_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_df =
        __builtin_amdgcn_make_buffer_rsrc(a.delta + (size_t)b * blk_elems, 0, (int)(blk_elems * 4), 0x00020000);
    const int toff = live ? (int)(((size_t)s * T_ * M + 4 * lg + 2 * hi) * 4) : -1;
    int voff[NTF];
#pragma unroll
    for (int tf = 0; tf < NTF; ++tf) voff[tf] = (toff < 0 || 16 * (RT0 + tf) + 4 * lg + 2 * hi >= M) ? -1 : toff;
    const __amdgpu_buffer_rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.traj), 0, 0, 0x00020000);
    struct At { int tf, t; };
    auto at = [&](int tf, int t) { return At{tf, t}; };
    auto soff = [&](const At& p) { return (p.t * M + 16 * (RT0 + p.tf)) * 4; };
    auto load2 = [&](const __amdgpu_buffer_rsrc_t& rs, bool on, const At& p, float& x0, float& x1) {
        const fv2 q = __builtin_bit_cast(fv2, __builtin_amdgcn_raw_buffer_load_b64(on ? rs : rs_none, voff[p.tf], on ? soff(p) : 0, 0));
        x0 = q.x; x1 = q.y;
    };
    // f'(u) and x of three consecutive steps in a three-deep queue A <- B <- C <- load.  (The two-draw sweep rotates its sets
    // statically, three copies of the step; here every copy of the loop body is another register assignment for 172
    // accumulator registers that the back edge has to permute, so ONE body per loop and two moves per value and stream: the
    // move reads a register whose load was issued a whole step earlier, and there are no stores to wait behind.)
    float eps[NE], gta[NE], carry[NE], dsum[NE], xn[NE], xc[NE], dfA[NE], dfB[NE], dfC[NE], xA[NE], xB[NE], xC[NE];
    bool rowok[NE];
    auto direct = [&](int i, int tau, float xm) {     // dL/dx_tau inside the penalty window (time average, rate and dynamics terms)
        float gg = gta[i] + ((xc[i] > a.theta) ? a.c_rate : 0.f);
        if (tau <= T_ - 1) gg -= 2.f * a.c_dyn * (xn[i] - xc[i]);
        if (tau >= a.skip + 2) gg += 2.f * a.c_dyn * (xc[i] - xm);
        return gg;
    };
    __syncthreads();                                                          // (A) max |W|
    const int wexp = duo_w_exp(*wmax);
    Ops ops;
    ops.load(rsrc, M, li, lg, duo_pow2(wexp), lds + FL::WLDS + WV * FL::NPL * 1024, lane);
    float m0 = 0.f;
#pragma unroll
    for (int tf = 0; tf < NTF; ++tf) {
        const int row = 16 * (RT0 + tf) + 4 * lg + 2 * hi;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * tf + e;
            rowok[i] = row + e < M;
            eps[i] = (row + e < M && live) ? (row + e < N ? a.eps_E : a.eps_I) : 0.f;
            gta[i] = (live && row + e < M) ? a.g_time_avg[((size_t)b * a.NB + s) * M + row + e] * inv : 0.f;
            carry[i] = dsum[i] = xn[i] = 0.f;
        }
        load2(rs_traj, true, at(tf, T_ - 1), xc[2 * tf], xc[2 * tf + 1]);                        // x_T
        load2(rs_traj, T_ >= 2, at(tf, T_ - 2), xA[2 * tf], xA[2 * tf + 1]);                     // x_{T-1}: step T
        load2(rs_traj, T_ >= 3, at(tf, T_ - 3), xB[2 * tf], xB[2 * tf + 1]);                     // x_{T-2}: step T - 1
        load2(rs_traj, T_ >= 4, at(tf, T_ - 4), xC[2 * tf], xC[2 * tf + 1]);                     // x_{T-3}: step T - 2
        load2(rs_df, true, at(tf, T_ - 1), dfA[2 * tf], dfA[2 * tf + 1]);                        // f'(u_T)
        load2(rs_df, T_ >= 2, at(tf, T_ - 2), dfB[2 * tf], dfB[2 * tf + 1]);                     // f'(u_{T-1})
        load2(rs_df, T_ >= 3, at(tf, T_ - 3), dfC[2 * tf], dfC[2 * tf + 1]);                     // f'(u_{T-2})
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * tf + e;
            if (!rowok[i] || !live) continue;
            const float a_T = (T_ >= a.skip + 1) ? direct(i, T_, xA[i]) : 0.f;
            m0 = fmaxf(m0, fmaxf(__builtin_fabsf(eps[i] * dfA[i] * a_T), __builtin_fabsf(eps[i] * a_T) * 9.5367431640625e-07f));
        }
    }
    {
        const unsigned wm0 = duo_wave_max_bits(m0);
        if (lane == 0) __hip_atomic_fetch_max((__attribute__((address_space(3))) unsigned*)(lds + FL::SLOTS) + (T_ + 1 + 3) % 3, wm0,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    mf4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (mf4){0.f, 0.f, 0.f, 0.f};
    mf4 gwf[NFR > 0 ? NFR : 1][NRT], gwl[NLR > 0 ? NLR : 1][CW > 0 ? CW : 1];          // full rows WV, WV + 4, ...; left-over rows, my column run
#pragma unroll
    for (int r = 0; r < NFR; ++r)
#pragma unroll
        for (int c = 0; c < NRT; ++c) gwf[r][c] = (mf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NLR; ++r)
#pragma unroll
        for (int c = 0; c < CW; ++c) gwl[r][c] = (mf4){0.f, 0.f, 0.f, 0.f};
    float dmx = 0.f;                                // max |delta| of the lane's values over the sweep (a.dmax)
    bool poisoned = false;
    int bused = 0;                                  // the scale exponent the draw's delta in the chain's image was written with
    int sexp = 0x7fffffff;                          // running minimum of bexp: the scale exponent of the accumulators
    unsigned lastref = 0u;
    auto chain = [&](unsigned img) {
        if (SSN_FUSE_ABLATE & 2) return;
        ops.chain(b_rd + (unsigned)FL::DIMG + img, acc);
        if constexpr (WS::TAIL_SHARED) *(LdsF4)(size_t)(xs + (unsigned)(WV * S::XS)) = acc[NT - 1];
    };
    // the rank-8 update of one step from the two images of that step (image offsets: 0 or BB), MFMA by MFMA at sites that the
    // serial part of the NEXT step spreads over its own instruction stream: the matrix pipe works on one while the vector
    // instructions up to the next site issue
    hv8 aopf[NFR > 0 ? NFR : 1], aopl[NLR > 0 ? NLR : 1], ubn, ubop;
    unsigned uimg = 0u;
    auto update_begin = [&](unsigned img) {
        uimg = img;
#pragma unroll
        for (int r = 0; r < NFR; ++r) aopf[r] = fuse_read_tr(tr_a + (unsigned)FL::DIMG + img + (unsigned)((WV + 4 * r) * 512));
#pragma unroll
        for (int r = 0; r < NLR; ++r) aopl[r] = fuse_read_tr(tr_a + (unsigned)FL::DIMG + img + (unsigned)((FT::NF + r) * 512));
        ubn = fuse_read_tr(tr_b + (unsigned)FL::XIMG + img);
    };
    // MFMA number j of the step's NM, column by column (a column in the wave's run of the left-over rows has NLR more)
    constexpr int NM = NFR * NRT + NLR * CW;
    auto update_one = [&](int j) {                      // (j is a constant after inlining)
        int c = 0, i = j;
        for (; c < NRT; ++c) {
            const int n = NFR + ((c >= FT::C0 && c < FT::C1) ? NLR : 0);
            if (i < n) break;
            i -= n;
        }
        if (c >= NRT) return;
        if (i == 0) {
            ubop = ubn;
            if (c + 1 < NRT) ubn = fuse_read_tr(tr_b + (unsigned)FL::XIMG + uimg + (unsigned)((c + 1) * 512));
        }
        if (i < NFR) SSN_FUSE_ASM("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(gwf[i < NFR ? i : 0][c < NRT ? c : 0]) : "v"(aopf[i < NFR ? i : 0]), "v"(ubop));
        else {
            const int r = i - NFR < NLR ? i - NFR : 0, cc = c - FT::C0 < CW ? (c - FT::C0 >= 0 ? c - FT::C0 : 0) : 0;
            SSN_FUSE_ASM("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(gwl[r][cc]) : "v"(aopl[r]), "v"(ubop));
        }
    };
    // site k of NSITE in the serial part issues MFMAs k NM / NSITE ... (k + 1) NM / NSITE - 1: one, sometimes two -- a second MFMA
    // right behind the first waits 12 cycles for the pipe, and every vector instruction behind it waits too
    constexpr int NSITE = 2 * NTF + 3 * NE + 4 * NTF + 2;
    auto update_slot = [&](auto UPD, auto KK) {
        if constexpr (decltype(UPD)::value && !(SSN_FUSE_ABLATE & 1)) {
            constexpr int k = decltype(KK)::value, j0 = k * NM / NSITE, j1 = (k + 1) * NM / NSITE;
            static_assert(k < NSITE && j1 - j0 <= 3, "MFMAs per site");
            // (fences: left alone, the scheduler gathers the MFMAs back into groups of three or four)
            if constexpr (j1 > j0) __builtin_amdgcn_sched_barrier(0);
            if constexpr (j1 > j0) update_one(j0);
            if constexpr (j1 > j0 + 1) update_one(j0 + 1);
            if constexpr (j1 > j0 + 2) update_one(j0 + 2);
            if constexpr (j1 > j0) __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto join_tile = [&](auto UPD, auto TF, float usc) {
        constexpr int tf = decltype(TF)::value;
        mf4 sm = acc[tf];
        if constexpr (WS::HEAD_SHARED) {
            if (tf == 0) {
                const mf4 xp = *(LdsF4)(size_t)(xs + (unsigned)((WV - 1) * S::XS));
                sm.x += xp.x; sm.y += xp.y; sm.z += xp.z; sm.w += xp.w;
            }
        }
        carry[2 * tf] = fmaf(duo_join<true>(sm.x, sm.z, hi, sm.y, sm.w), usc, carry[2 * tf]);
        update_slot(UPD, std::integral_constant<int, 2 * tf>{});
        carry[2 * tf + 1] = fmaf(duo_join<false>(sm.y, sm.w, hi), usc, carry[2 * tf + 1]);
        update_slot(UPD, std::integral_constant<int, 2 * tf + 1>{});
    };
    auto serial = [&](auto WIN, auto UPD, int tau) {
        constexpr bool win_on = decltype(WIN)::value;
        if constexpr (decltype(UPD)::value && !(SSN_FUSE_ABLATE & 1)) update_begin((unsigned)(((tau + 1) & 1) * S::BB));
        float (&dfc)[NE] = dfA;
        float (&xm)[NE] = xA;                            // x_{tau-1}: the operand of this step's update, and of the window terms
        const unsigned img = (unsigned)((tau & 1) * S::BB);
        // scale of this step's delta from the previous step's maximum (kept when that was exactly zero)
        const unsigned mprev = (unsigned)__builtin_amdgcn_readfirstlane((int)*(LdsUC)(size_t)slot(tau + 1));
        if (WV == 0 && lane == 0) *(LdsU)(size_t)slot(tau + 2) = 0u;
        const unsigned ref = mprev ? mprev : lastref;
        lastref = ref;
        int bexp = 7 - ((int)((ref >> 23) & 0xffu) - 127);
        bexp = ref == 0u ? 0 : (bexp > 100 ? 100 : (bexp < -100 ? -100 : bexp));
        if constexpr (decltype(UPD)::value) {              // (every step but the first: tau < T)
            const float usc = duo_pow2(-wexp - bused);
            static_for<NTF>([&](auto TF) { join_tile(UPD, TF, usc); });
        }
        float delta[NE], dm = 0.f;
        static_for<NE>([&](auto II) {
            constexpr int i = decltype(II)::value;
            float a_t = carry[i];
            if constexpr (win_on) a_t += direct(i, tau, xm[i]);
            update_slot(UPD, std::integral_constant<int, 2 * NTF + 3 * i>{});
            delta[i] = eps[i] * dfc[i] * a_t;
            dm = fmaxf(dm, __builtin_fabsf(delta[i]));
            update_slot(UPD, std::integral_constant<int, 2 * NTF + 3 * i + 1>{});
            carry[i] = fmaf(-eps[i], a_t, a_t);                                   // (1 - eps) a_t
            if (GEXT) dsum[i] += delta[i];
            update_slot(UPD, std::integral_constant<int, 2 * NTF + 3 * i + 2>{});
        });
        const float rs = live ? duo_pow2(bexp) : 0.f;
        if (!(dm * rs < 65504.f)) { delta[0] = __builtin_nanf(""); poisoned = true; }   // outgrew the lagged scale: poison, do not clamp
        dmx = fmaxf(dmx, live ? dm : 0.f);
        const int snew = bexp < sexp ? bexp : sexp;
        // x_{tau-1} under the accumulators' scale: 2^(xexp + snew - bexp), at most 2^xexp
        int xe = xexp + snew - bexp;
        xe = xe < -100 ? -100 : xe;
        const float xsc = duo_pow2(xe);
        static_for<NTF>([&](auto TF) {
            constexpr int tf = decltype(TF)::value;
            const int rt = RT0 + tf;
            unsigned h, m;
            constexpr int s0 = 2 * NTF + 3 * NE + 4 * tf;
            duo_split2(delta[2 * tf], delta[2 * tf + 1], rs, h, m);
            update_slot(UPD, std::integral_constant<int, s0>{});
            const unsigned wr = b_wr + (unsigned)FL::DIMG + img + (unsigned)(((rt >> 1) * 4 + 2 * (rt & 1)) * S::BROW);
            *(LdsU)(size_t)wr = h;
            *(LdsU)(size_t)(wr + 128u) = m;
            update_slot(UPD, std::integral_constant<int, s0 + 1>{});
            unsigned xh, xl;
            duo_split2(xm[2 * tf], xm[2 * tf + 1], xsc, xh, xl);
            update_slot(UPD, std::integral_constant<int, s0 + 2>{});
            const unsigned wx = b_wr + (unsigned)FL::XIMG + img + (unsigned)(((rt >> 1) * 4 + 2 * (rt & 1)) * S::BROW);
            *(LdsU)(size_t)wx = xh;
            *(LdsU)(size_t)(wx + 128u) = xl;
            update_slot(UPD, std::integral_constant<int, s0 + 3>{});
        });
        if constexpr (win_on) {
#pragma unroll
            for (int i = 0; i < NE; ++i) { xn[i] = xc[i]; xc[i] = xm[i]; }
        }
        // the queues move up (B and C were loaded one and two steps ago), then the loads for step tau - 3 go into C
#pragma unroll
        for (int i = 0; i < NE; ++i) { dfA[i] = dfB[i]; dfB[i] = dfC[i]; xA[i] = xB[i]; xB[i] = xC[i]; }
        update_slot(UPD, std::integral_constant<int, NSITE - 2>{});
#pragma unroll
        for (int tf = 0; tf < NTF; ++tf) {
            load2(rs_df, tau >= 4, at(tf, tau - 4), dfC[2 * tf], dfC[2 * tf + 1]);           // f'(u_{tau-3})
            load2(rs_traj, tau >= 5, at(tf, tau - 5), xC[2 * tf], xC[2 * tf + 1]);           // x_{tau-4}: step tau - 3 pairs it with delta_{tau-3}
        }
        bused = bexp;
        update_slot(UPD, std::integral_constant<int, NSITE - 1>{});
        const unsigned wm = duo_wave_max_bits(live ? dm : 0.f);
        if (lane == 0) __hip_atomic_fetch_max((__attribute__((address_space(3))) unsigned*)(lds + FL::SLOTS) + (tau + 3) % 3, wm,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return snew;
    };
    constexpr std::integral_constant<bool, false> W0{};
    constexpr std::integral_constant<bool, true> W1{};
    __syncthreads();                                                          // (B)
#if SSN_FUSE_STAMP
    unsigned long long st_a = 0, st_b1 = 0, st_c = 0, st_b2 = 0; int st_n = 0;
#endif
    auto step = [&](auto WIN, auto UPD, int tau) {
#if SSN_FUSE_STAMP
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        // phase A: the update of step tau + 1 (images of the other parity) and the serial part of step tau, one instruction stream
        const int snew = serial(WIN, UPD, tau);
        if (snew != sexp) {                                                    // |delta| reached a new binade: the sums follow
            if (sexp != 0x7fffffff) {
                // through LDS, tile by tile: the accumulator file is not a VALU operand, and a multiply written on the tiles
                // themselves costs the whole kernel its register assignment (170 spilled registers)
                const float f = duo_pow2(snew - sexp < -120 ? -120 : snew - sexp);
                // (the update's MFMAs are inline assembly: the compiler does not count their result latency for us)
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                auto scale_tile = [&](mf4& t) {
                    asm volatile("ds_write_b128 %0, %1" : : "v"(rsc), "a"(t) : "memory");
                    mf4 v = *(LdsF4)(size_t)rsc;
                    v.x *= f; v.y *= f; v.z *= f; v.w *= f;
                    *(LdsF4)(size_t)rsc = v;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=a"(t) : "v"(rsc) : "memory");
                };
#pragma unroll
                for (int r = 0; r < NFR; ++r)
#pragma unroll
                    for (int c = 0; c < NRT; ++c) scale_tile(gwf[r][c]);
#pragma unroll
                for (int r = 0; r < NLR; ++r)
#pragma unroll
                    for (int c = 0; c < CW; ++c) scale_tile(gwl[r][c]);
            }
            sexp = snew;
        }
#if SSN_FUSE_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        duo_phase_barrier();
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        chain((unsigned)((tau & 1) * S::BB));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        duo_phase_barrier();
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        if (!decltype(WIN)::value) { st_a += t1 - t0; st_b1 += t2 - t1; st_c += t3 - t2; st_b2 += t4 - t3; ++st_n; }
#else
        duo_phase_barrier();
        chain((unsigned)((tau & 1) * S::BB));
        duo_phase_barrier();
#endif
    };
    const int tw = a.skip + 1 > 1 ? a.skip + 1 : 1;     // window steps first (time runs backwards): tau = T ... tw
    step(W1, W0, T_);                                   // (step T lies in the window: skip < T; no update before it)
    int tau = T_ - 1;
    for (; tau >= tw; --tau) step(W1, W1, tau);
    for (; tau >= 1; --tau) step(W0, W1, tau);
    // (the update of step 1 pairs delta_1 with x_0 = 0: nothing to add)
#if SSN_FUSE_STAMP
    if (blockIdx.x == 0 && lane == 0) {
        unsigned long long* o = fuse_stamps + 8 * WV;
        o[0] = st_a; o[1] = st_b1; o[2] = st_c; o[3] = st_b2; o[4] = (unsigned long long)st_n;
    }
#endif
    // ---- dL/dW of this draw: acc 2^-(sexp + xexp); accumulator lane (lg, li) holds rows 4 lg .. 4 lg + 3 of column li
    {
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        const int se = sexp == 0x7fffffff ? 0 : sexp;
        int fe = -(se + xexp);
        fe = fe > 120 ? 120 : (fe < -120 ? -120 : fe);
        const float fin = duo_pow2(fe);
        float* const out = gW + (size_t)b * M * M;
        auto put = [&](const mf4& t, int rt, int ct) {
            const int j = 16 * ct + li;
            const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 16 * rt + 4 * lg + e;
                if (i < M && j < M) out[(size_t)i * M + j] = v[e] * fin;
            }
        };
#pragma unroll
        for (int r = 0; r < NFR; ++r)
#pragma unroll
            for (int c = 0; c < NRT; ++c) put(gwf[r][c], WV + 4 * r, c);
#pragma unroll
        for (int r = 0; r < NLR; ++r)
#pragma unroll
            for (int c = 0; c < CW; ++c) put(gwl[r][c], FT::NF + r, FT::C0 + c);
    }
    if (a.dmax) {                                   // max |delta| of the draw, NaN for a poisoned one (as gen_backward_duo_kernel)
        const unsigned wm = duo_wave_max_bits(dmx);
        if (lane == 0) atomicMax(a.dmax + b, wm);
        if (poisoned) atomicMax(a.dmax + b, 0x7fc00000u);
    }
    if (GEXT && live) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int row = 16 * (RT0 + i / 2) + 4 * lg + 2 * hi + (i & 1);
            if (row < M) a.g_ext[((size_t)b * a.NB + s) * M + row] = dsum[i];
        }
    }
}

// grid: one workgroup of 256 threads per draw (NB <= 8)
template <int MK, bool GEXT>
__global__ void __launch_bounds__(256) gen_backward_fused_kernel(GenBwdArgs<float> a, float* gW, int xexp) {
    using FL = FuseLds<MK>;
    __shared__ __align__(16) char lds[FL::TOTAL];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < FL::WLDS / 4; c += blockDim.x) reinterpret_cast<unsigned*>(lds)[c] = 0u;
    __syncthreads();
    switch (wave) {
        case 0: fuse_backward_wave<MK, 0, GEXT>(a, gW, xexp, b, lane, lds); break;
        case 1: fuse_backward_wave<MK, 1, GEXT>(a, gW, xexp, b, lane, lds); break;
        case 2: fuse_backward_wave<MK, 2, GEXT>(a, gW, xexp, b, lane, lds); break;
        default: fuse_backward_wave<MK, 3, GEXT>(a, gW, xexp, b, lane, lds); break;
    }
}

static int fuse_pick_mk(int M) {
    const int ladder[] = {104, 152, 208};
    for (int mk : ladder) if (M <= mk) return mk;
    return 0;
}
bool gen_backward_fused_supported(const GenBwdArgs<float>& a, float xmax) {
    return a.B > 0 && a.NB >= 1 && a.NB <= 8 && a.M >= 2 && (a.M & 1) == 0 && fuse_pick_mk(a.M) != 0 && a.seqlen >= 1 &&
           xmax > 0.f && xmax < __builtin_inff();
}
template <int MK>
static hipError_t launch_fused_mk(const GenBwdArgs<float>& a, float* gW, int xexp, hipStream_t st) {
    if (a.g_ext) hipLaunchKernelGGL((gen_backward_fused_kernel<MK, true>), dim3(a.B), dim3(256), 0, st, a, gW, xexp);
    else hipLaunchKernelGGL((gen_backward_fused_kernel<MK, false>), dim3(a.B), dim3(256), 0, st, a, gW, xexp);
    return hipGetLastError();
}
// a.delta = f'(u) [B][NB][T][M] (read only here), a.traj the trajectory; gW [B][M][M] out; xmax >= every |traj| element
hipError_t launch_gen_backward_fused(const GenBwdArgs<float>& a, float* gW, float xmax, hipStream_t st) {
    if (!gen_backward_fused_supported(a, xmax) || !gW) return hipErrorInvalidValue;
    // x 2^xexp < 2^15: the fp16 parts of the largest rate stay finite
    const int e = (int)((__builtin_bit_cast(unsigned, xmax) >> 23) & 0xffu) - 127;          // floor(log2 xmax)
    int xexp = 14 - e - 1;
    xexp = xexp > 100 ? 100 : (xexp < -100 ? -100 : xexp);
    switch (fuse_pick_mk(a.M)) {
        case 104: return launch_fused_mk<104>(a, gW, xexp, st);
        case 152: return launch_fused_mk<152>(a, gW, xexp, st);
        case 208: return launch_fused_mk<208>(a, gW, xexp, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ssn

#if SSN_FUSE_STAMP
extern "C" int ssn_debug_fuse_stamps(unsigned long long* out32) {
    return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(ssn::fuse_stamps), 32 * sizeof(unsigned long long));
}
#endif
