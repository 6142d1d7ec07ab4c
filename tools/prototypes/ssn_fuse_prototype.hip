// PROTOTYPE, NOT BUILT INTO THE LIBRARY (round 4; DESIGN 3.7d holds the verdict).  It compiles
//   (cd tc_gan_amd/csrc && hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -I. -S --cuda-device-only
//    ../../tools/prototypes/ssn_fuse_prototype.hip -o /tmp/fuse.s)
// and that is what it is for: at 2N = 208 the register allocation comes out with 3900-4100 spilled registers and 31,000
// v_accvgpr moves (114,000 lines of assembly), i.e. the design below does not fit the register file through this compiler.
//
// The BPTT adjoint sweep with dL/dW accumulated ON CHIP (round 4): one launch replaces gen_backward_duo_kernel (ssn_duo.hip) +
// gw_split_kernel (ssn_gw.hip), and the 7.9 GB delta stream between them (written by one, read back with the trajectory by
// the other) never exists.
//
//   workgroup = 4 waves = ONE draw, one wave per SIMD, so a wave may use the whole 512-entry register file: its quarter of
//   W^T as two fp16 parts (184 registers, as in the two-draw kernels) AND a quadrant of the 13 x 13 grid of 16 x 16
//   accumulator tiles of dL/dW (49 / 42 / 42 / 36 tiles = 196 ... 144 registers), which stay there for the whole sweep.
//   Per step tau, two phases, one s_barrier each:
//     A: serial part of step tau (as gen_backward_duo: join of the chain's sums, delta_tau = eps f'(u_tau) a_tau, carry,
//        lagged power-of-two scale, delta_tau as two fp16 parts into the B image of this step's chain) + x_{tau-1} as two
//        fp16 parts into a second image of the same layout + the rank-8 update of step tau + 1:
//            gW[i][j] += sum_s delta_{tau+1}[s][i] x_tau[s][j]
//        = one v_mfma_f32_16x16x32_f16 per tile: K = 32 = 8 stimuli x {d_h x_h, d_h x_m, d_m x_h, d_m x_m};
//     B: chain W^T delta_tau (46 MFMAs per wave).
//   Both operands of the update are per-neuron vectors over (part, stimulus), the transpose of what the images hold
//   ([column = 8 part + stimulus][8 neurons]): ds_read_b64_tr_b16 delivers them transposed, two reads per operand, no
//   shuffles and no second copy of delta (tools/microbench/tr_read_check.hip pins the address map on the hardware).
//   The update's MFMAs touch no register of the serial part: one instruction stream carries both, the matrix pipe works
//   while the vector instructions of the serial part issue.
//
// Scales.  The chain's image of delta_tau carries the lagged scale 2^bexp(tau) of the two-draw sweep (max |delta_{tau+1}|
// at 2^7).  The accumulators need ONE scale per draw for the whole sum, so they follow the running minimum sexp of bexp
// (|delta| growing backwards in time lowers it); the factor 2^(sexp - bexp) <= 1 goes into the x image of the same step
// (x 2^(xexp + sexp - bexp), xexp from the rate bound of the saturating I/O function) -- a step whose delta is far below
// the largest one seen so far loses low bits of a product that is negligible against the sum -- and when sexp drops, the
// accumulators are multiplied by the power of two once (at most a few dozen times per sweep).  At the end
// gW = acc 2^-(sexp + xexp).  A delta that outgrows the lagged scale poisons its draw with NaN as before.
#include "ssn_duo_core.h"       // (tc_gan_amd/csrc: compile with -I tc_gan_amd/csrc)

#ifndef SSN_FUSE_STAMP
#define SSN_FUSE_STAMP 0        // diagnostic build: s_memtime ticks of workgroup 0 per segment (ssn_debug_fuse_stamps)
#endif
#ifndef SSN_FUSE_NL
#define SSN_FUSE_NL 16          // 2N > 152: units of every wave whose low part W_m lives in LDS (W 184 + accumulators 196 + state do not fit 512)
#endif
#ifndef SSN_FUSE_ABLATE
#define SSN_FUSE_ABLATE 0       // timing only (wrong results): 1 = no rank-8 update, 2 = no chain
#endif

namespace ssn {

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

// quadrants of the NRT x NRT tile grid: wave 0 (lightest serial part) the largest, wave 3 (four row tiles to finish) the smallest
template <int MK, int WV>
struct FuseTiles {
    using S = Duo16<MK>;
    static constexpr int H = (S::NRT + 1) / 2;
    static constexpr int R0 = (WV & 2) ? H : 0, R1 = (WV & 2) ? S::NRT : H;
    static constexpr int C0 = (WV & 1) ? H : 0, C1 = (WV & 1) ? S::NRT : H;
    static constexpr int NR = R1 - R0, NC = C1 - C0;
};

template <int MK>
struct FuseLds {
    using S = Duo16<MK>;
    static constexpr int NL = MK > 152 ? SSN_FUSE_NL : 0;
    static constexpr int DIMG = 0, XIMG = 2 * S::BB, XSL = 4 * S::BB, SLOTS = XSL + (S::WM - 1) * S::XS, WMAX = SLOTS + 16,
                         WLDS = WMAX + 16, TOTAL = WLDS + S::WM * NL * 1024;
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
    constexpr int NR = FT::NR, NC = FT::NC;
    const int M = a.M, N = a.M / 2, T_ = a.seqlen;
    const int li = lane & 15, lg = lane >> 4, hi = li >> 3, st = li & 7;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W + (size_t)b * M * M), 0, M * M * 4, 0x00020000);
    using Ops = DuoOperands<MK, WV, FuseLds<MK>::NL, true>;
    using LdsH8 = const __attribute__((address_space(3))) hv8*;
    using LdsF4 = __attribute__((address_space(3))) mf4*;
    using LdsU = __attribute__((address_space(3))) unsigned*;
    using LdsUC = volatile const __attribute__((address_space(3))) unsigned*;
    const unsigned base = (unsigned)(size_t)(LdsH8)lds;
    unsigned* const wmax = reinterpret_cast<unsigned*>(lds + FL::WMAX);
    atomicMax(wmax, __builtin_bit_cast(unsigned, Ops::max_abs(rsrc, M, li, lg)));
    const unsigned xs = base + (unsigned)FL::XSL + (unsigned)(lane * 16);
    const unsigned b_rd = base + (unsigned)(lg * S::BROW + li * 16);                                  // + image, + kt * 4 * BROW
    const unsigned b_wr = base + (unsigned)((lg >> 1) * S::BROW + st * 16 + (lg & 1) * 8 + hi * 4);   // + image, + row tile part
    const unsigned slots = base + (unsigned)FL::SLOTS;
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
    // f'(u) and x of three consecutive steps in rotating register sets (static rotation: see gen_backward_duo): the step with
    // phase PH uses set (PH + 1) % 3 and loads, two steps ahead, into set PH
    float eps[NE], gta[NE], carry[NE], dsum[NE], xn[NE], xc[NE], df3[3][NE], xr3[3][NE];
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
    ops.load(rsrc, M, li, lg, duo_pow2(wexp), lds + FL::WLDS + WV * FL::NL * 1024, lane);
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
        load2(rs_traj, T_ >= 2, at(tf, T_ - 2), xr3[1][2 * tf], xr3[1][2 * tf + 1]);             // x_{T-1}: phase 0 uses set 1
        load2(rs_traj, T_ >= 3, at(tf, T_ - 3), xr3[2][2 * tf], xr3[2][2 * tf + 1]);             // x_{T-2}: phase 1 uses set 2
        load2(rs_df, true, at(tf, T_ - 1), df3[1][2 * tf], df3[1][2 * tf + 1]);                  // f'(u_T)
        load2(rs_df, T_ >= 2, at(tf, T_ - 2), df3[2][2 * tf], df3[2][2 * tf + 1]);               // f'(u_{T-1})
        df3[0][2 * tf] = df3[0][2 * tf + 1] = xr3[0][2 * tf] = xr3[0][2 * tf + 1] = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * tf + e;
            if (!rowok[i] || !live) continue;
            const float a_T = (T_ >= a.skip + 1) ? direct(i, T_, xr3[1][i]) : 0.f;
            m0 = fmaxf(m0, fmaxf(__builtin_fabsf(eps[i] * df3[1][i] * a_T), __builtin_fabsf(eps[i] * a_T) * 9.5367431640625e-07f));
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
    mf4 gw[NR][NC];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) gw[r][c] = (mf4){0.f, 0.f, 0.f, 0.f};
    int bused = 0;                                  // the scale exponent the draw's delta in the chain's image was written with
    int sexp = 0x7fffffff;                          // running minimum of bexp: the scale exponent of the accumulators
    unsigned lastref = 0u;
    auto chain = [&](unsigned img) {
        if (SSN_FUSE_ABLATE & 2) return;
        ops.chain(b_rd + img, acc);
        if constexpr (WS::TAIL_SHARED) *(LdsF4)(size_t)(xs + (unsigned)(WV * S::XS)) = acc[NT - 1];
    };
    // the rank-8 update of one step from the two images of that step (image offsets: 0 or BB)
    auto update = [&](unsigned img) {
        if (SSN_FUSE_ABLATE & 1) return;
        hv8 aop[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) aop[r] = fuse_read_tr(tr_a + (unsigned)FL::DIMG + img + (unsigned)((FT::R0 + r) * 512));
        hv8 bnext = fuse_read_tr(tr_b + (unsigned)FL::XIMG + img + (unsigned)(FT::C0 * 512));
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const hv8 bop = bnext;
            if (c + 1 < NC) bnext = fuse_read_tr(tr_b + (unsigned)FL::XIMG + img + (unsigned)((FT::C0 + c + 1) * 512));
            // (the accumulators are pinned to the accumulator half of the register file: left to itself the compiler keeps
            // them where the rare rescaling multiply below can reach them -- in the vector half, which W^T and the sweep's
            // state already fill -- and spills thousands of registers.  Every tile gets ONE MFMA per step and is next read
            // a phase later: no wait states to spell out)
#pragma unroll
            for (int r = 0; r < NR; ++r)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(gw[r][c]) : "v"(aop[r]), "v"(bop));
        }
    };
    auto join_tile = [&](int tf, float usc) {
        mf4 sm = acc[tf];
        if constexpr (WS::HEAD_SHARED) {
            if (tf == 0) {
                const mf4 xp = *(LdsF4)(size_t)(xs + (unsigned)((WV - 1) * S::XS));
                sm.x += xp.x; sm.y += xp.y; sm.z += xp.z; sm.w += xp.w;
            }
        }
        carry[2 * tf] = fmaf(duo_join<true>(sm.x, sm.z, hi, sm.y, sm.w), usc, carry[2 * tf]);
        carry[2 * tf + 1] = fmaf(duo_join<false>(sm.y, sm.w, hi), usc, carry[2 * tf + 1]);
    };
    auto serial = [&](auto WIN, auto PH, int tau) {
        constexpr bool win_on = decltype(WIN)::value;
        constexpr int ph = decltype(PH)::value;
        float (&dfc)[NE] = df3[(ph + 1) % 3];
        float (&ndf)[NE] = df3[ph];
        float (&xm)[NE] = xr3[(ph + 1) % 3];             // x_{tau-1}: the operand of this step's update, and of the window terms
        float (&nxr)[NE] = xr3[ph];
        const unsigned img = (unsigned)((tau & 1) * S::BB);
        // scale of this step's delta from the previous step's maximum (kept when that was exactly zero)
        const unsigned mprev = (unsigned)__builtin_amdgcn_readfirstlane((int)*(LdsUC)(size_t)slot(tau + 1));
        if (WV == 0 && lane == 0) *(LdsU)(size_t)slot(tau + 2) = 0u;
        const unsigned ref = mprev ? mprev : lastref;
        lastref = ref;
        int bexp = 7 - ((int)((ref >> 23) & 0xffu) - 127);
        bexp = ref == 0u ? 0 : (bexp > 100 ? 100 : (bexp < -100 ? -100 : bexp));
        if (tau < T_) {
            const float usc = duo_pow2(-wexp - bused);
#pragma unroll
            for (int tf = 0; tf < NTF; ++tf) join_tile(tf, usc);
        }
        float delta[NE], dm = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            float a_t = carry[i];
            if constexpr (win_on) a_t += direct(i, tau, xm[i]);
            delta[i] = eps[i] * dfc[i] * a_t;
            dm = fmaxf(dm, __builtin_fabsf(delta[i]));
            carry[i] = fmaf(-eps[i], a_t, a_t);                                   // (1 - eps) a_t
            if (GEXT) dsum[i] += delta[i];
        }
        const float rs = live ? duo_pow2(bexp) : 0.f;
        if (!(dm * rs < 65504.f)) delta[0] = __builtin_nanf("");                   // outgrew the lagged scale: poison, do not clamp
        const int snew = bexp < sexp ? bexp : sexp;
        // x_{tau-1} under the accumulators' scale: 2^(xexp + snew - bexp), at most 2^xexp
        int xe = xexp + snew - bexp;
        xe = xe < -100 ? -100 : xe;
        const float xsc = duo_pow2(xe);
#pragma unroll
        for (int tf = 0; tf < NTF; ++tf) {
            const int rt = RT0 + tf;
            unsigned h, m;
            duo_split2(delta[2 * tf], delta[2 * tf + 1], rs, h, m);
            const unsigned wr = b_wr + (unsigned)FL::DIMG + img + (unsigned)(((rt >> 1) * 4 + 2 * (rt & 1)) * S::BROW);
            *(LdsU)(size_t)wr = h;
            *(LdsU)(size_t)(wr + 128u) = m;
            unsigned xh, xl;
            duo_split2(xm[2 * tf], xm[2 * tf + 1], xsc, xh, xl);
            const unsigned wx = b_wr + (unsigned)FL::XIMG + img + (unsigned)(((rt >> 1) * 4 + 2 * (rt & 1)) * S::BROW);
            *(LdsU)(size_t)wx = xh;
            *(LdsU)(size_t)(wx + 128u) = xl;
        }
        if constexpr (win_on) {
#pragma unroll
            for (int i = 0; i < NE; ++i) { xn[i] = xc[i]; xc[i] = xm[i]; }
        }
        // loads for two steps ahead into the sets this step has just finished with
#pragma unroll
        for (int tf = 0; tf < NTF; ++tf) {
            load2(rs_df, tau >= 3, at(tf, tau - 3), ndf[2 * tf], ndf[2 * tf + 1]);
            load2(rs_traj, tau >= 4, at(tf, tau - 4), nxr[2 * tf], nxr[2 * tf + 1]);       // x_{tau-3}: step tau - 2 pairs it with delta_{tau-2}
        }
        bused = bexp;
        const unsigned wm = duo_wave_max_bits(live ? dm : 0.f);
        if (lane == 0) __hip_atomic_fetch_max((__attribute__((address_space(3))) unsigned*)(lds + FL::SLOTS) + (tau + 3) % 3, wm,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return snew;
    };
    constexpr std::integral_constant<bool, false> W0{};
    constexpr std::integral_constant<bool, true> W1{};
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    constexpr std::integral_constant<int, 2> P2{};
    __syncthreads();                                                          // (B)
#if SSN_FUSE_STAMP
    unsigned long long st_a = 0, st_b1 = 0, st_c = 0, st_b2 = 0; int st_n = 0;
#endif
    auto step = [&](auto WIN, auto PH, int tau) {
#if SSN_FUSE_STAMP
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        // phase A: the update of step tau + 1 (images of the other parity) and the serial part of step tau, one instruction stream
        if (tau < T_) update((unsigned)(((tau + 1) & 1) * S::BB));
        const int snew = serial(WIN, PH, tau);
        if (snew != sexp) {                                                    // |delta| reached a new binade: the sums follow
            if (sexp != 0x7fffffff) {
                const float f = duo_pow2(snew - sexp < -120 ? -120 : snew - sexp);
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int c = 0; c < NC; ++c) { gw[r][c].x *= f; gw[r][c].y *= f; gw[r][c].z *= f; gw[r][c].w *= f; }
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
    int tau = T_, ph = 0;
    auto step_any = [&](auto WIN) {
        if (ph == 0) step(WIN, P0, tau); else if (ph == 1) step(WIN, P1, tau); else step(WIN, P2, tau);
        ph = ph == 2 ? 0 : ph + 1;
        --tau;
    };
    const int tw = a.skip + 1 > 1 ? a.skip + 1 : 1;     // window steps first (time runs backwards): tau = T ... tw
    for (; tau - 2 >= tw; tau -= 3) {
        step(W1, P0, tau);
        step(W1, P1, tau - 1);
        step(W1, P2, tau - 2);
    }
    while (tau >= tw) step_any(W1);
    while (tau >= 1 && ph != 0) step_any(W0);
    for (; tau >= 3; tau -= 3) {
        step(W0, P0, tau);
        step(W0, P1, tau - 1);
        step(W0, P2, tau - 2);
    }
    while (tau >= 1) step_any(W0);
    // (the update of step 1 pairs delta_1 with x_0 = 0: nothing to add)
#if SSN_FUSE_STAMP
    if (blockIdx.x == 0 && lane == 0) {
        unsigned long long* o = fuse_stamps + 8 * WV;
        o[0] = st_a; o[1] = st_b1; o[2] = st_c; o[3] = st_b2; o[4] = (unsigned long long)st_n;
    }
#endif
    // ---- dL/dW of this draw: acc 2^-(sexp + xexp); accumulator lane (lg, li) holds rows 4 lg .. 4 lg + 3 of column li
    {
        const int se = sexp == 0x7fffffff ? 0 : sexp;
        int fe = -(se + xexp);
        fe = fe > 120 ? 120 : (fe < -120 ? -120 : fe);
        const float fin = duo_pow2(fe);
        float* const out = gW + (size_t)b * M * M;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int j = 16 * (FT::C0 + c) + li;
                const float v[4] = {gw[r][c].x, gw[r][c].y, gw[r][c].z, gw[r][c].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 16 * (FT::R0 + r) + 4 * lg + e;
                    if (i < M && j < M) out[(size_t)i * M + j] = v[e] * fin;
                }
            }
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
