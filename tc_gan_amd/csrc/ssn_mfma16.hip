// The SSN recurrence for NB >= 4 stimuli per weight draw (generator forward first) with W . r on the fp16 matrix cores
// as an EXACT-PRODUCT SPLIT: fp32 operands are carried as sums of fp16 numbers, every partial product of two fp16
// numbers is exact in the fp32 accumulator, and the fp16 MFMA runs at 16x the rate of the fp32 one that
// ssn_mfma.hip uses (v_mfma_f32_16x16x32_f16: 16 cycles per 16 x 16 x 32 tile per SIMD).
//
//   r (the state, fp32 in the serial waves' registers)  =  r_h + r_m + r_l    three fp16 numbers, EXACT
//       (11 + 11 + 2 significant bits by truncation; below 2^-24 / rscale in absolute value the tail is dropped);
//   W (stationary, registers of the matrix waves)       =  W_h + W_m          two fp16 numbers, 22 significant bits:
//       relative representation error <= 2^-22 = 2.4e-7 per element (fp32 itself: 6e-8) -- a third part would need
//       half as many registers again (273 per lane) and half as many MFMAs again.  That is a fixed perturbation of the
//       weights four times the fp32 rounding of the W-build itself and below the error of the fp32 power law
//       k v^n = exp2(n log2 v + log2 k) of the same step (tests/test_generator_gpu.py measures both against fp64).
//   All parts share one power-of-two scale per operand (W: per draw, from max |W|, found in the prologue; r: from the
//   rate bound of the saturating I/O function), so the six products land in one accumulator and parts of small elements
//   that fall below the fp16 range only lose bits that are below 2^-24 of the largest element.
//
// B operand = 16 columns = 4 stimuli x [r_h, r_m, r_l, (unused)], so one MFMA with A = W_h and one with A = W_m give all
// six products of a 16 x 32 tile of W for one group of 4 stimuli; the serial lane of (row quad, stimulus) adds the three
// part columns.  The two 4-stimulus groups of a draw run half a step apart exactly as in ssn_mfma.hip: in phase p the
// matrix waves run the chain of (group p & 1, step p >> 1) while the serial waves finish (group (p - 1) & 1,
// step (p - 1) >> 1): nonlinearity, Euler update, windowed reductions, trajectory stores, split of the new state.
//
// Workgroup = 4 matrix waves + 4 serial waves (one of each per SIMD, 256 registers).  The NRT x NKT tiles of the padded
// W (13 x 7 at 2N = 200) are dealt to the matrix waves in row-major order, 22-23 each, so a row tile is finished by
// one wave or by two neighbours; the second partial sum of a shared row tile goes to an "extra" slot that the serial
// lanes of that tile add (all others read a row of zeros).
//
// Only the saturating I/O function (asym_tanh, the default of every driver) bounds the rates, hence the choice of the
// r scale; the other two run the fp32 MFMA kernel.
//
// Three kernels share the matrix-wave code: gen_forward_split_kernel (this comment), gen_backward_split_kernel (adjoint
// sweep: W^T, delta with a scale that follows max |delta| step by step; any I/O function) and solve_split_kernel (the
// fixed-point solver with its stop protocol; scale from max(rate bound, max |r0|)).  Forward and solver also exist in a
// WIDE form (gen_forward_wide_kernel, solve_wide_kernel: all 8 stimuli of a draw in one chain per step, the state as two
// fp16 parts), which is what two-group launches run by default.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "ssn_device.h"
#include "ssn_host.h"
#include "ssn_mfma_io.h"

#ifndef SSN_SPLIT_ABLATE
#define SSN_SPLIT_ABLATE 0      // diagnostic builds: 1 = no serial part, 2 = no chains, 4 = sums not stored, 8 = one B tile
                                // read seven times (wrong results, timing only)
#endif

#ifndef SSN_SPLIT_STAMP
#define SSN_SPLIT_STAMP 0       // diagnostic build: cycle stamps of the adjoint's serial wave (ssn_debug_split_stamps)
#endif

namespace ssn {

#if SSN_SPLIT_STAMP
__device__ unsigned long long split_stamps[8];
#endif

typedef _Float16 hv8 __attribute__((ext_vector_type(8)));
typedef unsigned uv2 __attribute__((ext_vector_type(2)));

// a = 14 - floor(log2 max |W|): W 2^a fills [2^14, 2^15), the top of the fp16 range
__device__ __forceinline__ int split_w_exp(unsigned maxbits) {
    const int biased = (int)((maxbits >> 23) & 0xffu);
    const int a = 14 - ((biased ? biased : 1) - 127);
    return a > 100 ? 100 : (a < -100 ? -100 : a);
}
__device__ __forceinline__ float split_w_scale(unsigned maxbits) {
    return __builtin_bit_cast(float, (unsigned)(127 + split_w_exp(maxbits)) << 23);
}
// accumulator -> input current: 2^-(a + rshift)
__device__ __forceinline__ float split_u_scale(unsigned maxbits, int rshift) {
    return __builtin_bit_cast(float, (unsigned)(127 - split_w_exp(maxbits) - rshift) << 23);
}

typedef float fv2 __attribute__((ext_vector_type(2)));
// x * sc -> three fp16 parts by truncation (the masked values have 11 significant bits: their conversion is exact),
// written to the lane's k slots of the B operand (part columns 64 bytes apart); packed fp32 arithmetic throughout
__device__ __forceinline__ void split3_store(const float (&x)[4], fv2 sc01, fv2 sc23, unsigned bw) {
    using LdsU2 = __attribute__((address_space(3))) uv2*;
    const fv2 s01 = (fv2){x[0], x[1]} * sc01, s23 = (fv2){x[2], x[3]} * sc23;
    auto top = [](fv2 v) { return __builtin_bit_cast(fv2, __builtin_bit_cast(uv2, v) & (uv2){0xffffe000u, 0xffffe000u}); };
    auto pk = [](fv2 v) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v.x, v.y)); };
    const fv2 h01 = top(s01), h23 = top(s23);
    const fv2 d01 = s01 - h01, d23 = s23 - h23;
    const fv2 m01 = top(d01), m23 = top(d23);
    const fv2 l01 = d01 - m01, l23 = d23 - m23;
    *(LdsU2)(size_t)bw = (uv2){pk(h01), pk(h23)};
    *(LdsU2)(size_t)(bw + 64u) = (uv2){pk(m01), pk(m23)};
    *(LdsU2)(size_t)(bw + 128u) = (uv2){pk(l01), pk(l23)};
}

template <int MK>
struct Split16 {
    static constexpr int NRT = (MK + 15) / 16, NKT = (MK + 31) / 32;     // row tiles (16 rows), k tiles (32 columns)
    static constexpr int UNITS = NRT * NKT;
    static constexpr int WM = 4;                                         // matrix waves = serial waves
    static constexpr int start(int w) { return UNITS * w / WM; }         // first unit (rt * NKT + kt) of matrix wave w
    static constexpr int MAXU = (UNITS + WM - 1) / WM;
    // LDS rows = (tile, 16-lane group of the MFMA layout) x 16 columns x 16 B.  The two buffers are banked for their
    // ds_read_b128 side, whose lane groups are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS):
    // the matrix lanes' B-tile read takes columns 0-3, 12-15 of one row and 4-11 of the next per group -- conflict-free on
    // rows of exactly 256 B (BROW); the serial lanes' column read of the sums takes 4 columns of rows {0,3,5,6} / {1,2,4,7}
    // of an 8-row run per group -- conflict-free with 64 B of skew per row (ROW).
    static constexpr int BROW = 256;
    static constexpr int ROW = 320;
    static constexpr int BB = NKT * 4 * BROW;        // B operand of one group: [k tile][k octet][column][8 fp16]
    static constexpr int AB = NRT * 4 * ROW;         // sums of one group:      [row tile][row quad][column][4 fp32]
    static constexpr int XB = (WM - 1) * 4 * ROW;    // second partial sums of the row tiles shared by waves w-1 | w
    static constexpr int LDS = 2 * BB + 2 * AB + 2 * XB + ROW + 16;
    static_assert(NRT * 16 <= 256, "serial lanes: 4 waves x 64 lanes x 4 rows");
    static_assert(start(1) >= NKT, "a row tile is shared by at most two waves");
};

// One matrix wave: its slice of W (TRANSPOSED: of W^T, for the adjoint sweep) in registers for the whole launch, one
// chain per phase p in [p0, p1) for group (p - p0) & 1, a barrier at the end of each of the `nphase` phases.
template <int MK, int WV, bool TRANSPOSED>
__device__ __forceinline__ void split_matrix_wave(const float* __restrict__ Wd, int M, int lane, int p0, int p1, int nphase,
                                                  int gpw, char* bbuf, char* abuf, char* xbuf, unsigned* wmax) {
    using S = Split16<MK>;
    constexpr int U0 = S::start(WV), U1 = S::start(WV + 1), NU = U1 - U0;
    constexpr int RT0 = U0 / S::NKT, RT1 = (U1 - 1) / S::NKT, NT = RT1 - RT0 + 1;
    constexpr bool HEAD_SHARED = (U0 % S::NKT) != 0;       // my first row tile was started by wave WV - 1
    const int li = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wd), 0, M * M * 4, 0x00020000);

    // the lane's 8 elements of unit u: A[16 rt + li][32 kt + 8 lg .. + 7], A = W or W^T, zero outside M x M
    auto fetch = [&](int u, float (&w)[8]) {
        const int row = 16 * (u / S::NKT) + li, k0 = 32 * (u % S::NKT) + 8 * lg;
        const int rowc = row < M ? row : M - 1;
        float v[8];
        if constexpr (!TRANSPOSED) {
            const int voff = (rowc * M + k0) * 4;
            const mf4 lo = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
            const mf4 hi = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16, 0, 0));
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {          // A[row][k] = W[k][row]: 64-byte runs over the 16 lanes of a row tile
                const int k = k0 + e < M ? k0 + e : M - 1;
                v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (k * M + rowc) * 4, 0, 0));
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = (row < M && k0 + e < M) ? v[e] : 0.f;
    };
    // pass 1: max |W| of the draw (non-negative floats order like their bit patterns)
    float mx = 0.f;
    for (int u = U0; u < U1; ++u) {
        float w[8];
        fetch(u, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, __builtin_fabsf(w[e]));
    }
    atomicMax(wmax, __builtin_bit_cast(unsigned, mx));
    __syncthreads();                                                          // (A)
    const float sa = split_w_scale(*wmax);
    // pass 2: W 2^a = W_h + W_m
    hv8 Ah[NU], Am[NU];
#pragma unroll
    for (int ui = 0; ui < NU; ++ui) {
        float w[8];
        fetch(U0 + ui, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s = w[e] * sa;
            const _Float16 h = (_Float16)s;
            Ah[ui][e] = h;
            Am[ui][e] = (_Float16)(s - (float)h);
        }
    }
    using LdsH8 = const __attribute__((address_space(3))) hv8*;
    using LdsF4 = __attribute__((address_space(3))) mf4*;
    const unsigned boff = (unsigned)(lg * S::ROW + li * 16), boff_b = (unsigned)(lg * S::BROW + li * 16);
    __syncthreads();                                                          // (B)
    for (int p = 0; p < nphase; ++p) {
        if (p >= p0 && p < p1 && ((p - p0) & 1) < gpw && !(SSN_SPLIT_ABLATE & 2)) {
            const int g = (p - p0) & 1;
            const unsigned bb = (unsigned)(size_t)(LdsH8)(bbuf + g * S::BB) + boff_b;
            hv8 bt[S::NKT];
#pragma unroll
            for (int kt = 0; kt < S::NKT; ++kt)
                bt[kt] = *(LdsH8)(size_t)(bb + (unsigned)(((SSN_SPLIT_ABLATE & 8) ? 0 : kt) * 4 * S::BROW));
            mf4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = (mf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < S::NKT; ++kt) {
#pragma unroll
                for (int part = 0; part < 2; ++part) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int u = (RT0 + t) * S::NKT + kt;
                        if (u >= U0 && u < U1)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(part ? Am[u - U0] : Ah[u - U0], bt[kt], acc[t], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                char* dst = (t == 0 && HEAD_SHARED) ? xbuf + g * S::XB + (WV - 1) * 4 * S::ROW
                                                    : abuf + g * S::AB + (RT0 + t) * 4 * S::ROW;
                if (!(SSN_SPLIT_ABLATE & 4) || acc[t].x == 12345.f) *(LdsF4)(size_t)((unsigned)(size_t)(LdsF4)dst + boff) = acc[t];
            }
        }
        __syncthreads();
    }
}

template <int MK, bool SAVE>
__global__ void __launch_bounds__(512, 2) gen_forward_split_kernel(GenFwdArgs<float> a, int rshift) {
    using S = Split16<MK>;
    __shared__ __align__(16) char lds[S::LDS];
    char* const bbuf = lds;
    char* const abuf = lds + 2 * S::BB;
    char* const xbuf = abuf + 2 * S::AB;
    char* const zrow = xbuf + 2 * S::XB;
    unsigned* const wmax = reinterpret_cast<unsigned*>(zrow + S::ROW);
    const int M = a.M, N = a.M / 2, T_ = a.seqlen;
    const int gpw = a.mfma_groups;                // stimulus groups of 4 in this workgroup (2; 1: group 1 idle)
    const int ngroups = (a.NB + 4 * gpw - 1) / (4 * gpw);
    const int b = blockIdx.x / ngroups;
    const int s0 = (blockIdx.x % ngroups) * 4 * gpw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < S::LDS / 4; c += blockDim.x) reinterpret_cast<unsigned*>(lds)[c] = 0u;
    __syncthreads();                              // (zeroed before any wave records max |W|)

    if (wave < S::WM) {
        const float* Wd = a.W + (size_t)b * M * M;
        switch (wave) {       // phases 0 .. 2T: chains in 0 .. 2T - 1
            case 0: split_matrix_wave<MK, 0, false>(Wd, M, lane, 0, 2 * T_, 2 * T_ + 1, gpw, bbuf, abuf, xbuf, wmax); break;
            case 1: split_matrix_wave<MK, 1, false>(Wd, M, lane, 0, 2 * T_, 2 * T_ + 1, gpw, bbuf, abuf, xbuf, wmax); break;
            case 2: split_matrix_wave<MK, 2, false>(Wd, M, lane, 0, 2 * T_, 2 * T_ + 1, gpw, bbuf, abuf, xbuf, wmax); break;
            default: split_matrix_wave<MK, 3, false>(Wd, M, lane, 0, 2 * T_, 2 * T_ + 1, gpw, bbuf, abuf, xbuf, wmax); break;
        }
        return;
    }

    // ================================ serial wave ================================
    const int sw = wave - S::WM;
    const int blk = lane >> 2, j = lane & 3;
    const int er = 64 * sw + 4 * blk;             // first of the 4 rows this lane finishes
    const IoSelect io(a.io);
    float eps[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) eps[v] = (er + v < N) ? a.eps_E : a.eps_I;
    bool live[2];
    float rc[2][4], ex[2][4], ta[2][4], dp[2][4], rpn[2][4];
    int toff[2];                                  // byte offset of (my stimulus, step 0, row er) within this draw's block
    const size_t blk_elems = (size_t)a.NB * T_ * M;
    __amdgpu_buffer_rsrc_t rs_traj, rs_df;
    if constexpr (SAVE) {
        rs_traj = __builtin_amdgcn_make_buffer_rsrc(a.traj + (size_t)b * blk_elems, 0, (int)(blk_elems * 4), 0x00020000);
        rs_df = __builtin_amdgcn_make_buffer_rsrc(a.df + (size_t)b * blk_elems, 0, (int)(blk_elems * 4), 0x00020000);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int s = s0 + 4 * g + j;
        live[g] = s < a.NB && g < gpw;
        toff[g] = (live[g] && er < M) ? (int)(((size_t)s * T_ * M + er) * 4) : -1;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            rc[g][v] = ta[g][v] = dp[g][v] = rpn[g][v] = 0.f;
            ex[g][v] = (live[g] && er + v < M) ? a.ext[((size_t)b * a.NB + s) * M + er + v] : 0.f;
        }
    }
    // LDS addresses of this lane (bytes, relative to a group's buffer)
    const int rt = er / 16 < S::NRT ? er / 16 : S::NRT - 1, rq = (er / 4) & 3;
    const unsigned a_off = (unsigned)((rt * 4 + rq) * S::ROW + j * 16);       // column 4 part + j: + 64 part
    int x_slot = -1;                                                          // extra slot of my row tile, if shared
#pragma unroll
    for (int w = 1; w < S::WM; ++w)
        if (S::start(w) % S::NKT != 0 && S::start(w) / S::NKT == rt) x_slot = w - 1;
    const unsigned x_off = (unsigned)((x_slot * 4 + rq) * S::ROW + j * 16);
    const bool b_live = er < 32 * S::NKT;
    const unsigned b_off = (unsigned)(((er / 32) * 4 + ((er & 31) >> 3)) * S::BROW + j * 16 + ((er & 7) >> 2) * 8);
    const float rs = __builtin_bit_cast(float, (unsigned)(127 + rshift) << 23);          // 2^rshift
    const fv2 rs01 = {er < M ? rs : 0.f, er + 1 < M ? rs : 0.f}, rs23 = {er + 2 < M ? rs : 0.f, er + 3 < M ? rs : 0.f};
    using LdsF4 = const __attribute__((address_space(3))) mf4*;
    using LdsU2 = __attribute__((address_space(3))) uv2*;

    __syncthreads();                                                          // (A) max |W| of the draw is known
    const float usc = split_u_scale(*wmax, rshift);                           // 2^-(a + rshift)

    auto serial = [&](auto G, auto WIN, int it) {
        constexpr int g = decltype(G)::value;
        constexpr bool win_on = decltype(WIN)::value;
        const unsigned ab = (unsigned)(size_t)(LdsF4)(abuf + g * S::AB) + a_off;
        const unsigned xb = x_slot < 0 ? (unsigned)(size_t)(LdsF4)zrow + (unsigned)(j * 16)
                                       : (unsigned)(size_t)(LdsF4)(xbuf + g * S::XB) + x_off;
        const mf4 p0 = *(LdsF4)(size_t)ab, p1 = *(LdsF4)(size_t)(ab + 64u), p2 = *(LdsF4)(size_t)(ab + 128u);
        const mf4 q0 = *(LdsF4)(size_t)xb, q1 = *(LdsF4)(size_t)(xb + 64u), q2 = *(LdsF4)(size_t)(xb + 128u);
        const mf4 acc = ((p0 + q0) + (p1 + q1)) + (p2 + q2);
        const float accs[4] = {acc.x, acc.y, acc.z, acc.w};
        const float win2 = (it > a.skip) ? 1.f : 0.f;
        float rnew[4], dfn[4] = {0.f, 0.f, 0.f, 0.f};
        float uu[4], ff[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) uu[v] = fmaf(accs[v], usc, ex[g][v]);
        io.template eval4<SAVE>(uu, ff, dfn);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float f = ff[v];
            const float r1 = fmaf(eps[v], f - rc[g][v], rc[g][v]);             // (1 - eps) r + eps f(u)
            const float dd = r1 - rc[g][v];
            if constexpr (win_on) {
                ta[g][v] += r1;
                rpn[g][v] += fmaxf(r1 - a.theta, 0.f);
                dp[g][v] = fmaf(win2 * dd, dd, dp[g][v]);
            }
            rc[g][v] = r1;
            rnew[v] = r1;              // (rows >= M: never stored, and split with scale 0)
        }
        if constexpr (SAVE) {
            if (er + 3 < M) {       // whole quad inside the matrix (always when M % 4 == 0)
                const int off = toff[g] < 0 ? -1 : toff[g] + it * M * 4;
                const mf4 rv = {rnew[0], rnew[1], rnew[2], rnew[3]}, dv = {dfn[0], dfn[1], dfn[2], dfn[3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(
                    unsigned __attribute__((ext_vector_type(4))), rv), rs_traj, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(
                    unsigned __attribute__((ext_vector_type(4))), dv), rs_df, off, 0, 0);
            } else {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int off = (toff[g] < 0 || er + v >= M) ? -1 : toff[g] + (it * M + v) * 4;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rnew[v]), rs_traj, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dfn[v]), rs_df, off, 0, 0);
                }
            }
        }
        // new state -> three fp16 parts, written where the matrix lanes read their B operands (rows >= M: scale 0)
        if (b_live) split3_store(rnew, rs01, rs23, (unsigned)(size_t)(LdsU2)(bbuf + g * S::BB) + b_off);
    };
    constexpr std::integral_constant<int, 0> G0{};
    constexpr std::integral_constant<int, 1> G1{};
    __syncthreads();                                  // (B)
    __syncthreads();                                  // phase 0: nothing to finish yet
    constexpr std::integral_constant<bool, false> W0{};
    constexpr std::integral_constant<bool, true> W1{};
    const int nskip = a.skip < T_ ? (a.skip > 0 ? a.skip : 0) : T_;
    for (int it = 0; it < nskip; ++it) {
        if (!(SSN_SPLIT_ABLATE & 1)) serial(G0, W0, it);     // phase 2 it + 1
        __syncthreads();
        if (!(SSN_SPLIT_ABLATE & 1) && gpw == 2) serial(G1, W0, it);     // phase 2 it + 2
        __syncthreads();
    }
    for (int it = nskip; it < T_; ++it) {
        if (!(SSN_SPLIT_ABLATE & 1)) serial(G0, W1, it);
        __syncthreads();
        if (!(SSN_SPLIT_ABLATE & 1) && gpw == 2) serial(G1, W1, it);
        __syncthreads();
    }

    const float inv = 1.f / (float)(T_ - a.skip);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (!live[g]) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if (er + v >= M) continue;
            const size_t o = ((size_t)b * a.NB + s0 + 4 * g + j) * M + er + v;
            a.time_avg[o] = ta[g][v] * inv;
            a.dyn_row[o] = dp[g][v];
            a.rate_row[o] = rpn[g][v];
        }
    }
}



// ---------------------------------------------------------------------------------------------------------------
// "Wide" form of the split forward: ALL 8 stimuli of a draw in the 16 operand columns -- column s = r_h, column 8 + s =
// r_m of stimulus s -- so one chain of 2 MFMAs per tile serves a whole step (the two-group form above spends 2 per tile
// per group of four).  RP = 3 keeps the state exact with a third MFMA per tile: A = W_h against a second operand whose
// columns s hold r_l (the product lands in the accumulator column of W_h . r_h); RP = 2 drops r_l: the state enters the
// product with 22 significant bits, like W.  No second group to alternate with: a step is [chain] barrier [serial parts
// of both groups] barrier, the matrix waves idle through the serial parts and vice versa -- fewer MFMAs against no
// overlap.  Two groups per workgroup only (a.mfma_groups == 2).
// ---------------------------------------------------------------------------------------------------------------
template <int RP>
__device__ __forceinline__ void wide_store(const float (&x)[4], fv2 sc01, fv2 sc23, unsigned bw, unsigned second) {
    using LdsU2 = __attribute__((address_space(3))) uv2*;
    const fv2 s01 = (fv2){x[0], x[1]} * sc01, s23 = (fv2){x[2], x[3]} * sc23;
    auto top = [](fv2 v) { return __builtin_bit_cast(fv2, __builtin_bit_cast(uv2, v) & (uv2){0xffffe000u, 0xffffe000u}); };
    auto pk = [](fv2 v) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v.x, v.y)); };
    if constexpr (RP == 3) {
        const fv2 h01 = top(s01), h23 = top(s23);
        const fv2 d01 = s01 - h01, d23 = s23 - h23;
        *(LdsU2)(size_t)bw = (uv2){pk(h01), pk(h23)};
        const fv2 m01 = top(d01), m23 = top(d23);
        *(LdsU2)(size_t)(bw + 128u) = (uv2){pk(m01), pk(m23)};
        *(LdsU2)(size_t)(bw + second) = (uv2){pk(d01 - m01), pk(d23 - m23)};
    } else {
        // two parts: both by ROUND TO NEAREST (h = rn(s), m = rn(s - h): |s - h - m| <= 2^-23 |s|, no bias; round 2
        // truncated both, 2^-21 and biased toward zero)
        typedef _Float16 hv2_ __attribute__((ext_vector_type(2)));
        const hv2_ hn01 = __builtin_convertvector(s01, hv2_), hn23 = __builtin_convertvector(s23, hv2_);
        const fv2 r01 = s01 - __builtin_convertvector(hn01, fv2), r23 = s23 - __builtin_convertvector(hn23, fv2);
        *(LdsU2)(size_t)bw = (uv2){__builtin_bit_cast(unsigned, hn01), __builtin_bit_cast(unsigned, hn23)};
        *(LdsU2)(size_t)(bw + 128u) = (uv2){__builtin_bit_cast(unsigned, __builtin_convertvector(r01, hv2_)),
                                            __builtin_bit_cast(unsigned, __builtin_convertvector(r23, hv2_))};
    }
}

template <int MK, int WV, int RP>
__device__ __forceinline__ void wide_matrix_wave(const float* __restrict__ Wd, int M, int lane, int T_, char* bbuf, char* abuf,
                                                 char* xbuf, unsigned* wmax) {
    using S = Split16<MK>;
    constexpr int U0 = S::start(WV), U1 = S::start(WV + 1), NU = U1 - U0;
    constexpr int RT0 = U0 / S::NKT, RT1 = (U1 - 1) / S::NKT, NT = RT1 - RT0 + 1;
    constexpr bool HEAD_SHARED = (U0 % S::NKT) != 0;
    const int li = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wd), 0, M * M * 4, 0x00020000);
    auto fetch = [&](int u, float (&w)[8]) {
        const int row = 16 * (u / S::NKT) + li, k0 = 32 * (u % S::NKT) + 8 * lg;
        const int voff = ((row < M ? row : M - 1) * M + k0) * 4;
        const mf4 lo = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
        const mf4 hi = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16, 0, 0));
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = (row < M && k0 + e < M) ? v[e] : 0.f;
    };
    float mx = 0.f;
    for (int u = U0; u < U1; ++u) {
        float w[8];
        fetch(u, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, __builtin_fabsf(w[e]));
    }
    atomicMax(wmax, __builtin_bit_cast(unsigned, mx));
    __syncthreads();                                                          // (A)
    const float sa = split_w_scale(*wmax);
    hv8 Ah[NU], Am[NU];
#pragma unroll
    for (int ui = 0; ui < NU; ++ui) {
        float w[8];
        fetch(U0 + ui, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sc = w[e] * sa;
            const _Float16 h = (_Float16)sc;
            Ah[ui][e] = h;
            Am[ui][e] = (_Float16)(sc - (float)h);
        }
    }
    using LdsH8 = const __attribute__((address_space(3))) hv8*;
    using LdsF4 = __attribute__((address_space(3))) mf4*;
    const unsigned boff = (unsigned)(lg * S::ROW + li * 16);
    const unsigned bb = (unsigned)(size_t)(LdsH8)bbuf + (unsigned)(lg * S::BROW + li * 16);
    __syncthreads();                                                          // (B)
    for (int it = 0; it < T_; ++it) {
        mf4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (mf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < S::NKT; ++kt) {
            const hv8 b1 = *(LdsH8)(size_t)(bb + (unsigned)(kt * 4 * S::BROW));
#pragma unroll
            for (int part = 0; part < 2; ++part) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int u = (RT0 + t) * S::NKT + kt;
                    if (u >= U0 && u < U1)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(part ? Am[u - U0] : Ah[u - U0], b1, acc[t], 0, 0, 0);
                }
            }
            if constexpr (RP == 3) {
                const hv8 b2 = *(LdsH8)(size_t)(bb + (unsigned)(S::BB + kt * 4 * S::BROW));
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int u = (RT0 + t) * S::NKT + kt;
                    if (u >= U0 && u < U1) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[u - U0], b2, acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            char* dst = (t == 0 && HEAD_SHARED) ? xbuf + (WV - 1) * 4 * S::ROW : abuf + (RT0 + t) * 4 * S::ROW;
            *(LdsF4)(size_t)((unsigned)(size_t)(LdsF4)dst + boff) = acc[t];
        }
        __syncthreads();                              // sums stored: the serial waves take over
        __syncthreads();                              // new state stored
    }
}

template <int MK, bool SAVE, int RP>
__global__ void __launch_bounds__(512, 2) gen_forward_wide_kernel(GenFwdArgs<float> a, int rshift) {
    using S = Split16<MK>;
    __shared__ __align__(16) char lds[S::LDS];          // (laid out for two groups; this kernel uses the first halves)
    char* const bbuf = lds;                               // B1: columns s (r_h) and 8 + s (r_m) of stimulus s; B2 = bbuf + BB: r_l
    char* const abuf = lds + 2 * S::BB;
    char* const xbuf = abuf + 2 * S::AB;
    char* const zrow = xbuf + 2 * S::XB;
    unsigned* const wmax = reinterpret_cast<unsigned*>(zrow + S::ROW);
    const int M = a.M, N = a.M / 2, T_ = a.seqlen;
    const int gpw = a.mfma_groups;                // stimulus groups of 4 in this workgroup (2; 1: group 1 idle)
    const int ngroups = (a.NB + 4 * gpw - 1) / (4 * gpw);
    const int b = blockIdx.x / ngroups;
    const int s0 = (blockIdx.x % ngroups) * 4 * gpw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < S::LDS / 4; c += blockDim.x) reinterpret_cast<unsigned*>(lds)[c] = 0u;
    __syncthreads();                              // (zeroed before any wave records max |W|)

    if (wave < S::WM) {
        const float* Wd = a.W + (size_t)b * M * M;
        switch (wave) {
            case 0: wide_matrix_wave<MK, 0, RP>(Wd, M, lane, T_, bbuf, abuf, xbuf, wmax); break;
            case 1: wide_matrix_wave<MK, 1, RP>(Wd, M, lane, T_, bbuf, abuf, xbuf, wmax); break;
            case 2: wide_matrix_wave<MK, 2, RP>(Wd, M, lane, T_, bbuf, abuf, xbuf, wmax); break;
            default: wide_matrix_wave<MK, 3, RP>(Wd, M, lane, T_, bbuf, abuf, xbuf, wmax); break;
        }
        return;
    }

    // ================================ serial wave ================================
    const int sw = wave - S::WM;
    const int blk = lane >> 2, j = lane & 3;
    const int er = 64 * sw + 4 * blk;             // first of the 4 rows this lane finishes
    const IoSelect io(a.io);
    float eps[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) eps[v] = (er + v < N) ? a.eps_E : a.eps_I;
    bool live[2];
    float rc[2][4], ex[2][4], ta[2][4], dp[2][4], rpn[2][4];
    int toff[2];                                  // byte offset of (my stimulus, step 0, row er) within this draw's block
    const size_t blk_elems = (size_t)a.NB * T_ * M;
    __amdgpu_buffer_rsrc_t rs_traj, rs_df;
    if constexpr (SAVE) {
        rs_traj = __builtin_amdgcn_make_buffer_rsrc(a.traj + (size_t)b * blk_elems, 0, (int)(blk_elems * 4), 0x00020000);
        rs_df = __builtin_amdgcn_make_buffer_rsrc(a.df + (size_t)b * blk_elems, 0, (int)(blk_elems * 4), 0x00020000);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int s = s0 + 4 * g + j;
        live[g] = s < a.NB && g < gpw;
        toff[g] = (live[g] && er < M) ? (int)(((size_t)s * T_ * M + er) * 4) : -1;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            rc[g][v] = ta[g][v] = dp[g][v] = rpn[g][v] = 0.f;
            ex[g][v] = (live[g] && er + v < M) ? a.ext[((size_t)b * a.NB + s) * M + er + v] : 0.f;
        }
    }
    // LDS addresses of this lane (bytes, relative to a group's buffer)
    const int rt = er / 16 < S::NRT ? er / 16 : S::NRT - 1, rq = (er / 4) & 3;
    const unsigned a_off = (unsigned)((rt * 4 + rq) * S::ROW + j * 16);       // column 4 part + j: + 64 part
    int x_slot = -1;                                                          // extra slot of my row tile, if shared
#pragma unroll
    for (int w = 1; w < S::WM; ++w)
        if (S::start(w) % S::NKT != 0 && S::start(w) / S::NKT == rt) x_slot = w - 1;
    const unsigned x_off = (unsigned)((x_slot * 4 + rq) * S::ROW + j * 16);
    const bool b_live = er < 32 * S::NKT;
    const unsigned b_off = (unsigned)(((er / 32) * 4 + ((er & 31) >> 3)) * S::BROW + j * 16 + ((er & 7) >> 2) * 8);
    const float rs = __builtin_bit_cast(float, (unsigned)(127 + rshift) << 23);          // 2^rshift
    const fv2 rs01 = {er < M ? rs : 0.f, er + 1 < M ? rs : 0.f}, rs23 = {er + 2 < M ? rs : 0.f, er + 3 < M ? rs : 0.f};
    using LdsF4 = const __attribute__((address_space(3))) mf4*;
    using LdsU2 = __attribute__((address_space(3))) uv2*;

    __syncthreads();                                                          // (A) max |W| of the draw is known
    const float usc = split_u_scale(*wmax, rshift);                           // 2^-(a + rshift)

    auto serial = [&](auto G, auto WIN, int it) {
        constexpr int g = decltype(G)::value;
        constexpr bool win_on = decltype(WIN)::value;
        // columns 4 g + j (W . r_h, and W_h . r_l on top when RP = 3) and 8 + 4 g + j (W . r_m) of the one sums buffer
        const unsigned ab = (unsigned)(size_t)(LdsF4)abuf + a_off + (unsigned)(64 * g);
        const unsigned xb = x_slot < 0 ? (unsigned)(size_t)(LdsF4)zrow + (unsigned)(j * 16)
                                       : (unsigned)(size_t)(LdsF4)xbuf + x_off + (unsigned)(64 * g);
        const mf4 p0 = *(LdsF4)(size_t)ab, p1 = *(LdsF4)(size_t)(ab + 128u);
        const mf4 q0 = *(LdsF4)(size_t)xb, q1 = *(LdsF4)(size_t)(xb + 128u);
        const mf4 acc = (p0 + q0) + (p1 + q1);
        const float accs[4] = {acc.x, acc.y, acc.z, acc.w};
        const float win2 = (it > a.skip) ? 1.f : 0.f;
        float rnew[4], dfn[4] = {0.f, 0.f, 0.f, 0.f};
        float uu[4], ff[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) uu[v] = fmaf(accs[v], usc, ex[g][v]);
        io.template eval4<SAVE>(uu, ff, dfn);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float f = ff[v];
            const float r1 = fmaf(eps[v], f - rc[g][v], rc[g][v]);             // (1 - eps) r + eps f(u)
            const float dd = r1 - rc[g][v];
            if constexpr (win_on) {
                ta[g][v] += r1;
                rpn[g][v] += fmaxf(r1 - a.theta, 0.f);
                dp[g][v] = fmaf(win2 * dd, dd, dp[g][v]);
            }
            rc[g][v] = r1;
            rnew[v] = r1;              // (rows >= M: never stored, and split with scale 0)
        }
        if constexpr (SAVE) {
            if (er + 3 < M) {       // whole quad inside the matrix (always when M % 4 == 0)
                const int off = toff[g] < 0 ? -1 : toff[g] + it * M * 4;
                const mf4 rv = {rnew[0], rnew[1], rnew[2], rnew[3]}, dv = {dfn[0], dfn[1], dfn[2], dfn[3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(
                    unsigned __attribute__((ext_vector_type(4))), rv), rs_traj, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(
                    unsigned __attribute__((ext_vector_type(4))), dv), rs_df, off, 0, 0);
            } else {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int off = (toff[g] < 0 || er + v >= M) ? -1 : toff[g] + (it * M + v) * 4;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rnew[v]), rs_traj, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dfn[v]), rs_df, off, 0, 0);
                }
            }
        }
        // new state -> three fp16 parts, written where the matrix lanes read their B operands (rows >= M: scale 0)
        if (b_live) wide_store<RP>(rnew, rs01, rs23, (unsigned)(size_t)(LdsU2)bbuf + b_off + (unsigned)(64 * g), (unsigned)S::BB);
    };
    constexpr std::integral_constant<int, 0> G0{};
    constexpr std::integral_constant<int, 1> G1{};
    __syncthreads();                                  // (B)
    constexpr std::integral_constant<bool, false> W0{};
    constexpr std::integral_constant<bool, true> W1{};
    const int nskip = a.skip < T_ ? (a.skip > 0 ? a.skip : 0) : T_;
    for (int it = 0; it < nskip; ++it) {
        __syncthreads();                              // the chain of step it is done
        serial(G0, W0, it);
        serial(G1, W0, it);
        __syncthreads();
    }
    for (int it = nskip; it < T_; ++it) {
        __syncthreads();
        serial(G0, W1, it);
        serial(G1, W1, it);
        __syncthreads();
    }

    const float inv = 1.f / (float)(T_ - a.skip);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (!live[g]) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if (er + v >= M) continue;
            const size_t o = ((size_t)b * a.NB + s0 + 4 * g + j) * M + er + v;
            a.time_avg[o] = ta[g][v] * inv;
            a.dyn_row[o] = dp[g][v];
            a.rate_row[o] = rpn[g][v];
        }
    }
}


// max over the wave of a non-negative float (bit patterns order like the values), in every lane's SGPR copy
__device__ __forceinline__ unsigned wave_max_bits(float x) {
    unsigned v = __builtin_bit_cast(unsigned, x);
    auto dpp_max = [&](auto CTRL, auto ROWMASK) {
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, decltype(CTRL)::value, decltype(ROWMASK)::value, 0xf, false);
        v = o > v ? o : v;
    };
    using I = std::integral_constant<int, 0>;
    (void)sizeof(I);
    dpp_max(std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
    dpp_max(std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
    dpp_max(std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});   // row_half_mirror
    dpp_max(std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});   // row_mirror: every lane = row max
    dpp_max(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast15 -> rows 1, 3
    dpp_max(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast31 -> rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Reverse-time adjoint sweep with W^T delta on the fp16 matrix cores: the matrix waves hold W^T as two fp16 parts
// (the forward's layout and scale), the serial waves own the adjoint state (the recurrence, HBM streams and the in-place
// shifted delta of gen_backward_mfma_kernel, ssn_mfma.hip) and hand delta_tau over as three fp16 parts, exact.
// delta has no a-priori bound, so its power-of-two scale follows the data: the scale of step tau is taken from
// max |delta_{tau+1}| of the same group (one step of lag keeps the reduction off the critical path: a wave maximum by DPP
// and one LDS atomic per wave, three rotating slots), placed at 2^7 so that delta may grow or shrink by 2^8 per step
// before bits are lost (a part that overflows saturates: v_cvt_pkrtz never produces infinity); the first step's scale
// comes from its own delta, found in the prologue.
template <int MK>
__global__ void __launch_bounds__(512, 2) gen_backward_split_kernel(GenBwdArgs<float> a) {
    using S = Split16<MK>;
    __shared__ __align__(16) char lds[S::LDS + 32];
    char* const bbuf = lds;
    char* const abuf = lds + 2 * S::BB;
    char* const xbuf = abuf + 2 * S::AB;
    char* const zrow = xbuf + 2 * S::XB;
    unsigned* const wmax = reinterpret_cast<unsigned*>(zrow + S::ROW);
    unsigned* const dmax = wmax + 4;              // [group][3 rotating slots]: max |delta| bit patterns
    const int M = a.M, N = a.M / 2, T_ = a.seqlen;
    const int gpw = a.mfma_groups;
    const int ngroups = (a.NB + 4 * gpw - 1) / (4 * gpw);
    const int b = blockIdx.x / ngroups;
    const int s0 = (blockIdx.x % ngroups) * 4 * gpw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < (S::LDS + 32) / 4; c += blockDim.x) reinterpret_cast<unsigned*>(lds)[c] = 0u;
    __syncthreads();

    if (wave < S::WM) {
        const float* Wd = a.W + (size_t)b * M * M;
        switch (wave) {       // phases 0 .. 2T - 1: chains in 1 .. 2T - 1 (phase 0 is the first serial part)
            case 0: split_matrix_wave<MK, 0, true>(Wd, M, lane, 1, 2 * T_, 2 * T_, gpw, bbuf, abuf, xbuf, wmax); break;
            case 1: split_matrix_wave<MK, 1, true>(Wd, M, lane, 1, 2 * T_, 2 * T_, gpw, bbuf, abuf, xbuf, wmax); break;
            case 2: split_matrix_wave<MK, 2, true>(Wd, M, lane, 1, 2 * T_, 2 * T_, gpw, bbuf, abuf, xbuf, wmax); break;
            default: split_matrix_wave<MK, 3, true>(Wd, M, lane, 1, 2 * T_, 2 * T_, gpw, bbuf, abuf, xbuf, wmax); break;
        }
        return;
    }

    const int sw = wave - S::WM;
    const int blk = lane >> 2, j = lane & 3;
    const int er = 64 * sw + 4 * blk;
    const float inv = 1.f / (float)(T_ - a.skip);
    float eps[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) eps[v] = (er + v < N) ? a.eps_E : a.eps_I;
    const size_t blk_elems = (size_t)a.NB * T_ * M;
    const __amdgpu_buffer_rsrc_t rs_traj =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.traj) + (size_t)b * blk_elems, 0, (int)(blk_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dlt =
        __builtin_amdgcn_make_buffer_rsrc(a.delta + (size_t)b * blk_elems, 0, (int)(blk_elems * 4), 0x00020000);
    const bool quad = er + 3 < M;
    bool live[2];
    int toff[2];
    float gta[2][4], carry[2][4], xn[2][4], xc[2][4], xm[2][4], dfc[2][4], dsum[2][4];
    float pxm[2][4], pdf[2][4];                   // loads in flight for the END of the group's next serial part
    int bused[2] = {0, 0};                        // the scale exponent the group's delta in LDS was written with
    unsigned lastref[2] = {0u, 0u};
    unsigned runmax[2] = {0u, 0u};                // max over the steps so far of max |delta_tau| of the group: GenBwdArgs::dmax
    auto load4 = [&](const __amdgpu_buffer_rsrc_t& rs, int off, float (&out)[4]) {
        if (quad) {
            const mf4 q = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w;
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v)
                out[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (off < 0 || er + v >= M) ? -1 : off + 4 * v, 0, 0));
        }
    };
    auto store4 = [&](const __amdgpu_buffer_rsrc_t& rs, int off, const float (&val)[4]) {
        if (quad) {
            const mf4 q = {val[0], val[1], val[2], val[3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(unsigned __attribute__((ext_vector_type(4))), q), rs, off, 0, 0);
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val[v]), rs, (off < 0 || er + v >= M) ? -1 : off + 4 * v, 0, 0);
        }
    };
    auto at_step = [&](int g, int t) { return toff[g] < 0 ? -1 : toff[g] + t * M * 4; };   // byte offset of index t
    using LdsAtom = __attribute__((address_space(3))) unsigned*;
    using LdsUC = volatile const __attribute__((address_space(3))) unsigned*;
    const unsigned dmax_addr = (unsigned)(size_t)(LdsAtom)dmax;
    auto slot = [&](int g, int tau) { return dmax_addr + 4u * (unsigned)(3 * g + (tau + 3) % 3); };
    // direct gradient of the loss w.r.t. x_tau inside the penalty window
    auto direct = [&](int g, int v, int tau) {
        float gg = gta[g][v] + ((xc[g][v] > a.theta) ? a.c_rate : 0.f);
        if (tau <= T_ - 1) gg -= 2.f * a.c_dyn * (xn[g][v] - xc[g][v]);
        if (tau >= a.skip + 2) gg += 2.f * a.c_dyn * (xc[g][v] - xm[g][v]);
        return gg;
    };
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int s = s0 + 4 * g + j;
        live[g] = s < a.NB && g < gpw;
        toff[g] = (live[g] && er < M) ? (int)(((size_t)s * T_ * M + er) * 4) : -1;
        const float zero4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            gta[g][v] = (live[g] && er + v < M) ? a.g_time_avg[((size_t)b * a.NB + s) * M + er + v] * inv : 0.f;
            carry[g][v] = dsum[g][v] = xn[g][v] = 0.f;
        }
        load4(rs_traj, at_step(g, T_ - 1), xc[g]);                                   // x_T
        if (T_ >= 2) load4(rs_traj, at_step(g, T_ - 2), xm[g]); else { for (int v = 0; v < 4; ++v) xm[g][v] = 0.f; }
        load4(rs_dlt, at_step(g, T_ - 1), dfc[g]);                                   // f'(u_T)
        store4(rs_dlt, at_step(g, T_ - 1), zero4);                                   // slot T-1 of the shifted delta stays zero
        if (T_ >= 3) load4(rs_traj, at_step(g, T_ - 3), pxm[g]); else { for (int v = 0; v < 4; ++v) pxm[g][v] = 0.f; }
        if (T_ >= 2) load4(rs_dlt, at_step(g, T_ - 2), pdf[g]); else { for (int v = 0; v < 4; ++v) pdf[g][v] = 0.f; }
        // the first step's scale: max |delta_T| (carry = 0), or of eps |a_T| 2^-20 if f' vanishes everywhere
        float m0 = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if (er + v >= M) continue;
            const float at = (T_ >= a.skip + 1) ? direct(g, v, T_) : 0.f;
            m0 = fmaxf(m0, fmaxf(__builtin_fabsf(eps[v] * dfc[g][v] * at), __builtin_fabsf(eps[v] * at) * 9.5367431640625e-07f));
        }
        const unsigned wm0 = wave_max_bits(m0);
        if (lane == 0 && g < gpw) __hip_atomic_fetch_max((LdsAtom)(size_t)slot(g, T_ + 1), wm0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // LDS addresses of this lane (as in the forward kernel)
    const int rt = er / 16 < S::NRT ? er / 16 : S::NRT - 1, rq = (er / 4) & 3;
    const unsigned a_off = (unsigned)((rt * 4 + rq) * S::ROW + j * 16);
    int x_slot = -1;
#pragma unroll
    for (int w = 1; w < S::WM; ++w)
        if (S::start(w) % S::NKT != 0 && S::start(w) / S::NKT == rt) x_slot = w - 1;
    const unsigned x_off = (unsigned)((x_slot * 4 + rq) * S::ROW + j * 16);
    const bool b_live = er < 32 * S::NKT;
    const unsigned b_off = (unsigned)(((er / 32) * 4 + ((er & 31) >> 3)) * S::BROW + j * 16 + ((er & 7) >> 2) * 8);
    using LdsF4 = const __attribute__((address_space(3))) mf4*;
    using LdsU2 = __attribute__((address_space(3))) uv2*;

    __syncthreads();                                                          // (A) max |W| and the first delta scale
    const int wexp = split_w_exp(*wmax);

    // serial part of (group g, step tau): first = no matrix result yet (tau == T)
    auto serial = [&](auto G, auto WIN, int tau) {
        constexpr int g = decltype(G)::value;
        constexpr bool win_on = decltype(WIN)::value;
        float nxm[4] = {0.f, 0.f, 0.f, 0.f}, ndf[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (win_on) {
            if (tau >= 4 && tau >= a.skip + 3) load4(rs_traj, at_step(g, tau - 4), nxm);
        }
        if (tau >= 3) load4(rs_dlt, at_step(g, tau - 3), ndf);
        // scale of this step's delta from the previous step's maximum (kept when that was exactly zero)
        const unsigned mprev = __builtin_amdgcn_readfirstlane(*(LdsUC)(size_t)slot(g, tau + 1));
        if (lane == 0) *(LdsAtom)(size_t)slot(g, tau + 2) = 0u;                 // next step's slot (last read a phase pair ago)
        const unsigned ref = mprev ? mprev : lastref[g];
        lastref[g] = ref;
        runmax[g] = runmax[g] > mprev ? runmax[g] : mprev;       // (delta_T .. delta_2 are stored; their maxima are read at tau = T - 1 .. 1)
        int bexp = 7 - ((int)((ref >> 23) & 0xffu) - 127);
        bexp = ref == 0u ? 0 : (bexp > 100 ? 100 : (bexp < -100 ? -100 : bexp));
        if (tau < T_) {
            const unsigned ab = (unsigned)(size_t)(LdsF4)(abuf + g * S::AB) + a_off;
            const unsigned xb = x_slot < 0 ? (unsigned)(size_t)(LdsF4)zrow + (unsigned)(j * 16)
                                           : (unsigned)(size_t)(LdsF4)(xbuf + g * S::XB) + x_off;
            const mf4 p0 = *(LdsF4)(size_t)ab, p1 = *(LdsF4)(size_t)(ab + 64u), p2 = *(LdsF4)(size_t)(ab + 128u);
            const mf4 q0 = *(LdsF4)(size_t)xb, q1 = *(LdsF4)(size_t)(xb + 64u), q2 = *(LdsF4)(size_t)(xb + 128u);
            const mf4 acc = ((p0 + q0) + (p1 + q1)) + (p2 + q2);                   // W^T delta_{tau+1} 2^(a + bused)
            const float usc = __builtin_bit_cast(float, (unsigned)(127 - wexp - bused[g]) << 23);
            carry[g][0] = fmaf(acc.x, usc, carry[g][0]); carry[g][1] = fmaf(acc.y, usc, carry[g][1]);
            carry[g][2] = fmaf(acc.z, usc, carry[g][2]); carry[g][3] = fmaf(acc.w, usc, carry[g][3]);
        }
        float delta[4], dm = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float gg = 0.f;
            if constexpr (win_on) gg = direct(g, v, tau);
            const float at = gg + carry[g][v];
            delta[v] = (er + v < M) ? eps[v] * dfc[g][v] * at : 0.f;
            dm = fmaxf(dm, __builtin_fabsf(delta[v]));
            carry[g][v] = fmaf(-eps[v], at, at);                                       // (1 - eps) a_t
            dsum[g][v] += delta[v];
            if constexpr (win_on) {
                xn[g][v] = xc[g][v]; xc[g][v] = xm[g][v]; xm[g][v] = pxm[g][v]; pxm[g][v] = nxm[v];
            }
            dfc[g][v] = pdf[g][v];
            pdf[g][v] = ndf[v];
        }
        {
            // The scale lags one step behind the data: an adjoint that grows by more than 2^8 within one step (unstable
            // draws, the unbounded I/O functions) would SATURATE in v_cvt_pkrtz and come out finite but clamped.  Such a
            // step is poisoned instead: NaN into the hand-over and into the delta stream, so that the gradient is NaN like
            // the fp32 kernels' overflow and the drivers' NaN guards see it.
            const float rs = live[g] ? __builtin_bit_cast(float, (unsigned)(127 + bexp) << 23) : 0.f;
            if (!(dm * rs < 65504.f)) {
                delta[0] = __builtin_nanf("");
                if (a.dmax) atomicMax(a.dmax + b, 0x7fc00000u);       // the draw's hand-over word says "poisoned" (NaN; host side counts them)
            }
            if (b_live) split3_store(delta, (fv2){rs, rs}, (fv2){rs, rs}, (unsigned)(size_t)(LdsU2)(bbuf + g * S::BB) + b_off);
        }
        bused[g] = bexp;
        const unsigned wm = wave_max_bits(live[g] ? dm : 0.f);
        if (lane == 0) __hip_atomic_fetch_max((LdsAtom)(size_t)slot(g, tau), wm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (tau >= 2) store4(rs_dlt, at_step(g, tau - 2), delta);                      // shifted: pairs with x_{tau-1}
    };
    constexpr std::integral_constant<int, 0> G0{};
    constexpr std::integral_constant<int, 1> G1{};
    constexpr std::integral_constant<bool, false> W0{};
    constexpr std::integral_constant<bool, true> W1{};
    __syncthreads();                                                          // (B)
    int tau = T_;
    for (; tau >= a.skip + 1 && tau >= 1; --tau) {    // window steps first (time runs backwards)
        serial(G0, W1, tau);                          // phase 2 (T - tau)
        __syncthreads();
        if (gpw == 2) serial(G1, W1, tau);            // phase 2 (T - tau) + 1
        __syncthreads();
    }
#if SSN_SPLIT_STAMP
    {   // diagnostic build: s_memtime per segment of the non-window steps, serial wave 0 of workgroup 0
        unsigned long long ts = 0, tb = 0; int n = 0;
        for (; tau >= 1; --tau) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            serial(G0, W0, tau);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            __syncthreads();
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            if (gpw == 2) serial(G1, W0, tau);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
            __syncthreads();
            const unsigned long long t4 = __builtin_amdgcn_s_memtime();
            ts += (t1 - t0) + (t3 - t2); tb += (t2 - t1) + (t4 - t3); ++n;
        }
        if (blockIdx.x == 0 && sw == 0 && lane == 0) { split_stamps[0] = ts; split_stamps[1] = tb; split_stamps[2] = (unsigned long long)n; }
    }
#else
    for (; tau >= 1; --tau) {
        serial(G0, W0, tau);
        __syncthreads();
        if (gpw == 2) serial(G1, W0, tau);
        __syncthreads();
    }
#endif
    // the bound ssn_gw.hip scales the draw's delta with (ssn_weight_grad_scaled_f32); the unscaled delta is what was stored,
    // whatever its fp16 image in LDS saturated to
    if (a.dmax && sw == 0 && lane == 0) atomicMax(a.dmax + b, runmax[0] > runmax[1] ? runmax[0] : runmax[1]);
    if (a.g_ext) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (!live[g]) continue;
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (er + v < M) a.g_ext[((size_t)b * a.NB + s0 + 4 * g + j) * M + er + v] = dsum[g][v];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fixed-point solver (Euler loop of tc_gan/ext/ssnode.c:69-187; contract, stop protocol and phase structure of
// solve_mfma_kernel, ssn_mfma.hip) with W . r on the fp16 matrix cores as above: W two fp16 parts, the state three.
// The state scale comes from max(rate_hard_bound, max |r0|) of the workgroup's (draw, 8 stimuli): with the saturating
// I/O function and dt <= tau every later state is a convex combination of values inside that bound.
// ---------------------------------------------------------------------------------------------------------------
template <int MK, int WV>
__device__ __forceinline__ void solve_split_matrix_wave(const SolveArgs<float>& a, int b, int lane, char* bbuf, char* abuf,
                                                        char* xbuf, unsigned* wmax, int (*flags)[8], int s0) {
    using S = Split16<MK>;
    constexpr int U0 = S::start(WV), U1 = S::start(WV + 1), NU = U1 - U0;
    constexpr int RT0 = U0 / S::NKT, RT1 = (U1 - 1) / S::NKT, NT = RT1 - RT0 + 1;
    constexpr bool HEAD_SHARED = (U0 % S::NKT) != 0;
    const int M = a.M, max_iter = a.st.max_iter;
    const int li = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W + (size_t)b * M * M), 0, M * M * 4, 0x00020000);
    auto fetch = [&](int u, float (&w)[8]) {
        const int row = 16 * (u / S::NKT) + li, k0 = 32 * (u % S::NKT) + 8 * lg;
        const int voff = ((row < M ? row : M - 1) * M + k0) * 4;
        const mf4 lo = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
        const mf4 hi = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16, 0, 0));
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = (row < M && k0 + e < M) ? v[e] : 0.f;
    };
    float mx = 0.f;
    for (int u = U0; u < U1; ++u) {
        float w[8];
        fetch(u, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, __builtin_fabsf(w[e]));
    }
    atomicMax(wmax, __builtin_bit_cast(unsigned, mx));
    __syncthreads();                                                          // (A)
    const float sa = split_w_scale(*wmax);
    hv8 Ah[NU], Am[NU];
#pragma unroll
    for (int ui = 0; ui < NU; ++ui) {
        float w[8];
        fetch(U0 + ui, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sc = w[e] * sa;
            const _Float16 h = (_Float16)sc;
            Ah[ui][e] = h;
            Am[ui][e] = (_Float16)(sc - (float)h);
        }
    }
    using LdsH8 = const __attribute__((address_space(3))) hv8*;
    using LdsF4 = __attribute__((address_space(3))) mf4*;
    const unsigned boff = (unsigned)(lg * S::ROW + li * 16), boff_b = (unsigned)(lg * S::BROW + li * 16);
    auto chain = [&](int g) {
        const unsigned bb = (unsigned)(size_t)(LdsH8)(bbuf + g * S::BB) + boff_b;
        hv8 bt[S::NKT];
#pragma unroll
        for (int kt = 0; kt < S::NKT; ++kt) bt[kt] = *(LdsH8)(size_t)(bb + (unsigned)(kt * 4 * S::BROW));
        mf4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (mf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < S::NKT; ++kt) {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int u = (RT0 + t) * S::NKT + kt;
                    if (u >= U0 && u < U1)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(part ? Am[u - U0] : Ah[u - U0], bt[kt], acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            char* dst = (t == 0 && HEAD_SHARED) ? xbuf + g * S::XB + (WV - 1) * 4 * S::ROW
                                                : abuf + g * S::AB + (RT0 + t) * 4 * S::ROW;
            *(LdsF4)(size_t)((unsigned)(size_t)(LdsF4)dst + boff) = acc[t];
        }
    };
    // bookkeeping identical in every wave (see solve_mfma_kernel)
    bool my_frozen = lane >= 8 || s0 + lane >= a.NB;
    unsigned frozen = (unsigned)__builtin_amdgcn_ballot_w64(my_frozen) & 0xffu;
    auto verdict = [&](int g, int f) {
        const bool fnc = (f & 0xffff) != 0, fhb = (f >> 16) != 0;
        const bool stop = lane < 8 && (lane >> 2) == g && !my_frozen && (!fnc || fhb);
        my_frozen = my_frozen || stop;
        frozen = (unsigned)__builtin_amdgcn_ballot_w64(my_frozen) & 0xffu;
    };
    __syncthreads();                                                          // (B)
    for (int it = 0; it <= max_iter; ++it) {
        int f4 = 0;
        if (it >= 1) f4 = flags[(it - 1) % 3][lane & 7];
        if (it < max_iter) chain(0);
        if (it >= 1) verdict(0, f4);
        if (frozen == 0xffu) break;
        __syncthreads();
        if (it >= 1) f4 = flags[(it - 1) % 3][lane & 7];
        if (it < max_iter) chain(1);
        if (it >= 1) verdict(1, f4);
        if (frozen == 0xffu) break;
        __syncthreads();
    }
}

template <int MK>
__global__ void __launch_bounds__(512, 2) solve_split_kernel(SolveArgs<float> a) {
    using S = Split16<MK>;
    __shared__ __align__(16) char lds[S::LDS + 128];
    char* const bbuf = lds;
    char* const abuf = lds + 2 * S::BB;
    char* const xbuf = abuf + 2 * S::AB;
    char* const zrow = xbuf + 2 * S::XB;
    unsigned* const wmax = reinterpret_cast<unsigned*>(zrow + S::ROW);       // [0] max |W|, [1] max |r0|
    int (*flags)[8] = reinterpret_cast<int (*)[8]>(zrow + S::ROW + 32);       // [3][8]
    const int M = a.M, N = a.N, max_iter = a.st.max_iter;
    const int ngroups = (a.NB + 7) / 8;
    const int b = blockIdx.x / ngroups;
    const int s0 = (blockIdx.x % ngroups) * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < (S::LDS + 128) / 4; c += blockDim.x) reinterpret_cast<unsigned*>(lds)[c] = 0u;
    __syncthreads();

    if (wave < S::WM) {
        switch (wave) {
            case 0: solve_split_matrix_wave<MK, 0>(a, b, lane, bbuf, abuf, xbuf, wmax, flags, s0); break;
            case 1: solve_split_matrix_wave<MK, 1>(a, b, lane, bbuf, abuf, xbuf, wmax, flags, s0); break;
            case 2: solve_split_matrix_wave<MK, 2>(a, b, lane, bbuf, abuf, xbuf, wmax, flags, s0); break;
            default: solve_split_matrix_wave<MK, 3>(a, b, lane, bbuf, abuf, xbuf, wmax, flags, s0); break;
        }
        return;
    }

    // ================================ serial wave ================================
    const int sw = wave - S::WM;
    const int blk = lane >> 2, j = lane & 3;
    const int er = 64 * sw + 4 * blk;
    int my_code = 1, my_steps = max_iter;
    bool my_frozen = lane >= 8 || s0 + lane >= a.NB;
    unsigned frozen = (unsigned)__builtin_amdgcn_ballot_w64(my_frozen) & 0xffu;
    auto verdict = [&](int g, int it, int f) {
        const bool fnc = (f & 0xffff) != 0, fhb = (f >> 16) != 0;
        const bool stop = lane < 8 && (lane >> 2) == g && !my_frozen && (!fnc || fhb);
        my_code = stop ? (fnc ? 2 : 0) : my_code;
        my_steps = stop ? it + 1 : my_steps;
        my_frozen = my_frozen || stop;
        frozen = (unsigned)__builtin_amdgcn_ballot_w64(my_frozen) & 0xffu;
    };
    float eps[4], rc[2][4], rp[2][4], ex[2][4];
    bool live[2], rowok[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { eps[v] = (er + v < N) ? a.st.eps_E : a.st.eps_I; rowok[v] = er + v < M; }
    float r0max = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int s = s0 + 4 * g + j;
        live[g] = s < a.NB;
        const size_t vec = ((size_t)b * a.NB + (live[g] ? s : 0)) * M;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bool ok = live[g] && er + v < M;
            rc[g][v] = rp[g][v] = ok ? a.r[vec + er + v] : 0.f;
            ex[g][v] = ok ? a.ext[(a.ext_per_draw ? vec : (size_t)s * M) + er + v] : 0.f;
            r0max = fmaxf(r0max, __builtin_fabsf(rc[g][v]));
        }
    }
    {
        const unsigned wm0 = wave_max_bits(r0max);
        if (lane == 0) atomicMax(wmax + 1, wm0);
    }
    const IoSelect io(a.io);
    const int rt = er / 16 < S::NRT ? er / 16 : S::NRT - 1, rq = (er / 4) & 3;
    const unsigned a_off = (unsigned)((rt * 4 + rq) * S::ROW + j * 16);
    int x_slot = -1;
#pragma unroll
    for (int w = 1; w < S::WM; ++w)
        if (S::start(w) % S::NKT != 0 && S::start(w) / S::NKT == rt) x_slot = w - 1;
    const unsigned x_off = (unsigned)((x_slot * 4 + rq) * S::ROW + j * 16);
    const bool b_live = er < 32 * S::NKT;
    const unsigned b_off = (unsigned)(((er / 32) * 4 + ((er & 31) >> 3)) * S::BROW + j * 16 + ((er & 7) >> 2) * 8);
    using LdsF4 = const __attribute__((address_space(3))) mf4*;
    using LdsU2 = __attribute__((address_space(3))) uv2*;
    __syncthreads();                                                          // (A) max |W|, max |r0|
    // state scale: bound 2^rshift < 2^14 for bound = max(rate_hard_bound, max |r0|)
    const float bound = fmaxf(a.io.hard, __builtin_bit_cast(float, wmax[1]));
    int rshift = 13 - ((int)((__builtin_bit_cast(unsigned, bound) >> 23) & 0xffu) - 127);
    rshift = rshift > 100 ? 100 : (rshift < -100 ? -100 : rshift);
    const float usc = split_u_scale(*wmax, rshift);
    const float rs = __builtin_bit_cast(float, (unsigned)(127 + rshift) << 23);
    const fv2 rs01 = {er < M ? rs : 0.f, er + 1 < M ? rs : 0.f}, rs23 = {er + 2 < M ? rs : 0.f, er + 3 < M ? rs : 0.f};
#pragma unroll
    for (int g = 0; g < 2; ++g)
        if (b_live) split3_store(rc[g], rs01, rs23, (unsigned)(size_t)(LdsU2)(bbuf + g * S::BB) + b_off);
    auto serial = [&](auto G, int it) {
        constexpr int g = decltype(G)::value;
        const unsigned ab = (unsigned)(size_t)(LdsF4)(abuf + g * S::AB) + a_off;
        const unsigned xb = x_slot < 0 ? (unsigned)(size_t)(LdsF4)zrow + (unsigned)(j * 16)
                                       : (unsigned)(size_t)(LdsF4)(xbuf + g * S::XB) + x_off;
        const mf4 p0 = *(LdsF4)(size_t)ab, p1 = *(LdsF4)(size_t)(ab + 64u), p2 = *(LdsF4)(size_t)(ab + 128u);
        const mf4 q0 = *(LdsF4)(size_t)xb, q1 = *(LdsF4)(size_t)(xb + 64u), q2 = *(LdsF4)(size_t)(xb + 128u);
        const mf4 acc = ((p0 + q0) + (p1 + q1)) + (p2 + q2);
        const float accs[4] = {acc.x, acc.y, acc.z, acc.w};
        float uu[4], ff[4], dummy[4], r1[4], dabs[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) uu[v] = fmaf(accs[v], usc, ex[g][v]);
        io.template eval4<false>(uu, ff, dummy);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            r1[v] = rc[g][v] + (-rc[g][v] + ff[v]) * eps[v];                   // ssnode.c:64-67
            dabs[v] = rowok[v] ? fabsf(r1[v] - rc[g][v]) : -1.f;
        }
        if (sw == 0 && lane < 4) flags[(it + 1) % 3][4 * g + lane] = 0;
        if (live[g] && !((frozen >> (4 * g + j)) & 1u) && er < M) {
            const float dmax = fmaxf(fmaxf(dabs[0], dabs[1]), fmaxf(dabs[2], dabs[3]));
            const float rmax = fmaxf(fmaxf(r1[0], r1[1]), fmaxf(r1[2], r1[3]));
            short* fw = reinterpret_cast<short*>(&flags[it % 3][4 * g + j]);
            if (dmax >= a.st.atol) fw[0] = 1;
            if (a.st.check_hard && rmax >= a.st.hard_stop) fw[1] = 1;
#pragma unroll
            for (int v = 0; v < 4; ++v) { rp[g][v] = rc[g][v]; rc[g][v] = rowok[v] ? r1[v] : rc[g][v]; }
            split3_store(rc[g], rs01, rs23, (unsigned)(size_t)(LdsU2)(bbuf + g * S::BB) + b_off);
        }
    };
    constexpr std::integral_constant<int, 0> G0{};
    constexpr std::integral_constant<int, 1> G1{};
    __syncthreads();                                                          // (B)
    for (int it = 0; it <= max_iter; ++it) {
        int f4 = 0;
        if (it >= 1) f4 = flags[(it - 1) % 3][lane & 7];                        // phase 2 it: serial part of (1, it - 1)
        if (it >= 1) serial(G1, it - 1);
        if (it >= 1) verdict(0, it - 1, f4);
        if (frozen == 0xffu) break;
        __syncthreads();
        if (it >= 1) f4 = flags[(it - 1) % 3][lane & 7];                        // phase 2 it + 1: serial part of (0, it)
        if (it < max_iter) serial(G0, it);
        if (it >= 1) verdict(1, it - 1, f4);
        if (frozen == 0xffu) break;
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (!live[g]) continue;
        const size_t unit = (size_t)b * a.NB + s0 + 4 * g + j;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if (er + v >= M) continue;
            a.r[unit * M + er + v] = rc[g][v];
            if (a.r_prev) a.r_prev[unit * M + er + v] = rp[g][v];
        }
    }
    if (sw == 0 && lane < 8 && s0 + lane < a.NB) {
        const size_t unit = (size_t)b * a.NB + s0 + lane;
        a.codes[unit] = my_code;
        if (a.steps) a.steps[unit] = my_steps;
    }
}


// Wide form of the split solver (see gen_forward_wide_kernel): all 8 stimuli in one chain per step, the state as two fp16
// parts; a step is [chain] barrier [serial parts of both groups] barrier, the stop flags of a step are read by every
// wave at the start of the next one.
template <int MK, int WV>
__device__ __forceinline__ void solve_wide_matrix_wave(const SolveArgs<float>& a, int b, int lane, char* bbuf, char* abuf,
                                                        char* xbuf, unsigned* wmax, int (*flags)[8], int s0) {
    using S = Split16<MK>;
    constexpr int U0 = S::start(WV), U1 = S::start(WV + 1), NU = U1 - U0;
    constexpr int RT0 = U0 / S::NKT, RT1 = (U1 - 1) / S::NKT, NT = RT1 - RT0 + 1;
    constexpr bool HEAD_SHARED = (U0 % S::NKT) != 0;
    const int M = a.M, max_iter = a.st.max_iter;
    const int li = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W + (size_t)b * M * M), 0, M * M * 4, 0x00020000);
    auto fetch = [&](int u, float (&w)[8]) {
        const int row = 16 * (u / S::NKT) + li, k0 = 32 * (u % S::NKT) + 8 * lg;
        const int voff = ((row < M ? row : M - 1) * M + k0) * 4;
        const mf4 lo = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
        const mf4 hi = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16, 0, 0));
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = (row < M && k0 + e < M) ? v[e] : 0.f;
    };
    float mx = 0.f;
    for (int u = U0; u < U1; ++u) {
        float w[8];
        fetch(u, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, __builtin_fabsf(w[e]));
    }
    atomicMax(wmax, __builtin_bit_cast(unsigned, mx));
    __syncthreads();                                                          // (A)
    const float sa = split_w_scale(*wmax);
    hv8 Ah[NU], Am[NU];
#pragma unroll
    for (int ui = 0; ui < NU; ++ui) {
        float w[8];
        fetch(U0 + ui, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sc = w[e] * sa;
            const _Float16 h = (_Float16)sc;
            Ah[ui][e] = h;
            Am[ui][e] = (_Float16)(sc - (float)h);
        }
    }
    using LdsH8 = const __attribute__((address_space(3))) hv8*;
    using LdsF4 = __attribute__((address_space(3))) mf4*;
    const unsigned boff = (unsigned)(lg * S::ROW + li * 16), boff_b = (unsigned)(lg * S::BROW + li * 16);
    auto chain = [&]() {
        const unsigned bb = (unsigned)(size_t)(LdsH8)bbuf + boff_b;
        hv8 bt[S::NKT];
#pragma unroll
        for (int kt = 0; kt < S::NKT; ++kt) bt[kt] = *(LdsH8)(size_t)(bb + (unsigned)(kt * 4 * S::BROW));
        mf4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (mf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < S::NKT; ++kt) {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int u = (RT0 + t) * S::NKT + kt;
                    if (u >= U0 && u < U1)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(part ? Am[u - U0] : Ah[u - U0], bt[kt], acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            char* dst = (t == 0 && HEAD_SHARED) ? xbuf + (WV - 1) * 4 * S::ROW : abuf + (RT0 + t) * 4 * S::ROW;
            *(LdsF4)(size_t)((unsigned)(size_t)(LdsF4)dst + boff) = acc[t];
        }
    };
    // bookkeeping identical in every wave (see solve_mfma_kernel)
    bool my_frozen = lane >= 8 || s0 + lane >= a.NB;
    unsigned frozen = (unsigned)__builtin_amdgcn_ballot_w64(my_frozen) & 0xffu;
    auto verdict = [&](int g, int f) {
        const bool fnc = (f & 0xffff) != 0, fhb = (f >> 16) != 0;
        const bool stop = lane < 8 && (lane >> 2) == g && !my_frozen && (!fnc || fhb);
        my_frozen = my_frozen || stop;
        frozen = (unsigned)__builtin_amdgcn_ballot_w64(my_frozen) & 0xffu;
    };
    __syncthreads();                                                          // (B)
    for (int it = 0; it <= max_iter; ++it) {
        if (it >= 1) {                                // flags of step it - 1, both groups: complete since the last barrier
            const int f4 = flags[(it - 1) % 3][lane & 7];
            verdict(0, f4);
            verdict(1, f4);
        }
        if (frozen == 0xffu) break;
        if (it < max_iter) chain();
        __syncthreads();                              // sums stored
        __syncthreads();                              // new states stored
    }
}

template <int MK>
__global__ void __launch_bounds__(512, 2) solve_wide_kernel(SolveArgs<float> a) {
    using S = Split16<MK>;
    __shared__ __align__(16) char lds[S::LDS + 128];
    char* const bbuf = lds;
    char* const abuf = lds + 2 * S::BB;
    char* const xbuf = abuf + 2 * S::AB;
    char* const zrow = xbuf + 2 * S::XB;
    unsigned* const wmax = reinterpret_cast<unsigned*>(zrow + S::ROW);       // [0] max |W|, [1] max |r0|
    int (*flags)[8] = reinterpret_cast<int (*)[8]>(zrow + S::ROW + 32);       // [3][8]
    const int M = a.M, N = a.N, max_iter = a.st.max_iter;
    const int ngroups = (a.NB + 7) / 8;
    const int b = blockIdx.x / ngroups;
    const int s0 = (blockIdx.x % ngroups) * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < (S::LDS + 128) / 4; c += blockDim.x) reinterpret_cast<unsigned*>(lds)[c] = 0u;
    __syncthreads();

    if (wave < S::WM) {
        switch (wave) {
            case 0: solve_wide_matrix_wave<MK, 0>(a, b, lane, bbuf, abuf, xbuf, wmax, flags, s0); break;
            case 1: solve_wide_matrix_wave<MK, 1>(a, b, lane, bbuf, abuf, xbuf, wmax, flags, s0); break;
            case 2: solve_wide_matrix_wave<MK, 2>(a, b, lane, bbuf, abuf, xbuf, wmax, flags, s0); break;
            default: solve_wide_matrix_wave<MK, 3>(a, b, lane, bbuf, abuf, xbuf, wmax, flags, s0); break;
        }
        return;
    }

    // ================================ serial wave ================================
    const int sw = wave - S::WM;
    const int blk = lane >> 2, j = lane & 3;
    const int er = 64 * sw + 4 * blk;
    int my_code = 1, my_steps = max_iter;
    bool my_frozen = lane >= 8 || s0 + lane >= a.NB;
    unsigned frozen = (unsigned)__builtin_amdgcn_ballot_w64(my_frozen) & 0xffu;
    auto verdict = [&](int g, int it, int f) {
        const bool fnc = (f & 0xffff) != 0, fhb = (f >> 16) != 0;
        const bool stop = lane < 8 && (lane >> 2) == g && !my_frozen && (!fnc || fhb);
        my_code = stop ? (fnc ? 2 : 0) : my_code;
        my_steps = stop ? it + 1 : my_steps;
        my_frozen = my_frozen || stop;
        frozen = (unsigned)__builtin_amdgcn_ballot_w64(my_frozen) & 0xffu;
    };
    float eps[4], rc[2][4], rp[2][4], ex[2][4];
    bool live[2], rowok[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { eps[v] = (er + v < N) ? a.st.eps_E : a.st.eps_I; rowok[v] = er + v < M; }
    float r0max = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int s = s0 + 4 * g + j;
        live[g] = s < a.NB;
        const size_t vec = ((size_t)b * a.NB + (live[g] ? s : 0)) * M;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bool ok = live[g] && er + v < M;
            rc[g][v] = rp[g][v] = ok ? a.r[vec + er + v] : 0.f;
            ex[g][v] = ok ? a.ext[(a.ext_per_draw ? vec : (size_t)s * M) + er + v] : 0.f;
            r0max = fmaxf(r0max, __builtin_fabsf(rc[g][v]));
        }
    }
    {
        const unsigned wm0 = wave_max_bits(r0max);
        if (lane == 0) atomicMax(wmax + 1, wm0);
    }
    const IoSelect io(a.io);
    const int rt = er / 16 < S::NRT ? er / 16 : S::NRT - 1, rq = (er / 4) & 3;
    const unsigned a_off = (unsigned)((rt * 4 + rq) * S::ROW + j * 16);
    int x_slot = -1;
#pragma unroll
    for (int w = 1; w < S::WM; ++w)
        if (S::start(w) % S::NKT != 0 && S::start(w) / S::NKT == rt) x_slot = w - 1;
    const unsigned x_off = (unsigned)((x_slot * 4 + rq) * S::ROW + j * 16);
    const bool b_live = er < 32 * S::NKT;
    const unsigned b_off = (unsigned)(((er / 32) * 4 + ((er & 31) >> 3)) * S::BROW + j * 16 + ((er & 7) >> 2) * 8);
    using LdsF4 = const __attribute__((address_space(3))) mf4*;
    using LdsU2 = __attribute__((address_space(3))) uv2*;
    __syncthreads();                                                          // (A) max |W|, max |r0|
    // state scale: bound 2^rshift < 2^14 for bound = max(rate_hard_bound, max |r0|)
    const float bound = fmaxf(a.io.hard, __builtin_bit_cast(float, wmax[1]));
    int rshift = 13 - ((int)((__builtin_bit_cast(unsigned, bound) >> 23) & 0xffu) - 127);
    rshift = rshift > 100 ? 100 : (rshift < -100 ? -100 : rshift);
    const float usc = split_u_scale(*wmax, rshift);
    const float rs = __builtin_bit_cast(float, (unsigned)(127 + rshift) << 23);
    const fv2 rs01 = {er < M ? rs : 0.f, er + 1 < M ? rs : 0.f}, rs23 = {er + 2 < M ? rs : 0.f, er + 3 < M ? rs : 0.f};
#pragma unroll
    for (int g = 0; g < 2; ++g)
        if (b_live) wide_store<2>(rc[g], rs01, rs23, (unsigned)(size_t)(LdsU2)bbuf + b_off + (unsigned)(64 * g), 0u);
    auto serial = [&](auto G, int it) {
        constexpr int g = decltype(G)::value;
        const unsigned ab = (unsigned)(size_t)(LdsF4)abuf + a_off + (unsigned)(64 * g);      // columns 4 g + j and 8 + 4 g + j
        const unsigned xb = x_slot < 0 ? (unsigned)(size_t)(LdsF4)zrow + (unsigned)(j * 16)
                                       : (unsigned)(size_t)(LdsF4)xbuf + x_off + (unsigned)(64 * g);
        const mf4 p0 = *(LdsF4)(size_t)ab, p1 = *(LdsF4)(size_t)(ab + 128u);
        const mf4 q0 = *(LdsF4)(size_t)xb, q1 = *(LdsF4)(size_t)(xb + 128u);
        const mf4 acc = (p0 + q0) + (p1 + q1);
        const float accs[4] = {acc.x, acc.y, acc.z, acc.w};
        float uu[4], ff[4], dummy[4], r1[4], dabs[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) uu[v] = fmaf(accs[v], usc, ex[g][v]);
        io.template eval4<false>(uu, ff, dummy);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            r1[v] = rc[g][v] + (-rc[g][v] + ff[v]) * eps[v];                   // ssnode.c:64-67
            dabs[v] = rowok[v] ? fabsf(r1[v] - rc[g][v]) : -1.f;
        }
        if (sw == 0 && lane < 4) flags[(it + 1) % 3][4 * g + lane] = 0;
        if (live[g] && !((frozen >> (4 * g + j)) & 1u) && er < M) {
            const float dmax = fmaxf(fmaxf(dabs[0], dabs[1]), fmaxf(dabs[2], dabs[3]));
            const float rmax = fmaxf(fmaxf(r1[0], r1[1]), fmaxf(r1[2], r1[3]));
            short* fw = reinterpret_cast<short*>(&flags[it % 3][4 * g + j]);
            if (dmax >= a.st.atol) fw[0] = 1;
            if (a.st.check_hard && rmax >= a.st.hard_stop) fw[1] = 1;
#pragma unroll
            for (int v = 0; v < 4; ++v) { rp[g][v] = rc[g][v]; rc[g][v] = rowok[v] ? r1[v] : rc[g][v]; }
            wide_store<2>(rc[g], rs01, rs23, (unsigned)(size_t)(LdsU2)bbuf + b_off + (unsigned)(64 * g), 0u);
        }
    };
    constexpr std::integral_constant<int, 0> G0{};
    constexpr std::integral_constant<int, 1> G1{};
    __syncthreads();                                                          // (B)
    for (int it = 0; it <= max_iter; ++it) {
        if (it >= 1) {
            const int f4 = flags[(it - 1) % 3][lane & 7];
            verdict(0, it - 1, f4);
            verdict(1, it - 1, f4);
        }
        if (frozen == 0xffu) break;
        __syncthreads();                              // the chain of step it is done
        if (it < max_iter) { serial(G0, it); serial(G1, it); }
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (!live[g]) continue;
        const size_t unit = (size_t)b * a.NB + s0 + 4 * g + j;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if (er + v >= M) continue;
            a.r[unit * M + er + v] = rc[g][v];
            if (a.r_prev) a.r_prev[unit * M + er + v] = rp[g][v];
        }
    }
    if (sw == 0 && lane < 8 && s0 + lane < a.NB) {
        const size_t unit = (size_t)b * a.NB + s0 + lane;
        a.codes[unit] = my_code;
        if (a.steps) a.steps[unit] = my_steps;
    }
}

static int split_pick_mk(int M) {
    const int ladder[] = {104, 152, 208};
    for (int mk : ladder) if (M <= mk) return mk;
    return 0;
}

// rshift: r 2^rshift stays below the fp16 range for every reachable rate (r <= rate_hard_bound with the saturating
// I/O function and Euler factors <= 1); -1: the split kernel does not apply
int gen_split_rshift(const GenFwdArgs<float>& a) {
    if (a.io.io_type != SSN_IO_TANH || !(a.io.hard > 0.f) || !(a.io.hard < 3.0e4f) || !(a.io.soft >= 0.f)) return -1;
    if (!(a.eps_E > 0.f && a.eps_E <= 1.f && a.eps_I > 0.f && a.eps_I <= 1.f)) return -1;
    if ((a.M & 1) || a.NB < 4 || split_pick_mk(a.M) == 0) return -1;
    int sh = 0;
    while (sh < 14 && a.io.hard * (float)(2 << sh) <= 32768.f) ++sh;
    return sh;
}

// two-group launches: 2 = wide form with the state as two fp16 parts (default), 3 = wide form with three (exact state,
// three MFMAs per tile: no faster than the alternating form), 0 = alternating two-group form; SSN_FWD_WIDE overrides
int gen_split_wide_parts() {
    static const int rp = [] { const char* e = getenv("SSN_FWD_WIDE"); return (e && (e[0] == '0' || e[0] == '2' || e[0] == '3')) ? e[0] - '0' : 2; }();
    return rp;
}
template <int MK>
static hipError_t launch_split_mk(const GenFwdArgs<float>& a, int rshift, hipStream_t st) {
    const int ngroups = (a.NB + 4 * a.mfma_groups - 1) / (4 * a.mfma_groups);
    if (a.mfma_groups == 2 && !a.split_narrow && gen_split_wide_parts()) {
        const bool three = gen_split_wide_parts() == 3;
        if (a.traj) {
            if (three) hipLaunchKernelGGL((gen_forward_wide_kernel<MK, true, 3>), dim3(a.B * ngroups), dim3(512), 0, st, a, rshift);
            else hipLaunchKernelGGL((gen_forward_wide_kernel<MK, true, 2>), dim3(a.B * ngroups), dim3(512), 0, st, a, rshift);
        } else {
            if (three) hipLaunchKernelGGL((gen_forward_wide_kernel<MK, false, 3>), dim3(a.B * ngroups), dim3(512), 0, st, a, rshift);
            else hipLaunchKernelGGL((gen_forward_wide_kernel<MK, false, 2>), dim3(a.B * ngroups), dim3(512), 0, st, a, rshift);
        }
        return hipGetLastError();
    }
    if (a.traj) hipLaunchKernelGGL((gen_forward_split_kernel<MK, true>), dim3(a.B * ngroups), dim3(512), 0, st, a, rshift);
    else hipLaunchKernelGGL((gen_forward_split_kernel<MK, false>), dim3(a.B * ngroups), dim3(512), 0, st, a, rshift);
    return hipGetLastError();
}
hipError_t launch_gen_forward_split(const GenFwdArgs<float>& a, hipStream_t st) {
    const int rshift = gen_split_rshift(a);
    if (rshift < 0) return hipErrorInvalidValue;
    switch (split_pick_mk(a.M)) {
        case 104: return launch_split_mk<104>(a, rshift, st);
        case 152: return launch_split_mk<152>(a, rshift, st);
        case 208: return launch_split_mk<208>(a, rshift, st);
        default: return hipErrorInvalidValue;
    }
}

bool solve_split_supported(const SolveArgs<float>& a) {
    return a.io.io_type == SSN_IO_TANH && a.io.hard > 0.f && a.io.hard < 3.0e4f && a.io.soft >= 0.f && a.st.eps_E > 0.f &&
           a.st.eps_E <= 1.f && a.st.eps_I > 0.f && a.st.eps_I <= 1.f && !(a.M & 1) && a.NB >= 4 && split_pick_mk(a.M) != 0;
}
template <int MK>
static hipError_t launch_solve_split_mk(const SolveArgs<float>& a, hipStream_t st) {
    if (gen_split_wide_parts() && !a.split_narrow) hipLaunchKernelGGL((solve_wide_kernel<MK>), dim3(a.B * ((a.NB + 7) / 8)), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((solve_split_kernel<MK>), dim3(a.B * ((a.NB + 7) / 8)), dim3(512), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_solve_split(const SolveArgs<float>& a, hipStream_t st) {
    if (!solve_split_supported(a)) return hipErrorInvalidValue;
    switch (split_pick_mk(a.M)) {
        case 104: return launch_solve_split_mk<104>(a, st);
        case 152: return launch_solve_split_mk<152>(a, st);
        case 208: return launch_solve_split_mk<208>(a, st);
        default: return hipErrorInvalidValue;
    }
}

bool gen_split_backward_supported(int M, int NB) { return !(M & 1) && NB >= 4 && split_pick_mk(M) != 0; }

template <int MK>
static hipError_t launch_split_bwd_mk(const GenBwdArgs<float>& a, hipStream_t st) {
    const int ngroups = (a.NB + 4 * a.mfma_groups - 1) / (4 * a.mfma_groups);
    hipLaunchKernelGGL((gen_backward_split_kernel<MK>), dim3(a.B * ngroups), dim3(512), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_gen_backward_split(const GenBwdArgs<float>& a, hipStream_t st) {
    switch (split_pick_mk(a.M)) {
        case 104: return launch_split_bwd_mk<104>(a, st);
        case 152: return launch_split_bwd_mk<152>(a, st);
        case 208: return launch_split_bwd_mk<208>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ssn

#if SSN_SPLIT_STAMP
extern "C" int ssn_debug_split_stamps(unsigned long long* out8) {
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(ssn::split_stamps), 8 * sizeof(unsigned long long));
}
#endif
