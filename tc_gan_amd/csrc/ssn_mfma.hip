// MFMA form of the fixed-time SSN recurrence for SEVERAL stimuli per weight draw (fp32).
//
// Every real caller drives one W with NB = 8 stimuli (the 8 bandwidths of a tuning curve), so the per-step
// product W (M x M) * R (M x NB) is a small GEMM.  v_mfma_f32_4x4x1_16b_f32 maps it without waste:
//   16 blocks x (4 x 1) * (1 x 4):  A = one column of W for 64 rows (lane l <-> row l: the wave's 64-row slab of W
//   lives in VGPRs, one row per lane, M registers),  B = r[stimulus l % 4][k] (the lane's stimulus),  D = 4 VGPRs:
//   lane (blk = l / 4, j = l % 4) ends with the COMPLETE sums of rows 4 blk .. 4 blk + 3 for stimulus j.
// No cross-lane reduction, no partial sums; all 64 lanes finish 4 (row, stimulus) pairs each.  fp32 MFMA peak equals
// the packed-FMA VALU peak (157 TFLOP/s) but one wave per SIMD keeps the matrix pipe busy by itself, whereas the VALU
// tile kernels (ssn_tile.hip) are bound by per-wave issue, LDS traffic and their serial part:
// tools/microbench/mfma_matvec_rate.hip: 4195 cycles per step for 256 rows x 200 columns x 8 stimuli on one CU
// = 524 cycles per (draw, stimulus) step, against ~1000 for the split tile shape.
//
// Workgroup = 2 * ceil(M / 64) waves (slab form; K-split form: 3 * ceil(M / 64), see KSplit) for one (draw, group of
// 8 stimuli) = two MFMA groups of 4 stimuli, SPECIALISED:
//   * "matrix" waves 0 .. wm-1 hold the W slabs and do nothing but MFMA chains (4 interleaved partial chains per group),
//   * "serial" waves wm .. 2 wm - 1 (same lane -> (rows, stimulus) map, a few dozen registers) run the serial part of
//     every step: nonlinearity, Euler update, windowed reductions, trajectory stores, state write.
// A wave issues in order, one instruction every ~4 cycles, so MFMAs and the ~260 VALU instructions of the serial part
// do NOT overlap inside one wave (measured: 16.8 ms vs 7.9 ms without the serial part at the C3 shape, however the
// two were interleaved); issued from two different waves of the same SIMD they do.
// The two stimulus groups run HALF A STEP APART; phase p (one barrier each):
//   matrix waves: chain of (group p & 1, step p >> 1) -> accumulators to LDS (abuf)
//   serial waves: serial part of (group (p-1) & 1, step (p-1) >> 1) from abuf -> new state to LDS (rbuf)
// State in LDS: rbuf[2][8][MK+4], per stimulus contiguous in the neuron index; a lane reads the B operands of 4
// consecutive k with one ds_read_b128 (the 4 stimuli of a group sit on disjoint banks) and the serial lane writes its
// 4 finished rows with one ds_write_b128.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "ssn_device.h"
#include "ssn_host.h"
#include "ssn_mfma_io.h"

#ifndef SSN_MFMA_ABLATE
#define SSN_MFMA_ABLATE 0      // diagnostic builds: 1 = no serial part (wrong results, timing only)
#endif

namespace ssn {


// f(v) and f'(v) of ssn_gen.hip (kept in sync: ssnode.c:25-53 / ssnode.py:129-149)
__device__ __forceinline__ void mfma_io_eval_grad(float v, const IoConsts<float>& c, float& f, float& df) {
    if (!(v > 0.f)) { f = (v != v) ? v : 0.f; df = 0.f; return; }
    if (c.io_type == SSN_IO_POWER || v <= c.v0) { f = pow_rate(v, c.k, c.n); df = c.n * f / v; return; }
    if (c.io_type == SSN_IO_LINEAR) { f = c.soft + c.lin_slope * (v - c.v0); df = c.lin_slope; return; }
    const float th = tanh_pos(c.tanh_gain * (v - c.v0));
    f = c.soft + c.span * th;
    df = c.span_gain * (1.f - th * th);
}

// the wave's 64-row slab of the row-major M x M matrix A (TRANSPOSED: of its transpose): wr[k] = slab[lane][k]
template <int MK, bool TRANSPOSED>
__device__ __forceinline__ void slab_load(const float* A, int M, int row, float (&wr)[MK]) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, M * M * 4, 0x00020000);
    const int rowc = row < M ? row : M - 1;
    if constexpr (!TRANSPOSED) {
        const int voff = rowc * M * 4;
#pragma unroll
        for (int k4 = 0; k4 < MK; k4 += 4) {
            // (whole-vector bit_cast: see ssn_tile_core.h)
            const mf4 v = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, k4 * 4, 0));
            wr[k4] = v.x; wr[k4 + 1] = v.y; wr[k4 + 2] = v.z; wr[k4 + 3] = v.w;
        }
    } else {
        const int voff = rowc * 4;                       // element (row, k) = A[k][row]: coalesced over lanes
#pragma unroll
        for (int k = 0; k < MK; ++k)
            wr[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                rsrc, voff, __builtin_amdgcn_readfirstlane((k < M ? k : M - 1) * M * 4), 0));
    }
#pragma unroll
    for (int k = 0; k < MK; ++k) wr[k] = (row < M && k < M) ? wr[k] : 0.f;
}


// acc[v] = sum_k wr[k] * x[k] for the lane's stimulus row `xs` (LDS), four interleaved partial chains.  The order
// [read quad q + DEPTH][4 MFMAs of quad q] is pinned with data dependencies (empty asm on the LDS address and the
// accumulators): left alone, the compiler emits read - wait - use per quad.
template <int MK>
__device__ __forceinline__ mf4 slab_chain(const float (&wr)[MK], const float* xs) {
    constexpr int NQ = MK / 4, DEPTH = 4;         // B-operand reads run DEPTH quads (16 MFMAs) ahead of their use
    using LdsV4 = const __attribute__((address_space(3))) mf4*;
    using LdsF = const __attribute__((address_space(3))) float*;
    unsigned xa = (unsigned)(size_t)(LdsF)xs;
    mf4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    mf4 bq[NQ];
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) bq[q] = *(LdsV4)(size_t)(xa + 16u * q);
    asm volatile("" : "+v"(xa));
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q + DEPTH < NQ) bq[q + DEPTH] = *(LdsV4)(size_t)(xa + 16u * (q + DEPTH));
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * q], bq[q].x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * q + 1], bq[q].y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * q + 2], bq[q].z, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * q + 3], bq[q].w, a3, 0, 0, 0);
        asm volatile("" : "+v"(xa), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));     // VGPR ties: no AGPR split of the file
    }
    return (a0 + a1) + (a2 + a3);
}

// ---- balanced K-split operand layout --------------------------------------------------------------------
// The slab form gives every matrix wave 64 rows x MK columns: with M = 200 the fourth wave holds 8 real rows and the
// chain is 200 MFMAs long in every wave.  The rows of one MFMA block need not be related to the other blocks', and
// the four interleaved chains of a wave need not belong to the same rows, so the unit of work is a SLOT
// (wave, chain, block) = (row quad q, column part p): 4 rows x KP columns.  NQ = MK / 4 quads x P parts fill
// WM * 64 slots almost exactly (MK = 200: 50 x 5 = 250 of 256) and every wave runs 4 chains of KP = 40 MFMAs: 160
// per phase instead of 200.  A lane ends with P partial sums per (row, stimulus) spread over P slots; they go to LDS
// slot by slot and the serial lane of (quad, stimulus) adds them (P 16-byte reads).
template <int MK>
struct KSplit {
    static constexpr int SW = (MK + 63) / 64;                  // serial waves (64 lanes x 4 rows each)
    static constexpr int SLOTS = SW * 64 * 4 / 4;              // (wave, chain, block) slots: 4 chains of 16 blocks per 64 rows
    static constexpr int NQ = MK / 4;                          // row quads
    static constexpr int P = SLOTS / NQ;                       // column parts per quad
    static constexpr int KP = ((MK + P - 1) / P + 3) / 4 * 4;  // columns per part (16-byte B-operand reads)
    static constexpr int NS4 = KP / 4;
    static constexpr bool enabled = 4 * KP < MK;               // shorter chains than the slab form
    // The 4 chains per SIMD are issued by TWO matrix waves of 2 chains each (CH = 2, WM = 2 SW matrix waves): one wave
    // alone leaves the matrix pipe idle while it waits for its B operands from LDS (129 vs 143 TFLOP/s in
    // tools/microbench/mfma_matvec_rate.hip); 80 W registers per wave, 3 waves per SIMD with the serial wave.
    static constexpr int CH = 2;
    static constexpr int WM = SW * 4 / CH;                     // matrix waves
    static constexpr int THREADS = (WM + SW) * 64;
    static constexpr int RS = (P * KP > MK ? P * KP : MK) + 8; // state row stride: the (stimulus, part) reads of one
                                                               // wave instruction on disjoint banks (MK = 200, 152)
    // slot of (wave, chain, block) -> (part, quad); slots >= NQ * P idle (part P: all columns out of range)
    __device__ static __forceinline__ int slot(int wave, int c, int blk) { return (wave * CH + c) * 16 + blk; }
};

// wr[c][s] = A[4 q_c + i][p_c KP + s] (TRANSPOSED: A[p_c KP + s][4 q_c + i]) for lane (blk, i), zero outside M x M
template <int MK, bool TRANSPOSED>
__device__ __forceinline__ void ksplit_load(const float* A, int M, int wave, int lane, float (&wr)[KSplit<MK>::CH][KSplit<MK>::KP]) {
    using KS = KSplit<MK>;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, M * M * 4, 0x00020000);
#pragma unroll
    for (int c = 0; c < KS::CH; ++c) {
        const int u = KS::slot(wave, c, lane >> 2);
        const int part = u / KS::NQ, row = 4 * (u % KS::NQ) + (lane & 3);
        const int k0 = part * KS::KP;
        const int rowc = row < M ? row : M - 1;
        if constexpr (!TRANSPOSED) {
            const int voff = (rowc * M + k0) * 4;
#pragma unroll
            for (int s4 = 0; s4 < KS::NS4; ++s4) {
                const mf4 v = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, s4 * 16, 0));
                wr[c][4 * s4] = v.x; wr[c][4 * s4 + 1] = v.y; wr[c][4 * s4 + 2] = v.z; wr[c][4 * s4 + 3] = v.w;
            }
        } else {
            const int voff = (k0 * M + rowc) * 4;
#pragma unroll
            for (int s = 0; s < KS::KP; ++s)
                wr[c][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rsrc, voff, __builtin_amdgcn_readfirstlane(s * M * 4), 0));
        }
#pragma unroll
        for (int s = 0; s < KS::KP; ++s) wr[c][s] = (row < M && k0 + s < M) ? wr[c][s] : 0.f;
    }
}

// The chains of one wave in one phase: chain c multiplies its slot's 4 x KP block with x[k0_c .. k0_c + KP) of the
// lane's stimulus row (LDS byte address xa[c]) and stores the 4 partial sums to out[c] (LDS).  Read/MFMA order pinned as
// in slab_chain: the B operands of step group s4 + DEPTH are requested before the MFMAs of group s4.
template <int MK>
__device__ __forceinline__ void ksplit_chain(const float (&wr)[KSplit<MK>::CH][KSplit<MK>::KP], unsigned (&xa)[KSplit<MK>::CH],
                                             mf4* const (&out)[KSplit<MK>::CH]) {
    using KS = KSplit<MK>;
    constexpr int NS4 = KS::NS4, CH = KS::CH, DEPTH = 3;
    using LdsV4 = const __attribute__((address_space(3))) mf4*;
    mf4 acc[CH];
    mf4 bq[NS4][CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = (mf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < DEPTH && s4 < NS4; ++s4)
#pragma unroll
        for (int c = 0; c < CH; ++c) bq[s4][c] = *(LdsV4)(size_t)(xa[c] + 16u * s4);
#pragma unroll
    for (int c = 0; c < CH; ++c) asm volatile("" : "+v"(xa[c]));
#pragma unroll
    for (int s4 = 0; s4 < NS4; ++s4) {
        if (s4 + DEPTH < NS4) {
#pragma unroll
            for (int c = 0; c < CH; ++c) bq[s4 + DEPTH][c] = *(LdsV4)(size_t)(xa[c] + 16u * (s4 + DEPTH));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < CH; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[c][4 * s4 + e], bq[s4][c][e], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CH; ++c) asm volatile("" : "+v"(xa[c]), "+v"(acc[c]));
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) *out[c] = acc[c];
}

// serial side: complete sums of (quad q, stimulus j) = the P partial sums of slots p NQ + q
template <int MK>
__device__ __forceinline__ mf4 ksplit_gather(const mf4* ab /* [SLOTS][4] of one group */, int q, int j) {
    using KS = KSplit<MK>;
    const int qc = q < KS::NQ ? q : KS::NQ - 1;
    mf4 part[KS::P];
#pragma unroll
    for (int p = 0; p < KS::P; ++p) part[p] = ab[(p * KS::NQ + qc) * 4 + j];
    mf4 acc = part[0];
#pragma unroll
    for (int p = 1; p < KS::P; ++p) acc += part[p];
    return acc;
}

template <int MK, bool SAVE>
__global__ void __launch_bounds__(KSplit<MK>::enabled ? KSplit<MK>::THREADS : 512, KSplit<MK>::enabled ? 3 : 2) gen_forward_mfma_kernel(GenFwdArgs<float> a) {
    using KS = KSplit<MK>;
    constexpr bool KSP = KS::enabled;
    constexpr int RS = KSP ? KS::RS : MK + 4;     // LDS row stride: the 4 stimuli of a group on disjoint banks
    __shared__ __align__(16) float rbuf[2][8][RS];
    // accumulator hand-off: slab form [group][matrix wave][lane], K-split form [group][slot][stimulus]
    __shared__ __align__(16) mf4 abuf[2][KSP ? KS::SLOTS * 4 : 4 * 64];
    const int M = a.M, N = a.M / 2, T_ = a.seqlen;
    const int gpw = a.mfma_groups;                // stimulus groups of 4 in this workgroup (2; 1: group 1 idle)
    const int ngroups = (a.NB + 4 * gpw - 1) / (4 * gpw);
    const int b = blockIdx.x / ngroups;
    const int s0 = (blockIdx.x % ngroups) * 4 * gpw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = KSplit<MK>::enabled ? KSplit<MK>::WM : (int)(blockDim.x >> 7);     // matrix waves
    const int blk = lane >> 2, j = lane & 3;
    const int nphase = 2 * T_ + 1;
    for (int c = threadIdx.x; c < 2 * 8 * RS; c += blockDim.x) (&rbuf[0][0][0])[c] = 0.f;

    if (wave < wm) {
        // ================================ matrix wave ================================
        if constexpr (KSP) {
            float wr[KS::CH][KS::KP];
            ksplit_load<MK, false>(a.W + (size_t)b * M * M, M, wave, lane, wr);
            using LdsF = const __attribute__((address_space(3))) float*;
            const unsigned rb0 = (unsigned)(size_t)(LdsF)&rbuf[0][0][0];
            unsigned xoff[KS::CH];                // byte offset of (my stimulus row, first column of chain c's part)
#pragma unroll
            for (int c = 0; c < KS::CH; ++c) xoff[c] = (unsigned)((j * RS + (KS::slot(wave, c, blk) / KS::NQ) * KS::KP) * 4);
            __syncthreads();
            for (int p = 0; p < nphase; ++p) {
                if (p < 2 * T_ && (p & 1) < gpw && !(SSN_MFMA_ABLATE & 2)) {
                    const int g = p & 1, it = p >> 1;
                    const unsigned base = rb0 + (unsigned)((((it & 1) * 8 + 4 * g) * RS) * 4);
                    static_assert(KS::CH == 2, "operand lists below are written out for two chains per wave");
                    unsigned xa[2] = {base + xoff[0], base + xoff[1]};
                    mf4* const out[2] = {&abuf[g][KS::slot(wave, 0, blk) * 4 + j], &abuf[g][KS::slot(wave, 1, blk) * 4 + j]};
                    ksplit_chain<MK>(wr, xa, out);
                }
                __syncthreads();
            }
        } else {
            float wr[MK];
            slab_load<MK, false>(a.W + (size_t)b * M * M, M, 64 * wave + lane, wr);
            __syncthreads();
            for (int p = 0; p < nphase; ++p) {
                if (p < 2 * T_ && (p & 1) < gpw && !(SSN_MFMA_ABLATE & 2)) {
                    const int g = p & 1, it = p >> 1;
                    abuf[g][wave * 64 + lane] = slab_chain<MK>(wr, &rbuf[it & 1][4 * g + j][0]);
                }
                __syncthreads();
            }
        }
        return;
    }

    // ================================ serial wave ================================
    const int sw = wave - wm;
    const int er = 64 * sw + 4 * blk;             // first of the 4 rows this lane finishes
    const IoSelect io(a.io);
    float eps[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) eps[v] = (er + v < N) ? a.eps_E : a.eps_I;
    // per group g: my stimulus s0 + 4 g + j.  Trajectory stores go through a raw buffer whose out-of-range offsets
    // are dropped by the hardware: lanes without a real (stimulus, row) get offset -1 instead of a branch.
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
    // serial part of (group g, step it): u = acc + ext -> f, f' -> Euler step, windowed reductions, trajectory, state
    // WIN: step inside the averaging window (it >= skip); before it the three windowed sums are not touched
    auto serial = [&](auto G, auto WIN, int it) {
        constexpr int g = decltype(G)::value;
        constexpr bool win_on = decltype(WIN)::value;
        mf4 acc;
        if constexpr (KSP) acc = ksplit_gather<MK>(&abuf[g][0], 16 * sw + blk, j);
        else acc = abuf[g][sw * 64 + lane];
        const float accs[4] = {acc.x, acc.y, acc.z, acc.w};
        const float win2 = (it > a.skip) ? 1.f : 0.f;
        float rnew[4], dfn[4] = {0.f, 0.f, 0.f, 0.f};   // scalars: see the bit_cast note in ssn_tile_core.h
        float uu[4], ff[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) uu[v] = accs[v] + ex[g][v];
        if (SSN_MFMA_ABLATE & 4) { for (int v = 0; v < 4; ++v) ff[v] = uu[v] * 0.5f; } else io.template eval4<SAVE>(uu, ff, dfn);
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
            rnew[v] = (er + v < M) ? r1 : 0.f;
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
        if (er < M) *reinterpret_cast<mf4*>(&rbuf[(it + 1) & 1][4 * g + j][er]) = (mf4){rnew[0], rnew[1], rnew[2], rnew[3]};
    };
    constexpr std::integral_constant<int, 0> G0{};
    constexpr std::integral_constant<int, 1> G1{};
    __syncthreads();
    __syncthreads();                                  // phase 0: nothing to finish yet
    constexpr std::integral_constant<bool, false> W0{};
    constexpr std::integral_constant<bool, true> W1{};
    const int nskip = a.skip < T_ ? (a.skip > 0 ? a.skip : 0) : T_;
    for (int it = 0; it < nskip; ++it) {
        if (!(SSN_MFMA_ABLATE & 1)) serial(G0, W0, it);     // phase 2 it + 1
        __syncthreads();
        if (!(SSN_MFMA_ABLATE & 1) && gpw == 2) serial(G1, W0, it);     // phase 2 it + 2
        __syncthreads();
    }
    for (int it = nskip; it < T_; ++it) {
        if (!(SSN_MFMA_ABLATE & 1)) serial(G0, W1, it);
        __syncthreads();
        if (!(SSN_MFMA_ABLATE & 1) && gpw == 2) serial(G1, W1, it);
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

// Reverse-time adjoint sweep (same recurrence as gen_backward_kernel, ssn_gen.hip) in the specialised MFMA form:
// the matrix waves hold the slabs of W^T and compute W^T delta_tau, the serial waves own the adjoint state.
// Per group: serial(tau) [a_tau = g_tau + carry; delta_tau = eps f'(u_tau) a_tau -> LDS, HBM] -> matrix(tau) ->
// serial(tau - 1) [carry = (1 - eps) a_tau + W^T delta_tau] ...; group 0 serial phases are the even ones, group 1 the
// odd ones, the matrix waves serve the other group in every phase.
template <int MK>
__global__ void __launch_bounds__(KSplit<MK>::enabled ? KSplit<MK>::THREADS : 512, KSplit<MK>::enabled ? 3 : 2) gen_backward_mfma_kernel(GenBwdArgs<float> a) {
    using KS = KSplit<MK>;
    constexpr bool KSP = KS::enabled;
    constexpr int RS = KSP ? KS::RS : MK + 4;
    __shared__ __align__(16) float dbuf[8][RS];          // delta_tau per stimulus, contiguous in the neuron index
    __shared__ __align__(16) mf4 abuf[2][KSP ? KS::SLOTS * 4 : 4 * 64];    // hand-off, as in the forward kernel
    const int M = a.M, N = a.M / 2, T_ = a.seqlen;
    const int gpw = a.mfma_groups;
    const int ngroups = (a.NB + 4 * gpw - 1) / (4 * gpw);
    const int b = blockIdx.x / ngroups;
    const int s0 = (blockIdx.x % ngroups) * 4 * gpw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = KSplit<MK>::enabled ? KSplit<MK>::WM : (int)(blockDim.x >> 7);
    const int blk = lane >> 2, j = lane & 3;
    const int nphase = 2 * T_;
    for (int c = threadIdx.x; c < 8 * RS; c += blockDim.x) (&dbuf[0][0])[c] = 0.f;

    if (wave < wm) {
        if constexpr (KSP) {
            float wr[KS::CH][KS::KP];
            ksplit_load<MK, true>(a.W + (size_t)b * M * M, M, wave, lane, wr);
            using LdsF = const __attribute__((address_space(3))) float*;
            const unsigned db0 = (unsigned)(size_t)(LdsF)&dbuf[0][0];
            unsigned xoff[KS::CH];
#pragma unroll
            for (int c = 0; c < KS::CH; ++c) xoff[c] = (unsigned)((j * RS + (KS::slot(wave, c, blk) / KS::NQ) * KS::KP) * 4);
            __syncthreads();
            for (int p = 0; p < nphase; ++p) {
                if (p >= 1 && ((p - 1) & 1) < gpw) {
                    const int g = (p - 1) & 1;
                    const unsigned base = db0 + (unsigned)(4 * g * RS * 4);
                    static_assert(KS::CH == 2, "operand lists below are written out for two chains per wave");
                    unsigned xa[2] = {base + xoff[0], base + xoff[1]};
                    mf4* const out[2] = {&abuf[g][KS::slot(wave, 0, blk) * 4 + j], &abuf[g][KS::slot(wave, 1, blk) * 4 + j]};
                    ksplit_chain<MK>(wr, xa, out);
                }
                __syncthreads();
            }
        } else {
            float wr[MK];
            slab_load<MK, true>(a.W + (size_t)b * M * M, M, 64 * wave + lane, wr);
            __syncthreads();
            for (int p = 0; p < nphase; ++p) {
                if (p >= 1 && ((p - 1) & 1) < gpw) {
                    const int g = (p - 1) & 1;
                    abuf[g][wave * 64 + lane] = slab_chain<MK>(wr, &dbuf[4 * g + j][0]);
                }
                __syncthreads();
            }
        }
        return;
    }

    const int sw = wave - wm;
    const int er = 64 * sw + 4 * blk;
    const float inv = 1.f / (float)(T_ - a.skip);
    float eps[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) eps[v] = (er + v < N) ? a.eps_E : a.eps_I;
    // trajectory / delta streams of this draw through raw buffers (out-of-range offsets read 0 / are dropped)
    const size_t blk_elems = (size_t)a.NB * T_ * M;
    const __amdgpu_buffer_rsrc_t rs_traj =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.traj) + (size_t)b * blk_elems, 0, (int)(blk_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dlt =
        __builtin_amdgcn_make_buffer_rsrc(a.delta + (size_t)b * blk_elems, 0, (int)(blk_elems * 4), 0x00020000);
    const bool quad = er + 3 < M;                 // whole quad of rows inside the matrix (always when M % 4 == 0)
    bool live[2];
    int toff[2];
    float gta[2][4], carry[2][4], xn[2][4], xc[2][4], xm[2][4], dfc[2][4], dsum[2][4];
    float pxm[2][4], pdf[2][4];                   // loads in flight for the END of the group's next serial part
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
        // what serial(T) hands over at its end: x_{T-2} and f'(u_{T-1})
        if (T_ >= 3) load4(rs_traj, at_step(g, T_ - 3), pxm[g]); else { for (int v = 0; v < 4; ++v) pxm[g][v] = 0.f; }
        if (T_ >= 2) load4(rs_dlt, at_step(g, T_ - 2), pdf[g]); else { for (int v = 0; v < 4; ++v) pdf[g][v] = 0.f; }
    }
    // serial part of (group g, step tau): first = no matrix result yet (tau == T)
    // WIN: step inside the penalty window (tau >= skip + 1): only there the trajectory enters (direct gradient of the
    // three loss terms); before the window the serial part is the bare adjoint recurrence and x is neither read nor kept.
    auto serial = [&](auto G, auto WIN, int tau) {
        constexpr int g = decltype(G)::value;
        constexpr bool win_on = decltype(WIN)::value;
        // HBM reads run TWO serial parts ahead (a serial part is too short to cover their latency): issued here,
        // consumed at the end of serial(tau - 1); what this call consumes was issued by serial(tau + 1)
        float nxm[4] = {0.f, 0.f, 0.f, 0.f}, ndf[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (win_on) {
            // x_{tau-3} is used by the serial parts of steps tau-2 .. tau-4, and only inside the window
            if (tau >= 4 && tau >= a.skip + 3) load4(rs_traj, at_step(g, tau - 4), nxm);
        }
        if (tau >= 3) load4(rs_dlt, at_step(g, tau - 3), ndf);
        if (tau < T_) {
            mf4 acc;                                                                  // W^T delta_{tau+1}
            if constexpr (KSP) acc = ksplit_gather<MK>(&abuf[g][0], 16 * sw + blk, j);
            else acc = abuf[g][sw * 64 + lane];
            carry[g][0] += acc.x; carry[g][1] += acc.y; carry[g][2] += acc.z; carry[g][3] += acc.w;
        }
        float delta[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float gg = 0.f;                      // direct gradient of the loss w.r.t. x_tau
            if constexpr (win_on) {
                gg = gta[g][v] + ((xc[g][v] > a.theta) ? a.c_rate : 0.f);
                if (tau <= T_ - 1) gg -= 2.f * a.c_dyn * (xn[g][v] - xc[g][v]);
                if (tau >= a.skip + 2) gg += 2.f * a.c_dyn * (xc[g][v] - xm[g][v]);
            }
            const float at = gg + carry[g][v];
            delta[v] = (er + v < M) ? eps[v] * dfc[g][v] * at : 0.f;
            carry[g][v] = fmaf(-eps[v], at, at);                                       // (1 - eps) a_t
            dsum[g][v] += delta[v];
            if constexpr (win_on) {
                xn[g][v] = xc[g][v]; xc[g][v] = xm[g][v]; xm[g][v] = pxm[g][v]; pxm[g][v] = nxm[v];
            }
            dfc[g][v] = pdf[g][v];
            pdf[g][v] = ndf[v];
        }
        if (live[g] && er < M) *reinterpret_cast<mf4*>(&dbuf[4 * g + j][er]) = (mf4){delta[0], delta[1], delta[2], delta[3]};
        if (tau >= 2) store4(rs_dlt, at_step(g, tau - 2), delta);                      // shifted: pairs with x_{tau-1}
    };
    constexpr std::integral_constant<int, 0> G0{};
    constexpr std::integral_constant<int, 1> G1{};
    constexpr std::integral_constant<bool, false> W0{};
    constexpr std::integral_constant<bool, true> W1{};
    __syncthreads();
    int tau = T_;
    for (; tau >= a.skip + 1 && tau >= 1; --tau) {    // window steps first (time runs backwards)
        serial(G0, W1, tau);                          // phase 2 (T - tau)
        __syncthreads();
        if (gpw == 2) serial(G1, W1, tau);                          // phase 2 (T - tau) + 1
        __syncthreads();
    }
    for (; tau >= 1; --tau) {
        serial(G0, W0, tau);
        __syncthreads();
        if (gpw == 2) serial(G1, W0, tau);
        __syncthreads();
    }
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

// Fixed-point solver (Euler loop of tc_gan/ext/ssnode.c:69-187, contract of solve_tile_kernel) in the specialised MFMA
// form, for NB >= 4 stimuli per draw.  Serial part of (group g, step it): r1 = r + (-r + f(W r + ext)) eps, the two
// stop tests, freeze per stimulus.  Stop protocol: the flags of (g, it) (one word per stimulus: low half "some row not
// converged", high half "some row hit the rate bound"; three rotating slots) are complete after the barrier of the
// phase in which serial(g, it) ran; in the NEXT phase EVERY wave reads them and updates the same frozen mask / codes /
// step counts, so matrix and serial waves leave the loop in the same phase.
template <int MK>
__global__ void __launch_bounds__(KSplit<MK>::enabled ? KSplit<MK>::THREADS : 512, KSplit<MK>::enabled ? 3 : 2) solve_mfma_kernel(SolveArgs<float> a) {
    using KS = KSplit<MK>;
    constexpr bool KSP = KS::enabled;
    constexpr int RS = KSP ? KS::RS : MK + 4;
    __shared__ __align__(16) float rbuf[2][8][RS];
    __shared__ __align__(16) mf4 abuf[2][KSP ? KS::SLOTS * 4 : 4 * 64];    // hand-off, as in the generator kernels
    __shared__ __align__(16) int flags[3][8];
    const int M = a.M, N = a.N, max_iter = a.st.max_iter;
    const int ngroups = (a.NB + 7) / 8;
    const int b = blockIdx.x / ngroups;
    const int s0 = (blockIdx.x % ngroups) * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = KSplit<MK>::enabled ? KSplit<MK>::WM : (int)(blockDim.x >> 7);
    const int blk = lane >> 2, j = lane & 3;
    const bool matrix = wave < wm;
    const int sw = wave - wm;
    const int er = 64 * (matrix ? wave : sw) + 4 * blk;

    for (int c = threadIdx.x; c < 2 * 8 * RS; c += blockDim.x) (&rbuf[0][0][0])[c] = 0.f;
    if (threadIdx.x < 24) (&flags[0][0])[threadIdx.x] = 0;
    __syncthreads();

    // ---- bookkeeping, identical in every wave: lane s (< 8) follows stimulus s0 + s --------------------
    // (per-lane VALU + one ballot per phase; a scalar version costs ~40 dependent SALU instructions per stimulus group
    // and phase, in order, in front of the matrix wave's next MFMA chain)
    int my_code = 1, my_steps = max_iter;
    bool my_frozen = lane >= 8 || s0 + lane >= a.NB;             // lanes without a stimulus count as stopped
    unsigned frozen = (unsigned)__builtin_amdgcn_ballot_w64(my_frozen) & 0xffu;   // bit s: stimulus s0 + s has stopped
    // The flag word is READ at the start of a phase and APPLIED at its end (after the phase's work, before the
    // barrier): the LDS round trip would otherwise sit in front of the MFMA chain / serial part of every phase.
    auto flags_read = [&](auto, int it) { return flags[it % 3][lane & 7]; };
    auto verdict = [&](auto G, int it, int f) {                  // flags of (group G, step it) -> my_* / frozen
        constexpr int g = decltype(G)::value;
        const bool fnc = (f & 0xffff) != 0, fhb = (f >> 16) != 0;
        const bool stop = lane < 8 && (lane >> 2) == g && !my_frozen && (!fnc || fhb);
        my_code = stop ? (fnc ? 2 : 0) : my_code;
        my_steps = stop ? it + 1 : my_steps;
        my_frozen = my_frozen || stop;
        frozen = (unsigned)__builtin_amdgcn_ballot_w64(my_frozen) & 0xffu;
    };
    constexpr std::integral_constant<int, 0> G0{};
    constexpr std::integral_constant<int, 1> G1{};

    // The two roles run the SAME phase structure (same barriers, same exits) in separate loops, so that the
    // registers of one role (200 for the slab) are not live in the other.
    if (matrix) {
        if constexpr (KSP) {
            float wr[KS::CH][KS::KP];
            ksplit_load<MK, false>(a.W + (size_t)b * M * M, M, wave, lane, wr);
            using LdsF = const __attribute__((address_space(3))) float*;
            const unsigned rb0 = (unsigned)(size_t)(LdsF)&rbuf[0][0][0];
            unsigned xoff[KS::CH];
#pragma unroll
            for (int c = 0; c < KS::CH; ++c) xoff[c] = (unsigned)((j * RS + (KS::slot(wave, c, blk) / KS::NQ) * KS::KP) * 4);
            auto chain = [&](int g, int it) {
                const unsigned base = rb0 + (unsigned)((((it & 1) * 8 + 4 * g) * RS) * 4);
                static_assert(KS::CH == 2, "operand lists below are written out for two chains per wave");
                unsigned xa[2] = {base + xoff[0], base + xoff[1]};
                mf4* const out[2] = {&abuf[g][KS::slot(wave, 0, blk) * 4 + j], &abuf[g][KS::slot(wave, 1, blk) * 4 + j]};
                ksplit_chain<MK>(wr, xa, out);
            };
            __syncthreads();
            for (int it = 0; it <= max_iter; ++it) {
                int f4 = 0;
                if (it >= 1) f4 = flags_read(G0, it - 1);                       // phase 2 it
                if (it < max_iter) chain(0, it);
                if (it >= 1) verdict(G0, it - 1, f4);
                if (frozen == 0xffu) break;
                __syncthreads();
                if (it >= 1) f4 = flags_read(G1, it - 1);                       // phase 2 it + 1
                if (it < max_iter) chain(1, it);
                if (it >= 1) verdict(G1, it - 1, f4);
                if (frozen == 0xffu) break;
                __syncthreads();
            }
        } else {
            float wr[MK];
            slab_load<MK, false>(a.W + (size_t)b * M * M, M, 64 * wave + lane, wr);
            __syncthreads();
            for (int it = 0; it <= max_iter; ++it) {
                int f4 = 0;
                if (it >= 1) f4 = flags_read(G0, it - 1);                       // phase 2 it
                if (it < max_iter) abuf[0][wave * 64 + lane] = slab_chain<MK>(wr, &rbuf[it & 1][j][0]);
                if (it >= 1) verdict(G0, it - 1, f4);
                if (frozen == 0xffu) break;
                __syncthreads();
                if (it >= 1) f4 = flags_read(G1, it - 1);                       // phase 2 it + 1
                if (it < max_iter) abuf[1][wave * 64 + lane] = slab_chain<MK>(wr, &rbuf[it & 1][4 + j][0]);
                if (it >= 1) verdict(G1, it - 1, f4);
                if (frozen == 0xffu) break;
                __syncthreads();
            }
        }
        return;
    }

    float eps[4], rc[2][4], rp[2][4], ex[2][4];
    bool live[2];
#pragma unroll
    for (int v = 0; v < 4; ++v) eps[v] = (er + v < N) ? a.st.eps_E : a.st.eps_I;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int s = s0 + 4 * g + j;
        live[g] = s < a.NB;
        const size_t vec = ((size_t)b * a.NB + (live[g] ? s : 0)) * M;
        float r4[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bool ok = live[g] && er + v < M;
            rc[g][v] = rp[g][v] = r4[v] = ok ? a.r[vec + er + v] : 0.f;
            ex[g][v] = ok ? a.ext[(a.ext_per_draw ? vec : (size_t)s * M) + er + v] : 0.f;
        }
        if (live[g] && er < M) *reinterpret_cast<mf4*>(&rbuf[0][4 * g + j][er]) = (mf4){r4[0], r4[1], r4[2], r4[3]};
    }
    const IoSelect io(a.io);
    bool rowok[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) rowok[v] = er + v < M;
    auto serial = [&](auto G, int it) {            // serial part of (group G, step it)
        constexpr int g = decltype(G)::value;
        mf4 acc;
        if constexpr (KSP) acc = ksplit_gather<MK>(&abuf[g][0], 16 * sw + blk, j);
        else acc = abuf[g][sw * 64 + lane];
        const float accs[4] = {acc.x, acc.y, acc.z, acc.w};
        float uu[4], ff[4], dummy[4], r1[4], dabs[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) uu[v] = accs[v] + ex[g][v];
        io.template eval4<false>(uu, ff, dummy);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            r1[v] = rc[g][v] + (-rc[g][v] + ff[v]) * eps[v];                   // ssnode.c:64-67
            dabs[v] = rowok[v] ? fabsf(r1[v] - rc[g][v]) : -1.f;               // rows beyond M never "not converged"
        }
        if (sw == 0 && lane < 4) flags[(it + 1) % 3][4 * g + lane] = 0;
        // one lane-divergent region for everything a stopped (or absent) stimulus must not do
        if (live[g] && !((frozen >> (4 * g + j)) & 1u) && er < M) {
            const float dmax = fmaxf(fmaxf(dabs[0], dabs[1]), fmaxf(dabs[2], dabs[3]));
            const float rmax = fmaxf(fmaxf(r1[0], r1[1]), fmaxf(r1[2], r1[3]));   // rows beyond M hold 0
            short* fw = reinterpret_cast<short*>(&flags[it % 3][4 * g + j]);
            if (dmax >= a.st.atol) fw[0] = 1;
            if (a.st.check_hard && rmax >= a.st.hard_stop) fw[1] = 1;
#pragma unroll
            for (int v = 0; v < 4; ++v) { rp[g][v] = rc[g][v]; rc[g][v] = rowok[v] ? r1[v] : rc[g][v]; }
            *reinterpret_cast<mf4*>(&rbuf[(it + 1) & 1][4 * g + j][er]) = (mf4){rc[g][0], rc[g][1], rc[g][2], rc[g][3]};
        }
    };
    __syncthreads();
    for (int it = 0; it <= max_iter; ++it) {
        int f4 = 0;
        if (it >= 1) f4 = flags_read(G0, it - 1);                               // phase 2 it: serial part of (1, it - 1)
        if (it >= 1) serial(G1, it - 1);
        if (it >= 1) verdict(G0, it - 1, f4);
        if (frozen == 0xffu) break;
        __syncthreads();
        if (it >= 1) f4 = flags_read(G1, it - 1);                               // phase 2 it + 1: serial part of (0, it)
        if (it < max_iter) serial(G0, it);
        if (it >= 1) verdict(G1, it - 1, f4);
        if (frozen == 0xffu) break;
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (!live[g]) continue;
        const int s = 4 * g + j;
        const size_t unit = (size_t)b * a.NB + s0 + s;
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

static int mfma_pick_mk(int M) {
    const int ladder[] = {104, 152, 200, 208};
    for (int mk : ladder) if (M <= mk) return mk;
    return 0;
}
bool gen_mfma_supported(int M, int NB) { return (M % 2 == 0) && NB >= 4 && mfma_pick_mk(M) != 0; }

template <int MK>
static hipError_t launch_fwd_mk(const GenFwdArgs<float>& a, hipStream_t st) {
    const int threads = KSplit<MK>::enabled ? KSplit<MK>::THREADS : 128 * ((a.M + 63) / 64);
    const int ngroups = (a.NB + 4 * a.mfma_groups - 1) / (4 * a.mfma_groups);
    if (a.traj) hipLaunchKernelGGL((gen_forward_mfma_kernel<MK, true>), dim3(a.B * ngroups), dim3(threads), 0, st, a);
    else hipLaunchKernelGGL((gen_forward_mfma_kernel<MK, false>), dim3(a.B * ngroups), dim3(threads), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_gen_forward_mfma(const GenFwdArgs<float>& a, hipStream_t st) {
    switch (mfma_pick_mk(a.M)) {
        case 104: return launch_fwd_mk<104>(a, st);
        case 152: return launch_fwd_mk<152>(a, st);
        case 200: return launch_fwd_mk<200>(a, st);
        case 208: return launch_fwd_mk<208>(a, st);
        default: return hipErrorInvalidValue;
    }
}

template <int MK>
static hipError_t launch_bwd_mk(const GenBwdArgs<float>& a, hipStream_t st) {
    const int threads = KSplit<MK>::enabled ? KSplit<MK>::THREADS : 128 * ((a.M + 63) / 64);
    const int ngroups = (a.NB + 4 * a.mfma_groups - 1) / (4 * a.mfma_groups);
    hipLaunchKernelGGL((gen_backward_mfma_kernel<MK>), dim3(a.B * ngroups), dim3(threads), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_gen_backward_mfma(const GenBwdArgs<float>& a, hipStream_t st) {
    switch (mfma_pick_mk(a.M)) {
        case 104: return launch_bwd_mk<104>(a, st);
        case 152: return launch_bwd_mk<152>(a, st);
        case 200: return launch_bwd_mk<200>(a, st);
        case 208: return launch_bwd_mk<208>(a, st);
        default: return hipErrorInvalidValue;
    }
}

template <int MK>
static hipError_t launch_solve_mk(const SolveArgs<float>& a, hipStream_t st) {
    const int threads = KSplit<MK>::enabled ? KSplit<MK>::THREADS : 128 * ((a.M + 63) / 64);
    const int ngroups = (a.NB + 7) / 8;
    hipLaunchKernelGGL((solve_mfma_kernel<MK>), dim3(a.B * ngroups), dim3(threads), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_solve_mfma(const SolveArgs<float>& a, hipStream_t st) {
    switch (mfma_pick_mk(a.M)) {
        case 104: return launch_solve_mk<104>(a, st);
        case 152: return launch_solve_mk<152>(a, st);
        case 200: return launch_solve_mk<200>(a, st);
        case 208: return launch_solve_mk<208>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ssn
