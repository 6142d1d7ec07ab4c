// "Tile" register-stationary SSN solver (variant 2) for MI355X (gfx950).
//
// Same contract as solve_regw_kernel (ssn_solver.hip) -- the Euler loop of
// tc_gan/ext/ssnode.c:69-187 over B draws x NB stimuli -- with a different
// mapping of W onto the register file, chosen from measurements on the chip
// (tools/microbench): a v_fmac_f32 whose r operand is DPP-broadcast issues at
// ~2/3 of the rate of a plain VGPR-operand v_fmac_f32, and with one matrix row
// per lane a 200-neuron solve keeps only 200 of 256 lanes busy.
//
// Layout: a wave is an 8 x 8 grid of lanes, lane = 8*rg + cg.
//   * row group rg owns RA consecutive rows:  row = (8*wave + rg)*RA + a
//   * column group cg owns C consecutive columns:  col = cg*C + c
//   * each lane holds the RA x C tile W[rows(rg)][cols(cg)] in VGPRs for the
//     whole solve (W is read from HBM once);
//   * per Euler step the lane reads ITS C values of r from LDS with 16-byte
//     reads (the 8 lanes of a column group read the same address -> broadcast;
//     slabs are padded so the 8 column groups hit disjoint banks) and issues
//     RA*C plain FMAs: every loaded r value feeds RA FMAs;
//   * the 8 partial sums of a row sit in 8 ADJACENT lanes; a transpose-reduce
//     (reduce8_to_lane, ssn_tile_core.h: 7 DPP adds + selects, the first stage as
//     masked v_add_f32_dpp) leaves lane cg with the complete sum of row a = cg;
//   * lane cg < RA then finishes row a = cg: + ext, I/O nonlinearity, Euler
//     update, stop tests, and writes r' back to LDS.  ONE barrier per step.
// For 2N = 200: C = 25, RA = 7, 4 waves (224 row slots, 200 used; all 200
// columns exact).  Default shape: split residency (RL = 2: five rows of every
// lane's tile in VGPRs as packed pairs, two in LDS), 167 VGPRs, three workgroups
// per CU; the all-register shape (175 W registers per lane, two workgroups per
// CU) is variant 4.  fp64: 7 rows x C <= 13 up to 2N = 104, 4 rows x C = 19 / 26
// (one workgroup of 5-7 waves per CU) up to 2N = 208.
#include <hip/hip_runtime.h>
#include "ssn_device.h"
#include "ssn_host.h"
#include "ssn_tile_core.h"

// diagnostic switch: 0 = the uniform 7-row kernel also at 2N = 170..200 (A/B builds)
#ifndef SSN_TILE_MIXED
#define SSN_TILE_MIXED 1
#endif

namespace ssn {

// The Euler loop of one wave: its RA x C tile of W (rows rowbase + a of row group rg), the step loop with ONE workgroup
// barrier per step, the stop protocol and the final stores.  Every wave of a workgroup runs this with the same C, NB
// and barrier sequence; RA / RL / PK (row pairs packed for v_pk_fma_f32) may differ between waves (mixed kernel below).
template <typename T, int RA, int C, int RL, int NB, bool PK>
__device__ __forceinline__ void tile_wave_body(const SolveArgs<T>& a, T* wlds, T (*rbuf)[NB][8 * SlabPad<C>::value],
                                               int (*flags)[NB], const int rowbase) {
    constexpr int CP = SlabPad<C>::value;
    constexpr int NQ = (C + 3) / 4;              // 16-B reads per lane per stimulus
    static_assert(RA <= 8, "a row group has 8 lanes to finish its rows");
    using V4 = T __attribute__((ext_vector_type(4)));
    const int M = a.M, N = a.N;
    const int ngroups = (a.NB + NB - 1) / NB;
    const int b = blockIdx.x / ngroups;
    const int s0 = (blockIdx.x % ngroups) * NB;
    const int lane = threadIdx.x & 63;
    const int cg = lane & 7;
    const int colbase = cg * C;

    // ---- prologue: my RA x C tile of W -> registers (and LDS for the last RL rows) -----------------
    constexpr bool SPLIT = RL > 0 || PK;        // SplitTile: packed row pairs in VGPRs (+ RL rows in LDS)
    T w[!SPLIT ? RA : 1][!SPLIT ? C : 1];       // plain all-register shape
    SplitTile<T, RA, C, RL> sw;
    if constexpr (SPLIT) sw.template load<false>(a.W + (size_t)b * M * M, M, rowbase, colbase, wlds, threadIdx.x);
    else tile_load<T, RA, C, false>(a.W + (size_t)b * M * M, M, rowbase, colbase, w);
    // the row this lane finishes each step (a = cg), its LDS slot, input and state
    const int myrow = rowbase + cg;
    const bool fin = (cg < RA) && (myrow < M);
    const int myslot = (myrow / C) * CP + (myrow % C);
    T rc[NB], rp[NB], ex[NB];            // x_k, x_{k-1} of my row; input of my row
    bool live[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        live[s] = (s0 + s) < a.NB;
        const size_t vec = ((size_t)b * a.NB + (live[s] ? s0 + s : 0)) * M;
        rc[s] = (fin && live[s]) ? a.r[vec + myrow] : (T)0;
        rp[s] = rc[s];
        ex[s] = (fin && live[s]) ? a.ext[(a.ext_per_draw ? vec : (size_t)(s0 + s) * M) + myrow] : (T)0;
    }
    for (int c = threadIdx.x; c < 2 * NB * 8 * CP; c += blockDim.x) (&rbuf[0][0][0])[c] = (T)0;
    if (threadIdx.x < 3 * NB) (&flags[0][0])[threadIdx.x] = 0;
    __syncthreads();
    if (fin) {
#pragma unroll
        for (int s = 0; s < NB; ++s) rbuf[0][s][myslot] = rc[s];
    }
    __syncthreads();

    const T eps = (myrow < N) ? a.st.eps_E : a.st.eps_I;
    int code[NB], nsteps[NB];
    bool frozen[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) { code[s] = 1; nsteps[s] = a.st.max_iter; frozen[s] = !live[s]; }

    // Stop protocol (no vote, no second barrier, no exposed LDS latency): in iteration i every lane
    // whose row fails the convergence test (or hits the rate bound) stores 1 to flags[i%3]; the
    // verdict of iteration i is READ in iteration i+1 (the loads are issued before the FMA block and
    // consumed after it, BEFORE that iteration's state update, so rc/rp still hold x_{i+1}/x_i) and
    // slot (i+1)%3 is cleared for iteration i+1.  The FMAs of the extra iteration are discarded.
    int cur = 0;
    int step = 0;
    for (; step < a.st.max_iter; ++step) {
        int fl[NB];
        if (!(SSN_ABLATE & 4) && step > 0) {
#pragma unroll
            for (int s = 0; s < NB; ++s) fl[s] = flags[(step + 2) % 3][s];
        }
        // ---- partial sums over my C columns for my RA rows ---------------------------
        T acc[NB][8];
#pragma unroll
        for (int s = 0; s < NB; ++s)
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[s][r] = (T)0;
        if constexpr (SPLIT) {
            sw.template matvec<NB>(wlds, threadIdx.x, &rbuf[cur][0][0], cg, acc);
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                V4 rv[NB];
#pragma unroll
                for (int s = 0; s < NB; ++s) {
                    if constexpr (SSN_ABLATE & 8) { rv[s].x = rc[s]; rv[s].y = rc[s] + 1; rv[s].z = rc[s] + 2; rv[s].w = rc[s] + 3; }
                    else rv[s] = *reinterpret_cast<const V4*>(&rbuf[cur][s][cg * CP + 4 * q]);
                }
#pragma unroll
                for (int s = 0; s < NB; ++s) {
                    const T rr[4] = {rv[s].x, rv[s].y, rv[s].z, rv[s].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (4 * q + e < C) {
#pragma unroll
                            for (int r = 0; r < RA; ++r) acc[s][r] = fma(w[r][4 * q + e], rr[e], acc[s][r]);
                        }
                    }
                }
            }
        }
        // ---- verdict of the previous iteration (uniform over the workgroup) -----------
        if (!(SSN_ABLATE & 4) && step > 0) {
            // keep the consumption of the flag loads BELOW the FMA block: without this tie the compiler
            // hoists the branch to the loop head and waits for the flag read before issuing the r reads
            // (two exposed LDS latencies per step instead of one)
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                asm volatile("" : "+v"(fl[s]));
#pragma unroll
                for (int r = 0; r < RA; ++r) asm volatile("" : "+v"(acc[s][r]));
            }
            int nfrozen = 0;
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                // every lane read the same word: make the verdict (and with it frozen/code/nsteps/cur and the loop
                // control) scalar
                const int fu = __builtin_amdgcn_readfirstlane(fl[s]);
                const bool fnc = fu & 0xffff, fhb = fu >> 16;
                if (!frozen[s] && (!fnc || fhb)) {
                    code[s] = fnc ? 2 : 0;
                    nsteps[s] = step;                 // it stopped in iteration step-1
                    frozen[s] = true;
                }
                nfrozen += frozen[s];
            }
            if (nfrozen == NB) break;
        }
        // ---- combine the 8 column groups; lane cg ends with the total of row a = cg ----
        const int slot = step % 3;
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const T u = (SSN_ABLATE & 2) ? acc[s][0] : reduce_rows_to_lane<RA>(acc[s], cg);
            // ---- Euler update + stop tests for my row ---------------------------------
            const T fu = (SSN_ABLATE & 1) ? (u + ex[s]) : io_eval(u + ex[s], a.io);
            const T r1 = rc[s] + (-rc[s] + fu) * eps;
            if (fin && !frozen[s]) {
                if (!(SSN_ABLATE & 4)) {
                    short* fw = reinterpret_cast<short*>(&flags[slot][s]);
                    if (abs_t(r1 - rc[s]) >= a.st.atol) fw[0] = 1;
                    if (a.st.check_hard && r1 >= a.st.hard_stop) fw[1] = 1;
                }
                if constexpr (NB > 1) rp[s] = rc[s];     // NB == 1: x_{k-1} is read back from LDS at the end
                rc[s] = r1;
            }
            if (fin) rbuf[cur ^ 1][s][myslot] = rc[s];
        }
        if (!(SSN_ABLATE & 4) && threadIdx.x < NB) flags[(step + 1) % 3][threadIdx.x] = 0;
        if constexpr (!(SSN_ABLATE & 16)) __syncthreads();
        cur ^= 1;
    }
    // NB == 1: nothing runs after the stop, so the buffer that was read in the last update still holds x_{k-1}
    if constexpr (NB == 1) { if (step > 0 && fin) rp[0] = rbuf[cur ^ 1][0][myslot]; }
    // verdict of the last executed iteration (no extra step was taken, so no roll-back)
    if (!(SSN_ABLATE & 4) && step == a.st.max_iter && step > 0) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            if (!frozen[s]) {
                const int f = flags[(step + 2) % 3][s];
                const int nc = f & 0xffff, hb = f >> 16;
                if (!nc) { code[s] = 0; nsteps[s] = step; }
                else if (hb) { code[s] = 2; nsteps[s] = step; }
            }
        }
    }

#pragma unroll
    for (int s = 0; s < NB; ++s) {
        if (!live[s]) continue;
        const size_t vec = ((size_t)b * a.NB + s0 + s) * M;
        if (fin) {
            a.r[vec + myrow] = rc[s];
            if (a.r_prev) a.r_prev[vec + myrow] = rp[s];
        }
        if (threadIdx.x == 0) {
            a.codes[(size_t)b * a.NB + s0 + s] = code[s];
            if (a.steps) a.steps[(size_t)b * a.NB + s0 + s] = nsteps[s];
        }
    }
}

template <typename T, int RA, int C, int RL, int NB, int MAXTHREADS, int MINWAVES>
__global__ void __launch_bounds__(MAXTHREADS, MINWAVES) solve_tile_kernel(SolveArgs<T> a) {
    constexpr int CP = SlabPad<C>::value;
    using Split = TileSplit<RA, C, RL>;       // RL > 0: last RL rows of every lane's tile live in LDS
    __shared__ __align__(16) T wlds[Split::lds_elems(MAXTHREADS)];
    __shared__ __align__(16) T rbuf[2][NB][8 * CP];
    // per slot and stimulus one 32-bit word: low half = "some row has not converged", high half = "some row hit
    // the rate bound"; written with 16-bit stores (no race between the two kinds), read and cleared as one word
    __shared__ int flags[3][NB];
    if constexpr (RL > 0) set_rank_priority((blockIdx.x >> 8) % 3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    tile_wave_body<T, RA, C, RL, NB, false>(a, wlds, rbuf, flags, (8 * wave + (lane >> 3)) * RA);
}

// Split shapes whose LAST wave is left with at most 40 rows (2N = 170..200 at C = 25: waves 0-2 own 56 rows each, 32 remain;
// 2N = 114..152 at C = 19: 40 remain): that wave runs an LRA-row tile (LRA = 4 or 5), all in VGPRs as packed row pairs --
// 50 (or 57) FMA instructions per step instead of 100 (76) on a 7-row tile that is mostly padding, and no LDS-resident
// rows to re-read.  Same barriers, same stop protocol, same results.
template <typename T, int C, int RL, int NW, int LRA, int NB, int MINWAVES>
__global__ void __launch_bounds__(64 * NW, MINWAVES) solve_tile_mixed_kernel(SolveArgs<T> a) {
    constexpr int CP = SlabPad<C>::value;
    using Split = TileSplit<7, C, RL>;
    __shared__ __align__(16) T wlds[Split::lds_elems(64 * (NW - 1))];      // heavy waves only
    __shared__ __align__(16) T rbuf[2][NB][8 * CP];
    __shared__ int flags[3][NB];
    set_rank_priority((blockIdx.x >> 8) % 3);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < NW - 1) tile_wave_body<T, 7, C, RL, NB, false>(a, wlds, rbuf, flags, (8 * wave + (lane >> 3)) * 7);
    else tile_wave_body<T, LRA, C, 0, NB, true>(a, wlds, rbuf, flags, 56 * (NW - 1) + (lane >> 3) * LRA);
}

// ---------------------------------------------------------------------------------
// dispatch: C = columns per column group (8*C >= M), RA = 7 rows per lane, waves = ceil(M / 56) <= 4.
// Shapes (fp32): "split" = RL rows of every lane's tile in LDS so that the kernel fits 168 VGPRs and
// three workgroups share a CU (C = 19: RL = 1, C = 25: RL = 2; C = 26 would need 3 x 55 KB of LDS);
// "reg" = whole tile in VGPRs (2 waves/SIMD at C >= 19).  fp64: all-register only.
// ---------------------------------------------------------------------------------
static constexpr int TILE_RA = 7;

static int tile_pick_c(int M, int elem_bytes) {
    const int need = (M + 7) / 8;
    const int ladder32[] = {4, 8, 13, 19, 25, 26};
    const int ladder64[] = {4, 8, 13, 19, 26};         // fp64: 2 VGPRs per value; C >= 19 runs with 4 rows per lane
    if (elem_bytes == 4) { for (int c : ladder32) if (need <= c) return c; }
    else                 { for (int c : ladder64) if (need <= c) return c; }
    return 0;
}

template <typename T>
bool tile_supported(int M, int NB) { (void)NB; return (M % 2 == 0) && tile_pick_c(M, (int)sizeof(T)) != 0; }
template bool tile_supported<float>(int, int);
template bool tile_supported<double>(int, int);

template <typename T, int RA, int C, int RL, int NB, int MINW>
static hipError_t launch_tile_k(const SolveArgs<T>& a, hipStream_t st) {
    constexpr int MAXW = (8 * C + 8 * RA - 1) / (8 * RA);
    const int waves = (a.M + 8 * RA - 1) / (8 * RA);
    const int ngroups = (a.NB + NB - 1) / NB;
    hipLaunchKernelGGL((solve_tile_kernel<T, RA, C, RL, NB, 64 * MAXW, MINW>), dim3(a.B * ngroups),
                       dim3(64 * waves), 0, st, a);
    return hipGetLastError();
}

template <typename T, int C>
static hipError_t launch_tile_nb(const SolveArgs<T>& a, hipStream_t st, bool split) {
    if constexpr (sizeof(T) == 4) {
        // split shapes; when the last wave is left with <= 40 rows it gets a lighter tile (mixed kernel)
        const int waves = (a.M + 55) / 56, last_rows = a.M - 56 * (waves - 1);
        if constexpr (C == 25) {                                                      // (C = 26 spills in the loop)
            if (SSN_TILE_MIXED && split && waves == 4 && last_rows <= 32) {
                hipLaunchKernelGGL((solve_tile_mixed_kernel<T, C, 2, 4, 4, 1, 3>), dim3(a.B * a.NB), dim3(256), 0, st, a);
                return hipGetLastError();
            }
            if (split) return launch_tile_k<T, TILE_RA, C, 2, 1, 3>(a, st);
        }
        if constexpr (C == 19) {
            if (SSN_TILE_MIXED && split && waves == 3 && last_rows <= 40) {
                if (last_rows <= 32) hipLaunchKernelGGL((solve_tile_mixed_kernel<T, C, 1, 3, 4, 1, 3>), dim3(a.B * a.NB), dim3(192), 0, st, a);
                else hipLaunchKernelGGL((solve_tile_mixed_kernel<T, C, 1, 3, 5, 1, 3>), dim3(a.B * a.NB), dim3(192), 0, st, a);
                return hipGetLastError();
            }
            if (split) return launch_tile_k<T, TILE_RA, C, 1, 1, 3>(a, st);
        }
        // 7*C W registers + 8*NB accumulators + 4*NB states must stay under 256 VGPRs (no spills)
        if constexpr (C <= 19) { if (a.NB >= 4) return launch_tile_k<T, TILE_RA, C, 0, 4, 2>(a, st); }
        if (a.NB >= 2) return launch_tile_k<T, TILE_RA, C, 0, 2, 2>(a, st);
        return launch_tile_k<T, TILE_RA, C, 0, 1, 2>(a, st);
    } else {
        // fp64 (2 VGPRs per value).  Up to 2N = 104: 7 rows x C <= 13 columns per lane, whole tile in VGPRs.  Beyond
        // (the reference's default N = 102 gives 2N = 204, tc_gan/ssnode.py:28): 4 rows per lane and ceil(2N / 32)
        // waves -- one workgroup of up to 7 waves owns the CU, W (333 KB at 2N = 204) sits in its VGPRs and, at C = 26,
        // one row of every lane's tile in LDS (94 KB) so that the kernel stays under 256 VGPRs (two waves per SIMD).
        if constexpr (C == 26) return launch_tile_k<T, 4, C, 1, 1, 2>(a, st);
        else if constexpr (C == 19) return launch_tile_k<T, 4, C, 0, 1, 2>(a, st);
        else return launch_tile_k<T, TILE_RA, C, 0, 1, 1>(a, st);
    }
}

// shape: 0 = library default, 1 = split residency where instantiated, 2 = all-register
template <typename T> hipError_t launch_tile(const SolveArgs<T>& a, hipStream_t st, int shape);
template <> hipError_t launch_tile<float>(const SolveArgs<float>& a, hipStream_t st, int shape) {
    const bool split = shape != 2;
    switch (tile_pick_c(a.M, 4)) {
        case 4: return launch_tile_nb<float, 4>(a, st, split);
        case 8: return launch_tile_nb<float, 8>(a, st, split);
        case 13: return launch_tile_nb<float, 13>(a, st, split);
        case 19: return launch_tile_nb<float, 19>(a, st, split);
        case 25: return launch_tile_nb<float, 25>(a, st, split);
        case 26: return launch_tile_nb<float, 26>(a, st, split);
        default: return hipErrorInvalidValue;
    }
}
template <> hipError_t launch_tile<double>(const SolveArgs<double>& a, hipStream_t st, int shape) {
    (void)shape;
    switch (tile_pick_c(a.M, 8)) {
        case 4: return launch_tile_nb<double, 4>(a, st, false);
        case 8: return launch_tile_nb<double, 8>(a, st, false);
        case 13: return launch_tile_nb<double, 13>(a, st, false);
        case 19: return launch_tile_nb<double, 19>(a, st, false);
        case 26: return launch_tile_nb<double, 26>(a, st, false);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ssn
