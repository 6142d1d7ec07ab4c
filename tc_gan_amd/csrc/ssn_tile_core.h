// Shared device pieces of the "tile" register-stationary layout (see ssn_tile.hip):
// 8x8 lane grid per wave, RA x C tile of the matrix per lane, in-wave transpose-reduce.
#pragma once
#include <hip/hip_runtime.h>
#include "ssn_device.h"

// Diagnostic builds only (tools/microbench/tile_ablate.hip): bit mask of loop phases to stub out so that
// their cost can be measured.  The product library is always built with SSN_ABLATE == 0.
#ifndef SSN_ABLATE
#define SSN_ABLATE 0
#endif
// diagnostic switch: 0 no wave priorities, 2 static priority per workgroup rank (product setting)
#ifndef SSN_PRIO_MODE
#define SSN_PRIO_MODE 2
#endif
// diagnostic switch: 0 = leave the packed FMAs of the split tile to the compiler (A/B builds), 1 = asm with op_sel broadcast
#ifndef SSN_PK_ASM
#define SSN_PK_ASM 1
#endif
#ifndef SSN_REDUCE4
#define SSN_REDUCE4 1            // diagnostic: 0 = 8-row transpose-reduce also for tiles of <= 4 rows
#endif

namespace ssn {

// x + (x from the lane selected by a DPP control); folds to v_add_f32_dpp.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    const float y = __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
    return x + y;
}
template <int CTRL>
__device__ __forceinline__ double dpp_add(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return x + __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// Sum over the 8 adjacent lanes {8k .. 8k+7}; every lane ends with the total.
template <typename T>
__device__ __forceinline__ T sum8(T x) {
    x = dpp_add<0xB1>(x);    // quad_perm:[1,0,3,2]
    x = dpp_add<0x4E>(x);    // quad_perm:[2,3,0,1]
    x = dpp_add<0x141>(x);   // row_half_mirror
    return x;
}

template <int CTRL, typename T>
__device__ __forceinline__ T dpp_get(T x);
template <int CTRL>
__device__ __forceinline__ float dpp_get_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_get_f(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Transpose-reduce over the 8 adjacent lanes of a row group: in: 8 per-lane partial sums
// acc[0..7] (row a of the group, this lane's columns); out: lane cg holds the TOTAL of row cg.
// Each stage halves the rows a lane still carries (keep the half selected by one bit of cg, send
// the other half to the partner that keeps it): 4+2+1 = 7 DPP adds and 14 selects, instead of
// 3 DPP adds per row for all 8 rows.
template <typename T>
__device__ __forceinline__ T reduce8_to_lane(const T (&acc)[8], int cg) {
    const bool bA = cg & 4, bB = cg & 2, bC = cg & 1;
    T n4[4], n2[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const T keep = bA ? acc[k + 4] : acc[k];
        const T send = bA ? acc[k] : acc[k + 4];
        n4[k] = keep + dpp_get_f<0x141>(send);          // row_half_mirror: lane i <-> 7-i
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const T keep = bB ? n4[k + 2] : n4[k];
        const T send = bB ? n4[k] : n4[k + 2];
        n2[k] = keep + dpp_get_f<0x4E>(send);           // quad_perm:[2,3,0,1]: lane i <-> i^2
    }
    const T keep = bC ? n2[1] : n2[0];
    const T send = bC ? n2[0] : n2[1];
    return keep + dpp_get_f<0xB1>(send);                // quad_perm:[1,0,3,2]: lane i <-> i^1
}

// fp32: the first exchange without selects.  Bit 2 of the lane index (= bit 2 of cg) is a DPP bank boundary, so the
// two halves are formed by two v_add_f32_dpp, the second one writing only the lanes that keep rows 4..7
// (bank_mask 0xa = lanes 4-7 and 12-15 of every row of 16): 8 instructions instead of 12 for this stage.
#ifndef SSN_REDUCE_PLAIN
template <>
__device__ __forceinline__ float reduce8_to_lane<float>(const float (&acc)[8], int cg) {
    const bool bB = cg & 2, bC = cg & 1;
    float n4[4], n2[2];
    // A DPP read of a VGPR needs two wait states after the VALU write of that VGPR and the hardware does not
    // interlock; the compiler inserts them for its own DPP instructions but cannot see into inline asm.  One s_nop
    // that "redefines" all eight inputs puts every producing FMA in front of it, and with it two wait states in front
    // of every DPP read below.
    float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3], a4 = acc[4], a5 = acc[5], a6 = acc[6], a7 = acc[7];
    asm volatile("s_nop 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    const float lo[4] = {a0, a1, a2, a3}, hi[4] = {a4, a5, a6, a7};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float r;
        asm("v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(lo[k]));
        asm("v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa" : "+v"(r) : "v"(hi[k]));
        n4[k] = r;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float keep = bB ? n4[k + 2] : n4[k];
        const float send = bB ? n4[k] : n4[k + 2];
        n2[k] = keep + dpp_get_f<0x4E>(send);           // quad_perm:[2,3,0,1]: lane i <-> i^2
    }
    const float keep = bC ? n2[1] : n2[0];
    const float send = bC ? n2[0] : n2[1];
    return keep + dpp_get_f<0xB1>(send);                // quad_perm:[1,0,3,2]: lane i <-> i^1
}
#endif

// Tiles of at most 4 rows (acc[4..7] == 0): the first exchange only has to fold the two halves of the lane group
// together -- four DPP adds, no second masked add, no zero inputs to materialise; lanes 0-3 end with rows 0-3.
template <int ROWS, typename T>
__device__ __forceinline__ T reduce_rows_to_lane(const T (&acc)[8], int cg) {
    if constexpr (ROWS > 4 || sizeof(T) != 4 || !SSN_REDUCE4) {
        return reduce8_to_lane(acc, cg);
    } else {
        const bool bB = cg & 2, bC = cg & 1;
        T n4[4], n2[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) n4[k] = acc[k] + dpp_get_f<0x141>(acc[k]);      // row_half_mirror: lane i <-> 7-i
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const T keep = bB ? n4[k + 2] : n4[k];
            const T send = bB ? n4[k] : n4[k + 2];
            n2[k] = keep + dpp_get_f<0x4E>(send);           // quad_perm:[2,3,0,1]: lane i <-> i^2
        }
        const T keep = bC ? n2[1] : n2[0];
        const T send = bC ? n2[0] : n2[1];
        return keep + dpp_get_f<0xB1>(send);                // quad_perm:[1,0,3,2]: lane i <-> i^1
    }
}

template <int C> struct SlabPad {
    // floats per column-group slab in LDS: multiple of 4 (16-B reads) with an ODD number of
    // 16-B units, so the 8 slabs start on distinct 4-bank groups (conflict-free ds_read_b128).
    static constexpr int q = (C + 3) / 4;
    static constexpr int value = 4 * ((q & 1) ? q : q + 1);
};


// acc[s][a] = sum_c w[a][c] * x_s[col(cg, c)] for the lane's RA rows: x is read from an LDS
// image laid out as 8 column-group slabs of SlabPad<C> floats per stimulus.
template <typename T, int RA, int C, int NB>
__device__ __forceinline__ void tile_matvec(const T (&w)[RA][C], const T* xs /* [NB][8*CP] */, int cg,
                                            T (&acc)[NB][8]) {
    constexpr int CP = SlabPad<C>::value;
    constexpr int NQ = (C + 3) / 4;
    using V4 = T __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int s = 0; s < NB; ++s)
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[s][r] = (T)0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        V4 rv[NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) rv[s] = *reinterpret_cast<const V4*>(&xs[s * 8 * CP + cg * CP + 4 * q]);
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

// One element of a raw buffer (byte offset = voff + soff); out-of-range offsets return 0 instead of faulting.
__device__ __forceinline__ float buffer_load_elem(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, float) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
}
__device__ __forceinline__ double buffer_load_elem(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, double) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0));
}

// Load the lane's RA x C tile of a row-major M x M matrix A (TRANSPOSED = false) or of its transpose
// (TRANSPOSED = true: tile element (row, col) = A[col][row]); entries outside M x M are zero.
// Buffer loads: one address VGPR per tile row (clamped row), the column as an immediate / scalar offset;
// out-of-tile columns may read the next row (or past the matrix: the buffer returns 0) and are masked.
// No per-element 64-bit addresses, no per-element branches.
template <typename T, int RA, int C, bool TRANSPOSED>
__device__ __forceinline__ void tile_load(const T* A, int M, int rowbase, int colbase, T (&w)[RA][C]) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A), 0, M * M * (int)sizeof(T), 0x00020000);
    const int ncol = M - colbase;                       // columns c < ncol are inside the matrix
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int row = rowbase + r;
        const int rowc = row < M ? row : M - 1;
        const int voff = (TRANSPOSED ? colbase * M + rowc : rowc * M + colbase) * (int)sizeof(T);
        if constexpr (!TRANSPOSED && sizeof(T) == 4) {
            // the row is contiguous in memory: 16-byte loads (a row group's 8 lanes read one matrix row back to back)
            // (bit_cast of the WHOLE vector: extracting the integer elements first is folded to a splat of one dword
            // load by this compiler)
            using V4 = float __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int c4 = 0; c4 + 4 <= C; c4 += 4) {
                const V4 v = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, c4 * 4, 0));
                w[r][c4] = v.x; w[r][c4 + 1] = v.y; w[r][c4 + 2] = v.z; w[r][c4 + 3] = v.w;
            }
#pragma unroll
            for (int c = C & ~3; c < C; ++c) w[r][c] = buffer_load_elem(rsrc, voff, c * 4, (T)0);
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c)
                w[r][c] = buffer_load_elem(rsrc, voff, (TRANSPOSED ? __builtin_amdgcn_readfirstlane(c * M) : c) * (int)sizeof(T), (T)0);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) w[r][c] = (row < M && c < ncol) ? w[r][c] : (T)0;
    }
}

// ---------------------------------------------------------------------------------------------
// Split residency: the first RA-RL rows of a lane's RA x C tile stay in VGPRs, the last RL rows live in
// LDS.  At 2N = 200 (RA = 7, C = 25) the all-register tile needs 175 + ~70 VGPRs -> 2 waves/SIMD, and a
// single wave issues only one VALU instruction per ~5 cycles; with RL = 2 the kernel fits 168 VGPRs ->
// 3 waves/SIMD, three workgroups per CU (3 x 50 KB of LDS), which is what fills the VALU.
// LDS image per lane: values j = c*RL + rl (column-major over the RL rows), zero-padded to NF4 16-byte
// units; unit k of lane tid sits at base[(k*WS + tid)*4] (conflict-free, one address VGPR + immediate
// offsets).  WS = number of lanes whose row group can hold a real row (8*ceil(8C/RA), 232 of 256 at
// C = 25): the idle lanes of the last wave alias the next unit -- they never write, and what they read
// only reaches accumulators of row groups that have no rows.
// ---------------------------------------------------------------------------------------------
template <int RA, int C, int RL> struct TileSplit {
    static_assert(RL >= 0 && RL <= 2 && RL < RA, "0, 1 or 2 LDS-resident rows");
    static constexpr int RR = RA - RL;
    static constexpr int NL = RL * C;
    static constexpr int NF4 = (NL + 3) / 4;
    static constexpr int WS = 8 * ((8 * C + RA - 1) / RA);
    static constexpr int lds_elems(int wg) { return NF4 > 0 ? ((NF4 - 1) * WS + wg) * 4 : 4; }
};

// The split tile of one lane.  Register rows are stored as PAIRS of rows per column so that one
// v_pk_fma_f32 (r value broadcast to both halves through op_sel) updates two rows: a lone wave issues a
// VALU instruction only every ~4 cycles, and the packed form does twice the work per issue slot, so the
// FMA block keeps the pipe full even while the other waves of the SIMD sit at the barrier.
// At RA = 7, RL = 2: rows (0,1), (2,3) packed, row 4 plain, LDS rows (5,6) packed -> 4 instead of 7
// instructions per column.  (fp64 has no packed form; the same code compiles to scalar v_fma_f64.)
template <typename T, int RA, int C, int RL> struct SplitTile {
    using S = TileSplit<RA, C, RL>;
    static constexpr int RR = S::RR, NP = RR / 2, ODD = RR % 2;
    using V2 = T __attribute__((ext_vector_type(2)));
    using V4 = T __attribute__((ext_vector_type(4)));
    V2 p[NP > 0 ? NP : 1][C];      // p[k][c] = (W[row 2k][c], W[row 2k+1][c])
    T o[ODD ? C : 1];              // last register row when RR is odd

    template <bool TRANSPOSED>
    __device__ __forceinline__ void load(const T* A, int M, int rowbase, int colbase, T* wl, int tid) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A), 0, M * M * (int)sizeof(T), 0x00020000);
        const int ncol = M - colbase;                       // columns c < ncol are inside the matrix
        auto row_voff = [&](int row) {
            const int rowc = row < M ? row : M - 1;
            return (TRANSPOSED ? colbase * M + rowc : rowc * M + colbase) * (int)sizeof(T);
        };
        // one tile row -> t[0..C): 16-byte buffer loads where the row is contiguous in memory (a row group's 8
        // lanes then read one whole matrix row back to back), single elements for the transposed tile / fp64
        auto fetch_row = [&](int row, T (&t)[C]) {
            const int voff = row_voff(row);
            if constexpr (!TRANSPOSED && sizeof(T) == 4) {
                // (bit_cast of the WHOLE vector: extracting the integer elements first is folded to a splat of one
                // dword load by this compiler)
#pragma unroll
                for (int c4 = 0; c4 + 4 <= C; c4 += 4) {
                    const V4 v = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, c4 * 4, 0));
                    t[c4] = v.x; t[c4 + 1] = v.y; t[c4 + 2] = v.z; t[c4 + 3] = v.w;
                }
#pragma unroll
                for (int c = C & ~3; c < C; ++c) t[c] = buffer_load_elem(rsrc, voff, c * 4, (T)0);
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    t[c] = buffer_load_elem(rsrc, voff, (TRANSPOSED ? __builtin_amdgcn_readfirstlane(c * M) : c) * (int)sizeof(T), (T)0);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) t[c] = (row < M && c < ncol) ? t[c] : (T)0;
        };
        // LDS rows first, fenced off from the register rows (fewer values in flight at once)
        if (RL > 0 && tid < S::WS) {
            T t[RL > 0 ? RL : 1][C];
#pragma unroll
            for (int rl = 0; rl < RL; ++rl) fetch_row(rowbase + RR + rl, t[rl]);
#pragma unroll
            for (int k = 0; k < S::NF4; ++k) {
                T v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * k + e, c = j / (RL > 0 ? RL : 1), rl = j % (RL > 0 ? RL : 1);
                    v[e] = (j < S::NL) ? t[rl][c] : (T)0;
                }
                V4 q; q.x = v[0]; q.y = v[1]; q.z = v[2]; q.w = v[3];
                *reinterpret_cast<V4*>(&wl[((size_t)k * S::WS + tid) * 4]) = q;
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            T t[C];
            fetch_row(rowbase + r, t);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (ODD && r == RR - 1) o[c] = t[c];
                else if (r & 1) p[r / 2][c].y = t[c];
                else p[r / 2][c].x = t[c];
            }
        }
    }

    // acc[s][a] = sum_c W[a][c] * x_s[col(cg, c)], rows a < RR from VGPRs, rows a >= RR from the LDS image.
    // Software pipeline per quad q of columns: issue the x reads of quad q+1 and the W units of quad q, run the
    // register FMAs of quad q (which cover the LDS latency), then the LDS-row FMAs.
    template <int NB>
    __device__ __forceinline__ void matvec(const T* wl, int tid, const T* xs /* [NB][8*CP] */, int cg,
                                           T (&acc)[NB][8]) const {
        constexpr int CP = SlabPad<C>::value;
        constexpr int NQ = (C + 3) / 4;
        // Stage boundaries are enforced with data dependencies (an empty asm that "redefines" the LDS
        // addresses and the accumulators): the reads of stage q+1 cannot be hoisted above it, the FMAs of
        // stage q cannot sink below it.  __builtin_amdgcn_sched_barrier alone is not enough -- instruction
        // selection already reorders the (side-effect free) FMAs across it and every read ends up at the top
        // (~80 staging VGPRs).  The tied values are 32-bit LDS byte addresses: tying a generic pointer would
        // lose its address space and turn every ds_read into a flat_load.
        using LdsV4 = const __attribute__((address_space(3))) V4*;
        using LdsT = const __attribute__((address_space(3))) T*;
        unsigned xa = (unsigned)(size_t)(LdsT)(xs + cg * CP);
        unsigned wa = (unsigned)(size_t)(LdsT)(wl + tid * 4);
        V2 a2[NB][NP > 0 ? NP : 1], al2[NB];
        T ao[NB], al1[NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) {
#pragma unroll
            for (int k = 0; k < NP; ++k) a2[s][k] = (V2){(T)0, (T)0};
            al2[s] = (V2){(T)0, (T)0};
            ao[s] = al1[s] = (T)0;
        }
        V4 rv[NQ][NB];
        V4 wu[S::NF4 > 0 ? S::NF4 : 1];
        const V4 stub = {(T)cg, (T)(cg + 1), (T)(cg + 2), (T)(cg + 3)};     // diagnostic builds (SSN_ABLATE) only
        auto load_quad = [&](int q) {
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                if constexpr (SSN_ABLATE & 8) rv[q][s] = stub;
                else rv[q][s] = *(LdsV4)(size_t)(xa + (unsigned)sizeof(T) * (s * 8 * CP + 4 * q));
            }
        };
        load_quad(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + 1 < NQ) load_quad(q + 1);
#pragma unroll
            for (int uu = 0; uu < RL; ++uu) {                         // values 4*q*RL .. 4*(q+1)*RL - 1
                const int k = q * RL + uu;
                if (k < S::NF4) {
                    if constexpr (SSN_ABLATE & 32) wu[k] = stub;
                    else wu[k] = *(LdsV4)(size_t)(wa + (unsigned)sizeof(T) * (k * S::WS * 4));
                }
            }
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                const T rr[4] = {rv[q][s].x, rv[q][s].y, rv[q][s].z, rv[q][s].w};
                // fp32: acc += w2 * (r, r) with r taken from ONE half of the aligned register pair the 16-byte read left it
                // in (op_sel broadcasts the low or the high half to both lanes of v_pk_fma_f32).  Written as asm because
                // the compiler, short of registers, copies every fourth r into a fresh pair first (one v_mov per quad and
                // accumulator group: ~15 of ~190 instructions per step).
                const V2 rlo = {rv[q][s].x, rv[q][s].y}, rhi = {rv[q][s].z, rv[q][s].w};
                auto pk_bcast = [&](V2& acc, const V2& w2, int e) {
                    if constexpr (sizeof(T) == 4 && SSN_PK_ASM) {
                        if (e == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w2), "v"(rlo));
                        else if (e == 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w2), "v"(rlo));
                        else if (e == 2) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w2), "v"(rhi));
                        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w2), "v"(rhi));
                    } else {
                        acc = __builtin_elementwise_fma(w2, (V2){rr[e], rr[e]}, acc);
                    }
                };
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * q + e;
                    if (c < C) {
#pragma unroll
                        for (int k = 0; k < NP; ++k) pk_bcast(a2[s][k], p[k][c], e);
                        if (ODD) ao[s] = fma(o[c], rr[e], ao[s]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * q + e;
                    if (c < C) {
                        if constexpr (RL == 2) {
                            const int j = c * 2;
                            const V2 wp = {wu[j / 4][j % 4], wu[j / 4][j % 4 + 1]};
                            pk_bcast(al2[s], wp, e);
                        } else if constexpr (RL == 1) {
                            al1[s] = fma(wu[c / 4][c % 4], rr[e], al1[s]);
                        }
                    }
                }
            }
            if (q + 1 < NQ) {
#pragma unroll
                for (int s = 0; s < NB; ++s) {
#pragma unroll
                    for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(a2[s][k]));
                    if (ODD) asm volatile("" : "+v"(ao[s]));
                    if (RL == 2) asm volatile("" : "+v"(al2[s]));
                    if (RL == 1) asm volatile("" : "+v"(al1[s]));
                }
                asm volatile("" : "+v"(xa), "+v"(wa));
            }
        }
#pragma unroll
        for (int s = 0; s < NB; ++s) {
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[s][r] = (T)0;
#pragma unroll
            for (int k = 0; k < NP; ++k) { acc[s][2 * k] = a2[s][k].x; acc[s][2 * k + 1] = a2[s][k].y; }
            if (ODD) acc[s][RR - 1] = ao[s];
            if (RL == 2) { acc[s][RR] = al2[s].x; acc[s][RR + 1] = al2[s].y; }
            if (RL == 1) acc[s][RR] = al1[s];
        }
    }
};


// Static wave priority by the workgroup's (guessed) rank among the three workgroups sharing its CU, so that the
// arbiter prefers one workgroup's waves on all four SIMDs instead of round-robining (which keeps the co-resident
// workgroups in phase: all in their FMA block, then all in their serial part).  Worth ~2 % at the C2 shape.
__device__ __forceinline__ void set_rank_priority(int rank) {
    if (SSN_PRIO_MODE != 2) return;
    if (rank == 0) __builtin_amdgcn_s_setprio(3);
    else if (rank == 1) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(1);
}   // all-register kernels use tile_load / tile_matvec

}  // namespace ssn
