// Shared device pieces of the "tile" register-stationary layout (see ssn_tile.hip):
// 8x8 lane grid per wave, RA x C tile of the matrix per lane, in-wave transpose-reduce.
#pragma once
#include <hip/hip_runtime.h>
#include "ssn_device.h"

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

// Load the lane's RA x C tile of a row-major M x M matrix A (TRANSPOSED = false) or of its transpose
// (TRANSPOSED = true: tile element (row, col) = A[col][row]); entries outside M x M are zero.
// Unconditional loads from clamped addresses, masked by multiplication (no per-element branches).
template <typename T, int RA, int C, bool TRANSPOSED>
__device__ __forceinline__ void tile_load(const T* A, int M, int rowbase, int colbase, T (&w)[RA][C]) {
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int row = rowbase + r;
        const int rowc = row < M ? row : M - 1;
        const T rmask = (row < M) ? (T)1 : (T)0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int col = colbase + c;
            const int colc = col < M ? col : M - 1;
            const T v = TRANSPOSED ? A[(size_t)colc * M + rowc] : A[(size_t)rowc * M + colc];
            w[r][c] = v * ((col < M) ? rmask : (T)0);
        }
    }
}

}  // namespace ssn
