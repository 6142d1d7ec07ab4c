// dL/dW of the BPTT generator update: gW[b] = Delta[b]^T X[b], one M x M matrix per weight draw, contracted over
// K = NB * T (stimulus, time) pairs:  gW[b][i][j] = sum_k Delta[b][k][i] * X[b][k][j]
// (Delta = the adjoint sweep's shifted delta_t, X = the saved trajectory x_{t-1}; reference: the `theano.grad` of
// networks/wgan.py:236-242 through the scan of networks/ssn.py:354-385).  At BASELINE config 3 this is 1024 GEMMs of
// 200 x 200 x 9600 = 786 GFLOP and 15.8 GB of operands read once: as much arithmetic as the forward recurrence.
//
// gw_split_kernel (fp32, 2N <= 224): the bf16 matrix cores run 16x the fp32 MFMA rate, so each fp32 operand is split
// EXACTLY into three bf16 terms  x = h + m + l  (8 + 8 + 8 significand bits = the 24 of fp32; h = bf16(x),
// m = bf16(x - h), l = bf16(x - h - m), every subtraction exact) and the product is formed from the six partial products
// that reach 2^-24 relative:  d x = dh xh + dh xm + dm xh + dm xm + dh xl + dl xh  (dropped: m l, l m, l l <= 2^-24 |d x|),
// each an exact product of two bf16 values accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs cost 6/16 of
// one fp32 MFMA: the GEMM needs 2.4 ms of matrix-pipe time at C3 instead of 5.0 and becomes HBM-bound (15.8 GB).
// The result carries fp32 input precision (tests/test_generator_gpu.py checks it against fp64 to ~1e-6).
//
// gw_split_kernel<NT, true> (round 3): the same GEMM with every operand as TWO fp16 numbers by round to nearest,
// x 2^e = h + m, |x 2^e - h - m| <= 2^-24 |x 2^e| (h = rn16(s), m = rn16(s - h): 11 + 1 + 11 significant bits while m stays
// above the fp16 subnormal step, i.e. for elements within 2^15 of the largest), and the three partial products that reach
// 2^-24: dh xh + dh xm + dm xh (dropped: dm xm <= 2^-24 |d x|) on v_mfma_f32_32x32x16_f16: HALF the matrix work of the bf16
// form at the same accuracy.  fp16 has 5 exponent bits, so each operand needs a power-of-two scale that brings its largest
// magnitude to [2^14, 2^15): one per draw for Delta (max |delta| of the draw, which the two-draw adjoint sweep of
// ssn_duo.hip tracks for its own scaling and hands over), one per call for X (the rate bound of the saturating I/O
// function).  Values 2^-40 of the largest and below are lost -- against sum_k |d||x| of the same draw they do not count.
//
// One workgroup (8 waves) per draw keeps the whole M x M accumulator on chip (7 x 7 tiles of 32 x 32 at 2N = 200: wave
// (rh, cq) owns tile rows rh*4.. and tile columns cq*2..), streams Delta and X in slabs of 16 k through LDS (split once
// per element, shared by all waves; [row][k] layout, 48-byte row stride: conflict-free 16-byte operand reads), double
// buffered, one barrier per slab, global loads two slabs ahead.  HBM traffic = the operands once + the result once.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <type_traits>
#include <atomic>
#include "ssn_host.h"

// Ablation builds for timing only (tools/ab_one.sh <tag> ssn_gw -DGW_ABLATE=n; results are wrong): 1 no MFMAs and no operand
// reads, 2 no global loads inside the loop, 3 no split and no LDS stores
#ifndef GW_ABLATE
#define GW_ABLATE 0
#endif

namespace ssn {

typedef float gf16 __attribute__((ext_vector_type(16)));
typedef short gb8 __attribute__((ext_vector_type(8)));
typedef unsigned gu4 __attribute__((ext_vector_type(4)));

// ---- workgroup shape -------------------------------------------------------------------------------------------
// NT x NT tiles of 32 x 32 (NT = 7 covers 2N <= 224).  Tiles -> waves for NT = 7 (49 tiles, 8 waves, two per SIMD: waves w and
// w + 4 share one): six 2 x 3 blocks tile the leading 6 x 6, wave 6 takes tile row 6 (7 tiles), wave 7 the rest of tile
// column 6 (6 tiles): 6,6,6,6,6,6,7,6 tiles -> 12,12,13,12 per SIMD.  For NT = 4: eight 2 x 1 blocks.
template <int NT> struct GwBlocks;
template <> struct GwBlocks<7> {
    // class 0: 2 x 3 block at (2 (w / 2), 3 (w % 2)); class 1: 1 x 7 at (6, 0); class 2: 6 x 1 at (0, 6)
    __device__ static int cls(int w) { return w < 6 ? 0 : w - 5; }
    __device__ static int r0(int w) { return w < 6 ? 2 * (w >> 1) : (w == 6 ? 6 : 0); }
    __device__ static int c0(int w) { return w < 6 ? 3 * (w & 1) : (w == 6 ? 0 : 6); }
};
template <> struct GwBlocks<4> {
    __device__ static int cls(int) { return 3; }                   // class 3: 2 x 1 block at (2 (w / 4), w % 4)
    __device__ static int r0(int w) { return 2 * (w >> 2); }
    __device__ static int c0(int w) { return w & 3; }
};

// One slab = 16 k.  Staging task of thread t: matrix t >> 8 (wave-uniform), k half (t >> 7) & 1, rows 2 p and 2 p + 1
// with p = t & 127: eight 8-byte loads (coalesced: a wave reads 512 contiguous bytes per k), split, six 16-byte LDS
// stores.  LDS: [buffer][matrix][part] images of 256 rows (rows >= 2N hold zeros), layout below.
struct GwStage {
    __amdgpu_buffer_rsrc_t rs;
    float scale;         // fp16 form: the power of two this thread's matrix is multiplied with before the split
    int voff;            // byte offset of (k = 8 half, row 2 p), or -1 (row pair beyond the matrix: loads return 0)
    int k_stride;        // M * 4
    int pad;             // (16 - K % 16) % 16: slab s starts at k = 16 s - pad
    unsigned lds;        // byte offset of (matrix, part 0, row 2 p, half) inside one buffer
};

// Slab s covers k in [16 s - pad, 16 s - pad + 16), pad = (16 - K % 16) % 16: only slab 0 can reach below k = 0 (it is
// fetched by gw_fetch_first, which masks those k); every later slab is whole, and a slab past the end starts at or
// beyond K.  The hardware bounds-checks the VGPR part of a raw-buffer offset only (not the SGPR part), so the slab's
// base goes into the VGPR offset -- a slab past the end is rejected as a whole and reads zeros -- and only the k index
// inside the slab (< 16 rows, in range by construction) rides in the SGPR offset.
__device__ __forceinline__ void gw_fetch(const GwStage& g, int slab, float (&raw)[16]) {
    const int vo = g.voff < 0 ? -1 : g.voff + (slab * 16 - g.pad) * g.k_stride;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        // (bit_cast of the WHOLE result to a float vector: extracting integer elements first is folded to a splat of
        // one dword load by this compiler, see ssn_tile_core.h)
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 v = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(g.rs, vo, e * g.k_stride, 0));
        raw[e] = v.x;                                      // row 2 p,     k = k0 + 8 half + e
        raw[8 + e] = v.y;                                  // row 2 p + 1
    }
}
__device__ __forceinline__ void gw_fetch_first(const GwStage& g, int half, float (&raw)[16]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const int k = 8 * half + e - g.pad;
        const int vo = (g.voff < 0 || k < 0) ? -1 : g.voff + (e - g.pad) * g.k_stride;
        const f2 v = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(g.rs, vo, 0, 0));
        raw[e] = v.x;
        raw[8 + e] = v.y;
    }
}

typedef __bf16 gbf2 __attribute__((ext_vector_type(2)));
typedef float gf2 __attribute__((ext_vector_type(2)));
// (a, b) -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned gw_pack(float a, float b) {
    const gf2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gbf2));
}
__device__ __forceinline__ float gw_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float gw_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// 8 consecutive k of one row: x = h + m + l, each packed as 8 bf16 (5.5 VALU per element)
__device__ __forceinline__ void gw_split_row(const float* v, gu4& ph, gu4& pm, gu4& pl) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = v[2 * q], b = v[2 * q + 1];
        const unsigned h = gw_pack(a, b);
        const float ra = a - gw_lo(h), rb = b - gw_hi(h);
        const unsigned m = gw_pack(ra, rb);
        const float sa = ra - gw_lo(m), sb = rb - gw_hi(m);
        ph[q] = h; pm[q] = m; pl[q] = gw_pack(sa, sb);
    }
}

// LDS image of one (matrix, part): row r, k half h at gw_row(r) + 16 h.  Even rows fill 128 slots of 48 bytes, odd rows
// a second plane 128 bytes further on: a staging thread writes rows 2 p and 2 p + 1 at a lane stride of 48 bytes in both
// planes (conflict-free 16-byte stores), and the 16 consecutive rows of one ds_read_b128 lane group fall on 64 distinct
// banks (the odd plane's skew of 32 banks = what 8 more even slots would add).
constexpr unsigned GW_ROWB = 48, GW_ODD = 128 * GW_ROWB + 128, GW_PARTB = 2 * 128 * GW_ROWB + 256;
template <bool F16> struct GwFmt {
    static constexpr unsigned NP = F16 ? 2 : 3;          // parts per operand
    static constexpr unsigned MATB = NP * GW_PARTB, BUFB = 2 * MATB;
};
__device__ __forceinline__ unsigned gw_row(unsigned r) { return (r >> 1) * GW_ROWB + (r & 1) * GW_ODD; }

typedef _Float16 gh2 __attribute__((ext_vector_type(2)));
typedef _Float16 gh8 __attribute__((ext_vector_type(8)));
// 8 consecutive k of one row, fp16 form: s = x * scale = h + m by round to nearest (v_cvt_pk_f16_f32; 2.5 VALU per element)
__device__ __forceinline__ void gw_split_row_f16(const float* v, float sc, gu4& ph, gu4& pm) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = v[2 * q] * sc, b = v[2 * q + 1] * sc;
        const gh2 h = __builtin_convertvector((gf2){a, b}, gh2);
        const float ra = a - (float)h[0], rb = b - (float)h[1];
        ph[q] = __builtin_bit_cast(unsigned, h);
        pm[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((gf2){ra, rb}, gh2));
    }
}

template <bool F16>
__device__ __forceinline__ void gw_stage(const GwStage& g, char* buf, const float (&raw)[16]) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        char* p = buf + g.lds + rr * GW_ODD;
        if constexpr (F16) {
            gu4 ph, pm;
            gw_split_row_f16(&raw[8 * rr], g.scale, ph, pm);
            *reinterpret_cast<gu4*>(p) = ph;
            *reinterpret_cast<gu4*>(p + GW_PARTB) = pm;
        } else {
            gu4 ph, pm, pl;
            gw_split_row(&raw[8 * rr], ph, pm, pl);
            *reinterpret_cast<gu4*>(p) = ph;
            *reinterpret_cast<gu4*>(p + GW_PARTB) = pm;
            *reinterpret_cast<gu4*>(p + 2 * GW_PARTB) = pl;
        }
    }
}

// The main loop of one wave class: an NR x NC block of tiles at tile (r0, c0).  Every class runs the same barriers.
template <int NR, int NC, bool F16>
__device__ __forceinline__ void gw_run(const GwStage& g, char* sm, int nslab, int r0, int c0, int lane, int half, float* out, int M,
                                       float osc0, float osc1) {
    constexpr unsigned GW_MATB = GwFmt<F16>::MATB, GW_BUFB = GwFmt<F16>::BUFB;
    gf16 acc[NR][NC];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
    const int orow = lane & 31, ohalf = lane >> 5;
    const unsigned a_off = gw_row(r0 * 32 + orow) + ohalf * 16;                 // matrix 0, part 0
    const unsigned b_off = GW_MATB + gw_row(c0 * 32 + orow) + ohalf * 16;       // matrix 1, part 0

    float raw[3][16];
    gw_fetch_first(g, half, raw[0]);
    gw_stage<F16>(g, sm, raw[0]);
    gw_fetch(g, 1, raw[1]);           // (slabs past the end read zeros)
    gw_fetch(g, 2, raw[2]);
    gw_fetch(g, 3, raw[0]);
    __syncthreads();

    auto mma = [&](const char* buf) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const char* pa = buf + a_off + r * 16 * GW_ROWB;
            if constexpr (F16) {
                const gh8 ah = *reinterpret_cast<const gh8*>(pa);
                const gh8 am = *reinterpret_cast<const gh8*>(pa + GW_PARTB);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const char* pb = buf + b_off + c * 16 * GW_ROWB;
                    const gh8 bh = *reinterpret_cast<const gh8*>(pb);
                    const gh8 bm = *reinterpret_cast<const gh8*>(pb + GW_PARTB);
                    gf16 a = acc[r][c];                  // smallest partial products first
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, a, 0, 0, 0);
                    acc[r][c] = a;
                }
                continue;
            }
            const gb8 ah = *reinterpret_cast<const gb8*>(pa);
            const gb8 am = *reinterpret_cast<const gb8*>(pa + GW_PARTB);
            const gb8 al = *reinterpret_cast<const gb8*>(pa + 2 * GW_PARTB);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const char* pb = buf + b_off + c * 16 * GW_ROWB;
                const gb8 bh = *reinterpret_cast<const gb8*>(pb);
                const gb8 bm = *reinterpret_cast<const gb8*>(pb + GW_PARTB);
                const gb8 bl = *reinterpret_cast<const gb8*>(pb + 2 * GW_PARTB);
                gf16 a = acc[r][c];                      // smallest partial products first
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a, 0, 0, 0);
                acc[r][c] = a;
            }
        }
    };
    // Slab s is multiplied from buffer s & 1 while slab s + 1 (in registers since three iterations) is split into the
    // other buffer and slab s + 4 is requested from HBM: three slabs = 77 KB per CU in flight (with two, the kernel ran at
    // 4.3 TB/s = the bytes in flight over the loaded latency).  The three are independent and sit in one basic block: the
    // compiler puts the VALU of the split between the MFMAs (explicit sched_group_barrier pipelines changed nothing).
    // Register sets rotate with period 3, LDS buffers with period 2: six slabs per loop iteration, spelled out.
    auto it = [&](int s, auto R, auto P) {
        constexpr int r = decltype(R)::value, p = decltype(P)::value;
#if GW_ABLATE != 1
        mma(sm + p * GW_BUFB);
#endif
#if GW_ABLATE != 3
        gw_stage<F16>(g, sm + (1 - p) * GW_BUFB, raw[r]);
#endif
#if GW_ABLATE != 2
        gw_fetch(g, s + 4, raw[r]);
#endif
        __syncthreads();
    };
    constexpr std::integral_constant<int, 0> I0{};
    constexpr std::integral_constant<int, 1> I1{};
    constexpr std::integral_constant<int, 2> I2{};
    for (int s = 0; s < nslab; s += 6) {
        it(s, I1, I0);
        it(s + 1, I2, I1);
        it(s + 2, I0, I0);
        it(s + 3, I1, I1);
        it(s + 4, I2, I0);
        it(s + 5, I0, I1);
    }
    // (a slab count that is not a multiple of six multiplies slabs of zeros at the end)

    // ---- epilogue: C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = (c0 + c) * 32 + orow;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = (r0 + r) * 32 + (e & 3) + 8 * (e >> 2) + 4 * ohalf;
                if (i < M && j < M) out[(size_t)i * M + j] = F16 ? acc[r][c][e] * osc0 * osc1 : acc[r][c][e];
            }
        }
}

// e = 14 - floor(log2 v) from the bit pattern of v > 0, clamped to +-100 (0 and denormals: 100; inf / NaN: -100, the
// products are then inf / NaN like the operands)
__device__ __forceinline__ int gw_exp(unsigned bits) {
    const int biased = (int)((bits >> 23) & 0xffu);
    const int e = 14 - ((biased ? biased : 1) - 127);
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
__device__ __forceinline__ float gw_pow2(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }

// F16: dmax[b] = bit pattern of max |delta[b]| (or of any bound on it), xmax = a bound on |traj|
template <int NT, bool F16>
__global__ void __launch_bounds__(512, 2) gw_split_kernel(const float* __restrict__ delta, const float* __restrict__ traj,
                                                          float* __restrict__ gW, long K, int M,
                                                          const unsigned* __restrict__ dmax, float xmax) {
    constexpr unsigned GW_MATB = GwFmt<F16>::MATB;
    extern __shared__ __align__(16) char sm[];           // 2 buffers of GW_BUFB bytes
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long KM = K * (long)M;
    const int mat = __builtin_amdgcn_readfirstlane(tid >> 8);
    const float* src = (mat == 0 ? delta : traj) + (size_t)b * KM;
    GwStage g;
    g.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(KM * 4), 0x00020000);
    const int half = (tid >> 7) & 1, pair = tid & 127;
    g.voff = (2 * pair < M) ? (8 * half * M + 2 * pair) * 4 : -1;
    g.k_stride = M * 4;
    g.pad = (int)((16 - K % 16) % 16);
    g.lds = (unsigned)mat * GW_MATB + gw_row(2 * pair) + half * 16;
    float osc0 = 1.f, osc1 = 1.f;
    g.scale = 1.f;
    if constexpr (F16) {
        const int ed = gw_exp(dmax[b]), ex = gw_exp(__builtin_bit_cast(unsigned, xmax));
        g.scale = gw_pow2(mat == 0 ? ed : ex);
        osc0 = gw_pow2(-ed); osc1 = gw_pow2(-ex);
    }
    const int nslab = (int)((K + 15) / 16);
    float* out = gW + (size_t)b * M * M;
    using BL = GwBlocks<NT>;
    const int cls = __builtin_amdgcn_readfirstlane(BL::cls(wave));
    const int r0 = __builtin_amdgcn_readfirstlane(BL::r0(wave)), c0 = __builtin_amdgcn_readfirstlane(BL::c0(wave));
    if constexpr (NT == 7) {
        if (cls == 0) gw_run<2, 3, F16>(g, sm, nslab, r0, c0, lane, half, out, M, osc0, osc1);
        else if (cls == 1) gw_run<1, 7, F16>(g, sm, nslab, r0, c0, lane, half, out, M, osc0, osc1);
        else gw_run<6, 1, F16>(g, sm, nslab, r0, c0, lane, half, out, M, osc0, osc1);
    } else {
        gw_run<2, 1, F16>(g, sm, nslab, r0, c0, lane, half, out, M, osc0, osc1);
    }
}

// Any size and fp64: classic 16 x 16 output tile per workgroup, 16-deep slabs through LDS, plain FMAs in k order.
template <typename T>
__global__ void __launch_bounds__(256) gw_simple_kernel(const T* __restrict__ delta, const T* __restrict__ traj,
                                                        T* __restrict__ gW, long K, int M) {
    __shared__ T As[16][17], Bs[16][17];
    const int nt = (M + 15) / 16;
    const int b = blockIdx.x / (nt * nt), tile = blockIdx.x % (nt * nt);
    const int i0 = (tile / nt) * 16, j0 = (tile % nt) * 16;
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    const T* D = delta + (size_t)b * K * M;
    const T* X = traj + (size_t)b * K * M;
    T acc = 0;
    for (long k0 = 0; k0 < K; k0 += 16) {
        const long k = k0 + ti;                          // thread (ti, tj) stages element (k0 + ti, tile offset tj)
        As[ti][tj] = (k < K && i0 + tj < M) ? D[k * M + i0 + tj] : (T)0;
        Bs[ti][tj] = (k < K && j0 + tj < M) ? X[k * M + j0 + tj] : (T)0;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc += As[kk][ti] * Bs[kk][tj];
        __syncthreads();
    }
    if (i0 + ti < M && j0 + tj < M) gW[(size_t)b * M * M + (size_t)(i0 + ti) * M + j0 + tj] = acc;
}

template <int NT, bool F16>
static hipError_t gw_launch_split(const float* delta, const float* traj, float* gW, int B, long K, int M, const unsigned* dmax,
                                  float xmax, hipStream_t st) {
    const size_t lds = 2 * (size_t)GwFmt<F16>::BUFB;     // 150528 (bf16 x 3) / 100352 (fp16 x 2) bytes: one workgroup per CU
    // The dynamic-LDS limit is an attribute of the function ON THE CURRENT DEVICE: one flag per device (a process that
    // drives several cards launches this kernel on each), set after the attribute call has succeeded; two threads racing
    // on a device both make the (idempotent) call.
    static std::atomic<bool> done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return hipErrorInvalidDevice;
    if (dev >= 64 || !done[dev].load(std::memory_order_acquire)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gw_split_kernel<NT, F16>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev < 64) done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((gw_split_kernel<NT, F16>), dim3(B), dim3(512), lds, st, delta, traj, gW, K, M, dmax, xmax);
    return hipGetLastError();
}
static bool gw_split_ok(long K, int M) { return M <= 224 && K * (long)M * 4 < (1L << 31); }

template <typename T>
hipError_t launch_weight_grad(const T* delta, const T* traj, T* gW, int B, long K, int M, int kernel, hipStream_t st) {
    if (B <= 0 || M <= 0) return hipSuccess;
    if (K <= 0) return hipMemsetAsync(gW, 0, sizeof(T) * (size_t)B * M * M, st);
    if constexpr (sizeof(T) == 4) {
        // kernel: 0 automatic, 1 plain-FMA kernel, 2 split-bf16 MFMA kernel
        const bool split_ok = gw_split_ok(K, M);
        if (kernel == 2 && !split_ok) return hipErrorInvalidValue;
        if ((kernel == 0 && split_ok && M > 32) || kernel == 2) {
            if (M <= 128) return gw_launch_split<4, false>(delta, traj, gW, B, K, M, nullptr, 0.f, st);
            return gw_launch_split<7, false>(delta, traj, gW, B, K, M, nullptr, 0.f, st);
        }
    }
    const int nt = (M + 15) / 16;
    hipLaunchKernelGGL((gw_simple_kernel<T>), dim3((unsigned)((long)B * nt * nt)), dim3(256), 0, st, delta, traj, gW, K, M);
    return hipGetLastError();
}
// The fp16 two-part form: dmax[b] = bit pattern of a bound on |delta[b]| (device), xmax = a bound on |traj| (> 0, finite)
hipError_t launch_weight_grad_scaled(const float* delta, const float* traj, float* gW, int B, long K, int M, const unsigned* dmax,
                                     float xmax, hipStream_t st) {
    if (B <= 0 || M <= 0) return hipSuccess;
    if (K <= 0) return hipMemsetAsync(gW, 0, sizeof(float) * (size_t)B * M * M, st);
    if (!gw_split_ok(K, M) || !dmax || !(xmax > 0.f) || !(xmax < __builtin_inff())) return hipErrorInvalidValue;
    if (M <= 128) return gw_launch_split<4, true>(delta, traj, gW, B, K, M, dmax, xmax, st);
    return gw_launch_split<7, true>(delta, traj, gW, B, K, M, dmax, xmax, st);
}
template hipError_t launch_weight_grad<float>(const float*, const float*, float*, int, long, int, int, hipStream_t);
template hipError_t launch_weight_grad<double>(const double*, const double*, double*, int, long, int, int, hipStream_t);

}  // namespace ssn
