// WGAN-GP critic on MI355X (gfx950): MLP forward / backward / gradient-penalty
// double-backward as a chain of small MFMA GEMMs, plus the optimizer kernels.
//
// Reference semantics (Lasagne/Theano graphs; restated in oracle/gan_torch.py):
//   network  networks/cwgan.py:123-175 + simple_discriminator.py:139-165 ('none' normalization):
//            h0 = [x, contrast, |norm_probe|, cell_type]; h_l = relu(h_{l-1} W_l + b_l);
//            D = h_L w_out (no bias).
//   loss     cwgan.py:190-214: mean D(xg) - mean D(xd) + lambda * mean((||dD(xp)/dxp||_2 - 1)^2),
//            gradient taken w.r.t. the tuning-curve part of the input only.
//   updates  wgan.py:111-165 (Updater: adam-wgan / rmsprop / sgd, L1/L2 penalty and decay).
//
// The gradient penalty needs the gradient of an input-gradient.  For a ReLU MLP the
// input gradient is a LINEAR chain in the weights once the activation masks m_l are
// fixed:  v_L = m_L * w_out^T,  v_{l-1} = m_{l-1} * (v_l W_l^T),  g = v_1 W_1^T,
// so its parameter gradient is ordinary backprop through that chain with upstream
// dP/dg = 2 (||g_x|| - 1)/||g_x|| * g_x / batch (Theano's jacobian builds a
// batch x batch x NB tensor for the same quantity, cwgan.py:210-211).
//
// GEMM kernel: 64x64 tile per 256-thread workgroup, 4 waves x one 32x32 MFMA tile.
//   bf16 path  v_mfma_f32_32x32x16_bf16, operands converted from the fp32 master copies
//              while staging into LDS (fp32 accumulate, fp32 results);
//   fp32 path  v_mfma_f32_32x32x2_f32 (exact fp32 products) for parity tests.
// The layer GEMMs here are tiny (batch x 11..512 x 512): they are launch/latency bound,
// not MFMA bound; the tile is chosen for simplicity and full generality in the strides.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <type_traits>
#include <cstdlib>
#include "ssn_host.h"
#include "ssn_critic_dev.h"

namespace ssn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short to_bf16(float x) {
    // round-to-nearest-even (plain cast semantics; keeps NaN a NaN)
    __hip_bfloat16 h = __float2bfloat16(x);
    return __builtin_bit_cast(unsigned short, h);
}

enum { EPI_PLAIN = 0, EPI_BIAS_RELU = 1, EPI_MASK = 2 };

struct GemmArgs {
    const float* A; long sam, sak;     // op(A)(m,k) = A[m*sam + k*sak]
    const float* B; long sbk, sbn;     // op(B)(k,n) = B[k*sbk + n*sbn]
    float* C; long ldc;                // C[m*ldc + n]
    int M, N, K;
    float alpha, beta;                 // C = alpha*acc + beta*C   (PLAIN only)
    const float* bias;                 // [N]          (BIAS_RELU)
    const float* mask; long ldm;       // [M][ldm] > 0 (MASK): C = acc * (mask > 0 ? 1 : leak)
    float leak;                        // slope of the hidden nonlinearity below zero: 0 rectify, 0.01 / 1/3 leaky, 1 linear
    int epilogue;
    int kchunk;                        // K range per blockIdx.z (split-K; PLAIN with beta == 1 only)
    float* partial;                    // split-K: slice z writes its tile sums to partial[z][m*N + n] (no atomics)
};

// ---- epilogue of one 32 x 32 wave tile: C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -----------
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16& acc, int mw, int nw, int lane, int bz, int nz) {
    const int n = nw + (lane & 31);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int m = mw + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (m < g.M && n < g.N) {
            float v = acc[reg];
            float* c = g.C + m * g.ldc + n;
            if (g.epilogue == EPI_BIAS_RELU) { v += g.bias[n]; v = v > 0.f ? v : g.leak * v; }
            else if (g.epilogue == EPI_MASK) { v = (g.mask[m * g.ldm + n] > 0.f) ? v : g.leak * v; }
            else if (g.partial) { g.partial[((long)bz * g.M + m) * g.N + n] = v; continue; }   // slab of K slice bz
            else { v = g.alpha * v + (g.beta != 0.f ? g.beta * (*c) : 0.f); }
            *c = v;
        }
    }
}

template <bool BF16>
__global__ void __launch_bounds__(256) gemm_mfma_kernel(GemmArgs g) {
    constexpr int BM = 64, BN = 64, BK = BF16 ? 32 : 16;
    // staged as [outer index][k], k contiguous: both MFMA operands then read consecutive k
    __shared__ __align__(16) unsigned short As16[BF16 ? BM : 1][BF16 ? BK + 8 : 1];
    __shared__ __align__(16) unsigned short Bs16[BF16 ? BN : 1][BF16 ? BK + 8 : 1];
    __shared__ float As32[BF16 ? 1 : BM][BF16 ? 1 : BK + 1];
    __shared__ float Bs32[BF16 ? 1 : BN][BF16 ? 1 : BK + 1];

    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;          // 2 x 2 waves, 32 x 32 each
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
    // Each thread stages EPT elements of op(A) and of op(B) per K tile.  The global loads of tile k+1 are issued
    // into registers before the MFMAs of tile k (software pipelining): these GEMMs are small (a handful of
    // workgroups per CU at best), so an exposed load latency per K step was most of their run time.
    constexpr int EPT = BM * BK / 256;
    float ra[EPT], rb[EPT];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * 256;
            // make the index that is contiguous in memory the fastest-varying one
            int mi, ki;
            if (g.sak == 1) { ki = e % BK; mi = e / BK; } else { mi = e % BM; ki = e / BM; }
            const int m = m0 + mi, k = k0 + ki;
            ra[i] = (m < g.M && k < kend) ? g.A[m * g.sam + k * g.sak] : 0.f;
            int ni, kj;
            if (g.sbk == 1) { kj = e % BK; ni = e / BK; } else { ni = e % BN; kj = e / BN; }
            const int n = n0 + ni, k2 = k0 + kj;
            rb[i] = (n < g.N && k2 < kend) ? g.B[k2 * g.sbk + n * g.sbn] : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * 256;
            int mi, ki;
            if (g.sak == 1) { ki = e % BK; mi = e / BK; } else { mi = e % BM; ki = e / BM; }
            if constexpr (BF16) As16[mi][ki] = to_bf16(ra[i]); else As32[mi][ki] = ra[i];
            int ni, kj;
            if (g.sbk == 1) { kj = e % BK; ni = e / BK; } else { ni = e % BN; kj = e / BN; }
            if constexpr (BF16) Bs16[ni][kj] = to_bf16(rb[i]); else Bs32[ni][kj] = rb[i];
        }
    };
    if (kbeg < kend) load_tile(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        store_tile();
        __syncthreads();
        if (k0 + BK < kend) load_tile(k0 + BK);
        if constexpr (BF16) {
            // lane l: A[row l&31][k = 8*(l>>5) + j], B[k = 8*(l>>5) + j][col l&31], j = 0..7 (per 16-deep step)
#pragma unroll
            for (int kk = 0; kk < BK; kk += 16) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(&As16[wr * 32 + (lane & 31)][kk + 8 * (lane >> 5)]);
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(&Bs16[wc * 32 + (lane & 31)][kk + 8 * (lane >> 5)]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
            }
        } else {
            // lane l: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31] (per 2-deep step)
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float af = As32[wr * 32 + (lane & 31)][kk + (lane >> 5)];
                const float bf = Bs32[wc * 32 + (lane & 31)][kk + (lane >> 5)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    gemm_epilogue(g, acc, m0 + wr * 32, n0 + wc * 32, lane, blockIdx.z, gridDim.z);
}

// ---- deterministic split-K ---------------------------------------------------------------------------------------
// Weight-gradient GEMMs contract over the batch (K = 1024..3072) into a small M x N: few output tiles, long K loop, so
// K is split over blockIdx.z to fill the chip.  The slices do NOT add into C with atomics (the order of fp32 atomic
// adds changes from run to run, and Adam turns a last-bit difference of a near-zero gradient into a full-size step:
// two identical-seed GAN runs drifted apart within a few updates).  Each slice stores its sums to a scratch slab and ONE
// reduction kernel per critic update adds the slabs of every split GEMM of that update in slice order:
// C += alpha * (p_0 + p_1 + ...).  The plan lives on the calling thread between begin and flush.
struct SplitKEntry { float* C; const float* partial; long mn; int splits; float alpha; };
constexpr int kMaxSplitK = 24;
struct SplitKPlan { float* scratch; size_t cap, used; int n; SplitKEntry e[kMaxSplitK]; };
static thread_local SplitKPlan tl_plan{nullptr, 0, 0, 0, {}};
// entries that accumulate into the same C (the [xg; xd] chain and the penalty chain both add into one weight's gradient)
// form one group, handled by the same threads one entry after the other, in the order the GEMMs were issued
struct SplitKReduceArgs { SplitKEntry e[kMaxSplitK]; int group_of[kMaxSplitK][4]; int group_len[kMaxSplitK]; };

__global__ void __launch_bounds__(256) splitk_reduce_kernel(SplitKReduceArgs a) {
    const int grp = blockIdx.y, len = a.group_len[grp];
    const long mn = a.e[a.group_of[grp][0]].mn;
    float* C = a.e[a.group_of[grp][0]].C;
    // (four elements per thread where the tensor allows it: 16-byte reads of the slabs; the arithmetic per element is the same)
    bool vec = (mn & 3) == 0 && (reinterpret_cast<size_t>(C) & 15) == 0;
    for (int k = 0; k < len; ++k) vec = vec && (reinterpret_cast<size_t>(a.e[a.group_of[grp][k]].partial) & 15) == 0;
    if (vec) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        for (long i = 4 * (blockIdx.x * 256L + threadIdx.x); i < mn; i += 4 * gridDim.x * 256L) {
            f4 c = *reinterpret_cast<const f4*>(C + i);
            for (int k = 0; k < len; ++k) {
                const SplitKEntry& en = a.e[a.group_of[grp][k]];
                f4 s = *reinterpret_cast<const f4*>(en.partial + i);
                for (int z = 1; z < en.splits; ++z) {
                    const f4 t = *reinterpret_cast<const f4*>(en.partial + z * mn + i);
                    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                }
                c.x = __builtin_fmaf(en.alpha, s.x, c.x); c.y = __builtin_fmaf(en.alpha, s.y, c.y);
                c.z = __builtin_fmaf(en.alpha, s.z, c.z); c.w = __builtin_fmaf(en.alpha, s.w, c.w);
            }
            *reinterpret_cast<f4*>(C + i) = c;
        }
        return;
    }
    for (long i = blockIdx.x * 256L + threadIdx.x; i < mn; i += gridDim.x * 256L) {
        float c = C[i];
        for (int k = 0; k < len; ++k) {
            const SplitKEntry& en = a.e[a.group_of[grp][k]];
            float s = en.partial[i];
            for (int z = 1; z < en.splits; ++z) s += en.partial[z * mn + i];
            c = __builtin_fmaf(en.alpha, s, c);
        }
        C[i] = c;
    }
}


// ---------------------------------------------------------------------------------
// The bf16 GEMM for the shapes the critic spends its time in (batch x 512 x 512 and the like: K >= 64).  Same tile (64 x 64 per workgroup, 2 x 2 waves of one 32 x 32 MFMA tile), same k order of the
// accumulation -- what changes is how the operands arrive: the general kernel above issues one K tile of scalar loads per
// iteration and waits for it, and at 16 iterations of about a microsecond of load latency each a 0.5 GFLOP GEMM took
// 18 us.  Here every thread fetches 16-byte vectors along the contiguous index, THREE K tiles of 64 ahead (24 loads in
// flight per thread), LDS is double buffered and an iteration has one barrier.
// ---------------------------------------------------------------------------------
typedef float gf4 __attribute__((ext_vector_type(4)));
#ifndef SSN_GEMM_NO_XCD_DEAL
#define SSN_GEMM_NO_XCD_DEAL 0     // 1: tiles in launch order (A/B builds)
#endif
// operand modes: 0 = k contiguous (16-byte vectors along k), 1 = m / n contiguous (vectors along m / n), 2 = any strides
// (scalar loads, k fastest: the 11-wide input side of the first layer, whose extents are not multiples of four)
constexpr int GP_BM = 64, GP_BN = 64, GP_BK = 64, GP_LDK = GP_BK + 8;
// one 64 x 64 tile: workgroup (bx, by) of a gx x gy grid, K slice bz of nz
template <int AMODE, int BMODE>
__device__ __forceinline__ void gemm_pipe_tile(const GemmArgs& g, int bx, int by, int bz, int gx, int gy, int nz,
                                               unsigned short (&As)[2][GP_BM][GP_LDK], unsigned short (&Bs)[2][GP_BN][GP_LDK]) {
    constexpr int BM = GP_BM, BN = GP_BN, BK = GP_BK, LDK = GP_LDK, NST = 3;
    // Workgroups go to the 8 XCDs round robin by their linear id, and each XCD has an L2 of its own.  With x (the column tile)
    // running fastest and 8 column tiles, XCD k would compute column k of EVERY row tile: all of A through every L2.  The tiles
    // are dealt so that an XCD gets a contiguous run of them (whole row tiles: 1 / 8 of A, all of B, once).  Same tiles, same
    // arithmetic: the results do not change by a bit.
    {
        const int tiles = gx * gy;
        if ((tiles & 7) == 0 && !SSN_GEMM_NO_XCD_DEAL) {
            const int lin = by * gx + bx, t = (lin & 7) * (tiles >> 3) + (lin >> 3);
            by = t / gx; bx = t - by * gx;
        }
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int kbeg = bz * g.kchunk;
    const int kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
    const int nt = (kend - kbeg + BK - 1) / BK;
    const int q = tid & 15, p = tid >> 4;             // vector index along the contiguous extent, line within a group of 16
    gf4 ra[NST][4], rb[NST][4];
    const gf4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // one operand tile (64 lines x 64 k) into 4 vectors per thread; X(line, k) = base[line * sl + k * sk], `lines` valid lines
    auto fetch = [&](auto MODE, const float* base, long sl, long sk, int l0, int lines, int k0, gf4 (&r)[4]) {
        constexpr int mode = decltype(MODE)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (mode == 0) {                // line = p + 16 i, vector = 4 consecutive k
                const int l = l0 + p + 16 * i, k = k0 + 4 * q;
                r[i] = (l < lines && k < kend) ? *reinterpret_cast<const gf4*>(base + (long)l * sl + k) : zero4;
            } else if constexpr (mode == 1) {         // k = p + 16 i, vector = 4 consecutive lines
                const int k = k0 + p + 16 * i, l = l0 + 4 * q;
                r[i] = (l < lines && k < kend) ? *reinterpret_cast<const gf4*>(base + (long)k * sk + l) : zero4;
            } else {                                  // the element layout of mode 0, one load each
                const int l = l0 + p + 16 * i;
                gf4 v = zero4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + 4 * q + e;
                    if (l < lines && k < kend) v[e] = base[(long)l * sl + (long)k * sk];
                }
                r[i] = v;
            }
        }
    };
    auto stash = [&](auto MODE, unsigned short (&T)[BM][LDK], const gf4 (&r)[4]) {
        constexpr int mode = decltype(MODE)::value;
        typedef unsigned short us4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const gf4 a = r[i];
            if constexpr (mode != 1) {
                *reinterpret_cast<us4*>(&T[p + 16 * i][4 * q]) = (us4){to_bf16(a.x), to_bf16(a.y), to_bf16(a.z), to_bf16(a.w)};
            } else {
                const int k = p + 16 * i;
                T[4 * q][k] = to_bf16(a.x); T[4 * q + 1][k] = to_bf16(a.y);
                T[4 * q + 2][k] = to_bf16(a.z); T[4 * q + 3][k] = to_bf16(a.w);
            }
        }
    };
    constexpr std::integral_constant<int, AMODE> AM{};
    constexpr std::integral_constant<int, BMODE> BMD{};
    auto load = [&](auto ST, int k0) {
        constexpr int st = decltype(ST)::value;
        fetch(AM, g.A, g.sam, g.sak, m0, g.M, k0, ra[st]);
        fetch(BMD, g.B, g.sbn, g.sbk, n0, g.N, k0, rb[st]);
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // iteration t: registers of stage t % 3 -> LDS buffer t & 1, barrier, refill the stage with tile t + 3, MFMAs.  One
    // barrier is enough: a wave stores into buffer (t + 1) & 1 only behind barrier t, which every wave reaches after its
    // MFMA reads of iteration t - 1 from that buffer.
    auto step = [&](auto ST, int t) {
        constexpr int st = decltype(ST)::value;
        const int buf = t & 1;
        stash(AM, As[buf], ra[st]);
        stash(BMD, Bs[buf], rb[st]);
        __syncthreads();
        if (t + NST < nt) load(ST, kbeg + (t + NST) * BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(&As[buf][wr * 32 + (lane & 31)][kk + 8 * (lane >> 5)]);
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(&Bs[buf][wc * 32 + (lane & 31)][kk + 8 * (lane >> 5)]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
        }
    };
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    constexpr std::integral_constant<int, 2> S2{};
    if (nt > 0) load(S0, kbeg);
    if (nt > 1) load(S1, kbeg + BK);
    if (nt > 2) load(S2, kbeg + 2 * BK);
    for (int t = 0; t < nt; t += NST) {
        step(S0, t);
        if (t + 1 < nt) step(S1, t + 1);
        if (t + 2 < nt) step(S2, t + 2);
    }
    gemm_epilogue(g, acc, m0 + wr * 32, n0 + wc * 32, lane, bz, nz);
}
template <int AMODE, int BMODE>
__global__ void __launch_bounds__(256) gemm_bf16_pipe_kernel(GemmArgs g) {
    __shared__ __align__(16) unsigned short As[2][GP_BM][GP_LDK];
    __shared__ __align__(16) unsigned short Bs[2][GP_BN][GP_LDK];
    gemm_pipe_tile<AMODE, BMODE>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, gridDim.z, As, Bs);
}
// SEVERAL GEMMs in one launch (the weight-gradient GEMMs of a critic update on the row-block path: nothing orders them among
// themselves, every one writes slabs or a C of its own): workgroup id -> (problem, tile, K slice), the tile code is the one above
constexpr int kMaxBatch = 8;
struct GemmBatchArgs { GemmArgs g[kMaxBatch]; int first[kMaxBatch + 1]; int gx[kMaxBatch], gy[kMaxBatch], nz[kMaxBatch], am[kMaxBatch], bm[kMaxBatch]; int n; };
__global__ void __launch_bounds__(256) gemm_bf16_pipe_batch_kernel(GemmBatchArgs b) {
    __shared__ __align__(16) unsigned short As[2][GP_BM][GP_LDK];
    __shared__ __align__(16) unsigned short Bs[2][GP_BN][GP_LDK];
    int p = 0;
    while (p + 1 < b.n && (int)blockIdx.x >= b.first[p + 1]) ++p;
    const int lin = blockIdx.x - b.first[p], gx = b.gx[p], gy = b.gy[p], nz = b.nz[p];
    const int bz = lin / (gx * gy), rem = lin - bz * gx * gy, by = rem / gx, bx = rem - by * gx;
    const GemmArgs& g = b.g[p];
    const int mode = b.am[p] * 3 + b.bm[p];
    switch (mode) {                         // (the operand modes the weight gradients meet; anything else: the general form)
        case 4: gemm_pipe_tile<1, 1>(g, bx, by, bz, gx, gy, nz, As, Bs); break;
        case 7: gemm_pipe_tile<2, 1>(g, bx, by, bz, gx, gy, nz, As, Bs); break;
        case 5: gemm_pipe_tile<1, 2>(g, bx, by, bz, gx, gy, nz, As, Bs); break;
        default: gemm_pipe_tile<2, 2>(g, bx, by, bz, gx, gy, nz, As, Bs); break;
    }
}
// how gemm_bf16_pipe_kernel fetches an operand X(line, k) = base[line * sl + k * sk] with `lines` lines and K columns:
// 16-byte vectors need an aligned base, a unit stride along the vector and the other stride and the extent in fours
static int gemm_pipe_mode(const float* base, long sl, long sk, int lines, int K) {
    const bool al = (reinterpret_cast<size_t>(base) & 15) == 0;
    if (al && sk == 1 && K % 4 == 0 && sl % 4 == 0) return 0;
    if (al && sl == 1 && lines % 4 == 0 && sk % 4 == 0) return 1;
    return 2;
}
static bool gemm_pipe_ok(const GemmArgs& g) {
    // SSN_GEMM_PIPE=0: every GEMM through the general kernel (A/B runs and tests/test_critic_gpu.py; read once)
    static const bool on = [] { const char* e = getenv("SSN_GEMM_PIPE"); return !(e && e[0] == '0'); }();
    return on && g.K >= 64;
}
template <int AMODE>
static void gemm_pipe_launch_b(int bmode, dim3 grid, hipStream_t st, const GemmArgs& g) {
    if (bmode == 0)      hipLaunchKernelGGL((gemm_bf16_pipe_kernel<AMODE, 0>), grid, dim3(256), 0, st, g);
    else if (bmode == 1) hipLaunchKernelGGL((gemm_bf16_pipe_kernel<AMODE, 1>), grid, dim3(256), 0, st, g);
    else                 hipLaunchKernelGGL((gemm_bf16_pipe_kernel<AMODE, 2>), grid, dim3(256), 0, st, g);
}

static int choose_splits(int M, int N, int K) {
    const int tiles = ((N + 63) / 64) * ((M + 63) / 64);
    if (K < 512 || tiles >= 256) return 1;
    int splits = (512 + tiles - 1) / tiles;
    const int maxs = K / 128;
    if (splits > maxs) splits = maxs;
    return splits < 1 ? 1 : splits;
}
// upper bound of the scratch one critic update needs: two weight-gradient GEMMs per layer and per output vector
size_t critic_splitk_scratch_floats(const int* dims, int nlayers, int rows) {
    size_t n = 0;
    for (int l = 0; l < nlayers; ++l) n += 2 * (size_t)choose_splits(dims[l], dims[l + 1], rows) * dims[l] * dims[l + 1];
    n += 2 * (size_t)choose_splits(dims[nlayers], 1, rows) * dims[nlayers];
    return n + 64;
}
void critic_splitk_begin(float* scratch, size_t cap) { tl_plan.scratch = scratch; tl_plan.cap = cap; tl_plan.used = 0; tl_plan.n = 0; }
hipError_t critic_splitk_flush(hipStream_t st) {
    hipError_t e = hipSuccess;
    if (tl_plan.n > 0) {
        SplitKReduceArgs a;
        int groups = 0;
        long most = 0;
        for (int i = 0; i < tl_plan.n; ++i) {
            a.e[i] = tl_plan.e[i];
            if (a.e[i].mn > most) most = a.e[i].mn;
            int g = 0;
            while (g < groups && a.e[a.group_of[g][0]].C != a.e[i].C) ++g;
            if (g == groups) a.group_len[groups++] = 0;
            a.group_of[g][a.group_len[g]++] = i;
        }
        long bx = (most + 255) / 256;
        if (bx > 256) bx = 256;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)bx, groups), dim3(256), 0, st, a);
        e = hipGetLastError();
    }
    tl_plan.scratch = nullptr; tl_plan.cap = tl_plan.used = 0; tl_plan.n = 0;
    return e;
}

// GEMMs collected for one launch (gemm_bf16_pipe_batch_kernel): between gemm_batch_begin and gemm_batch_flush every GEMM the
// pipelined kernel would take is queued here instead of launched -- same tiles, same K slices, same slabs
struct GemmBatch { bool on; int blocks; GemmBatchArgs b; };
static thread_local GemmBatch tl_batch{false, 0, {}};
static void gemm_batch_begin() { tl_batch.on = true; tl_batch.blocks = 0; tl_batch.b.n = 0; tl_batch.b.first[0] = 0; }
static hipError_t gemm_batch_flush(hipStream_t st) {
    tl_batch.on = false;
    if (tl_batch.b.n == 0) return hipSuccess;
    hipLaunchKernelGGL(gemm_bf16_pipe_batch_kernel, dim3(tl_batch.blocks), dim3(256), 0, st, tl_batch.b);
    tl_batch.b.n = 0;
    return hipGetLastError();
}
static hipError_t gemm(GemmArgs g, bool bf16, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    int splits = 1;
    bool slab = false;
    if (g.epilogue == EPI_PLAIN && g.beta == 1.f && g.ldc == g.N && tl_plan.scratch && tl_plan.n < kMaxSplitK) {
        splits = choose_splits(g.M, g.N, g.K);
        int same = 0;
        for (int i = 0; i < tl_plan.n; ++i) same += tl_plan.e[i].C == g.C;
        // no room, or a 5th split GEMM into one C: run it unsplit.  An unsplit GEMM adds into C directly, BEFORE the
        // slabs of earlier split ones are added at the flush -- a fixed order either way.
        if (same >= 4 || tl_plan.used + (size_t)splits * g.M * g.N > tl_plan.cap) splits = 1;
        // (inside a batch two GEMMs may accumulate into one C -- the two halves' gradients of a weight: only GEMMs that write
        // slabs of their own are batched; an unsplit one is launched behind everything queued so far, as it always was)
        slab = tl_batch.on && splits > 1;
    }
    const bool pipe = bf16 && gemm_pipe_ok(g);
    if (tl_batch.on && !(pipe && slab && tl_batch.b.n < kMaxBatch)) {      // not for the batch: everything queued so far goes first
        hipError_t e = gemm_batch_flush(st);
        if (e != hipSuccess) return e;
        gemm_batch_begin();
        slab = false;
    }
    const int bk = pipe ? 64 : (bf16 ? 32 : 16);
    g.kchunk = ((g.K + splits - 1) / splits + bk - 1) / bk * bk;
    splits = (g.K + g.kchunk - 1) / g.kchunk;
    g.partial = nullptr;
    slab = slab && splits > 1;
    if (splits > 1) {
        g.partial = tl_plan.scratch + tl_plan.used;
        tl_plan.e[tl_plan.n++] = SplitKEntry{g.C, g.partial, (long)g.M * g.N, splits, g.alpha};
        tl_plan.used += (size_t)splits * g.M * g.N;
    }
    dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, splits);
    if (pipe && tl_batch.on && slab) {
        GemmBatchArgs& b = tl_batch.b;
        const int i = b.n++;
        b.g[i] = g; b.gx[i] = grid.x; b.gy[i] = grid.y; b.nz[i] = grid.z;
        b.am[i] = gemm_pipe_mode(g.A, g.sam, g.sak, g.M, g.K); b.bm[i] = gemm_pipe_mode(g.B, g.sbn, g.sbk, g.N, g.K);
        tl_batch.blocks += (int)(grid.x * grid.y * grid.z);
        b.first[i + 1] = tl_batch.blocks;
        return hipSuccess;
    }
    if (pipe) {
        const int am = gemm_pipe_mode(g.A, g.sam, g.sak, g.M, g.K), bm = gemm_pipe_mode(g.B, g.sbn, g.sbk, g.N, g.K);
        if (am == 0)      gemm_pipe_launch_b<0>(bm, grid, st, g);
        else if (am == 1) gemm_pipe_launch_b<1>(bm, grid, st, g);
        else              gemm_pipe_launch_b<2>(bm, grid, st, g);
    } else if (bf16) hipLaunchKernelGGL((gemm_mfma_kernel<true>), grid, dim3(256), 0, st, g);
    else      hipLaunchKernelGGL((gemm_mfma_kernel<false>), grid, dim3(256), 0, st, g);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// small elementwise / reduction kernels of the critic
// ---------------------------------------------------------------------------------
// h0[b] = [x[b][0:nx], c0, |c1|, c2]   (cwgan.py:164-170; hide_cell_type zeroes c2, 178-187); nc = 0: the unconditional critic
// of networks/wgan.py:66-97, h0[b] = x[b] (cond is not read)
__global__ void __launch_bounds__(256) critic_input_kernel(const float* __restrict__ x, const float* __restrict__ cond,
                                                           float* __restrict__ h0, int batch, int nx, int hide_cell_type, int nc) {
    const int n0 = nx + nc;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < (long)batch * n0; e += gridDim.x * 256L) {
        const int b = (int)(e / n0), j = (int)(e % n0);
        float v;
        if (j < nx) v = x[(long)b * nx + j];
        else if (j == nx) v = cond[b * 3 + 0];
        else if (j == nx + 1) v = fabsf(cond[b * 3 + 1]);
        else v = hide_cell_type ? 0.f : cond[b * 3 + 2];
        h0[e] = v;
    }
}

// The inputs of a whole critic step in ONE launch (single-process loop): the penalty points xp = eps xd + (1 - eps) xg
// (cwgan.py:476-481; the bits of interpolate_kernel) and the three input blocks critic_input_kernel would build from xg, xd and
// xp with the one condition array they share: h0 = [rows of xg; rows of xd] (2 n rows), hp0 = rows of xp.
__global__ void __launch_bounds__(256) critic_step_inputs_kernel(const float* __restrict__ xg, const float* __restrict__ xd,
                                                                 const float* __restrict__ cond, const float* __restrict__ eps,
                                                                 float* __restrict__ xp, float* __restrict__ h0,
                                                                 float* __restrict__ hp0, int n, int nx, int hide_cell_type, int nc) {
    const int n0 = nx + nc;
    const long total = (long)n * n0;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += gridDim.x * 256L) {
        const int b = (int)(e / n0), j = (int)(e % n0);
        if (j < nx) {
            const long i = (long)b * nx + j;
            const float g = xg[i], d = xd[i], w = eps[b];
            const float p = __fadd_rn(__fmul_rn(w, d), __fmul_rn(1.f - w, g));
            xp[i] = p;
            h0[e] = g; h0[total + e] = d; hp0[e] = p;
        } else {
            float v;
            if (j == nx) v = cond[b * 3 + 0];
            else if (j == nx + 1) v = fabsf(cond[b * 3 + 1]);
            else v = hide_cell_type ? 0.f : cond[b * 3 + 2];
            h0[e] = v; h0[total + e] = v; hp0[e] = v;
        }
    }
}

// v_L[b][k] = (h_L[b][k] > 0 ? 1 : leak) * w_out[k] * up[b]   (up = per-sample upstream of D; nullptr -> 1)
__global__ void __launch_bounds__(256) critic_outgrad_kernel(const float* __restrict__ hL, const float* __restrict__ wout,
                                                             const float* __restrict__ up, float* __restrict__ vL,
                                                             int batch, int nL, float leak) {
    for (long e = blockIdx.x * 256L + threadIdx.x; e < (long)batch * nL; e += gridDim.x * 256L) {
        const int b = (int)(e / nL), k = (int)(e % nL);
        const float v = wout[k] * (up ? up[b] : 1.f);
        vL[e] = (hL[e] > 0.f) ? v : leak * v;
    }
}

// column sums: out[n] = sum_b X[b][n] + beta * out[n]    (bias gradients).  One workgroup per 64 columns: 16 lanes x
// float4 per row, 64 row slots, fixed-order tree in LDS -- deterministic.  Accumulation in fp64: the bias gradient of
// a unit that is active on equally many generated and data rows is a sum of +c and -c terms, exactly zero in the
// reference's arithmetic; fp32 partial sums (3c, 5c, ...) round, leave ~1e-8 of noise, and Adam (eps 1e-8) turns
// that noise into full-size steps of the bias.
// (axpy_to: instead of `out`, the sums s go to axpy_to[n] += axpy_a * (float)s -- a column sum into a scratch vector followed by
// axpy_kernel, in one go)
__device__ __forceinline__ void colsum_block(const float* __restrict__ X, float* __restrict__ out, int batch, int n, float beta,
                                             int bx, double (&red)[64][64], double (&red2)[16][64], float* axpy_to, float axpy_a) {
    const int cq = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const int col = bx * 64 + 4 * cq;
    double s[4] = {0., 0., 0., 0.};
    if (col + 3 < n && (n & 3) == 0) {
#pragma unroll 4
        for (int b = slot; b < batch; b += 64) {
            const float4 v = *reinterpret_cast<const float4*>(X + (long)b * n + col);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        }
    } else {
        for (int b = slot; b < batch; b += 64)
#pragma unroll
            for (int e = 0; e < 4; ++e) if (col + e < n) s[e] += X[(long)b * n + col + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[slot][4 * cq + e] = s[e];
    __syncthreads();
    {
        const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
        red2[g][c] = (red[4 * g][c] + red[4 * g + 1][c]) + (red[4 * g + 2][c] + red[4 * g + 3][c]);
    }
    __syncthreads();
    if (threadIdx.x < 64 && bx * 64 + threadIdx.x < n) {
        double t = 0.;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red2[g][threadIdx.x];
        if (axpy_to) {
            float* y = axpy_to + bx * 64 + threadIdx.x;
            *y = __builtin_fmaf(axpy_a, (float)(t + 0.), *y);
        } else {
            float* o = out + bx * 64 + threadIdx.x;
            *o = (float)(t + (beta != 0.f ? (double)beta * (double)*o : 0.));
        }
    }
}
__global__ void __launch_bounds__(1024) colsum_kernel(const float* __restrict__ X, float* __restrict__ out, int batch, int n,
                                                      float beta) {
    __shared__ double red[64][64];
    __shared__ double red2[16][64];
    colsum_block(X, out, batch, n, beta, blockIdx.x, red, red2, nullptr, 0.f);
}
// the column sums of a critic update on the row-block path in ONE launch (bias gradients of every layer, the w_out term of the
// penalty added straight into its gradient)
struct ColsumBatchArgs {
    const float* X[10]; float* out[10]; int batch[10], n[10], first[11]; float beta[10]; float* axpy_to[10]; float axpy_a[10]; int cnt;
    // one more workgroup (the last) for the update's statistics (critic_stats_block) when stats != nullptr
    const float* dvals; const float* dnorm; float* stats; int ng, nd, np; float lmd;
};
__global__ void __launch_bounds__(1024) colsum_batch_kernel(ColsumBatchArgs a) {
    __shared__ double red[64][64];
    __shared__ double red2[16][64];
    if ((int)blockIdx.x == a.first[a.cnt]) {
        critic_stats_block(a.dvals, a.dnorm, a.stats, a.ng, a.nd, a.np, a.lmd, *reinterpret_cast<float (*)[3][256]>(&red[0][0]));
        return;
    }
    int p = 0;
    while (p + 1 < a.cnt && (int)blockIdx.x >= a.first[p + 1]) ++p;
    colsum_block(a.X[p], a.out[p], a.batch[p], a.n[p], a.beta[p], blockIdx.x - a.first[p], red, red2, a.axpy_to[p], a.axpy_a[p]);
}

// gradient-penalty head: per sample norm of g[:, :nx]; writes ghat = 2 (norm-1)/norm * g_x / batch (zero
// beyond nx) and accumulates sum_b (norm_b - 1)^2 into pen[0] (one workgroup; fixed order -> deterministic).
__global__ void __launch_bounds__(256) gp_head_kernel(const float* __restrict__ g, float* __restrict__ ghat,
                                                      float* __restrict__ pen, int batch, int n0, int nx) {
    __shared__ float red[256];
    float local = 0.f;
    for (int b = threadIdx.x; b < batch; b += 256) {
        float coef;
        const float d = critic_gp_row(g + (long)b * n0, nx, batch, coef);
        local = __builtin_fmaf(d, d, local);
        for (int j = 0; j < n0; ++j) ghat[(long)b * n0 + j] = (j < nx) ? coef * g[(long)b * n0 + j] : 0.f;
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) pen[0] = red[0] / (float)batch;
}

// out[0] = mean(d[0:ng]), out[1] = mean(d[ng:ng+nd])   (one workgroup)
__global__ void __launch_bounds__(256) two_means_kernel(const float* __restrict__ d, float* __restrict__ out, int ng, int nd) {
    __shared__ float red[2][256];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < ng; i += 256) a += d[i];
    for (int i = threadIdx.x; i < nd; i += 256) b += d[ng + i];
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) { red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = ng ? red[0][0] / ng : 0.f; out[1] = nd ? red[1][0] / nd : 0.f; }
}

__global__ void __launch_bounds__(256) fill_updown_kernel(float* up, int ng, int nd) {
    // upstream of mean D(xg) - mean D(xd) for the concatenated [xg; xd] batch
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ng + nd; i += gridDim.x * 256)
        up[i] = critic_updown(i, ng, nd);
}

// stats[3] = mean D(xg) - mean D(xd) + lmd * penalty
__global__ void loss_combine_kernel(float* stats, float lmd) { stats[3] = critic_loss_value(stats[0], stats[1], stats[2], lmd); }

// gx[b][j] = s * v0[b][j], j < nx   (tuning-curve part of the input gradient)
__global__ void __launch_bounds__(256) gather_scale_kernel(const float* __restrict__ v0, float* __restrict__ gx, int batch,
                                                           int n0, int nx, float s) {
    for (long e = blockIdx.x * 256L + threadIdx.x; e < (long)batch * nx; e += gridDim.x * 256L) {
        const int b = (int)(e / nx), j = (int)(e % nx);
        gx[e] = s * v0[(long)b * n0 + j];
    }
}

__global__ void __launch_bounds__(256) axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long n) {
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += gridDim.x * 256L) y[e] = __builtin_fmaf(a, x[e], y[e]);
}

// ---------------------------------------------------------------------------------
// critic passes
// ---------------------------------------------------------------------------------
struct CriticNet {
    int nlayers;           // hidden layers L
    int dims[10];          // n_0 .. n_L
    const float* W[9];
    const float* b[9];
    const float* wout;     // [n_L]
    long nparams;
    float leak;            // hidden nonlinearity: x > 0 ? x : leak * x (0 = rectify; the sign of h_l is the sign of its pre-activation)
};

static bool parse_net(const float* params, const int* dims, int nlayers, CriticNet& net, float leak = 0.f) {
    if (nlayers < 0 || nlayers > 8 || !(leak >= 0.f) || leak > 1.f) return false;
    net.nlayers = nlayers;
    net.leak = leak;
    long off = 0;
    for (int l = 0; l <= nlayers; ++l) net.dims[l] = dims[l];
    for (int l = 0; l < nlayers; ++l) {
        net.W[l] = params + off; off += (long)dims[l] * dims[l + 1];
        net.b[l] = params + off; off += dims[l + 1];
    }
    net.wout = params + off; off += dims[nlayers];
    net.nparams = off;
    return true;
}

static int blocks_for(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }

// activations h[l] (l = 0..L) for `batch` rows live at act + offsets; returns D in dout[batch]
static hipError_t critic_forward_pass(const CriticNet& net, float* const* h, float* dout, int batch, bool bf16, hipStream_t st) {
    hipError_t e;
    for (int l = 0; l < net.nlayers; ++l) {
        GemmArgs g{};
        g.A = h[l]; g.sam = net.dims[l]; g.sak = 1;
        g.B = net.W[l]; g.sbk = net.dims[l + 1]; g.sbn = 1;
        g.C = h[l + 1]; g.ldc = net.dims[l + 1];
        g.M = batch; g.N = net.dims[l + 1]; g.K = net.dims[l];
        g.bias = net.b[l]; g.epilogue = EPI_BIAS_RELU; g.leak = net.leak;
        if ((e = gemm(g, bf16, st)) != hipSuccess) return e;
    }
    GemmArgs g{};
    const int L = net.nlayers;
    g.A = h[L]; g.sam = net.dims[L]; g.sak = 1;
    g.B = net.wout; g.sbk = 1; g.sbn = 1;
    g.C = dout; g.ldc = 1; g.M = batch; g.N = 1; g.K = net.dims[L];
    g.alpha = 1.f; g.beta = 0.f; g.epilogue = EPI_PLAIN;
    return gemm(g, bf16, st);
}

// Backward chain with masks from h[]: v[L] given; computes v[l-1] = m_{l-1} * (v[l] W_l^T) down to
// v[0] = v[1] W_1^T (no mask on the input).  If grads != nullptr also accumulates parameter gradients
// of "sum_b up_b D_b" style losses: dW_l += h_{l-1}^T v_l,  db_l += colsum(v_l).
struct CriticFork;
static hipError_t critic_hand_over(CriticFork* fk, hipStream_t from);
static hipStream_t critic_grad_stream(CriticFork* fk, hipStream_t st);
static hipError_t critic_backward_chain(const CriticNet& net, float* const* h, float* const* v, int batch, float* grads,
                                        bool want_input_grad, bool bf16, hipStream_t st, CriticFork* fk = nullptr) {
    hipError_t e;
    const hipStream_t sg = critic_grad_stream(fk, st);        // weight gradients and bias sums (st itself without a fork)
    long off_end = net.nparams - net.dims[net.nlayers];
    long off = off_end;
    for (int l = net.nlayers - 1; l >= 0; --l) {
        const int nin = net.dims[l], nout = net.dims[l + 1];
        off -= nout;                    // bias of layer l
        const long off_b = off;
        off -= (long)nin * nout;        // weight of layer l
        const long off_w = off;
        if (grads) {
            GemmArgs g{};               // dW_l[i][k] += scale * sum_b h_l[b][i] v_{l+1}[b][k]
            g.A = h[l]; g.sam = 1; g.sak = nin;            // op(A)(i, b) = h[b][i]
            g.B = v[l + 1]; g.sbk = nout; g.sbn = 1;        // op(B)(b, k)
            g.C = grads + off_w; g.ldc = nout; g.M = nin; g.N = nout; g.K = batch;
            g.alpha = 1.f; g.beta = 1.f; g.epilogue = EPI_PLAIN;
            if ((e = critic_hand_over(fk, st)) != hipSuccess) return e;          // v[l + 1] is queued on st
            if ((e = gemm(g, bf16, sg)) != hipSuccess) return e;
            hipLaunchKernelGGL(colsum_kernel, dim3((nout + 63) / 64), dim3(1024), 0, sg, v[l + 1], grads + off_b, batch, nout, 1.f);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        if (l > 0 || want_input_grad) {
            GemmArgs g{};               // v_l = (v_{l+1} W_l^T) [* mask_l]
            g.A = v[l + 1]; g.sam = nout; g.sak = 1;
            g.B = net.W[l]; g.sbk = 1; g.sbn = nout;        // op(B)(k, i) = W[i][k]
            g.C = v[l]; g.ldc = nin; g.M = batch; g.N = nin; g.K = nout;
            if (l > 0) { g.epilogue = EPI_MASK; g.mask = h[l]; g.ldm = nin; g.leak = net.leak; }
            else { g.alpha = 1.f; g.beta = 0.f; g.epilogue = EPI_PLAIN; }
            if ((e = gemm(g, bf16, st)) != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

size_t critic_workspace_floats(const int* dims, int nlayers, int batch_gd, int batch_p) {
    // (1) h_0..h_L and v_0..v_L for the [xg; xd] rows + upstream; (2) h, v, e for the xp rows + D(xp); + scratch
    long per_row = 0, maxd = 0;
    for (int l = 0; l <= nlayers; ++l) { per_row += dims[l]; if (dims[l] > maxd) maxd = dims[l]; }
    return (size_t)(2L * batch_gd * per_row + batch_gd + 3L * batch_p * per_row + batch_p + maxd + 64) +
           critic_splitk_scratch_floats(dims, nlayers, batch_gd + batch_p) + critic_rows_workspace_floats(dims, nlayers, batch_p);
}

// Wide plain critics on bf16 operands: D (mode 0) or D and the input gradient (mode 1) of `batch` rows whose input block h0 is
// built, in two launches -- weights to B fragments, the row-block kernel (ssn_critic_rows.hip).  `free_ws`: workspace behind h0.
static hipError_t critic_rows_eval(const float* params, const int* dims, int nlayers, float* h0, int batch, float* dvals, int mode,
                                   float* gx, float scale, int nx, float leak, float* free_ws, hipStream_t st) {
    RowsArgs ra{};
    ra.L = nlayers; ra.leak = leak; ra.mode = mode; ra.nx = nx; ra.dvals = dvals; ra.gx = gx; ra.scale = scale;
    for (int l = 0; l <= nlayers; ++l) ra.dims[l] = dims[l];
    if (mode == 0) { ra.h[0] = h0; ra.ng = batch; } else { ra.hp[0] = h0; ra.np = batch; }
    hipError_t e = critic_rows_pack(params, dims, nlayers, free_ws, ra, nullptr, 0, st);
    return e != hipSuccess ? e : critic_rows_launch(ra, st);
}

// D values for a batch (inference / accuracy): out[batch]
hipError_t critic_forward(const float* params, const int* dims, int nlayers, const float* x, const float* cond, int batch,
                          int hide_cell_type, float* out, float* ws, bool bf16, hipStream_t st, float leak) {
    CriticNet net;
    if (!parse_net(params, dims, nlayers, net, leak)) return hipErrorInvalidValue;
    const int nc = cond ? 3 : 0, nx = dims[0] - nc;          // (no condition columns: the unconditional critic)
    float* h[10];
    float* p = ws;
    for (int l = 0; l <= nlayers; ++l) { h[l] = p; p += (long)batch * dims[l]; }
    hipLaunchKernelGGL(critic_input_kernel, dim3(blocks_for((long)batch * dims[0])), dim3(256), 0, st, x, cond, h[0], batch, nx, hide_cell_type, nc);
    if (bf16 && batch > 0 && critic_rows_supported(dims, nlayers))
        return critic_rows_eval(params, dims, nlayers, h[0], batch, out, 0, nullptr, 0.f, nx, leak, h[1], st);
    return critic_forward_pass(net, h, out, batch, bf16, st);
}

// D values of TWO batches in one pass over the stacked rows [xa; xb] (the accuracy mean D(xg) - mean D(xd) after every critic
// update, cwgan.py:505-507): every output row depends on its own input row alone and a forward GEMM is never split over K,
// so the values are those of two separate forwards, bit for bit, for one chain of launches instead of two.
// out[na + nb]; ws as critic_forward with batch = na + nb (<= 2 max(na, nb): what the callers reserve).
hipError_t critic_forward2(const float* params, const int* dims, int nlayers, const float* xa, const float* ca, int na,
                           const float* xb, const float* cb, int nb, int hide_cell_type, float* out, float* ws, bool bf16,
                           hipStream_t st, float leak, bool inputs_ready) {
    CriticNet net;
    if (!parse_net(params, dims, nlayers, net, leak)) return hipErrorInvalidValue;
    if ((ca == nullptr) != (cb == nullptr) && na > 0 && nb > 0) return hipErrorInvalidValue;
    const int nc = (ca || cb) ? 3 : 0, nx = dims[0] - nc, batch = na + nb;
    if (batch == 0) return hipSuccess;
    float* h[10];
    float* p = ws;
    for (int l = 0; l <= nlayers; ++l) { h[l] = p; p += (long)batch * dims[l]; }
    if (na > 0 && !inputs_ready) hipLaunchKernelGGL(critic_input_kernel, dim3(blocks_for((long)na * dims[0])), dim3(256), 0, st, xa, ca, h[0], na, nx, hide_cell_type, nc);
    if (nb > 0 && !inputs_ready) hipLaunchKernelGGL(critic_input_kernel, dim3(blocks_for((long)nb * dims[0])), dim3(256), 0, st, xb, cb, h[0] + (long)na * dims[0], nb, nx, hide_cell_type, nc);
    if (bf16 && critic_rows_supported(dims, nlayers))
        return critic_rows_eval(params, dims, nlayers, h[0], batch, out, 0, nullptr, 0.f, nx, leak, h[1], st);
    return critic_forward_pass(net, h, out, batch, bf16, st);
}

// The two halves of a critic update -- the Wasserstein term on [xg; xd] and the gradient penalty on xp -- are independent
// chains of ~20 small launches each (forward, backward chain, weight gradients), latency-bound one after the other: the
// penalty half runs on a second stream beside the first (fork behind the inputs, join in front of the final sums), and the
// weight-gradient GEMMs and bias sums of BOTH halves -- which nothing in the chains waits for -- on a third, each behind an
// event that says its operands are queued.  The
// results do not change by a bit: every weight-gradient GEMM is split over K into slabs of its own, and the slabs are added
// by ONE kernel after the join, in the order the GEMMs were ISSUED (critic_splitk_flush) -- which is the host's order, not
// the order of execution.  A shape whose weight-gradient GEMMs would add into the gradient directly (choose_splits == 1:
// K < 512 or >= 256 output tiles) keeps the one stream.  SSN_CRITIC_PAR=0 switches the second stream off (A/B runs).
struct CriticFork {
    hipStream_t aux, gs;               // penalty half; weight-gradient GEMMs and bias column sums of both halves
    hipEvent_t fork, join, join_g;
    hipEvent_t ready[24]; unsigned nready;  // "operands of the next gradient GEMM are there", one per hand-over, reused every call
    // the gradient stream takes its next launch behind everything `from` holds so far
    hipError_t hand_over(hipStream_t from) {
        hipEvent_t ev = ready[nready++ % 24];
        hipError_t e = hipEventRecord(ev, from);
        return e != hipSuccess ? e : hipStreamWaitEvent(gs, ev, 0);
    }
};
static CriticFork* critic_fork() {
    static thread_local CriticFork* per_dev[64] = {};   // per calling thread and device: the event ring is not shared
    static const bool on = [] { const char* e = getenv("SSN_CRITIC_PAR"); return !(e && e[0] == '0'); }();
    if (!on) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!per_dev[dev]) {
        CriticFork* f = new CriticFork{};
        bool ok = hipStreamCreateWithFlags(&f->aux, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&f->gs, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&f->fork, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&f->join, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&f->join_g, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 24 && ok; ++i) ok = hipEventCreateWithFlags(&f->ready[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) { delete f; return nullptr; }
        per_dev[dev] = f;
    }
    return per_dev[dev];
}

static hipError_t critic_hand_over(CriticFork* fk, hipStream_t from) { return fk ? fk->hand_over(from) : hipSuccess; }
static hipStream_t critic_grad_stream(CriticFork* fk, hipStream_t st) { return fk ? fk->gs : st; }

// Full critic loss + gradient.  stats[0..3] = mean D(xg), mean D(xd), penalty, loss.
hipError_t critic_loss_grad(const float* params, const int* dims, int nlayers, const float* xg, const float* cg,
                            const float* xd, const float* cd, const float* xp, const float* cp, int ng, int nd, int np,
                            float lmd, int hide_cell_type, float* grads, float* stats, float* dvals, float* ws, bool bf16,
                            hipStream_t st, float leak, const float* eps, float* xp_out) {
    // eps != nullptr: xp is not an input -- it is formed here, together with the three input blocks, in one launch (xp_out
    // receives it; ng = nd = np and ONE condition array for the three inputs: the single-process critic step)
    CriticNet net;
    if (!parse_net(params, dims, nlayers, net, leak)) return hipErrorInvalidValue;
    if (eps && (!xp_out || ng != nd || nd != np || cg != cd || cd != cp)) return hipErrorInvalidValue;
    hipError_t e;
    const int nc = (cg || cd || cp) ? 3 : 0;
    if (nc && ((ng && !cg) || (nd && !cd) || (np && !cp))) return hipErrorInvalidValue;   // conditions for all inputs or for none
    const int L = nlayers, nx = dims[0] - nc;
    const int bgd = ng + nd;
    // (the row-block path of wide plain critics zeroes the gradient in its weight-packing launch)
    const bool rows_path = bf16 && bgd > 0 && np > 0 && critic_rows_supported(dims, nlayers);
    if (!rows_path && (e = hipMemsetAsync(grads, 0, net.nparams * sizeof(float), st)) != hipSuccess) return e;
    float* p = ws;
    // ---------------- (1) mean D(xg) - mean D(xd) on the concatenated batch -------------------
    float *h[10], *v[10];
    for (int l = 0; l <= L; ++l) { h[l] = p; p += (long)bgd * dims[l]; }
    for (int l = 0; l <= L; ++l) { v[l] = p; p += (long)bgd * dims[l]; }
    float* up = p; p += bgd;
    float *hp[10], *vp[10], *ep[10];
    for (int l = 0; l <= L; ++l) { hp[l] = p; p += (long)np * dims[l]; }
    for (int l = 0; l <= L; ++l) { vp[l] = p; p += (long)np * dims[l]; }
    for (int l = 0; l <= L; ++l) { ep[l] = p; p += (long)np * dims[l]; }
    float* dp = p; p += np;
    float* tmp = p; p += dims[L];
    critic_splitk_begin(p, critic_splitk_scratch_floats(dims, nlayers, bgd + np));     // the rest of the workspace
    struct PlanScope { ~PlanScope() { critic_splitk_begin(nullptr, 0); } } plan_scope;    // closed on every return path
    // second stream for the penalty half, when every weight-gradient GEMM of both halves goes to slabs (see critic_fork)
    bool par = np > 0 && bgd > 0 && choose_splits(dims[L], 1, bgd) > 1;
    for (int l = 0; l < L && par; ++l) par = choose_splits(dims[l], dims[l + 1], bgd) > 1 && choose_splits(dims[l], dims[l + 1], np) > 1;
    CriticFork* const fk = par ? critic_fork() : nullptr;
    hipStream_t sp = st;                            // stream of the penalty half
    const hipStream_t sg = critic_grad_stream(fk, st);   // stream of the weight-gradient GEMMs and bias sums
    // Once the side streams carry work, EVERY way out of this function joins them to st first: an error return that left
    // them running would hand the caller back a workspace that is still being read and written.
    struct JoinScope {
        CriticFork* fk; hipStream_t st; bool forked = false, joined = false;
        hipError_t join() {
            if (!fk || !forked || joined) return hipSuccess;
            joined = true;
            hipError_t e = hipEventRecord(fk->join, fk->aux), e2;
            if (e == hipSuccess) e = hipStreamWaitEvent(st, fk->join, 0);
            e2 = hipEventRecord(fk->join_g, fk->gs);
            if (e2 == hipSuccess) e2 = hipStreamWaitEvent(st, fk->join_g, 0);
            return e != hipSuccess ? e : e2;
        }
        ~JoinScope() { (void)join(); }
    } join_scope{fk, st};
    // ---------------- wide plain critics on bf16 operands: the row-local chains of both halves in ONE launch (ssn_critic_rows.hip),
    // then the weight-gradient GEMMs, whose operands are all there, on the three streams.  Same bits as the chains below.
    if (rows_path) {
        if (eps) hipLaunchKernelGGL(critic_step_inputs_kernel, dim3(blocks_for((long)ng * dims[0])), dim3(256), 0, st, xg, xd, cg, eps, xp_out,
                                    h[0], hp[0], ng, nx, hide_cell_type, nc);
        else {
            hipLaunchKernelGGL(critic_input_kernel, dim3(blocks_for((long)ng * dims[0])), dim3(256), 0, st, xg, cg, h[0], ng, nx, hide_cell_type, nc);
            hipLaunchKernelGGL(critic_input_kernel, dim3(blocks_for((long)nd * dims[0])), dim3(256), 0, st, xd, cd, h[0] + (long)ng * dims[0], nd, nx, hide_cell_type, nc);
            hipLaunchKernelGGL(critic_input_kernel, dim3(blocks_for((long)np * dims[0])), dim3(256), 0, st, xp, cp, hp[0], np, nx, hide_cell_type, nc);
        }
        RowsArgs ra{};
        ra.L = L; ra.leak = leak; ra.mode = 2;
        for (int l = 0; l <= L; ++l) { ra.dims[l] = dims[l]; ra.h[l] = h[l]; ra.v[l] = v[l]; ra.hp[l] = hp[l]; ra.vp[l] = vp[l]; ra.ep[l] = ep[l]; }
        ra.up = up; ra.dvals = dvals; ra.ng = ng; ra.nd = nd; ra.np = np; ra.nx = nx;
        float* const rws = p + critic_splitk_scratch_floats(dims, nlayers, bgd + np);
        ra.dnorm = rws;
        if ((e = critic_rows_pack(params, dims, L, rws + np, ra, grads, net.nparams, st)) != hipSuccess) return e;
        if ((e = critic_rows_launch(ra, st)) != hipSuccess) return e;
        // every operand of the weight gradients is there: ONE launch for the GEMMs (split over K into slabs of their own, in the
        // order of the chains below), one for the bias sums, the w_out term of the penalty and the statistics, one for the slabs
        // (`par`: every weight-gradient GEMM of both halves is split over K -- the condition of the three streams below; any other
        // shape issues them one by one, in the order and with the direct additions of the chains below)
        if (par) gemm_batch_begin();
        struct BatchScope { ~BatchScope() { tl_batch.on = false; tl_batch.b.n = 0; } } batch_scope;
        ColsumBatchArgs cs{};
        auto add_colsum = [&](const float* X, float* out, int batch, int n, float beta, float* axpy_to, float axpy_a) {
            const int i = cs.cnt++;
            cs.X[i] = X; cs.out[i] = out; cs.batch[i] = batch; cs.n[i] = n; cs.beta[i] = beta; cs.axpy_to[i] = axpy_to; cs.axpy_a[i] = axpy_a;
            cs.first[i + 1] = cs.first[i] + (n + 63) / 64;
        };
        {
            GemmArgs g{};                               // d/dw_out = sum_b up_b h_L[b][:]
            g.A = h[L]; g.sam = 1; g.sak = dims[L];
            g.B = up; g.sbk = 1; g.sbn = 1;
            g.C = grads + (net.nparams - dims[L]); g.ldc = 1; g.M = dims[L]; g.N = 1; g.K = bgd;
            g.alpha = 1.f; g.beta = 1.f; g.epilogue = EPI_PLAIN;
            if ((e = gemm(g, bf16, st)) != hipSuccess) return e;
        }
        long off = net.nparams - dims[L];
        for (int l = L - 1; l >= 0; --l) {              // dW_l += h_l^T v_{l+1},  db_l += colsum(v_{l+1})
            const int nin = dims[l], nout = dims[l + 1];
            off -= nout;
            const long off_b = off;
            off -= (long)nin * nout;
            GemmArgs g{};
            g.A = h[l]; g.sam = 1; g.sak = nin;
            g.B = v[l + 1]; g.sbk = nout; g.sbn = 1;
            g.C = grads + off; g.ldc = nout; g.M = nin; g.N = nout; g.K = bgd;
            g.alpha = 1.f; g.beta = 1.f; g.epilogue = EPI_PLAIN;
            if ((e = gemm(g, bf16, st)) != hipSuccess) return e;
            add_colsum(v[l + 1], grads + off_b, bgd, nout, 1.f, nullptr, 0.f);
        }
        off = 0;
        for (int l = 0; l < L; ++l) {                   // dW_l += lmd e_l^T v'_{l+1}
            const int nin = dims[l], nout = dims[l + 1];
            GemmArgs g{};
            g.A = ep[l]; g.sam = 1; g.sak = nin;
            g.B = vp[l + 1]; g.sbk = nout; g.sbn = 1;
            g.C = grads + off; g.ldc = nout; g.M = nin; g.N = nout; g.K = np;
            g.alpha = lmd; g.beta = 1.f; g.epilogue = EPI_PLAIN;
            if ((e = gemm(g, bf16, st)) != hipSuccess) return e;
            off += (long)nin * nout + nout;
        }
        if ((e = gemm_batch_flush(st)) != hipSuccess) return e;
        // d/dw_out[k] += lmd * sum_b e_L[b][k]   (behind the GEMMs: one of them may have added into that gradient directly)
        add_colsum(ep[L], tmp, np, dims[L], 0.f, grads + (net.nparams - dims[L]), lmd);
        cs.dvals = dvals; cs.dnorm = ra.dnorm; cs.stats = stats; cs.ng = ng; cs.nd = nd; cs.np = np; cs.lmd = lmd;
        hipLaunchKernelGGL(colsum_batch_kernel, dim3(cs.first[cs.cnt] + 1), dim3(1024), 0, st, cs);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        return critic_splitk_flush(st);
    }
    if (eps)        // (in front of the fork: the penalty half reads hp[0])
        hipLaunchKernelGGL(critic_step_inputs_kernel, dim3(blocks_for((long)ng * dims[0])), dim3(256), 0, st, xg, xd, cg, eps, xp_out,
                           h[0], hp[0], ng, nx, hide_cell_type, nc);
    if (fk) {
        if ((e = hipEventRecord(fk->fork, st)) != hipSuccess) return e;          // behind the memset and everything the caller queued
        join_scope.forked = true;
        if ((e = hipStreamWaitEvent(fk->aux, fk->fork, 0)) != hipSuccess) return e;
        sp = fk->aux;
    }
    if (!eps) {
        hipLaunchKernelGGL(critic_input_kernel, dim3(blocks_for((long)ng * dims[0])), dim3(256), 0, st, xg, cg, h[0], ng, nx, hide_cell_type, nc);
        hipLaunchKernelGGL(critic_input_kernel, dim3(blocks_for((long)nd * dims[0])), dim3(256), 0, st, xd, cd, h[0] + (long)ng * dims[0], nd, nx, hide_cell_type, nc);
    }
    if ((e = critic_forward_pass(net, h, dvals, bgd, bf16, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(two_means_kernel, dim3(1), dim3(256), 0, st, dvals, stats, ng, nd);
    hipLaunchKernelGGL(fill_updown_kernel, dim3(blocks_for(bgd)), dim3(256), 0, st, up, ng, nd);
    // d/dw_out = sum_b up_b h_L[b][:]
    {
        GemmArgs g{};
        g.A = h[L]; g.sam = 1; g.sak = dims[L];      // op(A)(k, b) = h_L[b][k]
        g.B = up; g.sbk = 1; g.sbn = 1;
        g.C = grads + (net.nparams - dims[L]); g.ldc = 1; g.M = dims[L]; g.N = 1; g.K = bgd;
        g.alpha = 1.f; g.beta = 1.f; g.epilogue = EPI_PLAIN;
        if ((e = critic_hand_over(fk, st)) != hipSuccess) return e;              // h[L] and up are queued on st
        if ((e = gemm(g, bf16, sg)) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(critic_outgrad_kernel, dim3(blocks_for((long)bgd * dims[L])), dim3(256), 0, st, h[L], net.wout, up, v[L], bgd, dims[L], net.leak);
    if ((e = critic_backward_chain(net, h, v, bgd, grads, false, bf16, st, fk)) != hipSuccess) return e;

    // ---------------- (2) gradient penalty on xp ------------------------------------------------
    if (!eps) hipLaunchKernelGGL(critic_input_kernel, dim3(blocks_for((long)np * dims[0])), dim3(256), 0, sp, xp, cp, hp[0], np, nx, hide_cell_type, nc);
    if ((e = critic_forward_pass(net, hp, dp, np, bf16, sp)) != hipSuccess) return e;
    // input gradient g = dD/dh0 per sample: v_L = m_L * w_out, chain down to vp[0]
    hipLaunchKernelGGL(critic_outgrad_kernel, dim3(blocks_for((long)np * dims[L])), dim3(256), 0, sp, hp[L], net.wout, (const float*)nullptr, vp[L], np, dims[L], net.leak);
    if ((e = critic_backward_chain(net, hp, vp, np, nullptr, true, bf16, sp)) != hipSuccess) return e;
    // penalty and its gradient w.r.t. g: ep[0] = ghat (np x n0)
    hipLaunchKernelGGL(gp_head_kernel, dim3(1), dim3(256), 0, sp, vp[0], ep[0], stats + 2, np, dims[0], nx);
    // backprop through the linear chain g = v_1 W_1^T, v_{l} = m_l * (v_{l+1} W_{l+1}^T), v_L = m_L * w_out:
    //   dW_l[i][k] += lmd * sum_b e_{l-1}[b][i] v_l[b][k],   e_l = m_l * (e_{l-1} W_l)   (e_0 = ghat)
    long off = 0;
    for (int l = 0; l < L; ++l) {
        const int nin = dims[l], nout = dims[l + 1];
        {
            GemmArgs g{};
            g.A = ep[l]; g.sam = 1; g.sak = nin;               // op(A)(i, b) = e_l[b][i]
            g.B = vp[l + 1]; g.sbk = nout; g.sbn = 1;
            g.C = grads + off; g.ldc = nout; g.M = nin; g.N = nout; g.K = np;
            g.alpha = lmd; g.beta = 1.f; g.epilogue = EPI_PLAIN;
            if ((e = critic_hand_over(fk, sp)) != hipSuccess) return e;          // e_l (and v_{l+1} before it) are queued on sp
            if ((e = gemm(g, bf16, sg)) != hipSuccess) return e;
        }
        {
            GemmArgs g{};                                       // e_{l+1} = m_{l+1} * (e_l W_l)
            g.A = ep[l]; g.sam = nin; g.sak = 1;
            g.B = net.W[l]; g.sbk = nout; g.sbn = 1;
            g.C = ep[l + 1]; g.ldc = nout; g.M = np; g.N = nout; g.K = nin;
            g.epilogue = EPI_MASK; g.mask = hp[l + 1]; g.ldm = nout; g.leak = net.leak;
            if ((e = gemm(g, bf16, sp)) != hipSuccess) return e;
        }
        off += (long)nin * nout + nout;
    }
    // d/dw_out[k] += lmd * sum_b e_L[b][k]      (v_L = m_L * w_out, mask already applied in e_L)
    hipLaunchKernelGGL(colsum_kernel, dim3((dims[L] + 63) / 64), dim3(1024), 0, sp, ep[L], tmp, np, dims[L], 0.f);
    if (fk) {                                       // join: the sums below and the slab reduction see both halves
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = join_scope.join()) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(axpy_kernel, dim3(blocks_for(dims[L])), dim3(256), 0, st, grads + (net.nparams - dims[L]), tmp, lmd, (long)dims[L]);
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(1), 0, st, stats, lmd);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return critic_splitk_flush(st);                   // grads += the split GEMMs' slabs, in slice order
}

// Gradient of  -mean D(x)  w.r.t. the tuning-curve part of the input (generator side, wgan.py:236):
// gx[batch][nx];  also returns mean D in stats[0].
hipError_t critic_input_grad(const float* params, const int* dims, int nlayers, const float* x, const float* cond, int batch,
                             int hide_cell_type, float scale, float* gx, float* stats, float* ws, bool bf16, hipStream_t st,
                             float leak) {
    CriticNet net;
    if (!parse_net(params, dims, nlayers, net, leak)) return hipErrorInvalidValue;
    hipError_t e;
    const int L = nlayers, nc = cond ? 3 : 0, nx = dims[0] - nc;
    float *h[10], *v[10];
    float* p = ws;
    for (int l = 0; l <= L; ++l) { h[l] = p; p += (long)batch * dims[l]; }
    for (int l = 0; l <= L; ++l) { v[l] = p; p += (long)batch * dims[l]; }
    float* dv = p; p += batch;
    hipLaunchKernelGGL(critic_input_kernel, dim3(blocks_for((long)batch * dims[0])), dim3(256), 0, st, x, cond, h[0], batch, nx, hide_cell_type, nc);
    if (bf16 && batch > 0 && critic_rows_supported(dims, nlayers)) {
        if ((e = critic_rows_eval(params, dims, nlayers, h[0], batch, dv, 1, gx, scale, nx, leak, p, st)) != hipSuccess) return e;
        hipLaunchKernelGGL(two_means_kernel, dim3(1), dim3(256), 0, st, dv, stats, batch, 0);
        return hipGetLastError();
    }
    if ((e = critic_forward_pass(net, h, dv, batch, bf16, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(two_means_kernel, dim3(1), dim3(256), 0, st, dv, stats, batch, 0);
    hipLaunchKernelGGL(critic_outgrad_kernel, dim3(blocks_for((long)batch * dims[L])), dim3(256), 0, st, h[L], net.wout, (const float*)nullptr, v[L], batch, dims[L], net.leak);
    if ((e = critic_backward_chain(net, h, v, batch, nullptr, true, bf16, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(gather_scale_kernel, dim3(blocks_for((long)batch * nx)), dim3(256), 0, st, v[0], gx, batch, dims[0], nx, scale);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// building blocks exported to ssn_critic_ln.hip
// ---------------------------------------------------------------------------------
hipError_t critic_gemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                       int M, int N, int K, float alpha, float beta, bool bf16, hipStream_t st) {
    GemmArgs g{};
    g.A = A; g.sam = sam; g.sak = sak; g.B = B; g.sbk = sbk; g.sbn = sbn; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta; g.epilogue = EPI_PLAIN;
    return gemm(g, bf16, st);
}
hipError_t critic_colsum(const float* X, float* out, int batch, int n, float beta, hipStream_t st) {
    hipLaunchKernelGGL(colsum_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, X, out, batch, n, beta);
    return hipGetLastError();
}
hipError_t critic_make_input(const float* x, const float* cond, float* h0, int batch, int nx, int hide, hipStream_t st) {
    if (batch == 0) return hipSuccess;
    const int nc = cond ? 3 : 0;
    hipLaunchKernelGGL(critic_input_kernel, dim3(blocks_for((long)batch * (nx + nc))), dim3(256), 0, st, x, cond, h0, batch, nx, hide, nc);
    return hipGetLastError();
}
hipError_t critic_gp_head(const float* g, float* ghat, float* pen, int batch, int n0, int nx, hipStream_t st) {
    hipLaunchKernelGGL(gp_head_kernel, dim3(1), dim3(256), 0, st, g, ghat, pen, batch, n0, nx);
    return hipGetLastError();
}
hipError_t critic_two_means(const float* d, float* out, int ng, int nd, hipStream_t st) {
    hipLaunchKernelGGL(two_means_kernel, dim3(1), dim3(256), 0, st, d, out, ng, nd);
    return hipGetLastError();
}
hipError_t critic_loss_combine(float* stats, float lmd, hipStream_t st) {
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(1), 0, st, stats, lmd);
    return hipGetLastError();
}
hipError_t critic_gather_scale(const float* v0, float* gx, int batch, int n0, int nx, float s, hipStream_t st) {
    hipLaunchKernelGGL(gather_scale_kernel, dim3(blocks_for((long)batch * nx)), dim3(256), 0, st, v0, gx, batch, n0, nx, s);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// optimizers (wgan.py:111-165 on top of lasagne.updates.{adam,rmsprop,sgd})
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) optimizer_kernel(OptArgs o) {
    // (compared in fp32, the precision in which the caller reads the penalty back: host and device take the same decision;
    // NaN compares false: the update is made, as cwgan.py:494 does)
    if (o.gate && (float)*o.gate > (float)o.gate_bound) return;
    if (o.skip_nonfinite) {                    // (every thread reads the few gradient elements: one verdict for the launch)
        bool bad = false;
        for (long e = 0; e < o.n; ++e) bad = bad || !(fabsf(o.g[e]) <= 3.402823466e38f);
        if (bad) {
            for (long e = blockIdx.x * 256L + threadIdx.x; e < o.n; e += gridDim.x * 256L)
                if (o.record) { o.record[e] = o.p[e]; if (e == 0) o.record[o.n] = __builtin_nanf(""); }
            return;
        }
    }
    for (long e = blockIdx.x * 256L + threadIdx.x; e < o.n; e += gridDim.x * 256L) {
        const float p0 = o.p[e];
        float g = o.g[e];
        // penalties enter through the loss (wgan.py:144-151): d/dp (l2 * p^2) = 2 l2 p, d/dp (l1 |p|) = l1 sgn p
        g += 2.f * o.l2_penalty * p0 + o.l1_penalty * ((p0 > 0.f) - (p0 < 0.f));
        float pn;
        if (o.kind == 1) {
            const float m = o.beta1 * o.s1[e] + (1.f - o.beta1) * g;
            const float v = o.beta2 * o.s2[e] + (1.f - o.beta2) * g * g;
            o.s1[e] = m; o.s2[e] = v;
            pn = p0 - o.a_t * m / (sqrtf(v) + o.eps);
        } else if (o.kind == 2) {
            const float acc = o.rho * o.s1[e] + (1.f - o.rho) * g * g;
            o.s1[e] = acc;
            pn = p0 - o.lr * g / sqrtf(acc + o.eps);
        } else {
            pn = p0 - o.lr * g;
        }
        // decoupled decay on the OLD value (wgan.py:158-163, apply_l2_decay / apply_l1_decay)
        pn -= o.lr * o.l2_decay * p0 + o.lr * o.l1_decay * ((p0 > 0.f) - (p0 < 0.f));
        // wgan.py:244-251.  A NaN update stays NaN, as it does in Theano's clip (a switch on comparisons): fmaxf(NaN, lo) would
        // hand back the lower bound and hide a poisoned gradient from the drivers' NaN guards
        if (pn == pn) {
            if (o.clip_lo_v) pn = fminf(fmaxf(pn, o.clip_lo_v[e]), o.clip_hi_v[e]);      // (bounds per element: the reference's numpy clip broadcasts)
            else if (o.clip) pn = fminf(fmaxf(pn, o.clip_lo), o.clip_hi);
        }
        o.p[e] = pn;
        if (o.record) {
            o.record[e] = pn;
            if (e == 0 && o.record_tail) o.record[o.n] = *o.record_tail;
        }
    }
}
hipError_t optimizer_step(const OptArgs& o, hipStream_t st) {
    if (o.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(optimizer_kernel, dim3(blocks_for(o.n)), dim3(256), 0, st, o);
    return hipGetLastError();
}

}  // namespace ssn
