// Wide plain critics (hidden widths multiples of 32 up to 512: BASELINE config 3's 3 x 512) -- the ROW-LOCAL part of a
// critic update as ONE launch on the bf16 matrix cores.
//
// The layer-by-layer path (ssn_critic.hip) runs a critic update as two chains of ~20 dependent launches (Wasserstein half:
// forward, D, upstream, backward chain; penalty half: forward, input-gradient chain, penalty head, second chain), each a
// 128-256 workgroup GEMM of 8-16 us that is bound by its own latency (launch, first operand fetch, eight barriers), not by
// arithmetic: 0.41 ms per update at the C3 shape.  Everything in those chains combines numbers of ONE row of the stacked
// batch only (ssn_critic_fused.hip says the same of the small critics), so a workgroup that owns a block of 32 rows can walk
// it through the whole sequence with nobody to wait for:
//
//   critic_pack_kernel   the weights as bf16 MFMA B-fragments, once per update: for every layer the operand of the forward
//                        form  op(B)(k, n) = W_l[k][n]  and of the transposed form  op(B)(k, i) = W_l[i][k], tile by tile
//                        ([n tile][k step][lane][8 values] = one coalesced 1 KB read per MFMA), and w_out as a one-column tile;
//   critic_rows_kernel   one workgroup (16 waves) per 32 rows.  The A operand (the rows' current activations, bf16) lives in
//                        LDS, two buffers in ping-pong, one barrier per layer; B fragments stream from L2 (3 MB of packed
//                        weights: every XCD's L2 holds them all) eight k steps ahead, the first eight of a pass requested in
//                        front of the epilogue of the pass before; every wave owns the n tiles w, w + 16, ... of the layer's
//                        output (one at 512 wide), finishes them (bias + nonlinearity or mask), writes them to the next A buffer
//                        and, as fp32, to the global arrays the weight-gradient GEMMs read afterwards (h, v, e: the layout of
//                        critic_loss_grad's workspace).  The activation masks never leave the lane: the lane that finished
//                        h_l[m][n] is the one that masks v_l[m][n] and e_l[m][n] (same tile, same accumulator layout).
//
// (Not to be confused with critic_rows_kernel<RB> of ssn_critic_fused.hip, the fp32 row-block kernel of the small critics: this
// one takes RowsArgs and is no template.)  The kernel has three modes: the update (2), D only (0: forward, accuracy), D and the
// input gradient (1: the generator side).
//
// SAME BITS as the layer-by-layer path: same instruction (v_mfma_f32_32x32x16_bf16), same operand rounding (fp32 master
// values to bf16 by round-to-nearest-even), same k order of every accumulation, same epilogue expressions; the penalty head is
// the per-row loop of gp_head_kernel (critic_gp_row), and the cross-row sums (mean D(xg), mean D(xd), penalty) are taken by
// critic_stats_block (one workgroup of colsum_batch_kernel) in the orders of two_means_kernel / gp_head_kernel / loss_combine_kernel.
// tests/test_critic_gpu.py::test_rows_path_matches_layer_path compares loss, statistics and every gradient bit for bit.
//
// Reference semantics: networks/cwgan.py:123-214, simple_discriminator.py:139-165 (restated in oracle/gan_torch.py).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdlib>
#include "ssn_host.h"
#include "ssn_critic_dev.h"

namespace ssn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int RW_RB = 32;                 // rows per workgroup = one MFMA row tile
constexpr int RW_MAXW = 512;              // widest layer
constexpr int RW_LDA = RW_MAXW + 8;       // LDS row stride of the A operand (halfs): rows 16 banks apart
#ifndef SSN_ROWS_WAVES
#define SSN_ROWS_WAVES 16
#endif
#ifndef SSN_ROWS_DEPTH
#define SSN_ROWS_DEPTH 8
#endif
#ifndef SSN_ROWS_NT
#define SSN_ROWS_NT 0                     // 1: the fp32 copies for the weight-gradient GEMMs leave with the non-temporal policy
#endif
constexpr int RW_WAVES = SSN_ROWS_WAVES;
constexpr int RW_TPW = RW_MAXW / 32 / RW_WAVES;   // n tiles per wave
constexpr int RW_DEPTH = SSN_ROWS_DEPTH;  // k steps of B fragments in flight per tile

__device__ __forceinline__ unsigned short rows_bf16(float x) {
    return __builtin_bit_cast(unsigned short, __float2bfloat16(x));
}

__device__ __forceinline__ void rows_store(float* p, float v) {
    if (SSN_ROWS_NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// ------------------------------------------------------------------------------------------------------------------
// weights -> B fragments
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) critic_pack_kernel(RowsPackArgs a) {
    const long gid = blockIdx.x * 256L + threadIdx.x;             // one thread per (tile, k step, lane): 8 values, 16 bytes
    const long pack_threads = (a.total + 255) / 256 * 256;
    if (gid >= pack_threads) {                                    // the blocks behind: the update's gradient vector starts at zero
        for (long e = gid - pack_threads; e < a.nzero; e += (gridDim.x * 256L - pack_threads)) a.zero[e] = 0.f;
        return;
    }
    if (gid >= a.total) return;
    int s = 0;
    while (s + 1 < a.nseg && gid >= a.seg[s + 1].start) ++s;
    const RowsPackSeg sg = a.seg[s];
    const long e = gid - sg.start;
    const int lane = (int)(e & 63);
    const long tk = e >> 6;
    const int ks = (int)(tk % sg.KS), t = (int)(tk / sg.KS);
    const int n = 32 * t + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    unsigned short v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        float x = 0.f;
        if (sg.kind == 0) { if (k < sg.nin && n < sg.nout) x = sg.src[(long)k * sg.nout + n]; }        // op(B)(k, n) = W[k][n]
        else if (sg.kind == 1) { if (k < sg.nout && n < sg.nin) x = sg.src[(long)n * sg.nout + k]; }   // op(B)(k, i) = W[i][k]
        else { if (n == 0 && k < sg.nin) x = sg.src[k]; }                                              // op(B)(k, 0) = w_out[k]
        v[j] = rows_bf16(x);
    }
    typedef unsigned short us8 __attribute__((ext_vector_type(8)));
    *reinterpret_cast<us8*>(sg.dst + e * 8) = (us8){v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
}

// ------------------------------------------------------------------------------------------------------------------
// one layer of one row block: acc[j] = A (32 x 16 KS, LDS) . B (tile wave + RW_WAVES j of `pack`), in two pieces -- the first
// RW_DEPTH B fragments of a pass are requested BEFORE the epilogue of the pass in front of it (they depend on nothing the block
// computes, and a load issued behind the epilogue's stores would wait for them: vmcnt counts in issue order)
// ------------------------------------------------------------------------------------------------------------------
struct RowsPass { const unsigned short* pack; int KS, NT; };

__device__ __forceinline__ void rows_prefetch(const RowsPass& ps, bf16x8 (&q)[RW_TPW][RW_DEPTH], int wave, int lane) {
#pragma unroll
    for (int j = 0; j < RW_TPW; ++j) {
        const int t = wave + RW_WAVES * j;
        if (t < ps.NT) {
            const unsigned short* bp = ps.pack + ((long)t * ps.KS) * 512 + lane * 8;
#pragma unroll
            for (int d = 0; d < RW_DEPTH; ++d)
                if (d < ps.KS) q[j][d] = *reinterpret_cast<const bf16x8*>(bp + (long)d * 512);
        }
    }
}
__device__ __forceinline__ void rows_mma(const unsigned short* A, const RowsPass& ps, bf16x8 (&q)[RW_TPW][RW_DEPTH], f32x16 (&acc)[RW_TPW],
                                         int wave, int lane) {
    const int KS = ps.KS;
    const unsigned short* ap = A + (lane & 31) * RW_LDA + 8 * (lane >> 5);
    const unsigned short* bp[RW_TPW];
    bool on[RW_TPW];
#pragma unroll
    for (int j = 0; j < RW_TPW; ++j) {
        const int t = wave + RW_WAVES * j;
        on[j] = t < ps.NT;
        bp[j] = ps.pack + ((long)(on[j] ? t : 0) * KS) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    }
    if (!on[0]) return;                                  // (tiles are dealt in order: no first tile, no second)
    for (int ks0 = 0; ks0 < KS; ks0 += RW_DEPTH) {
#pragma unroll
        for (int d = 0; d < RW_DEPTH; ++d) {
            const int ks = ks0 + d;
            if (ks < KS) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(ap + 16 * ks);
#pragma unroll
                for (int j = 0; j < RW_TPW; ++j) {
                    if (on[j]) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, q[j][d], acc[j], 0, 0, 0);
                        if (ks + RW_DEPTH < KS) q[j][d] = *reinterpret_cast<const bf16x8*>(bp[j] + (long)(ks + RW_DEPTH) * 512);
                    }
                }
            }
        }
    }
}

// Finish the wave's tiles.  EPI 0: bias + nonlinearity (records the mask), 1: mask, 2: plain.  Values go to the next A buffer
// (bf16) and, for rows that exist, to gout[row][n] (fp32); C layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
template <int EPI>
__device__ __forceinline__ void rows_finish(const f32x16 (&acc)[RW_TPW], int N, int NT, int r0, int nrows, const float* __restrict__ bias,
                                            float leak, unsigned& mask, float* __restrict__ gout, unsigned short* Anext, int wave, int lane) {
    unsigned bits = EPI == 0 ? 0u : mask;
#pragma unroll
    for (int j = 0; j < RW_TPW; ++j) {
        const int t = wave + RW_WAVES * j;
        if (t >= NT) continue;
        const int n = 32 * t + (lane & 31);
        if (n >= N) continue;
        const float bn = EPI == 0 ? bias[n] : 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            float v = acc[j][reg];
            if (EPI == 0) {
                v += bn;
                const bool pos = v > 0.f;
                v = pos ? v : leak * v;
                bits |= (pos ? 1u : 0u) << (16 * j + reg);
            } else if (EPI == 1) {
                v = ((bits >> (16 * j + reg)) & 1u) ? v : leak * v;
            }
            Anext[m * RW_LDA + n] = rows_bf16(v);
            if (gout && r0 + m < nrows) rows_store(gout + (long)(r0 + m) * N + n, v);
        }
    }
    if (EPI == 0) mask = bits;
}

// grid: ceil(np / 32) penalty blocks first (their chain is twice as long), then ceil((ng + nd) / 32) blocks of [xg; xd] rows
__global__ void __launch_bounds__(64 * RW_WAVES) critic_rows_kernel(RowsArgs a) {
    __shared__ __align__(16) unsigned short Abuf[2][RW_RB * RW_LDA];
    __shared__ unsigned maskw[9][64 * RW_WAVES];
    __shared__ float G[RW_RB][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.L;
    // mode 2: the update (penalty blocks + [xg; xd] blocks); mode 0: D of ng + nd rows only; mode 1: D and the input gradient of np rows
    const int mode = a.mode;
    const bool keep = mode == 2;                         // the fp32 copies are for the weight-gradient GEMMs
    const int nbp = (a.np + RW_RB - 1) / RW_RB;
    const bool pen = (int)blockIdx.x < nbp;
    const int blk = pen ? blockIdx.x : blockIdx.x - nbp;
    const int r0 = blk * RW_RB, nrows = pen ? a.np : a.ng + a.nd;
    auto hh = [&](int l) { return pen ? a.hp[l] : a.h[l]; };      // (uniform selects of kernel arguments: no copy of `a`)
    auto vv = [&](int l) { return pen ? a.vp[l] : a.v[l]; };
    const float leak = a.leak;
    int cur = 0;
    // the block's passes in order: forward 0 .. L-1, backward L-1 .. 1, and for penalty rows g = v_1 W_0^T and the second chain
    auto pass_at = [&](int i) {
        RowsPass ps{nullptr, 0, 0};
        if (i < L) { ps.pack = a.pf[i]; ps.KS = (a.dims[i] + 15) / 16; ps.NT = a.dims[i + 1] / 32; }
        else if (mode == 0) {}
        else if (i < 2 * L - 1) { const int l = 2 * L - 1 - i; ps.pack = a.pb[l]; ps.KS = a.dims[l + 1] / 16; ps.NT = a.dims[l] / 32; }
        else if (pen && i == 2 * L - 1) { ps.pack = a.pb[0]; ps.KS = a.dims[1] / 16; ps.NT = 1; }
        else if (pen && mode == 2 && i < 3 * L) { const int l = i - 2 * L; ps.pack = a.pf[l]; ps.KS = (a.dims[l] + 15) / 16; ps.NT = a.dims[l + 1] / 32; }
        return ps;
    };
    bf16x8 q[RW_TPW][RW_DEPTH];
    int ip = 0;
    RowsPass ps = pass_at(0);
    rows_prefetch(ps, q, wave, lane);
    // acc = the current pass on the A buffer `cur`; then the next pass's first fragments are on their way
    auto run_pass = [&](f32x16 (&acc_)[RW_TPW]) {
        rows_mma(Abuf[cur], ps, q, acc_, wave, lane);
        ps = pass_at(++ip);
        if (ps.pack) rows_prefetch(ps, q, wave, lane);
    };
    // ---- the rows' input block (built by critic_input_kernel / critic_step_inputs_kernel), zero beyond dims[0] and the last row
    {
        const int n0 = a.dims[0], kp = 16 * ((n0 + 15) / 16);
        for (int e = tid; e < RW_RB * kp; e += 64 * RW_WAVES) {
            const int m = e / kp, k = e % kp;
            const float x = (k < n0 && r0 + m < nrows) ? hh(0)[(long)(r0 + m) * n0 + k] : 0.f;
            Abuf[0][m * RW_LDA + k] = rows_bf16(x);
        }
    }
    __syncthreads();
    f32x16 acc[RW_TPW];
    unsigned mk = 0;
    // ---- forward: h_{l+1} = f(h_l W_l + b_l)
    for (int l = 0; l < L; ++l) {
        const int N = a.dims[l + 1], NT = N / 32;
        run_pass(acc);
        rows_finish<0>(acc, N, NT, r0, nrows, a.b[l], leak, mk, keep ? hh(l + 1) : nullptr, Abuf[cur ^ 1], wave, lane);
        maskw[l + 1][tid] = mk;
        cur ^= 1;
        __syncthreads();
    }
    const int NL = a.dims[L], NTL = NL / 32, KSL = NL / 16;
    // ---- D = h_L w_out ([xg; xd] rows; the penalty half of an update never uses its D)
    if ((!pen || mode == 1) && wave == 0) {
        f32x16 d;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = 0.f;
        const unsigned short* ap = Abuf[cur] + (lane & 31) * RW_LDA + 8 * (lane >> 5);
        const unsigned short* bp = a.po + lane * 8;
        for (int ks = 0; ks < KSL; ++ks)
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(ap + 16 * ks),
                                                        *reinterpret_cast<const bf16x8*>(bp + (long)ks * 512), d, 0, 0, 0);
        if ((lane & 31) == 0) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = r0 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (row < nrows) a.dvals[row] = d[reg];
            }
        }
    }
    if (mode == 0) return;
    if (!pen && tid < RW_RB && r0 + tid < nrows) a.up[r0 + tid] = critic_updown(r0 + tid, a.ng, a.nd);
    // ---- v_L = m_L * w_out * up   (critic_outgrad_kernel; the penalty half: up = 1)
    {
        const unsigned bits = maskw[L][tid];
        float* const vL = vv(L);
#pragma unroll
        for (int j = 0; j < RW_TPW; ++j) {
            const int t = wave + RW_WAVES * j;
            if (t >= NTL) continue;
            const int n = 32 * t + (lane & 31);
            const float wn = a.wout[n];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const float v = wn * (pen ? 1.f : critic_updown(r0 + m, a.ng, a.nd));
                const float o = ((bits >> (16 * j + reg)) & 1u) ? v : leak * v;
                Abuf[cur ^ 1][m * RW_LDA + n] = rows_bf16(o);
                if (keep && r0 + m < nrows) rows_store(vL + (long)(r0 + m) * NL + n, o);
            }
        }
    }
    cur ^= 1;
    __syncthreads();
    // ---- backward chain: v_l = m_l * (v_{l+1} W_l^T), l = L - 1 .. 1
    for (int l = L - 1; l >= 1; --l) {
        const int N = a.dims[l], NT = N / 32;
        run_pass(acc);
        mk = maskw[l][tid];
        rows_finish<1>(acc, N, NT, r0, nrows, nullptr, leak, mk, keep ? vv(l) : nullptr, Abuf[cur ^ 1], wave, lane);
        cur ^= 1;
        __syncthreads();
    }
    if (!pen) return;
    // ---- penalty rows: g = v_1 W_0^T (no mask), the penalty head per row, then e_{l+1} = m_{l+1} * (e_l W_l)
    {
        const int n0 = a.dims[0];
        run_pass(acc);
        if (wave == 0) {
            const int n = lane & 31;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) G[(reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)][n] = acc[0][reg];
        }
        __syncthreads();
        if (mode == 1) {                                  // gx = scale * g[:, 0:nx]   (gather_scale_kernel)
            for (int e = tid; e < RW_RB * a.nx; e += 64 * RW_WAVES) {
                const int m = e / a.nx, j = e % a.nx;
                if (r0 + m < nrows) a.gx[(long)(r0 + m) * a.nx + j] = a.scale * G[m][j];
            }
            return;
        }
        const int kp = 16 * ((n0 + 15) / 16);
        if (tid < RW_RB) {
            const int m = tid;
            float coef;
            const float d = critic_gp_row(G[m], a.nx, a.np, coef);
            for (int j = 0; j < kp; ++j) {
                const float gh = (j < a.nx) ? coef * G[m][j] : 0.f;
                Abuf[cur ^ 1][m * RW_LDA + j] = rows_bf16(gh);
                if (j < n0 && r0 + m < nrows) a.ep[0][(long)(r0 + m) * n0 + j] = gh;
            }
            if (r0 + m < nrows) a.dnorm[r0 + m] = d;
        }
        cur ^= 1;
        __syncthreads();
    }
    for (int l = 0; l < L; ++l) {
        const int N = a.dims[l + 1], NT = N / 32;
        run_pass(acc);
        mk = maskw[l + 1][tid];
        rows_finish<1>(acc, N, NT, r0, nrows, nullptr, leak, mk, a.ep[l + 1], Abuf[cur ^ 1], wave, lane);
        cur ^= 1;
        if (l + 1 < L) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
bool critic_rows_supported(const int* dims, int nlayers) {
    static const bool enabled = [] { const char* v = std::getenv("SSN_CRITIC_ROWS"); return !(v && v[0] == '0'); }();
    if (!enabled || nlayers < 1 || nlayers > 8 || dims[0] < 1 || dims[0] > 32) return false;
    for (int l = 1; l <= nlayers; ++l) if (dims[l] % 32 != 0 || dims[l] > RW_MAXW || dims[l] < 32) return false;
    return true;
}
// halfs of the packed weights: forward and transposed form of every layer, w_out
static long rows_pack_halfs(const int* dims, int L, long* off_f, long* off_b, long* off_o) {
    long n = 0;
    for (int l = 0; l < L; ++l) {
        if (off_f) off_f[l] = n;
        n += (long)(dims[l + 1] / 32) * ((dims[l] + 15) / 16) * 512;
        if (off_b) off_b[l] = n;
        n += (long)((dims[l] + 31) / 32) * (dims[l + 1] / 16) * 512;
    }
    if (off_o) *off_o = n;
    n += (long)(dims[L] / 16) * 512;
    return n;
}
size_t critic_rows_workspace_floats(const int* dims, int nlayers, int batch_p) {
    if (!critic_rows_supported(dims, nlayers)) return 0;
    return (size_t)(rows_pack_halfs(dims, nlayers, nullptr, nullptr, nullptr) + 1) / 2 + (size_t)batch_p + 16;
}

hipError_t critic_rows_pack(const float* params, const int* dims, int L, float* ws_pack, RowsArgs& ra, float* zero, long nzero, hipStream_t st) {
    long off_f[9], off_b[9], off_o;
    rows_pack_halfs(dims, L, off_f, off_b, &off_o);
    unsigned short* base = reinterpret_cast<unsigned short*>((reinterpret_cast<size_t>(ws_pack) + 15) & ~(size_t)15);
    RowsPackArgs pa{};
    long start = 0, poff = 0;
    int ns = 0;
    for (int l = 0; l < L; ++l) {
        const int nin = dims[l], nout = dims[l + 1];
        const float* W = params + poff;
        RowsPackSeg f{W, base + off_f[l], nin, nout, nout / 32, (nin + 15) / 16, 0, start};
        pa.seg[ns++] = f; start += (long)f.NT * f.KS * 64;
        RowsPackSeg b{W, base + off_b[l], nin, nout, (nin + 31) / 32, nout / 16, 1, start};
        pa.seg[ns++] = b; start += (long)b.NT * b.KS * 64;
        ra.pf[l] = f.dst; ra.pb[l] = b.dst;
        ra.b[l] = W + (long)nin * nout;
        poff += (long)nin * nout + nout;
    }
    RowsPackSeg o{params + poff, base + off_o, dims[L], 1, 1, dims[L] / 16, 2, start};
    pa.seg[ns++] = o; start += (long)o.NT * o.KS * 64;
    ra.po = o.dst; ra.wout = params + poff;
    pa.nseg = ns; pa.total = start;
    pa.zero = zero; pa.nzero = zero ? nzero : 0;
    long zblocks = (pa.nzero + 1023) / 1024;            // four elements per thread of the zeroing blocks
    if (zblocks > 1024) zblocks = 1024;
    hipLaunchKernelGGL(critic_pack_kernel, dim3((unsigned)((start + 255) / 256 + zblocks)), dim3(256), 0, st, pa);
    return hipGetLastError();
}

hipError_t critic_rows_launch(const RowsArgs& ra, hipStream_t st) {
    const int nb = (ra.np + RW_RB - 1) / RW_RB + (ra.ng + ra.nd + RW_RB - 1) / RW_RB;
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(critic_rows_kernel, dim3(nb), dim3(64 * RW_WAVES), 0, st, ra);
    return hipGetLastError();
}
}  // namespace ssn
