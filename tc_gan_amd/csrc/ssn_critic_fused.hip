// Small critics (every layer width <= 128: the paper's 4 x 128 LayerNorm critic, scripts/fig4/gan/run.json) in THREE
// launches per update instead of ~70: at these sizes every kernel of the layer-by-layer path (ssn_critic.hip,
// ssn_critic_ln.hip) costs ~5 us of fixed latency whatever it computes, and a critic update is a chain of ~70 of them.
//
// Everything in a critic update is ROW-LOCAL except the parameter gradients: forward pass, LayerNorm statistics, the
// input-gradient chain, the gradient-penalty head and the two sweeps of its double backward (formulas: header of
// ssn_critic_ln.hip; reference: networks/cwgan.py:190-214, simple_discriminator.py:6-75, 139-165) only ever combine
// numbers of ONE row of the stacked batch [xg; xd; xp].  So:
//   1. critic_rows_kernel   one workgroup per block of RB rows walks that block through the WHOLE sequence -- the device
//                           functions below are the kernels of ssn_critic_ln.hip turned into per-row routines (one wave
//                           per row), the layer GEMMs are done by the workgroup for its rows (plain fp32 FMAs: exact
//                           fp32 whatever `precision` says -- arithmetic is not what these sizes are bound by);
//                           intermediates go to the same global scratch arrays as before (L2-resident);
//   2. critic_wgrad_kernel  every parameter gradient = a sum over rows of products of two of those arrays: one launch,
//                           one workgroup per 16 x 16 output tile of each tensor, rows in order -> deterministic;
//   3. the optimizer step   (unchanged, ssn_critic.hip).
// The loss statistics are per-block partial sums added in block order by workgroup 0 of launch 2.
#include <hip/hip_runtime.h>
#include "ssn_host.h"

namespace ssn {

constexpr float FLN_EPS = 1e-4f;       // Lasagne BatchNormLayer default, inherited by LayerNormLayer
constexpr int FMAXW = 128;             // widest layer handled here
constexpr int FMAXL = 8;

struct FusedNet {
    int L; int dims[FMAXL + 2]; int ln[FMAXL + 1];
    const float* W[FMAXL + 1]; const float* b[FMAXL + 1]; const float* wout;
    long offW[FMAXL + 1], offb[FMAXL + 1], offout, nparams;
};
// per-row work arrays, [rows][dims[l]] each (same meaning as `Acts` of ssn_critic_ln.hip)
struct FusedActs { float *h[FMAXL + 2], *y[FMAXL + 2], *invs[FMAXL + 2], *u[FMAXL + 2], *p[FMAXL + 2], *c[FMAXL + 2]; };
// penalty rows only, [np][dims[l]]
struct FusedPen { float *du[FMAXL + 2], *dc[FMAXL + 2], *dyA[FMAXL + 2], *dpre[FMAXL + 2], *da[FMAXL + 2], *dsA[FMAXL + 2]; };

struct FusedArgs {
    FusedNet net;
    FusedActs A;          // stacked rows [xg; xd; xp] (modes 0/1: the one batch)
    FusedPen P;
    const float *xg, *cg, *xd, *cd, *xp, *cp;
    int ng, nd, np, hide;
    float lmd, scale;
    float* up;            // [rows]
    float* dall;          // [rows]  D per row
    float* dout;          // optional copy of D for the first `ndout` rows (the caller's output array)
    int ndout;
    float* part;          // [blocks][4]  per-block partial sums: sum D(xg), sum D(xd), sum (norm - 1)^2
    float* gx;            // mode 1: [batch][nx]
    int mode;             // 0 forward only, 1 input gradient, 2 loss gradient
    int nb_gd;            // blocks covering rows [0, ng + nd); the rest cover the penalty rows
};

__device__ __forceinline__ float fwave_sum(float x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// out[r][n] = sum_k Ain[r][k] * W[k * ldw + n]   (NN)   or   sum_k Ain[r][k] * W[n * ldw + k]   (NT: W^T)
// for the block's rows r0 .. r0 + nr - 1 (nr <= RB); K, N <= 128.  A is staged transposed in LDS (As[k][r]); thread
// (half, n) accumulates RB / 2 rows of column n.
template <int RB, bool NT>
__device__ __forceinline__ void block_gemm(const float* __restrict__ Ain, int lda, const float* __restrict__ W, int ldw,
                                           float* __restrict__ out, int ldo, int r0, int nr, int K, int N, float* As) {
    constexpr int HR = RB / 2;
    const int tid = threadIdx.x;
    __syncthreads();                                   // As free; producers of Ain done
    for (int e = tid; e < RB * K; e += 256) {
        const int r = e / K, k = e % K;                // coalesced over k
        As[k * RB + r] = (r < nr) ? Ain[(size_t)(r0 + r) * lda + k] : 0.f;
    }
    __syncthreads();
    const int n = tid & 127, half = tid >> 7;
    if (n < N) {
        float acc[HR];
#pragma unroll
        for (int i = 0; i < HR; ++i) acc[i] = 0.f;
        // one workgroup per CU and four waves: nothing hides a load's latency but the loads themselves -- eight weights
        // are requested before the FMAs of the first (the k loop is a chain of L2 round trips otherwise)
        constexpr int U = 32;
        const float* wp = NT ? W + (size_t)n * ldw : W + n;      // element k: wp[k] (NT) / wp[k * ldw] (NN)
        const size_t ks = NT ? 1 : (size_t)ldw;
        int k = 0;
        for (; k + U <= K; k += U) {
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = wp[(size_t)(k + u) * ks];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < HR; ++i) acc[i] = fmaf(As[(k + u) * RB + half * HR + i], w[u], acc[i]);
        }
        for (; k < K; ++k) {
            const float w = wp[(size_t)k * ks];
#pragma unroll
            for (int i = 0; i < HR; ++i) acc[i] = fmaf(As[k * RB + half * HR + i], w, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            const int r = half * HR + i;
            if (r < nr) out[(size_t)(r0 + r) * ldo + n] = acc[i];
        }
    }
    __syncthreads();
}

template <int RB>
__global__ void __launch_bounds__(256) critic_rows_kernel(FusedArgs a) {
    __shared__ float As[FMAXW * RB];
    __shared__ float red[3][4];
    const FusedNet& net = a.net;
    const int L = net.L, nx = net.dims[0] - 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bgd = a.ng + a.nd;
    const bool pen_block = (int)blockIdx.x >= a.nb_gd;
    // rows of this block in the stacked numbering, and (penalty blocks) in the penalty-row numbering
    const int r0 = pen_block ? bgd + ((int)blockIdx.x - a.nb_gd) * RB : (int)blockIdx.x * RB;
    const int rend = pen_block ? bgd + a.np : bgd;
    const int nr = (rend - r0 < RB) ? rend - r0 : RB;
    const int q0 = r0 - bgd;                           // first penalty-row index (penalty blocks)
    const FusedActs& A = a.A;

    // ---- h_0 = [x, contrast, |norm_probe|, cell_type] (cwgan.py:164-170; hide_cell_type zeroes the last, 178-187) ----
    for (int e = tid; e < nr * net.dims[0]; e += 256) {
        const int r = r0 + e / net.dims[0], j = e % net.dims[0];
        const float *x, *c; int rr;
        if (r < a.ng) { x = a.xg; c = a.cg; rr = r; }
        else if (r < bgd) { x = a.xd; c = a.cd; rr = r - a.ng; }
        else { x = a.xp; c = a.cp; rr = r - bgd; }
        float v;
        if (j < nx) v = x[(size_t)rr * nx + j];
        else if (j == nx) v = c[rr * 3 + 0];
        else if (j == nx + 1) v = fabsf(c[rr * 3 + 1]);
        else v = a.hide ? 0.f : c[rr * 3 + 2];
        A.h[0][(size_t)r * net.dims[0] + j] = v;
    }
    // ---- forward: a = h_{l-1} W_l -> (y, 1/s, h_l = relu(y + b_l)) ------------------------------------------------
    for (int l = 1; l <= L; ++l) {
        const int nin = net.dims[l - 1], n = net.dims[l];
        block_gemm<RB, false>(A.h[l - 1], nin, net.W[l - 1], n, A.u[l], n, r0, nr, nin, n, As);
        for (int rr = wave; rr < nr; rr += 4) {
            const size_t o0 = (size_t)(r0 + rr) * n;
            const float* ar = A.u[l] + o0;
            float mu = 0.f, is = 1.f;
            if (net.ln[l - 1]) {
                float s = 0.f;
                for (int j = lane; j < n; j += 64) s += ar[j];
                mu = fwave_sum(s) / n;
                float v = 0.f;
                for (int j = lane; j < n; j += 64) { const float d = ar[j] - mu; v += d * d; }
                is = rsqrtf(fwave_sum(v) / n + FLN_EPS);
            }
            for (int j = lane; j < n; j += 64) {
                const float yy = (ar[j] - mu) * is;
                A.y[l][o0 + j] = yy;
                const float pre = yy + net.b[l - 1][j];
                A.h[l][o0 + j] = pre > 0.f ? pre : 0.f;
            }
            if (lane == 0) A.invs[l][r0 + rr] = is;
        }
    }
    __syncthreads();
    // ---- D = h_L . w_out; upstream of D per row; per-block sums of D ------------------------------------------------
    float sdg = 0.f, sdd = 0.f;
    for (int rr = wave; rr < nr; rr += 4) {
        const int r = r0 + rr, nL = net.dims[L];
        float s = 0.f;
        for (int j = lane; j < nL; j += 64) s += A.h[L][(size_t)r * nL + j] * net.wout[j];
        s = fwave_sum(s);
        if (lane == 0) {
            a.dall[r] = s;
            if (a.dout && r < a.ndout) a.dout[r] = s;
            if (r < a.ng) sdg += s; else if (r < bgd) sdd += s;
            a.up[r] = (a.mode != 2) ? 1.f : (r < a.ng ? 1.f / (float)a.ng : (r < bgd ? -1.f / (float)a.nd : 1.f));
        }
    }
    if (lane == 0) { red[0][wave] = sdg; red[1][wave] = sdd; red[2][wave] = 0.f; }
    __syncthreads();
    if (a.mode == 0) {
        if (tid == 0 && a.part) {
            a.part[blockIdx.x * 4 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            a.part[blockIdx.x * 4 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
            a.part[blockIdx.x * 4 + 2] = 0.f;
        }
        return;
    }
    // ---- input-gradient chain: u_L = up w_out;  p_l = m_l u_l;  c_l = LNback(p_l);  u_{l-1} = c_l W_l^T -----------
    for (int e = tid; e < nr * net.dims[L]; e += 256) {
        const int r = r0 + e / net.dims[L], j = e % net.dims[L];
        A.u[L][(size_t)r * net.dims[L] + j] = net.wout[j] * a.up[r];
    }
    __syncthreads();
    for (int l = L; l >= 1; --l) {
        const int nin = net.dims[l - 1], n = net.dims[l];
        for (int rr = wave; rr < nr; rr += 4) {
            const size_t o0 = (size_t)(r0 + rr) * n;
            float sx = 0.f, sxy = 0.f;
            for (int j = lane; j < n; j += 64) {
                const float v = (A.h[l][o0 + j] > 0.f) ? A.u[l][o0 + j] : 0.f;
                sx += v; sxy += v * A.y[l][o0 + j];
            }
            const float mx = fwave_sum(sx) / n, mxy = fwave_sum(sxy) / n, is = A.invs[l][r0 + rr];
            for (int j = lane; j < n; j += 64) {
                const float v = (A.h[l][o0 + j] > 0.f) ? A.u[l][o0 + j] : 0.f;
                A.p[l][o0 + j] = v;
                A.c[l][o0 + j] = net.ln[l - 1] ? (v - mx - A.y[l][o0 + j] * mxy) * is : v;
            }
        }
        block_gemm<RB, true>(A.c[l], n, net.W[l - 1], n, A.u[l - 1], nin, r0, nr, n, nin, As);
    }
    if (a.mode == 1) {
        // generator side: gx = scale * dD/dx (tuning-curve part), per-block sum of D in part[][0]
        for (int e = tid; e < nr * nx; e += 256) {
            const int r = r0 + e / nx, j = e % nx;
            a.gx[(size_t)r * nx + j] = a.scale * A.u[0][(size_t)r * net.dims[0] + j];
        }
        if (tid == 0) {
            a.part[blockIdx.x * 4 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            a.part[blockIdx.x * 4 + 1] = 0.f; a.part[blockIdx.x * 4 + 2] = 0.f;
        }
        return;
    }
    float spen = 0.f;
    if (pen_block) {
        const FusedPen& P = a.P;
        const int n0 = net.dims[0];
        // ---- gradient-penalty head: norm of g_x per row; du_0 = lmd * 2 (norm - 1)/norm * g_x / np (zero beyond nx) ----
        for (int rr = wave; rr < nr; rr += 4) {
            const int r = r0 + rr, q = q0 + rr;
            float s = 0.f;
            for (int j = lane; j < nx; j += 64) { const float v = A.u[0][(size_t)r * n0 + j]; s += v * v; }
            const float nrm = sqrtf(fwave_sum(s));
            const float coef = (nrm > 0.f) ? a.lmd * 2.f * (nrm - 1.f) / nrm / (float)a.np : 0.f;
            for (int j = lane; j < n0; j += 64) P.du[0][(size_t)q * n0 + j] = (j < nx) ? coef * A.u[0][(size_t)r * n0 + j] : 0.f;
            if (lane == 0) spen += (nrm - 1.f) * (nrm - 1.f);
        }
        // ---- sweep 1: dc_l = du_{l-1} W_l;  du_l = m_l LNback(dc_l);  dyA_l, dsA_l ------------------------------------
        for (int l = 1; l <= L; ++l) {
            const int nin = net.dims[l - 1], n = net.dims[l];
            block_gemm<RB, false>(P.du[l - 1], nin, net.W[l - 1], n, P.dc[l], n, q0, nr, nin, n, As);
            for (int rr = wave; rr < nr; rr += 4) {
                const size_t o0 = (size_t)(r0 + rr) * n, p0 = (size_t)(q0 + rr) * n;
                float sdc = 0.f, sdcy = 0.f, spy = 0.f, sdcc = 0.f;
                for (int j = lane; j < n; j += 64) {
                    const float dc = P.dc[l][p0 + j], y = A.y[l][o0 + j];
                    sdc += dc; sdcy += dc * y; spy += A.p[l][o0 + j] * y; sdcc += dc * A.c[l][o0 + j];
                }
                const float mdc = fwave_sum(sdc) / n, rm = fwave_sum(sdcy) / n, qm = fwave_sum(spy) / n;
                const float is = A.invs[l][r0 + rr], tot = fwave_sum(sdcc);
                const int ln = net.ln[l - 1];
                for (int j = lane; j < n; j += 64) {
                    const float dc = P.dc[l][p0 + j], y = A.y[l][o0 + j];
                    const float dp = ln ? (dc - mdc - y * rm) * is : dc;
                    P.du[l][p0 + j] = (A.h[l][o0 + j] > 0.f) ? dp : 0.f;
                    P.dyA[l][p0 + j] = ln ? -(dc * qm + A.p[l][o0 + j] * rm) * is : 0.f;
                }
                if (lane == 0) P.dsA[l][q0 + rr] = ln ? -tot * is : 0.f;
            }
        }
        __syncthreads();
        // ---- sweep 2 (l = L..1, dh_L = 0): dpre = m_l dh_l;  da = LNback(dyA + dpre) + dsA y / n;  dh_{l-1} = da W_l^T ----
        for (int l = L; l >= 1; --l) {
            const int nin = net.dims[l - 1], n = net.dims[l];
            const bool have_dh = l < L;                 // dh_l was written into dc[l] by the step above
            for (int rr = wave; rr < nr; rr += 4) {
                const size_t o0 = (size_t)(r0 + rr) * n, p0 = (size_t)(q0 + rr) * n;
                float sd = 0.f, sdy = 0.f;
                for (int j = lane; j < n; j += 64) {
                    const float dp = (have_dh && A.h[l][o0 + j] > 0.f) ? P.dc[l][p0 + j] : 0.f;
                    P.dpre[l][p0 + j] = dp;
                    const float d = P.dyA[l][p0 + j] + dp;
                    sd += d; sdy += d * A.y[l][o0 + j];
                }
                const float md = fwave_sum(sd) / n, mdy = fwave_sum(sdy) / n, is = A.invs[l][r0 + rr], ds = P.dsA[l][q0 + rr];
                const int ln = net.ln[l - 1];
                for (int j = lane; j < n; j += 64) {
                    const float d = P.dyA[l][p0 + j] + P.dpre[l][p0 + j], y = A.y[l][o0 + j];
                    P.da[l][p0 + j] = ln ? (d - md - y * mdy) * is + ds * y / n : d;
                }
            }
            if (l > 1) block_gemm<RB, true>(P.da[l], n, net.W[l - 1], n, P.dc[l - 1], nin, q0, nr, n, nin, As);
        }
    }
    if (lane == 0) red[2][wave] = spen;
    __syncthreads();
    if (tid == 0) {
        a.part[blockIdx.x * 4 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        a.part[blockIdx.x * 4 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        a.part[blockIdx.x * 4 + 2] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    }
}

// ---- parameter gradients -----------------------------------------------------------------------------------------------
// Every gradient tensor is  G[i][j] = sum over terms t of sum_r X_t[r][i] * Y_t[r][j]  (weights: three terms -- the
// [xg; xd] chain, sweep 1, sweep 2 of the penalty rows), or a column sum of up to two arrays (biases, w_out).
// One workgroup per 16 x 16 tile; rows in increasing order, terms in a fixed order: the result does not depend on
// scheduling.  Workgroup 0 also adds the loss statistics.
struct WgradTerm { const float* X; int ldx; const float* Y; int ldy; int rows; };
struct WgradTensor {
    float* G; int M, N;         // output [M][N] (biases / w_out: M = 1; X == nullptr -> plain column sum of Y; X != nullptr with
    int nterms;                 //  ldx == 0 -> sum_r X[r] * Y[r][j], the w_out term weighted by `up`)
    WgradTerm t[3];
    int tile0;                  // first workgroup index of this tensor
};
struct WgradArgs {
    WgradTensor ten[2 * FMAXL + 2];
    int ntensors, ntiles;
    const float* part; int nblocks, nb_gd, ng, nd, np;
    float lmd; float* stats;
};

__global__ void __launch_bounds__(256) critic_wgrad_kernel(WgradArgs a) {
    __shared__ float Xs[16][17], Ys[16][17];
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    if (blockIdx.x == 0 && tid == 0 && a.stats) {
        // stats[0..3] = mean D(xg), mean D(xd), penalty, loss  (block order)
        float sg = 0.f, sd = 0.f, sp = 0.f;
        for (int b = 0; b < a.nblocks; ++b) { sg += a.part[b * 4]; sd += a.part[b * 4 + 1]; sp += a.part[b * 4 + 2]; }
        const float mg = a.ng ? sg / a.ng : 0.f, md = a.nd ? sd / a.nd : 0.f, pen = a.np ? sp / a.np : 0.f;
        a.stats[0] = mg; a.stats[1] = md; a.stats[2] = pen; a.stats[3] = mg - md + a.lmd * pen;
    }
    if (a.ntensors == 0) return;
    int k = 0;
    while (k + 1 < a.ntensors && (int)blockIdx.x >= a.ten[k + 1].tile0) ++k;
    const WgradTensor& T = a.ten[k];
    const int tile = blockIdx.x - T.tile0, ntn = (T.N + 15) / 16;
    const int i0 = (tile / ntn) * 16, j0 = (tile % ntn) * 16;
    // Row-vector tensors (bias and w_out gradients) accumulate in fp64: the bias gradient of a unit that is active on
    // equally many generated and data rows is a sum of +c and -c terms, exactly zero in the reference's arithmetic;
    // fp32 partial sums (3c, 5c, ...) round, leave ~1e-8 of noise, and Adam (eps 1e-8) turns that noise into full-size
    // steps of the bias (tests/test_critic_gpu.py::test_bias_gradients_that_cancel_are_exactly_zero).
    const bool vec = T.M == 1;
    float acc = 0.f;
    double accd = 0.0;
    for (int t = 0; t < T.nterms; ++t) {
        const WgradTerm& w = T.t[t];
        for (int r0 = 0; r0 < w.rows; r0 += 16) {
            // thread (ti, tj) stages element (row r0 + ti, offset tj) of both operands.  Row-vector tensors (biases,
            // w_out: M = 1): X is absent (column sum: 1) or one weight per row (ldx == 0); every column of Xs gets it.
            const int r = r0 + ti;
            float xv = 0.f, yv = 0.f;
            if (r < w.rows) {
                if (w.X == nullptr) xv = 1.f;
                else if (w.ldx == 0) xv = w.X[r];
                else if (i0 + tj < T.M) xv = w.X[(size_t)r * w.ldx + i0 + tj];
                if (j0 + tj < T.N) yv = w.Y[(size_t)r * w.ldy + j0 + tj];
            }
            Xs[ti][tj] = xv; Ys[ti][tj] = yv;
            __syncthreads();
            if (vec) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) accd += (double)Xs[kk][ti] * (double)Ys[kk][tj];
            } else {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) acc = fmaf(Xs[kk][ti], Ys[kk][tj], acc);
            }
            __syncthreads();
        }
    }
    if (i0 + ti < T.M && j0 + tj < T.N) T.G[(size_t)(i0 + ti) * T.N + j0 + tj] = vec ? (float)accd : acc;
}

// ---- host side ---------------------------------------------------------------------------------------------------------
bool critic_fused_supported(const int* dims, int nlayers) {
    if (nlayers < 0 || nlayers > FMAXL) return false;
    for (int l = 0; l <= nlayers; ++l) if (dims[l] > FMAXW || dims[l] < 1) return false;
    return dims[0] > 3;
}

static bool parse_fused(const float* params, const int* dims, const int* norm, int nlayers, FusedNet& net) {
    if (!critic_fused_supported(dims, nlayers)) return false;
    net.L = nlayers;
    long off = 0;
    for (int l = 0; l <= nlayers; ++l) net.dims[l] = dims[l];
    for (int l = 0; l < nlayers; ++l) {
        net.ln[l] = norm ? norm[l] : 0;
        net.offW[l] = off; net.W[l] = params + off; off += (long)dims[l] * dims[l + 1];
        net.offb[l] = off; net.b[l] = params + off; off += dims[l + 1];
    }
    net.offout = off; net.wout = params + off; off += dims[nlayers];
    net.nparams = off;
    return true;
}

static float* fcarve(float*& p, long n) { float* r = p; p += n; return r; }

size_t critic_fused_workspace_floats(const int* dims, int nlayers, int batch_gd, int batch_p) {
    long per_row = 0;
    for (int l = 0; l <= nlayers; ++l) per_row += dims[l];
    const long rows = (long)batch_gd + batch_p;
    const long blocks = (batch_gd + 3) / 4 + (batch_p + 3) / 4 + 2;
    return (size_t)(rows * (5 * per_row + (nlayers + 1) + 2) + (long)batch_p * (5 * per_row + (nlayers + 1)) + blocks * 4 + 64);
}

static void carve_fused(float*& p, const FusedNet& net, int rows, int np, FusedArgs& a) {
    for (int l = 0; l <= net.L; ++l) {
        a.A.h[l] = fcarve(p, (long)rows * net.dims[l]); a.A.y[l] = fcarve(p, (long)rows * net.dims[l]);
        a.A.invs[l] = fcarve(p, rows); a.A.u[l] = fcarve(p, (long)rows * net.dims[l]);
        a.A.p[l] = fcarve(p, (long)rows * net.dims[l]); a.A.c[l] = fcarve(p, (long)rows * net.dims[l]);
    }
    a.up = fcarve(p, rows); a.dall = fcarve(p, rows);
    for (int l = 0; l <= net.L; ++l) {
        a.P.du[l] = fcarve(p, (long)np * net.dims[l]); a.P.dc[l] = fcarve(p, (long)np * net.dims[l]);
        a.P.dyA[l] = fcarve(p, (long)np * net.dims[l]); a.P.dpre[l] = fcarve(p, (long)np * net.dims[l]);
        a.P.da[l] = fcarve(p, (long)np * net.dims[l]); a.P.dsA[l] = fcarve(p, np);
    }
}

// rows per block: as few as still leaves the chip short of workgroups (the kernel is a latency chain per workgroup)
static int fused_rb(int rows) { return rows <= 1024 ? 4 : (rows <= 2048 ? 8 : 16); }

static hipError_t launch_rows(const FusedArgs& a, int rb, int nblocks, hipStream_t st) {
    if (nblocks <= 0) return hipSuccess;
    if (rb == 4) hipLaunchKernelGGL((critic_rows_kernel<4>), dim3(nblocks), dim3(256), 0, st, a);
    else if (rb == 8) hipLaunchKernelGGL((critic_rows_kernel<8>), dim3(nblocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((critic_rows_kernel<16>), dim3(nblocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t critic_fused_forward(const float* params, const int* dims, const int* norm, int nlayers, const float* x,
                                const float* cond, int batch, int hide, float* out, float* ws, hipStream_t st) {
    FusedArgs a{};
    if (!parse_fused(params, dims, norm, nlayers, a.net)) return hipErrorInvalidValue;
    float* p = ws;
    carve_fused(p, a.net, batch, 0, a);
    a.xg = x; a.cg = cond; a.ng = batch; a.nd = 0; a.np = 0; a.hide = hide; a.mode = 0; a.part = nullptr;
    const int rb = fused_rb(batch);
    a.nb_gd = (batch + rb - 1) / rb;
    a.dout = out; a.ndout = batch;
    return launch_rows(a, rb, a.nb_gd, st);
}

hipError_t critic_fused_input_grad(const float* params, const int* dims, const int* norm, int nlayers, const float* x,
                                   const float* cond, int batch, int hide, float scale, float* gx, float* stats, float* ws,
                                   hipStream_t st) {
    FusedArgs a{};
    if (!parse_fused(params, dims, norm, nlayers, a.net)) return hipErrorInvalidValue;
    float* p = ws;
    carve_fused(p, a.net, batch, 0, a);
    a.xg = x; a.cg = cond; a.ng = batch; a.nd = 0; a.np = 0; a.hide = hide; a.mode = 1; a.scale = scale; a.gx = gx;
    const int rb = fused_rb(batch);
    a.nb_gd = (batch + rb - 1) / rb;
    a.part = fcarve(p, (long)a.nb_gd * 4);
    hipError_t e = launch_rows(a, rb, a.nb_gd, st);
    if (e != hipSuccess) return e;
    WgradArgs w{};                                       // no tensors: workgroup 0 only adds the block sums of D
    w.ntensors = 0; w.ntiles = 1; w.part = a.part; w.nblocks = a.nb_gd; w.nb_gd = a.nb_gd; w.ng = batch; w.nd = 0; w.np = 0;
    w.lmd = 0.f; w.stats = stats;
    w.ten[0].tile0 = 0; w.ten[0].M = 0; w.ten[0].N = 0; w.ten[0].nterms = 0; w.ten[0].G = nullptr;
    hipLaunchKernelGGL(critic_wgrad_kernel, dim3(1), dim3(256), 0, st, w);
    return hipGetLastError();
}

hipError_t critic_fused_loss_grad(const float* params, const int* dims, const int* norm, int nlayers, const float* xg,
                                  const float* cg, const float* xd, const float* cd, const float* xp, const float* cp, int ng,
                                  int nd, int np, float lmd, int hide, float* grads, float* stats, float* dvals, float* ws,
                                  hipStream_t st) {
    FusedArgs a{};
    if (!parse_fused(params, dims, norm, nlayers, a.net)) return hipErrorInvalidValue;
    const FusedNet& net = a.net;
    const int L = net.L, bgd = ng + nd, rows = bgd + np;
    float* p = ws;
    carve_fused(p, net, rows, np, a);
    a.xg = xg; a.cg = cg; a.xd = xd; a.cd = cd; a.xp = xp; a.cp = cp; a.ng = ng; a.nd = nd; a.np = np; a.hide = hide;
    a.lmd = lmd; a.mode = 2;
    const int rb = fused_rb(rows);
    a.nb_gd = (bgd + rb - 1) / rb;
    const int nb_p = (np + rb - 1) / rb, nblocks = a.nb_gd + nb_p;
    a.part = fcarve(p, (long)nblocks * 4);
    a.dout = dvals; a.ndout = bgd;
    hipError_t e = launch_rows(a, rb, nblocks, st);
    if (e != hipSuccess) return e;

    WgradArgs w{};
    int nt = 0, tiles = 0;
    auto add = [&](float* G, int M, int N) -> WgradTensor& {
        WgradTensor& T = w.ten[nt++];
        T.G = G; T.M = M; T.N = N; T.nterms = 0; T.tile0 = tiles;
        tiles += ((M + 15) / 16) * ((N + 15) / 16);
        return T;
    };
    auto term = [&](WgradTensor& T, const float* X, int ldx, const float* Y, int ldy, int nrows) {
        if (nrows > 0) T.t[T.nterms++] = WgradTerm{X, ldx, Y, ldy, nrows};
    };
    const FusedActs& A = a.A;
    for (int l = 1; l <= L; ++l) {
        const int nin = net.dims[l - 1], n = net.dims[l];
        // dW_l = h_{l-1}^T c_l ([xg; xd] rows)  +  du_{l-1}^T c_l (penalty rows, sweep 1)  +  h_{l-1}^T da_l (sweep 2)
        WgradTensor& TW = add(grads + net.offW[l - 1], nin, n);
        term(TW, A.h[l - 1], nin, A.c[l], n, bgd);
        term(TW, a.P.du[l - 1], nin, A.c[l] + (size_t)bgd * n, n, np);
        term(TW, A.h[l - 1] + (size_t)bgd * nin, nin, a.P.da[l], n, np);
        // db_l = colsum(p_l) ([xg; xd] rows) + colsum(dpre_l) (penalty rows)
        WgradTensor& Tb = add(grads + net.offb[l - 1], 1, n);
        term(Tb, nullptr, 0, A.p[l], n, bgd);
        term(Tb, nullptr, 0, a.P.dpre[l], n, np);
    }
    // dw_out = sum_r up_r h_L[r] ([xg; xd] rows) + colsum(du_L) (penalty rows)
    WgradTensor& To = add(grads + net.offout, 1, net.dims[L]);
    term(To, a.up, 0, A.h[L], net.dims[L], bgd);
    term(To, nullptr, 0, a.P.du[L], net.dims[L], np);
    w.ntensors = nt; w.ntiles = tiles;
    w.part = a.part; w.nblocks = nblocks; w.nb_gd = a.nb_gd; w.ng = ng; w.nd = nd; w.np = np; w.lmd = lmd; w.stats = stats;
    hipLaunchKernelGGL(critic_wgrad_kernel, dim3(tiles), dim3(256), 0, st, w);
    return hipGetLastError();
}

}  // namespace ssn
