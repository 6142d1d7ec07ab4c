// Fixed-time SSN generator on MI355X (gfx950): forward with fused reductions and the
// BPTT adjoint (reverse-time) sweep, both in the "tile" register-stationary layout of
// ssn_tile.hip / ssn_tile_core.h.
//
// Reference semantics (Theano/Lasagne graphs, restated in oracle/gan_torch.py):
//   forward   networks/ssn.py:555-576  r_{t+1} = (1-eps) r_t + eps f(W r_t + I),  r_0 = 0,
//             eps = dt/tau per neuron; 598-633: output index t = state after t+1 steps,
//             rs = outputs[skip:], time_avg = mean_t rs, dynamics_penalty = mean (rs[t+1]-rs[t])^2,
//             rate_penalty = mean relu(rs - theta).
//   backward  theano.grad of the generator loss through the scan (wgan.py:236-242): the
//             reverse-time adjoint  a_{t-1} = g_{t-1} + (1-eps) a_t + W^T (eps f'(u_t) a_t),
//             dL/dW = sum_t (eps f'(u_t) a_t) x_{t-1}^T.
//
// Storage for BPTT: the forward writes the trajectory x_t and f'(u_t) (fp32,
// [B][NB][T][M] each; 7.9 GB each at the C3 shape -- HBM capacity and bandwidth are idle
// on this path, FLOPs are not, so nothing is recomputed).  The backward overwrites the
// f' buffer in place with delta_t = eps f'(u_t) a_t shifted by one step, so that
// dL/dW[b] = delta[b]^T . traj[b] is one plain batched GEMM over K = NB*T.
#include <hip/hip_runtime.h>
#include "ssn_device.h"
#include "ssn_host.h"
#include "ssn_tile_core.h"

namespace ssn {

// f(v) and f'(v) (branch form of ssnode.c:25-53; derivative of the selected branch, as the
// clip/where graph of ssnode.py:129-149 differentiates).
template <typename T>
__device__ __forceinline__ void io_eval_grad(T v, const IoConsts<T>& c, T& f, T& df) {
    if (!(v > (T)0)) { f = (v != v) ? v : (T)0; df = (T)0; return; }
    if (c.io_type == SSN_IO_POWER || v <= c.v0) {
        f = pow_rate(v, c.k, c.n);
        df = c.n * f / v;
        return;
    }
    if (c.io_type == SSN_IO_LINEAR) { f = c.soft + c.lin_slope * (v - c.v0); df = c.lin_slope; return; }
    const T th = tanh_pos(c.tanh_gain * (v - c.v0));
    f = c.soft + c.span * th;
    df = c.span_gain * ((T)1 - th * th);
}

// fp32: k v^n = 2^(n log2 v + log2 k) without the sign test in front (v = 0 -> 0, v < 0 -> NaN, selected away; NaN
// stays NaN) and f' = n f / v through v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division.
template <>
__device__ __forceinline__ void io_eval_grad<float>(float v, const IoConsts<float>& c, float& f, float& df) {
    const float pw = __builtin_amdgcn_exp2f(fmaf(c.n, __builtin_amdgcn_logf(v), c.log2k));
    f = (v < 0.f) ? 0.f : pw;
    df = (v > 0.f) ? c.n * pw * __builtin_amdgcn_rcpf(v) : 0.f;
    if (c.io_type != SSN_IO_POWER && v > c.v0) {
        if (c.io_type == SSN_IO_LINEAR) { f = c.soft + c.lin_slope * (v - c.v0); df = c.lin_slope; }
        else {
            const float th = tanh_pos(c.tanh_gain * (v - c.v0));
            f = c.soft + c.span * th;
            df = c.span_gain * (1.f - th * th);
        }
    }
}

template <typename T, int RA, int C, int RL, int NB, int MAXTHREADS, int MINWAVES>
__global__ void __launch_bounds__(MAXTHREADS, MINWAVES) gen_forward_kernel(GenFwdArgs<T> a) {
    constexpr int CP = SlabPad<C>::value;
    __shared__ __align__(16) T rbuf[2][NB][8 * CP];
    __shared__ __align__(16) T wlds[TileSplit<RA, C, RL>::lds_elems(MAXTHREADS)];   // RL > 0: see ssn_tile_core.h
    const int M = a.M, N = a.M / 2, T_ = a.seqlen;
    const int ngroups = (a.NB + NB - 1) / NB;
    const int b = blockIdx.x / ngroups;
    const int s0 = (blockIdx.x % ngroups) * NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & 7, rg = lane >> 3;
    const int rowbase = (8 * wave + rg) * RA;

    T w[RL == 0 ? RA : 1][RL == 0 ? C : 1];     // all-register shape
    SplitTile<T, RA, C, RL> sw;                 // split shape (RL > 0)
    if constexpr (RL > 0) sw.template load<false>(a.W + (size_t)b * M * M, M, rowbase, cg * C, wlds, threadIdx.x);
    else tile_load<T, RA, C, false>(a.W + (size_t)b * M * M, M, rowbase, cg * C, w);

    const int myrow = rowbase + cg;
    const bool fin = (cg < RA) && (myrow < M);
    const int myslot = (myrow / C) * CP + (myrow % C);
    const T eps = (myrow < N) ? a.eps_E : a.eps_I;
    T rc[NB], ex[NB], ta[NB], dp[NB], rpn[NB];
    bool live[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        live[s] = fin && (s0 + s) < a.NB;
        rc[s] = ta[s] = dp[s] = rpn[s] = (T)0;
        ex[s] = live[s] ? a.ext[((size_t)b * a.NB + s0 + s) * M + myrow] : (T)0;
    }
    for (int c = threadIdx.x; c < 2 * NB * 8 * CP; c += blockDim.x) (&rbuf[0][0][0])[c] = (T)0;
    __syncthreads();

    // trajectory addressing: uniform (SGPR) pointer to row `it` of the block's first stimulus + a 32-bit lane
    // offset, instead of one 64-bit VGPR address pair per stream
    const size_t blk = ((size_t)b * a.NB + s0) * T_ * M;
    const unsigned stim_stride = (unsigned)T_ * (unsigned)M;
    int cur = 0;
    for (int it = 0; it < T_; ++it) {
        T acc[NB][8];
        if constexpr (RL > 0) sw.template matvec<NB>(wlds, threadIdx.x, &rbuf[cur][0][0], cg, acc);
        else tile_matvec<T, RA, C, NB>(w, &rbuf[cur][0][0], cg, acc);
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const T u = reduce8_to_lane(acc[s], cg) + ex[s];
            T f, dfv;
            io_eval_grad(u, a.io, f, dfv);
            const T r1 = fma(eps, f - rc[s], rc[s]);          // (1 - eps) r + eps f(u)
            if (live[s]) {
                if (it >= a.skip) {
                    ta[s] += r1;
                    rpn[s] += (r1 > a.theta) ? (r1 - a.theta) : (T)0;
                    if (it > a.skip) { const T d = r1 - rc[s]; dp[s] += d * d; }
                }
                if (a.traj) {
                    const unsigned lo = (unsigned)s * stim_stride + (unsigned)myrow;
                    (a.traj + blk + (size_t)it * M)[lo] = r1;
                    (a.df + blk + (size_t)it * M)[lo] = dfv;
                }
                rc[s] = r1;
                rbuf[cur ^ 1][s][myslot] = r1;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    const T inv = (T)1 / (T)(T_ - a.skip);
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        if (!live[s]) continue;
        const size_t o = ((size_t)b * a.NB + s0 + s) * M + myrow;
        a.time_avg[o] = ta[s] * inv;
        a.dyn_row[o] = dp[s];
        a.rate_row[o] = rpn[s];
    }
}

// Reverse-time adjoint sweep.  The lane that finishes row j owns a_t[j]; the tile holds W^T.
template <typename T, int RA, int C, int RL, int NB, int MAXTHREADS, int MINWAVES>
__global__ void __launch_bounds__(MAXTHREADS, MINWAVES) gen_backward_kernel(GenBwdArgs<T> a) {
    constexpr int CP = SlabPad<C>::value;
    __shared__ __align__(16) T dbuf[2][NB][8 * CP];
    __shared__ __align__(16) T wlds[TileSplit<RA, C, RL>::lds_elems(MAXTHREADS)];
    const int M = a.M, N = a.M / 2, T_ = a.seqlen;
    const int ngroups = (a.NB + NB - 1) / NB;
    const int b = blockIdx.x / ngroups;
    const int s0 = (blockIdx.x % ngroups) * NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & 7, rg = lane >> 3;
    const int rowbase = (8 * wave + rg) * RA;

    T wt[RL == 0 ? RA : 1][RL == 0 ? C : 1];   // tile element (row j, col i) = W[i][j]
    SplitTile<T, RA, C, RL> sw;
    if constexpr (RL > 0) sw.template load<true>(a.W + (size_t)b * M * M, M, rowbase, cg * C, wlds, threadIdx.x);
    else tile_load<T, RA, C, true>(a.W + (size_t)b * M * M, M, rowbase, cg * C, wt);

    const int myrow = rowbase + cg;
    const bool fin = (cg < RA) && (myrow < M);
    const int myslot = (myrow / C) * CP + (myrow % C);
    const T eps = (myrow < N) ? a.eps_E : a.eps_I;
    const T inv = (T)1 / (T)(T_ - a.skip);
    bool live[NB];
    // uniform (SGPR) block pointers + 32-bit lane offsets (lo[s] = s*T*M + myrow)
    const size_t blk = ((size_t)b * a.NB + s0) * T_ * M;
    const T* trj = a.traj + blk;
    T* dlt = a.delta + blk;
    unsigned lo[NB];
    T gta[NB], carry[NB], xn[NB], xc[NB], xm[NB], dfc[NB], dsum[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        live[s] = fin && (s0 + s) < a.NB;
        lo[s] = live[s] ? (unsigned)s * (unsigned)T_ * (unsigned)M + (unsigned)myrow : 0u;
        gta[s] = live[s] ? a.g_time_avg[((size_t)b * a.NB + s0 + s) * M + myrow] * inv : (T)0;
        carry[s] = (T)0;
        dsum[s] = (T)0;
        xn[s] = (T)0;
        xc[s] = live[s] ? (trj + (size_t)(T_ - 1) * M)[lo[s]] : (T)0;              // x_T
        xm[s] = (live[s] && T_ >= 2) ? (trj + (size_t)(T_ - 2) * M)[lo[s]] : (T)0;  // x_{T-1}
        dfc[s] = live[s] ? (dlt + (size_t)(T_ - 1) * M)[lo[s]] : (T)0;             // f'(u_T)
        if (live[s]) (dlt + (size_t)(T_ - 1) * M)[lo[s]] = (T)0;   // slot T-1 of the shifted delta stays zero
    }
    for (int c = threadIdx.x; c < 2 * NB * 8 * CP; c += blockDim.x) (&dbuf[0][0][0])[c] = (T)0;
    __syncthreads();

    int cur = 0;
    for (int tau = T_; tau >= 1; --tau) {
        // prefetch what the NEXT iteration (tau-1) needs: x_{tau-2} and f'(u_{tau-1})
        T xmm[NB], dfn[NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            // opaque per iteration: otherwise trj + lo is hoisted as a loop-invariant 64-bit VGPR pair per stream
            // (registers this kernel does not have) instead of SGPR base + 32-bit VGPR offset addressing
            asm volatile("" : "+v"(lo[s]));
            // (x_{tau-2} is only used inside the penalty window, by the steps tau-1 .. tau-3 >= skip + 1)
            xmm[s] = (live[s] && tau >= 3 && tau >= a.skip + 3) ? (trj + (size_t)(tau - 3) * M)[lo[s]] : (T)0;
            dfn[s] = (live[s] && tau >= 2) ? (dlt + (size_t)(tau - 2) * M)[lo[s]] : (T)0;
        }
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            // direct gradient of the loss w.r.t. x_tau (window: tau >= skip+1)
            T g = (T)0;
            if (tau >= a.skip + 1) {
                g = gta[s] + ((xc[s] > a.theta) ? a.c_rate : (T)0);
                if (tau <= T_ - 1) g -= (T)2 * a.c_dyn * (xn[s] - xc[s]);
                if (tau >= a.skip + 2) g += (T)2 * a.c_dyn * (xc[s] - xm[s]);
            }
            const T at = g + carry[s];
            const T delta = eps * dfc[s] * at;
            carry[s] = fma(-eps, at, at);              // (1 - eps) a_t, + (W^T delta)[j] below
            dsum[s] += delta;                          // dL/d ext = sum_t delta_t  (u_t = W x_{t-1} + ext)
            if (live[s]) {
                dbuf[cur][s][myslot] = delta;
                if (tau >= 2) (dlt + (size_t)(tau - 2) * M)[lo[s]] = delta;   // shifted: pairs with x_{tau-1}
            }
        }
        __syncthreads();
        T acc[NB][8];
        if constexpr (RL > 0) sw.template matvec<NB>(wlds, threadIdx.x, &dbuf[cur][0][0], cg, acc);
        else tile_matvec<T, RA, C, NB>(wt, &dbuf[cur][0][0], cg, acc);
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            carry[s] += reduce8_to_lane(acc[s], cg);
            xn[s] = xc[s]; xc[s] = xm[s]; xm[s] = xmm[s]; dfc[s] = dfn[s];
        }
        cur ^= 1;
    }
    if (a.g_ext) {
#pragma unroll
        for (int s = 0; s < NB; ++s)
            if (live[s]) a.g_ext[((size_t)b * a.NB + s0 + s) * M + myrow] = dsum[s];
    }
}

// dL/dJ, dL/dD, dL/dS partial sums from dL/dW:  W = wnn * (sgn J + sgn D z), wnn = exp(-dx^2 / 2S^2)
// (make_w_batch.py:19-34).  One workgroup per (draw, p, q) block; out[b][pq][3] in fp64.
template <typename T>
__global__ void __launch_bounds__(256) jds_grad_kernel(const T* __restrict__ gW, const T* __restrict__ z,
                                                       JDSv<T> p, double* __restrict__ out, int N) {
    const int M = 2 * N;
    const int b = blockIdx.x >> 2, pq = blockIdx.x & 3, pp = pq >> 1, qq = pq & 1;
    const T inv_nm1 = (N > 1) ? (T)1 / (T)(N - 1) : (T)0;
    const T sgn = qq ? (T)-1 : (T)1;
    double sj = 0, sd = 0, ss = 0;
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
        const int i = e / N, j = e - i * N;
        const size_t o = ((size_t)b * M + pp * N + i) * M + qq * N + j;
        const T dx = (T)(i - j) * inv_nm1;
        const T wnn = exp(-(dx * dx) * p.inv2s2[pq]);
        const T g = gW[o], zz = z[o];
        sj += (double)(g * sgn * wnn);
        sd += (double)(g * sgn * wnn * zz);
        ss += (double)(g * wnn * (sgn * p.J[pq] + sgn * p.D[pq] * zz) * dx * dx * p.inv_s3[pq]);
    }
    __shared__ double red[3][256];
    red[0][threadIdx.x] = sj; red[1][threadIdx.x] = sd; red[2][threadIdx.x] = ss;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
            red[2][threadIdx.x] += red[2][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) out[((size_t)b * 4 + pq) * 3 + threadIdx.x] = red[threadIdx.x][0];
}

// ---------------------------------------------------------------------------------
// dispatch (same ladder as the tile solver)
// ---------------------------------------------------------------------------------
static constexpr int GEN_RA = 7;
static int gen_pick_c(int M, int elem_bytes) {
    const int need = (M + 7) / 8;
    const int ladder32[] = {4, 8, 13, 19, 25, 26};
    const int ladder64[] = {4, 8, 13};
    if (elem_bytes == 4) { for (int c : ladder32) if (need <= c) return c; }
    else                 { for (int c : ladder64) if (need <= c) return c; }
    return 0;
}
template <typename T> bool gen_supported(int M) {
    return (M % 2 == 0) && (gen_pick_c(M, (int)sizeof(T)) != 0 || gen_stream_supported(M));
}
template bool gen_supported<float>(int);
template bool gen_supported<double>(int);

template <typename T, int C, int RL, int NB, bool FWD, typename Args>
static hipError_t launch_gen_k(const Args& a, hipStream_t st) {
    constexpr int MAXW = (8 * C + 8 * GEN_RA - 1) / (8 * GEN_RA);
    constexpr int MINW = (sizeof(T) == 4) ? (RL > 0 ? 3 : 2) : 1;
    const int waves = (a.M + 8 * GEN_RA - 1) / (8 * GEN_RA);
    const int ngroups = (a.NB + NB - 1) / NB;
    if constexpr (FWD)
        hipLaunchKernelGGL((gen_forward_kernel<T, GEN_RA, C, RL, NB, 64 * MAXW, MINW>), dim3(a.B * ngroups),
                           dim3(64 * waves), 0, st, a);
    else
        hipLaunchKernelGGL((gen_backward_kernel<T, GEN_RA, C, RL, NB, 64 * MAXW, MINW>), dim3(a.B * ngroups),
                           dim3(64 * waves), 0, st, a);
    return hipGetLastError();
}
template <typename T, int C, bool FWD, typename Args>
static hipError_t launch_gen_nb(const Args& a, hipStream_t st) {
    // stimuli per workgroup chosen so that no instantiation spills (7*C tile + 8*NB accumulators + state)
    if constexpr (sizeof(T) == 4) {
        if constexpr (C <= 13) { if (a.NB >= 4) return launch_gen_k<T, C, 0, 4, FWD>(a, st); }
        // (two stimuli per workgroup at C >= 19 measured no faster at the C3 shape: the loop is VALU-issue bound)
        if constexpr (C <= 13) { if (a.NB >= 2) return launch_gen_k<T, C, 0, 2, FWD>(a, st); }
        // split VGPR/LDS residency (3 waves/SIMD, three workgroups per CU), as in the solver
        if constexpr (C == 25 || C == 26) return launch_gen_k<T, C, 2, 1, FWD>(a, st);
        if constexpr (C == 19) return launch_gen_k<T, C, 1, 1, FWD>(a, st);
    }
    return launch_gen_k<T, C, 0, 1, FWD>(a, st);
}
template <typename T, bool FWD, typename Args>
static hipError_t launch_gen_c(const Args& a, hipStream_t st) {
    const int c = gen_pick_c(a.M, (int)sizeof(T));
    if constexpr (sizeof(T) == 4) {
        switch (c) {
            case 4: return launch_gen_nb<T, 4, FWD>(a, st);
            case 8: return launch_gen_nb<T, 8, FWD>(a, st);
            case 13: return launch_gen_nb<T, 13, FWD>(a, st);
            case 19: return launch_gen_nb<T, 19, FWD>(a, st);
            case 25: return launch_gen_nb<T, 25, FWD>(a, st);
            case 26: return launch_gen_nb<T, 26, FWD>(a, st);
            default: break;
        }
    } else {
        switch (c) {
            case 4: return launch_gen_nb<T, 4, FWD>(a, st);
            case 8: return launch_gen_nb<T, 8, FWD>(a, st);
            case 13: return launch_gen_nb<T, 13, FWD>(a, st);
            default: break;
        }
    }
    // sizes without a register-resident instantiation: streaming kernels (ssn_gen_stream.hip)
    if constexpr (FWD) return launch_gen_forward_stream<T>(a, st);
    else return launch_gen_backward_stream<T>(a, st);
}

template <typename T> hipError_t launch_gen_forward(const GenFwdArgs<T>& a, hipStream_t st) { return launch_gen_c<T, true>(a, st); }
template <typename T> hipError_t launch_gen_backward(const GenBwdArgs<T>& a, hipStream_t st) { return launch_gen_c<T, false>(a, st); }
template hipError_t launch_gen_forward<float>(const GenFwdArgs<float>&, hipStream_t);
template hipError_t launch_gen_forward<double>(const GenFwdArgs<double>&, hipStream_t);
template hipError_t launch_gen_backward<float>(const GenBwdArgs<float>&, hipStream_t);
template hipError_t launch_gen_backward<double>(const GenBwdArgs<double>&, hipStream_t);

template <typename T>
hipError_t launch_jds_grad(const T* gW, const T* z, const T* jds12, double* out, int B, int N, hipStream_t st) {
    JDSv<T> p;
    for (int q = 0; q < 4; ++q) {
        p.J[q] = jds12[q]; p.D[q] = jds12[4 + q];
        const T s = jds12[8 + q];
        p.inv2s2[q] = (T)1 / ((T)2 * s * s);
        p.inv_s3[q] = (T)1 / (s * s * s);
    }
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL((jds_grad_kernel<T>), dim3(B * 4), dim3(256), 0, st, gW, z, p, out, N);
    return hipGetLastError();
}
template hipError_t launch_jds_grad<float>(const float*, const float*, const float*, double*, int, int, hipStream_t);
template hipError_t launch_jds_grad<double>(const double*, const double*, const double*, double*, int, int, hipStream_t);

}  // namespace ssn
