// Fixed-time SSN generator and its BPTT adjoint for ANY size (fp32 / fp64): the fallback behind the register-resident
// kernels of ssn_gen.hip / ssn_mfma.hip, which are instantiated up to 2N = 208 (fp32) / 104 (fp64).  Same
// recurrence, reductions and storage layout (networks/ssn.py:555-576, 598-633; see the header of ssn_gen.hip), W
// streamed from global memory / L2 every step as in solve_stream_kernel -- correct for every even 2N, not fast.
//
// One 256-thread workgroup per (draw, stimulus).
//   forward:  a wave per row, lanes stride the columns, wavefront shuffle sum; per-row window sums live in LDS.
//   backward: a thread owns columns j = t, t + 256, ... of W (rows of W^T): W[i][j] is read coalesced over the
//             threads for every i, delta_tau[i] is an LDS broadcast, no cross-lane reduction.
#include <hip/hip_runtime.h>
#include "ssn_device.h"
#include "ssn_host.h"

namespace ssn {

template <typename T>
__device__ __forceinline__ void stream_io_eval_grad(T v, const IoConsts<T>& c, T& f, T& df) {   // = io_eval_grad (ssn_gen.hip)
    if (!(v > (T)0)) { f = (v != v) ? v : (T)0; df = (T)0; return; }
    if (c.io_type == SSN_IO_POWER || v <= c.v0) { f = pow_rate(v, c.k, c.n); df = c.n * f / v; return; }
    if (c.io_type == SSN_IO_LINEAR) { f = c.soft + c.lin_slope * (v - c.v0); df = c.lin_slope; return; }
    const T th = tanh_pos(c.tanh_gain * (v - c.v0));
    f = c.soft + c.span * th;
    df = c.span_gain * ((T)1 - th * th);
}

template <typename T>
__device__ __forceinline__ T stream_wave_sum(T x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

template <typename T>
__global__ void __launch_bounds__(256) gen_forward_stream_kernel(GenFwdArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int M = a.M, N = a.M / 2, T_ = a.seqlen;
    T* rbuf = reinterpret_cast<T*>(smem_raw);          // [2][M]
    T* ta = rbuf + 2 * M;                              // [M] window sums
    T* dp = ta + M;
    T* rpn = dp + M;
    const int b = blockIdx.x / a.NB, s = blockIdx.x % a.NB;
    const size_t unit = (size_t)b * a.NB + s;
    const T* W = a.W + (size_t)b * M * M;
    const T* ext = a.ext + unit * M;
    T* traj = a.traj ? a.traj + unit * T_ * M : nullptr;
    T* df = a.df ? a.df + unit * T_ * M : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int c = threadIdx.x; c < 5 * M; c += blockDim.x) rbuf[c] = (T)0;
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < T_; ++it) {
        const T* rc = rbuf + cur * M;
        T* rn = rbuf + (cur ^ 1) * M;
        for (int i = wave; i < M; i += nwaves) {
            const T* wrow = W + (size_t)i * M;
            T part = (T)0;
            for (int j = lane; j < M; j += 64) part = fma(wrow[j], rc[j], part);
            const T u = stream_wave_sum(part) + ext[i];
            T f, dfv;
            stream_io_eval_grad(u, a.io, f, dfv);
            const T r0 = rc[i];
            const T r1 = fma(i < N ? a.eps_E : a.eps_I, f - r0, r0);          // (1 - eps) r + eps f(u)
            if (lane == 0) {
                if (it >= a.skip) {
                    ta[i] += r1;
                    rpn[i] += (r1 > a.theta) ? (r1 - a.theta) : (T)0;
                    if (it > a.skip) { const T d = r1 - r0; dp[i] += d * d; }
                }
                if (traj) { traj[(size_t)it * M + i] = r1; df[(size_t)it * M + i] = dfv; }
                rn[i] = r1;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    const T inv = (T)1 / (T)(T_ - a.skip);
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        a.time_avg[unit * M + i] = ta[i] * inv;
        a.dyn_row[unit * M + i] = dp[i];
        a.rate_row[unit * M + i] = rpn[i];
    }
}

constexpr int STREAM_JT = 8;                           // columns per thread: 2N <= 2048

template <typename T>
__global__ void __launch_bounds__(256) gen_backward_stream_kernel(GenBwdArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int M = a.M, N = a.M / 2, T_ = a.seqlen;
    T* dbuf = reinterpret_cast<T*>(smem_raw);          // [M] delta_tau
    const int b = blockIdx.x / a.NB, s = blockIdx.x % a.NB;
    const size_t unit = (size_t)b * a.NB + s;
    const T* W = a.W + (size_t)b * M * M;
    const T* trj = a.traj + unit * T_ * M;
    T* dlt = a.delta + unit * T_ * M;
    const T inv = (T)1 / (T)(T_ - a.skip);
    T gta[STREAM_JT], carry[STREAM_JT], dsum[STREAM_JT], dfc[STREAM_JT];
#pragma unroll
    for (int q = 0; q < STREAM_JT; ++q) {
        const int j = threadIdx.x + 256 * q;
        const bool ok = j < M;
        gta[q] = ok ? a.g_time_avg[unit * M + j] * inv : (T)0;
        carry[q] = dsum[q] = (T)0;
        dfc[q] = ok ? dlt[(size_t)(T_ - 1) * M + j] : (T)0;               // f'(u_T)
        if (ok) dlt[(size_t)(T_ - 1) * M + j] = (T)0;                      // slot T-1 of the shifted delta stays zero
    }
    for (int tau = T_; tau >= 1; --tau) {
#pragma unroll
        for (int q = 0; q < STREAM_JT; ++q) {
            const int j = threadIdx.x + 256 * q;
            if (j >= M) continue;
            T g = (T)0;                                 // direct gradient of the loss w.r.t. x_tau (window: tau >= skip+1)
            if (tau >= a.skip + 1) {
                const T xc = trj[(size_t)(tau - 1) * M + j];
                g = gta[q] + ((xc > a.theta) ? a.c_rate : (T)0);
                if (tau <= T_ - 1) g -= (T)2 * a.c_dyn * (trj[(size_t)tau * M + j] - xc);
                if (tau >= a.skip + 2) g += (T)2 * a.c_dyn * (xc - trj[(size_t)(tau - 2) * M + j]);
            }
            const T eps = (j < N) ? a.eps_E : a.eps_I;
            const T at = g + carry[q];
            const T delta = eps * dfc[q] * at;
            carry[q] = fma(-eps, at, at);               // (1 - eps) a_t, + (W^T delta)[j] below
            dsum[q] += delta;
            dbuf[j] = delta;
            if (tau >= 2) {
                dfc[q] = dlt[(size_t)(tau - 2) * M + j];                   // f'(u_{tau-1}), read before the slot is reused
                dlt[(size_t)(tau - 2) * M + j] = delta;                    // shifted: pairs with x_{tau-1}
            }
        }
        __syncthreads();
        T acc[STREAM_JT];
#pragma unroll
        for (int q = 0; q < STREAM_JT; ++q) acc[q] = (T)0;
        for (int i = 0; i < M; ++i) {
            const T d = dbuf[i];
            const T* wrow = W + (size_t)i * M;
#pragma unroll
            for (int q = 0; q < STREAM_JT; ++q) {
                const int j = threadIdx.x + 256 * q;
                if (j < M) acc[q] = fma(wrow[j], d, acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < STREAM_JT; ++q) carry[q] += acc[q];
        __syncthreads();
    }
    if (a.g_ext) {
#pragma unroll
        for (int q = 0; q < STREAM_JT; ++q) {
            const int j = threadIdx.x + 256 * q;
            if (j < M) a.g_ext[unit * M + j] = dsum[q];
        }
    }
}

bool gen_stream_supported(int M) { return M > 0 && (M % 2 == 0) && M <= 256 * STREAM_JT; }

template <typename T>
hipError_t launch_gen_forward_stream(const GenFwdArgs<T>& a, hipStream_t st) {
    if (!gen_stream_supported(a.M)) return hipErrorInvalidValue;
    const size_t smem = 5 * (size_t)a.M * sizeof(T);
    hipLaunchKernelGGL((gen_forward_stream_kernel<T>), dim3(a.B * a.NB), dim3(256), smem, st, a);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_gen_backward_stream(const GenBwdArgs<T>& a, hipStream_t st) {
    if (!gen_stream_supported(a.M)) return hipErrorInvalidValue;
    const size_t smem = (size_t)a.M * sizeof(T);
    hipLaunchKernelGGL((gen_backward_stream_kernel<T>), dim3(a.B * a.NB), dim3(256), smem, st, a);
    return hipGetLastError();
}
template hipError_t launch_gen_forward_stream<float>(const GenFwdArgs<float>&, hipStream_t);
template hipError_t launch_gen_forward_stream<double>(const GenFwdArgs<double>&, hipStream_t);
template hipError_t launch_gen_backward_stream<float>(const GenBwdArgs<float>&, hipStream_t);
template hipError_t launch_gen_backward_stream<double>(const GenBwdArgs<double>&, hipStream_t);

}  // namespace ssn
