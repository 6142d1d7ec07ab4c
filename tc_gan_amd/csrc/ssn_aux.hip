// Streaming helper kernels around the solver (gfx950): connectivity from noise,
// stimulus profiles, array form of the I/O nonlinearity, and a dot product.
// All are HBM-bound elementwise/reduction kernels.
#include <hip/hip_runtime.h>
#include "ssn_device.h"
#include "ssn_host.h"

namespace ssn {

// W[b][pN+i][qN+j] = exp(-(x_i-x_j)^2/(2 S_pq^2)) * sign_q * (J_pq + D_pq z[b][pN+i][qN+j])
// (gradient_expressions/make_w_batch.py:8-34; weight_gen.py:13-26).  8 B/element of HBM
// traffic; one thread per 4 consecutive columns when M % 4 == 0 (16-B accesses).
// pdev != nullptr: J, D, S as DEVICE T[12] (what an optimizer launch queued just before has left there -- the host has not
// seen the values yet); the constants derived from them by the host's own expressions, so that the same values give the same W
template <typename T, int VEC>
__global__ void __launch_bounds__(256) build_w_kernel(const T* __restrict__ z, T* __restrict__ W, JDS<T> p,
                                                      int N, long total_vec, const T* __restrict__ pdev) {
    if (pdev) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            p.J[q] = pdev[q];
            p.D[q] = pdev[4 + q];
            const T s = pdev[8 + q];
            T two_s2;
            {
#pragma clang fp contract(off)
                two_s2 = (T)2 * s * s;
            }
            p.inv2s2[q] = (T)1 / two_s2;
        }
    }
    const int M = 2 * N;
    const T inv_nm1 = (N > 1) ? (T)1 / (T)(N - 1) : (T)0;
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < total_vec; v += (long)gridDim.x * blockDim.x) {
        const long e0 = v * VEC;
        const int col0 = (int)(e0 % M);
        const int row = (int)((e0 / M) % M);
        T zin[VEC], wout[VEC];
        if constexpr (VEC == 4) {
            using V4 = T __attribute__((ext_vector_type(4)));
            const V4 q = *reinterpret_cast<const V4*>(z + e0);
            zin[0] = q.x; zin[1] = q.y; zin[2] = q.z; zin[3] = q.w;
        } else {
            zin[0] = z[e0];
        }
#pragma unroll
        for (int t = 0; t < VEC; ++t) wout[t] = w_from_z<T>(p, N, inv_nm1, row, col0 + t, zin[t]);
        if constexpr (VEC == 4) {
            using V4 = T __attribute__((ext_vector_type(4)));
            V4 q; q.x = wout[0]; q.y = wout[1]; q.z = wout[2]; q.w = wout[3];
            *reinterpret_cast<V4*>(W + e0) = q;
        } else {
            W[e0] = wout[0];
        }
    }
}

template <typename T>
hipError_t launch_build_w(const T* z, const T* jds12, T* W, int B, int N, hipStream_t st, const T* jds12_dev) {
    JDS<T> p = {};
    for (int q = 0; q < 4 && !jds12_dev; ++q) {
        p.J[q] = jds12[q];
        p.D[q] = jds12[4 + q];
        p.inv2s2[q] = (T)1 / ((T)2 * jds12[8 + q] * jds12[8 + q]);
    }
    const int M = 2 * N;
    const long total = (long)B * M * M;
    if (total == 0) return hipSuccess;
    const bool vec4 = (M % 4 == 0) && (((uintptr_t)z | (uintptr_t)W) % (4 * sizeof(T)) == 0);
    const long nvec = vec4 ? total / 4 : total;
    const int blocks = (int)((nvec + 255) / 256 < 256 * 8 ? (nvec + 255) / 256 : 256 * 8);
    if (vec4) hipLaunchKernelGGL((build_w_kernel<T, 4>), dim3(blocks), dim3(256), 0, st, z, W, p, N, nvec, jds12_dev);
    else      hipLaunchKernelGGL((build_w_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, z, W, p, N, nvec, jds12_dev);
    return hipGetLastError();
}
template hipError_t launch_build_w<float>(const float*, const float*, float*, int, int, hipStream_t, const float*);
template hipError_t launch_build_w<double>(const double*, const double*, double*, int, int, hipStream_t, const double*);

// ext[b][s][pN+i] = c * sig((x_i + bw/2)/l) * sig((bw/2 - x_i)/l)   (stimuli.py:3-10)
template <typename T>
__device__ __forceinline__ void stimulus_body(const T* __restrict__ bw, const T* __restrict__ con, T inv_l,
                                              const T* __restrict__ amp, int NB, T* __restrict__ ext, int N, long total,
                                              long blk, long nblk) {
    const int M = 2 * N;
    const T step = (N > 1) ? (T)1 / (T)(N - 1) : (T)0;
    for (long e = blk * 256L + threadIdx.x; e < total; e += nblk * 256L) {
        const int m = (int)(e % M);
        const long bs = e / M;
        const int i = m >= N ? m - N : m;
        const T x = (T)-0.5 + step * (T)i;
        const T hb = bw[bs] * (T)0.5;
        const T s1 = (T)1 / ((T)1 + exp(-(x + hb) * inv_l));
        const T s2 = (T)1 / ((T)1 + exp(-(hb - x) * inv_l));
        // heterogeneous input (networks/ssn.py:645-700): stimulus * (1 + v_pop * z_in), amp[b][m] given per draw
        const T gain = amp ? amp[(bs / NB) * M + m] : (T)1;
        ext[e] = gain * con[bs] * s1 * s2;
    }
}
template <typename T>
__global__ void __launch_bounds__(256) stimulus_kernel(const T* __restrict__ bw, const T* __restrict__ con, T inv_l,
                                                       const T* __restrict__ amp, int NB,
                                                       T* __restrict__ ext, int N, long total) {
    stimulus_body<T>(bw, con, inv_l, amp, NB, ext, N, total, blockIdx.x, gridDim.x);
}
template <typename T>
hipError_t launch_stimulus(const T* bw, const T* con, T smooth, const T* amp, T* ext, int B, int NB, int N, hipStream_t st) {
    const long total = (long)B * NB * 2 * N;
    if (total == 0) return hipSuccess;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL((stimulus_kernel<T>), dim3(blocks), dim3(256), 0, st, bw, con, (T)1 / smooth, amp, NB, ext, N, total);
    return hipGetLastError();
}
// The heterogeneous-input stimulus with its amplification formed in the launch: gain = 1 + v z_in (networks/ssn.py:679-686),
// the two roundings of the torch expression `1 + vs[None, :] * zin` that used to be three small launches in front of this one.
// v: per neuron [2N] (nv = 2N), per population [2] (first N neurons v[0], the others v[1]) or one value [1].
__global__ void __launch_bounds__(256) stimulus_hetero_kernel(const float* __restrict__ bw, const float* __restrict__ con, float inv_l,
                                                              const float* __restrict__ zin, const float* __restrict__ v, int nv,
                                                              int NB, float* __restrict__ ext, int N, long total) {
    const int M = 2 * N;
    const float step = (N > 1) ? 1.f / (float)(N - 1) : 0.f;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
        const int m = (int)(e % M);
        const long bs = e / M;
        const int i = m >= N ? m - N : m;
        const float x = -0.5f + step * (float)i;
        const float hb = bw[bs] * 0.5f;
        const float s1 = 1.f / (1.f + exp(-(x + hb) * inv_l));
        const float s2 = 1.f / (1.f + exp(-(hb - x) * inv_l));
        const float vm = v[nv == M ? m : (nv == 2 ? (m >= N ? 1 : 0) : 0)];
        float t = vm * zin[(bs / NB) * M + m];
        asm volatile("" : "+v"(t));                      // (the product is rounded before the addition: no contraction into one fma)
        const float gain = 1.f + t;
        ext[e] = gain * con[bs] * s1 * s2;
    }
}
hipError_t launch_stimulus_hetero(const float* bw, const float* con, float smooth, const float* zin, const float* v, int nv, float* ext,
                                  int B, int NB, int N, hipStream_t st) {
    const long total = (long)B * NB * 2 * N;
    if (total == 0) return hipSuccess;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(stimulus_hetero_kernel, dim3(blocks), dim3(256), 0, st, bw, con, 1.f / smooth, zin, v, nv, NB, ext, N, total);
    return hipGetLastError();
}
template hipError_t launch_stimulus<float>(const float*, const float*, float, const float*, float*, int, int, int, hipStream_t);
template hipError_t launch_stimulus<double>(const double*, const double*, double, const double*, double*, int, int, int, hipStream_t);

template <typename T>
__global__ void __launch_bounds__(256) io_eval_kernel(const T* __restrict__ v, T* __restrict__ out, long count, IoConsts<T> io) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < count; e += (long)gridDim.x * blockDim.x)
        out[e] = io_eval(v[e], io);
}
template <typename T>
hipError_t launch_io_eval(const T* v, T* out, long count, const IoConsts<T>& io, hipStream_t st) {
    if (count == 0) return hipSuccess;
    const int blocks = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
    hipLaunchKernelGGL((io_eval_kernel<T>), dim3(blocks), dim3(256), 0, st, v, out, count, io);
    return hipGetLastError();
}
template hipError_t launch_io_eval<float>(const float*, float*, long, const IoConsts<float>&, hipStream_t);
template hipError_t launch_io_eval<double>(const double*, double*, long, const IoConsts<double>&, hipStream_t);

// Single-workgroup dot product, sequential-in-chunks (ssnode.c:10-19 `dot`).
template <typename T>
__global__ void __launch_bounds__(256) dot_kernel(const T* __restrict__ x, const T* __restrict__ y, T* out, int dim) {
    __shared__ T part[256];
    T s = (T)0;
    for (int j = threadIdx.x; j < dim; j += 256) s = fma(x[j], y[j], s);
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0];
}
template <typename T>
hipError_t launch_dot(const T* x, const T* y, T* out, int dim, hipStream_t st) {
    hipLaunchKernelGGL((dot_kernel<T>), dim3(1), dim3(256), 0, st, x, y, out, dim);
    return hipGetLastError();
}
template hipError_t launch_dot<float>(const float*, const float*, float*, int, hipStream_t);
template hipError_t launch_dot<double>(const double*, const double*, double*, int, hipStream_t);


// ---------------------------------------------------------------------------------
// Moment matching (networks/moment_matching.py:91-243): per tuning-curve channel d the sample mean and
// (population) variance over the minibatch, the weighted squared distance to the data moments, and its
// gradient w.r.t. every generated tuning curve.  x[B][D] row-major.  Sums in fp64.
// sums[2][D] = (sum_b x, sum_b x^2): produced per rank, all-reduced by the caller when the minibatch is
// sharded over GPUs, then consumed with the GLOBAL batch size.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) moment_sums_kernel(const float* __restrict__ x, int B, int D, double* __restrict__ sums) {
    const int d = blockIdx.x;
    double s1 = 0, s2 = 0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const double v = (double)x[(size_t)b * D + d];
        s1 += v; s2 += v * v;
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) { red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sums[d] = red[0][0]; sums[D + d] = red[1][0]; }
}

// loss L0 = mean over the (2, D) array of w * (data - gen)^2 ;  gen = (m, s), m = S1/Bg, s = S2/Bg - m^2
// dL0/dx[b][d] = (1 / (2 D)) * ( 2 w0 (m - mu) / Bg + 2 w1 (s - sigma) * 2 (x - m) / Bg )
// out[0] = L0, out[1 + d] = m_d, out[1 + D + d] = s_d   (fp64)
__global__ void __launch_bounds__(256) moment_loss_grad_kernel(const float* __restrict__ x, const double* __restrict__ sums,
                                                               double Bg, const double* __restrict__ data_moments,
                                                               const double* __restrict__ weights, int B, int D,
                                                               float* __restrict__ gx, double* __restrict__ out) {
    const int d = blockIdx.x;
    const double m = sums[d] / Bg, s = sums[D + d] / Bg - m * m;
    const double em = m - data_moments[d], es = s - data_moments[D + d];
    const double w0 = weights[d], w1 = weights[D + d];
    const double c0 = w0 * em / (Bg * D), c1 = 2.0 * w1 * es / (Bg * D);
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const double xv = (double)x[(size_t)b * D + d];
        gx[(size_t)b * D + d] = (float)(c0 + c1 * (xv - m));
    }
    if (threadIdx.x == 0) { out[1 + d] = m; out[1 + D + d] = s; }
}
// L0 = mean over the (2, D) array of w (data - gen)^2, summed in channel order by ONE workgroup (a tree with a fixed
// shape: the value does not depend on which channel's workgroup finished first, as an atomic sum would)
__global__ void __launch_bounds__(256) moment_loss_sum_kernel(const double* __restrict__ data_moments,
                                                              const double* __restrict__ weights, int D, double* __restrict__ out) {
    __shared__ double red[256];
    double t = 0.0;
    for (int d = threadIdx.x; d < D; d += 256) {
        const double em = out[1 + d] - data_moments[d], es = out[1 + D + d] - data_moments[D + d];
        t += weights[d] * em * em + weights[D + d] * es * es;
    }
    red[threadIdx.x] = t;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] / (2.0 * D);
}

hipError_t launch_moment_sums(const float* x, int B, int D, double* sums, hipStream_t st) {
    if (D == 0) return hipSuccess;
    hipLaunchKernelGGL(moment_sums_kernel, dim3(D), dim3(256), 0, st, x, B, D, sums);
    return hipGetLastError();
}
hipError_t launch_moment_loss_grad(const float* x, const double* sums, double Bg, const double* data_moments,
                                   const double* weights, int B, int D, float* gx, double* out, hipStream_t st) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st);
    if (e != hipSuccess || D == 0) return e;
    hipLaunchKernelGGL(moment_loss_grad_kernel, dim3(D), dim3(256), 0, st, x, sums, Bg, data_moments, weights, B, D, gx, out);
    hipLaunchKernelGGL(moment_loss_sum_kernel, dim3(1), dim3(256), 0, st, data_moments, weights, D, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// Counter-based uniform noise (Philox4x32-10, Salmon et al. 2011): out[i] = u(seed, offset + i) in [0, 1).
// Element g of the stream is word g % 4 of the Philox block with counter (g / 4, 0, 0, 0) and key (seed lo, seed hi),
// so any slice of the stream can be produced by itself: rank r of a data-parallel run fills only ITS rows of the
// global z tensor and the job as a whole draws exactly the numbers a single process draws.  HBM-bound (4 or 8 B
// written per element).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <typename T>
__global__ void __launch_bounds__(256) philox_uniform_kernel(unsigned long long seed, unsigned long long offset, T* __restrict__ out,
                                                             unsigned long long n) {
    const unsigned long long first_blk = offset >> 2, last_blk = (offset + n + 3) >> 2;      // Philox blocks touched
    for (unsigned long long blk = first_blk + blockIdx.x * 256ull + threadIdx.x; blk < last_blk; blk += gridDim.x * 256ull) {
        unsigned w[4];
        philox4x32_10((unsigned)blk, (unsigned)(blk >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long g = 4 * blk + j;
            if (g >= offset && g < offset + n) out[g - offset] = (T)((float)(w[j] >> 8) * (1.0f / 16777216.0f));
        }
    }
}
template <typename T>
hipError_t launch_philox_uniform(unsigned long long seed, unsigned long long offset, T* out, unsigned long long n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    unsigned long long blocks = ((n + 3) / 4 + 1 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((philox_uniform_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, seed, offset, out, n);
    return hipGetLastError();
}
// W straight from the noise stream: element e of this call's z is stream element offset + e (the numbers
// philox_uniform_kernel writes for the same seed and offset), W as build_w_kernel forms it; z itself is written only when
// the caller keeps it (the generator step's chain rule needs it, the five critic-phase forwards of a GAN iteration do not):
// 4 instead of 12 bytes of HBM traffic per element and one launch instead of two.  One thread per 4 consecutive columns.
template <typename T>
__device__ __forceinline__ void build_w_philox_body(unsigned long long seed, unsigned long long offset, T* __restrict__ W,
                                                    T* __restrict__ zout, const JDS<T>& p, int N, long total_vec, long blk0, long nblk) {
    const int M = 2 * N;
    const T inv_nm1 = (N > 1) ? (T)1 / (T)(N - 1) : (T)0;
    const unsigned mis = (unsigned)(offset & 3ull);           // (uniform) the 4 elements straddle two Philox blocks when != 0
    for (long v = blk0 * 256L + threadIdx.x; v < total_vec; v += nblk * 256L) {
        const long e0 = v * 4;
        const unsigned long long g0 = offset + (unsigned long long)e0, blk = g0 >> 2;
        unsigned w[8];
        philox4x32_10((unsigned)blk, (unsigned)(blk >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
        if (mis) philox4x32_10((unsigned)(blk + 1), (unsigned)((blk + 1) >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w + 4);
        int col = (int)(e0 % M), row = (int)((e0 / M) % M);       // (M even: 4 | M * M, so a group of 4 never leaves its draw,
        T zin[4], wout[4];                                          //  but with M % 4 == 2 it may run over the end of a row)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned bits = mis == 0 ? w[t] : (mis == 1 ? w[t + 1] : (mis == 2 ? w[t + 2] : w[t + 3]));
            zin[t] = (T)((float)(bits >> 8) * (1.0f / 16777216.0f));
            wout[t] = w_from_z<T>(p, N, inv_nm1, row, col, zin[t]);
            if (++col == M) { col = 0; if (++row == M) row = 0; }
        }
        using V4 = T __attribute__((ext_vector_type(4)));
        V4 q; q.x = wout[0]; q.y = wout[1]; q.z = wout[2]; q.w = wout[3];
        *reinterpret_cast<V4*>(W + e0) = q;
        if (zout) {
            V4 zq; zq.x = zin[0]; zq.y = zin[1]; zq.z = zin[2]; zq.w = zin[3];
            *reinterpret_cast<V4*>(zout + e0) = zq;
        }
    }
}
template <typename T>
__global__ void __launch_bounds__(256) build_w_philox_kernel(unsigned long long seed, unsigned long long offset, T* __restrict__ W,
                                                             T* __restrict__ zout, JDS<T> p, int N, long total_vec) {
    build_w_philox_body<T>(seed, offset, W, zout, p, N, total_vec, blockIdx.x, gridDim.x);
}
// M even and 16-byte aligned outputs (the caller checks; other shapes take the two-kernel path)
template <typename T>
hipError_t launch_build_w_philox(unsigned long long seed, unsigned long long offset, const T* jds12, T* W, T* zout, int B, int N,
                                 hipStream_t st) {
    JDS<T> p;
    for (int q = 0; q < 4; ++q) {
        p.J[q] = jds12[q];
        p.D[q] = jds12[4 + q];
        p.inv2s2[q] = (T)1 / ((T)2 * jds12[8 + q] * jds12[8 + q]);
    }
    const int M = 2 * N;
    const long total = (long)B * M * M;
    if (total == 0) return hipSuccess;
    if ((((uintptr_t)W | (uintptr_t)zout) % (4 * sizeof(T))) != 0) return hipErrorInvalidValue;      // (M = 2N: 4 | M * M)
    const long nvec = total / 4;
    const int blocks = (int)((nvec + 255) / 256 < 256 * 16 ? (nvec + 255) / 256 : 256 * 16);
    hipLaunchKernelGGL((build_w_philox_kernel<T>), dim3(blocks), dim3(256), 0, st, seed, offset, W, zout, p, N, nvec);
    return hipGetLastError();
}
template hipError_t launch_build_w_philox<float>(unsigned long long, unsigned long long, const float*, float*, float*, int, int, hipStream_t);
template hipError_t launch_build_w_philox<double>(unsigned long long, unsigned long long, const double*, double*, double*, int, int, hipStream_t);

// The heterogeneous-input SSN's per-neuron input variability in one launch (networks/ssn.py:679-720): element i of the
// draw gets z = +1 / -1 (u < 0.5, `dist_in = 'bernoulli'`) or 2 u - 1 ('uniform') from the same stream element u as
// philox_uniform_kernel would give it, and amp = 1 + v[i % M] * z (v = the population's input variability per neuron).
template <typename T>
__global__ void __launch_bounds__(256) philox_amp_kernel(unsigned long long seed, unsigned long long offset, const T* __restrict__ v,
                                                         T* __restrict__ zin, T* __restrict__ amp, unsigned long long n, int M,
                                                         int bernoulli) {
    const unsigned long long first_blk = offset >> 2, last_blk = (offset + n + 3) >> 2;
    for (unsigned long long blk = first_blk + blockIdx.x * 256ull + threadIdx.x; blk < last_blk; blk += gridDim.x * 256ull) {
        unsigned w[4];
        philox4x32_10((unsigned)blk, (unsigned)(blk >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long g = 4 * blk + j;
            if (g >= offset && g < offset + n) {
                const unsigned long long i = g - offset;
                const T u = (T)((float)(w[j] >> 8) * (1.0f / 16777216.0f));
                const T z = bernoulli ? (u < (T)0.5 ? (T)1 : (T)-1) : u * (T)2 - (T)1;
                zin[i] = z;
                amp[i] = (T)1 + v[i % (unsigned long long)M] * z;
            }
        }
    }
}
template <typename T>
hipError_t launch_philox_amp(unsigned long long seed, unsigned long long offset, const T* v, T* zin, T* amp,
                             unsigned long long n, int M, int bernoulli, hipStream_t st) {
    if (n == 0) return hipSuccess;
    unsigned long long blocks = ((n + 3) / 4 + 1 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((philox_amp_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, seed, offset, v, zin, amp, n, M, bernoulli);
    return hipGetLastError();
}
template hipError_t launch_philox_amp<float>(unsigned long long, unsigned long long, const float*, float*, float*, unsigned long long, int, int, hipStream_t);
template hipError_t launch_philox_amp<double>(unsigned long long, unsigned long long, const double*, double*, double*, unsigned long long, int, int, hipStream_t);

// The inputs of a device-noise forward in ONE launch (ssn_gen_inputs_philox_f32; three launches until round 4): the grid is
// cut in three ranges of workgroups that run the bodies of philox_amp_kernel (heterogeneous-input noise z and 1 + v z),
// stimulus_kernel and build_w_philox_kernel.  The stimulus needs 1 + v z of its own (draw, neuron): it forms the number
// again from the same stream element (ten Philox rounds per element, on a kernel of a few microseconds) instead of waiting
// for the first range -- same expression, same bits as the stored `amp`.
__global__ void __launch_bounds__(256) gen_inputs_kernel(GenInputsArgs a) {
    const long blk = blockIdx.x;
    const int M = 2 * a.N;
    if (blk < a.nb_w) {
        build_w_philox_body<float>(a.seed, a.off_z, a.W, a.z, a.p, a.N, a.total_vec, blk, a.nb_w);
    } else if (blk < a.nb_w + a.nb_s) {
        const long b0 = blk - a.nb_w;
        if (!a.v) { stimulus_body<float>(a.bw, a.con, a.inv_l, nullptr, a.NB, a.ext, a.N, a.total_s, b0, a.nb_s); return; }
        const float step = (a.N > 1) ? 1.f / (float)(a.N - 1) : 0.f;
        for (long e = b0 * 256L + threadIdx.x; e < a.total_s; e += a.nb_s * 256L) {
            const int m = (int)(e % M);
            const long bs = e / M;
            const int i = m >= a.N ? m - a.N : m;
            const float x = -0.5f + step * (float)i;
            const float hb = a.bw[bs] * 0.5f;
            const float s1 = 1.f / (1.f + exp(-(x + hb) * a.inv_l));
            const float s2 = 1.f / (1.f + exp(-(hb - x) * a.inv_l));
            const unsigned long long g = a.off_zin + (unsigned long long)((bs / a.NB) * M + m);
            unsigned w[4];
            philox4x32_10((unsigned)(g >> 2), (unsigned)(g >> 34), 0u, 0u, (unsigned)a.seed, (unsigned)(a.seed >> 32), w);
            const float u = (float)(w[g & 3ull] >> 8) * (1.0f / 16777216.0f);
            const float z = a.bernoulli ? (u < 0.5f ? 1.f : -1.f) : u * 2.f - 1.f;
            const float gain = 1.f + a.v[m] * z;
            a.ext[e] = gain * a.con[bs] * s1 * s2;
        }
    } else {
        const long b0 = blk - a.nb_w - a.nb_s, nblk = (long)gridDim.x - a.nb_w - a.nb_s;
        const unsigned long long n = (unsigned long long)a.B * M, first_blk = a.off_zin >> 2, last_blk = (a.off_zin + n + 3) >> 2;
        for (unsigned long long pb = first_blk + (unsigned long long)b0 * 256ull + threadIdx.x; pb < last_blk; pb += (unsigned long long)nblk * 256ull) {
            unsigned w[4];
            philox4x32_10((unsigned)pb, (unsigned)(pb >> 32), 0u, 0u, (unsigned)a.seed, (unsigned)(a.seed >> 32), w);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned long long g = 4 * pb + j;
                if (g >= a.off_zin && g < a.off_zin + n) {
                    const unsigned long long i = g - a.off_zin;
                    const float u = (float)(w[j] >> 8) * (1.0f / 16777216.0f);
                    const float z = a.bernoulli ? (u < 0.5f ? 1.f : -1.f) : u * 2.f - 1.f;
                    a.zin[i] = z;
                    a.amp[i] = 1.f + a.v[i % (unsigned long long)M] * z;
                }
            }
        }
    }
}
hipError_t launch_gen_inputs(GenInputsArgs a, const float* jds12, hipStream_t st) {
    for (int q = 0; q < 4; ++q) {
        a.p.J[q] = jds12[q];
        a.p.D[q] = jds12[4 + q];
        a.p.inv2s2[q] = 1.f / (2.f * jds12[8 + q] * jds12[8 + q]);
    }
    const int M = 2 * a.N;
    const long total = (long)a.B * M * M;
    if (total == 0) return hipSuccess;
    if ((((uintptr_t)a.W | (uintptr_t)a.z) % 16) != 0) return hipErrorInvalidValue;
    a.total_vec = total / 4;
    a.total_s = (long)a.B * a.NB * M;
    a.nb_w = (a.total_vec + 255) / 256 < 256 * 16 ? (a.total_vec + 255) / 256 : 256 * 16;       // (the grids of the three launches)
    a.nb_s = a.total_s ? ((a.total_s + 255) / 256 < 2048 ? (a.total_s + 255) / 256 : 2048) : 0;
    long nb_a = 0;
    if (a.v) {
        nb_a = (long)((((unsigned long long)a.B * M + 3) / 4 + 1 + 255) / 256);
        if (nb_a > 8192) nb_a = 8192;
    }
    hipLaunchKernelGGL(gen_inputs_kernel, dim3((unsigned)(a.nb_w + a.nb_s + nb_a)), dim3(256), 0, st, a);
    return hipGetLastError();
}

template hipError_t launch_philox_uniform<float>(unsigned long long, unsigned long long, float*, unsigned long long, hipStream_t);
template hipError_t launch_philox_uniform<double>(unsigned long long, unsigned long long, double*, unsigned long long, hipStream_t);

// ---------------------------------------------------------------------------------
// The two penalty means of the fixed-time generator (networks/ssn.py:626, 632): out[0] = scale_dyn * sum(dyn_row),
// out[1] = scale_rate * sum(rate_row), fp64, in ONE launch: every workgroup writes its fp64 partial sums to `ws`,
// takes a ticket, and the workgroup that draws the last ticket adds the partials as a fixed tree over the workgroup
// index (deterministic) and resets the ticket.  ws: [2 * PEN_BLOCKS] doubles + the ticket (one int, zero before the
// first launch).
// ---------------------------------------------------------------------------------
constexpr int PEN_BLOCKS = 256;
// (optional rider, round 4: the conditional prober's gather tc[k][s] = time_avg[ids[k]][s][probes[k]] (cwgan.py:91-98) of the
// same forward, which used to be a launch of its own between this one and the critic -- plain copies, spread over the grid)
template <typename T>
struct ProbeGather { const T* time_avg; const long* ids; const long* probes; T* tc; long count; int NB, M; };
template <typename T>
__global__ void __launch_bounds__(256) penalty_means_kernel(const T* __restrict__ dyn, const T* __restrict__ rate, long n,
                                                            double scale_dyn, double scale_rate, double* __restrict__ ws,
                                                            double* __restrict__ out, ProbeGather<T> pg) {
    __shared__ double red[2][256];
    __shared__ int last;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < pg.count; e += (long)gridDim.x * 256L) {
        const long k = e / pg.NB;
        const int sidx = (int)(e % pg.NB);
        pg.tc[e] = pg.time_avg[((size_t)pg.ids[k] * pg.NB + sidx) * pg.M + pg.probes[k]];
    }
    double s0 = 0.0, s1 = 0.0;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256L) { s0 += (double)dyn[e]; s1 += (double)rate[e]; }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) { red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; }
        __syncthreads();
    }
    int* ticket = reinterpret_cast<int*>(ws + 2 * PEN_BLOCKS);
    if (threadIdx.x == 0) {
        ws[2 * blockIdx.x] = red[0][0]; ws[2 * blockIdx.x + 1] = red[1][0];
        __threadfence();                                           // partials visible before the ticket
        last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (!last) return;
    __threadfence();                                               // acquire: see every workgroup's partials
    // the last workgroup adds the (at most 256) partials as a fixed tree: thread b takes workgroup b's pair
    const bool have = threadIdx.x < gridDim.x;
    red[0][threadIdx.x] = have ? __builtin_nontemporal_load(ws + 2 * threadIdx.x) : 0.0;
    red[1][threadIdx.x] = have ? __builtin_nontemporal_load(ws + 2 * threadIdx.x + 1) : 0.0;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) { red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = red[0][0] * scale_dyn; out[1] = red[1][0] * scale_rate;
        *ticket = 0;
    }
}
// out[0] = mean(d[0:ng]) - mean(d[ng:ng+nd])   (one workgroup, fixed order: the critic's "accuracy", cwgan.py:139-147)
__global__ void __launch_bounds__(256) mean_diff_kernel(const float* __restrict__ d, float* __restrict__ out, int ng, int nd) {
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
    if (threadIdx.x == 0) out[0] = (ng ? red[0][0] / ng : 0.f) - (nd ? red[1][0] / nd : 0.f);
}
// head of the critic step's scalar record: tail[0..1] = the forward's two penalties (fp64 -> fp32), tail[2] = the loss
__global__ void step_head_kernel(const double* __restrict__ pens, const float* __restrict__ stats, float* __restrict__ tail) {
    tail[0] = pens ? (float)pens[0] : 0.f;
    tail[1] = pens ? (float)pens[1] : 0.f;
    tail[2] = stats[3];
}
hipError_t launch_step_head(const double* pens, const float* stats, float* tail, hipStream_t st) {
    hipLaunchKernelGGL(step_head_kernel, dim3(1), dim3(1), 0, st, pens, stats, tail);
    return hipGetLastError();
}
hipError_t launch_mean_diff(const float* d, int ng, int nd, float* out, hipStream_t st) {
    hipLaunchKernelGGL(mean_diff_kernel, dim3(1), dim3(256), 0, st, d, out, ng, nd);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_penalty_means(const T* dyn, const T* rate, long n, double scale_dyn, double scale_rate, double* ws,
                                double* out, hipStream_t st, const T* time_avg, const long* ids, const long* probes, T* tc,
                                int nsamp, int NB, int M) {
    long blocks = (n + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > PEN_BLOCKS) blocks = PEN_BLOCKS;
    ProbeGather<T> pg{time_avg, ids, probes, tc, tc ? (long)nsamp * NB : 0L, NB, M};
    hipLaunchKernelGGL((penalty_means_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, dyn, rate, n, scale_dyn, scale_rate, ws, out, pg);
    return hipGetLastError();
}
template hipError_t launch_penalty_means<float>(const float*, const float*, long, double, double, double*, double*, hipStream_t,
                                                const float*, const long*, const long*, float*, int, int, int);
template hipError_t launch_penalty_means<double>(const double*, const double*, long, double, double, double*, double*, hipStream_t,
                                                 const double*, const long*, const long*, double*, int, int, int);

// ---------------------------------------------------------------------------------
// Small per-step helpers of the GAN loop (one launch each instead of three to five element-wise ones).
// segment_sqnorms: out[t] = sum of x[i]^2 over i in [bounds[t], bounds[t + 1]) -- the per-tensor parameter statistics that
// recorders.py:275-311 logs after every critic step; fp64 partial sums, fixed order (below).
// interpolate: xp = eps * xd + (1 - eps) * xg per row (cwgan.py:476-481).
// ---------------------------------------------------------------------------------
// Two stages, both in a fixed order: stage 1 cuts every tensor into SQ_CHUNKS contiguous chunks, one workgroup per
// (chunk, tensor) -- 64 x n workgroups instead of n, so a 0.8 M-parameter critic uses the whole chip instead of 7 CUs
// (round 2: one workgroup per tensor, 255 us per call) -- and writes its fp64 sum to ws[tensor][chunk]; stage 2 adds the
// chunks of a tensor in chunk order.  No atomics: the same bits every run.
constexpr int SQ_CHUNKS = 64;
__global__ void __launch_bounds__(256) segment_sqnorms_kernel(const float* __restrict__ x, const long* __restrict__ bounds,
                                                             double* __restrict__ ws) {
    __shared__ double red[256];
    const long lo = bounds[blockIdx.y], hi = bounds[blockIdx.y + 1];
    const long per = ((hi - lo + SQ_CHUNKS - 1) / SQ_CHUNKS + 3) & ~3L;          // chunk length, a multiple of 4 elements
    const long c0 = lo + per * blockIdx.x, c1 = (c0 + per < hi) ? c0 + per : hi;
    double s = 0.0;
    for (long i = c0 + threadIdx.x; i < c1; i += 256) { const double v = (double)x[i]; s += v * v; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) ws[(size_t)blockIdx.y * SQ_CHUNKS + blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(64) segment_sqnorms_finish_kernel(const double* __restrict__ ws, float* __restrict__ out, int n) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n) return;
    double s = 0.0;
    for (int c = 0; c < SQ_CHUNKS; ++c) s += ws[(size_t)t * SQ_CHUNKS + c];
    out[t] = (float)s;
}
// The end of a one-call critic step in ONE launch (three until round 4: mean_diff_kernel, segment_sqnorms_finish_kernel,
// step_head_kernel): tail[0..1] = the forward's penalties, tail[2] = the loss, tail[3] = mean D(xg) - mean D(xd) of the
// updated critic (fixed order, as mean_diff_kernel), tail[4 + t] = sum of squares of parameter tensor t (chunks in chunk
// order, as segment_sqnorms_finish_kernel): the same bits as the three launches.
__global__ void __launch_bounds__(256) step_finish_kernel(const float* __restrict__ d, int ng, int nd, const double* __restrict__ pens,
                                                          const float* __restrict__ stats, const double* __restrict__ ws, int nseg,
                                                          float* __restrict__ tail) {
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
    if (threadIdx.x == 0) {
        tail[0] = pens ? (float)pens[0] : 0.f;
        tail[1] = pens ? (float)pens[1] : 0.f;
        tail[2] = stats[3];
        tail[3] = (ng ? red[0][0] / ng : 0.f) - (nd ? red[1][0] / nd : 0.f);
    }
    for (int t = threadIdx.x; t < nseg; t += 256) {
        double s = 0.0;
        for (int c = 0; c < SQ_CHUNKS; ++c) s += ws[(size_t)t * SQ_CHUNKS + c];
        tail[4 + t] = (float)s;
    }
}
hipError_t launch_step_finish(const float* params, const long* bounds, int nseg, double* ws, const float* dvals, int ng, int nd,
                              const double* pens, const float* stats, float* tail, hipStream_t st) {
    if (nseg > 0) hipLaunchKernelGGL(segment_sqnorms_kernel, dim3(SQ_CHUNKS, nseg), dim3(256), 0, st, params, bounds, ws);
    hipLaunchKernelGGL(step_finish_kernel, dim3(1), dim3(256), 0, st, dvals, ng, nd, pens, stats, ws, nseg, tail);
    return hipGetLastError();
}
long segment_sqnorms_ws_doubles(int n) { return (long)(n > 0 ? n : 0) * SQ_CHUNKS; }
hipError_t launch_segment_sqnorms(const float* x, const long* bounds, int n, float* out, double* ws, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(segment_sqnorms_kernel, dim3(SQ_CHUNKS, n), dim3(256), 0, st, x, bounds, ws);
    hipLaunchKernelGGL(segment_sqnorms_finish_kernel, dim3((n + 63) / 64), dim3(64), 0, st, ws, out, n);
    return hipGetLastError();
}
__global__ void __launch_bounds__(256) interpolate_kernel(const float* __restrict__ eps, const float* __restrict__ xd,
                                                         const float* __restrict__ xg, float* __restrict__ xp, long n, int cols) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n) return;
    const float e = eps[i / cols];
    xp[i] = __fadd_rn(__fmul_rn(e, xd[i]), __fmul_rn(1.f - e, xg[i]));       // (two products, one sum, each rounded: numpy's bits; critic_step_inputs_kernel forms the same)
}
hipError_t launch_interpolate(const float* eps, const float* xd, const float* xg, float* xp, int rows, int cols, hipStream_t st) {
    const long n = (long)rows * cols;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(interpolate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, eps, xd, xg, xp, n, cols);
    return hipGetLastError();
}

// Adjoint of the conditional prober's gather tuning_curve[k][s] = time_avg[ids[k]][s][probes[k]] (cwgan.py:91-98):
// g_ta[b][s][m] = sum over the samples k with ids[k] == b and probes[k] == m of g[k][s], zero elsewhere.  One workgroup
// per model b (ONE wave) zeroes its slab and lets thread s add the matching samples in k order: deterministic whatever the
// collisions, no atomics, no sort, no host wait.
template <typename T>
__global__ void __launch_bounds__(64) probe_scatter_kernel(const T* __restrict__ g, const long* __restrict__ ids,
                                                          const long* __restrict__ probes, T* __restrict__ g_ta, int n, int NB, int M) {
    const int b = blockIdx.x;
    T* slab = g_ta + (size_t)b * NB * M;
    for (int i = threadIdx.x; i < NB * M; i += 64) slab[i] = (T)0;
    __syncthreads();
    // the wave looks at 64 samples at a time (one ballot), then walks the matches in k order: the additions of a (stimulus,
    // probe) cell are made by one thread in the order of the plain loop over k (round 4: every thread scanned all n ids itself)
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + (int)threadIdx.x;
        unsigned long long hit = __ballot(k < n && ids[k] == b);
        while (hit) {
            const int kk = k0 + __builtin_ctzll(hit);
            hit &= hit - 1;
            const long m = probes[kk];
            for (int s = threadIdx.x; s < NB; s += 64) slab[(size_t)s * M + m] += g[(size_t)kk * NB + s];
        }
    }
}
template <typename T>
hipError_t launch_probe_scatter(const T* g, const long* ids, const long* probes, T* g_ta, int n, int B, int NB, int M, hipStream_t st) {
    if (B <= 0 || NB <= 0 || M <= 0) return hipSuccess;
    hipLaunchKernelGGL((probe_scatter_kernel<T>), dim3(B), dim3(64), 0, st, g, ids, probes, g_ta, n, NB, M);
    return hipGetLastError();
}
template hipError_t launch_probe_scatter<float>(const float*, const long*, const long*, float*, int, int, int, int, hipStream_t);
template hipError_t launch_probe_scatter<double>(const double*, const long*, const long*, double*, int, int, int, int, hipStream_t);

}  // namespace ssn
