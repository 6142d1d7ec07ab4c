// Fixed-point implicit gradient (tc_gan/gradient_expressions/SS_grad.py:17-99, make_w_batch.py:36-121).
// At a fixed point r = f(W r + I):  dr/dtheta = (1 - Phi W)^-1 Phi (dW/dtheta r),  Phi = diag f'(W r + I).
// Two kernels build what the reference builds symbolically -- dW/dJ, dW/dD, dW/dS as [nz][M][M][2][2] tensors
// and the batched linear systems (A = 1 - Phi W, rhs = Phi (dW r)) -- and the batched LU solve is a library call.
#include <hip/hip_runtime.h>
#include "ssn_device.h"
#include "ssn_host.h"

namespace ssn {

// dW[b][i][j][p][q] = d W[b][i][j] / d theta[p][q], theta = J (which 0), D (1), S (2); W = wnn (sgn J + sgn D z),
// wnn = exp(-dx^2 / 2 S^2)  (make_w_batch.py:36-121 with identity dJ'/dJ).  Nonzero only for the (p, q) of the block.
template <typename T>
__global__ void __launch_bounds__(256) build_dw_kernel(const T* __restrict__ z, JDSv<T> p, int which, int N,
                                                       T* __restrict__ dW, long total) {
    const int M = 2 * N;
    const T inv_nm1 = (N > 1) ? (T)1 / (T)(N - 1) : (T)0;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int col = (int)(e % M), row = (int)((e / M) % M);
        const int pp = row >= N, qq = col >= N, i = row - pp * N, j = col - qq * N, pq = pp * 2 + qq;
        const T dx = (T)(i - j) * inv_nm1;
        const T wnn = exp(-(dx * dx) * p.inv2s2[pq]);
        const T sgn = qq ? (T)-1 : (T)1;
        T v;
        if (which == 0) v = sgn * wnn;
        else if (which == 1) v = sgn * wnn * z[e];
        else v = wnn * dx * dx * p.inv_s3[pq] * (sgn * p.J[pq] + sgn * p.D[pq] * z[e]);
        T* o = dW + e * 4;
        o[0] = o[1] = o[2] = o[3] = (T)0;
        o[pq] = v;
    }
}

template <typename T>
__device__ __forceinline__ T io_slope(T v, const IoConsts<T>& c) {
    // f'(v) of the selected branch (SS_grad.py:76-99)
    if (!(v > (T)0)) return (T)0;
    if (c.io_type == SSN_IO_POWER || v <= c.v0) return c.n * pow_rate(v, c.k, c.n) / v;
    if (c.io_type == SSN_IO_LINEAR) return c.lin_slope;
    const T th = tanh_pos(c.tanh_gain * (v - c.v0));
    return c.span_gain * ((T)1 - th * th);
}

// one workgroup per (draw z, stimulus b, row i):
//   v = W[z][i][:] . R[z][b][:] + I;  phi = f'(v);  A[z][b][i][j] = delta_ij - phi W[z][i][j];
//   rhs[z][b][i][c] = phi * sum_j dW[z or 0][i][j][c] R[z][b][j]      (c = 2 p + q)
template <typename T>
__global__ void __launch_bounds__(256) ss_system_kernel(const T* __restrict__ R, const T* __restrict__ W,
                                                        const T* __restrict__ dW, int dw_per_draw,
                                                        const T* __restrict__ I, int i_per_draw, IoConsts<T> io,
                                                        int nb, int M, T* __restrict__ A, T* __restrict__ rhs) {
    const long blk = blockIdx.x;
    const int i = (int)(blk % M);
    const long zb = blk / M;
    const int b = (int)(zb % nb);
    const long zi = zb / nb;
    const T* w = W + (zi * M + i) * M;
    const T* dw = dW + ((dw_per_draw ? zi : 0) * M + i) * (long)M * 4;
    const T* r = R + zb * M;
    T s[5] = {0, 0, 0, 0, 0};
    for (int j = threadIdx.x; j < M; j += blockDim.x) {
        const T rj = r[j];
        s[0] += w[j] * rj;
        s[1] += dw[4 * j] * rj; s[2] += dw[4 * j + 1] * rj; s[3] += dw[4 * j + 2] * rj; s[4] += dw[4 * j + 3] * rj;
    }
    __shared__ T red[5][256];
#pragma unroll
    for (int c = 0; c < 5; ++c) red[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) {
#pragma unroll
            for (int c = 0; c < 5; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + off];
        }
        __syncthreads();
    }
    const T v = red[0][0] + I[(i_per_draw ? zb : b) * M + i];
    const T phi = io_slope(v, io);
    T* a = A + (zb * M + i) * (long)M;
    for (int j = threadIdx.x; j < M; j += blockDim.x) a[j] = ((j == i) ? (T)1 : (T)0) - phi * w[j];
    if (threadIdx.x < 4) rhs[(zb * M + i) * 4 + threadIdx.x] = phi * red[1 + threadIdx.x][0];
}

template <typename T>
hipError_t launch_build_dw(const T* z, const T* jds12, int which, T* dW, int B, int N, hipStream_t st) {
    JDSv<T> p;
    for (int q = 0; q < 4; ++q) {
        p.J[q] = jds12[q]; p.D[q] = jds12[4 + q];
        const T s = jds12[8 + q];
        p.inv2s2[q] = (T)1 / ((T)2 * s * s);
        p.inv_s3[q] = (T)1 / (s * s * s);
    }
    const long total = (long)B * 4 * N * N;
    if (total == 0) return hipSuccess;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL((build_dw_kernel<T>), dim3(blocks), dim3(256), 0, st, z, p, which, N, dW, total);
    return hipGetLastError();
}
template hipError_t launch_build_dw<float>(const float*, const float*, int, float*, int, int, hipStream_t);
template hipError_t launch_build_dw<double>(const double*, const double*, int, double*, int, int, hipStream_t);

template <typename T>
hipError_t launch_ss_system(const T* R, const T* W, const T* dW, int dw_per_draw, const T* I, int i_per_draw,
                            const IoConsts<T>& io, int nz, int nb, int M, T* A, T* rhs, hipStream_t st) {
    const long blocks = (long)nz * nb * M;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL((ss_system_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, R, W, dW, dw_per_draw, I,
                       i_per_draw, io, nb, M, A, rhs);
    return hipGetLastError();
}
template hipError_t launch_ss_system<float>(const float*, const float*, const float*, int, const float*, int,
                                            const IoConsts<float>&, int, int, int, float*, float*, hipStream_t);
template hipError_t launch_ss_system<double>(const double*, const double*, const double*, int, const double*, int,
                                             const IoConsts<double>&, int, int, int, double*, double*, hipStream_t);

// ---------------------------------------------------------------------------------------------------------------
// Batched dense solve A x = b for the systems above (SS_grad.py:44 `solve` in a scan over (draw, stimulus)):
// Gaussian elimination with partial (row) pivoting, one workgroup per system, in place -- A ends as its LU factors,
// rhs as the solution.  The matrix stays in global memory (160 KB fp32 / 320 KB fp64 per system at 2N = 200: it lives in
// L2 while its workgroup works on it); pivot row and multipliers of the current step are staged in LDS.  Per step:
// pivot search (LDS tree), row swap, multipliers, rank-1 update of the trailing block and of the right-hand sides;
// then a column-oriented back substitution.  ~4 M barriers per system; M^3 / 3 FMAs in k order.
// info[s] = 0, or 1 + the first step whose pivot column was entirely zero (singular to working precision).
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int NRHS>
__global__ void __launch_bounds__(256) lu_solve_kernel(T* __restrict__ Aall, T* __restrict__ Ball, int* __restrict__ info, int M) {
    extern __shared__ __align__(16) unsigned char lu_smem[];
    T* rowk = reinterpret_cast<T*>(lu_smem);              // [M + NRHS]  pivot row (columns k.., then its right-hand sides)
    T* lcol = rowk + M + NRHS;                            // [M]         multipliers of the current column
    __shared__ T red_v[256];
    __shared__ int red_i[256];
    __shared__ int bad;
    const int tid = threadIdx.x, nt = blockDim.x;
    T* A = Aall + (size_t)blockIdx.x * M * M;
    T* B = Ball + (size_t)blockIdx.x * M * NRHS;
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int k = 0; k < M; ++k) {
        // ---- pivot: largest |A[i][k]|, i >= k (ties: the smallest i, as LAPACK's i?amax) ------------------
        T best = (T)-1; int bi = k;
        for (int i = k + tid; i < M; i += nt) {
            const T v = abs_t(A[(size_t)i * M + k]);
            if (v > best) { best = v; bi = i; }
        }
        red_v[tid] = best; red_i[tid] = bi;
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) {
            if (tid < off) {
                const T v = red_v[tid + off]; const int i2 = red_i[tid + off];
                if (v > red_v[tid] || (v == red_v[tid] && i2 < red_i[tid])) { red_v[tid] = v; red_i[tid] = i2; }
            }
            __syncthreads();
        }
        const int p = red_i[0];
        const T pv = red_v[0];
        if (!(pv > (T)0)) { if (tid == 0 && bad == 0) bad = k + 1; }
        // ---- swap rows k and p (columns k.. and the right-hand sides); stage the new row k in LDS -----------
        for (int j = k + tid; j < M + NRHS; j += nt) {
            T* ak = (j < M) ? &A[(size_t)k * M + j] : &B[(size_t)k * NRHS + (j - M)];
            T* ap = (j < M) ? &A[(size_t)p * M + j] : &B[(size_t)p * NRHS + (j - M)];
            const T vk = *ak, vp = *ap;
            if (p != k) { *ak = vp; *ap = vk; }
            rowk[j] = vp;
        }
        __syncthreads();
        const T piv = rowk[k];
        const T rinv = (piv != (T)0) ? (T)1 / piv : (T)0;
        for (int i = k + 1 + tid; i < M; i += nt) {
            const T l = A[(size_t)i * M + k] * rinv;
            lcol[i] = l;
            A[(size_t)i * M + k] = l;
        }
        __syncthreads();
        // ---- trailing update: A[i][j] -= l_i rowk[j] (j > k) and the right-hand sides ----------------------
        const int w = M + NRHS - (k + 1);                  // columns k+1 .. M+NRHS-1
        const long cells = (long)(M - k - 1) * w;
        for (long e = tid; e < cells; e += nt) {
            const int i = k + 1 + (int)(e / w), j = k + 1 + (int)(e % w);
            T* a = (j < M) ? &A[(size_t)i * M + j] : &B[(size_t)i * NRHS + (j - M)];
            *a = fma(-lcol[i], rowk[j], *a);
        }
        __syncthreads();
    }
    // ---- back substitution, column oriented: x_k = b_k / u_kk, then b_i -= u_ik x_k for i < k --------------
    for (int k = M - 1; k >= 0; --k) {
        if (tid < NRHS) {
            const T ukk = A[(size_t)k * M + k];
            const T x = (ukk != (T)0) ? B[(size_t)k * NRHS + tid] / ukk : (T)0;
            B[(size_t)k * NRHS + tid] = x;
            rowk[tid] = x;
        }
        __syncthreads();
        for (int e = tid; e < k * NRHS; e += nt) {
            const int i = e / NRHS, c = e % NRHS;
            B[(size_t)i * NRHS + c] = fma(-A[(size_t)i * M + k], rowk[c], B[(size_t)i * NRHS + c]);
        }
        __syncthreads();
    }
    if (info && tid == 0) info[blockIdx.x] = bad;
}

template <typename T>
hipError_t launch_lu_solve(T* A, T* rhs, int* info, int nsys, int M, int nrhs, hipStream_t st) {
    if (nsys <= 0 || M <= 0) return hipSuccess;
    if (nrhs != 4 && nrhs != 1) return hipErrorInvalidValue;
    const size_t lds = (size_t)(2 * M + nrhs) * sizeof(T);
    if (nrhs == 4) hipLaunchKernelGGL((lu_solve_kernel<T, 4>), dim3(nsys), dim3(256), lds, st, A, rhs, info, M);
    else hipLaunchKernelGGL((lu_solve_kernel<T, 1>), dim3(nsys), dim3(256), lds, st, A, rhs, info, M);
    return hipGetLastError();
}
template hipError_t launch_lu_solve<float>(float*, float*, int*, int, int, int, hipStream_t);
template hipError_t launch_lu_solve<double>(double*, double*, int*, int, int, int, hipStream_t);

}  // namespace ssn
