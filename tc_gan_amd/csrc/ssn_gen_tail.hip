// End of a generator step of the BPTT GANs (GeneratorTrainer, tc_gan/networks/wgan.py:218-260) without a tensor library in
// between: what the adjoint sweep, dL/dW and the chain rule through make_W_with_x left on the device becomes the flat
// gradient vector [dL/dV, dL/dJ, dL/dD, dL/dS] and the step's loss in ONE launch (gen_grads_kernel), and -- behind the job's
// all-reduce, when there is one -- the optimizer update of every parameter with its own clip bounds plus the record the
// host reads back in ONE more (optimizer_kernel with per-element bounds, ssn_critic.hip).  Round 3 spent ~20 launches of
// element-wise and reduction kernels of a tensor library on the same arithmetic (the paper's shape runs 3.6 ms per GAN
// iteration: every launch is a percent).
//
//   dL/dtheta_pq = sum_b part[b][pq][t]                      (t = 0, 1, 2: J, D, S; part from jds_grad_kernel)
//   dL/dV_pop    = sum_{b, s, m in pop} g_ext[b][s][m] ext_base[b][s][m] z_in[b][m]      (networks/ssn.py:679-686:
//                  ext = (1 + V_pop z_in) ext_base; 'deg-heteroin': one V for both populations = the sum of the two)
//   loss         = -mean D(G(z)) + dynamics_cost dynamics_penalty + rate_cost rate_penalty        (wgan.py:236-241)
// All sums in fp64 and in a fixed order (workgroup partials, then the workgroup that draws the last ticket adds them as a
// tree): the same bits every run.
#include <hip/hip_runtime.h>
#include "ssn_host.h"

namespace ssn {

constexpr int GT_BLOCKS = 128;

__device__ __forceinline__ double gt_block_sum(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(256) gen_grads_kernel(GenGradsArgs a) {
    __shared__ double red[256];
    __shared__ int last;
    const int N = a.M / 2;
    // ---- partial sums of this workgroup: the two populations' input-variability gradients
    double vE = 0.0, vI = 0.0;
    if (a.nv > 0) {
        const long per_draw = (long)a.NB * a.M, total = (long)a.B * per_draw;
        for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
            const long b = e / per_draw;
            const int m = (int)(e % a.M);
            const double t = (double)a.g_ext[e] * (double)a.ext_base[e] * (double)a.zin[b * a.M + m];
            if (m < N) vE += t; else vI += t;
        }
    }
    const double sE = gt_block_sum(vE, red), sI = gt_block_sum(vI, red);
    int* ticket = reinterpret_cast<int*>(a.ws + 2 * GT_BLOCKS);
    if (threadIdx.x == 0) {
        a.ws[2 * blockIdx.x] = sE; a.ws[2 * blockIdx.x + 1] = sI;
        __threadfence();
        last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // ---- the last workgroup: everything else, in a fixed order
    const bool have = threadIdx.x < gridDim.x;
    const double tE = gt_block_sum(have ? __builtin_nontemporal_load(a.ws + 2 * threadIdx.x) : 0.0, red);
    const double tI = gt_block_sum(have ? __builtin_nontemporal_load(a.ws + 2 * threadIdx.x + 1) : 0.0, red);
    if (threadIdx.x == 0) {
        if (a.nv == 1) a.out[0] = (float)(tE + tI);
        else if (a.nv == 2) { a.out[0] = (float)tE; a.out[1] = (float)tI; }
    }
    for (int c = 0; c < 12; ++c) {                    // c = 3 q + t of part[b][q][t]; out: J (t = 0), D (1), S (2) blocks of four
        double s = 0.0;
        for (int b = threadIdx.x; b < a.B; b += 256) s += a.part[(size_t)b * 12 + c];
        const double tot = gt_block_sum(s, red);
        if (threadIdx.x == 0) a.out[a.nv + (c % 3) * 4 + c / 3] = (float)tot;
    }
    if (threadIdx.x == 0) {
        const double dyn = a.pens ? a.pens[0] : 0.0, rate = a.pens ? a.pens[1] : 0.0;
        a.out[a.nv + 12] = (float)(-(double)a.dmean[0] + a.dynamics_cost * dyn + a.rate_cost * rate);
        *ticket = 0;
    }
}

hipError_t launch_gen_grads(const GenGradsArgs& a, hipStream_t st) {
    long blocks = 1;
    if (a.nv > 0) {
        blocks = ((long)a.B * a.NB * a.M + 255) / 256;
        if (blocks < 1) blocks = 1;
        if (blocks > GT_BLOCKS) blocks = GT_BLOCKS;
    }
    hipLaunchKernelGGL(gen_grads_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace ssn
