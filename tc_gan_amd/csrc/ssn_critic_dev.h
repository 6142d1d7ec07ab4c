// Per-row and per-update scalar pieces of the critic's loss that more than one kernel evaluates (the layer-by-layer path of
// ssn_critic.hip and the row-block path of ssn_critic_rows.hip): ONE definition each, with the fused multiply-adds written
// out, so that both paths round the same way whatever the compiler would contract.
#pragma once
#include <hip/hip_runtime.h>

namespace ssn {

// upstream of mean D(xg) - mean D(xd) for row i of the stacked batch [xg; xd]   (cwgan.py:190-200)
__device__ __forceinline__ float critic_updown(int i, int ng, int nd) { return (i < ng) ? 1.f / (float)ng : -1.f / (float)nd; }

// gradient-penalty head of one row (cwgan.py:203-214): g[0:nx] = dD/dx of the row; returns norm - 1 and, in coef, the factor
// of the row's upstream  ghat = coef g = 2 (norm - 1) / norm / batch * g
__device__ __forceinline__ float critic_gp_row(const float* g, int nx, int batch, float& coef) {
    float s = 0.f;
    for (int j = 0; j < nx; ++j) { const float v = g[j]; s = __builtin_fmaf(v, v, s); }
    const float nrm = sqrtf(s);
    const float d = nrm - 1.f;
    coef = (nrm > 0.f) ? 2.f * d / nrm / (float)batch : 0.f;
    return d;
}

// loss = mean D(xg) - mean D(xd) + lambda * penalty
__device__ __forceinline__ float critic_loss_value(float dg, float dd, float pen, float lmd) { return __builtin_fmaf(lmd, pen, dg - dd); }

// stats[0] = mean D(xg), stats[1] = mean D(xd) (the order of two_means_kernel), stats[2] = mean (norm - 1)^2 over the penalty rows
// (the order of gp_head_kernel's sum; dnorm[b] = norm_b - 1), stats[3] = the loss; threads 0 .. 255 of one workgroup
__device__ __forceinline__ void critic_stats_block(const float* __restrict__ d, const float* __restrict__ dnorm, float* __restrict__ stats,
                                                   int ng, int nd, int np, float lmd, float (&red)[3][256]) {
    const int t = threadIdx.x;
    if (t < 256) {
        float x = 0.f, y = 0.f, p = 0.f;
        for (int i = t; i < ng; i += 256) x += d[i];
        for (int i = t; i < nd; i += 256) y += d[ng + i];
        for (int b = t; b < np; b += 256) { const float e = dnorm[b]; p = __builtin_fmaf(e, e, p); }
        red[0][t] = x; red[1][t] = y; red[2][t] = p;
    }
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (t < off) { red[0][t] += red[0][t + off]; red[1][t] += red[1][t + off]; red[2][t] += red[2][t + off]; }
        __syncthreads();
    }
    if (t == 0) {
        const float s0 = ng ? red[0][0] / ng : 0.f, s1 = nd ? red[1][0] / nd : 0.f, s2 = red[2][0] / (float)np;
        stats[0] = s0; stats[1] = s1; stats[2] = s2;
        stats[3] = critic_loss_value(s0, s1, s2, lmd);
    }
}

}  // namespace ssn
