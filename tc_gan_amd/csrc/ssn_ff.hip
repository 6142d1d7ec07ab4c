// Feed-forward tuning-curve generator (FF_lalazar model) on MI355X (gfx950).
//
// Reference: FF_functions/lalazar_func.py:16-45 (get_FF_output) as used by
// FF_lalazar_model.py:154-177 (BASELINE config 5; restated in oracle/ff_torch.py):
//   e[s,i,g]   = exp(-|pos_g - stim_i|^2 / (2 (RF_w[s,g] RF_d + RF_l)^2))
//   out[s,i,h] = relu( sum_g e[s,i,g] J FF_con[s,h,g] FF_str[s,h,g] / sum_g e[s,i,g]  -  thr[s,h] )
//   thr[s,h]   = TH + sgn(TH_sam[s,h]) |TH_sam[s,h]|^a TH_d
// over G = box^3 grid points of linspace(-3, 3, box)^3 (positions derived from the index, never loaded).
//
// HBM-bound streaming reduction: per sample the three per-point streams RF_w, FF_con, FF_str
// (12 B per grid point, 16-byte accesses) are read ONCE and feed all NI stimuli; one workgroup per
// (sample, hidden unit) keeps 2*NI running sums in registers, finishes with a wavefront shuffle +
// LDS reduction.  Algorithmic traffic: nsam * nhid * G * 12 B (786 MB per 1024 samples at box 40).
// The backward kernel is a second pass of the same shape that produces the per-sample partial
// derivatives w.r.t. RF_l and RF_d; the remaining parameter gradients follow from the forward sums.
#include <hip/hip_runtime.h>
#include "ssn_host.h"

namespace ssn {

constexpr int FF_MAX_NI = 32;

template <int NI_T>
__device__ __forceinline__ void ff_block_reduce(float (&v)[2 * NI_T], float* red /* [4][2*NI_T] */, int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 2 * NI_T; ++k) {
        float x = v[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
        if (lane == 0 && k < n) red[wave * 2 * NI_T + k] = x;
    }
    __syncthreads();
}

// grid = nsam * nhid workgroups of 256 threads.  sums[s][h][0][i] = sum_g e w, sums[s][h][1][i] = sum_g e.
template <int NI_T>
__global__ void __launch_bounds__(256) ff_forward_kernel(FFArgs a) {
    __shared__ float stim[NI_T][3];
    __shared__ float red[4][2 * NI_T];
    const int s = blockIdx.x / a.nhid, h = blockIdx.x % a.nhid;
    const int G = a.box * a.box * a.box;
    if (threadIdx.x < a.ni * 3) (&stim[0][0])[threadIdx.x] = a.stim[threadIdx.x];
    __syncthreads();
    const float* rfw = a.RF_w + (size_t)s * G;
    const float* con = a.FF_con + ((size_t)s * a.nhid + h) * G;
    const float* str = a.FF_str + ((size_t)s * a.nhid + h) * G;
    const float step = (a.box > 1) ? 6.f / (float)(a.box - 1) : 0.f;
    float acc[2 * NI_T];
#pragma unroll
    for (int k = 0; k < 2 * NI_T; ++k) acc[k] = 0.f;
    using V4 = float __attribute__((ext_vector_type(4)));
    const bool vec = (G % 4 == 0);
    const int nv = vec ? G / 4 : G;
    for (int v = threadIdx.x; v < nv; v += 256) {
        float w4[4], c4[4], s4[4];
        const int g0 = vec ? 4 * v : v;
        const int cnt = vec ? 4 : 1;
        if (vec) {
            const V4 q0 = *reinterpret_cast<const V4*>(rfw + g0), q1 = *reinterpret_cast<const V4*>(con + g0),
                     q2 = *reinterpret_cast<const V4*>(str + g0);
            w4[0] = q0.x; w4[1] = q0.y; w4[2] = q0.z; w4[3] = q0.w;
            c4[0] = q1.x; c4[1] = q1.y; c4[2] = q1.z; c4[3] = q1.w;
            s4[0] = q2.x; s4[1] = q2.y; s4[2] = q2.z; s4[3] = q2.w;
        } else { w4[0] = rfw[g0]; c4[0] = con[g0]; s4[0] = str[g0]; }
        for (int t = 0; t < cnt; ++t) {
            const int g = g0 + t;
            const int iz = g % a.box, iy = (g / a.box) % a.box, ix = g / (a.box * a.box);
            const float px = -3.f + step * ix, py = -3.f + step * iy, pz = -3.f + step * iz;
            const float sig = w4[t] * a.RF_d + a.RF_l;
            const float inv2s2 = 0.5f * __builtin_amdgcn_rcpf(sig * sig);
            const float wgt = a.J * c4[t] * s4[t];
#pragma unroll
            for (int i = 0; i < NI_T; ++i) {
                if (i < a.ni) {
                    const float dx = px - stim[i][0], dy = py - stim[i][1], dz = pz - stim[i][2];
                    const float e = __expf(-(dx * dx + dy * dy + dz * dz) * inv2s2);
                    acc[i] += e * wgt;
                    acc[NI_T + i] += e;
                }
            }
        }
    }
    ff_block_reduce<NI_T>(acc, &red[0][0], 2 * NI_T);
    if (threadIdx.x < a.ni) {
        const int i = threadIdx.x;
        float num = 0.f, den = 0.f;
        for (int wv = 0; wv < 4; ++wv) { num += red[wv][i]; den += red[wv][NI_T + i]; }
        const float ts = a.TH_sam[(size_t)s * a.nhid + h];
        const float thr = a.TH + ((ts > 0.f) - (ts < 0.f)) * __powf(fabsf(ts), a.a) * a.TH_d;
        const size_t o = ((size_t)s * a.ni + i) * a.nhid + h;
        const float q = num / den;
        a.out[o] = fmaxf(q - thr, 0.f);
        if (a.q) { a.q[o] = q; a.den[o] = den; }
    }
}

// Fast path for the model script's stimulus set {X} x {Y} x {Z} (3 x 3 x 3 lattice, i = (a*3 + b)*3 + c,
// FF_lalazar_model.py:139): exp(-|p - s_i|^2 c) factorises into ex[a] ey[b] ez[c], so a grid point costs
// 9 exponentials and 36 multiplies instead of 27 exponentials and 27 distance evaluations.
__global__ void __launch_bounds__(256) ff_forward_lattice_kernel(FFArgs a, FFLattice lat) {
    constexpr int NI_T = 27;
    __shared__ float red[4][2 * NI_T];
    const int s = blockIdx.x / a.nhid, h = blockIdx.x % a.nhid;
    const int G = a.box * a.box * a.box;
    const float* rfw = a.RF_w + (size_t)s * G;
    const float* con = a.FF_con + ((size_t)s * a.nhid + h) * G;
    const float* str = a.FF_str + ((size_t)s * a.nhid + h) * G;
    const float step = (a.box > 1) ? 6.f / (float)(a.box - 1) : 0.f;
    using V2 = float __attribute__((ext_vector_type(2)));
    V2 acc2[NI_T];                               // {sum e w, sum e} per stimulus: one v_pk_fma_f32 per update
#pragma unroll
    for (int k = 0; k < NI_T; ++k) acc2[k] = V2{0.f, 0.f};
    using V4 = float __attribute__((ext_vector_type(4)));
    const bool vec = (G % 4 == 0);
    const int nv = vec ? G / 4 : G;
    // software prefetch: the three 16-byte loads of iteration v+256 are issued before the ~150 VALU
    // instructions of iteration v (2 waves per SIMD do not hide an HBM round trip on their own)
    V4 n0 = V4{0, 0, 0, 0}, n1 = n0, n2 = n0;
    auto fetch = [&](int v, V4& q0, V4& q1, V4& q2) {
        if (v < nv) {
            if (vec) {
                q0 = *reinterpret_cast<const V4*>(rfw + 4 * v); q1 = *reinterpret_cast<const V4*>(con + 4 * v);
                q2 = *reinterpret_cast<const V4*>(str + 4 * v);
            } else { q0.x = rfw[v]; q1.x = con[v]; q2.x = str[v]; }
        }
    };
    fetch(threadIdx.x, n0, n1, n2);
    // grid coordinates of the thread's current first point, advanced incrementally (no per-point div/mod)
    const int pts = vec ? 4 : 1;
    int bz, by, bx;
    { const int g = threadIdx.x * pts; bz = g % a.box; by = (g / a.box) % a.box; bx = g / (a.box * a.box); }
    const int stride = 256 * pts;
    const int sx = stride / (a.box * a.box), sy = (stride % (a.box * a.box)) / a.box, sz = stride % a.box;
    // one point: 9 exponentials, 9 + 9 + 3 multiplies, 27 packed FMAs.  e = ex ey ez is never formed:
    // sum e w += (ex ey) (ez w), sum e += (ex ey) ez.
    auto point = [&](const float (&dxx)[3], const float (&dyy)[3], const float (&dzz)[3], float w, float c, float sv) {
        const float sig = w * a.RF_d + a.RF_l;
        const float c2 = -0.72134752044448170f * __builtin_amdgcn_rcpf(sig * sig);   // -log2(e) / (2 sig^2), 1 ulp
        const float wj = a.J * c * sv;
        float ex[3], ey[3], ez[3], ezw[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ex[k] = __builtin_amdgcn_exp2f(dxx[k] * c2);
            ey[k] = __builtin_amdgcn_exp2f(dyy[k] * c2);
            ez[k] = __builtin_amdgcn_exp2f(dzz[k] * c2);
            ezw[k] = ez[k] * wj;
        }
#pragma unroll
        for (int ia = 0; ia < 3; ++ia)
#pragma unroll
            for (int ib = 0; ib < 3; ++ib) {
                const float exy = ex[ia] * ey[ib];
#pragma unroll
                for (int ic = 0; ic < 3; ++ic) {
                    const int i = (ia * 3 + ib) * 3 + ic;
                    acc2[i] = __builtin_elementwise_fma(V2{exy, exy}, V2{ezw[ic], ez[ic]}, acc2[i]);
                }
            }
    };
    auto sq3 = [&](float p, const float (&l)[3], float (&out)[3]) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float d = p - l[k]; out[k] = d * d; }
    };
    if (vec && a.box % 4 == 0) {
        // the four points of a 16-byte load are consecutive in z and share (ix, iy): the squared x and y distances to the
        // lattice are formed once per load (6 + 6 VALU per four points instead of per point), fully unrolled
        for (int v = threadIdx.x; v < nv; v += 256) {
            const V4 q0 = n0, q1 = n1, q2 = n2;
            fetch(v + 256, n0, n1, n2);
            float dxx[3], dyy[3], dzz[3];
            sq3(-3.f + step * bx, lat.x, dxx);
            sq3(-3.f + step * by, lat.y, dyy);
            const float pz0 = -3.f + step * bz;
            sq3(pz0, lat.z, dzz);             point(dxx, dyy, dzz, q0.x, q1.x, q2.x);
            sq3(pz0 + step, lat.z, dzz);      point(dxx, dyy, dzz, q0.y, q1.y, q2.y);
            sq3(pz0 + 2.f * step, lat.z, dzz); point(dxx, dyy, dzz, q0.z, q1.z, q2.z);
            sq3(pz0 + 3.f * step, lat.z, dzz); point(dxx, dyy, dzz, q0.w, q1.w, q2.w);
            bz += sz; if (bz >= a.box) { bz -= a.box; ++by; }
            by += sy; if (by >= a.box) { by -= a.box; ++bx; }
            bx += sx;
        }
    } else {
        for (int v = threadIdx.x; v < nv; v += 256) {
            const V4 q0 = n0, q1 = n1, q2 = n2;
            fetch(v + 256, n0, n1, n2);
            const float w4[4] = {q0.x, q0.y, q0.z, q0.w}, c4[4] = {q1.x, q1.y, q1.z, q1.w}, s4[4] = {q2.x, q2.y, q2.z, q2.w};
            const int cnt = vec ? 4 : 1;
            int iz = bz, iy = by, ix = bx;
            for (int t = 0; t < cnt; ++t) {
                if (t > 0) { if (++iz >= a.box) { iz = 0; if (++iy >= a.box) { iy = 0; ++ix; } } }
                float dxx[3], dyy[3], dzz[3];
                sq3(-3.f + step * ix, lat.x, dxx);
                sq3(-3.f + step * iy, lat.y, dyy);
                sq3(-3.f + step * iz, lat.z, dzz);
                point(dxx, dyy, dzz, w4[t], c4[t], s4[t]);
            }
            bz += sz; if (bz >= a.box) { bz -= a.box; ++by; }
            by += sy; if (by >= a.box) { by -= a.box; ++bx; }
            bx += sx;
        }
    }
    float acc[2 * NI_T];
#pragma unroll
    for (int k = 0; k < NI_T; ++k) { acc[k] = acc2[k].x; acc[NI_T + k] = acc2[k].y; }
    ff_block_reduce<NI_T>(acc, &red[0][0], 2 * NI_T);
    if (threadIdx.x < NI_T) {
        const int i = threadIdx.x;
        float num = 0.f, den = 0.f;
        for (int wv = 0; wv < 4; ++wv) { num += red[wv][i]; den += red[wv][NI_T + i]; }
        const float ts = a.TH_sam[(size_t)s * a.nhid + h];
        const float thr = a.TH + ((ts > 0.f) - (ts < 0.f)) * __powf(fabsf(ts), a.a) * a.TH_d;
        const size_t o = ((size_t)s * a.ni + i) * a.nhid + h;
        const float q = num / den;
        a.out[o] = fmaxf(q - thr, 0.f);
        if (a.q) { a.q[o] = q; a.den[o] = den; }
    }
}

// Second pass: dsig[s][h] = sum_{i,g} gq[s,i,h] e d^2/sig^3 (w_g - q)/den  and the same weighted by RF_w[s,g]
// (gq = upstream gradient of the pre-threshold drive q, already masked by out > 0).
template <int NI_T>
__global__ void __launch_bounds__(256) ff_backward_kernel(FFArgs a, const float* __restrict__ gq, float* __restrict__ dsig) {
    __shared__ float stim[NI_T][3];
    __shared__ float coef[NI_T], qv[NI_T];
    __shared__ float red[4][2 * NI_T];
    const int s = blockIdx.x / a.nhid, h = blockIdx.x % a.nhid;
    const int G = a.box * a.box * a.box;
    if (threadIdx.x < a.ni * 3) (&stim[0][0])[threadIdx.x] = a.stim[threadIdx.x];
    if (threadIdx.x < a.ni) {
        const size_t o = ((size_t)s * a.ni + threadIdx.x) * a.nhid + h;
        coef[threadIdx.x] = gq[o] / a.den[o];
        qv[threadIdx.x] = a.q[o];
    }
    __syncthreads();
    const float* rfw = a.RF_w + (size_t)s * G;
    const float* con = a.FF_con + ((size_t)s * a.nhid + h) * G;
    const float* str = a.FF_str + ((size_t)s * a.nhid + h) * G;
    const float step = (a.box > 1) ? 6.f / (float)(a.box - 1) : 0.f;
    float acc[2 * NI_T];
#pragma unroll
    for (int k = 0; k < 2 * NI_T; ++k) acc[k] = 0.f;
    for (int g = threadIdx.x; g < G; g += 256) {
        const int iz = g % a.box, iy = (g / a.box) % a.box, ix = g / (a.box * a.box);
        const float px = -3.f + step * ix, py = -3.f + step * iy, pz = -3.f + step * iz;
        const float rw = rfw[g];
        const float sig = rw * a.RF_d + a.RF_l;
        const float inv2s2 = 0.5f * __builtin_amdgcn_rcpf(sig * sig), inv_s3 = __builtin_amdgcn_rcpf(sig * sig * sig);
        const float wgt = a.J * con[g] * str[g];
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NI_T; ++i) {
            if (i < a.ni) {
                const float dx = px - stim[i][0], dy = py - stim[i][1], dz = pz - stim[i][2];
                const float d2 = dx * dx + dy * dy + dz * dz;
                t += coef[i] * __expf(-d2 * inv2s2) * d2 * (wgt - qv[i]);
            }
        }
        acc[0] += t * inv_s3;
        acc[1] += t * inv_s3 * rw;
    }
    ff_block_reduce<NI_T>(acc, &red[0][0], 2);
    if (threadIdx.x < 2) {
        float v = 0.f;
        for (int wv = 0; wv < 4; ++wv) v += red[wv][threadIdx.x];
        dsig[((size_t)s * a.nhid + h) * 2 + threadIdx.x] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same model on its real data structure (round 4).  FF_con is box^3 / 100 ones per unit drawn with
// np.random.choice (FF_lalazar_model.py:154-167): 99 % of the FF_con / FF_str streams above are zeros, and they are two
// thirds of the dense kernel's bytes.  Here a unit's connections come as lists -- conn_idx[s][h][c] (grid index, -1 =
// empty slot) and conn_str[s][h][c] -- and the work splits in two:
//   den[s][i]    = sum over ALL grid points of e[s][i][g]            one streaming pass over RF_w (4 B per point), shared by
//                                                                    every hidden unit of the sample;
//   num[s][h][i] = sum over the unit's connections of e J str        RF_w gathered at <= ncon points (L2 / MALL hits: the pass
//                                                                    above has just read the sample's 256 KB).
// One workgroup per sample.  The streaming pass is transcendental-bound, not HBM-bound: 9 exponentials per point at a quarter
// of the plain rate are 36 of its ~64 issue slots per point; two points per lane share every packed multiply and FMA
// (v_pk_mul_f32 / v_pk_fma_f32 on {even point, odd point} pairs).
// ---------------------------------------------------------------------------------------------------------------
struct FFSparse { const int* idx; const float* str; int ncon; };
#ifndef FF_SPARSE_PACK
#define FF_SPARSE_PACK 1
#endif
#ifndef FF_SPARSE_DEPTH
#define FF_SPARSE_DEPTH 3
#endif
#ifndef FF_SPARSE_ABLATE
#define FF_SPARSE_ABLATE 0
#endif

__global__ void __launch_bounds__(256) ff_forward_sparse_lattice_kernel(FFArgs a, FFLattice lat, FFSparse sp) {
    constexpr int NI_T = 27;
    __shared__ float red[4][2 * NI_T];
    __shared__ float dens[NI_T];
    const int s = blockIdx.x;
    const int G = a.box * a.box * a.box;
    const float* rfw = a.RF_w + (size_t)s * G;
    const float step = (a.box > 1) ? 6.f / (float)(a.box - 1) : 0.f;
    using V2 = float __attribute__((ext_vector_type(2)));
    using V4 = float __attribute__((ext_vector_type(4)));
    const float KC = -0.72134752044448170f;            // -log2(e) / 2
    auto sq3 = [&](float p, const float (&l)[3], float (&out)[3]) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float d = p - l[k]; out[k] = d * d; }
    };
    // ---- pass 1: denominators, four z-consecutive points per 16-byte load, two points per packed instruction
    V2 den2[NI_T];
#pragma unroll
    for (int k = 0; k < NI_T; ++k) den2[k] = V2{0.f, 0.f};
    const int nv = G / 4;                              // (box % 4 == 0: checked by the launcher)
    int bz, by, bx;
    { const int g = threadIdx.x * 4; bz = g % a.box; by = (g / a.box) % a.box; bx = g / (a.box * a.box); }
    const int stride = 256 * 4;
    const int sx = stride / (a.box * a.box), sy = (stride % (a.box * a.box)) / a.box, sz = stride % a.box;
    auto pair = [&](const float (&dxx)[3], const float (&dyy)[3], const float (&dz0)[3], const float (&dz1)[3], float w0, float w1) {
#if !FF_SPARSE_PACK
        // (A/B form: the two points one after the other on plain fp32 instructions, sums in the .x halves)
        const float ww[2] = {w0, w1};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float sg = ww[t] * a.RF_d + a.RF_l;
            const float cc = KC * __builtin_amdgcn_rcpf(sg * sg);
            float fx[3], fy[3], fz[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                fx[k] = __builtin_amdgcn_exp2f(dxx[k] * cc);
                fy[k] = __builtin_amdgcn_exp2f(dyy[k] * cc);
                fz[k] = __builtin_amdgcn_exp2f((t ? dz1[k] : dz0[k]) * cc);
            }
#pragma unroll
            for (int ia = 0; ia < 3; ++ia)
#pragma unroll
                for (int ib = 0; ib < 3; ++ib) {
                    const float fxy = fx[ia] * fy[ib];
#pragma unroll
                    for (int ic = 0; ic < 3; ++ic) den2[(ia * 3 + ib) * 3 + ic].x = fmaf(fxy, fz[ic], den2[(ia * 3 + ib) * 3 + ic].x);
                }
        }
        return;
#endif
        const V2 sig = V2{w0, w1} * V2{a.RF_d, a.RF_d} + V2{a.RF_l, a.RF_l};
        const V2 s2 = sig * sig;
        const V2 c2 = V2{KC, KC} * V2{__builtin_amdgcn_rcpf(s2.x), __builtin_amdgcn_rcpf(s2.y)};
        V2 ex[3], ey[3], ez[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const V2 ax = V2{dxx[k], dxx[k]} * c2, ay = V2{dyy[k], dyy[k]} * c2, az = V2{dz0[k], dz1[k]} * c2;
#if FF_SPARSE_ABLATE == 1                                    // (timing only: no exponentials)
            ex[k] = ax; ey[k] = ay; ez[k] = az;
#else
            ex[k] = V2{__builtin_amdgcn_exp2f(ax.x), __builtin_amdgcn_exp2f(ax.y)};
            ey[k] = V2{__builtin_amdgcn_exp2f(ay.x), __builtin_amdgcn_exp2f(ay.y)};
            ez[k] = V2{__builtin_amdgcn_exp2f(az.x), __builtin_amdgcn_exp2f(az.y)};
#endif
        }
#pragma unroll
        for (int ia = 0; ia < 3; ++ia)
#pragma unroll
            for (int ib = 0; ib < 3; ++ib) {
                const V2 exy = ex[ia] * ey[ib];
#pragma unroll
                for (int ic = 0; ic < 3; ++ic) {
                    const int i = (ia * 3 + ib) * 3 + ic;
#if FF_SPARSE_ABLATE == 2                                    // (timing only: 9 of the 27 sums)
                    if (ic == 0) den2[i] = __builtin_elementwise_fma(exy, ez[ic] + ez[1] + ez[2], den2[i]);
#else
                    den2[i] = __builtin_elementwise_fma(exy, ez[ic], den2[i]);
#endif
                }
            }
    };
    // FF_SPARSE_DEPTH 16-byte loads in flight per lane (one iteration of ~180 instructions does not cover an HBM round trip:
    // with one load ahead the pass ran at the bytes-in-flight limit, 1.9 ms with or without its exponentials)
    constexpr int DEPTH = FF_SPARSE_DEPTH;
    V4 ring[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        ring[d] = V4{0, 0, 0, 0};
        if ((int)threadIdx.x + 256 * d < nv) ring[d] = *reinterpret_cast<const V4*>(rfw + 4 * (threadIdx.x + 256 * d));
    }
    for (int v = threadIdx.x; v < nv; v += 256) {
        const V4 q0 = ring[0];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d) ring[d] = ring[d + 1];
        if (v + 256 * DEPTH < nv) ring[DEPTH - 1] = *reinterpret_cast<const V4*>(rfw + 4 * (v + 256 * DEPTH));
        float dxx[3], dyy[3], dz0[3], dz1[3], dz2[3], dz3[3];
        sq3(-3.f + step * bx, lat.x, dxx);
        sq3(-3.f + step * by, lat.y, dyy);
        const float pz0 = -3.f + step * bz;
        sq3(pz0, lat.z, dz0); sq3(pz0 + step, lat.z, dz1); sq3(pz0 + 2.f * step, lat.z, dz2); sq3(pz0 + 3.f * step, lat.z, dz3);
        pair(dxx, dyy, dz0, dz1, q0.x, q0.y);
        pair(dxx, dyy, dz2, dz3, q0.z, q0.w);
        bz += sz; if (bz >= a.box) { bz -= a.box; ++by; }
        by += sy; if (by >= a.box) { by -= a.box; ++bx; }
        bx += sx;
    }
    float acc[2 * NI_T];
#pragma unroll
    for (int k = 0; k < NI_T; ++k) { acc[k] = den2[k].x + den2[k].y; acc[NI_T + k] = 0.f; }
    ff_block_reduce<NI_T>(acc, &red[0][0], NI_T);
    if (threadIdx.x < NI_T) dens[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
    // ---- pass 2: numerators of every hidden unit from its connection list
    for (int h = 0; h < a.nhid; ++h) {
        const int* idx = sp.idx + ((size_t)s * a.nhid + h) * sp.ncon;
        const float* str = sp.str + ((size_t)s * a.nhid + h) * sp.ncon;
#pragma unroll
        for (int k = 0; k < 2 * NI_T; ++k) acc[k] = 0.f;
        for (int c = threadIdx.x; c < sp.ncon; c += 256) {
            const int g = idx[c];
            if (g < 0 || g >= G) continue;
            const int iz = g % a.box, iy = (g / a.box) % a.box, ix = g / (a.box * a.box);
            float dxx[3], dyy[3], dzz[3];
            sq3(-3.f + step * ix, lat.x, dxx);
            sq3(-3.f + step * iy, lat.y, dyy);
            // (the z coordinate as the streaming pass forms it: base of the 4-point group + multiple of the step)
            sq3((-3.f + step * (iz & ~3)) + (float)(iz & 3) * step, lat.z, dzz);
            const float sig = rfw[g] * a.RF_d + a.RF_l;
            const float c2 = KC * __builtin_amdgcn_rcpf(sig * sig);
            const float wj = a.J * str[c];
            float ex[3], ey[3], ezw[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ex[k] = __builtin_amdgcn_exp2f(dxx[k] * c2);
                ey[k] = __builtin_amdgcn_exp2f(dyy[k] * c2);
                ezw[k] = __builtin_amdgcn_exp2f(dzz[k] * c2) * wj;
            }
#pragma unroll
            for (int ia = 0; ia < 3; ++ia)
#pragma unroll
                for (int ib = 0; ib < 3; ++ib) {
                    const float exy = ex[ia] * ey[ib];
#pragma unroll
                    for (int ic = 0; ic < 3; ++ic) acc[(ia * 3 + ib) * 3 + ic] += exy * ezw[ic];
                }
        }
        __syncthreads();                               // (red is reused)
        ff_block_reduce<NI_T>(acc, &red[0][0], NI_T);
        if (threadIdx.x < NI_T) {
            const int i = threadIdx.x;
            const float num = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
            const float den = dens[i];
            const float ts = a.TH_sam[(size_t)s * a.nhid + h];
            const float thr = a.TH + ((ts > 0.f) - (ts < 0.f)) * __powf(fabsf(ts), a.a) * a.TH_d;
            const size_t o = ((size_t)s * a.ni + i) * a.nhid + h;
            const float q = num / den;
            a.out[o] = fmaxf(q - thr, 0.f);
            if (a.q) { a.q[o] = q; a.den[o] = den; }
        }
        __syncthreads();
    }
}

// any stimulus set / box: the same split (denominators over all points, numerators over the connection list), one
// exponential per (stimulus, point)
template <int NI_T>
__global__ void __launch_bounds__(256) ff_forward_sparse_kernel(FFArgs a, FFSparse sp) {
    __shared__ float stim[NI_T][3];
    __shared__ float red[4][2 * NI_T];
    __shared__ float dens[NI_T];
    const int s = blockIdx.x;
    const int G = a.box * a.box * a.box;
    if (threadIdx.x < a.ni * 3) (&stim[0][0])[threadIdx.x] = a.stim[threadIdx.x];
    __syncthreads();
    const float* rfw = a.RF_w + (size_t)s * G;
    const float step = (a.box > 1) ? 6.f / (float)(a.box - 1) : 0.f;
    float acc[2 * NI_T];
#pragma unroll
    for (int k = 0; k < 2 * NI_T; ++k) acc[k] = 0.f;
    auto point = [&](int g, float w, float wgt, bool with_w) {
        const int iz = g % a.box, iy = (g / a.box) % a.box, ix = g / (a.box * a.box);
        const float px = -3.f + step * ix, py = -3.f + step * iy, pz = -3.f + step * iz;
        const float sig = w * a.RF_d + a.RF_l;
        const float inv2s2 = 0.5f * __builtin_amdgcn_rcpf(sig * sig);
#pragma unroll
        for (int i = 0; i < NI_T; ++i) {
            if (i < a.ni) {
                const float dx = px - stim[i][0], dy = py - stim[i][1], dz = pz - stim[i][2];
                const float e = __expf(-(dx * dx + dy * dy + dz * dz) * inv2s2);
                if (with_w) acc[i] += e * wgt; else acc[i] += e;
            }
        }
    };
    for (int g = threadIdx.x; g < G; g += 256) point(g, rfw[g], 0.f, false);
    ff_block_reduce<NI_T>(acc, &red[0][0], NI_T);
    if (threadIdx.x < a.ni) dens[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
    for (int h = 0; h < a.nhid; ++h) {
        const int* idx = sp.idx + ((size_t)s * a.nhid + h) * sp.ncon;
        const float* str = sp.str + ((size_t)s * a.nhid + h) * sp.ncon;
#pragma unroll
        for (int k = 0; k < 2 * NI_T; ++k) acc[k] = 0.f;
        for (int c = threadIdx.x; c < sp.ncon; c += 256) {
            const int g = idx[c];
            if (g >= 0 && g < G) point(g, rfw[g], a.J * str[c], true);
        }
        __syncthreads();
        ff_block_reduce<NI_T>(acc, &red[0][0], NI_T);
        if (threadIdx.x < a.ni) {
            const int i = threadIdx.x;
            const float num = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
            const float den = dens[i];
            const float ts = a.TH_sam[(size_t)s * a.nhid + h];
            const float thr = a.TH + ((ts > 0.f) - (ts < 0.f)) * __powf(fabsf(ts), a.a) * a.TH_d;
            const size_t o = ((size_t)s * a.ni + i) * a.nhid + h;
            const float q = num / den;
            a.out[o] = fmaxf(q - thr, 0.f);
            if (a.q) { a.q[o] = q; a.den[o] = den; }
        }
        __syncthreads();
    }
}

// Second pass of the connection-list form: dsig[s][h] = sum_{i,g} gq e d^2 / sig^3 (w_g - q) / den (and the same weighted by
// RF_w) with w_g = J str on the unit's list and 0 elsewhere: the -q part runs over all grid points (RF_w only), the w part
// over the list.  One workgroup per (sample, hidden unit), one exponential per (stimulus, point), as ff_backward_kernel.
template <int NI_T>
__global__ void __launch_bounds__(256) ff_backward_sparse_kernel(FFArgs a, FFSparse sp, const float* __restrict__ gq, float* __restrict__ dsig) {
    __shared__ float stim[NI_T][3];
    __shared__ float coef[NI_T], cq[NI_T];
    __shared__ float red[4][2 * NI_T];
    const int s = blockIdx.x / a.nhid, h = blockIdx.x % a.nhid;
    const int G = a.box * a.box * a.box;
    if (threadIdx.x < a.ni * 3) (&stim[0][0])[threadIdx.x] = a.stim[threadIdx.x];
    if (threadIdx.x < a.ni) {
        const size_t o = ((size_t)s * a.ni + threadIdx.x) * a.nhid + h;
        coef[threadIdx.x] = gq[o] / a.den[o];
        cq[threadIdx.x] = coef[threadIdx.x] * a.q[o];
    }
    __syncthreads();
    const float* rfw = a.RF_w + (size_t)s * G;
    const float step = (a.box > 1) ? 6.f / (float)(a.box - 1) : 0.f;
    float acc[2 * NI_T];
#pragma unroll
    for (int k = 0; k < 2 * NI_T; ++k) acc[k] = 0.f;
    auto point = [&](int g, float rw, const float* cf, float sgn) {
        const int iz = g % a.box, iy = (g / a.box) % a.box, ix = g / (a.box * a.box);
        const float px = -3.f + step * ix, py = -3.f + step * iy, pz = -3.f + step * iz;
        const float sig = rw * a.RF_d + a.RF_l;
        const float inv2s2 = 0.5f * __builtin_amdgcn_rcpf(sig * sig), inv_s3 = __builtin_amdgcn_rcpf(sig * sig * sig);
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NI_T; ++i) {
            if (i < a.ni) {
                const float dx = px - stim[i][0], dy = py - stim[i][1], dz = pz - stim[i][2];
                const float d2 = dx * dx + dy * dy + dz * dz;
                t += cf[i] * __expf(-d2 * inv2s2) * d2;
            }
        }
        acc[0] += sgn * t * inv_s3;
        acc[1] += sgn * t * inv_s3 * rw;
    };
    for (int g = threadIdx.x; g < G; g += 256) point(g, rfw[g], cq, -1.f);
    const int* idx = sp.idx + ((size_t)s * a.nhid + h) * sp.ncon;
    const float* str = sp.str + ((size_t)s * a.nhid + h) * sp.ncon;
    for (int c = threadIdx.x; c < sp.ncon; c += 256) {
        const int g = idx[c];
        if (g >= 0 && g < G) point(g, rfw[g], coef, a.J * str[c]);
    }
    ff_block_reduce<NI_T>(acc, &red[0][0], 2);
    if (threadIdx.x < 2) {
        float v = 0.f;
        for (int wv = 0; wv < 4; ++wv) v += red[wv][threadIdx.x];
        dsig[((size_t)s * a.nhid + h) * 2 + threadIdx.x] = v;
    }
}
hipError_t launch_ff_backward_sparse(const FFArgs& a, const int* idx, const float* str, int ncon, const float* gq, float* dsig, hipStream_t st) {
    if (a.nsam == 0) return hipSuccess;
    if (a.ni > FF_MAX_NI || a.ni < 1 || !a.q || !a.den || ncon < 0 || (ncon > 0 && (!idx || !str))) return hipErrorInvalidValue;
    const FFSparse sp{idx, str, ncon};
    hipLaunchKernelGGL((ff_backward_sparse_kernel<FF_MAX_NI>), dim3(a.nsam * a.nhid), dim3(256), 0, st, a, sp, gq, dsig);
    return hipGetLastError();
}

hipError_t launch_ff_forward_sparse(const FFArgs& a, const FFLattice* lat, const int* idx, const float* str, int ncon, hipStream_t st) {
    if (a.nsam == 0) return hipSuccess;
    if (a.ni > FF_MAX_NI || a.ni < 1 || ncon < 0 || (ncon > 0 && (!idx || !str))) return hipErrorInvalidValue;
    const FFSparse sp{idx, str, ncon};
    const int G = a.box * a.box * a.box;
    if (lat && a.ni == 27 && a.box % 4 == 0 && G % 4 == 0 && ((uintptr_t)a.RF_w % 16) == 0) {
        hipLaunchKernelGGL(ff_forward_sparse_lattice_kernel, dim3(a.nsam), dim3(256), 0, st, a, *lat, sp);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((ff_forward_sparse_kernel<FF_MAX_NI>), dim3(a.nsam), dim3(256), 0, st, a, sp);
    return hipGetLastError();
}

hipError_t launch_ff_forward(const FFArgs& a, const FFLattice* lat, hipStream_t st) {
    if (a.nsam == 0) return hipSuccess;
    if (a.ni > FF_MAX_NI || a.ni < 1) return hipErrorInvalidValue;
    if (lat && a.ni == 27) {
        hipLaunchKernelGGL(ff_forward_lattice_kernel, dim3(a.nsam * a.nhid), dim3(256), 0, st, a, *lat);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((ff_forward_kernel<FF_MAX_NI>), dim3(a.nsam * a.nhid), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_ff_backward(const FFArgs& a, const float* gq, float* dsig, hipStream_t st) {
    if (a.nsam == 0) return hipSuccess;
    if (a.ni > FF_MAX_NI || a.ni < 1 || !a.q || !a.den) return hipErrorInvalidValue;
    hipLaunchKernelGGL((ff_backward_kernel<FF_MAX_NI>), dim3(a.nsam * a.nhid), dim3(256), 0, st, a, gq, dsig);
    return hipGetLastError();
}

}  // namespace ssn
