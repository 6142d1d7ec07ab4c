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
