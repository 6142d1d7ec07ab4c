// Device-side building blocks shared by the SSN kernels (gfx950 only).
//
// I/O nonlinearities follow tc_gan/ext/ssnode.c:25-53 (branch form) which is
// value-identical to the clip/where form of tc_gan/ssnode.py:129-149.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/ssnode_mi355x.h"

namespace ssn {

// Per-launch constants derived on the host in fp64, then narrowed to T.
template <typename T>
struct IoConsts {
    int io_type;   // SSN_IO_*
    T k, n;
    T v0;          // (soft/k)^(1/n)                        ssnode.c:21-23, 74
    T soft, hard;  // rate_soft_bound, rate_hard_bound
    T lin_slope;   // k * v0^(n-1) * n                      ssnode.c:41
    T tanh_gain;   // n * soft / ((hard - soft) * v0)       ssnode.c:51
    T span;        // hard - soft        (host-side so that it arrives as a kernel argument in an SGPR
    T span_gain;   // span * tanh_gain    instead of being recomputed into a VGPR by every wave)
    T log2k;       // log2(k): k v^n = 2^(n log2 v + log2 k) in the fp32 kernels
};

template <typename T>
struct StepConsts {
    T eps_E, eps_I;  // dt / tau_E, dt / tau_I             ssnode.c:72-73
    T atol;
    T hard_stop;     // bound of the code-2 test (ssnode.c:98-102); +inf disables
    int check_hard;  // 0 for SSN_IO_TANH (ssnode.c:168-185 has no such test)
    int max_iter;
};

template <typename T>
inline IoConsts<T> make_io_consts(const ssn_solver_params& p) {
    IoConsts<T> c;
    c.io_type = p.io_type;
    c.k = (T)p.k;
    c.n = (T)p.n;
    const double v0 = pow(p.rate_soft_bound / p.k, 1.0 / p.n);
    c.v0 = (T)v0;
    c.soft = (T)p.rate_soft_bound;
    c.hard = (T)p.rate_hard_bound;
    c.lin_slope = (T)(p.k * pow(v0, p.n - 1.0) * p.n);
    c.tanh_gain = (T)(p.n * p.rate_soft_bound / ((p.rate_hard_bound - p.rate_soft_bound) * v0));
    c.span = c.hard - c.soft;
    c.span_gain = c.span * c.tanh_gain;
    c.log2k = (T)log2(p.k);
    return c;
}

template <typename T>
inline StepConsts<T> make_step_consts(const ssn_solver_params& p) {
    StepConsts<T> c;
    c.eps_E = (T)(p.dt / p.tau_E);
    c.eps_I = (T)(p.dt / p.tau_I);
    c.atol = (T)p.atol;
    c.hard_stop = (T)p.rate_hard_bound;
    c.check_hard = (p.io_type != SSN_IO_TANH);
    c.max_iter = p.max_iter;
    return c;
}

// k * v^n for v > 0.
__device__ __forceinline__ float pow_rate(float v, float k, float n) {
    // v_log_f32 / v_exp_f32 are base-2, ~1 ulp each; |n*log2 v| <= ~25 here so the
    // relative error of the result stays below 1e-6 (tolerance of the path: 1e-4).
    return k * __builtin_amdgcn_exp2f(n * __builtin_amdgcn_logf(v));
}
__device__ __forceinline__ double pow_rate(double v, double k, double n) { return k * pow(v, n); }

__device__ __forceinline__ float tanh_pos(float x) {
    // x >= 0 here.  1 - 2/(1+e^{2x}); e^{2x} -> inf gives exactly 1.
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // 2*log2(e)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ double tanh_pos(double x) { return tanh(x); }

template <typename T>
__device__ __forceinline__ T io_eval(T v, const IoConsts<T>& c) {
    if (!(v > (T)0)) return (v != v) ? v : (T)0;   // v <= 0 -> 0; NaN propagates like pow(NaN)
    if (c.io_type == SSN_IO_POWER || v <= c.v0) return pow_rate(v, c.k, c.n);
    if (c.io_type == SSN_IO_LINEAR) return c.soft + c.lin_slope * (v - c.v0);
    return c.soft + c.span * tanh_pos(c.tanh_gain * (v - c.v0));
}

// fp32: the power law without the v > 0 test in front of it.  v = 0 -> 2^-inf = 0; v < 0 -> NaN, selected away;
// NaN stays NaN.  Same values as the generic form to ~1e-7 relative.
template <>
__device__ __forceinline__ float io_eval<float>(float v, const IoConsts<float>& c) {
    const float pw = __builtin_amdgcn_exp2f(fmaf(c.n, __builtin_amdgcn_logf(v), c.log2k));
    float f = (v < 0.f) ? 0.f : pw;
    if (c.io_type != SSN_IO_POWER && v > c.v0) {
        f = (c.io_type == SSN_IO_LINEAR) ? c.soft + c.lin_slope * (v - c.v0)
                                         : c.soft + c.span * tanh_pos(c.tanh_gain * (v - c.v0));
    }
    return f;
}

__device__ __forceinline__ float abs_t(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double abs_t(double x) { return __builtin_fabs(x); }

// Broadcast lane (16*(lane/16) + LANE16) of x to all lanes of its 16-lane DPP row.
template <int LANE16>
__device__ __forceinline__ float row_bcast(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + LANE16, 0xf, 0xf, true));
}
template <int LANE16>
__device__ __forceinline__ double row_bcast(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x150 + LANE16, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + LANE16, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

}  // namespace ssn
