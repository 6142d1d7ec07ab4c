// Pieces shared by the matrix-core forms of the SSN recurrence (ssn_mfma.hip: fp32 MFMA; ssn_mfma16.hip: fp16-split MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include "ssn_device.h"

namespace ssn {

typedef float mf4 __attribute__((ext_vector_type(4)));

// Branch-free f(v), f'(v): the serial part must stay in the basic block of the MFMA chain it hides behind, so the
// lane-dependent cases (v <= 0, v <= v0) are selects and the (uniform) I/O type is folded into coefficients --
// a select on a uniform condition would be turned back into a branch.  Same values as io_eval_grad (ssn_gen.hip).
struct IoSelect {
    float k, n, log2k, v0_low, soft, gain, c_lin, c_tanh, c_tanh_gain;
    __device__ __forceinline__ explicit IoSelect(const IoConsts<float>& c) {
        const bool lin = c.io_type == SSN_IO_LINEAR, th = c.io_type == SSN_IO_TANH;
        k = c.k; n = c.n; log2k = __builtin_log2f(c.k); soft = c.soft; gain = c.tanh_gain;
        v0_low = (c.io_type == SSN_IO_POWER) ? __builtin_inff() : c.v0;    // v <= v0_low: power-law branch
        c_lin = lin ? c.lin_slope : 0.f;
        c_tanh = th ? c.span : 0.f;
        c_tanh_gain = th ? c.span_gain : 0.f;
        v0 = c.v0;
    }
    float v0;
    // WANT_DF = false skips f' (only the backward needs it); the saturating branch is evaluated only when some lane
    // of the wave is above v0 (rates above the soft bound are rare): one wave-uniform branch.
    template <bool WANT_DF>
    __device__ __forceinline__ void eval4(const float (&v)[4], float (&f)[4], float (&df)[4]) const {
        // The power law k v^n = 2^(n log2 v + log2 k) is evaluated for all four values UNCONDITIONALLY (the empty asm
        // pins it): left alone, the compiler sinks log/exp under a `v > 0` branch per value, which serialises the
        // four dependent chains.  v = 0 gives 2^-inf = 0 by itself, v < 0 gives NaN and is selected away, NaN stays NaN.
        float pw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pw[i] = __builtin_amdgcn_exp2f(fmaf(n, __builtin_amdgcn_logf(v[i]), log2k));
        asm volatile("" : "+v"(pw[0]), "+v"(pw[1]), "+v"(pw[2]), "+v"(pw[3]));
        float rv[4];
        if (WANT_DF) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rv[i] = n * pw[i] * __builtin_amdgcn_rcpf(v[i]);
            asm volatile("" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[i] = (v[i] < 0.f) ? 0.f : pw[i];
            if (WANT_DF) df[i] = (v[i] > 0.f) ? rv[i] : 0.f;
        }
        float vmax;                               // NaN operands are ignored by v_max: they take no saturating branch
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(vmax) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
        asm("v_max_f32 %0, %1, %2" : "=v"(vmax) : "v"(vmax), "v"(v[3]));
        const bool any_high = vmax > v0_low;
        if (__builtin_amdgcn_ballot_w64(any_high) != 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = v[i] - v0;
                const float th = tanh_pos(gain * d);
                const bool high = v[i] > v0_low;
                f[i] = high ? fmaf(c_tanh, th, fmaf(c_lin, d, soft)) : f[i];
                if (WANT_DF) df[i] = high ? fmaf(c_tanh_gain, 1.f - th * th, c_lin) : df[i];
            }
        }
    }
    // The same for NV values at once (the two-draw kernels of ssn_duo.hip finish 6 or 8 values per lane and step).
    template <bool WANT_DF, int NV>
    __device__ __forceinline__ void evaln(const float (&v)[NV], float (&f)[NV], float (&df)[NV]) const {
        float pw[NV], rv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            pw[i] = __builtin_amdgcn_exp2f(fmaf(n, __builtin_amdgcn_logf(v[i]), log2k));
            asm volatile("" : "+v"(pw[i]));
        }
        if (WANT_DF) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                rv[i] = n * pw[i] * __builtin_amdgcn_rcpf(v[i]);
                asm volatile("" : "+v"(rv[i]));
            }
        }
        float vmax = v[0];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f[i] = (v[i] < 0.f) ? 0.f : pw[i];
            if (WANT_DF) df[i] = (v[i] > 0.f) ? rv[i] : 0.f;
        }
        // max of the NV values, two per instruction (v_max / v_max3 ignore NaN operands: NaN takes no saturating branch)
#pragma unroll
        for (int i = 1; i < NV; i += 2) {
            if (i + 1 < NV) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(vmax) : "v"(vmax), "v"(v[i]), "v"(v[i + 1]));
            else asm("v_max_f32 %0, %1, %2" : "=v"(vmax) : "v"(vmax), "v"(v[i]));
        }
        if (__builtin_amdgcn_ballot_w64(vmax > v0_low) != 0) {
            // all NV saturating values unconditionally (pinned: left alone the compiler puts each one under its own
            // `v > v0` branch, NV dependent exp / rcp chains one after the other), then selects.  The empty asm keeps this
            // block behind its wave-uniform branch: for small NV the compiler would otherwise run it every time.
            asm volatile("");
            float th[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                th[i] = tanh_pos(gain * fmaxf(v[i] - v0, 0.f));
                asm volatile("" : "+v"(th[i]));
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float d = v[i] - v0;
                const bool high = v[i] > v0_low;
                f[i] = high ? fmaf(c_tanh, th[i], fmaf(c_lin, d, soft)) : f[i];
                if (WANT_DF) df[i] = high ? fmaf(c_tanh_gain, 1.f - th[i] * th[i], c_lin) : df[i];
            }
        }
    }
};

}  // namespace ssn
