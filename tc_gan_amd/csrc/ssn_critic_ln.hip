// The layer-by-layer critic in its general form (networks/simple_discriminator.py:6-75, 139-165): per hidden layer
//   plain:       Dense + bias -> nonlinearity
//   normalised:  Dense(no bias) -> LayerNorm (no parameters, biased variance, eps = 1e-4 of Lasagne's BatchNormLayer)
//                -> [ScaleLayer: one learnable factor per unit, for every nonlinearity but rectify, :57-75] -> Bias -> nonlinearity
// with ANY of lasagne.nonlinearities' rectify / leaky_rectify / very_leaky_rectify / linear / tanh / sigmoid / softplus / elu
// (`--disc-nonlinearity`, run/bptt_wgan.py:153), including the WGAN-GP double backward through the normalisation, the scale
// and the curvature of a smooth nonlinearity.  The rectify-only fast paths are ssn_critic.hip and ssn_critic_fused.hip.
//
// Per layer l (a = h_{l-1} W_l; y = LN(a) or a; pre = g_l y + b_l (g_l = 1 without scale); h_l = f(pre); s = f'(pre), t = f''(pre)):
//   LNback(x; y, 1/sd) = (x - mean(x) - y mean(x.y)) / sd          (row-wise, symmetric operator)
//   input-gradient chain   p_l = s_l u_l,  c_l = LNback(g_l p_l),  u_{l-1} = c_l W_l^T,  g = u_0
//   parameter gradient of sum_b up_b D_b:  dW_l += h_{l-1}^T c_l;  db_l += colsum(p_l);  dg_l += colsum(p_l y_l)
//   penalty backward, sweep 1 (l = 1..L, du_0 = dP/dg):
//       dW_l += du_{l-1}^T c_l;  dc_l = du_{l-1} W_l;  dq_l = LNback(dc_l);  dg_l += colsum(dq_l p_l);  dp_l = g_l dq_l;
//       dyA_l = -(dc_l q + g_l p_l r)/sd  with q = mean(g p.y), r = mean(dc.y);   dsA_l = -sum(dc.c)/sd;
//       du_l = s_l dp_l;   preA_l = dp_l u_l t_l  (the curvature term: zero for the piecewise-linear nonlinearities);
//       dw_out += colsum(du_L)
//   sweep 2 (l = L..1, dh_L = 0): ordinary backprop of the a_l-dependence:
//       dpre = preA_l + s_l dh_l;  db_l += colsum(dpre);  dg_l += colsum(dpre y_l);  dy = dyA_l + g_l dpre;
//       da = LNback(dy) + dsA_l y / n;  dW_l += h_{l-1}^T da;  dh_{l-1} = da W_l^T
// (derivation in DESIGN.md section 3.8; checked against torch autograd of oracle/gan_torch.py).
#include <hip/hip_runtime.h>
#include "ssn_host.h"

namespace ssn {

// from ssn_critic.hip
hipError_t critic_gemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                       int M, int N, int K, float alpha, float beta, bool bf16, hipStream_t st);
hipError_t critic_colsum(const float* X, float* out, int batch, int n, float beta, hipStream_t st);
hipError_t critic_make_input(const float* x, const float* cond, float* h0, int batch, int nx, int hide, hipStream_t st);
hipError_t critic_gp_head(const float* g, float* ghat, float* pen, int batch, int n0, int nx, hipStream_t st);
hipError_t critic_two_means(const float* d, float* out, int ng, int nd, hipStream_t st);
hipError_t critic_loss_combine(float* stats, float lmd, hipStream_t st);
size_t critic_splitk_scratch_floats(const int* dims, int nlayers, int rows);
void critic_splitk_begin(float* scratch, size_t cap);
hipError_t critic_splitk_flush(hipStream_t st);
hipError_t critic_gather_scale(const float* v0, float* gx, int batch, int n0, int nx, float s, hipStream_t st);

constexpr float LN_EPS = 1e-4f;

// hidden nonlinearity: kind 0 = x > 0 ? x : leak x (rectify 0, leaky_rectify 0.01, very_leaky_rectify 1/3, linear 1),
// 1 tanh, 2 sigmoid, 3 softplus (log1p(exp x)), 4 elu (x > 0 ? x : expm1 x)  -- lasagne.nonlinearities
struct ActSpec { int kind; float leak; };
__device__ __forceinline__ float act_f(const ActSpec a, float x) {
    switch (a.kind) {
        case 1: return tanhf(x);
        case 2: return 1.f / (1.f + __expf(-x));
        case 3: return x > 20.f ? x : log1pf(__expf(x));
        case 4: return x > 0.f ? x : expm1f(x);
        default: return x > 0.f ? x : a.leak * x;
    }
}
__device__ __forceinline__ float act_d1(const ActSpec a, float x) {
    switch (a.kind) {
        case 1: { const float t = tanhf(x); return 1.f - t * t; }
        case 2: { const float g = 1.f / (1.f + __expf(-x)); return g * (1.f - g); }
        case 3: return 1.f / (1.f + __expf(-x));
        case 4: return x > 0.f ? 1.f : __expf(x);
        default: return x > 0.f ? 1.f : a.leak;
    }
}
__device__ __forceinline__ float act_d2(const ActSpec a, float x) {
    switch (a.kind) {
        case 1: { const float t = tanhf(x); return -2.f * t * (1.f - t * t); }
        case 2: { const float g = 1.f / (1.f + __expf(-x)); return g * (1.f - g) * (1.f - 2.f * g); }
        case 3: { const float g = 1.f / (1.f + __expf(-x)); return g * (1.f - g); }
        case 4: return x > 0.f ? 0.f : __expf(x);
        default: return 0.f;
    }
}

__device__ __forceinline__ float wave_sum_f(float x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// One wave per row.  forward  a -> (y, invs, h = f(g y + b));  ln = 0: y = a.
__global__ void __launch_bounds__(64) ln_forward_kernel(const float* __restrict__ a, const float* __restrict__ gamma,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        float* __restrict__ invs, float* __restrict__ h, int n, int ln,
                                                        const ActSpec act) {
    const long row = blockIdx.x;
    const float* ar = a + row * n;
    float mu = 0.f, is = 1.f;
    if (ln) {
        float s = 0.f;
        for (int j = threadIdx.x; j < n; j += 64) s += ar[j];
        mu = wave_sum_f(s) / n;
        float v = 0.f;
        for (int j = threadIdx.x; j < n; j += 64) { const float d = ar[j] - mu; v += d * d; }
        is = rsqrtf(wave_sum_f(v) / n + LN_EPS);
    }
    for (int j = threadIdx.x; j < n; j += 64) {
        const float yy = (ar[j] - mu) * is;
        y[row * n + j] = yy;
        h[row * n + j] = act_f(act, (gamma ? gamma[j] : 1.f) * yy + bias[j]);
    }
    if (threadIdx.x == 0) invs[row] = is;
}

// p = f'(pre) x;  out = LNback(g p; y, invs)   (ln = 0: out = g p);  py = p y when asked for (the scale's gradient)
__global__ void __launch_bounds__(64) ln_back_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ bias, const float* __restrict__ y,
                                                     const float* __restrict__ invs, float* __restrict__ out,
                                                     float* __restrict__ pout, float* __restrict__ py, int n, int ln,
                                                     const ActSpec act) {
    const long row = blockIdx.x;
    float sx = 0.f, sxy = 0.f;
    for (int j = threadIdx.x; j < n; j += 64) {
        const long o = row * n + j;
        const float g = gamma ? gamma[j] : 1.f;
        const float v = act_d1(act, g * y[o] + bias[j]) * x[o];
        pout[o] = v;
        if (py) py[o] = v * y[o];
        sx += g * v; sxy += g * v * y[o];
    }
    const float mx = wave_sum_f(sx) / n, mxy = wave_sum_f(sxy) / n, is = invs[row];
    for (int j = threadIdx.x; j < n; j += 64) {
        const long o = row * n + j;
        const float v = (gamma ? gamma[j] : 1.f) * pout[o];
        out[o] = ln ? (v - mx - y[o] * mxy) * is : v;
    }
}

// Sweep-1 row op: given dc, p, c, y, invs, u:  dq = LNback(dc);  dp = g dq;  du = f' dp;  preA = dp u f'';  gprod = dq p;
// dyA = -(dc q + g p r) invs;  dsA = -sum(dc c) invs
__global__ void __launch_bounds__(64) ln_sweep1_kernel(const float* __restrict__ dc, const float* __restrict__ p,
                                                       const float* __restrict__ c, const float* __restrict__ y,
                                                       const float* __restrict__ invs, const float* __restrict__ gamma,
                                                       const float* __restrict__ bias, const float* __restrict__ u,
                                                       float* __restrict__ du, float* __restrict__ dyA,
                                                       float* __restrict__ dsA, float* __restrict__ preA,
                                                       float* __restrict__ gprod, int n, int ln, const ActSpec act) {
    const long row = blockIdx.x;
    float sdc = 0.f, sdcy = 0.f, spy = 0.f, sdcc = 0.f;
    for (int j = threadIdx.x; j < n; j += 64) {
        const long o = row * n + j;
        const float pg = (gamma ? gamma[j] : 1.f) * p[o];
        sdc += dc[o]; sdcy += dc[o] * y[o]; spy += pg * y[o]; sdcc += dc[o] * c[o];
    }
    const float mdc = wave_sum_f(sdc) / n, r = wave_sum_f(sdcy) / n, q = wave_sum_f(spy) / n, is = invs[row];
    const float tot = wave_sum_f(sdcc);
    for (int j = threadIdx.x; j < n; j += 64) {
        const long o = row * n + j;
        const float g = gamma ? gamma[j] : 1.f;
        const float pre = g * y[o] + bias[j];
        const float dq = ln ? (dc[o] - mdc - y[o] * r) * is : dc[o];
        const float dp = g * dq;
        du[o] = act_d1(act, pre) * dp;
        dyA[o] = ln ? -(dc[o] * q + g * p[o] * r) * is : 0.f;
        if (preA) preA[o] = dp * u[o] * act_d2(act, pre);
        if (gprod) gprod[o] = dq * p[o];
    }
    if (threadIdx.x == 0) dsA[row] = ln ? -tot * is : 0.f;
}

// Sweep-2 row op: dpre = preA + f' dh (each may be null = 0);  dy = dyA + g dpre;  da = LNback(dy) + dsA y / n
// (ln = 0: da = dy);  dprey = dpre y when asked for
__global__ void __launch_bounds__(64) ln_sweep2_kernel(const float* __restrict__ dh, const float* __restrict__ gamma,
                                                       const float* __restrict__ bias, const float* __restrict__ preA,
                                                       const float* __restrict__ dyA, const float* __restrict__ dsA,
                                                       const float* __restrict__ y, const float* __restrict__ invs,
                                                       float* __restrict__ dpre, float* __restrict__ da,
                                                       float* __restrict__ dprey, int n, int ln, const ActSpec act) {
    const long row = blockIdx.x;
    float sd = 0.f, sdy = 0.f;
    for (int j = threadIdx.x; j < n; j += 64) {
        const long o = row * n + j;
        const float g = gamma ? gamma[j] : 1.f;
        float dp = preA ? preA[o] : 0.f;
        if (dh) dp += act_d1(act, g * y[o] + bias[j]) * dh[o];
        dpre[o] = dp;
        if (dprey) dprey[o] = dp * y[o];
        const float d = (dyA ? dyA[o] : 0.f) + g * dp;
        sd += d; sdy += d * y[o];
    }
    const float md = wave_sum_f(sd) / n, mdy = wave_sum_f(sdy) / n, is = invs[row];
    const float ds = dsA ? dsA[row] : 0.f;
    for (int j = threadIdx.x; j < n; j += 64) {
        const long o = row * n + j;
        const float d = (dyA ? dyA[o] : 0.f) + (gamma ? gamma[j] : 1.f) * dpre[o];
        da[o] = ln ? (d - md - y[o] * mdy) * is + ds * y[o] / n : d;
    }
}

// p_L[b][k] = (h_L > 0) * w_out[k] * up[b]
__global__ void __launch_bounds__(256) top_seed_kernel(const float* __restrict__ wout, const float* __restrict__ up,
                                                       float* __restrict__ uL, int batch, int nL) {
    for (long e = blockIdx.x * 256L + threadIdx.x; e < (long)batch * nL; e += gridDim.x * 256L)
        uL[e] = wout[e % nL] * (up ? up[e / nL] : 1.f);
}
// upstream of D per stacked row: +1/ng (generated), -1/nd (data), 1 (penalty rows: plain input gradient of D)
__global__ void __launch_bounds__(256) fill_updown2_kernel(float* up, int ng, int nd, int np) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ng + nd + np; i += gridDim.x * 256)
        up[i] = (i < ng) ? 1.f / (float)ng : ((i < ng + nd) ? -1.f / (float)nd : 1.f);
}
__global__ void __launch_bounds__(256) scale_kernel(float* x, float a, long n) {
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += gridDim.x * 256L) x[e] *= a;
}

static int nblk(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }

struct NormNet {
    int L; int dims[10]; int ln[9];
    const float* W[9]; const float* g[9]; const float* b[9]; const float* wout; long offW[9], offg[9], offb[9], offout, nparams;
    ActSpec act; bool smooth, scaled;
};
// layer flags: bit 0 = layer normalisation, bit 1 = learnable scale after it (parameter order W, scales, b: lasagne's
// get_all_params of Dense(no bias) -> LayerNorm -> ScaleLayer -> BiasLayer).  act: the public activation codes.
static bool act_from_code(int code, ActSpec& a) {
    switch (code) {
        case 0: a = {0, 0.f}; return true;            // rectify
        case 1: a = {0, 0.01f}; return true;          // leaky_rectify
        case 2: a = {0, 1.f / 3.f}; return true;      // very_leaky_rectify
        case 3: a = {0, 1.f}; return true;            // linear / identity
        case 4: a = {1, 0.f}; return true;            // tanh
        case 5: a = {2, 0.f}; return true;            // sigmoid
        case 6: a = {3, 0.f}; return true;            // softplus
        case 7: a = {4, 0.f}; return true;            // elu
        default: return false;
    }
}
static bool parse_norm_net(const float* params, const int* dims, const int* norm, int nlayers, NormNet& net, int act_code = 0) {
    if (nlayers < 0 || nlayers > 8) return false;
    if (!act_from_code(act_code, net.act)) return false;
    net.smooth = net.act.kind != 0;
    net.scaled = false;
    net.L = nlayers;
    long off = 0;
    for (int l = 0; l <= nlayers; ++l) net.dims[l] = dims[l];
    for (int l = 0; l < nlayers; ++l) {
        const int flags = norm ? norm[l] : 0;
        if (flags < 0 || flags > 3 || flags == 2) return false;      // a scale only follows a normalisation
        net.ln[l] = flags & 1;
        net.offW[l] = off; net.W[l] = params ? params + off : nullptr; off += (long)dims[l] * dims[l + 1];
        net.g[l] = nullptr; net.offg[l] = -1;
        if (flags & 2) { net.offg[l] = off; net.g[l] = params ? params + off : nullptr; off += dims[l + 1]; net.scaled = true; }
        net.offb[l] = off; net.b[l] = params ? params + off : nullptr; off += dims[l + 1];
    }
    net.offout = off; net.wout = params ? params + off : nullptr; off += dims[nlayers];
    net.nparams = off;
    return true;
}
long critic_act_num_params(const int* dims, const int* flags, int nlayers) {
    NormNet net;
    return parse_norm_net(nullptr, dims, flags, nlayers, net) ? net.nparams : -1;
}

struct Acts {            // per batch of rows
    float *h[10], *y[10], *invs[10], *u[10], *p[10], *c[10], *py[10];
};
static float* carve(float*& p, long n) { float* r = p; p += n; return r; }
static void carve_acts(float*& p, const NormNet& net, int rows, Acts& A) {
    for (int l = 0; l <= net.L; ++l) {
        A.h[l] = carve(p, (long)rows * net.dims[l]);
        A.y[l] = carve(p, (long)rows * net.dims[l]);
        A.invs[l] = carve(p, rows);
        A.u[l] = carve(p, (long)rows * net.dims[l]);
        A.p[l] = carve(p, (long)rows * net.dims[l]);
        A.c[l] = carve(p, (long)rows * net.dims[l]);
        A.py[l] = net.scaled ? carve(p, (long)rows * net.dims[l]) : nullptr;
    }
}

static hipError_t norm_forward(const NormNet& net, const Acts& A, float* dout, int rows, bool bf16, hipStream_t st) {
    hipError_t e;
    for (int l = 0; l < net.L; ++l) {
        const int nin = net.dims[l], nout = net.dims[l + 1];
        // a -> stored temporarily in u[l+1]
        if ((e = critic_gemm(A.h[l], nin, 1, net.W[l], nout, 1, A.u[l + 1], nout, rows, nout, nin, 1.f, 0.f, bf16, st)) != hipSuccess) return e;
        hipLaunchKernelGGL(ln_forward_kernel, dim3(rows), dim3(64), 0, st, A.u[l + 1], net.g[l], net.b[l], A.y[l + 1], A.invs[l + 1],
                           A.h[l + 1], nout, net.ln[l], net.act);
    }
    return critic_gemm(A.h[net.L], net.dims[net.L], 1, net.wout, 1, 1, dout, 1, rows, 1, net.dims[net.L], 1.f, 0.f, bf16, st);
}

// input-gradient chain given per-row upstream `up` of D: fills p, c, u; u[0] = dD/dh0 (times up)
static hipError_t norm_chain(const NormNet& net, const Acts& A, const float* up, int rows, bool bf16, hipStream_t st) {
    hipError_t e;
    const int L = net.L;
    hipLaunchKernelGGL(top_seed_kernel, dim3(nblk((long)rows * net.dims[L])), dim3(256), 0, st, net.wout, up, A.u[L], rows, net.dims[L]);
    for (int l = L; l >= 1; --l) {
        const int nin = net.dims[l - 1], nout = net.dims[l];
        // p_l = f'(pre_l) u_l ; c_l = LNback(g_l p_l)
        hipLaunchKernelGGL(ln_back_kernel, dim3(rows), dim3(64), 0, st, A.u[l], net.g[l - 1], net.b[l - 1], A.y[l], A.invs[l], A.c[l],
                           A.p[l], net.g[l - 1] ? A.py[l] : nullptr, nout, net.ln[l - 1], net.act);
        // u_{l-1} = c_l W_l^T
        if ((e = critic_gemm(A.c[l], nout, 1, net.W[l - 1], 1, nout, A.u[l - 1], nin, rows, nin, nout, 1.f, 0.f, bf16, st)) != hipSuccess) return e;
    }
    return hipGetLastError();
}

// standard parameter gradient of sum_b up_b D_b given the chain results (c_l = dL/da_l, p_l = dL/dpre_l)
static hipError_t norm_param_grads(const NormNet& net, const Acts& A, const float* up, float* grads, int rows, bool bf16, hipStream_t st) {
    hipError_t e;
    const int L = net.L;
    // w_out: sum_b up_b h_L
    if ((e = critic_gemm(A.h[L], 1, net.dims[L], up, 1, 1, grads + net.offout, 1, net.dims[L], 1, rows, 1.f, 1.f, bf16, st)) != hipSuccess) return e;
    for (int l = 1; l <= L; ++l) {
        const int nin = net.dims[l - 1], nout = net.dims[l];
        if ((e = critic_gemm(A.h[l - 1], 1, nin, A.c[l], nout, 1, grads + net.offW[l - 1], nout, nin, nout, rows, 1.f, 1.f, bf16, st)) != hipSuccess) return e;
        if ((e = critic_colsum(A.p[l], grads + net.offb[l - 1], rows, nout, 1.f, st)) != hipSuccess) return e;
        if (net.g[l - 1] && (e = critic_colsum(A.py[l], grads + net.offg[l - 1], rows, nout, 1.f, st)) != hipSuccess) return e;
    }
    return hipSuccess;
}

size_t critic_norm_workspace_floats(const int* dims, int nlayers, int batch_gd, int batch_p) {
    long per_row = 0, maxd = 0;
    for (int l = 0; l <= nlayers; ++l) { per_row += dims[l]; if (dims[l] > maxd) maxd = dims[l]; }
    // (sized for the general form: p y per row for the scales' gradient, the curvature term and one product array per penalty row)
    const long acts = 6 * per_row + (nlayers + 1);
    return (size_t)((long)batch_gd * (acts + 2) + (long)batch_p * (acts + 7 * per_row + maxd + (nlayers + 1) + 2) + 2 * maxd + 64) +
           critic_splitk_scratch_floats(dims, nlayers, batch_gd + batch_p);
}

hipError_t critic_norm_forward(const float* params, const int* dims, const int* norm, int nlayers, const float* x,
                               const float* cond, int batch, int hide, float* out, float* ws, bool bf16, hipStream_t st, int act) {
    NormNet net;
    if (!parse_norm_net(params, dims, norm, nlayers, net, act)) return hipErrorInvalidValue;
    hipError_t e;
    float* p = ws;
    Acts A;
    carve_acts(p, net, batch, A);
    if ((e = critic_make_input(x, cond, A.h[0], batch, dims[0] - (cond ? 3 : 0), hide, st)) != hipSuccess) return e;
    return norm_forward(net, A, out, batch, bf16, st);
}

hipError_t critic_norm_input_grad(const float* params, const int* dims, const int* norm, int nlayers, const float* x,
                                  const float* cond, int batch, int hide, float scale, float* gx, float* stats, float* ws,
                                  bool bf16, hipStream_t st, int act) {
    NormNet net;
    if (!parse_norm_net(params, dims, norm, nlayers, net, act)) return hipErrorInvalidValue;
    hipError_t e;
    float* p = ws;
    Acts A;
    carve_acts(p, net, batch, A);
    float* dv = carve(p, batch);
    if ((e = critic_make_input(x, cond, A.h[0], batch, dims[0] - (cond ? 3 : 0), hide, st)) != hipSuccess) return e;
    if ((e = norm_forward(net, A, dv, batch, bf16, st)) != hipSuccess) return e;
    if ((e = critic_two_means(dv, stats, batch, 0, st)) != hipSuccess) return e;
    if ((e = norm_chain(net, A, nullptr, batch, bf16, st)) != hipSuccess) return e;
    return critic_gather_scale(A.u[0], gx, batch, dims[0], dims[0] - (cond ? 3 : 0), scale, st);
}

hipError_t critic_norm_loss_grad(const float* params, const int* dims, const int* norm, int nlayers, const float* xg,
                                 const float* cg, const float* xd, const float* cd, const float* xp, const float* cp, int ng,
                                 int nd, int np, float lmd, int hide, float* grads, float* stats, float* dvals, float* ws,
                                 bool bf16, hipStream_t st, int act) {
    NormNet net;
    if (!parse_norm_net(params, dims, norm, nlayers, net, act)) return hipErrorInvalidValue;
    hipError_t e;
    const int nc = (cg || cd || cp) ? 3 : 0;            // (no condition columns: the unconditional critic, networks/wgan.py:66-97)
    if (nc && ((ng && !cg) || (nd && !cd) || (np && !cp))) return hipErrorInvalidValue;
    const int L = nlayers, nx = dims[0] - nc, bgd = ng + nd;
    if ((e = hipMemsetAsync(grads, 0, net.nparams * sizeof(float), st)) != hipSuccess) return e;
    float* p = ws;
    // The rows of the three inputs are STACKED ([xg; xd; xp]): one forward pass and one backward chain over
    // bgd + np rows instead of two of each (every row depends on its own input only; the upstream of D is +1/ng, -1/nd
    // for the first two blocks and 1 for the penalty rows).  These passes are launch-bound (~5 us per kernel whatever
    // its size), so the saving is the ~19 launches per update.  A = all rows, used with `bgd` rows where only the first
    // two blocks matter; P = the penalty rows of the same arrays.
    const int rows = bgd + np;
    Acts A;
    carve_acts(p, net, rows, A);
    float* up = carve(p, rows);
    float* dall = carve(p, rows);
    Acts P;
    for (int l = 0; l <= L; ++l) {
        const long off = (long)bgd * dims[l];
        P.h[l] = A.h[l] + off; P.y[l] = A.y[l] + off; P.u[l] = A.u[l] + off; P.p[l] = A.p[l] + off; P.c[l] = A.c[l] + off;
        P.py[l] = A.py[l] ? A.py[l] + off : nullptr;
        P.invs[l] = A.invs[l] + bgd;
    }
    float *du[10], *dc[10], *dyA[10], *dsA[10], *dpre[10], *da[10], *preA[10];
    long maxd = 0;
    for (int l = 0; l <= L; ++l) {
        du[l] = carve(p, (long)np * dims[l]); dc[l] = carve(p, (long)np * dims[l]); dyA[l] = carve(p, (long)np * dims[l]);
        dpre[l] = carve(p, (long)np * dims[l]); da[l] = carve(p, (long)np * dims[l]); dsA[l] = carve(p, np);
        preA[l] = net.smooth ? carve(p, (long)np * dims[l]) : nullptr;
        if (dims[l] > maxd) maxd = dims[l];
    }
    float* prod = net.scaled ? carve(p, (long)np * maxd) : nullptr;       // dq p, then dpre y: summed over the rows at once
    // (dh_{l-1} of sweep 2 is written into dc[l-1], which is free by then)
    critic_splitk_begin(p, critic_splitk_scratch_floats(dims, nlayers, rows));      // the rest of the workspace
    struct PlanScope { ~PlanScope() { critic_splitk_begin(nullptr, 0); } } plan_scope;    // closed on every return path
    if ((e = critic_make_input(xg, cg, A.h[0], ng, nx, hide, st)) != hipSuccess) return e;
    if ((e = critic_make_input(xd, cd, A.h[0] + (long)ng * dims[0], nd, nx, hide, st)) != hipSuccess) return e;
    if ((e = critic_make_input(xp, cp, P.h[0], np, nx, hide, st)) != hipSuccess) return e;
    if ((e = norm_forward(net, A, dall, rows, bf16, st)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(dvals, dall, sizeof(float) * bgd, hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
    if ((e = critic_two_means(dall, stats, ng, nd, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(fill_updown2_kernel, dim3(nblk(rows)), dim3(256), 0, st, up, ng, nd, np);
    if ((e = norm_chain(net, A, up, rows, bf16, st)) != hipSuccess) return e;
    // ---- (1) mean D(xg) - mean D(xd): parameter gradients from the first bgd rows ---------------------
    if ((e = norm_param_grads(net, A, up, grads, bgd, bf16, st)) != hipSuccess) return e;
    // ---- (2) gradient penalty on the last np rows -----------------------------------------------------
    // du_0 = lmd * dP/dg
    if ((e = critic_gp_head(P.u[0], du[0], stats + 2, np, dims[0], nx, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(scale_kernel, dim3(nblk((long)np * dims[0])), dim3(256), 0, st, du[0], lmd, (long)np * dims[0]);
    // sweep 1
    for (int l = 1; l <= L; ++l) {
        const int nin = dims[l - 1], nout = dims[l];
        if ((e = critic_gemm(du[l - 1], 1, nin, P.c[l], nout, 1, grads + net.offW[l - 1], nout, nin, nout, np, 1.f, 1.f, bf16, st)) != hipSuccess) return e;
        if ((e = critic_gemm(du[l - 1], nin, 1, net.W[l - 1], nout, 1, dc[l], nout, np, nout, nin, 1.f, 0.f, bf16, st)) != hipSuccess) return e;
        float* gp = net.g[l - 1] ? prod : nullptr;
        hipLaunchKernelGGL(ln_sweep1_kernel, dim3(np), dim3(64), 0, st, dc[l], P.p[l], P.c[l], P.y[l], P.invs[l], net.g[l - 1],
                           net.b[l - 1], P.u[l], du[l], dyA[l], dsA[l], preA[l], gp, nout, net.ln[l - 1], net.act);
        if (gp && (e = critic_colsum(gp, grads + net.offg[l - 1], np, nout, 1.f, st)) != hipSuccess) return e;
    }
    // u_L = w_out (the same row for every sample): dw_out += colsum(du_L)
    if ((e = critic_colsum(du[L], grads + net.offout, np, dims[L], 1.f, st)) != hipSuccess) return e;
    // sweep 2
    const float* dh_cur = nullptr;
    for (int l = L; l >= 1; --l) {
        const int nin = dims[l - 1], nout = dims[l];
        float* gp = net.g[l - 1] ? prod : nullptr;
        hipLaunchKernelGGL(ln_sweep2_kernel, dim3(np), dim3(64), 0, st, dh_cur, net.g[l - 1], net.b[l - 1], preA[l], dyA[l], dsA[l],
                           P.y[l], P.invs[l], dpre[l], da[l], gp, nout, net.ln[l - 1], net.act);
        if ((e = critic_colsum(dpre[l], grads + net.offb[l - 1], np, nout, 1.f, st)) != hipSuccess) return e;
        if (gp && (e = critic_colsum(gp, grads + net.offg[l - 1], np, nout, 1.f, st)) != hipSuccess) return e;
        if ((e = critic_gemm(P.h[l - 1], 1, nin, da[l], nout, 1, grads + net.offW[l - 1], nout, nin, nout, np, 1.f, 1.f, bf16, st)) != hipSuccess) return e;
        if (l > 1) {
            if ((e = critic_gemm(da[l], nout, 1, net.W[l - 1], 1, nout, dc[l - 1], nin, np, nin, nout, 1.f, 0.f, bf16, st)) != hipSuccess) return e;
            dh_cur = dc[l - 1];
        }
    }
    if ((e = critic_loss_combine(stats, lmd, st)) != hipSuccess) return e;
    return critic_splitk_flush(st);                   // grads += the split GEMMs' slabs, in slice order
}

}  // namespace ssn
