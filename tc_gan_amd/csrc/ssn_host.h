// Host-side declarations shared between the kernel translation units and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "ssn_device.h"

namespace ssn {

template <typename T>
struct SolveArgs {
    const T* W;      // [B][M][M]
    const T* ext;    // [NB][M] or [B][NB][M]
    T* r;            // [B][NB][M] in/out (newest state)
    T* r_prev;       // [B][NB][M] or nullptr
    int* codes;      // [B][NB]
    int* steps;      // [B][NB] or nullptr
    int ext_per_draw;
    int B, NB, M, N;
    IoConsts<T> io;
    StepConsts<T> st;
    int split_narrow = 0;   // fp16-split solver: 1 = alternating two-group form (state as three parts) instead of the wide form
};

// Fixed-time generator forward (ssn_gen.hip)
template <typename T>
struct GenFwdArgs {
    const T* W;       // [B][M][M]
    const T* ext;     // [B][NB][M]
    T* time_avg;      // [B][NB][M]
    T* dyn_row;       // [B][NB][M]  sum over the window of (x_{t+1}-x_t)^2 per neuron
    T* rate_row;      // [B][NB][M]  sum over the window of relu(x_t - theta) per neuron
    T* traj;          // [B][NB][T][M] or nullptr: x_t, index t-1
    T* df;            // [B][NB][T][M] or nullptr: f'(u_t), index t-1
    int B, NB, M, seqlen, skip;
    T eps_E, eps_I, theta;
    IoConsts<T> io;
    int mfma_groups = 2;   // MFMA kernels: stimulus groups of 4 per workgroup (2, or 1 to spread few draws over the chip)
    int split_narrow = 0;  // fp16-split forward with two groups: 1 = the alternating two-group form (state as three parts,
                           // exact) instead of the wide form (all 8 stimuli in one chain, state as two parts)
};
// BPTT adjoint sweep (ssn_gen.hip)
template <typename T>
struct GenBwdArgs {
    const T* W;            // [B][M][M]
    const T* traj;         // [B][NB][T][M]
    T* delta;              // in: f'(u_t) (the forward's df); out: delta_t at index t-2, slot T-1 zero
    const T* g_time_avg;   // [B][NB][M]  dL/d time_avg
    T* g_ext;              // [B][NB][M] or nullptr: dL/d ext = sum_t delta_t
    int B, NB, M, seqlen, skip;
    T eps_E, eps_I, theta, c_dyn, c_rate;
    int mfma_groups = 2;
    int split_narrow = 0;  // as GenFwdArgs::split_narrow
    unsigned* dmax = nullptr;   // [B] or nullptr: atomic max of the bit patterns of |delta| the sweep stores for draw b (zeroed by
                                // the caller; the fp16-split sweeps of ssn_duo.hip / ssn_mfma16.hip only -- they track it for their own scaling)
};
template <typename T>
struct JDSv { T J[4], D[4], inv2s2[4], inv_s3[4]; };
template <typename T>
struct JDS { T J[4], D[4], inv2s2[4]; };
// One element of make_W_with_x (gradient_expressions/make_w_batch.py:8-34; weight_gen.py:13-26) from its z:
// W[pN + i][qN + j] = exp(-(x_i - x_j)^2 / (2 S_pq^2)) (+-J_pq +- D_pq z), x = linspace(-.5, .5, N), sign + for q = E, - for q = I.
// ONE definition for every kernel that forms W (build_w_kernel, the Philox and MT19937 forms that draw z in the same launch):
// products and sums rounded one by one (no contraction), so the same z gives the same bits wherever W is built.
// (in two halves, so that a kernel may keep the Gaussian factors of a launch in a table: the factor depends on the block pq and
// on |i - j| alone -- dx enters squared)
template <typename T>
__device__ __forceinline__ T w_gauss(const JDS<T>& p, T inv_nm1, int pq, int i_minus_j) {
#pragma clang fp contract(off)
    const T dx = (T)i_minus_j * inv_nm1;
    return exp(-(dx * dx) * p.inv2s2[pq]);
}
template <typename T>
__device__ __forceinline__ T w_combine_vals(T Jpq, T Dpq, int qq, T g, T z) {
#pragma clang fp contract(off)
    const T sgn = qq ? (T)-1 : (T)1;
    return g * (sgn * Jpq + sgn * Dpq * z);
}
template <typename T>
__device__ __forceinline__ T w_combine(const JDS<T>& p, int pq, int qq, T g, T z) {
    return w_combine_vals<T>(p.J[pq], p.D[pq], qq, g, z);
}
template <typename T>
__device__ __forceinline__ T w_from_z(const JDS<T>& p, int N, T inv_nm1, int row, int col, T z) {
    const int pp = row >= N, i = row - pp * N;
    const int qq = col >= N, j = col - qq * N;
    const int pq = pp * 2 + qq;
    return w_combine<T>(p, pq, qq, w_gauss<T>(p, inv_nm1, pq, i - j), z);
}
// inputs of a device-noise forward in one launch (ssn_aux.hip: gen_inputs_kernel)
struct GenInputsArgs {
    unsigned long long seed, off_z, off_zin;
    const float *bw, *con, *v; float inv_l; int bernoulli;
    float *W, *z, *zin, *amp, *ext;
    int B, NB, N;
    JDS<float> p; long total_vec, total_s, nb_w, nb_s;
};
hipError_t launch_gen_inputs(GenInputsArgs a, const float* jds12, hipStream_t st);
// gradient vector and loss of a generator step in one launch (ssn_gen_tail.hip)
struct GenGradsArgs {
    const double* part; int B;           // [B][4][3] from jds_grad_kernel
    int nv;                              // 0 no V, 1 one V for both populations, 2 (V_E, V_I)
    const float *g_ext, *ext_base, *zin; int NB, M;
    const float* dmean; const double* pens; double dynamics_cost, rate_cost;
    double* ws;                          // 2 * 128 + 1 doubles, the last 8 bytes zero before the first call
    float* out;                          // [nv + 12 + 1]
};
hipError_t launch_gen_grads(const GenGradsArgs& a, hipStream_t st);

// ssn_solver.hip
template <typename T> bool regw_supported(int M, int NB);
template <typename T> hipError_t launch_regw(const SolveArgs<T>& a, hipStream_t st);
template <typename T> hipError_t launch_stream(const SolveArgs<T>& a, hipStream_t st);

// ssn_tile.hip
template <typename T> bool tile_supported(int M, int NB);
template <typename T> hipError_t launch_tile(const SolveArgs<T>& a, hipStream_t st, int shape = 0);   // 0 default, 1 split residency, 2 all-register

// ssn_ssgrad.hip
template <typename T> hipError_t launch_build_dw(const T* z, const T* jds12, int which, T* dW, int B, int N, hipStream_t st);
template <typename T> hipError_t launch_ss_system(const T* R, const T* W, const T* dW, int dw_per_draw, const T* I, int i_per_draw,
                                                  const IoConsts<T>& io, int nz, int nb, int M, T* A, T* rhs, hipStream_t st);

template <typename T> hipError_t launch_lu_solve(T* A, T* rhs, int* info, int nsys, int M, int nrhs, hipStream_t st);

// ssn_mfma.hip (fp32, NB >= 4)
bool gen_mfma_supported(int M, int NB);
hipError_t launch_gen_forward_mfma(const GenFwdArgs<float>& a, hipStream_t st);
// ssn_mfma16.hip: fp16-split matrix-core forward; gen_split_rshift < 0: not applicable (I/O function without a rate bound, sizes)
int gen_split_rshift(const GenFwdArgs<float>& a);
hipError_t launch_gen_forward_split(const GenFwdArgs<float>& a, hipStream_t st);
int gen_split_wide_parts();   // two-group launches: 2 / 3 = wide form with that many state parts, 0 = alternating form
bool gen_split_backward_supported(int M, int NB);
bool solve_split_supported(const SolveArgs<float>& a);
hipError_t launch_solve_split(const SolveArgs<float>& a, hipStream_t st);
hipError_t launch_gen_backward_split(const GenBwdArgs<float>& a, hipStream_t st);
// ssn_duo.hip: two draws per workgroup, every wave both roles (wide operand layout, state as two fp16 parts by round to nearest)
hipError_t launch_gen_forward_duo(const GenFwdArgs<float>& a, hipStream_t st);
hipError_t launch_solve_duo(const SolveArgs<float>& a, hipStream_t st);
hipError_t launch_gen_backward_duo(const GenBwdArgs<float>& a, hipStream_t st);
// adjoint sweep + dL/dW in one launch (ssn_fuse.hip): a.delta = f'(u), read only; gW [B][M][M]; xmax >= max |traj|
bool gen_backward_fused_supported(const GenBwdArgs<float>& a, float xmax);
hipError_t launch_gen_backward_fused(const GenBwdArgs<float>& a, float* gW, float xmax, hipStream_t st);
hipError_t launch_gen_backward_mfma(const GenBwdArgs<float>& a, hipStream_t st);
hipError_t launch_solve_mfma(const SolveArgs<float>& a, hipStream_t st);

// ssn_gen.hip
template <typename T> bool gen_supported(int M);
// ssn_gen_stream.hip: any even 2N <= 2048 (fallback, W streamed from memory every step)
bool gen_stream_supported(int M);
template <typename T> hipError_t launch_gen_forward_stream(const GenFwdArgs<T>& a, hipStream_t st);
template <typename T> hipError_t launch_gen_backward_stream(const GenBwdArgs<T>& a, hipStream_t st);
template <typename T> hipError_t launch_gen_forward(const GenFwdArgs<T>& a, hipStream_t st);
template <typename T> hipError_t launch_gen_backward(const GenBwdArgs<T>& a, hipStream_t st);
template <typename T> hipError_t launch_jds_grad(const T* gW, const T* z, const T* jds12, double* out, int B, int N, hipStream_t st);

// ssn_gw.hip: gW[b] = delta[b]^T traj[b] over K rows ([K][M] row-major each); kernel 0 auto, 1 plain FMA, 2 split-bf16 MFMA
template <typename T> hipError_t launch_weight_grad(const T* delta, const T* traj, T* gW, int B, long K, int M, int kernel, hipStream_t st);
// the same on two fp16 parts per operand (3 partial products): dmax[b] bounds |delta[b]| (bit pattern, device), xmax bounds |traj|
hipError_t launch_weight_grad_scaled(const float* delta, const float* traj, float* gW, int B, long K, int M, const unsigned* dmax,
                                     float xmax, hipStream_t st);

// ssn_critic.hip
struct OptArgs {
    float* p; const float* g; float* s1; float* s2; long n;
    float lr, a_t, beta1, beta2, eps, rho;
    float l2_penalty, l1_penalty, l2_decay, l1_decay;
    float clip_lo, clip_hi; int clip;
    int kind;    // 0 sgd, 1 adam, 2 rmsprop
    // the whole update is skipped when *gate > gate_bound (device value, read by every thread; nullptr = always update):
    // the critic step that cwgan.py:493-498 drops when the batch's rate penalty exceeds its bound, decided on the device
    const double* gate = nullptr; double gate_bound = 0.0;
    // per-element clip bounds (instead of clip_lo / clip_hi when given), and the record of a generator step: record[e] = the
    // new value, record[n] = *record_tail (the step's loss), written by the thread of element 0
    const float* clip_lo_v = nullptr; const float* clip_hi_v = nullptr;
    float* record = nullptr; const float* record_tail = nullptr;
    // the whole update is skipped when any of the n gradient elements is not finite (n <= 64: the generator's handful of
    // parameters): parameters and optimizer state stay, record[e] = the old values, record[n] = NaN -- the caller recomputes
    // the gradient (a draw whose fp16 adjoint outgrew its scale) and applies that
    int skip_nonfinite = 0;
};
size_t critic_workspace_floats(const int* dims, int nlayers, int batch_gd, int batch_p);
hipError_t critic_forward(const float* params, const int* dims, int nlayers, const float* x, const float* cond, int batch,
                          int hide_cell_type, float* out, float* ws, bool bf16, hipStream_t st, float leak = 0.f);
hipError_t critic_forward2(const float* params, const int* dims, int nlayers, const float* xa, const float* ca, int na,
                           const float* xb, const float* cb, int nb, int hide_cell_type, float* out, float* ws, bool bf16,
                           hipStream_t st, float leak = 0.f, bool inputs_ready = false);
// (inputs_ready: ws already starts with the input block of [xa; xb] -- critic_loss_grad on the same rows leaves it there)
hipError_t critic_loss_grad(const float* params, const int* dims, int nlayers, const float* xg, const float* cg,
                            const float* xd, const float* cd, const float* xp, const float* cp, int ng, int nd, int np,
                            float lmd, int hide_cell_type, float* grads, float* stats, float* dvals, float* ws, bool bf16,
                            hipStream_t st, float leak = 0.f, const float* eps = nullptr, float* xp_out = nullptr);
// (leak: slope of the hidden nonlinearity below zero -- 0 rectify, 0.01 leaky_rectify, 1/3 very_leaky_rectify, 1 linear)
hipError_t critic_input_grad(const float* params, const int* dims, int nlayers, const float* x, const float* cond, int batch,
                             int hide_cell_type, float scale, float* gx, float* stats, float* ws, bool bf16, hipStream_t st,
                             float leak = 0.f);
hipError_t optimizer_step(const OptArgs& o, hipStream_t st);
// ssn_critic_ln.hip (per-layer LayerNorm flags; norm == nullptr -> all plain layers)
size_t critic_norm_workspace_floats(const int* dims, int nlayers, int batch_gd, int batch_p);
// (norm[l]: bit 0 layer normalisation, bit 1 learnable scale after it; act: 0 rectify, 1 leaky_rectify, 2 very_leaky_rectify,
// 3 linear, 4 tanh, 5 sigmoid, 6 softplus, 7 elu)
long critic_act_num_params(const int* dims, const int* flags, int nlayers);
hipError_t critic_norm_forward(const float* params, const int* dims, const int* norm, int nlayers, const float* x,
                               const float* cond, int batch, int hide, float* out, float* ws, bool bf16, hipStream_t st, int act = 0);
hipError_t critic_norm_input_grad(const float* params, const int* dims, const int* norm, int nlayers, const float* x,
                                  const float* cond, int batch, int hide, float scale, float* gx, float* stats, float* ws,
                                  bool bf16, hipStream_t st, int act = 0);
hipError_t critic_norm_loss_grad(const float* params, const int* dims, const int* norm, int nlayers, const float* xg,
                                 const float* cg, const float* xd, const float* cd, const float* xp, const float* cp, int ng,
                                 int nd, int np, float lmd, int hide, float* grads, float* stats, float* dvals, float* ws,
                                 bool bf16, hipStream_t st, int act = 0);

// ssn_critic_rows.hip: wide plain critics (hidden widths multiples of 32 up to 512), the row-local part of an update in one launch
struct RowsPackSeg { const float* src; unsigned short* dst; int nin, nout, NT, KS, kind; long start; };
struct RowsPackArgs { RowsPackSeg seg[19]; int nseg; long total; float* zero; long nzero; };    // zero[0:nzero] = 0 by the blocks behind `total`
struct RowsArgs {
    int L, dims[10];
    const float* b[9]; const float* wout; float leak;
    const unsigned short* pf[9];     // B fragments of op(B)(k, n) = W_l[k][n]   (forward, second chain of the penalty)
    const unsigned short* pb[9];     // ... of op(B)(k, i) = W_l[i][k]            (backward chains)
    const unsigned short* po;        // ... of the column w_out
    float* h[10]; float* v[10]; float* up; float* dvals;           // rows of [xg; xd]: activations, backward chain, upstream, D
    float* hp[10]; float* vp[10]; float* ep[10]; float* dnorm;     // penalty rows: ..., second chain, norm - 1 per row
    int ng, nd, np, nx;
    int mode;                        // 2: the update; 0: D of the ng + nd rows of h[0] only; 1: D and the input gradient of the np rows of hp[0]
    float* gx; float scale;          // mode 1: gx[np][nx] = scale * dD/dx
};
bool critic_rows_supported(const int* dims, int nlayers);
size_t critic_rows_workspace_floats(const int* dims, int nlayers, int batch_p);
hipError_t critic_rows_pack(const float* params, const int* dims, int L, float* ws_pack, RowsArgs& ra, float* zero, long nzero, hipStream_t st);
hipError_t critic_rows_launch(const RowsArgs& ra, hipStream_t st);

// ssn_critic_fused.hip: critics whose layer widths are all <= 128 (3 launches per update; fp32 arithmetic)
bool critic_fused_supported(const int* dims, int nlayers);
size_t critic_fused_workspace_floats(const int* dims, int nlayers, int batch_gd, int batch_p);
hipError_t critic_fused_forward(const float* params, const int* dims, const int* norm, int nlayers, const float* x,
                                const float* cond, int batch, int hide, float* out, float* ws, hipStream_t st);
hipError_t critic_fused_input_grad(const float* params, const int* dims, const int* norm, int nlayers, const float* x,
                                   const float* cond, int batch, int hide, float scale, float* gx, float* stats, float* ws,
                                   hipStream_t st);
hipError_t critic_fused_loss_grad(const float* params, const int* dims, const int* norm, int nlayers, const float* xg,
                                  const float* cg, const float* xd, const float* cd, const float* xp, const float* cp, int ng,
                                  int nd, int np, float lmd, int hide, float* grads, float* stats, float* dvals, float* ws,
                                  hipStream_t st);

// ssn_ff.hip
struct FFArgs {
    const float* RF_w;     // [nsam][G]
    const float* FF_con;   // [nsam][nhid][G]
    const float* FF_str;   // [nsam][nhid][G]
    const float* TH_sam;   // [nsam][nhid]
    const float* stim;     // [ni][3]
    float* out;            // [nsam][ni][nhid]
    float* q;              // [nsam][ni][nhid] or nullptr: pre-threshold drive
    float* den;            // [nsam][ni][nhid] or nullptr: sum_g e
    int nsam, nhid, ni, box;
    float RF_l, RF_d, TH, TH_d, J, a;
};
struct FFLattice { float x[3], y[3], z[3]; };      // stimuli = {x} x {y} x {z}, i = (a*3 + b)*3 + c
hipError_t launch_ff_forward(const FFArgs& a, const FFLattice* lat, hipStream_t st);
hipError_t launch_ff_backward(const FFArgs& a, const float* gq, float* dsig, hipStream_t st);
hipError_t launch_ff_forward_sparse(const FFArgs& a, const FFLattice* lat, const int* idx, const float* str, int ncon, hipStream_t st);
hipError_t launch_ff_backward_sparse(const FFArgs& a, const int* idx, const float* str, int ncon, const float* gq, float* dsig, hipStream_t st);

// ssn_aux.hip
hipError_t launch_moment_sums(const float* x, int B, int D, double* sums, hipStream_t st);
hipError_t launch_moment_loss_grad(const float* x, const double* sums, double Bg, const double* data_moments,
                                   const double* weights, int B, int D, float* gx, double* out, hipStream_t st);
// jds12_dev: J, D, S as device T[12] instead of the host's jds12 (which may then be null)
template <typename T> hipError_t launch_build_w(const T* z, const T* jds12, T* W, int B, int N, hipStream_t st, const T* jds12_dev = nullptr);
template <typename T> hipError_t launch_stimulus(const T* bw, const T* con, T smooth, const T* amp, T* ext, int B, int NB, int N, hipStream_t st);
hipError_t launch_stimulus_hetero(const float* bw, const float* con, float smooth, const float* zin, const float* v, int nv, float* ext,
                                  int B, int NB, int N, hipStream_t st);
template <typename T> hipError_t launch_io_eval(const T* v, T* out, long count, const IoConsts<T>& io, hipStream_t st);
template <typename T> hipError_t launch_philox_uniform(unsigned long long seed, unsigned long long offset, T* out, unsigned long long n, hipStream_t st);
template <typename T> hipError_t launch_build_w_philox(unsigned long long seed, unsigned long long offset, const T* jds12, T* W, T* zout, int B, int N, hipStream_t st);
template <typename T> hipError_t launch_philox_amp(unsigned long long seed, unsigned long long offset, const T* v, T* zin, T* amp, unsigned long long n, int M, int bernoulli, hipStream_t st);
template <typename T> hipError_t launch_probe_scatter(const T* g, const long* ids, const long* probes, T* g_ta, int n, int B, int NB, int M, hipStream_t st);
long segment_sqnorms_ws_doubles(int n);
hipError_t launch_segment_sqnorms(const float* x, const long* bounds, int n, float* out, double* ws, hipStream_t st);
hipError_t launch_interpolate(const float* eps, const float* xd, const float* xg, float* xp, int rows, int cols, hipStream_t st);
hipError_t launch_step_head(const double* pens, const float* stats, float* tail, hipStream_t st);
hipError_t launch_step_finish(const float* params, const long* bounds, int nseg, double* ws, const float* dvals, int ng, int nd,
                              const double* pens, const float* stats, float* tail, hipStream_t st);
hipError_t launch_mean_diff(const float* d, int ng, int nd, float* out, hipStream_t st);
template <typename T> hipError_t launch_penalty_means(const T* dyn, const T* rate, long n, double scale_dyn, double scale_rate, double* ws, double* out, hipStream_t st,
                                                      const T* time_avg = nullptr, const long* ids = nullptr, const long* probes = nullptr, T* tc = nullptr,
                                                      int nsamp = 0, int NB = 0, int M = 0);
template <typename T> hipError_t launch_dot(const T* x, const T* y, T* out, int dim, hipStream_t st);

// ssn_mt19937.hip: numpy's RandomState.random_sample on the device (key / pos: the host state, in/out)
int mt19937_jump_poly(unsigned long long nblocks, unsigned long long* bits);
hipError_t mt19937_draw(unsigned int* key, int* pos, unsigned long long total, unsigned long long skip, unsigned long long count,
                        void* out, int elem, hipStream_t st);
// What the caller's stream holds right behind the `total` doubles and is consumed with them: `total` elements, of which
// [skip, skip + count) are written to `out` as fp32.  kind 1: rng.choice(2, n) * 2 - 1 (one 32-bit output per element),
// kind 2: rng.rand(n) * 2 - 1 (one double per element), 0: nothing.
struct MtTail { int kind = 0; unsigned long long total = 0, skip = 0, count = 0; float* out = nullptr; };
// ticket == nullptr: the numbers only -- the state after the draw is not computed (a second window of a draw begun before)
hipError_t mt19937_begin(const unsigned int* key, int pos, unsigned long long total, unsigned long long skip, unsigned long long count,
                         void* out, int elem, hipStream_t st, int* ticket, float* W = nullptr, const float* jds12 = nullptr, int N = 0,
                         const MtTail* tail = nullptr);
hipError_t mt19937_finish(int ticket, unsigned int* key, int* pos);
bool mt19937_plan(int pos, unsigned long long total, unsigned long long skip, unsigned long long count, long* out, const MtTail* tail = nullptr);


}  // namespace ssn
