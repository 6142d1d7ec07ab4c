/*
 * ssnode_mi355x.h -- C ABI of libssnode.so (MI355X / gfx950 build).
 *
 * The library is a drop-in for the reference's tc_gan/ext/libssnode.so (built
 * from tc_gan/ext/ssnode.c, bound by tc_gan/clib.py:16-33) plus an additive
 * batched ABI for the GPU-resident hot path.  Plain C: pointers and sizes
 * only, no C++/torch types.  Every entry point runs on the GPU; there is no
 * CPU fallback -- without a usable HIP device the solver symbols return
 * SSN_ERR_BASE + hipError_t (> 900, the range the reference's Python reserves
 * for "library error", tc_gan/ssnode.py:267-268).
 *
 * Conventions
 *   M = 2N neurons (E block first, then I).  W is row-major, W[i*M + j] is
 *   the weight from neuron j to neuron i (ssnode.c:64-67 `W + dim * i`).
 *   "device" pointers are hipMalloc'ed (or torch CUDA tensor) addresses on
 *   the CURRENT HIP device of the calling thread; `stream` is a hipStream_t
 *   passed as void* (NULL = the default stream).  Batched calls are
 *   asynchronous on `stream` unless stated otherwise.
 */
#ifndef SSNODE_MI355X_H
#define SSNODE_MI355X_H

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 * 1. Drop-in symbols: exactly the exports of tc_gan/ext/ssnode.c.
 * ------------------------------------------------------------------------ */

/* Replaces ssnode.c:69-109 / 111-151 / 153-187 (ctypes decl clib.py:16-26).
 * Host fp64 buffers, caller owned: W[M*M], ext[M] read-only; r0[M] in/out;
 * r1[M] scratch.  Runs the fp64 HIP kernel (one workgroup), synchronous.
 * Returns 0 converged / 1 max_iter / 2 reached rate_hard_bound (power and
 * linear only) / >900 HIP failure.  On return the two caller buffers hold
 * what the reference's pointer-swapping loop leaves in them (newest state in
 * r0 on code 0; parity-dependent on codes 1 and 2 -- DESIGN.md "buffer
 * parity").  Re-entrant, no global mutable state beyond the HIP runtime. */
int solve_dynamics_asym_power_euler(int N, double *W, double *ext, double k, double n,
                                    double *r0, double *r1, double tau_E, double tau_I,
                                    double dt, int max_iter, double atol,
                                    double rate_soft_bound, double rate_hard_bound);
int solve_dynamics_asym_linear_euler(int N, double *W, double *ext, double k, double n,
                                     double *r0, double *r1, double tau_E, double tau_I,
                                     double dt, int max_iter, double atol,
                                     double rate_soft_bound, double rate_hard_bound);
int solve_dynamics_asym_tanh_euler(int N, double *W, double *ext, double k, double n,
                                   double *r0, double *r1, double tau_E, double tau_I,
                                   double dt, int max_iter, double atol,
                                   double rate_soft_bound, double rate_hard_bound);

/* Replace ssnode.c:21-53 (clib.py:28-33) and ssnode.c:10-19.  Scalar
 * conveniences with the reference's signatures; each call evaluates the
 * device function on the GPU for one element (see ssn_io_eval_f64 for the
 * array form).  NaN on HIP failure. */
double io_pow(double v, double r0, double r1, double v0, double k, double n);
double io_alin(double v, double r0, double r1, double v0, double k, double n);
double io_atanh(double v, double r0, double r1, double v0, double k, double n);
double rate_to_volt(double rate, double k, double n);
double dot(int dim, const double *x, const double *y);

/* ------------------------------------------------------------------------
 * 2. Additive batched ABI (new symbols).
 * ------------------------------------------------------------------------ */

#define SSN_IO_POWER  0   /* 'asym_power'  : ssnode.c io_pow   */
#define SSN_IO_LINEAR 1   /* 'asym_linear' : ssnode.c io_alin  */
#define SSN_IO_TANH   2   /* 'asym_tanh'   : ssnode.c io_atanh */
#define SSN_ERR_BASE  1000

/* Solver parameters; same meaning as the scalar arguments of the drop-in
 * symbols (ssnode.c:55-62) / tc_gan.ssnode.fixed_point (ssnode.py:159-166). */
typedef struct ssn_solver_params {
    int io_type;             /* SSN_IO_* */
    int max_iter;
    double k, n;
    double tau_E, tau_I;
    double dt;
    double atol;
    double rate_soft_bound;
    double rate_hard_bound;  /* for POWER/LINEAR: the rate_stop_at bound (ssnode.py:241-242); +inf disables */
} ssn_solver_params;

/* Library / device probes (host only). */
int         ssn_abi_version(void);
int         ssn_device_count(void);          /* <0: -hipError_t */
const char *ssn_last_error(void);            /* thread-local, "" if none */
/* Which register-resident kernel family covers a (M, NB, dtype) solve: 2 = "tile" kernel
 * (fp32: 2N <= 208, fp64: 2N <= 208), 1 = DPP kernel only, 0 = none (generic streaming
 * kernel).  The automatic dispatch of ssn_solve_batch_* additionally prefers the fp32 MFMA
 * kernels (variant 6 where it applies, else 5) for NB >= 4, 104 < 2N <= 208 and >= 192 (draw, 8 stimuli) workgroups.
 * dtype_bytes is 4 or 8. */
int         ssn_solver_fast_path(int M, int NB, int dtype_bytes);

/*
 * Batched fixed-point solve: B weight draws x NB stimuli.  Replaces the
 * thread-pool-over-ctypes loop of tc_gan/ssnode.py:423-510 for one round of
 * candidates.
 *   W      device [B][M][M]
 *   ext    device [NB][M] (ext_per_draw = 0) or [B][NB][M] (ext_per_draw = 1)
 *   r      device [B][NB][M]  in: initial rates; out: newest state
 *   r_prev device [B][NB][M] or NULL; out: state before the last executed step
 *   codes  device int32 [B][NB]: 0 / 1 / 2 as the drop-in symbols
 *   steps  device int32 [B][NB] or NULL: Euler steps executed
 * Every (draw, stimulus) pair stops independently at its own convergence /
 * bound step, exactly like an individual reference call.
 * Returns 0 or SSN_ERR_BASE + hipError_t (launch errors only; async).
 */
int ssn_solve_batch_f32(const float *W, const float *ext, int ext_per_draw,
                        float *r, float *r_prev, int *codes, int *steps,
                        int B, int NB, int M, const ssn_solver_params *p, void *stream);
int ssn_solve_batch_f64(const double *W, const double *ext, int ext_per_draw,
                        double *r, double *r_prev, int *codes, int *steps,
                        int B, int NB, int M, const ssn_solver_params *p, void *stream);
/* Force a kernel variant (testing / A-B benchmarking): 0 = generic streaming
 * kernel, 1 = register-stationary DPP kernel, 2 = register-stationary tile kernel
 * (shape chosen by the library), 3 = tile kernel with split VGPR/LDS residency,
 * 4 = tile kernel with the whole tile in VGPRs, 5 = fp32 MFMA kernel (NB >= 4), 6 = fp16-split MFMA kernel
 * (NB >= 4, asym_tanh, dt <= tau: W and the state carried as two fp16 parts each, exact products, all 8 stimuli in one
 * chain per step), 7 = the same in the alternating two-group form (state as three parts, exact), 8 = fp16-split kernel
 * with two draws per workgroup (csrc/ssn_duo.hip; what the automatic choice takes for more than 256 (draw, 8 stimuli)
 * units); error if the size has no instantiation; negative = automatic (MFMA kernel for
 * large fp32 batches with NB >= 4, otherwise tile > DPP > streaming). */
int ssn_solve_batch_f32_variant(int variant, const float *W, const float *ext, int ext_per_draw,
                                float *r, float *r_prev, int *codes, int *steps,
                                int B, int NB, int M, const ssn_solver_params *p, void *stream);
int ssn_solve_batch_f64_variant(int variant, const double *W, const double *ext, int ext_per_draw,
                                double *r, double *r_prev, int *codes, int *steps,
                                int B, int NB, int M, const ssn_solver_params *p, void *stream);

/* The variant (numbering above) that ssn_solve_batch_f32 / _f64 (dtype_bytes 4 / 8) picks by itself for this call shape
 * and these parameters under the current operand-precision setting; -1: the call would be refused.  Nothing is launched.
 * For benchmarks and logs that must name the kernel that ran. */
int ssn_solve_batch_variant_for(int B, int NB, int M, int dtype_bytes, const ssn_solver_params *p);

/* Operand precision of the AUTOMATIC kernel choice (ssn_solve_batch_* without a variant, ssn_gen_params.kernel = 0) on the
 * fp32 paths with NB >= 4 stimuli per draw.  Process-wide, atomic, takes effect with the next call; explicit variants /
 * kernel codes are never overridden.
 *   1 (initial value): the fp16-split matrix-core kernels where they apply (saturating I/O function, dt <= tau): W and the
 *      state enter the products as two fp16 parts each (23 significant bits by round to nearest in the two-draw form,
 *      22 by truncation of the state in the older wide form), every product exact, fp32 accumulation;
 *   0: fp32 operands only (fp32 MFMA / VALU kernels): the reference's floatX arithmetic.
 * The environment variable SSN_FWD_SPLIT=0 selects 0 as the initial value.  Returns the previous setting. */
int ssn_set_operand_precision(int mode);
int ssn_get_operand_precision(void);

/* Host-buffer convenience forms of the above (numpy callers): allocate,
 * copy in, solve, copy out, synchronise.  Same argument meaning, host
 * pointers.  The PCIe copies are inside the call. */
int ssn_solve_batch_host_f32(const float *W, const float *ext, int ext_per_draw,
                             float *r, float *r_prev, int *codes, int *steps,
                             int B, int NB, int M, const ssn_solver_params *p);
int ssn_solve_batch_host_f64(const double *W, const double *ext, int ext_per_draw,
                             double *r, double *r_prev, int *codes, int *steps,
                             int B, int NB, int M, const ssn_solver_params *p);

/*
 * Connectivity from noise: W[b] = make_W_with_x(z[b]; J, D, S)
 * (gradient_expressions/make_w_batch.py:8-34 == weight_gen.py:13-26).
 *   z, W   device [B][M][M];  J, D, S  HOST float/double[4] row-major 2x2
 *   (EE, EI, IE, II) -- signs applied inside (+ for E columns, - for I).
 */
int ssn_build_w_f32(const float *z, const float *J, const float *D, const float *S,
                    float *W, int B, int N, void *stream);
int ssn_build_w_f64(const double *z, const double *J, const double *D, const double *S,
                    double *W, int B, int N, void *stream);
/* ssn_build_w_f32 with J, D, S read on the DEVICE: jds_dev = device float[12] (J, D, S, each 2x2 row-major) -- e.g. the
 * parameter vector an ssn_gen_apply_f32 queued on the same stream is about to update, so that the next forward can be queued
 * before the host has read the new values.  Same arithmetic, same bits as ssn_build_w_f32 of the same values. */
int ssn_build_w_devparams_f32(const float *z, const float *jds_dev, float *W, int B, int N, void *stream);

/*
 * Stimulus: ext[b][s][pN + i] = c[b][s] * sig((x_i + bw/2)/l) * sig((bw/2 - x_i)/l),
 * x = linspace(-.5, .5, N), duplicated for E and I
 * (stimuli.py:3-10; networks/ssn.py:167-193).
 *   bandwidths, contrasts  device [B][NB];  ext device [B][NB][2N]
 */
int ssn_stimulus_f32(const float *bandwidths, const float *contrasts, float smoothness,
                     float *ext, int B, int NB, int N, void *stream);
int ssn_stimulus_f64(const double *bandwidths, const double *contrasts, double smoothness,
                     double *ext, int B, int NB, int N, void *stream);

/* ------------------------------------------------------------------------
 * 3. Fixed-time generator (BPTT path): replaces the compiled Theano graphs of
 *    tc_gan/networks/ssn.py (EulerSSNCore.get_output_for 555-576, EulerSSNModel
 *    598-633) and the reverse scan that theano.grad builds for
 *    GeneratorTrainer (networks/wgan.py:218-260).
 * ------------------------------------------------------------------------ */
typedef struct ssn_gen_params {
    int io_type;                 /* SSN_IO_* */
    int seqlen;                  /* T: Euler steps from r = 0 */
    int skip_steps;              /* first output index of the measurement window */
    int kernel;                  /* 0 library default, 1 VALU tile kernels, 2 fp32 MFMA kernels (fp32, NB >= 4), 3 the
                                  * same with one 4-stimulus group per workgroup (few draws: more workgroups), 4 / 5
                                  * forward: fp16-split MFMA kernel (W carried as two fp16 parts = 22 significant bits,
                                  * the state as three = exact, every product exact, fp32 accumulation; asym_tanh only)
                                  * with two / one group per workgroup -- with two groups all 8 stimuli share one MFMA
                                  * chain per step and the state enters it as two fp16 parts (22 bits, like W); 6 = two
                                  * groups in the alternating form, which carries the state as three parts (exact);
                                  * backward: the adjoint sweep in the alternating form
                                  * (W^T as two fp16 parts, delta as three with a scale that follows max |delta| step
                                  * by step; any I/O function); 8 = fp16-split forward with TWO DRAWS per workgroup
                                  * (csrc/ssn_duo.hip: the chain of one draw behind the serial part of the other, W and
                                  * state as two fp16 parts each by round to nearest = 23 bits; backward: the adjoint sweep in
                                  * the same two-draw form, W^T and delta as two parts each, any I/O function).  0 picks
                                  * 8 / 4 / 5 where they apply (8 when there are more than 256 (draw, 8 stimuli) units)
                                  * unless ssn_set_operand_precision(0) -- or SSN_FWD_SPLIT=0 as the initial value --
                                  * keeps the automatic choice on fp32 operands. */
    double k, n;
    double tau_E, tau_I, dt;     /* eps = dt / tau per neuron */
    double rate_soft_bound, rate_hard_bound;
    double rate_penalty_threshold;
} ssn_gen_params;

/* 1 if the register-stationary generator kernels cover this size (2N <= 208 fp32, <= 104 fp64). */
int ssn_gen_supported(int M, int dtype_bytes);
/* Which fp32 forward kernel ssn_gen_forward_f32 runs for this call shape and p->kernel (the numbering of p->kernel:
 * 1 VALU tile / streaming kernels, 2 / 3 fp32 MFMA, 4 / 5 / 6 / 8 fp16-split MFMA: 4 two groups in the wide form, 5 one
 * group, 6 two groups in the alternating form, 8 two draws per workgroup; 7: wide form with an exact state, SSN_FWD_WIDE=3);
 * save != 0: with trajectory stores;
 * -1: the call would be refused.  For benchmarks and tests that must name the kernel they measured. */
int ssn_gen_forward_variant(int B, int NB, int M, int seqlen, int save, const ssn_gen_params *p);

/*
 * Forward: r_{t+1} = (1-eps) r_t + eps f(W r_t + ext), r_0 = 0, T = seqlen steps.
 *   W        device [B][M][M];  ext device [B][NB][M]
 *   time_avg device [B][NB][M]  mean over output indices >= skip_steps
 *   dyn_row  device [B][NB][M]  SUM over the window of (x_{t+1}-x_t)^2, per neuron -- or, from the two-draw kernel (8),
 *            per group of neurons booked on one of them (zeros elsewhere): only sum(dyn_row) is defined
 *   rate_row device [B][NB][M]  the same for relu(x_t - threshold)
 *            (dynamics_penalty = sum(dyn_row)/(B*(T-skip-1)*NB*M), rate_penalty =
 *             sum(rate_row)/(B*(T-skip)*NB*M): the means of ssn.py:626,632)
 *   traj, df device [B][NB][T][M] or both NULL: trajectory and f'(u_t) kept for the backward
 */
int ssn_gen_forward_f32(const float *W, const float *ext, float *time_avg, float *dyn_row, float *rate_row,
                        float *traj, float *df, int B, int NB, int M, const ssn_gen_params *p, void *stream);
int ssn_gen_forward_f64(const double *W, const double *ext, double *time_avg, double *dyn_row, double *rate_row,
                        double *traj, double *df, int B, int NB, int M, const ssn_gen_params *p, void *stream);
/*
 * Backward (adjoint sweep).  g_time_avg = dL/d time_avg [B][NB][M]; c_dyn, c_rate = the
 * coefficients multiplying sum(dyn_row) and sum(rate_row) in L.  `df_delta` holds the
 * forward's df on entry and delta on exit, shifted so that
 *     dL/dW[b] = df_delta[b].reshape(NB*T, M)^T @ traj[b].reshape(NB*T, M).
 */
int ssn_gen_backward_f32(const float *W, const float *traj, float *df_delta, const float *g_time_avg,
                         double c_dyn, double c_rate, int B, int NB, int M, const ssn_gen_params *p, void *stream);
int ssn_gen_backward_f64(const double *W, const double *traj, double *df_delta, const double *g_time_avg,
                         double c_dyn, double c_rate, int B, int NB, int M, const ssn_gen_params *p, void *stream);
/* Same sweep, additionally writing g_ext[B][NB][M] = dL/d ext = sum_t delta_t (gradient path of the
 * input-variability parameter V of the heterogeneous-input SSN). */
int ssn_gen_backward_ext_f32(const float *W, const float *traj, float *df_delta, const float *g_time_avg, float *g_ext,
                             double c_dyn, double c_rate, int B, int NB, int M, const ssn_gen_params *p, void *stream);
int ssn_gen_backward_ext_f64(const double *W, const double *traj, double *df_delta, const double *g_time_avg, double *g_ext,
                             double c_dyn, double c_rate, int B, int NB, int M, const ssn_gen_params *p, void *stream);
/*
 * dL/dW of the BPTT update (the `theano.grad` of networks/wgan.py:236-242 through the scan of ssn.py:354-385):
 * gW[b][i][j] = sum_k delta[b][k][i] * traj[b][k][j],  delta / traj = the [B][K = NB*T][M] views of what
 * ssn_gen_backward_* and ssn_gen_forward_* leave, gW device [B][M][M].  kernel: 0 automatic; 1 plain FMAs in k order
 * (any size, fp64); 2 (fp32, M <= 224) bf16 matrix cores on an exact three-way split of every fp32 operand, six
 * partial products per product, fp32 accumulation -- fp32 input precision at 6/16 of the fp32 MFMA cost.
 */
int ssn_weight_grad_f32(const float *delta, const float *traj, float *gW, int B, long K, int M, int kernel, void *stream);
int ssn_weight_grad_f64(const double *delta, const double *traj, double *gW, int B, long K, int M, int kernel, void *stream);
/*
 * The same product on the fp16 matrix cores: every operand as two fp16 numbers by round to nearest (x 2^e = h + m to 2^-24),
 * three partial products per product, fp32 accumulation -- the accuracy of kernel 2 at half its matrix work.  fp16 has five
 * exponent bits, so the caller supplies bounds: dmax device [B], dmax[b] >= max |delta[b]| (one power-of-two scale per
 * draw; elements below 2^-40 of it are lost), and xmax >= max |traj| (host scalar, > 0 and finite: the rate bound of the
 * saturating I/O function).  A bound that is too small overflows fp16: inf / NaN in gW, never a silently wrong value.
 * ssn_gen_backward_max_f32 is ssn_gen_backward_ext_f32 plus that bound: when the sweep that runs tracks max |delta| per draw
 * (the fp16-split kernels 4 / 5 / 6 / 8, which need it for their own scaling) it fills dmax[B] and sets *tracked = 1; otherwise *tracked = 0,
 * dmax is untouched and ssn_weight_grad_f32 is the kernel to call.  fp32, M <= 224.
 */
/*
 * The adjoint sweep and dL/dW in ONE launch (the two halves of the same `theano.grad`, networks/wgan.py:236-242 through
 * ssn.py:354-385): what ssn_gen_backward_max_f32 followed by ssn_weight_grad_scaled_f32 compute, without the delta
 * stream between them -- a draw's dL/dW accumulates in the register file of the workgroup that runs its sweep.
 * df [B][NB][T][M] (f'(u) of the forward) is only read; gW device [B][M][M]; g_ext [B][NB][M] or NULL; dmax device [B]
 * or NULL (max |delta| per draw; NaN marks a draw whose delta outgrew its lagged fp16 scale, and its gW is NaN too);
 * xmax >= max |traj| as for ssn_weight_grad_scaled_f32.  fp32, NB <= 8, even 2N <= 208: ..._supported answers 1 / 0 for a
 * shape, and the entry refuses the others (no fallback inside).
 */
int ssn_gen_backward_fused_supported(int B, int NB, int M, const ssn_gen_params *p, float xmax);
int ssn_gen_backward_fused_f32(const float *W, const float *traj, const float *df, const float *g_time_avg, float *g_ext,
                               float *gW, float *dmax, float xmax, double c_dyn, double c_rate, int B, int NB, int M,
                               const ssn_gen_params *p, void *stream);
int ssn_gen_backward_max_f32(const float *W, const float *traj, float *df_delta, const float *g_time_avg, float *g_ext,
                             float *dmax, int *tracked, double c_dyn, double c_rate, int B, int NB, int M,
                             const ssn_gen_params *p, void *stream);
int ssn_weight_grad_scaled_f32(const float *delta, const float *traj, float *gW, int B, long K, int M, const float *dmax,
                               float xmax, void *stream);
/*
 * Chain rule W -> (J, D, S) of make_W_with_x (make_w_batch.py:19-34): out[b][pq][0..2] =
 * partial dL/dJ_pq, dL/dD_pq, dL/dS_pq of draw b (device fp64 [B][4][3]; sum over b on the caller's
 * side in a fixed order).  J, D, S are HOST arrays of 4.
 */
int ssn_jds_grad_f32(const float *gW, const float *z, const float *J, const float *D, const float *S,
                     double *out, int B, int N, void *stream);
int ssn_jds_grad_f64(const double *gW, const double *z, const double *J, const double *D, const double *S,
                     double *out, int B, int N, void *stream);

/* ------------------------------------------------------------------------
 * 4. WGAN-GP critic and optimizers: replace the compiled Theano graphs of
 *    ConditionalDiscriminator / ConditionalCriticTrainer (networks/cwgan.py:123-214),
 *    the critic part of GeneratorTrainer's loss (networks/wgan.py:236) and
 *    Updater (networks/wgan.py:111-165 over lasagne.updates.*).
 *    Network: h0 = [x(nx), contrast, |norm_probe|, cell_type]; L hidden ReLU layers
 *    (normalization 'none'); linear 1-unit output without bias.
 *    params / grads: ONE flat fp32 device buffer [W_1 (n0 x n1) row-major, b_1, ...,
 *    W_L, b_L, w_out (n_L)] -- the order of lasagne.layers.get_all_params.
 *    dims: HOST int[L+1] = {n0 = nx+3, n1, ..., nL}.
 *    Unconditional critic (UnConditionalDiscriminator, networks/wgan.py:66-97; CriticTrainer,
 *    wgan.py:194-215): pass NULL for EVERY condition pointer of a call (cond; cg, cd and cp
 *    together) -- then h0 = x and dims[0] = nx.  Conditions for some inputs of a call and not
 *    for others are refused.
 *    precision: 0 = bf16 MFMA operands (fp32 accumulate), 1 = fp32 MFMA.
 * ------------------------------------------------------------------------ */
long   ssn_critic_num_params(const int *dims, int nlayers);
size_t ssn_critic_workspace_floats(const int *dims, int nlayers, int batch_gd, int batch_p);
/* out[batch] = D(x, cond).  x device [batch][nx], cond device [batch][3] (contrast, norm_probe, cell_type). */
int ssn_critic_forward(const float *params, const int *dims, int nlayers, const float *x, const float *cond,
                       int batch, int hide_cell_type, float *out, float *workspace, int precision, void *stream);
/* loss = mean D(xg) - mean D(xd) + lmd * mean((||dD/dxp|| - 1)^2) and d loss / d params.
 * stats device float[4] = {mean D(xg), mean D(xd), penalty, loss}; dvals device [ng+nd] = D of [xg; xd]. */
int ssn_critic_loss_grad(const float *params, const int *dims, int nlayers,
                         const float *xg, const float *cg, const float *xd, const float *cd,
                         const float *xp, const float *cp, int ng, int nd, int np, float lmd,
                         int hide_cell_type, float *grads, float *stats, float *dvals,
                         float *workspace, int precision, void *stream);
/* gx[batch][nx] = scale * d(mean-free sum_b D_b)/dx, stats[0] = mean D(x)  (generator side: scale = -1/batch). */
int ssn_critic_input_grad(const float *params, const int *dims, int nlayers, const float *x, const float *cond,
                          int batch, int hide_cell_type, float scale, float *gx, float *stats,
                          float *workspace, int precision, void *stream);

/* The same three passes with the hidden nonlinearity x > 0 ? x : leak * x, 0 <= leak <= 1: lasagne's `rectify` (0),
 * `leaky_rectify` (0.01), `very_leaky_rectify` (1/3), `linear` / `identity` (1) -- the piecewise-linear choices of
 * simple_discriminator.py:139-152, for which the input gradient stays a linear chain with fixed slopes and the WGAN-GP double
 * backward keeps its form.  Plain (not layer-normalised) layers, layer-by-layer path. */
int ssn_critic_forward_leaky(const float *params, const int *dims, int nlayers, const float *x, const float *cond,
                             int batch, int hide_cell_type, float leak, float *out, float *workspace, int precision,
                             void *stream);
int ssn_critic_loss_grad_leaky(const float *params, const int *dims, int nlayers,
                               const float *xg, const float *cg, const float *xd, const float *cd,
                               const float *xp, const float *cp, int ng, int nd, int np, float lmd,
                               int hide_cell_type, float leak, float *grads, float *stats, float *dvals,
                               float *workspace, int precision, void *stream);
int ssn_critic_input_grad_leaky(const float *params, const int *dims, int nlayers, const float *x, const float *cond,
                                int batch, int hide_cell_type, float leak, float scale, float *gx, float *stats,
                                float *workspace, int precision, void *stream);

/* acc[0] = mean D(xg) - mean D(xd) (the "accuracy" of cwgan.py:139-147, logged after every critic update) in one call: two
 * forwards into dvals (device [ng + nd], scratch), one reduction in a fixed order.  layer_norm: HOST int[L] or NULL; leak as
 * above; workspace as for ssn_critic_forward* with batch = max(ng, nd). */
int ssn_critic_accuracy(const float *params, const int *dims, const int *layer_norm, int nlayers, float leak,
                        const float *xg, const float *cg, const float *xd, const float *cd, int ng, int nd,
                        int hide_cell_type, float *acc, float *dvals, float *workspace, int precision, void *stream);

/* The same three passes for a critic whose hidden layer l is layer-normalised when layer_norm[l] != 0
 * (simple_discriminator.py:51-75: Dense(no bias) -> LayerNorm (eps 1e-4, no parameters) -> Bias -> ReLU;
 * same [W_l, b_l] parameter layout).  layer_norm: HOST int[L] or NULL (all plain).  The WGAN-GP double
 * backward goes through the normalisation. */
size_t ssn_critic_norm_workspace_floats(const int *dims, int nlayers, int batch_gd, int batch_p);
int ssn_critic_forward_norm(const float *params, const int *dims, const int *layer_norm, int nlayers,
                            const float *x, const float *cond, int batch, int hide_cell_type, float *out,
                            float *workspace, int precision, void *stream);
int ssn_critic_loss_grad_norm(const float *params, const int *dims, const int *layer_norm, int nlayers,
                              const float *xg, const float *cg, const float *xd, const float *cd,
                              const float *xp, const float *cp, int ng, int nd, int np, float lmd,
                              int hide_cell_type, float *grads, float *stats, float *dvals,
                              float *workspace, int precision, void *stream);
int ssn_critic_input_grad_norm(const float *params, const int *dims, const int *layer_norm, int nlayers,
                               const float *x, const float *cond, int batch, int hide_cell_type, float scale,
                               float *gx, float *stats, float *workspace, int precision, void *stream);

/* The same passes for the critic in its general form (simple_discriminator.py:57-75, 139-152; `--disc-nonlinearity`,
 * run/bptt_wgan.py:153): hidden nonlinearity act = 0 rectify, 1 leaky_rectify (0.01), 2 very_leaky_rectify (1/3), 3 linear,
 * 4 tanh, 5 sigmoid, 6 softplus, 7 elu (lasagne.nonlinearities); layer_flags: HOST int[L] or NULL, per hidden layer 0 = plain
 * Dense + bias, 1 = Dense(no bias) -> LayerNorm -> Bias, 3 = Dense(no bias) -> LayerNorm -> ScaleLayer -> Bias (the reference
 * inserts the learnable per-unit scale for every nonlinearity but rectify).  Parameter layout per layer: W [in][out],
 * scales [out] (flag 3 only), b [out]; then w_out -- lasagne's get_all_params order.  The gradient penalty's double backward
 * carries the curvature f'' of the smooth nonlinearities and the scales' gradients.  Workspace: ssn_critic_norm_workspace_floats. */
long ssn_critic_num_params_act(const int *dims, const int *layer_flags, int nlayers);
int ssn_critic_forward_act(const float *params, const int *dims, const int *layer_flags, int nlayers, int act,
                           const float *x, const float *cond, int batch, int hide_cell_type, float *out,
                           float *workspace, int precision, void *stream);
int ssn_critic_loss_grad_act(const float *params, const int *dims, const int *layer_flags, int nlayers, int act,
                             const float *xg, const float *cg, const float *xd, const float *cd,
                             const float *xp, const float *cp, int ng, int nd, int np, float lmd,
                             int hide_cell_type, float *grads, float *stats, float *dvals,
                             float *workspace, int precision, void *stream);
int ssn_critic_input_grad_act(const float *params, const int *dims, const int *layer_flags, int nlayers, int act,
                              const float *x, const float *cond, int batch, int hide_cell_type, float scale,
                              float *gx, float *stats, float *workspace, int precision, void *stream);
int ssn_critic_accuracy_act(const float *params, const int *dims, const int *layer_flags, int nlayers, int act,
                            const float *xg, const float *cg, const float *xd, const float *cd, int ng, int nd,
                            int hide_cell_type, float *acc, float *dvals, float *workspace, int precision, void *stream);

typedef struct ssn_opt_params {
    int kind;                 /* 0 sgd, 1 adam, 2 rmsprop (lasagne.updates) */
    int step;                 /* 1-based update count (adam bias correction) */
    int clip;                 /* clamp the new value to [clip_lo, clip_hi] (wgan.py:244-251) */
    int reserved;             /* bit 0 (ssn_gen_apply_f32 with a record, n <= 64): make NO update when any gradient element is not
                               * finite -- parameters and optimizer state stay, record = the old values, record[n] = NaN */
    double learning_rate, beta1, beta2, epsilon, rho;
    double reg_l2_penalty, reg_l1_penalty, reg_l2_decay, reg_l1_decay;   /* wgan.py:121-129 */
    double clip_lo, clip_hi;
} ssn_opt_params;
/* In-place update of p[n] from g[n]; s1, s2 optimizer state (adam: m, v; rmsprop: s1 only; sgd: unused). */
int ssn_optimizer_step(float *p, const float *g, float *s1, float *s2, long n, const ssn_opt_params *o, void *stream);

/*
 * One critic step of the GAN loop in ONE call (single process; the data-parallel loop keeps the separate calls because its
 * all-reduce sits between the gradient and the optimizer step): penalty points xp = eps xd + (1 - eps) xg (cwgan.py:476-481),
 * loss + gradient (ssn_critic_loss_grad*), optimizer step on the critic's parameters, accuracy of the UPDATED critic
 * (cwgan.py:507), per-tensor sums of squares of the updated parameters (recorders.py:275-311), and the step's scalar record
 * tail[4 + nseg] = {dynamics penalty, rate penalty (from pens64, device double[2] or NULL), loss, accuracy, sums of squares}.
 * The same kernels in the same order as the separate calls: identical results.  xg, xd: device [n][nx]; cond [n][3] shared by
 * the three inputs; eps device [n]; layer_norm / leak as above; opt_s1 / opt_s2 / opt as ssn_optimizer_step; seg_bounds, seg_ws
 * as ssn_segment_sqnorms2_f32; acc_dvals device [2 n] scratch; workspace as ssn_critic_*workspace_floats(n + n, n).
 */
typedef struct ssn_critic_step {
    float *params; const int *dims; const int *layer_norm; int nlayers; float leak;
    const float *xg, *xd, *cond, *eps;
    int n, hide_cell_type, precision; float lmd;
    float *xp, *grads, *stats, *dvals, *workspace;
    float *opt_s1, *opt_s2; const ssn_opt_params *opt;
    const long *seg_bounds; int nseg; double *seg_ws;
    const double *pens64;
    float *acc_dvals, *tail;
} ssn_critic_step;
int ssn_critic_step_run(const ssn_critic_step *a, void *stream);
/* The same with the rule of cwgan.py:493-498 decided on the device: the optimizer step (parameters AND optimizer state) is
 * skipped when the batch's rate penalty pens64[1] exceeds rate_penalty_bound (> 0; pens64 required).  Everything else runs and
 * is written as in ssn_critic_step_run -- the caller reads tail[1], sees the bound exceeded and reports the step as skipped
 * (loss and accuracy NaN in the reference's record); no host round trip, no snapshot to roll back to. */
int ssn_critic_step_gated_run(const ssn_critic_step *a, double rate_penalty_bound, void *stream);

/*
 * End of a generator step (GeneratorTrainer, networks/wgan.py:218-260) in two calls, with the job's all-reduce (if any)
 * between them.  ssn_gen_grads_f32, ONE launch: out[nv + 12 + 1] = {dL/dV (nv = 0: none; 1: 'deg-heteroin', one V for both
 * populations; 2: 'heteroin', V_E, V_I), dL/dJ[4], dL/dD[4], dL/dS[4], loss} from jds_part (device [B][4][3], the output of
 * ssn_jds_grad_*), the input-variability pieces g_ext, ext_base (device [B][NB][M]) and zin (device [B][M]) --
 * dL/dV_pop = sum over draws, stimuli and the population's neurons of g_ext ext_base zin (networks/ssn.py:679-686) --
 * and loss = -dmean[0] + dynamics_cost pens64[0] + rate_cost pens64[1] (wgan.py:236-241; dmean: mean D(G(z)), stats[0] of
 * ssn_critic_input_grad*; pens64 may be NULL = 0).  Sums in fp64, fixed order.  ws: device scratch of
 * ssn_gen_grads_ws_doubles() doubles whose LAST 8 bytes are zero before the first call (the kernel leaves them zero).
 * ssn_gen_apply_f32, ONE launch: the update of ssn_optimizer_step on all n parameters at once with per-element clip bounds
 * clip_lo / clip_hi (device [n]; both NULL = opt's scalar rule) -- the reference clips with numpy broadcasting,
 * wgan.py:244-251 -- and record[n + 1] (device, or NULL) = the new values followed by grads[n] (the loss riding behind the
 * gradient vector).
 */
typedef struct ssn_gen_grads {
    const double *jds_part; int B;
    int nv;
    const float *g_ext, *ext_base, *zin; int NB, M;
    const float *dmean; const double *pens64;
    double dynamics_cost, rate_cost;
    double *ws;
    float *out;
} ssn_gen_grads;
long ssn_gen_grads_ws_doubles(void);
int ssn_gen_grads_f32(const ssn_gen_grads *a, void *stream);
int ssn_gen_apply_f32(float *params, const float *grads, float *s1, float *s2, int n, const ssn_opt_params *opt,
                      const float *clip_lo, const float *clip_hi, float *record, void *stream);

/* ------------------------------------------------------------------------
 * 5. Feed-forward tuning-curve generator (FF_lalazar model; BASELINE config 5): replaces the Theano
 *    graph of FF_functions/lalazar_func.py:16-45 (get_FF_output) compiled at
 *    FF_lalazar_model.py:175-179.  Grid = box^3 points of linspace(-3, 3, box)^3, z fastest.
 * ------------------------------------------------------------------------ */
typedef struct ssn_ff_params {
    int nsam, nhid, ni, box;
    double RF_l, RF_d, TH, TH_d, J, a;   /* already exponentiated where the model script does (FF_lalazar_model.py:175) */
} ssn_ff_params;
/* out[nsam][ni][nhid] = relu(sum_g e w / sum_g e - thr).  RF_w [nsam][G], FF_con/FF_str [nsam][nhid][G],
 * TH_sam [nsam][nhid], stim [ni][3] (ni <= 32), all device fp32.  q, den [nsam][ni][nhid] optional (NULL):
 * pre-threshold drive and sum_g e, kept for the backward. */
int ssn_ff_forward_f32(const float *RF_w, const float *FF_con, const float *FF_str, const float *TH_sam,
                       const float *stim, float *out, float *q, float *den, const ssn_ff_params *p, void *stream);
/* The same outputs from the model's own data structure: a unit has box^3 / 100 connections (FF_lalazar_model.py:154-167), so
 * instead of the dense FF_con / FF_str streams (99 % zeros, 8 of the 12 bytes per point) the connections come as lists:
 * conn_idx[nsam][nhid][ncon] (device int32, grid index z fastest, each index at most once per unit; < 0 = empty slot) and
 * conn_str[nsam][nhid][ncon] (the strengths FF_str at those indices).  sum_g e runs over all points of RF_w (4 B per point,
 * shared by the sample's hidden units), sum_g e w over the list. */
int ssn_ff_forward_sparse_f32(const float *RF_w, const int *conn_idx, const float *conn_str, int ncon, const float *TH_sam,
                              const float *stim, float *out, float *q, float *den, const ssn_ff_params *p, void *stream);
/* ssn_ff_backward_f32 (below) from the same lists: dsig as there. */
int ssn_ff_backward_sparse_f32(const float *RF_w, const int *conn_idx, const float *conn_str, int ncon, const float *stim,
                               const float *q, const float *den, const float *gq, float *dsig, const ssn_ff_params *p,
                               void *stream);
/* dsig[nsam][nhid][2] = per-sample partial derivatives of L w.r.t. RF_l and RF_d, given gq[nsam][ni][nhid] =
 * dL/d(drive) (upstream gradient times [out > 0]) and the forward's q, den. */
int ssn_ff_backward_f32(const float *RF_w, const float *FF_con, const float *FF_str, const float *stim,
                        const float *q, const float *den, const float *gq, float *dsig,
                        const ssn_ff_params *p, void *stream);

/* ---- 6. Moment matching (tc_gan/networks/moment_matching.py:91-104, 221-243; run/bptt_moments.py) ------------
 * x[B][D] generated tuning curves (device fp32).  sums[2][D] (device fp64) = (sum_b x, sum_b x^2): all-reduce it
 * over ranks when the minibatch is sharded, then pass the GLOBAL batch size.  data_moments, weights: [2][D] device
 * fp64 (BPTTMomentMatcher.set_dataset).  gx[B][D] = d L0 / d x with L0 = mean(weights * (data_moments - (m, s))^2);
 * out[1 + 2 D] (device fp64) = L0, m[D], s[D]  (sample mean / population variance of the minibatch). */
int ssn_moment_sums_f32(const float *x, int B, int D, double *sums, void *stream);
int ssn_moment_loss_grad_f32(const float *x, const double *sums, double global_batch, const double *data_moments,
                             const double *weights, int B, int D, float *gx, double *out, void *stream);

/* ---- 7. Fixed-point implicit gradient (tc_gan/gradient_expressions/SS_grad.py:17-99, make_w_batch.py:36-121) -----
 * dW[b][i][j][p][q] = d W[b][i][j] / d theta[p][q] for theta = J (which 0; independent of z, z may be NULL),
 * D (1) or S (2): the tensors make_WJ_with_x / make_WD_with_x / make_WS_with_x build (identity dJ'/dJ). */
int ssn_build_dw_f32(const float *z, const float *J, const float *D, const float *S, int which, float *dW, int B, int N,
                     void *stream);
int ssn_build_dw_f64(const double *z, const double *J, const double *D, const double *S, int which, double *dW, int B,
                     int N, void *stream);
/* The batched linear systems of WRgrad_batch at fixed points R[nz][nb][M]:  A[z][b] = 1 - Phi W[z],
 * rhs[z][b][i][c] = Phi_i * sum_j dW[z][i][j][c] R[z][b][j],  Phi = f'(W R + I) (I [nb][M], or [nz][nb][M] when
 * i_per_draw; dW [nz][M][M][4], or [1][M][M][4] when !dw_per_draw).  Solving A x = rhs gives dR/dtheta[z][b][M][2][2].
 * p: io_type, k, n, rate_soft_bound, rate_hard_bound are used.  All device pointers. */
int ssn_ss_grad_system_f32(const float *R, const float *W, const float *dW, int dw_per_draw, const float *I,
                           int i_per_draw, int nz, int nb, int M, const ssn_solver_params *p, float *A, float *rhs,
                           void *stream);
int ssn_ss_grad_system_f64(const double *R, const double *W, const double *dW, int dw_per_draw, const double *I,
                           int i_per_draw, int nz, int nb, int M, const ssn_solver_params *p, double *A, double *rhs,
                           void *stream);

/* Heterogeneous-input variant (networks/ssn.py:645-772): ext *= amp[b][m], amp = 1 + v_pop * z_in,
 * device [B][2N] (NULL = homogeneous). */
int ssn_stimulus_amp_f32(const float *bandwidths, const float *contrasts, float smoothness, const float *amp,
                         float *ext, int B, int NB, int N, void *stream);
int ssn_stimulus_amp_f64(const double *bandwidths, const double *contrasts, double smoothness, const double *amp,
                         double *ext, int B, int NB, int N, void *stream);
/* The same with the amplification formed in the launch (networks/ssn.py:679-686): amp[b][m] = 1 + v(m) * zin[b][m], rounded as
 * the two fp32 operations of `1 + vs[None, :] * zin`.  zin: device [B][2N]; v: device, one value per neuron (nv = 2N), per
 * population (nv = 2: neurons 0..N-1 take v[0], the others v[1]) or one for all (nv = 1). */
int ssn_stimulus_hetero_f32(const float *bandwidths, const float *contrasts, float smoothness, const float *zin, const float *v,
                            int nv, float *ext, int B, int NB, int N, void *stream);

/* I/O nonlinearity on arrays (device pointers), the device function the solver
 * kernels use: out[i] = io(v[i]).  p->k, n, rate_soft_bound, rate_hard_bound,
 * io_type are read; for SSN_IO_* semantics see ssnode.c:25-53. */
int ssn_io_eval_f32(const float *v, float *out, long count, const ssn_solver_params *p, void *stream);
int ssn_io_eval_f64(const double *v, double *out, long count, const ssn_solver_params *p, void *stream);

/* Batched dense solve for the implicit gradient (the `solve` of tc_gan/gradient_expressions/SS_grad.py:44, once per
 * (draw, stimulus)): A [nsys][M][M] row-major and rhs [nsys][M][nrhs] (nrhs = 4 or 1), device pointers, both
 * OVERWRITTEN -- A with its LU factors (partial row pivoting), rhs with the solution.  info (device, [nsys], may be
 * NULL): 0, or 1 + the first elimination step whose pivot column was all zero. */
int ssn_lu_solve_f32(float *A, float *rhs, int *info, int nsys, int M, int nrhs, void *stream);
int ssn_lu_solve_f64(double *A, double *rhs, int *info, int nsys, int M, int nrhs, void *stream);

/* The two scalar penalties of the fixed-time generator from the per-neuron window sums ssn_gen_forward_* leaves
 * (networks/ssn.py:626, 632): out[0] = scale_dyn * sum(dyn_row), out[1] = scale_rate * sum(rate_row), fp64, one launch,
 * deterministic.  ws: device scratch of 2 * 256 + 1 doubles whose LAST 8 bytes are zero before the first call (the
 * kernel leaves them zero); out: device [2]. */
int ssn_penalty_means_f32(const float *dyn_row, const float *rate_row, long n, double scale_dyn, double scale_rate,
                          double *ws, double *out, void *stream);
int ssn_penalty_means_f64(const double *dyn_row, const double *rate_row, long n, double scale_dyn, double scale_rate,
                          double *ws, double *out, void *stream);
/* The same launch with the conditional prober's gather riding along (cwgan.py:91-98; one launch instead of two):
 * tc[k][s] = time_avg[ids[k]][s][probes[k]] for k < nsamp, s < NB (time_avg: device [B][NB][M]; ids, probes: device int64
 * [nsamp], 0 <= ids < B, 0 <= probes < M; tc: device [nsamp][NB]).  nsamp = 0: exactly ssn_penalty_means_*. */
int ssn_penalty_means_probe_f32(const float *dyn_row, const float *rate_row, long n, double scale_dyn, double scale_rate,
                                double *ws, double *out, const float *time_avg, const long *ids, const long *probes,
                                float *tc, int nsamp, int NB, int M, void *stream);
int ssn_penalty_means_probe_f64(const double *dyn_row, const double *rate_row, long n, double scale_dyn, double scale_rate,
                                double *ws, double *out, const double *time_avg, const long *ids, const long *probes,
                                double *tc, int nsamp, int NB, int M, void *stream);
/* Per-step helpers of the GAN loop.  ssn_segment_sqnorms2_f32: out[t] = sum of x[i]^2 over
 * bounds[t] <= i < bounds[t + 1] (the per-tensor critic statistics of recorders.py:275-311; bounds: device [n + 1], out:
 * device [n]; ws: device scratch of ssn_segment_sqnorms_ws_doubles(n) doubles; every tensor is cut into chunks summed by
 * their own workgroups in fp64, the chunk sums added in chunk order: the same bits every run).
 * ssn_segment_sqnorms_f32 keeps the signature this symbol had before the scratch argument existed (same results; the
 * scratch comes from the stream-ordered allocator): a caller built against the older header still links AND runs right.
 * ssn_interpolate_f32: xp[r][c] = eps[r] xd[r][c] + (1 - eps[r]) xg[r][c], the
 * gradient-penalty points of cwgan.py:476-481 (all device; xd, xg, xp [rows][cols], eps [rows]). */
long ssn_segment_sqnorms_ws_doubles(int n);
int ssn_segment_sqnorms2_f32(const float *x, const long *bounds, int n, float *out, double *ws, void *stream);
int ssn_segment_sqnorms_f32(const float *x, const long *bounds, int n, float *out, void *stream);
/* Adjoint of the conditional prober's gather tuning_curve[k][s] = time_avg[ids[k]][s][probes[k]] (cwgan.py:91-98):
 * g_ta[b][s][m] = sum of g[k][s] over the samples k with ids[k] == b, probes[k] == m; the whole of g_ta is written.
 * g: device [n][NB], ids / probes: device int64 [n] (0 <= ids < B, 0 <= probes < M), g_ta: device [B][NB][M].
 * Samples are added in k order: deterministic. */
int ssn_probe_scatter_f32(const float *g, const long *ids, const long *probes, float *g_ta, int n, int B, int NB, int M, void *stream);
int ssn_probe_scatter_f64(const double *g, const long *ids, const long *probes, double *g_ta, int n, int B, int NB, int M, void *stream);
int ssn_interpolate_f32(const float *eps, const float *xd, const float *xg, float *xp, int rows, int cols, void *stream);

/* Device-side noise for the generator's z (replaces the host `rng.rand(batch, 2N, 2N)` of
 * tc_gan/networks/ssn.py:434-439 in the opt-in performance mode; the reference has no such mode, parity runs keep
 * host noise).  out[i] = u(seed, offset + i) in [0, 1) with 24 random bits, from Philox4x32-10 (Salmon, Moraes,
 * Dror, Shaw, SC'11; known-answer vectors checked in tests/test_noise.py): element g of the stream is word g % 4 of
 * the block with counter (g / 4, 0, 0, 0) and key (seed lo, seed hi).  Any slice can be generated by itself, so the
 * ranks of a data-parallel job each fill their own rows of ONE global stream.  `out` is a device pointer. */
int ssn_philox_uniform_f32(unsigned long long seed, unsigned long long offset, float *out, unsigned long long n, void *stream);
int ssn_philox_uniform_f64(unsigned long long seed, unsigned long long offset, double *out, unsigned long long n, void *stream);
/*
 * ssn_philox_uniform_* followed by ssn_build_w_* in one launch: W[b] = make_W_with_x(z[b]; J, D, S) with z = the B*M*M
 * stream elements from `offset` on (bit for bit the numbers ssn_philox_uniform_* writes).  z: device [B][M][M] to keep the
 * draw (the generator update's chain rule reads it), or NULL -- then the noise never touches memory.  W (and z) must be
 * 16-byte aligned; otherwise invalid-argument, and the two calls above are the path.
 */
int ssn_build_w_philox_f32(unsigned long long seed, unsigned long long offset, const float *J, const float *D, const float *S,
                           float *W, float *z, int B, int N, void *stream);
int ssn_build_w_philox_f64(unsigned long long seed, unsigned long long offset, const double *J, const double *D, const double *S,
                           double *W, double *z, int B, int N, void *stream);
/*
 * Everything a device-noise generator forward reads, in ONE call (the launches of ssn_philox_amp_f32, ssn_stimulus_amp_f32 and
 * ssn_build_w_philox_f32 in that order: the same numbers, without the host between them): zin / amp = the heterogeneous-input
 * signs and 1 + v zin from stream elements off_zin.. (v device [M]; v NULL: none, plain stimulus), ext[B][NB][M] the (amplified)
 * stimulus of bandwidths bw and contrasts con (device [B][NB]), W[B][M][M] from stream elements off_z.. (z device or NULL).
 */
typedef struct ssn_gen_inputs {
    unsigned long long seed, off_z, off_zin;
    const float *J, *D, *S;            /* HOST float[4] each */
    const float *bw, *con;
    float smoothness;
    const float *v;
    int bernoulli;
    float *W, *z, *zin, *amp, *ext;
    int B, NB, N;
} ssn_gen_inputs;
int ssn_gen_inputs_philox_f32(const ssn_gen_inputs *a, void *stream);
/* The heterogeneous-input SSN's noise from the same stream, in one launch (networks/ssn.py:679-720): zin[i] = +1 / -1
 * (u < 0.5; bernoulli != 0) or 2 u - 1, and amp[i] = 1 + v[i % M] * zin[i] (v: device [M], the input variability per
 * neuron); zin, amp: device [n]. */
int ssn_philox_amp_f32(unsigned long long seed, unsigned long long offset, const float *v, float *zin, float *amp,
                       unsigned long long n, int M, int bernoulli, void *stream);
int ssn_philox_amp_f64(unsigned long long seed, unsigned long long offset, const double *v, double *zin, double *amp,
                       unsigned long long n, int M, int bernoulli, void *stream);

/*
 * The reference's OWN noise stream on the device: numpy's RandomState.random_sample (= rng.rand), bit for bit.
 * tc_gan/networks/ssn.py:434-439 draws `zs = rng.rand(batchsize, 2N, 2N)` from the RandomState the GAN shares with its
 * minibatch sampler (stream order networks/cwgan.py:438-481) and hands it to Theano in floatX (utils/theanoutils.py:9-16:
 * fp64 -> fp32 by round to nearest even).  numpy (pinned 1.13.1, requirements-conda.txt:46) implements RandomState as
 * randomkit's MT19937; this call continues that generator from the caller's state:
 *   key[624], *pos   numpy's `RandomState.get_state()[1:3]` (HOST, in/out; 0 <= pos <= 624).  On return they are the state
 *                    numpy itself would have after `random_sample(total)`: pass them to `set_state` and the next host draw
 *                    (`choice`, `rand`) continues exactly as in the reference.  The call returns once that state is known; it
 *                    waits for a short chain on a stream of the library's own, never for `stream`.
 *   total            doubles the draw consumes (2 words each: (a >> 5, b >> 6) -> (a 2^26 + b) / 2^53)
 *   skip, count      doubles [skip, skip + count) of the draw are written to out[0 .. count) (device) on `stream`; the rest is
 *                    not generated -- a rank of a data-parallel job passes its own rows of the global draw.  count = 0 only
 *                    advances the state.
 * _f32 rounds each double to the nearest float (ties to even), as numpy's astype(float32) does.  Any slice costs jump-ahead
 * by precomputed polynomials (Haramoto et al. 2008; ssn_mt19937_poly.h), built at the first call (tens of ms per level).
 * ssn_mt19937_jump_poly is the host-side arithmetic alone, for tests: bits[313] = t^(624 nblocks) mod the characteristic
 * polynomial; needs no device.
 */
int ssn_mt19937_random_sample_f32(unsigned int *key, int *pos, unsigned long long total, unsigned long long skip,
                                  unsigned long long count, float *out, void *stream);
int ssn_mt19937_random_sample_f64(unsigned int *key, int *pos, unsigned long long total, unsigned long long skip,
                                  unsigned long long count, double *out, void *stream);
int ssn_mt19937_jump_poly(unsigned long long nblocks, unsigned long long *bits);
/* The same draw in two halves, for a caller that has launches to queue before it next needs its generator: _begin launches
 * everything (key / pos are read, not written) and returns a ticket at once; _finish(ticket) waits for the state after the
 * draw -- one small launch on the library's own stream -- and writes it to key / pos.  A ticket is finished at most once
 * (a second time: invalid argument) and may be finished any time later, after any number of other draws.  ssn_mt19937_random_sample_* = _begin followed by _finish. */
int ssn_mt19937_random_sample_begin_f32(const unsigned int *key, int pos, unsigned long long total, unsigned long long skip,
                                        unsigned long long count, float *out, void *stream, int *ticket);
int ssn_mt19937_random_sample_begin_f64(const unsigned int *key, int pos, unsigned long long total, unsigned long long skip,
                                        unsigned long long count, double *out, void *stream, int *ticket);
int ssn_mt19937_random_sample_finish(int ticket, unsigned int *key, int *pos);
/* ssn_mt19937_random_sample_begin_f32 with W = make_W_with_x(z; J, D, S) (ssn_build_w_f32's arithmetic, the same bits) formed in
 * the generation kernel itself: the draw is z[B_total][2N][2N] of the caller's stream, models b0 .. b0 + nb - 1 of it are this
 * call's (a rank's rows), W: device [nb][2N][2N]; z: device [nb][2N][2N] to keep the numbers (the generator update's chain rule
 * reads them) or NULL -- then they never touch memory.  J, D, S: HOST float[4].  Finish with ssn_mt19937_random_sample_finish. */
int ssn_build_w_mt19937_begin_f32(const unsigned int *key, int pos, int B_total, int b0, int nb, const float *J, const float *D,
                                  const float *S, float *W, float *z, int N, void *stream, int *ticket);
/* The two draws above followed IN THE SAME CALL by what the reference's heterogeneous-input models draw right behind zs
 * (tc_gan/networks/ssn.py:710-720, 764-767: `zs_in`), so that the host does not have to fetch the state between the two:
 *   tail_kind 1      rng.choice(2, n) * 2 - 1: one 32-bit output per element, its low bit (numpy's masked rejection with
 *                    mask 1 rejects nothing: `_rand_int64` / `random_bounded_uint64_fill` with rng = 1), as -1.f / 1.f
 *   tail_kind 2      rng.rand(n) * 2 - 1: one double per element, 2 x - 1 exact in double, then rounded to the nearest float
 *   tail_total       elements the tail consumes; [tail_skip, tail_skip + tail_count) of them go to tail_out (device, fp32)
 * The ticket's state is the one behind the tail.  ssn_build_w_mt19937_tail_begin_f32: the tail is zs_in[B_total][2N], rows
 * b0 .. b0 + nb - 1 go to zin: device [nb][2N].  Windows that lie far apart in the stream (a rank's rows) get a launch each. */
int ssn_mt19937_random_sample_tail_begin_f32(const unsigned int *key, int pos, unsigned long long total, unsigned long long skip,
                                             unsigned long long count, float *out, int tail_kind, unsigned long long tail_total,
                                             unsigned long long tail_skip, unsigned long long tail_count, float *tail_out,
                                             void *stream, int *ticket);
int ssn_build_w_mt19937_tail_begin_f32(const unsigned int *key, int pos, int B_total, int b0, int nb, const float *J, const float *D,
                                       const float *S, float *W, float *z, int N, int tail_kind, float *zin, void *stream, int *ticket);
/* Host arithmetic of a draw alone (no device): out[7] = {pos after the draw, regenerations of the key up to then, blocks per
 * segment, first and last segment this call generates, first and last block that holds a wanted word} for the given pos /
 * total / skip / count -- what the CPU tests check against numpy's own positions, rank by rank. */
int ssn_mt19937_plan(int pos, unsigned long long total, unsigned long long skip, unsigned long long count, long *out);
/* ... of a draw with a tail (ssn_mt19937_random_sample_tail_begin_f32): the position and the blocks cover the tail's words too */
int ssn_mt19937_plan_tail(int pos, unsigned long long total, unsigned long long skip, unsigned long long count, int tail_kind,
                          unsigned long long tail_total, unsigned long long tail_skip, unsigned long long tail_count, long *out);

#ifdef __cplusplus
}
#endif
#endif /* SSNODE_MI355X_H */
