/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the
 * product path (tc_gan_amd/).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * CPU restatement (fp64, plain C99) of the reference's forward-Euler SSN
 * fixed-point solver and its I/O nonlinearities:
 *     /root/reference/tc_gan/ext/ssnode.c
 *        dot                          :10-19
 *        rate_to_volt                 :21-23
 *        io_pow / io_alin / io_atanh  :25-53
 *        ODE_STEP                     :64-67
 *        solve_dynamics_asym_{power,linear,tanh}_euler :69-109, 111-151, 153-187
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file bit-for-bit
 * against (a) oracle/_ref/libssnode.so, the reference C file compiled
 * unmodified by oracle/Makefile, on seeded inputs, and (b) the committed
 * fixtures in tests/golden/ that were produced by that reference build
 * (tests/golden/make_golden.py), including the reference's own MATLAB-derived
 * .mat known-answer data.
 *
 * Semantics restated (one routine, io selected by code, instead of the
 * reference's three macro-expanded copies):
 *   state update    r1[i] = r0[i] + (-r0[i] + io(W[i,:].r0 + ext[i])) * (dt/tau_i)
 *                   tau_i = tau_E for i < N, tau_I otherwise
 *   v0              = (rate_soft_bound / k)^(1/n)
 *   after each step: if every |r1[i]-r0[i]| < atol -> newest state is copied
 *                   into the buffer that currently plays "r0", return 0;
 *                   else (power and linear only) if any r1[i] >= rate_hard_bound
 *                   -> return 2 (no copy, no swap);
 *                   else the two buffers exchange roles.
 *   after max_iter steps -> return 1.
 * The role exchange is on the local pointers only, so on return codes 1 and 2
 * which caller buffer holds the newest state depends on the parity of the
 * number of completed exchanges (SURVEY.md section 8(a) a1 "Quirk").
 */
#include <math.h>
#include <stddef.h>

enum { ORACLE_IO_POWER = 0, ORACLE_IO_LINEAR = 1, ORACLE_IO_TANH = 2 };

double oracle_rate_to_volt(double rate, double k, double n)
{
    return pow(rate / k, 1.0 / n);
}

/* io(v): rate as a function of "voltage" v.  soft = rate_soft_bound,
 * hard = rate_hard_bound, v0 = oracle_rate_to_volt(soft, k, n). */
double oracle_io(int io_type, double v, double soft, double hard,
                 double v0, double k, double n)
{
    if (v <= 0.0)
        return 0.0;
    if (io_type == ORACLE_IO_POWER || v <= v0)
        return k * pow(v, n);
    if (io_type == ORACLE_IO_LINEAR)
        return soft + k * pow(v0, n - 1.0) * n * (v - v0);
    /* ORACLE_IO_TANH */
    return soft + (hard - soft) * tanh(n * soft / (hard - soft) * (v - v0) / v0);
}

/* Sequential left-to-right accumulation, as the reference's `dot`.  (The
 * reference marks the loop `omp simd reduction`, which lets the compiler
 * re-associate; tests compare against the reference build with a tolerance of
 * a few ulp on the dot product and bit-exactly everywhere else when both are
 * compiled with the same flags.) */
static double row_dot(int dim, const double *a, const double *b)
{
    double s = 0.0;
#pragma omp simd reduction(+:s)
    for (int j = 0; j < dim; ++j)
        s += a[j] * b[j];
    return s;
}

double oracle_dot(int dim, const double *a, const double *b)
{
    return row_dot(dim, a, b);
}

/*
 * One solve.  Same argument meaning as the reference's solver symbols
 * (ssnode.c:55-62).  `steps_done`, if not NULL, receives the number of Euler
 * steps executed (the step that triggered the return included).
 */
int oracle_solve_euler(int io_type, int N, const double *W, const double *ext,
                       double k, double n, double *r0, double *r1,
                       double tau_E, double tau_I, double dt, int max_iter,
                       double atol, double rate_soft_bound,
                       double rate_hard_bound, int *steps_done)
{
    const int M = 2 * N;
    const double eps_E = dt / tau_E;
    const double eps_I = dt / tau_I;
    const double v0 = oracle_rate_to_volt(rate_soft_bound, k, n);
    double *cur = r0, *nxt = r1;
    int step;

    for (step = 0; step < max_iter; ++step) {
        for (int i = 0; i < M; ++i) {
            const double v = row_dot(M, W + (size_t)M * i, cur) + ext[i];
            const double f = oracle_io(io_type, v, rate_soft_bound,
                                       rate_hard_bound, v0, k, n);
            nxt[i] = cur[i] + (-cur[i] + f) * (i < N ? eps_E : eps_I);
        }

        int settled = 1;
        for (int i = 0; i < M; ++i) {
            if (fabs(nxt[i] - cur[i]) >= atol) {
                settled = 0;
                break;
            }
        }
        if (settled) {
            for (int i = 0; i < M; ++i)
                cur[i] = nxt[i];
            if (steps_done) *steps_done = step + 1;
            return 0;
        }

        if (io_type != ORACLE_IO_TANH) {
            for (int i = 0; i < M; ++i) {
                if (nxt[i] >= rate_hard_bound) {
                    if (steps_done) *steps_done = step + 1;
                    return 2;
                }
            }
        }

        double *tmp = cur; cur = nxt; nxt = tmp;
    }
    if (steps_done) *steps_done = max_iter;
    return 1;
}

/*
 * Batch driver used by tests and by bench.py's cpu_baseline ("port" kind):
 * B independent weight matrices x NB stimuli each.  W: [B][M][M], ext:
 * [NB][M] (shared by every draw), r: [B][NB][M] in = initial state, out =
 * contents of the "r0" caller buffer exactly as a reference call would leave
 * it.  codes/steps: [B][NB].  Parallel over draws when built with -fopenmp.
 * scratch: [B][NB][M] doubles (the "r1" buffers).
 */
void oracle_solve_batch(int io_type, int B, int NB, int N, const double *W,
                        const double *ext, double k, double n, double *r,
                        double *scratch, double tau_E, double tau_I, double dt,
                        int max_iter, double atol, double rate_soft_bound,
                        double rate_hard_bound, int *codes, int *steps)
{
    const int M = 2 * N;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        for (int s = 0; s < NB; ++s) {
            const size_t o = ((size_t)b * NB + s);
            codes[o] = oracle_solve_euler(
                io_type, N, W + (size_t)b * M * M, ext + (size_t)s * M, k, n,
                r + o * M, scratch + o * M, tau_E, tau_I, dt, max_iter, atol,
                rate_soft_bound, rate_hard_bound, steps ? steps + o : NULL);
        }
    }
}
