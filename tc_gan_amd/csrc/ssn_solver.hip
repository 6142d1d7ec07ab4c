// Batched SSN forward-Euler fixed-point solver for MI355X (gfx950).
//
// Replaces the hot loop of tc_gan/ext/ssnode.c:69-187 (one call per
// (weight draw, stimulus) pair from a Python thread pool, ssnode.py:423-510) by
// one launch over B draws x NB stimuli.
//
// Two kernels:
//
//  * solve_regw_kernel ("register-stationary", variant 1): one workgroup per
//    weight draw; thread i owns ROW i of W in VGPRs for the whole solve (W is read
//    from HBM exactly once), the NB state vectors live in LDS (double buffered), and
//    the per-step mat-vec  u_i = sum_j W_ij r_j  is a chain of v_fmac_f32 whose r_j
//    operand is a DPP row_newbcast of a register holding 16 consecutive r values --
//    no cross-lane reduction, no partial sums, ONE workgroup barrier per Euler step.
//    Templated on the number of 16-column chunks (KCH) and stimuli per group (NB).
//
//  * solve_stream_kernel ("generic", variant 0): any M; W rows are streamed from
//    global memory (L2 resident) every step, lanes stride the columns and a
//    wavefront shuffle reduction finishes each row.  Correctness baseline and
//    fallback for sizes the register kernel has no instantiation for.
//
// Semantics (both): identical to ssnode.c -- per step r1 = r0 + (-r0 + io(W r0 +
// ext)) * dt/tau; stop with code 0 at the first step where every |r1-r0| < atol,
// else (power/linear) code 2 when any r1 >= rate_hard_bound, else code 1 after
// max_iter steps.  Each (draw, stimulus) pair stops on its own step.
#include <hip/hip_runtime.h>
#include "ssn_device.h"
#include "ssn_host.h"

namespace ssn {


// ---------------------------------------------------------------------------------
// Generic streaming kernel: one workgroup (4 waves) per (draw, stimulus).
// ---------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

template <typename T>
__global__ void __launch_bounds__(256) solve_stream_kernel(SolveArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* rbuf = reinterpret_cast<T*>(smem_raw);                       // [2][M]
    int* flags = reinterpret_cast<int*>(rbuf + 2 * a.M);            // [3][2]
    const int M = a.M, N = a.N;
    const int b = blockIdx.x / a.NB, s = blockIdx.x % a.NB;
    const size_t vec = ((size_t)b * a.NB + s) * M;
    const T* W = a.W + (size_t)b * M * M;
    const T* ext = a.ext + (a.ext_per_draw ? vec : (size_t)s * M);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;

    for (int j = threadIdx.x; j < M; j += blockDim.x) { rbuf[j] = a.r[vec + j]; rbuf[M + j] = a.r[vec + j]; }
    if (threadIdx.x < 6) flags[threadIdx.x] = 0;
    __syncthreads();

    int cur = 0, code = 1, nsteps = a.st.max_iter;
    for (int step = 0; step < a.st.max_iter; ++step) {
        const T* rc = rbuf + cur * M;
        T* rn = rbuf + (cur ^ 1) * M;
        int nc = 0, hb = 0;
        for (int i = wave; i < M; i += nwaves) {
            const T* wrow = W + (size_t)i * M;
            T part = (T)0;
            for (int j = lane; j < M; j += 64) part = fma(wrow[j], rc[j], part);
            const T u = wave_sum(part) + ext[i];
            const T r0 = rc[i];
            const T r1 = r0 + (-r0 + io_eval(u, a.io)) * (i < N ? a.st.eps_E : a.st.eps_I);
            if (lane == 0) rn[i] = r1;
            nc |= (abs_t(r1 - r0) >= a.st.atol);
            hb |= (a.st.check_hard && r1 >= a.st.hard_stop);
        }
        const int slot = step % 3;
        if (lane == 0) {
            if (nc) flags[slot * 2 + 0] = 1;
            if (hb) flags[slot * 2 + 1] = 1;
        }
        if (threadIdx.x == 0) { const int nslot = (step + 1) % 3; flags[nslot * 2] = 0; flags[nslot * 2 + 1] = 0; }
        __syncthreads();
        cur ^= 1;
        const int any_nc = flags[slot * 2], any_hb = flags[slot * 2 + 1];
        if (!any_nc) { code = 0; nsteps = step + 1; break; }
        if (any_hb) { code = 2; nsteps = step + 1; break; }
    }
    // newest state is rbuf[cur]; the state one step earlier is rbuf[cur^1]
    for (int j = threadIdx.x; j < M; j += blockDim.x) {
        a.r[vec + j] = rbuf[cur * M + j];
        if (a.r_prev) a.r_prev[vec + j] = rbuf[(cur ^ 1) * M + j];
    }
    if (threadIdx.x == 0) {
        a.codes[(size_t)b * a.NB + s] = code;
        if (a.steps) a.steps[(size_t)b * a.NB + s] = nsteps;
    }
}

// ---------------------------------------------------------------------------------
// Register-stationary kernel.
// ---------------------------------------------------------------------------------

// acc += sum_{n<16} bcast16(rv, n) * w[BASE+n].  fp32: one v_fmac_f32_dpp per term (the
// compiler does not fold update_dpp into the fmac, so the sequence is spelled out).
template <int BASE, int NW>
__device__ __forceinline__ void chunk_fma(float& acc, float rv, const float (&w)[NW]) {
    // not `volatile`: the statement is a pure function of its operands, so the scheduler may hoist
    // the LDS reads of later chunks above it (with `volatile` every chunk waited for its own read).
    // (s_nop 1: a DPP read needs two wait states after a VALU write of its source and the compiler cannot see into
    // the asm; rv normally comes straight from an LDS read, but a copy through v_mov is the compiler's choice)
    asm(
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(rv), "v"(w[BASE + 0]), "v"(w[BASE + 1]), "v"(w[BASE + 2]), "v"(w[BASE + 3]), "v"(w[BASE + 4]),
          "v"(w[BASE + 5]), "v"(w[BASE + 6]), "v"(w[BASE + 7]), "v"(w[BASE + 8]), "v"(w[BASE + 9]),
          "v"(w[BASE + 10]), "v"(w[BASE + 11]), "v"(w[BASE + 12]), "v"(w[BASE + 13]), "v"(w[BASE + 14]),
          "v"(w[BASE + 15]));
}
template <int BASE, int NW>
__device__ __forceinline__ void chunk_fma(double& acc, double rv, const double (&w)[NW]) {
#define SSN_T(n) acc = fma(w[BASE + n], row_bcast<n>(rv), acc);
    SSN_T(0) SSN_T(1) SSN_T(2) SSN_T(3) SSN_T(4) SSN_T(5) SSN_T(6) SSN_T(7)
    SSN_T(8) SSN_T(9) SSN_T(10) SSN_T(11) SSN_T(12) SSN_T(13) SSN_T(14) SSN_T(15)
#undef SSN_T
}

template <int NB> struct NAcc { static constexpr int value = (NB == 1) ? 4 : (NB == 2 ? 2 : 1); };

// Static recursion over the KCH column chunks: chunk K reads its 16 r values per
// stimulus from LDS (lane & 15 selects the value; the 4 DPP rows read the same 16
// addresses -> broadcast, conflict free) and issues 16 DPP FMAs per stimulus.
template <int K, int KCH, int NB, typename T>
struct ChunkLoop {
    static __device__ __forceinline__ void run(T (&acc)[NB][NAcc<NB>::value], const T (&w)[16 * KCH],
                                               const T* rlane /* &rbuf[cur][0][lane&15] */) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const T rv = rlane[s * (16 * KCH) + 16 * K];
            chunk_fma<16 * K, 16 * KCH>(acc[s][K % NAcc<NB>::value], rv, w);
        }
        ChunkLoop<K + 1, KCH, NB, T>::run(acc, w, rlane);
    }
};
template <int KCH, int NB, typename T>
struct ChunkLoop<KCH, KCH, NB, T> {
    static __device__ __forceinline__ void run(T (&)[NB][NAcc<NB>::value], const T (&)[16 * KCH], const T*) {}
};

// One workgroup = one weight draw x one group of NB stimuli; blockDim = 64*ceil(M/64).
// grid = B * ceil(NBtot/NB).  LDS: rbuf[2][NB][16*KCH] + flags[3][2][NB].
template <typename T, int KCH, int NB, int MAXTHREADS, int MINWAVES>
__global__ void __launch_bounds__(MAXTHREADS, MINWAVES) solve_regw_kernel(SolveArgs<T> a) {
    constexpr int MP = 16 * KCH;
    constexpr int NACC = NAcc<NB>::value;
    __shared__ __align__(16) T rbuf[2][NB][MP];
    __shared__ int flags[3][2][NB];

    const int M = a.M, N = a.N;
    const int ngroups = (a.NB + NB - 1) / NB;
    const int b = blockIdx.x / ngroups;
    const int s0 = (blockIdx.x % ngroups) * NB;     // first stimulus of this group
    const int i = threadIdx.x;                       // my row
    const bool row_ok = i < M;
    const int lane16 = threadIdx.x & 15;

    // ---- prologue: my row of W -> registers (zero padded), state -> LDS -----------
    T w[MP];
    {
        // Unconditional loads from clamped (always valid) addresses, then masked by multiplication:
        // a select would let the compiler put every load under an exec branch with its own
        // s_waitcnt vmcnt(0), serialising 16*KCH HBM round trips per workgroup.
        const T* wrow = a.W + ((size_t)b * M + (row_ok ? i : 0)) * M;
        const T rowmask = row_ok ? (T)1 : (T)0;
        if ((M & 3) == 0) {
            using V4 = T __attribute__((ext_vector_type(4)));
            const V4* wrow4 = reinterpret_cast<const V4*>(wrow);
            const int nv = M >> 2;
#pragma unroll
            for (int c4 = 0; c4 < MP / 4; ++c4) {
                const V4 q = wrow4[c4 < nv ? c4 : nv - 1];
                const T m = (c4 < nv) ? rowmask : (T)0;
                w[4 * c4 + 0] = q.x * m; w[4 * c4 + 1] = q.y * m; w[4 * c4 + 2] = q.z * m; w[4 * c4 + 3] = q.w * m;
            }
        } else {
#pragma unroll
            for (int c = 0; c < MP; ++c) {
                const T v = wrow[c < M ? c : M - 1];
                w[c] = v * ((c < M) ? rowmask : (T)0);
            }
        }
    }
    T rc[NB], rp[NB], ex[NB];
    bool live[NB];           // stimulus exists (tail group may be partial)
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        live[s] = (s0 + s) < a.NB;
        const size_t vec = ((size_t)b * a.NB + (live[s] ? s0 + s : 0)) * M;
        rc[s] = (row_ok && live[s]) ? a.r[vec + i] : (T)0;
        rp[s] = rc[s];
        ex[s] = (row_ok && live[s]) ? a.ext[(a.ext_per_draw ? vec : (size_t)(s0 + s) * M) + i] : (T)0;
    }
    for (int c = threadIdx.x; c < 2 * NB * MP; c += blockDim.x) (&rbuf[0][0][0])[c] = (T)0;
    if (threadIdx.x < 3 * 2 * NB) (&flags[0][0][0])[threadIdx.x] = 0;
    __syncthreads();
    if (row_ok) {
#pragma unroll
        for (int s = 0; s < NB; ++s) rbuf[0][s][i] = rc[s];
    }
    __syncthreads();

    const T eps = (i < N) ? a.st.eps_E : a.st.eps_I;
    int code[NB], nsteps[NB];
    bool frozen[NB];
    int nlive = 0;
#pragma unroll
    for (int s = 0; s < NB; ++s) { code[s] = 1; nsteps[s] = a.st.max_iter; frozen[s] = !live[s]; nlive += live[s]; }

    int cur = 0;
    for (int step = 0; step < a.st.max_iter; ++step) {
        // ---- u = W r  (register-stationary W, DPP-broadcast r) ---------------------
        T acc[NB][NACC];
#pragma unroll
        for (int s = 0; s < NB; ++s)
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[s][q] = (T)0;
        ChunkLoop<0, KCH, NB, T>::run(acc, w, &rbuf[cur][0][lane16]);

        // ---- Euler update + per-row stop tests -------------------------------------
        const int slot = step % 3;
        unsigned ncmask = 0, hbmask = 0;
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            T u = acc[s][0];
#pragma unroll
            for (int q = 1; q < NACC; ++q) u += acc[s][q];
            u += ex[s];
            const T r1 = rc[s] + (-rc[s] + io_eval(u, a.io)) * eps;
            const bool upd = row_ok && !frozen[s];
            if (upd) {
                if (abs_t(r1 - rc[s]) >= a.st.atol) ncmask |= 1u << s;
                if (a.st.check_hard && r1 >= a.st.hard_stop) hbmask |= 1u << s;
                rp[s] = rc[s];
                rc[s] = r1;
            }
            if (row_ok) rbuf[cur ^ 1][s][i] = rc[s];
        }
        // wave-level vote, one LDS word per (flag kind, stimulus); plain stores of 1
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const bool nc = __any((ncmask >> s) & 1u);
            const bool hb = __any((hbmask >> s) & 1u);
            if ((threadIdx.x & 63) == 0) {
                if (nc) flags[slot][0][s] = 1;
                if (hb) flags[slot][1][s] = 1;
            }
        }
        if (threadIdx.x < 2 * NB) (&flags[(step + 1) % 3][0][0])[threadIdx.x] = 0;
        __syncthreads();
        cur ^= 1;

        // ---- uniform stop decisions (same for every thread of the workgroup) --------
        int nfrozen = 0;
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            if (!frozen[s]) {
                const int any_nc = flags[slot][0][s], any_hb = flags[slot][1][s];
                if (!any_nc) { code[s] = 0; nsteps[s] = step + 1; frozen[s] = true; }
                else if (any_hb) { code[s] = 2; nsteps[s] = step + 1; frozen[s] = true; }
            }
            nfrozen += frozen[s];
        }
        if (nfrozen == NB) break;
    }

    // ---- epilogue -----------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        if (!live[s]) continue;
        const size_t vec = ((size_t)b * a.NB + s0 + s) * M;
        if (row_ok) {
            a.r[vec + i] = rc[s];
            if (a.r_prev) a.r_prev[vec + i] = rp[s];
        }
        if (threadIdx.x == 0) {
            a.codes[(size_t)b * a.NB + s0 + s] = code[s];
            if (a.steps) a.steps[(size_t)b * a.NB + s0 + s] = nsteps[s];
        }
    }
}

// ---------------------------------------------------------------------------------
// Host-side dispatch
// ---------------------------------------------------------------------------------
template <typename T> struct RegwLimits;
template <> struct RegwLimits<float>  { static constexpr int max_kch = 13; };   // M <= 208
template <> struct RegwLimits<double> { static constexpr int max_kch = 7; };    // M <= 112 (2 VGPRs per value)

template <typename T>
static int pick_kch(int M) {
    const int need = (M + 15) / 16;
    const int ladder[] = {2, 4, 7, 10, 13};
    for (int k : ladder) if (need <= k && k <= RegwLimits<T>::max_kch) return k;
    return 0;
}
static int pick_nb(int NB) { return NB >= 8 ? 8 : (NB >= 4 ? 4 : (NB >= 2 ? 2 : 1)); }

template <typename T>
bool regw_supported(int M, int NB) { (void)NB; return (M % 2 == 0) && pick_kch<T>(M) != 0; }
template bool regw_supported<float>(int, int);
template bool regw_supported<double>(int, int);

template <typename T, int KCH, int NB>
static hipError_t launch_regw_k(const SolveArgs<T>& a, hipStream_t st) {
    // threads = 64*ceil(M/64) <= 64*ceil(16*KCH/64)
    constexpr int MAXT = 64 * ((16 * KCH + 63) / 64);
    // fp32: W row + a few dozen temporaries must fit 256 VGPRs to keep 2 waves per SIMD
    // (two workgroups of a 200-neuron solve per CU); fp64 needs the full 512.
    constexpr int MINW = (sizeof(T) == 4) ? 2 : 1;
    const int ngroups = (a.NB + NB - 1) / NB;
    const int threads = 64 * ((a.M + 63) / 64);
    hipLaunchKernelGGL((solve_regw_kernel<T, KCH, NB, MAXT, MINW>), dim3(a.B * ngroups), dim3(threads), 0, st, a);
    return hipGetLastError();
}

// Stimuli per workgroup.  fp32: up to 8 (KCH=13, NB=8 uses exactly 256 VGPRs, no spill).
// fp64 values take two VGPRs, so the "plumbing" fp64 path keeps NB <= 2 (1 at KCH=7).
template <typename T, int KCH>
static hipError_t launch_regw_nb(const SolveArgs<T>& a, hipStream_t st) {
    if constexpr (sizeof(T) == 4) {
        switch (pick_nb(a.NB)) {
            case 8: return launch_regw_k<T, KCH, 8>(a, st);
            case 4: return launch_regw_k<T, KCH, 4>(a, st);
            case 2: return launch_regw_k<T, KCH, 2>(a, st);
            default: return launch_regw_k<T, KCH, 1>(a, st);
        }
    } else {
        if (KCH < 7 && a.NB >= 2) return launch_regw_k<T, KCH, 2>(a, st);
        return launch_regw_k<T, KCH, 1>(a, st);
    }
}

template <typename T> hipError_t launch_regw(const SolveArgs<T>& a, hipStream_t st);
template <> hipError_t launch_regw<float>(const SolveArgs<float>& a, hipStream_t st) {
    switch (pick_kch<float>(a.M)) {
        case 2: return launch_regw_nb<float, 2>(a, st);
        case 4: return launch_regw_nb<float, 4>(a, st);
        case 7: return launch_regw_nb<float, 7>(a, st);
        case 10: return launch_regw_nb<float, 10>(a, st);
        case 13: return launch_regw_nb<float, 13>(a, st);
        default: return hipErrorInvalidValue;
    }
}
template <> hipError_t launch_regw<double>(const SolveArgs<double>& a, hipStream_t st) {
    switch (pick_kch<double>(a.M)) {
        case 2: return launch_regw_nb<double, 2>(a, st);
        case 4: return launch_regw_nb<double, 4>(a, st);
        case 7: return launch_regw_nb<double, 7>(a, st);
        default: return hipErrorInvalidValue;
    }
}

template <typename T>
hipError_t launch_stream(const SolveArgs<T>& a, hipStream_t st) {
    const size_t smem = 2 * (size_t)a.M * sizeof(T) + 6 * sizeof(int);
    hipLaunchKernelGGL((solve_stream_kernel<T>), dim3(a.B * a.NB), dim3(256), smem, st, a);
    return hipGetLastError();
}
template hipError_t launch_stream<float>(const SolveArgs<float>&, hipStream_t);
template hipError_t launch_stream<double>(const SolveArgs<double>&, hipStream_t);

}  // namespace ssn
