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
};

// ssn_solver.hip
template <typename T> bool regw_supported(int M, int NB);
template <typename T> hipError_t launch_regw(const SolveArgs<T>& a, hipStream_t st);
template <typename T> hipError_t launch_stream(const SolveArgs<T>& a, hipStream_t st);

// ssn_tile.hip
template <typename T> bool tile_supported(int M, int NB);
template <typename T> hipError_t launch_tile(const SolveArgs<T>& a, hipStream_t st);

// ssn_aux.hip
template <typename T> hipError_t launch_build_w(const T* z, const T* jds12, T* W, int B, int N, hipStream_t st);
template <typename T> hipError_t launch_stimulus(const T* bw, const T* con, T smooth, T* ext, int B, int NB, int N, hipStream_t st);
template <typename T> hipError_t launch_io_eval(const T* v, T* out, long count, const IoConsts<T>& io, hipStream_t st);
template <typename T> hipError_t launch_dot(const T* x, const T* y, T* out, int dim, hipStream_t st);

}  // namespace ssn
