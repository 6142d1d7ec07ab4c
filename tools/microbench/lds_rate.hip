// Microbenchmark 4: ds_read_b128 throughput per CU for the two access patterns of the split tile kernels:
// "private" = lane tid reads 16 B at (k*232 + tid)*16 (the LDS-resident rows of W), "bcast" = the 8 lanes of a
// row group share one of 8 addresses 112 B apart (the state vector).  3 workgroups x 256 threads per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_rate lds_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float V4 __attribute__((ext_vector_type(4)));
template <int NPRIV, int NBC>
__global__ void __launch_bounds__(256, 3) kern(float* out, int T) {
    __shared__ __align__(16) float wl[(12 * 232 + 256) * 4];
    __shared__ __align__(16) float xs[8 * 28];
    for (int i = threadIdx.x; i < (12 * 232 + 256) * 4; i += 256) wl[i] = i * 1e-6f;
    for (int i = threadIdx.x; i < 8 * 28; i += 256) xs[i] = i * 1e-3f;
    __syncthreads();
    const int cg = threadIdx.x & 7;
    V4 acc = {0, 0, 0, 0};
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int k = 0; k < NPRIV; ++k) {
            V4 v = *reinterpret_cast<const V4*>(&wl[(k * 232 + threadIdx.x) * 4]);
            asm volatile("" : "+v"(v));
            acc += v;
        }
#pragma unroll
        for (int q = 0; q < NBC; ++q) {
            V4 v = *reinterpret_cast<const V4*>(&xs[cg * 28 + 4 * q]);
            asm volatile("" : "+v"(v));
            acc += v;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int NPRIV, int NBC> void run(float* out) {
    const int T = 4000, blocks = 768;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<NPRIV, NBC><<<blocks, 256>>>(out, 100); hipDeviceSynchronize();
    hipEventRecord(e0); kern<NPRIV, NBC><<<blocks, 256>>>(out, T); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double reads_per_cu = 12.0 * (NPRIV + NBC) * T;      // wave-level ds_read_b128 per CU
    printf("private %2d bcast %2d: %.3f ms  -> %.2f cycles@2.4GHz per wave ds_read_b128 per CU (%.0f B/clk/CU)\n", NPRIV, NBC, ms,
           ms * 1e-3 * 2.4e9 / reads_per_cu, 1024.0 * reads_per_cu / (ms * 1e-3 * 2.4e9));
}
int main() {
    float* out; hipMalloc(&out, 768 * 256 * 4);
    run<13, 0>(out); run<0, 7>(out); run<13, 7>(out); run<4, 0>(out);
    return 0;
}
