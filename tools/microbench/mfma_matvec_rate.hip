// Microbenchmark 5: W (64 rows x 200 columns per wave, one row per lane, 200 VGPRs) times 8 state vectors per step with
// v_mfma_f32_4x4x1_16b_f32 (two or four interleaved accumulation chains), B operand from LDS (ds_read_b128 of 4 consecutive k for the lane's
// stimulus).  One 4-wave workgroup per CU (WPC=1) or two (WPC=2).  No epilogue, one barrier per step.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_matvec_rate mfma_matvec_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int MK = 200, RS = MK + 4;
template <int WPC, int CHAINS>
__global__ void __launch_bounds__(256, WPC) kern(const float* __restrict__ W, float* __restrict__ out, int T) {
    __shared__ __align__(16) float rbuf[8][RS];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    float wr[MK];
#pragma unroll
    for (int k = 0; k < MK; ++k) wr[k] = W[((blockIdx.x * 4 + w) * 64 + l) % 4096 * MK + k];
    for (int i = threadIdx.x; i < 8 * RS; i += 256) (&rbuf[0][0])[i] = 1e-3f * (i % 17);
    __syncthreads();
    f4 tot = {0, 0, 0, 0};
    for (int t = 0; t < T; ++t) {
        f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < MK / 4; ++q) {
            const f4 b0 = *reinterpret_cast<const f4*>(&rbuf[l & 3][4 * q]);
            const f4 b1 = *reinterpret_cast<const f4*>(&rbuf[4 + (l & 3)][4 * q]);
            if (CHAINS == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * q + e], b0[e], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * q + e], b1[e], a1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * q + e], b0[e], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * q + e], b1[e], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * q + e + 1], b0[e + 1], a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * q + e + 1], b1[e + 1], a3, 0, 0, 0);
                }
            }
        }
        tot += a0 + a1 + a2 + a3;
        __syncthreads();
        if (threadIdx.x < 8) rbuf[threadIdx.x][t & 63] = tot.x * 1e-9f;
        __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = tot.x + tot.y + tot.z + tot.w;
}
template <int WPC, int CHAINS> void run(const float* W, float* out) {
    const int T = 2000, blocks = 256 * WPC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<WPC, CHAINS><<<blocks, 256>>>(W, out, 50); hipDeviceSynchronize();
    hipEventRecord(e0); kern<WPC, CHAINS><<<blocks, 256>>>(W, out, T); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 256 * MK * 8 * (double)T * blocks;      // incl. padded rows (256 per workgroup)
    printf("chains=%d WPC=%d: %.3f ms, %.0f cycles@2.4GHz per step of %d workgroup(s) per CU, %.1f TFLOP/s (256-row slabs)\n", CHAINS, WPC, ms,
           ms * 1e-3 * 2.4e9 / T, WPC, flops / (ms * 1e-3) * 1e-12);
}
int main() {
    float *W, *out; hipMalloc(&W, 4096 * MK * 4); hipMalloc(&out, 512 * 256 * 4);
    hipMemset(W, 0, 4096 * MK * 4);
    run<1, 2>(W, out); run<2, 2>(W, out); run<1, 4>(W, out); run<2, 4>(W, out);
    return 0;
}
