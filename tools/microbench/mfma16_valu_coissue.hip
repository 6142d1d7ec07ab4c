// Microbenchmark: a wave streaming v_mfma_f32_16x16x32_f16 (46 per iteration, 4 accumulators: the chain of ssn_duo.hip)
// and a wave of the SAME SIMD running a VALU / transcendental stream shaped like that kernel's serial part
// (per iteration: 72 plain VALU + 14 transcendentals in ILP independent dependency chains).  512-thread workgroups:
// waves 0-3 matrix role, waves 4-7 vector role (wave i and i + 4 share a SIMD).  No barriers, no LDS.
// Reports cycles per iteration of each role alone and together, at s_setprio 0 and with the vector role at priority 3.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma16_valu_coissue mfma16_valu_coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float mf4 __attribute__((ext_vector_type(4)));
typedef _Float16 hv8 __attribute__((ext_vector_type(8)));
constexpr int NU = 23;

template <int ILP>
__global__ void __launch_bounds__(512, 2) kern(const float* __restrict__ W, float* out, int T, int roles, unsigned long long* clk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (!(roles & 1)) return;
        hv8 a[2 * NU];
#pragma unroll
        for (int k = 0; k < 2 * NU; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) a[k][e] = (_Float16)W[(lane * 8 + e + 37 * k) % 4096];
        hv8 b;
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = (_Float16)W[lane + e];
        mf4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int k = 0; k + 3 < 2 * NU; k += 4) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k], b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k + 1], b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k + 2], b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k + 3], b, c3, 0, 0, 0);
            }
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[44], b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[45], b, c1, 0, 0, 0);
            asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(b));
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        const mf4 s = (c0 + c1) + (c2 + c3);
        out[(size_t)blockIdx.x * 512 + threadIdx.x] = s.x + s.y + s.z + s.w;
        if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
    } else {
        if (!(roles & 2)) return;
        if (roles & 4) __builtin_amdgcn_s_setprio(3);
        float acc[8], x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i] = 0.f; x[i] = W[lane + 64 * i] * 0.01f; }
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int t = 0; t < T; ++t) {
            // 14 transcendentals + 72 plain, every instruction depending on the previous one of its chain (i % ILP)
#pragma unroll
            for (int i = 0; i < 86; ++i) {
                if (i % 6 == 0 && i < 84) asm volatile("v_exp_f32 %0, %0" : "+v"(acc[i % ILP]));
                else asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(acc[i % ILP]) : "v"(x[i & 7]));
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i];
        out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
        if (blockIdx.x == 0 && threadIdx.x == 256) clk[1] = t1 - t0;
    }
}

template <int ILP>
int run(const float* dW, float* dout, unsigned long long* dclk) {
    const int T = 4000, grid = 256;
    for (int roles : {1, 2, 3, 7}) {
        unsigned long long h[2] = {0, 0};
        CK(hipMemcpy(dclk, h, sizeof(h), hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern<ILP>, dim3(grid), dim3(512), 0, 0, dW, dout, T, roles, dclk);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern<ILP>, dim3(grid), dim3(512), 0, 0, dW, dout, T, roles, dclk);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost));
        printf("ILP %d roles=%d: kernel %.3f ms  matrix %.1f clk/iter (%.2f per MFMA)  vector %.1f clk/iter (72 valu + 14 trans)\n", ILP, roles, ms,
               (double)h[0] / T, (double)h[0] / T / 46, (double)h[1] / T);
    }
    return 0;
}

int main() {
    std::vector<float> hW(4096 + 64 * 8);
    for (size_t i = 0; i < hW.size(); ++i) hW[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
    float *dW, *dout; unsigned long long* dclk;
    CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dout, 256 * 512 * 4)); CK(hipMalloc(&dclk, 16));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    if (run<1>(dW, dout, dclk)) return 1;
    if (run<2>(dW, dout, dclk)) return 1;
    if (run<4>(dW, dout, dclk)) return 1;
    if (run<8>(dW, dout, dclk)) return 1;
    return 0;
}
