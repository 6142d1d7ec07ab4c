// Microbenchmark: do the fp32 MFMA chain of one wave and the VALU / transcendental stream of ANOTHER wave of the
// same SIMD overlap on gfx950?  512-thread workgroups, waves 0-3 = "matrix" role, waves 4-7 = "valu" role
// (wave i and i + 4 share SIMD i).  No barriers, no LDS.  Reports cycles per iteration of each role alone and together.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_coissue mfma_valu_coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float mf4 __attribute__((ext_vector_type(4)));
constexpr int NW = 200;

// roles: bit 0 matrix waves active, bit 1 valu waves active.  VK: 0 = v_fmac stream, 1 = v_exp stream, 2 = mix like the
// serial part (per 4 evaluations: 8 transcendentals + 60 plain)
template <int VK>
__global__ void __launch_bounds__(512, 2) kern(const float* __restrict__ W, float* out, int T, int roles, int nv, unsigned long long* clk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (!(roles & 1)) return;
        float w[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) w[k] = W[(size_t)((blockIdx.x * 4 + wave) * 64 + lane) % 4096 * NW + k];
        mf4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float b = W[lane];
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int k = 0; k < NW; k += 4) {
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[k], b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[k + 1], b, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[k + 2], b, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[k + 3], b, a3, 0, 0, 0);
            }
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b));
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        const mf4 s = (a0 + a1) + (a2 + a3);
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
            for (int r = 0; r < nv; ++r) {
                if constexpr (VK == 0) {
#pragma unroll
                    for (int i = 0; i < 64; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i & 7]) : "v"(x[i & 7]), "v"(x[(i + 3) & 7]));
                } else if constexpr (VK == 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(acc[i & 7]) : "v"(x[i & 7]));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(acc[i & 7]) : "v"(x[i & 7]));
#pragma unroll
                    for (int i = 0; i < 60; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i & 7]) : "v"(x[i & 7]), "v"(x[(i + 3) & 7]));
                }
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

template <int VK>
int run(const char* name, const float* dW, float* dout, unsigned long long* dclk, int nv) {
    const int T = 2000, grid = 256;
    for (int roles : {1, 2, 3, 7}) {
        unsigned long long h[2] = {0, 0};
        CK(hipMemcpy(dclk, h, sizeof(h), hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern<VK>, dim3(grid), dim3(512), 0, 0, dW, dout, T, roles, nv, dclk);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern<VK>, dim3(grid), dim3(512), 0, 0, dW, dout, T, roles, nv, dclk);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-28s nv=%d roles=%d: kernel %.3f ms  matrix %.1f clk/iter (%.2f per MFMA)  valu %.1f clk/iter\n", name, nv, roles, ms,
               (double)h[0] / T, (double)h[0] / T / NW, (double)h[1] / T);
    }
    return 0;
}

int main() {
    std::vector<float> hW(4096 * NW);
    for (size_t i = 0; i < hW.size(); ++i) hW[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
    float *dW, *dout; unsigned long long* dclk;
    CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dout, 256 * 512 * 4)); CK(hipMalloc(&dclk, 16));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    if (run<0>("fmac x64 per rep", dW, dout, dclk, 2)) return 1;
    if (run<0>("fmac x64 per rep", dW, dout, dclk, 6)) return 1;
    if (run<1>("exp x16 per rep", dW, dout, dclk, 2)) return 1;
    if (run<1>("exp x16 per rep", dW, dout, dclk, 6)) return 1;
    if (run<2>("8 exp + 60 fmac per rep", dW, dout, dclk, 2)) return 1;
    if (run<2>("8 exp + 60 fmac per rep", dW, dout, dclk, 4)) return 1;
    return 0;
}
