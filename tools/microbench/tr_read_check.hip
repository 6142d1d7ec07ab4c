// Check of the transposed LDS read the fused adjoint + dL/dW sweep relies on (ds_read_b64_tr_b16 through
// __builtin_amdgcn_ds_read_tr16_b64_v4i16): the B image of the two-draw kernels (csrc/ssn_duo.hip: row = (k tile, k octet),
// 256 B = 16 columns x 8 neurons of fp16, column c = 8 part + stimulus) read as the operand of a product that sums over
// (part, stimulus): lane (li, lg) must receive the 8 stimuli of part p(lg) for neuron 16 r + li.
// Also: a 256-thread kernel holding 100 accumulator tiles (400 registers) -- one wave per SIMD may use the 512-entry file.
// Build: hipcc --offload-arch=gfx950 -O3 -o tr_read_check tr_read_check.hip ; run: ./tr_read_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef short sv4 __attribute__((ext_vector_type(4)));
typedef float mf4 __attribute__((ext_vector_type(4)));
typedef _Float16 hv8 __attribute__((ext_vector_type(8)));
constexpr int NKT = 7, BROW = 256, BB = NKT * 4 * BROW;

__global__ void __launch_bounds__(64) tr_kernel(short* out /* [13][2 forms][64 lanes][8] */) {
    __shared__ __align__(16) short img[BB / 2];
    for (int e = threadIdx.x; e < BB / 2; e += 64) {
        const int row = e / 128, c = (e % 128) / 8, nn = e % 8;           // row = ktile * 4 + octet
        img[e] = (short)((c << 8) | (row * 8 + nn));                       // (column, neuron)
    }
    __syncthreads();
    const int lane = threadIdx.x, lg = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    using LdsV = __attribute__((address_space(3))) sv4*;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) short*)img;
    for (int r = 0; r < 13; ++r) {
        for (int form = 0; form < 2; ++form) {
            const int part = form == 0 ? (lg >> 1) : (lg & 1);            // A operand: parts h h m m; B operand: h m h m
            const unsigned a0 = base + (unsigned)(r * 512 + (pp >> 1) * 256 + (8 * part + q) * 16 + (pp & 1) * 8);
            const sv4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsV)(size_t)a0);
            const sv4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsV)(size_t)(a0 + 64u));
            short* o = out + ((r * 2 + form) * 64 + lane) * 8;
            o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w;
        }
    }
}

__global__ void __launch_bounds__(256) big_kernel(const float* in, float* out, int T) {
    hv8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)in[threadIdx.x + e]; b[e] = (_Float16)in[threadIdx.x + 8 + e]; }
    mf4 acc[100];
#pragma unroll
    for (int k = 0; k < 100; ++k) acc[k] = mf4{0, 0, 0, 0};
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int k = 0; k < 100; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 100; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    short* d; CK(hipMalloc(&d, 13 * 2 * 64 * 8 * sizeof(short)));
    hipLaunchKernelGGL(tr_kernel, dim3(1), dim3(64), 0, 0, d);
    CK(hipDeviceSynchronize());
    std::vector<short> h(13 * 2 * 64 * 8);
    CK(hipMemcpy(h.data(), d, h.size() * sizeof(short), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int r = 0; r < 13; ++r) for (int form = 0; form < 2; ++form) for (int lane = 0; lane < 64; ++lane) for (int s = 0; s < 8; ++s) {
        const int lg = lane >> 4, i = lane & 15, part = form == 0 ? (lg >> 1) : (lg & 1);
        const int want = ((8 * part + s) << 8) | (16 * r + i);
        const int got = (unsigned short)h[((r * 2 + form) * 64 + lane) * 8 + s];
        if (got != want) { if (bad < 8) printf("r %d form %d lane %d s %d: got (%d, %d) want (%d, %d)\n", r, form, lane, s, got >> 8, got & 255, want >> 8, want & 255); ++bad; }
    }
    printf("transposed read of the chain image: %d mismatches of %zu\n", bad, h.size());
    float *din, *dout; CK(hipMalloc(&din, 4096 * 4)); CK(hipMalloc(&dout, 256 * 256 * 4));
    CK(hipMemset(din, 0, 4096 * 4));
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(big_kernel)));
    printf("big_kernel: numRegs %d, local (spill) bytes %zu\n", fa.numRegs, (size_t)fa.localSizeBytes);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(big_kernel, dim3(256), dim3(256), 0, 0, din, dout, 1000);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(big_kernel, dim3(256), dim3(256), 0, 0, din, dout, 10000);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("big_kernel: %.3f ms for 10000 x 100 MFMAs per wave -> %.1f ns per MFMA (one wave per SIMD)\n", ms, ms * 1e6 / 1e6);
    return bad != 0;
}
