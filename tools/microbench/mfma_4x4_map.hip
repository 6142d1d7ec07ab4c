#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
// out[row][stim] = sum_k W[row][k] * R[stim][k], row < 64, stim < 4, K = 8
__global__ void k(const float* W, const float* R, float* out, int K) {
    const int l = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (int kk = 0; kk < K; ++kk) {
        const float a = W[l * K + kk];
        const float b = R[(l % 4) * K + kk];
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    }
    const int blk = l / 4, j = l % 4;
    for (int v = 0; v < 4; ++v) out[(4 * blk + v) * 4 + j] = acc[v];
}
int main() {
    const int K = 8;
    float hW[64 * K], hR[4 * K], hout[64 * 4];
    for (int i = 0; i < 64 * K; ++i) hW[i] = (float)((i * 7) % 13) - 6;
    for (int i = 0; i < 4 * K; ++i) hR[i] = (float)((i * 5) % 11) - 5;
    float *W, *R, *out; hipMalloc(&W, sizeof hW); hipMalloc(&R, sizeof hR); hipMalloc(&out, sizeof hout);
    hipMemcpy(W, hW, sizeof hW, hipMemcpyHostToDevice); hipMemcpy(R, hR, sizeof hR, hipMemcpyHostToDevice);
    k<<<1, 64>>>(W, R, out, K); hipMemcpy(hout, out, sizeof hout, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 64; ++r) for (int s = 0; s < 4; ++s) {
        float ref = 0; for (int kk = 0; kk < K; ++kk) ref += hW[r * K + kk] * hR[s * K + kk];
        if (ref != hout[r * 4 + s]) { if (bad < 5) printf("mismatch row %d stim %d: %g vs %g\n", r, s, hout[r*4+s], ref); ++bad; }
    }
    printf("bad=%d\n", bad);
    return 0;
}
