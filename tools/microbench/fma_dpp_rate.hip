// Microbenchmark: issue rate of v_fmac_f32 with a DPP row_newbcast operand vs plain VGPR / SGPR operands
// on gfx950, with the W row held in VGPRs (the register-stationary layout the SSN solver uses).
// Build: hipcc --offload-arch=gfx950 -O3 -o fma_dpp_rate fma_dpp_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <type_traits>

#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template<int K> struct Unroll{
  template<typename F> static __device__ __forceinline__ void run(F&& f){ Unroll<K-1>::run(f); f(std::integral_constant<int,K-1>{}); }
};
template<> struct Unroll<0>{ template<typename F> static __device__ __forceinline__ void run(F&&){} };

// 16 DPP FMAs in one asm statement: acc += bcast(r, n) * w[n]
#define F16(acc, r, w, base) \
  asm volatile( \
   "v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t" \
   "v_fmac_f32_dpp %0, %1, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t" \
   : "+v"(acc) : "v"(r), \
     "v"(w[base+0]),"v"(w[base+1]),"v"(w[base+2]),"v"(w[base+3]),"v"(w[base+4]),"v"(w[base+5]),"v"(w[base+6]),"v"(w[base+7]), \
     "v"(w[base+8]),"v"(w[base+9]),"v"(w[base+10]),"v"(w[base+11]),"v"(w[base+12]),"v"(w[base+13]),"v"(w[base+14]),"v"(w[base+15]))

constexpr int KCH = 13;           // 13 chunks of 16 columns = 208 W registers per lane
constexpr int NW = KCH * 16;

// MODE 0: DPP bcast (asm, 1 accumulator per chunk parity), MODE 1: plain VGPR operand (compiler), MODE 2: NACC=4 dpp
template<int MODE>
__global__ void __launch_bounds__(256, 2) kern(const float* __restrict__ W, const float* __restrict__ r0, float* __restrict__ out, int T){
  float w[NW];
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  #pragma unroll
  for(int j=0;j<NW;j++) w[j] = W[(size_t)(row % 4096) * NW + j];
  __shared__ float rs[2][NW];
  for(int j=threadIdx.x;j<NW;j+=blockDim.x) rs[0][j] = r0[j];
  __syncthreads();
  float mine = 0.f;
  for(int t=0;t<T;t++){
    const float* rb = rs[t&1];
    float rk[KCH];
    #pragma unroll
    for(int k=0;k<KCH;k++) rk[k] = rb[16*k + (threadIdx.x & 15)];
    float a0=0.f, a1=0.f;
    if constexpr (MODE==0){
      #pragma unroll
      for(int k=0;k<KCH;k++){
        if (k&1) { F16(a1, rk[k], w, 16*k); } else { F16(a0, rk[k], w, 16*k); }
      }
    } else {
      #pragma unroll
      for(int k=0;k<KCH;k++){
        #pragma unroll
        for(int n=0;n<16;n++){ if(n&1) a1 = fmaf(w[16*k+n], rk[k], a1); else a0 = fmaf(w[16*k+n], rk[k], a0); }
      }
    }
    mine = 0.999f*mine + 1e-3f*(a0+a1);
    if(threadIdx.x < NW) rs[(t+1)&1][threadIdx.x] = mine;
    __syncthreads();
  }
  out[row] = mine;
}

template<int MODE> int run(const char* name, int blocks, int T, const float* dW, const float* dr, float* dout){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<MODE><<<blocks,256>>>(dW,dr,dout,T/10); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  kern<MODE><<<blocks,256>>>(dW,dr,dout,T);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  double fma = (double)blocks*256*NW*T;
  printf("%-28s blocks=%5d T=%d  %.3f ms  %.2f TFLOP/s (fp32 FMA=2)  cycles/step/WG@2.4GHz=%.0f\n", name, blocks, T, ms, 2*fma/ms*1e-9,
         ms*1e-3*2.4e9/T/((blocks+255)/256));
  return 0;
}

int main(){
  const int T=4000;
  float *dW,*dr,*dout;
  std::vector<float> hW((size_t)4096*NW), hr(NW);
  for(size_t i=0;i<hW.size();i++) hW[i] = ((i*2654435761u)%1000)/1000.f*0.01f-0.005f;
  for(int i=0;i<NW;i++) hr[i]=0.1f*i;
  CK(hipMalloc(&dW,hW.size()*4)); CK(hipMalloc(&dr,NW*4)); CK(hipMalloc(&dout,4096*256*4));
  CK(hipMemcpy(dW,hW.data(),hW.size()*4,hipMemcpyHostToDevice)); CK(hipMemcpy(dr,hr.data(),NW*4,hipMemcpyHostToDevice));
  for(int blocks : {256, 512, 1024, 2048}){
    if(run<0>("dpp row_newbcast (asm)", blocks, T, dW, dr, dout)) return 1;
    if(run<1>("plain vgpr operand", blocks, T, dW, dr, dout)) return 1;
  }
  return 0;
}
