// Microbenchmark 2: pure VALU issue rates on gfx950 for the operand forms the SSN recurrence could use.
// No LDS, no barriers inside the timed loop.  W (208 values/lane) is register resident.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int NW = 208;

#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

// MODE 0 plain vgpr operand, 2 acc; 1 plain 8 acc; 2 sgpr operand; 3 dpp newbcast; 4 dpp quad_perm bcast; 5 pk_fma (vgpr pairs)
// 6 pk_fma with op_sel broadcast of r.lo; 7 readlane+fmac(sgpr) 1:4 ; 8 dpp row_shr:1
template<int MODE, int WPS>
__global__ void __launch_bounds__(256, WPS) kern(const float* __restrict__ W, float* __restrict__ out, int T, unsigned long long* clk){
  float w[NW];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  #pragma unroll
  for(int j=0;j<NW;j++) w[j] = W[(size_t)(gid % 4096) * NW + j];
  float r[16];
  #pragma unroll
  for(int j=0;j<16;j++) r[j] = W[(gid + j*64) % 4096] * 0.5f;
  float acc[8] = {0,0,0,0,0,0,0,0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  for(int t=0;t<T;t++){
    if constexpr (MODE==0){
      #pragma unroll
      for(int j=0;j<NW;j++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j&1]) : "v"(r[j&15]), "v"(w[j]));
    } else if constexpr (MODE==1){
      #pragma unroll
      for(int j=0;j<NW;j++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j&7]) : "v"(r[j&15]), "v"(w[j]));
    } else if constexpr (MODE==2){
      float s0 = __builtin_amdgcn_readfirstlane(r[0]), s1 = __builtin_amdgcn_readfirstlane(r[1]);
      float s2 = __builtin_amdgcn_readfirstlane(r[2]), s3 = __builtin_amdgcn_readfirstlane(r[3]);
      #pragma unroll
      for(int j=0;j<NW;j+=4){
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[0]) : "s"(s0), "v"(w[j]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[1]) : "s"(s1), "v"(w[j+1]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[2]) : "s"(s2), "v"(w[j+2]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[3]) : "s"(s3), "v"(w[j+3]));
      }
    } else if constexpr (MODE==3){
      #pragma unroll
      for(int k=0;k<NW/16;k++){
        #define M3(n) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(acc[n&3]) : "v"(r[k]), "v"(w[16*k+n]));
        R16(M3)
      }
    } else if constexpr (MODE==4){
      #pragma unroll
      for(int j=0;j<NW;j++) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc[j&3]) : "v"(r[j&15]), "v"(w[j]));
    } else if constexpr (MODE==5){
      v2f a0 = {acc[0],acc[1]}, a1 = {acc[2],acc[3]};
      #pragma unroll
      for(int j=0;j<NW;j+=4){
        v2f w0 = {w[j],w[j+1]}, w1 = {w[j+2],w[j+3]}; v2f rr = {r[j&15], r[(j+1)&15]};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(w0), "v"(rr));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(w1), "v"(rr));
      }
      acc[0]=a0.x; acc[1]=a0.y; acc[2]=a1.x; acc[3]=a1.y;
    } else if constexpr (MODE==6){
      v2f a0 = {acc[0],acc[1]}, a1 = {acc[2],acc[3]};
      #pragma unroll
      for(int j=0;j<NW;j+=4){
        v2f w0 = {w[j],w[j+1]}, w1 = {w[j+2],w[j+3]}; v2f rr = {r[j&15], r[(j+1)&15]};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a0) : "v"(w0), "v"(rr));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a1) : "v"(w1), "v"(rr));
      }
      acc[0]=a0.x; acc[1]=a0.y; acc[2]=a1.x; acc[3]=a1.y;
    } else if constexpr (MODE==7){
      #pragma unroll
      for(int j=0;j<NW;j+=4){
        float s;
        asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s) : "v"(r[(j>>2)&15]), "n"((j>>2)&63));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[0]) : "s"(s), "v"(w[j]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[1]) : "s"(s), "v"(w[j+1]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[2]) : "s"(s), "v"(w[j+2]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[3]) : "s"(s), "v"(w[j+3]));
      }
    } else if constexpr (MODE==8){
      #pragma unroll
      for(int j=0;j<NW;j++) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[j&3]) : "v"(r[j&15]), "v"(w[j]));
    }
    // keep r data dependent on acc so the loop cannot be hoisted, at negligible cost
    r[t&15] = acc[0]*1e-9f + r[t&15];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
  float s=0; 
  #pragma unroll
  for(int j=0;j<8;j++) s+=acc[j];
  out[gid] = s;
  if(threadIdx.x==0 && blockIdx.x==0){ clk[0]=t1-t0; clk[1]=rt1-rt0; }
}

template<int MODE, int WPS> int run(const char* name, int fma_per_instr, int instr_per_iter, const float* dW, float* dout, unsigned long long* dclk){
  const int T = 3000; const int blocks = 256*WPS;
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<MODE,WPS><<<blocks,256>>>(dW,dout,T/10,dclk); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  kern<MODE,WPS><<<blocks,256>>>(dW,dout,T,dclk);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  unsigned long long h[2]; CK(hipMemcpy(h,dclk,16,hipMemcpyDeviceToHost));
  double ghz = (double)h[0]/(double)h[1]*0.1;   // memtime ticks per 100MHz realtime tick
  double wave_instr_per_simd = (double)WPS * instr_per_iter * T;  // waves per SIMD * instrs
  double cyc = (double)h[0] / wave_instr_per_simd;
  double tf = 2.0*fma_per_instr*64.0*instr_per_iter*T*(blocks*4.0)/ (ms*1e-3) * 1e-12;
  printf("%-34s wps=%d  %.3f ms  clk=%.2f GHz  cycles/instr/SIMD=%.2f  -> %.1f TFLOP/s\n", name, WPS, ms, ghz, cyc, tf);
  return 0;
}

#define RUNALL(WPS) \
  if(run<0,WPS>("plain vgpr, 2 acc", 1, NW, dW,dout,dclk)) return 1; \
  if(run<1,WPS>("plain vgpr, 8 acc", 1, NW, dW,dout,dclk)) return 1; \
  if(run<2,WPS>("sgpr operand, 4 acc", 1, NW, dW,dout,dclk)) return 1; \
  if(run<3,WPS>("dpp row_newbcast, 4 acc", 1, NW, dW,dout,dclk)) return 1; \
  if(run<4,WPS>("dpp quad_perm, 4 acc", 1, NW, dW,dout,dclk)) return 1; \
  if(run<8,WPS>("dpp row_shr:1, 4 acc", 1, NW, dW,dout,dclk)) return 1; \
  if(run<5,WPS>("pk_fma vgpr", 2, NW/2, dW,dout,dclk)) return 1; \
  if(run<6,WPS>("pk_fma op_sel bcast", 2, NW/2, dW,dout,dclk)) return 1; \
  if(run<7,WPS>("readlane + 4 fmac(sgpr)", 1, NW + NW/4, dW,dout,dclk)) return 1;

int main(){
  float *dW,*dout; unsigned long long* dclk;
  std::vector<float> hW((size_t)4096*NW);
  for(size_t i=0;i<hW.size();i++) hW[i] = ((i*2654435761u)%1000)/1000.f*0.01f-0.005f;
  CK(hipMalloc(&dW,hW.size()*4)); CK(hipMalloc(&dout,4096*256*4)); CK(hipMalloc(&dclk,16));
  CK(hipMemcpy(dW,hW.data(),hW.size()*4,hipMemcpyHostToDevice));
  RUNALL(1)
  RUNALL(2)
  return 0;
}
