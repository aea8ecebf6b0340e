#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const short* a, float* out){
  __shared__ __attribute__((aligned(16))) short lds[1024];
  int l = threadIdx.x;
  for(int i=l;i<1024;i+=64) lds[i]=a[i];
  __syncthreads();
  bf16x8 x = *(bf16x8*)&lds[(l&15)*32 + (l>>4)*8];
  s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(&lds[ (l&15)/4*16 + ((l&3)*4) ]));
  f32x4 acc = {0,0,0,0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, acc, 0,0,0);
  out[l] = acc[0]+acc[1]+t0[0];
}
extern "C" int rg_test(const short* a, float* out, hipStream_t s){ hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, a, out); return (int)hipGetLastError(); }
