// Probe of LDS-DMA semantics on gfx950 (run on the GPU box): hipcc --offload-arch=gfx950 -O2 ldsdma_probe.hip -o ldsdma_probe
//  (1) what an out-of-range lane of `buffer_load_dwordx4 ... offen lds` leaves in LDS (zeros, or the old contents);
//  (2) whether M0 addresses the whole 160-KiB LDS (destination base above 64 KiB);
//  (3) whether the SGPR offset takes part in the range check.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 rsrc(const void* base, unsigned nrec) {
  const unsigned long b = (unsigned long)base;
  return (i32x4){(int)(unsigned)b, (int)((b >> 32) & 0xffffu), (int)nrec, 0x00020000};
}
__device__ __forceinline__ void glds16(unsigned lds_base, i32x4 r, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_base), "v"(voff), "s"(r), "s"(soff) : "memory");
}
__global__ void probe(const unsigned* src, unsigned nbytes, unsigned* out, unsigned lds_off, int soff) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* l = (unsigned*)smem;
  const int lane = threadIdx.x;
  for (int i = lane; i < (lds_off + 1024) / 4; i += 64) l[i] = 0xDEADBEEFu;
  __syncthreads();
  const i32x4 r = rsrc(src, nbytes);
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem + lds_off;
  // lanes 0..31 in range (16 B each), lanes 32..63: voffset beyond num_records
  const unsigned voff = lane < 32 ? lane * 16u : 0x80000000u;
  glds16(__builtin_amdgcn_readfirstlane(base), r, voff, soff);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = l[lds_off / 4 + lane * 4 + i];
}
int main() {
  unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  unsigned *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 1024); hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  struct { unsigned lds_off; unsigned nbytes; int soff; const char* what; } cases[] = {
    {0, 4096, 0, "base 0, all in range except parked lanes"}, {100 * 1024, 4096, 0, "LDS destination at 100 KiB"},
    {0, 256, 0, "num_records 256: lanes 16..31 beyond the records"}, {0, 512, 1024, "soffset 1024 with num_records 512 (soffset range-checked?)"}};
  for (auto& c : cases) {
    hipMemset(o, 0xff, 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), c.lds_off + 2048, 0, d, c.nbytes, o, c.lds_off, c.soff);
    unsigned r[256]; hipError_t e = hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    printf("%s: %s\n", c.what, hipGetErrorString(e));
    for (int lane : {0, 1, 15, 16, 31, 32, 63}) printf("  lane %2d: %08x %08x %08x %08x\n", lane, r[lane * 4], r[lane * 4 + 1], r[lane * 4 + 2], r[lane * 4 + 3]);
  }
  return 0;
}
