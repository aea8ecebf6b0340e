// Launch arguments shared by the implicit-GEMM convolution kernels (conv4.hip: igemm4 and its thin variants; gemm5.hip: igemm5).
#pragma once
#include "common.h"
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
struct Igemm4Args {
  const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y;
  int N, ID, IH, IW;
  int RD, RH, RW;
  int TD, TH, TW;
  int rs[3], tb[3], ts[3], os[3], oo[3];
  long xsn, xsd, xsh, xsw;
  long ysn, ysd, ysh, ysw;
  int cink, cin_lanes, cout, cout_lanes, wrows;
  int act; float slope;
  long M;
  int tiles_m, tiles_n;
  unsigned xbytes, wbytes;  // extents for the buffer descriptors (< 2^30)
  int krow, cp;             // FLAT: packed row length (elements, multiple of 64), 8-lane pieces per tap
  int ksplit; float* part;  // split-K (grids that leave most CUs idle): ksplit > 1 -> fp32 partial sums [split][row][cout_lanes] into `part`
  int nbatch; long xbatch, wbatch, ybatch;  // batched GEMM (rg_gemm_batched): element strides between the problems; no split-K
  // igemm5 only — the GEMM form of the weight gradient (gemm5.hip rg_wgrad_gemm): B rows `wld` elements apart (0: ntaps * cink), a second batch level
  // (batch = b2 * nbatch1 + b1: operand B of batch (b1, b2) starts wbatch * b1 + wbatch2 * b2 elements in), and raw fp32 output: element (m, n) of
  // batch b goes to raw + b * raw_batch + m * raw_ld + n (no bias / activation / 16-bit store); raw = nullptr: off
  long wld; int nbatch1; long wbatch2; float* raw; long raw_ld, raw_batch;
  // igemm5 only — stream-K (gemm5.hip): sk_grid > 0 workgroups share the (tile, K-tile) space in equal contiguous runs; a run that starts inside a
  // tile leaves its fp32 partial in sk_part[workgroup] and raises sk_flags[workgroup]; the workgroup that finishes the tile adds them and lowers the flags again
  int sk_grid; float* sk_part; int* sk_flags; int sk_epoch;
  int d2s_c;  // igemm5 only, 0 = off: the columns are 8 parity classes x d2s_c channels of a sub-pixel (stride-2 transposed) convolution — column
              // c goes to channel c % d2s_c of output voxel 2 * row + parity(c / d2s_c) (depth-to-space in the epilogue; os = 2, bias per channel)
};

