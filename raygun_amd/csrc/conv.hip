// Dispatcher of the implicit-GEMM convolution engine (Conv3d forward, Conv3d backward-data, ConvTranspose3d forward and its
// backward-data): the host describes the op as a grid of "rows" (output voxels of the launch), a tap lattice and affine
// row->input / row->output coordinate maps (include/raygun_hip.h), and this file picks the kernel family:
//   packing 64 (reduced channels padded per tap to 64): igemm4 (conv4.hip; activations straight into MFMA fragments) for
//               128/144-channel output tiles, igemm3 (conv3.hip; kw reuse through LDS) / igemm2 (conv2.hip) for narrow tiles;
//   packing 1  (FLAT: K flattened across taps): igemm4 FLAT and the thin-output kernels (conv4.hip).
// GEMM view:  D[channel][row] = sum_{tap, ci} Wp[channel][tap][ci] * X[row @ tap][ci]  — weights are the MFMA A operand,
// the gathered activations the B operand, so each lane ends up with 4 consecutive channels of one voxel (8-byte stores).
#include "common.h"
#include <stdlib.h>

extern "C" int rg_conv3d_kalign(int cin, int cout);
int rg_launch_igemm2(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p,
                     const void* zeros, hipStream_t st);
int rg_launch_igemm3(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p,
                     const void* zeros, hipStream_t st);
bool rg_igemm3_ok(const CL5& X, const rg_conv_params* p);
bool rg_igemm4_ok(const CL5& X, const rg_conv_params* p);
int rg_igemm4_tile_n(int cout);
bool rg_igemm5_wanted(const CL5& X, const CL5& Y, const rg_conv_params* p, size_t ws_bytes);   // gemm5.hip
size_t rg_igemm5_sk_bytes(const CL5& X, const CL5& Y, const rg_conv_params* p);
int rg_launch_igemm4(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p, hipStream_t st, int flat,
                     void* workspace, size_t ws_bytes);
__device__ __attribute__((aligned(128))) unsigned short rg_zero_page[64];  // zero-initialised with the code object
static const void* zero_page() {
  static const void* cache[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!cache[dev]) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(rg_zero_page)) != hipSuccess) return nullptr; cache[dev] = p; }
  return cache[dev];
}

int rg_launch_igemm5_d2s(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p, hipStream_t st, int dry);   // conv4.hip
// Stride-2 / pad-1 / 3^3 transposed convolution (ResNet.py:325-333's up path; the backward-data pass of the stride-2 convolutions :291-298) in its
// sub-pixel form with the depth-to-space shuffle in the GEMM's epilogue: p describes the (8 * C)-column, 2^3-tap convolution on the INPUT grid
// (R = input spatial size, cout = 8 * C, weights packed from ops._subpixel_weights_t), y is the (N, C, 2D, 2H, 2W) output, bias has C entries.
// Column block q = (qd, qh, qw) of row r lands on output voxel 2 r + q.  dry_run != 0: RG_OK if this problem can run this way, nothing is launched.
extern "C" int rg_conv3d_igemm_d2s(const rg_tensor* x, const void* w_packed, const float* bias, const rg_tensor* y, const rg_conv_params* p,
                                   int32_t dry_run, rg_stream s) {
  CL5 X, Y;
  int rc;
  if ((rc = rg_as_cl5(x, &X, "conv_d2s.x")) || (rc = rg_as_cl5(y, &Y, "conv_d2s.y"))) return rc;
  if (!p || (!dry_run && (!w_packed || (((uintptr_t)w_packed) & 15)))) RG_FAIL(RG_ERR_BAD_SHAPE, "conv_d2s: null / misaligned weights or params");
  if (p->cout != 8 * Y.C || Y.C % 16 || Y.lanes != Y.C || X.N != Y.N) RG_FAIL(RG_ERR_UNSUPPORTED, "conv_d2s: needs cout == 8 * C with C a multiple of 16 (C = %d, cout = %d)", Y.C, p->cout);
  if (Y.D != 2 * p->R[0] || Y.H != 2 * p->R[1] || Y.W != 2 * p->R[2]) RG_FAIL(RG_ERR_BAD_SHAPE, "conv_d2s: y must be twice the row grid along every axis");
  if (rg_conv3d_kalign(X.C, p->cout) != 64) RG_FAIL(RG_ERR_UNSUPPORTED, "conv_d2s: the 64-lane K packing only");
  for (int d = 0; d < 3; ++d) if (p->R[d] <= 0 || p->T[d] <= 0 || p->rstride[d] != 1) RG_FAIL(RG_ERR_BAD_SHAPE, "conv_d2s: unit row stride, non-empty grid");
  if ((rc = rg_launch_igemm5_d2s(X, Y, w_packed, bias, p, (hipStream_t)s, dry_run))) return rc;
  if (!dry_run) RG_CHECK_LAUNCH();
  return RG_OK;
}

extern "C" int rg_conv3d_igemm(const rg_tensor* x, const void* w_packed, const float* bias, const rg_tensor* y,
                               const rg_conv_params* p, rg_stream s) {
  return rg_conv3d_igemm2(x, w_packed, bias, y, p, nullptr, 0, s);
}
// Upper bound of what rg_conv3d_igemm2 can use: split-K partial sums, fp32 [<= 8 splits][rows][output lanes], for launches whose
// tile grid would leave most of the chip idle (few rows, long K).  0 when the problem is large enough never to split.
extern "C" int64_t rg_conv3d_igemm_workspace_bytes(const rg_tensor* x, const rg_tensor* y, const rg_conv_params* p) {
  CL5 X, Y;
  if (rg_as_cl5(x, &X, "conv.x") || rg_as_cl5(y, &Y, "conv.y") || !p) return -1;
  const long M = (long)X.N * p->R[0] * p->R[1] * p->R[2];
  const long tiles64 = ((M + 63) / 64) * ((Y.lanes + 143) / 144);   // the smallest tile the engine uses
  const int64_t sk = rg_conv3d_kalign(X.C, p->cout) == 64 ? (int64_t)rg_igemm5_sk_bytes(X, Y, p) : 0;   // stream-K partial slots (gemm5.hip)
  if (tiles64 > 1024 || Y.lanes < 64) return sk;                     // cannot be at most half a round of any tile shape / thin kernels
  const int64_t split = (int64_t)8 * M * Y.lanes * 4;
  return split > sk ? split : sk;
}
extern "C" int rg_conv3d_igemm2(const rg_tensor* x, const void* w_packed, const float* bias, const rg_tensor* y,
                                const rg_conv_params* p, void* workspace, size_t ws_bytes, rg_stream s) {
  CL5 X, Y;
  int rc;
  if ((rc = rg_as_cl5(x, &X, "conv.x")) || (rc = rg_as_cl5(y, &Y, "conv.y"))) return rc;
  if (!w_packed || !p) RG_FAIL(RG_ERR_BAD_SHAPE, "conv: null weights/params");
  if (((uintptr_t)w_packed) & 15) RG_FAIL(RG_ERR_BAD_SHAPE, "conv: packed weights must be 16B aligned");
  if (Y.C != p->cout || X.N != Y.N) RG_FAIL(RG_ERR_BAD_SHAPE, "conv: y channels %d != cout %d or batch mismatch", Y.C, p->cout);
  const int Yd[3] = {Y.D, Y.H, Y.W};
  for (int d = 0; d < 3; ++d) {
    if (p->R[d] <= 0 || p->T[d] <= 0) RG_FAIL(RG_ERR_BAD_SHAPE, "conv: empty row grid / tap lattice");
    const long last = (long)(p->R[d] - 1) * p->ostride[d] + p->ooffset[d];
    if (p->ooffset[d] < 0 || last >= Yd[d]) RG_FAIL(RG_ERR_BAD_SHAPE, "conv: row grid writes outside y along dim %d (last %ld, size %d)", d, last, Yd[d]);
  }
  if (rg_conv3d_kalign(X.C, p->cout) == 64) {
    const void* z = zero_page();
    if (!z) RG_FAIL(RG_ERR_HIP, "conv: cannot resolve the zero page symbol");
    // igemm4 (activations straight into fragments) for 128/144-channel tiles; the kw-reuse kernel keeps the unpadded
    // 128-channel stride-1 layers (787 vs 731 TFLOP/s) and both LDS kernels the narrow tiles
    const bool i3 = rg_igemm3_ok(X, p);
    if (rg_igemm4_ok(X, p) && (!(i3 && rg_igemm4_tile_n(p->cout) == 8) || rg_igemm5_wanted(X, Y, p, ws_bytes))) { if ((rc = rg_launch_igemm4(X, Y, w_packed, bias, p, (hipStream_t)s, 0, workspace, ws_bytes))) return rc; }
    else if (i3) rg_launch_igemm3(X, Y, w_packed, bias, p, z, (hipStream_t)s);
    else rg_launch_igemm2(X, Y, w_packed, bias, p, z, (hipStream_t)s);
    RG_CHECK_LAUNCH();
    return RG_OK;
  }
  if (rg_conv3d_kalign(X.C, p->cout) == 1) {  // K flattened across taps, activations straight into fragments
    if ((rc = rg_launch_igemm4(X, Y, w_packed, bias, p, (hipStream_t)s, 1, workspace, ws_bytes))) return rc;
    RG_CHECK_LAUNCH();
    return RG_OK;
  }
  RG_FAIL(RG_ERR_UNSUPPORTED, "conv: unknown packing mode");
}
