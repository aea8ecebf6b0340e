// Implicit-GEMM 3-D convolution engine for gfx950 (CDNA4), bf16 MFMA 16x16x32, fp32 accumulate.
//
// One kernel serves Conv3d forward, Conv3d backward-data, ConvTranspose3d forward and its
// backward-data: the host describes the op as a grid of "rows" (output voxels of the launch), a
// tap lattice and affine row->input / row->output coordinate maps (include/raygun_hip.h).
//
// GEMM view:  D[channel][row] = sum_{tap, ci} Wp[channel][tap][ci] * X[row @ tap][ci]
//   * the weights are the MFMA A operand and the gathered activations the B operand, so that each
//     lane ends up holding 4 CONSECUTIVE CHANNELS of one voxel -> 8-byte channels-last stores;
//   * K is walked tap by tap in 32-channel chunks; a chunk of a row is 64 contiguous bytes of the
//     channels-last input, fetched as four 16-byte lanes (coalesced), staged through LDS;
//   * LDS tiles are [rows][32 ch] with 64-byte rows and the 16-byte slot XOR-swizzled by
//     ((row>>3)&1)<<1, which makes every ds_read_b128 lane group hit 16 distinct slots;
//   * double-buffered LDS, register prefetch of the next K-step, one barrier per step.
#include "common.h"
#include <stdlib.h>

struct IgemmArgs {
  const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y;
  int N, ID, IH, IW;
  int RD, RH, RW;
  int TD, TH, TW;
  int rs[3], tb[3], ts[3], os[3], oo[3];
  long xsn, xsd, xsh, xsw;
  long ysn, ysd, ysh, ysw;
  int cin32;      // K per tap in the packed weights (multiple of 32)
  int cin_lanes;  // readable lanes of x (multiple of 8)
  int cout;       // real output channels
  int cout_lanes; // lanes of y to write (multiple of 8)
  int wrows;      // rows of packed weights (multiple of 16)
  int act; float slope;
  long M;         // total rows = N*RD*RH*RW
};

__device__ __forceinline__ int swz(int row, int slot) { return slot ^ (((row >> 3) & 1) << 1); }

template <int BM, int FN>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmArgs a) {
  constexpr int BN = FN * 16;
  constexpr int FM = BM / 64;            // 16-row fragments per wave (4 waves split the rows)
  constexpr int A_PER_T = BM * 4 / 256;  // 16-byte chunks of the activation tile per thread
  constexpr int B_CHUNKS = BN * 4;
  constexpr int B_PER_T = (B_CHUNKS + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * (BM + BN) * 32];
  bf16_t* As = lds;                  // [2][BM][32]
  bf16_t* Bs = lds + 2 * BM * 32;    // [2][BN][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- per-thread gather rows
  int rd0[A_PER_T], rh0[A_PER_T], rw0[A_PER_T];
  const bf16_t* xrow[A_PER_T];
  bool rvalid[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int q = tid + 256 * i;
    const long m = m0 + (q >> 2);
    rvalid[i] = m < a.M;
    long t = rvalid[i] ? m : 0;
    const int rw = (int)(t % a.RW); t /= a.RW;
    const int rh = (int)(t % a.RH); t /= a.RH;
    const int rd = (int)(t % a.RD); t /= a.RD;
    rd0[i] = rd * a.rs[0] + a.tb[0]; rh0[i] = rh * a.rs[1] + a.tb[1]; rw0[i] = rw * a.rs[2] + a.tb[2];
    xrow[i] = a.x + t * a.xsn;
  }
  const int agrp = tid & 3;

  const int nchunk = a.cin32 >> 5;
  const int ntaps = a.TD * a.TH * a.TW;
  const int nk = ntaps * nchunk;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  bf16x8 ra[A_PER_T], rb[B_PER_T];
  int tap = 0, chunk = 0, kd = 0, kh = 0, kw = 0;
  long toff[A_PER_T]; bool tin[A_PER_T];

  auto tap_setup = [&]() {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int id = rd0[i] + kd * a.ts[0], ih = rh0[i] + kh * a.ts[1], iw = rw0[i] + kw * a.ts[2];
      tin[i] = rvalid[i] && (unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW;
      toff[i] = id * a.xsd + ih * a.xsh + iw * a.xsw;
    }
  };
  auto gload = [&]() {
    const int c = chunk * 32 + agrp * 8;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      bf16x8 v = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (tin[i] && c < a.cin_lanes) v = *(const bf16x8*)(xrow[i] + toff[i] + c);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int q = tid + 256 * i;
      const int n = n0 + (q >> 2);
      bf16x8 v = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (q < B_CHUNKS && n < a.wrows) v = *(const bf16x8*)(a.w + ((long)n * ntaps + tap) * a.cin32 + chunk * 32 + (q & 3) * 8);
      rb[i] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int q = tid + 256 * i, row = q >> 2;
      *(bf16x8*)(As + (buf * BM + row) * 32 + swz(row, q & 3) * 8) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int q = tid + 256 * i, row = q >> 2;
      if (q < B_CHUNKS) *(bf16x8*)(Bs + (buf * BN + row) * 32 + swz(row, q & 3) * 8) = rb[i];
    }
  };
  auto advance = [&]() {
    if (++chunk == nchunk) {
      chunk = 0; ++tap;
      if (++kw == a.TW) { kw = 0; if (++kh == a.TH) { kh = 0; ++kd; } }
      tap_setup();
    }
  };

  tap_setup();
  gload();
  lstore(0);
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    const bool more = ks + 1 < nk;
    if (more) { advance(); gload(); }
    bf16x8 xf[FM], wf[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = wave * (BM / 4) + i * 16 + fr;
      xf[i] = *(const bf16x8*)(As + (buf * BM + row) * 32 + swz(row, fg) * 8);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int row = j * 16 + fr;
      wf[j] = *(const bf16x8*)(Bs + (buf * BN + row) * 32 + swz(row, fg) * 8);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds channels c0..c0+3 of voxel (lane&15) for every fragment
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const long m = m0 + wave * (BM / 4) + i * 16 + fr;
    if (m >= a.M) continue;
    long t = m;
    const int rw = (int)(t % a.RW); t /= a.RW;
    const int rh = (int)(t % a.RH); t /= a.RH;
    const int rd = (int)(t % a.RD); t /= a.RD;
    bf16_t* yrow = a.y + t * a.ysn + (long)(rd * a.os[0] + a.oo[0]) * a.ysd + (long)(rh * a.os[1] + a.oo[1]) * a.ysh +
                   (long)(rw * a.os[2] + a.oo[2]) * a.ysw;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int c = n0 + j * 16 + fg * 4;
      if (c >= a.cout_lanes) continue;
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = 0.f;
        if (c + r < a.cout) {
          v = acc[i][j][r] + (a.bias ? a.bias[c + r] : 0.f);
          v = act_fwd(v, a.act, a.slope);
        }
        o[r] = (short)f2bf(v);
      }
      *(bf16x4*)(yrow + c) = o;
    }
  }
}

// ---- thin outputs (Cout <= 16): direct-fragment kernel ---------------------------------------------------------------
// These layers (1/12-channel U-Net levels, 1x1x1 heads, 64->1 tails and the backward-data of 12->72 etc.) are HBM bound:
// 16 output channels cannot feed a tile pipeline.  No LDS, no barrier: every wave loads its MFMA operands straight from
// global memory in fragment layout — lane (row r = l&15, k-group g = l>>4) reads the 16 bytes (8 channel lanes) of voxel
// row r for K-group 4*step+g; consecutive rows are consecutive voxels, so one wave-instruction covers 16 rows x 16..64
// contiguous bytes per tap.  K is walked in 8-lane groups across taps (K = taps*lanes, no per-tap padding to 32), the
// (tap -> offset) table sits in LDS, loads of step k+1 are in flight during the MFMAs of step k; occupancy hides the rest.
struct GroupInfo { int off; short dd, dh, dw, pad; };
template <int FM, bool CHECK>
__global__ __launch_bounds__(256) void igemm_direct_kernel(IgemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) GroupInfo gtab[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gpt = a.cin_lanes >> 3;                 // 8-lane groups per tap
  const int ntaps = a.TD * a.TH * a.TW;
  const int ngroups = ntaps * gpt;
  for (int G = tid; G < ngroups; G += 256) {
    const int tap = G / gpt, c8 = G - tap * gpt;
    const int kw = tap % a.TW, kh = (tap / a.TW) % a.TH, kd = tap / (a.TW * a.TH);
    GroupInfo gi;
    gi.dd = (short)(kd * a.ts[0]); gi.dh = (short)(kh * a.ts[1]); gi.dw = (short)(kw * a.ts[2]); gi.pad = 0;
    gi.off = (int)(gi.dd * a.xsd + gi.dh * a.xsh + gi.dw * a.xsw) + c8 * 8;
    gtab[G] = gi;
  }
  __syncthreads();
  const int fr = lane & 15, fg = lane >> 4;
  const long mw = (long)blockIdx.x * (64 * FM) + wave * (16 * FM);
  int rd0[FM], rh0[FM], rw0[FM]; long rbase[FM]; bool rv[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const long m = mw + i * 16 + fr;
    rv[i] = m < a.M;
    long t = rv[i] ? m : 0;
    const int rw = (int)(t % a.RW); t /= a.RW;
    const int rh = (int)(t % a.RH); t /= a.RH;
    const int rd = (int)(t % a.RD); t /= a.RD;
    rd0[i] = rd * a.rs[0] + a.tb[0]; rh0[i] = rh * a.rs[1] + a.tb[1]; rw0[i] = rw * a.rs[2] + a.tb[2];
    rbase[i] = t * a.xsn + rd0[i] * a.xsd + rh0[i] * a.xsh + rw0[i] * a.xsw;
  }
  const int krow = ngroups * 8;                      // packed weight row length (no padding)
  const bf16_t* wrow = a.w + (long)fr * krow;
  const int nsteps = (ngroups + 3) >> 2;
  f32x4 acc[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 xf[FM], wf, xn[FM], wn;
  auto fetch = [&](int ks, bf16x8 (&xo)[FM], bf16x8& wo) {
    const int G = ks * 4 + fg;
    const bool gv = G < ngroups;
    const GroupInfo gi = gtab[gv ? G : 0];
    wo = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (gv) wo = *(const bf16x8*)(wrow + G * 8);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      bool ok = gv & rv[i];
      if (CHECK) ok = ok & ((unsigned)(rd0[i] + gi.dd) < (unsigned)a.ID) & ((unsigned)(rh0[i] + gi.dh) < (unsigned)a.IH) &
                      ((unsigned)(rw0[i] + gi.dw) < (unsigned)a.IW);
      xo[i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (ok) xo[i] = *(const bf16x8*)(a.x + rbase[i] + gi.off);
    }
  };
  fetch(0, xf, wf);
  for (int ks = 0; ks < nsteps; ++ks) {
    if (ks + 1 < nsteps) fetch(ks + 1, xn, wn);
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[i], acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < FM; ++i) xf[i] = xn[i];
    wf = wn;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const long m = mw + i * 16 + fr;
    if (m >= a.M) continue;
    long t = m;
    const int rw = (int)(t % a.RW); t /= a.RW;
    const int rh = (int)(t % a.RH); t /= a.RH;
    const int rd = (int)(t % a.RD); t /= a.RD;
    bf16_t* yrow = a.y + t * a.ysn + (long)(rd * a.os[0] + a.oo[0]) * a.ysd + (long)(rh * a.os[1] + a.oo[1]) * a.ysh +
                   (long)(rw * a.os[2] + a.oo[2]) * a.ysw;
    const int c = fg * 4;
    if (c >= a.cout_lanes) continue;
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = 0.f;
      if (c + r < a.cout) { v = acc[i][r] + (a.bias ? a.bias[c + r] : 0.f); v = act_fwd(v, a.act, a.slope); }
      o[r] = (short)f2bf(v);
    }
    *(bf16x4*)(yrow + c) = o;
  }
}

template <int BM, int FN>
static void launch_igemm(const IgemmArgs& a, hipStream_t s) {
  constexpr int BN = FN * 16;
  dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)((a.cout_lanes + BN - 1) / BN));
  rg_set_kernel_name("igemm_kernel<%d,%d>", BM, FN);
  hipLaunchKernelGGL((igemm_kernel<BM, FN>), grid, dim3(256), 0, s, a);
}

extern "C" int rg_conv3d_kalign(int cin, int cout);
int rg_launch_igemm2(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p,
                     const void* zeros, hipStream_t st);
int rg_launch_igemm3(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p,
                     const void* zeros, hipStream_t st);
bool rg_igemm3_ok(const CL5& X, const rg_conv_params* p);
bool rg_igemm4_ok(const CL5& X, const rg_conv_params* p);
int rg_igemm4_tile_n(int cout);
int rg_launch_igemm4(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p, hipStream_t st, int flat);
__device__ __attribute__((aligned(128))) unsigned short rg_zero_page[64];  // zero-initialised with the code object
static const void* zero_page() {
  static const void* cache[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!cache[dev]) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(rg_zero_page)) != hipSuccess) return nullptr; cache[dev] = p; }
  return cache[dev];
}

extern "C" int rg_conv3d_igemm(const rg_tensor* x, const void* w_packed, const float* bias, const rg_tensor* y,
                               const rg_conv_params* p, rg_stream s) {
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
    if (rg_igemm4_ok(X, p) && !(i3 && rg_igemm4_tile_n(p->cout) == 8)) { if ((rc = rg_launch_igemm4(X, Y, w_packed, bias, p, (hipStream_t)s, 0))) return rc; }
    else if (i3) rg_launch_igemm3(X, Y, w_packed, bias, p, z, (hipStream_t)s);
    else rg_launch_igemm2(X, Y, w_packed, bias, p, z, (hipStream_t)s);
    RG_CHECK_LAUNCH();
    return RG_OK;
  }
  if (rg_conv3d_kalign(X.C, p->cout) == 1) {  // K flattened across taps, activations straight into fragments
    if ((rc = rg_launch_igemm4(X, Y, w_packed, bias, p, (hipStream_t)s, 1))) return rc;
    RG_CHECK_LAUNCH();
    return RG_OK;
  }
  if (rg_conv3d_kalign(X.C, p->cout) == 8) {  // thin output: direct-fragment kernel (packed K = taps*lanes, no padding)
    IgemmArgs a;
    a.x = X.p; a.w = (const bf16_t*)w_packed; a.bias = bias; a.y = Y.p;
    a.N = X.N; a.ID = X.D; a.IH = X.H; a.IW = X.W;
    a.RD = p->R[0]; a.RH = p->R[1]; a.RW = p->R[2];
    a.TD = p->T[0]; a.TH = p->T[1]; a.TW = p->T[2];
    bool check = false;
    const int I[3] = {X.D, X.H, X.W};
    for (int d = 0; d < 3; ++d) {
      a.rs[d] = p->rstride[d]; a.tb[d] = p->tbase[d]; a.ts[d] = p->tstep[d]; a.os[d] = p->ostride[d]; a.oo[d] = p->ooffset[d];
      const long lo0 = p->tbase[d], lo1 = p->tbase[d] + (long)p->tstep[d] * (p->T[d] - 1);
      const long mn = lo0 < lo1 ? lo0 : lo1, mx = (lo0 < lo1 ? lo1 : lo0) + (long)(p->R[d] - 1) * p->rstride[d];
      if (mn < 0 || mx >= I[d]) check = true;
    }
    a.xsn = X.sn; a.xsd = X.sd; a.xsh = X.sh; a.xsw = X.sw;
    a.ysn = Y.sn; a.ysd = Y.sd; a.ysh = Y.sh; a.ysw = Y.sw;
    a.cin32 = X.lanes; a.cin_lanes = X.lanes; a.cout = p->cout; a.cout_lanes = Y.lanes; a.wrows = 16;
    a.act = p->act; a.slope = p->slope;
    a.M = (long)X.N * a.RD * a.RH * a.RW;
    const int ngroups = a.TD * a.TH * a.TW * (X.lanes / 8);
    const size_t shm = (size_t)ngroups * sizeof(GroupInfo);
    if (shm > 64 * 1024) RG_FAIL(RG_ERR_UNSUPPORTED, "conv: tap table of %d groups does not fit LDS", ngroups);
    if ((long)X.N * X.sn >= 0x7fffffffL) RG_FAIL(RG_ERR_UNSUPPORTED, "conv: input larger than 2^31 elements");
    dim3 grid((unsigned)((a.M + 255) / 256));
    rg_set_kernel_name("igemm_direct_kernel<4,%d>", (int)check);
    if (check) hipLaunchKernelGGL((igemm_direct_kernel<4, true>), grid, dim3(256), shm, (hipStream_t)s, a);
    else hipLaunchKernelGGL((igemm_direct_kernel<4, false>), grid, dim3(256), shm, (hipStream_t)s, a);
    RG_CHECK_LAUNCH();
    return RG_OK;
  }
  IgemmArgs a;
  a.x = X.p; a.w = (const bf16_t*)w_packed; a.bias = bias; a.y = Y.p;
  a.N = X.N; a.ID = X.D; a.IH = X.H; a.IW = X.W;
  a.RD = p->R[0]; a.RH = p->R[1]; a.RW = p->R[2];
  a.TD = p->T[0]; a.TH = p->T[1]; a.TW = p->T[2];
  for (int d = 0; d < 3; ++d) { a.rs[d] = p->rstride[d]; a.tb[d] = p->tbase[d]; a.ts[d] = p->tstep[d]; a.os[d] = p->ostride[d]; a.oo[d] = p->ooffset[d]; }
  a.xsn = X.sn; a.xsd = X.sd; a.xsh = X.sh; a.xsw = X.sw;
  a.ysn = Y.sn; a.ysd = Y.sd; a.ysh = Y.sh; a.ysw = Y.sw;
  a.cin32 = round_up(X.C, 32); a.cin_lanes = X.lanes;
  a.cout = p->cout; a.cout_lanes = Y.lanes; a.wrows = round_up(p->cout, 16);
  a.act = p->act; a.slope = p->slope;
  a.M = (long)X.N * a.RD * a.RH * a.RW;
  hipStream_t st = (hipStream_t)s;
  const int nf = a.wrows / 16;  // 16-channel fragments needed
  // tile choice: BN = 16*FN covering the channels with the least padding; rows 256 when there is enough work
  const bool big = a.M >= 256 * 512;
  if (nf == 1) { big ? launch_igemm<256, 1>(a, st) : launch_igemm<128, 1>(a, st); }
  else if (nf == 2) { big ? launch_igemm<256, 2>(a, st) : launch_igemm<128, 2>(a, st); }
  else if (nf == 3) { big ? launch_igemm<256, 3>(a, st) : launch_igemm<128, 3>(a, st); }
  else if (nf == 4) { big ? launch_igemm<256, 4>(a, st) : launch_igemm<128, 4>(a, st); }
  else if (nf == 5) { big ? launch_igemm<256, 5>(a, st) : launch_igemm<128, 5>(a, st); }
  else if (nf == 6) { big ? launch_igemm<256, 6>(a, st) : launch_igemm<128, 6>(a, st); }
  else if (nf % 9 == 0) { big ? launch_igemm<256, 9>(a, st) : launch_igemm<128, 9>(a, st); }
  else if (nf % 8 == 0 || nf > 16) { big ? launch_igemm<256, 8>(a, st) : launch_igemm<128, 8>(a, st); }
  else if (nf % 7 == 0) { big ? launch_igemm<256, 7>(a, st) : launch_igemm<128, 7>(a, st); }
  else if (nf % 5 == 0) { big ? launch_igemm<256, 5>(a, st) : launch_igemm<128, 5>(a, st); }
  else { big ? launch_igemm<256, 8>(a, st) : launch_igemm<128, 8>(a, st); }
  RG_CHECK_LAUNCH();
  return RG_OK;
}
