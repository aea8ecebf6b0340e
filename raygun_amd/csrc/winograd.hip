// Winograd F(2x2x2, 3x3x3) for the widest stride-1 3^3 convolutions (config U's 2592 -> 2592 level: 17^3 outputs, K = 27 * 2592).
//
// y = A^T [ sum_c (G g G^T)(k,c) . (B^T d B)(c) ] A per 2x2x2 output tile: 64 multiplications per tile and channel pair instead
// of 8 * 27 = 216, i.e. 64 batched GEMMs [tiles x Cin] x [Cin x Cout] (one per transform position) that run on the
// fragment-direct implicit-GEMM kernel of conv4.hip in its batched form (rg_gemm_batched), framed by three streaming passes:
//   rg_wino_input    d -> B^T d B, channels-last activations -> V[64][tiles][lanes]          (adds only, fp32, one rounding)
//   rg_wino_weights  g -> G g G^T, fp32 master weights -> U[64][Cout][Cin] and, for the backward-data pass, the same values
//                    as Ut[s(pos)][Cin][Cout]: the flipped kernel's transform is the original one with positions 0 <-> 3
//                    exchanged along every axis (G flip(g) = (Gg)[3,1,2,0]), so one pass over the weights serves both
//   rg_wino_output   M -> A^T M A + bias, activation, channels-last store (crops the odd last tile)
// Only worth it where K = Cin is long: the transform-domain tensors are 8x the activation size, and a GEMM with a handful of
// K-steps is all prologue (measured estimates for the 432-channel levels in DESIGN.md: no gain).
// Numerics: operands rounded once more in the transform domain; measured 2x the direct kernel's bf16 error on that layer.
#include "common.h"

struct WinoIO {
  CL5 T;                  // activation tensor (input of the input transform / output of the output transform)
  bf16_t* buf;            // V or M: [64][ntiles][cs]
  long cs, ps;            // channel (row) stride and position stride of buf, elements
  int TD, TH, TW;         // tiles per axis
  int o0, o1, o2;         // tensor coordinate of tile 0's first element (input transform: 0, or -2 for the full correlation)
  long ntiles;            // N * TD * TH * TW
  int pairs;              // channel pairs per tile handled by the grid
  const float* bias; int cout; int act; float slope;  // output transform only
};

__device__ __forceinline__ void wino_unpack(unsigned u, float& a, float& b) { a = bf2f((bf16_t)(u & 0xffffu)); b = bf2f((bf16_t)(u >> 16)); }
__device__ __forceinline__ unsigned wino_pack(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

// B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1] along one axis of a 4-vector
#define WINO_BT(v0, v1, v2, v3) do { const float t0_ = v0 - v2, t1_ = v1 + v2, t2_ = v2 - v1, t3_ = v1 - v3; v0 = t0_; v1 = t1_; v2 = t2_; v3 = t3_; } while (0)

__global__ __launch_bounds__(256) void wino_in_kernel(WinoIO a) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const int pr = (int)(gid % a.pairs);
  long tile = gid / a.pairs;
  if (tile >= a.ntiles) return;
  const long tile0 = tile;
  const int tw = (int)(tile % a.TW); tile /= a.TW;
  const int th = (int)(tile % a.TH); tile /= a.TH;
  const int td = (int)(tile % a.TD); const int n = (int)(tile / a.TD);
  const int c = pr * 2;
  const bool live = c < a.T.lanes;
  const int d0 = a.o0 + 2 * td, h0 = a.o1 + 2 * th, w0 = a.o2 + 2 * tw;
  const bf16_t* base = a.T.p + (long)n * a.T.sn + c;
  float v[4][4][4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = d0 + i, h = h0 + j, w = w0 + k;
        const bool in = live && (unsigned)d < (unsigned)a.T.D && (unsigned)h < (unsigned)a.T.H && (unsigned)w < (unsigned)a.T.W;
        unsigned u = 0;
        if (in) u = *(const unsigned*)(base + (long)d * a.T.sd + (long)h * a.T.sh + (long)w * a.T.sw);
        wino_unpack(u, v[i][j][k][0], v[i][j][k][1]);
      }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) WINO_BT(v[i][j][0][e], v[i][j][1][e], v[i][j][2][e], v[i][j][3][e]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) WINO_BT(v[i][0][k][e], v[i][1][k][e], v[i][2][k][e], v[i][3][k][e]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) WINO_BT(v[0][j][k][e], v[1][j][k][e], v[2][j][k][e], v[3][j][k][e]);
  }
  bf16_t* out = a.buf + tile0 * a.cs + c;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *(unsigned*)(out + (long)((i * 4 + j) * 4 + k) * a.ps) = wino_pack(v[i][j][k][0], v[i][j][k][1]);  // lanes past the tensor's: zeros
}

// A^T = [1 1 1 0; 0 1 -1 -1]
#define WINO_AT(o0, o1, m0, m1, m2, m3) do { o0 = m0 + m1 + m2; o1 = m1 - m2 - m3; } while (0)

__global__ __launch_bounds__(256) void wino_out_kernel(WinoIO a) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const int pr = (int)(gid % a.pairs);
  long tile = gid / a.pairs;
  if (tile >= a.ntiles) return;
  const long tile0 = tile;
  const int tw = (int)(tile % a.TW); tile /= a.TW;
  const int th = (int)(tile % a.TH); tile /= a.TH;
  const int td = (int)(tile % a.TD); const int n = (int)(tile / a.TD);
  const int c = pr * 2;
  const bf16_t* in = a.buf + tile0 * a.cs + c;
  float y[2][2][2][2];
  float m[4][4][4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) wino_unpack(*(const unsigned*)(in + (long)((i * 4 + j) * 4 + k) * a.ps), m[i][j][k][0], m[i][j][k][1]);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    float s1[4][4][2];  // after the w axis
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) WINO_AT(s1[i][j][0], s1[i][j][1], m[i][j][0][e], m[i][j][1][e], m[i][j][2][e], m[i][j][3][e]);
    float s2[4][2][2];  // after the h axis
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) WINO_AT(s2[i][0][k], s2[i][1][k], s1[i][0][k], s1[i][1][k], s1[i][2][k], s1[i][3][k]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k) WINO_AT(y[0][j][k][e], y[1][j][k][e], s2[0][j][k], s2[1][j][k], s2[2][j][k], s2[3][j][k]);
  }
  const float b0 = (a.bias && c < a.cout) ? a.bias[c] : 0.f, b1 = (a.bias && c + 1 < a.cout) ? a.bias[c + 1] : 0.f;
  bf16_t* base = a.T.p + (long)n * a.T.sn + c;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int d = a.o0 + 2 * td + i, h = a.o1 + 2 * th + j, w = a.o2 + 2 * tw + k;
        if (d >= a.T.D || h >= a.T.H || w >= a.T.W) continue;
        const float r0 = c < a.cout ? act_fwd(y[i][j][k][0] + b0, a.act, a.slope) : 0.f;
        const float r1 = c + 1 < a.cout ? act_fwd(y[i][j][k][1] + b1, a.act, a.slope) : 0.f;
        *(unsigned*)(base + (long)d * a.T.sd + (long)h * a.T.sh + (long)w * a.T.sw) = wino_pack(r0, r1);
      }
}

// G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1] along one axis: 3 -> 4
#define WINO_G(o0, o1, o2, o3, g0, g1, g2) do { const float s_ = 0.5f * (g0 + g2), h_ = 0.5f * g1; o0 = g0; o1 = s_ + h_; o2 = s_ - h_; o3 = g2; } while (0)

struct WinoW {
  const float* w; long sk, sc, st;   // element strides of w[k][c][tap] (tap = (kd*3+kh)*3+kw)
  int cout, cin;
  bf16_t* U; int urows, ured;        // U[64][urows][ured]: rows = cout, contiguous = cin
  bf16_t* Ut; int trows, tred;       // Ut[64][trows][tred]: rows = cin, contiguous = cout; position index mirrored (0 <-> 3 per axis)
};

// One (k, c) per thread, 32 x 32 tile per 1024-thread workgroup; both operands leave through a 16-position LDS slab so that
// their rows are written in 16-byte pieces (U: c-contiguous, Ut: k-contiguous).
__global__ __launch_bounds__(1024) void wino_w_kernel(WinoW a) {
  constexpr int LS = 40;  // LDS row length (elements): 80 B rows keep the 16-byte reads aligned and spread the 2-byte writes
  __shared__ __attribute__((aligned(16))) bf16_t lt[16 * 32 * LS];
  const int tid = threadIdx.x, cl = tid & 31, kl = tid >> 5;
  const int c = blockIdx.x * 32 + cl, k = blockIdx.y * 32 + kl;
  const bool live = c < a.cin && k < a.cout;
  float g[3][3][3];
#pragma unroll
  for (int t = 0; t < 27; ++t) g[t / 9][(t / 3) % 3][t % 3] = live ? a.w[(long)k * a.sk + (long)c * a.sc + (long)t * a.st] : 0.f;
  float u1[3][3][4], u2[3][4][4], u[4][4][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) WINO_G(u1[i][j][0], u1[i][j][1], u1[i][j][2], u1[i][j][3], g[i][j][0], g[i][j][1], g[i][j][2]);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) WINO_G(u2[i][0][kk], u2[i][1][kk], u2[i][2][kk], u2[i][3][kk], u1[i][0][kk], u1[i][1][kk], u1[i][2][kk]);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) WINO_G(u[0][j][kk], u[1][j][kk], u[2][j][kk], u[3][j][kk], u2[0][j][kk], u2[1][j][kk], u2[2][j][kk]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // forward operand: slab of the 16 positions (i, *, *) as [pos][k][c], written out in 16-byte pieces of 8 c
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) lt[((j * 4 + kk) * 32 + kl) * LS + cl] = f2bf(u[i][j][kk]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int item = tid + 1024 * q;            // 16 positions x 32 k x 4 pieces of 8 c
      const int piece = item & 3, kr = (item >> 2) & 31, pp = item >> 7;
      const int krow = blockIdx.y * 32 + kr, c0 = blockIdx.x * 32 + piece * 8;
      if (krow < a.urows && c0 < a.ured)
        *(bf16x8*)(a.U + ((long)(i * 16 + pp) * a.urows + krow) * a.ured + c0) = *(const bf16x8*)(lt + (pp * 32 + kr) * LS + piece * 8);
    }
    // transposed operand: the same slab as [pos][c][k], position index mirrored
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) lt[((j * 4 + kk) * 32 + cl) * LS + kl] = f2bf(u[i][j][kk]);
    __syncthreads();
    const int si = i == 0 ? 3 : (i == 3 ? 0 : i);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int item = tid + 1024 * q;            // 16 positions x 32 c x 4 pieces of 8 k
      const int piece = item & 3, cc = (item >> 2) & 31, pp = item >> 7;
      const int j = pp >> 2, kk = pp & 3;
      const int sj = j == 0 ? 3 : (j == 3 ? 0 : j), sk_ = kk == 0 ? 3 : (kk == 3 ? 0 : kk);
      const int crow = blockIdx.x * 32 + cc, k0 = blockIdx.y * 32 + piece * 8;
      if (crow < a.trows && k0 < a.tred)
        *(bf16x8*)(a.Ut + ((long)((si * 4 + sj) * 4 + sk_) * a.trows + crow) * a.tred + k0) = *(const bf16x8*)(lt + (pp * 32 + cc) * LS + piece * 8);
    }
  }
}

static int wino_io(const rg_tensor* t, CL5* T, const char* name) { return rg_as_cl5(t, T, name); }

extern "C" int rg_wino_input(const rg_tensor* x, void* V, int64_t v_cstride, const int32_t tiles[3], const int32_t origin[3], rg_stream stream) {
  WinoIO a; int rc;
  if ((rc = wino_io(x, &a.T, "wino_input.x"))) return rc;
  if (!V || !tiles || !origin || v_cstride < a.T.lanes || (v_cstride & 1)) RG_FAIL(RG_ERR_BAD_SHAPE, "wino_input: bad V / tiles / origin / stride");
  a.buf = (bf16_t*)V; a.cs = v_cstride;
  a.TD = tiles[0]; a.TH = tiles[1]; a.TW = tiles[2];
  a.o0 = origin[0]; a.o1 = origin[1]; a.o2 = origin[2];
  a.ntiles = (long)a.T.N * a.TD * a.TH * a.TW; a.ps = a.ntiles * a.cs;
  a.pairs = (int)(v_cstride / 2);
  a.bias = nullptr; a.cout = 0; a.act = 0; a.slope = 0.f;
  const long total = a.ntiles * a.pairs;
  if (total <= 0) return RG_OK;
  rg_set_kernel_name("wino_in_kernel");
  hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

extern "C" int rg_wino_output(const void* M, int64_t m_cstride, const int32_t tiles[3], rg_tensor* y, const float* bias, int32_t cout,
                              int32_t act, float slope, rg_stream stream) {
  WinoIO a; int rc;
  if ((rc = wino_io(y, &a.T, "wino_output.y"))) return rc;
  if (!M || !tiles || m_cstride < a.T.lanes || (m_cstride & 1)) RG_FAIL(RG_ERR_BAD_SHAPE, "wino_output: bad M / tiles / stride");
  a.buf = (bf16_t*)M; a.cs = m_cstride;
  a.TD = tiles[0]; a.TH = tiles[1]; a.TW = tiles[2];
  if (2 * a.TD < a.T.D || 2 * a.TH < a.T.H || 2 * a.TW < a.T.W) RG_FAIL(RG_ERR_BAD_SHAPE, "wino_output: the tiles do not cover y");
  a.o0 = a.o1 = a.o2 = 0;
  a.ntiles = (long)a.T.N * a.TD * a.TH * a.TW; a.ps = a.ntiles * a.cs;
  a.pairs = a.T.lanes / 2;
  a.bias = bias; a.cout = cout; a.act = act; a.slope = slope;
  const long total = a.ntiles * a.pairs;
  if (total <= 0) return RG_OK;
  rg_set_kernel_name("wino_out_kernel");
  hipLaunchKernelGGL(wino_out_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

extern "C" int64_t rg_wino_weights_rows(int32_t channels) { return round_up(channels, 32); }       // rows of U (cout) / Ut (cin)
extern "C" int64_t rg_wino_weights_red(int32_t channels) { return round_up(channels, 64); }        // contiguous length of U (cin) / Ut (cout)

extern "C" int rg_wino_weights(const void* w_f32, int32_t channels_last, int32_t cout, int32_t cin, void* U, void* Ut, rg_stream stream) {
  if (!w_f32 || !U || !Ut || cout <= 0 || cin <= 0) RG_FAIL(RG_ERR_BAD_SHAPE, "wino_weights: null pointer or empty weight");
  WinoW a;
  a.w = (const float*)w_f32; a.cout = cout; a.cin = cin;
  if (channels_last) { a.sk = 27L * cin; a.sc = 1; a.st = cin; } else { a.sk = 27L * cin; a.sc = 27; a.st = 1; }
  a.U = (bf16_t*)U; a.urows = (int)rg_wino_weights_rows(cout); a.ured = (int)rg_wino_weights_red(cin);
  a.Ut = (bf16_t*)Ut; a.trows = (int)rg_wino_weights_rows(cin); a.tred = (int)rg_wino_weights_red(cout);
  // the grid covers both padded extents (pad entries are written as zeros: the GEMM multiplies them)
  const int gx = (a.ured > a.trows ? a.ured : a.trows) / 32, gy = (a.urows > a.tred ? a.urows : a.tred) / 32;
  rg_set_kernel_name("wino_w_kernel");
  hipLaunchKernelGGL(wino_w_kernel, dim3(gx, gy), dim3(1024), 0, (hipStream_t)stream, a);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
