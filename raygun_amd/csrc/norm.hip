// InstanceNorm3d (affine=False, no running stats) forward/backward with fused activation and
// residual add, reflection/zero padding, and the fused loss value+gradient reductions.
// Statistics are fp32 sums per workgroup (one workgroup never straddles two samples), combined in
// double in a fixed order -> bitwise reproducible.
#include "common.h"
#include <stdlib.h>

static inline int grid_for(long n, int block = 256, int cap = 1 << 20) {
  long g = (n + block - 1) / block; if (g > cap) g = cap; if (g < 1) g = 1; return (int)g;
}

struct NormPlan { int gx, gy, vpb, nblk; };
static NormPlan norm_plan(const CL5& t) {
  NormPlan p;
  const int G8 = t.lanes / 8;
  int x = 1; while (x < G8 && x < 32) x <<= 1;
  p.gx = x; p.gy = 256 / x;
  const long nvox = (long)t.D * t.H * t.W;
  // workgroups per sample: 512 (two per CU).  Measured on the CycleGAN step (256 x 32^3 tensors): 2048 -> 97.8 ms, 1024 -> 97.7, 512 -> 96.5, 256 -> 99.7
  long want = 512 / (t.N > 0 ? t.N : 1); if (want < 8) want = 8;
  long v = (nvox + want - 1) / want; if (v < p.gy * 4) v = p.gy * 4;
  p.vpb = (int)v; p.nblk = (int)((nvox + v - 1) / v);
  return p;
}
extern "C" int64_t rg_instnorm_workspace_bytes(const rg_tensor* x) {
  CL5 t; if (rg_as_cl5(x, &t, "instnorm.x")) return -1;
  NormPlan p = norm_plan(t);
  return (int64_t)t.N * p.nblk * t.lanes * 2 * 4;
}

// Thread (tx, ty) of block (bx, by, n): channel group by*GX + tx (its 8 (mean, rstd) pairs are loaded ONCE), voxels
// bx*vpb + ty, + GY, ...  Voxel offsets are v * voxel stride when every tensor is dense in D,H,W (no per-element
// divisions); the strided path decodes (d,h,w) with 32-bit divisions.
__device__ __forceinline__ long vox_off(const CL5& t, int n, unsigned v, bool dense) {
  if (dense) return n * t.sn + (long)v * t.sw;
  const unsigned w = v % (unsigned)t.W; v /= (unsigned)t.W; const unsigned h = v % (unsigned)t.H; const unsigned d = v / (unsigned)t.H;
  return n * t.sn + d * t.sd + h * t.sh + w * t.sw;
}
static bool cl_dense(const CL5& t) { return t.sh == (long)t.W * t.sw && t.sd == (long)t.H * t.sh; }
// MODE 0: s0 = sum x, s1 = sum x^2.   MODE 1 (backward): g = gy*act'(y); s0 = sum g, s1 = sum g*xhat
template <int MODE>
__global__ void norm_reduce_kernel(CL5 x, CL5 y, CL5 gy, const float* mr, float* partial, int vpb, int act, float slope, int dense) {
  extern __shared__ float red[];  // [GY][GX][16]
  const int GX = blockDim.x, GY = blockDim.y;
  const int grp = blockIdx.y * GX + threadIdx.x;
  const int n = blockIdx.z;
  const long nvox = (long)x.D * x.H * x.W;
  const long v0 = (long)blockIdx.x * vpb;
  long v1 = v0 + vpb; if (v1 > nvox) v1 = nvox;
  float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool active = grp * 8 < x.lanes;
  float mean[8], rstd[8];
  if (MODE == 1 && active) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = grp * 8 + k;
      mean[k] = c < x.C ? mr[((long)n * x.C + c) * 2] : 0.f; rstd[k] = c < x.C ? mr[((long)n * x.C + c) * 2 + 1] : 0.f;
    }
  }
  if (active) {
    // NU voxels in flight per thread: all their loads are issued before the first use (the streams are latency-bound otherwise)
    constexpr int NU = 4;
    for (long vb = v0 + threadIdx.y; vb < v1; vb += (long)GY * NU) {
      bf16x8 xv[NU], gv[NU], yv[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const long v = vb + (long)u * GY;
        const bool in = v < v1;
        xv[u] = gv[u] = yv[u] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (in) {
          xv[u] = *(const bf16x8*)(x.p + vox_off(x, n, (unsigned)v, dense) + grp * 8);
          if (MODE == 1) {
            gv[u] = *(const bf16x8*)(gy.p + vox_off(gy, n, (unsigned)v, dense) + grp * 8);
            if (act != RG_ACT_NONE) yv[u] = *(const bf16x8*)(y.p + vox_off(y, n, (unsigned)v, dense) + grp * 8);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (MODE == 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) { const float f = bf2f((bf16_t)xv[u][k]); s0[k] += f; s1[k] += f * f; }  // out-of-range voxels were zero-filled
        } else if (vb + (long)u * GY < v1) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float g = bf2f((bf16_t)gv[u][k]) * act_bwd_from_y(bf2f((bf16_t)yv[u][k]), act, slope);
            const float xh = (bf2f((bf16_t)xv[u][k]) - mean[k]) * rstd[k];
            s0[k] += g; s1[k] += g * xh;
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[((threadIdx.y * GX + threadIdx.x) * 2) * 8 + k] = s0[k]; red[((threadIdx.y * GX + threadIdx.x) * 2 + 1) * 8 + k] = s1[k]; }
  __syncthreads();
  if (threadIdx.y == 0 && active) {
    for (int yy = 1; yy < GY; ++yy)
#pragma unroll
      for (int k = 0; k < 8; ++k) { s0[k] += red[((yy * GX + threadIdx.x) * 2) * 8 + k]; s1[k] += red[((yy * GX + threadIdx.x) * 2 + 1) * 8 + k]; }
    float* o = partial + (((long)n * gridDim.x + blockIdx.x) * x.lanes + grp * 8) * 2;
#pragma unroll
    for (int k = 0; k < 8; ++k) { o[k * 2] = s0[k]; o[k * 2 + 1] = s1[k]; }
  }
}
// MODE 0: out = (mean, rstd).  MODE 1: out = (mean_g, mean_gxhat).  One 64-lane wave per (n,c): strided partial sums in
// double, fixed-order shuffle tree (deterministic).
template <int MODE>
__global__ void norm_finalize_kernel(const float* partial, float* out, int N, int C, int lanes, int nblk, long nvox, float eps) {
  const int i = blockIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i % C;
  double a = 0, b = 0;
  for (int k = threadIdx.x; k < nblk; k += 64) { const float* p = partial + (((long)n * nblk + k) * lanes + c) * 2; a += p[0]; b += p[1]; }
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
  if (threadIdx.x != 0) return;
  if (MODE == 0) {
    const double mean = a / nvox; double var = b / nvox - mean * mean; if (var < 0) var = 0;
    out[i * 2] = (float)mean; out[i * 2 + 1] = (float)(1.0 / sqrt(var + eps));
  } else { out[i * 2] = (float)(a / nvox); out[i * 2 + 1] = (float)(b / nvox); }
}

extern "C" int rg_instnorm_stats(const rg_tensor* x, float* mean_rstd, float eps, void* workspace, size_t ws_bytes, rg_stream s) {
  CL5 X; int rc;
  if ((rc = rg_as_cl5(x, &X, "instnorm.x"))) return rc;
  NormPlan p = norm_plan(X);
  const size_t need = (size_t)X.N * p.nblk * X.lanes * 2 * 4;
  if (!workspace || ws_bytes < need || !mean_rstd) RG_FAIL(RG_ERR_WORKSPACE, "instnorm_stats: workspace %zu < %zu", ws_bytes, need);
  dim3 grid(p.nblk, (X.lanes / 8 + p.gx - 1) / p.gx, X.N), block(p.gx, p.gy);
  hipLaunchKernelGGL((norm_reduce_kernel<0>), grid, block, 256 * 16 * sizeof(float), (hipStream_t)s, X, X, X, (const float*)nullptr, (float*)workspace, p.vpb, 0, 0.f, (int)cl_dense(X));
  RG_CHECK_LAUNCH();
  hipLaunchKernelGGL((norm_finalize_kernel<0>), dim3(X.N * X.C), dim3(64), 0, (hipStream_t)s, (const float*)workspace, mean_rstd, X.N, X.C, X.lanes, p.nblk,
                     (long)X.D * X.H * X.W, eps);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// `ab` (optional): per-channel (gamma, beta) of an affine norm: y = act(xhat * gamma + beta) [+ res]
__global__ void norm_apply_kernel(CL5 x, CL5 res, CL5 y, const float* mr, const float* ab, int act, float slope, int has_res, int vpb, int dense) {
  const int grp = blockIdx.y * blockDim.x + threadIdx.x, n = blockIdx.z;
  if (grp * 8 >= y.lanes) return;
  float m[8], r[8], be[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = grp * 8 + k;
    m[k] = c < x.C ? mr[((long)n * x.C + c) * 2] : 0.f; r[k] = c < x.C ? mr[((long)n * x.C + c) * 2 + 1] : 0.f;
    be[k] = 0.f;
    if (ab && c < x.C) { r[k] *= ab[2 * c]; be[k] = ab[2 * c + 1]; }
  }
  const bool lin = act <= RG_ACT_LEAKY;
  const float nslope = rg_act_nslope(act, slope);
  const unsigned nvox = (unsigned)(y.D * y.H * y.W);
  const unsigned v0 = blockIdx.x * (unsigned)vpb;
  unsigned v1 = v0 + (unsigned)vpb; if (v1 > nvox) v1 = nvox;
  constexpr int NU = 4;  // voxels in flight per thread
  for (unsigned vb = v0 + threadIdx.y; vb < v1; vb += blockDim.y * NU) {
    bf16x8 xv[NU], rv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const unsigned v = vb + u * blockDim.y;
      xv[u] = rv[u] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (v < v1) {
        xv[u] = *(const bf16x8*)(x.p + vox_off(x, n, v, dense) + grp * 8);
        if (has_res) rv[u] = *(const bf16x8*)(res.p + vox_off(res, n, v, dense) + grp * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const unsigned v = vb + u * blockDim.y;
      if (v >= v1) break;
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float val = 0.f;
        {   // (the piecewise-linear activations as multiply + select; a generic act_fwd under the per-lane channel test predicates 32 inlined tanh bodies)
          const float xh = (bf2f((bf16_t)xv[u][k]) - m[k]) * r[k] + be[k];
          const float av = lin ? (xh < 0.f ? xh * nslope : xh) : act_fwd(xh, act, slope);
          val = grp * 8 + k < x.C ? av + bf2f((bf16_t)rv[u][k]) : 0.f;
        }
        o[k] = (short)f2bf(val);
      }
      *(bf16x8*)(y.p + vox_off(y, n, v, dense) + grp * 8) = o;
    }
  }
}
// launch shape shared by apply / bwd: ~2048 workgroups of (gx channel groups) x (256/gx voxels in flight)
static void norm_ew_plan(const CL5& t, dim3* grid, dim3* block, int* vpb) {
  const int G8 = t.lanes / 8;
  int gx = 1; while (gx < G8 && gx < 32) gx <<= 1;
  const int gy = 256 / gx, ngy = (G8 + gx - 1) / gx;
  const long nvox = (long)t.D * t.H * t.W;
  long want = 2048 / ((long)ngy * t.N); if (want < 1) want = 1;
  long v = (nvox + want - 1) / want; if (v < gy * 4) v = gy * 4;
  *vpb = (int)v; *grid = dim3((unsigned)((nvox + v - 1) / v), ngy, t.N); *block = dim3(gx, gy);
}
static bool same_shape(const CL5& a, const CL5& b) { return a.N == b.N && a.C == b.C && a.D == b.D && a.H == b.H && a.W == b.W; }

extern "C" int rg_instnorm_apply(const rg_tensor* x, const float* mean_rstd, const rg_tensor* res, const rg_tensor* y,
                                 int act, float slope, rg_stream s) {
  return rg_norm_apply_affine(x, mean_rstd, nullptr, res, y, act, slope, s);
}
extern "C" int rg_norm_apply_affine(const rg_tensor* x, const float* mean_rstd, const float* gamma_beta, const rg_tensor* res, const rg_tensor* y,
                                    int act, float slope, rg_stream s) {
  CL5 X, R, Y; int rc;
  if ((rc = rg_as_cl5(x, &X, "instnorm.x")) || (rc = rg_as_cl5(y, &Y, "instnorm.y"))) return rc;
  memset(&R, 0, sizeof(R));
  const int has_res = res && res->ptr;
  if (has_res) { if ((rc = rg_as_cl5(res, &R, "instnorm.res"))) return rc; if (!same_shape(R, X)) RG_FAIL(RG_ERR_BAD_SHAPE, "instnorm_apply: residual shape mismatch"); }
  if (!same_shape(X, Y) || !mean_rstd) RG_FAIL(RG_ERR_BAD_SHAPE, "instnorm_apply: shape mismatch / null stats");
  if ((long)Y.D * Y.H * Y.W >= (1L << 31)) RG_FAIL(RG_ERR_UNSUPPORTED, "instnorm_apply: more than 2^31 voxels per sample");
  dim3 grid, block; int vpb;
  norm_ew_plan(Y, &grid, &block, &vpb);
  const int dense = cl_dense(X) && cl_dense(Y) && (!has_res || cl_dense(R));
  hipLaunchKernelGGL(norm_apply_kernel, grid, block, 0, (hipStream_t)s, X, R, Y, mean_rstd, gamma_beta, act, slope, has_res, vpb, dense);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// `ab` (optional, affine norm): gx = gamma * rstd * (g - mean g - xhat * mean(g xhat)), g = gy * act'(y); the means `gm` are over the instance
// or, for batch statistics, already averaged over the batch
__global__ void norm_bwd_kernel(CL5 x, CL5 y, CL5 gy, CL5 gx, const float* mr, const float* gm, const float* ab, int act, float slope, int vpb, int dense) {
  const int grp = blockIdx.y * blockDim.x + threadIdx.x, n = blockIdx.z;
  if (grp * 8 >= gx.lanes) return;
  float m[8], r[8], g0[8], g1[8], ga[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = grp * 8 + k;
    const long q = ((long)n * x.C + (c < x.C ? c : 0)) * 2;
    m[k] = mr[q]; r[k] = mr[q + 1]; g0[k] = gm[q]; g1[k] = gm[q + 1];
    ga[k] = ab ? ab[2 * (c < x.C ? c : 0)] : 1.f;
  }
  const unsigned nvox = (unsigned)(gx.D * gx.H * gx.W);
  const unsigned v0 = blockIdx.x * (unsigned)vpb;
  unsigned v1 = v0 + (unsigned)vpb; if (v1 > nvox) v1 = nvox;
  constexpr int NU = 4;  // voxels in flight per thread
  for (unsigned vb = v0 + threadIdx.y; vb < v1; vb += blockDim.y * NU) {
    bf16x8 xv[NU], gv[NU], yv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const unsigned v = vb + u * blockDim.y;
      xv[u] = gv[u] = yv[u] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (v < v1) {
        xv[u] = *(const bf16x8*)(x.p + vox_off(x, n, v, dense) + grp * 8);
        gv[u] = *(const bf16x8*)(gy.p + vox_off(gy, n, v, dense) + grp * 8);
        if (act != RG_ACT_NONE) yv[u] = *(const bf16x8*)(y.p + vox_off(y, n, v, dense) + grp * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const unsigned v = vb + u * blockDim.y;
      if (v >= v1) break;
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float val = 0.f;
        if (grp * 8 + k < x.C) {
          const float g = bf2f((bf16_t)gv[u][k]) * act_bwd_from_y(bf2f((bf16_t)yv[u][k]), act, slope);
          const float xh = (bf2f((bf16_t)xv[u][k]) - m[k]) * r[k];
          val = ga[k] * r[k] * (g - g0[k] - xh * g1[k]);
        }
        o[k] = (short)f2bf(val);
      }
      *(bf16x8*)(gx.p + vox_off(gx, n, v, dense) + grp * 8) = o;
    }
  }
}
// gm[n][c] <- mean over n (BatchNorm: the statistics, hence the gradient's two means, are over batch x voxels)
__global__ void norm_batch_avg_kernel(float* gm, int N, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0, b = 0;
  for (int n = 0; n < N; ++n) { a += gm[((long)n * C + c) * 2]; b += gm[((long)n * C + c) * 2 + 1]; }
  const float fa = (float)(a / N), fb = (float)(b / N);
  for (int n = 0; n < N; ++n) { gm[((long)n * C + c) * 2] = fa; gm[((long)n * C + c) * 2 + 1] = fb; }
}
extern "C" int rg_instnorm_bwd(const rg_tensor* x, const float* mean_rstd, const rg_tensor* y, const rg_tensor* gy,
                               const rg_tensor* gx, int act, float slope, void* workspace, size_t ws_bytes, rg_stream s) {
  return rg_norm_bwd_affine(x, mean_rstd, nullptr, 0, y, gy, gx, act, slope, workspace, ws_bytes, s);
}
extern "C" int rg_norm_bwd_affine(const rg_tensor* x, const float* mean_rstd, const float* gamma_beta, int batch_stats, const rg_tensor* y,
                                  const rg_tensor* gy, const rg_tensor* gx, int act, float slope, void* workspace, size_t ws_bytes, rg_stream s) {
  CL5 X, Y, GY, GX; int rc;
  if ((rc = rg_as_cl5(x, &X, "instnorm_bwd.x")) || (rc = rg_as_cl5(gy, &GY, "instnorm_bwd.gy")) || (rc = rg_as_cl5(gx, &GX, "instnorm_bwd.gx"))) return rc;
  memset(&Y, 0, sizeof(Y));
  if (act != RG_ACT_NONE) { if ((rc = rg_as_cl5(y, &Y, "instnorm_bwd.y"))) return rc; if (!same_shape(Y, X)) RG_FAIL(RG_ERR_BAD_SHAPE, "instnorm_bwd: y shape"); }
  if (!same_shape(X, GY) || !same_shape(X, GX) || !mean_rstd) RG_FAIL(RG_ERR_BAD_SHAPE, "instnorm_bwd: shape mismatch");
  NormPlan p = norm_plan(X);
  const size_t part = (size_t)X.N * p.nblk * X.lanes * 2 * 4, need = part + (size_t)X.N * X.C * 2 * 4;
  if (!workspace || ws_bytes < need) RG_FAIL(RG_ERR_WORKSPACE, "instnorm_bwd: workspace %zu < %zu (use rg_instnorm_workspace_bytes + N*C*8)", ws_bytes, need);
  float* gm = (float*)((char*)workspace + part);
  dim3 grid(p.nblk, (X.lanes / 8 + p.gx - 1) / p.gx, X.N), block(p.gx, p.gy);
  hipLaunchKernelGGL((norm_reduce_kernel<1>), grid, block, 256 * 16 * sizeof(float), (hipStream_t)s, X, Y, GY, mean_rstd, (float*)workspace, p.vpb, act, slope, (int)(cl_dense(X) && cl_dense(GY) && (act == RG_ACT_NONE || cl_dense(Y))));
  RG_CHECK_LAUNCH();
  hipLaunchKernelGGL((norm_finalize_kernel<1>), dim3(X.N * X.C), dim3(64), 0, (hipStream_t)s, (const float*)workspace, gm, X.N, X.C, X.lanes, p.nblk,
                     (long)X.D * X.H * X.W, 0.f);
  RG_CHECK_LAUNCH();
  if (batch_stats && X.N > 1) { hipLaunchKernelGGL(norm_batch_avg_kernel, dim3((X.C + 63) / 64), dim3(64), 0, (hipStream_t)s, gm, X.N, X.C); RG_CHECK_LAUNCH(); }
  if ((long)GX.D * GX.H * GX.W >= (1L << 31)) RG_FAIL(RG_ERR_UNSUPPORTED, "instnorm_bwd: more than 2^31 voxels per sample");
  dim3 egrid, eblock; int evpb;
  norm_ew_plan(GX, &egrid, &eblock, &evpb);
  const int dense = cl_dense(X) && cl_dense(GY) && cl_dense(GX) && (act == RG_ACT_NONE || cl_dense(Y));
  hipLaunchKernelGGL(norm_bwd_kernel, egrid, eblock, 0, (hipStream_t)s, X, Y, GY, GX, mean_rstd, (const float*)gm, gamma_beta, act, slope, evpb, dense);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---------------------------------------------------------------------------- padding
__device__ __forceinline__ int reflect_idx(int i, int n) { if (i < 0) i = -i; if (i >= n) i = 2 * (n - 1) - i; return i; }

__global__ void pad_fwd_kernel(CL5 x, CL5 y, int pd, int ph, int pw, int mode) {
  const int G8 = y.lanes / 8;
  const long total = (long)y.N * y.D * y.H * y.W * G8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grp = (int)(i % G8); long t = i / G8;
    const int w = (int)(t % y.W); t /= y.W; const int h = (int)(t % y.H); t /= y.H; const int d = (int)(t % y.D); const int n = (int)(t / y.D);
    int sd = d - pd, sh = h - ph, sw = w - pw;
    bf16x8 v = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    bool ok = true;
    if (mode == 1) { sd = reflect_idx(sd, x.D); sh = reflect_idx(sh, x.H); sw = reflect_idx(sw, x.W); }
    else if (mode == 2) { sd = min(max(sd, 0), x.D - 1); sh = min(max(sh, 0), x.H - 1); sw = min(max(sw, 0), x.W - 1); }   // replicate: clamp to the border voxel
    else ok = (unsigned)sd < (unsigned)x.D && (unsigned)sh < (unsigned)x.H && (unsigned)sw < (unsigned)x.W;
    if (ok) v = *(const bf16x8*)(x.p + n * x.sn + (long)sd * x.sd + (long)sh * x.sh + (long)sw * x.sw + grp * 8);
    *(bf16x8*)(y.p + n * y.sn + d * y.sd + h * y.sh + w * y.sw + grp * 8) = v;
  }
}
// candidates in the padded tensor that map onto source index i (reflect): i+p, p-i (1<=i<=p), 2(n-1)-i+p (n-1-p<=i<=n-2)
__device__ __forceinline__ int pad_sources(int i, int n, int p, int mode, int out[3]) {
  int k = 0; out[k++] = i + p;
  if (mode == 1) { if (i >= 1 && i <= p) out[k++] = p - i; if (i >= n - 1 - p && i <= n - 2) out[k++] = 2 * (n - 1) - i + p; }
  return k;
}
// replicate: padded index range [lo, hi] that maps onto source index i (the border voxels collect their whole pad strip)
__device__ __forceinline__ void pad_range_replicate(int i, int n, int p, int& lo, int& hi) { lo = i == 0 ? 0 : i + p; hi = i == n - 1 ? n - 1 + 2 * p : i + p; }
__global__ void pad_bwd_kernel(CL5 gy, CL5 gx, int pd, int ph, int pw, int mode) {
  const int G8 = gx.lanes / 8;
  const long total = (long)gx.N * gx.D * gx.H * gx.W * G8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grp = (int)(i % G8); long t = i / G8;
    const int w = (int)(t % gx.W); t /= gx.W; const int h = (int)(t % gx.H); t /= gx.H; const int d = (int)(t % gx.D); const int n = (int)(t / gx.D);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (mode == 2) {
      int d0, d1, h0, h1, w0, w1;
      pad_range_replicate(d, gx.D, pd, d0, d1); pad_range_replicate(h, gx.H, ph, h0, h1); pad_range_replicate(w, gx.W, pw, w0, w1);
      for (int a = d0; a <= d1; ++a) for (int b = h0; b <= h1; ++b) for (int c = w0; c <= w1; ++c) {
        const bf16x8 v = *(const bf16x8*)(gy.p + n * gy.sn + (long)a * gy.sd + (long)b * gy.sh + (long)c * gy.sw + grp * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += bf2f((bf16_t)v[k]);
      }
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (short)f2bf(acc[k]);
      *(bf16x8*)(gx.p + n * gx.sn + d * gx.sd + h * gx.sh + w * gx.sw + grp * 8) = o;
      continue;
    }
    int ds[3], hs[3], ws[3];
    const int nd = pad_sources(d, gx.D, pd, mode, ds), nh = pad_sources(h, gx.H, ph, mode, hs), nw = pad_sources(w, gx.W, pw, mode, ws);
    for (int a = 0; a < nd; ++a) for (int b = 0; b < nh; ++b) for (int c = 0; c < nw; ++c) {
      const bf16x8 v = *(const bf16x8*)(gy.p + n * gy.sn + (long)ds[a] * gy.sd + (long)hs[b] * gy.sh + (long)ws[c] * gy.sw + grp * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += bf2f((bf16_t)v[k]);
    }
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (short)f2bf(acc[k]);
    *(bf16x8*)(gx.p + n * gx.sn + d * gx.sd + h * gx.sh + w * gx.sw + grp * 8) = o;
  }
}
static int pad_check(const CL5& X, const CL5& Y, const int32_t pad[3], int mode, const char* what) {
  if (Y.N != X.N || Y.C != X.C || Y.D != X.D + 2 * pad[0] || Y.H != X.H + 2 * pad[1] || Y.W != X.W + 2 * pad[2]) RG_FAIL(RG_ERR_BAD_SHAPE, "%s: padded shape mismatch", what);
  if (pad[0] < 0 || pad[1] < 0 || pad[2] < 0) RG_FAIL(RG_ERR_BAD_SHAPE, "%s: negative pad", what);
  if (mode == 1 && (pad[0] >= X.D || pad[1] >= X.H || pad[2] >= X.W)) RG_FAIL(RG_ERR_BAD_SHAPE, "%s: reflect pad must be smaller than the input", what);
  if (mode != 0 && mode != 1 && mode != 2) RG_FAIL(RG_ERR_UNSUPPORTED, "%s: mode %d (0 zeros, 1 reflect, 2 replicate)", what, mode);
  return RG_OK;
}
extern "C" int rg_pad3d_fwd(const rg_tensor* x, const rg_tensor* y, const int32_t pad[3], int mode, rg_stream s) {
  CL5 X, Y; int rc;
  if ((rc = rg_as_cl5(x, &X, "pad.x")) || (rc = rg_as_cl5(y, &Y, "pad.y")) || (rc = pad_check(X, Y, pad, mode, "pad_fwd"))) return rc;
  const long total = (long)Y.N * Y.D * Y.H * Y.W * (Y.lanes / 8);
  hipLaunchKernelGGL(pad_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, X, Y, pad[0], pad[1], pad[2], mode);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
extern "C" int rg_pad3d_bwd(const rg_tensor* gy, const rg_tensor* gx, const int32_t pad[3], int mode, rg_stream s) {
  CL5 GY, GX; int rc;
  if ((rc = rg_as_cl5(gy, &GY, "pad.gy")) || (rc = rg_as_cl5(gx, &GX, "pad.gx")) || (rc = pad_check(GX, GY, pad, mode, "pad_bwd"))) return rc;
  const long total = (long)GX.N * GX.D * GX.H * GX.W * (GX.lanes / 8);
  hipLaunchKernelGGL(pad_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, GY, GX, pad[0], pad[1], pad[2], mode);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---------------------------------------------------------------------------- losses (flat, contiguous fp32 tensors)
// out layout: out[0..2] results, out[3 + 3*b ...] per-block partials (b < LOSS_BLOCKS)
#define LOSS_BLOCKS 512
__global__ void loss_kernel(int kind, const float* p, const float* t, const float* w, float ct, float gscale, float* grad, float* out, long n) {
  __shared__ float red[3][256];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float pv = p[i];
    if (kind == 0) { const float d = pv - ct; a0 += d * d; if (grad) grad[i] = gscale * 2.f * d; }
    else if (kind == 3) {   // BCEWithLogitsLoss against the constant target ct (GANLoss "vanilla"): max(x,0) - x ct + log(1 + exp(-|x|))
      const float e = expf(-fabsf(pv));
      a0 += fmaxf(pv, 0.f) - pv * ct + log1pf(e);
      if (grad) grad[i] = gscale * ((pv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e)) - ct);
    } else if (kind == 4) {   // ct * x (GANLoss "wgangp": -mean(D(x)) for real, +mean for fake)
      a0 += ct * pv;
      if (grad) grad[i] = gscale * ct;
    }
    else if (kind == 1) {
      const float d = pv - t[i], ad = fabsf(d);
      a0 += ad < 1.f ? 0.5f * d * d : ad - 0.5f;
      if (grad) grad[i] = gscale * (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f));
    } else {
      const float d = pv - t[i], wv = w[i], sc = wv * d * d;
      if (wv > 0.f) { a0 += sc; a1 += 1.f; }
      if (sc != 0.f) a2 += 1.f;
      if (grad) grad[i] = wv > 0.f ? gscale * 2.f * wv * d : 0.f;
    }
  }
  red[0][threadIdx.x] = a0; red[1][threadIdx.x] = a1; red[2][threadIdx.x] = a2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; red[2][threadIdx.x] += red[2][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[3 + 3 * blockIdx.x] = red[0][0]; out[4 + 3 * blockIdx.x] = red[1][0]; out[5 + 3 * blockIdx.x] = red[2][0]; }
}
__global__ void loss_final_kernel(float* out, int nblk) {
  if (threadIdx.x < 3) { double s = 0; for (int b = 0; b < nblk; ++b) s += out[3 + 3 * b + threadIdx.x]; out[threadIdx.x] = (float)s; }
}
extern "C" int rg_loss_fwd_bwd(int kind, const rg_tensor* pred, const rg_tensor* target, const rg_tensor* weight, float const_target,
                               float grad_scale, const rg_tensor* grad, float* out, rg_stream s) {
  if (!pred || !pred->ptr || !out || pred->dtype != RG_F32 || pred->rank != 1 || pred->strides[0] != 1) RG_FAIL(RG_ERR_BAD_SHAPE, "loss: pred must be flat contiguous f32");
  const long n = pred->sizes[0];
  if (n <= 0) RG_FAIL(RG_ERR_BAD_SHAPE, "loss: empty input");
  if (kind < 0 || kind > 4) RG_FAIL(RG_ERR_UNSUPPORTED, "loss: kind %d", kind);
  const float *t = nullptr, *w = nullptr; float* g = nullptr;
  if (kind == 1 || kind == 2) { if (!target || !target->ptr || target->dtype != RG_F32 || target->sizes[0] != n) RG_FAIL(RG_ERR_BAD_SHAPE, "loss: target"); t = (const float*)target->ptr; }
  if (kind == 2) { if (!weight || !weight->ptr || weight->dtype != RG_F32 || weight->sizes[0] != n) RG_FAIL(RG_ERR_BAD_SHAPE, "loss: weight"); w = (const float*)weight->ptr; }
  if (grad && grad->ptr) { if (grad->dtype != RG_F32 || grad->sizes[0] != n) RG_FAIL(RG_ERR_BAD_SHAPE, "loss: grad"); g = (float*)grad->ptr; }
  int nblk = (int)((n + 256 * 8 - 1) / (256 * 8)); if (nblk > LOSS_BLOCKS) nblk = LOSS_BLOCKS; if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(loss_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)s, kind, (const float*)pred->ptr, t, w, const_target, grad_scale, g, out, n);
  RG_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, out, nblk);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
