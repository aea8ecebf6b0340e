// Train-time target generation for the MTLSD system on the GPU: what the reference's data pipe computes on
// the CPU for every batch (src/raygun/io/MTLSDDataPipe.py:99-131) —
//   gp.AddAffinities          rg_add_affinities    labels -> per-offset affinities (+ the same graph of the label mask)
//   gp.BalanceLabels          rg_balance_labels    affinities (+ mask) -> per-voxel loss weights, class-balanced over the masked-in voxels
//   AddLocalShapeDescriptor   rg_lsd_targets       labels -> 10 local shape descriptor channels (lsd's "gaussian" mode)
// HBM-bound integer / float32 voxel work (no GEMM shape): one thread per output voxel, coalesced along x; exact integer counts
// (64-bit atomics: order-independent, so deterministic), class weights in double like numpy's.  Oracle: oracle/targets.py (parity
// unpinned against gunpowder / lsd themselves: not installed, not vendored).
#include "common.h"

// ---- affinities -------------------------------------------------------------------------------------------------------------
// aff[e][p] = (seg[p] == seg[p + n_e]) && seg[p] > 0 && seg[p + n_e] > 0 for p and p + n_e inside the array, else 0
// (gunpowder/nodes/add_affinities.py seg_to_affgraph); p runs over the output ROI [off, off + oshape) of the label array.
struct AffArgs {
  const int32_t* seg; const uint8_t* mask; const int32_t* nhood; int K;
  int sz, sy, sx; int oz, oy, ox; int nz, ny, nx;   // array shape, ROI offset, ROI shape
  float* aff; float* aff_mask;
};
__global__ void add_affinities_kernel(AffArgs a) {
  const long nvox = (long)a.nz * a.ny * a.nx;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % a.nx), y = (int)((i / a.nx) % a.ny), z = (int)(i / ((long)a.nx * a.ny));
    const int pz = z + a.oz, py = y + a.oy, px = x + a.ox;
    const long p = ((long)pz * a.sy + py) * a.sx + px;
    const int s0 = a.seg[p];
    const int m0 = a.mask ? a.mask[p] : 1;
    for (int e = 0; e < a.K; ++e) {
      const int qz = pz + a.nhood[3 * e], qy = py + a.nhood[3 * e + 1], qx = px + a.nhood[3 * e + 2];
      float v = 0.f, mv = 0.f;
      if ((unsigned)qz < (unsigned)a.sz && (unsigned)qy < (unsigned)a.sy && (unsigned)qx < (unsigned)a.sx) {
        const long q = ((long)qz * a.sy + qy) * a.sx + qx;
        const int s1 = a.seg[q];
        v = (s0 == s1 && s0 > 0 && s1 > 0) ? 1.f : 0.f;
        if (a.mask) { const int m1 = a.mask[q]; mv = (m0 == m1 && m0 > 0 && m1 > 0) ? 1.f : 0.f; }
      }
      a.aff[(long)e * nvox + i] = v;
      if (a.aff_mask) a.aff_mask[(long)e * nvox + i] = mv;
    }
  }
}
extern "C" int rg_add_affinities(const int32_t* labels, const uint8_t* labels_mask, const int32_t* nhood, int32_t n_offsets, const int32_t shape[3],
                                 const int32_t out_offset[3], const int32_t out_shape[3], float* affs, float* affs_mask, rg_stream s) {
  if (!labels || !nhood || !affs || n_offsets <= 0 || (affs_mask && !labels_mask)) RG_FAIL(RG_ERR_BAD_SHAPE, "add_affinities: null pointer / no offsets / mask output without a label mask");
  for (int d = 0; d < 3; ++d)
    if (shape[d] <= 0 || out_shape[d] <= 0 || out_offset[d] < 0 || out_offset[d] + out_shape[d] > shape[d])
      RG_FAIL(RG_ERR_BAD_SHAPE, "add_affinities: the output ROI must lie inside the label array (dim %d)", d);
  AffArgs a;
  a.seg = labels; a.mask = labels_mask; a.nhood = nhood; a.K = n_offsets;
  a.sz = shape[0]; a.sy = shape[1]; a.sx = shape[2]; a.oz = out_offset[0]; a.oy = out_offset[1]; a.ox = out_offset[2];
  a.nz = out_shape[0]; a.ny = out_shape[1]; a.nx = out_shape[2]; a.aff = affs; a.aff_mask = affs_mask;
  const long nvox = (long)a.nz * a.ny * a.nx;
  long g = (nvox + 255) / 256; if (g > 65536) g = 65536;
  rg_set_kernel_name("add_affinities_kernel");
  hipLaunchKernelGGL(add_affinities_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)s, a);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---- label balancing ----------------------------------------------------------------------------------------------------------
// gunpowder/nodes/balance_labels.py, one slab, two classes: scale = mask * w[label],  w[c] = 1 / 2 / clip(count_c / masked_in).
// ws: 4 x uint64 {masked-in, class 0, class 1, other} zeroed by the count kernel's launch wrapper (hipMemsetAsync).
__global__ void balance_count_kernel(const float* lab, const float* mask, long n, unsigned long long* cnt) {
  unsigned long long in = 0, c0 = 0, c1 = 0, bad = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (mask && mask[i] == 0.f) continue;
    ++in;
    const long l = (long)lab[i];
    if (l == 0) ++c0; else if (l == 1) ++c1; else ++bad;
  }
  for (int o = 32; o > 0; o >>= 1) { in += __shfl_down(in, o); c0 += __shfl_down(c0, o); c1 += __shfl_down(c1, o); bad += __shfl_down(bad, o); }
  if ((threadIdx.x & 63) == 0) {
    if (in) atomicAdd(cnt + 0, in);
    if (c0) atomicAdd(cnt + 1, c0);
    if (c1) atomicAdd(cnt + 2, c1);
    if (bad) atomicAdd(cnt + 3, bad);
  }
}
__global__ void balance_scale_kernel(const float* lab, const float* mask, long n, const unsigned long long* cnt, double clipmin, double clipmax, float* scale) {
  double w[2];
  const double in = (double)cnt[0];
  for (int c = 0; c < 2; ++c) {
    double f = in > 0 ? (double)cnt[1 + c] / in : 0.0;
    f = f < clipmin ? clipmin : (f > clipmax ? clipmax : f);
    w[c] = cnt[1 + c] ? 1.0 / 2.0 / f : 0.0;       // (a class that does not occur keeps weight 0, as np.zeros(num_classes) does)
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float m = mask ? mask[i] : 1.f;
    const long l = (long)lab[i];
    scale[i] = (float)((double)m * ((l == 0 || l == 1) ? w[l] : 0.0));   // float32 array *= float64 array: product in double, one rounding
  }
}
extern "C" int64_t rg_balance_labels_workspace_bytes(void) { return 4 * sizeof(unsigned long long); }
extern "C" int rg_balance_labels(const float* labels, const float* mask, int64_t n, float clipmin, float clipmax, float* scale, void* workspace,
                                 size_t ws_bytes, rg_stream s) {
  if (!labels || !scale || n <= 0) RG_FAIL(RG_ERR_BAD_SHAPE, "balance_labels: null pointer or empty array");
  if (!workspace || ws_bytes < 4 * sizeof(unsigned long long)) RG_FAIL(RG_ERR_WORKSPACE, "balance_labels: workspace %zu < 32 bytes", ws_bytes);
  hipStream_t st = (hipStream_t)s;
  if (hipMemsetAsync(workspace, 0, 4 * sizeof(unsigned long long), st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "balance_labels: memset failed");
  long g = (n + 255) / 256; if (g > 4096) g = 4096;
  rg_set_kernel_name("balance_scale_kernel");
  hipLaunchKernelGGL(balance_count_kernel, dim3((unsigned)g), dim3(256), 0, st, labels, mask, (long)n, (unsigned long long*)workspace);
  hipLaunchKernelGGL(balance_scale_kernel, dim3((unsigned)g), dim3(256), 0, st, labels, mask, (long)n, (const unsigned long long*)workspace,
                     (double)clipmin, (double)clipmax, scale);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---- local shape descriptors ---------------------------------------------------------------------------------------------------
// lsd's LsdExtractor (gaussian mode, 3-D).  For output voxel P with label l != 0: on the grid subsampled by `df` (sub voxel
// s = P / df, seg_sub(q) = seg[q * df]) the Gaussian-weighted statistics of the sub voxels q with seg_sub(q) == l inside the
// truncated window — mass, mean offset, covariance — normalised as the package does.  The package filters one segment mask at a time
// (separable passes); the window sum below is the same quantity evaluated directly with relative coordinates d = (q - s) * sub_voxel_size.
struct LsdArgs {
  const int32_t* seg; int sz, sy, sx; int oz, oy, ox; int nz, ny, nx; int df; int uz, uy, ux;   // sub-grid shape
  const float* wz; const float* wy; const float* wx; int rz, ry, rx;                             // normalised 1-D weights, radii (sub voxels)
  float svz, svy, svx;       // sub voxel size (world units)
  float sgz, sgy, sgx;       // sigma (world units)
  float* out;
};
__global__ __launch_bounds__(256) void lsd_kernel(LsdArgs a) {
  const long nvox = (long)a.nz * a.ny * a.nx;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvox) return;
  const int x = (int)(i % a.nx), y = (int)((i / a.nx) % a.ny), z = (int)(i / ((long)a.nx * a.ny));
  const int pz = z + a.oz, py = y + a.oy, px = x + a.ox;
  const int l = a.seg[((long)pz * a.sy + py) * a.sx + px];
  float o[10];
#pragma unroll
  for (int c = 0; c < 10; ++c) o[c] = 0.f;
  if (l != 0) {
    const int s0 = pz / a.df, s1 = py / a.df, s2 = px / a.df;
    float cnt = 0.f, m0 = 0.f, m1 = 0.f, m2 = 0.f, c00 = 0.f, c11 = 0.f, c22 = 0.f, c01 = 0.f, c02 = 0.f, c12 = 0.f;
    const int z0 = max(0, s0 - a.rz), z1 = min(a.uz - 1, s0 + a.rz);
    const int y0 = max(0, s1 - a.ry), y1 = min(a.uy - 1, s1 + a.ry);
    const int x0 = max(0, s2 - a.rx), x1 = min(a.ux - 1, s2 + a.rx);
    for (int qz = z0; qz <= z1; ++qz) {
      const float wz = a.wz[qz - s0 + a.rz], dz = (float)(qz - s0) * a.svz;
      for (int qy = y0; qy <= y1; ++qy) {
        const float wzy = wz * a.wy[qy - s1 + a.ry], dy = (float)(qy - s1) * a.svy;
        const int32_t* row = a.seg + ((long)(qz * a.df) * a.sy + (long)qy * a.df) * a.sx;
        float r_c = 0.f, r_m = 0.f, r_mm = 0.f;    // row sums: weight, weight * dx, weight * dx^2
        for (int qx = x0; qx <= x1; ++qx) {
          if (row[(long)qx * a.df] == l) {
            const float w = a.wx[qx - s2 + a.rx], dx = (float)(qx - s2) * a.svx;
            r_c += w; r_m += w * dx; r_mm += w * dx * dx;
          }
        }
        const float wc = wzy * r_c, wm = wzy * r_m;
        cnt += wc; m0 += wc * dz; m1 += wc * dy; m2 += wm;
        c00 += wc * dz * dz; c11 += wc * dy * dy; c22 += wzy * r_mm; c01 += wc * dz * dy; c02 += wm * dz; c12 += wm * dy;
      }
    }
    float mo0, mo1, mo2;
    if (cnt == 0.f) {   // no sub voxel of this segment in the window: the package divides by 1 and subtracts the ABSOLUTE grid coordinate
      cnt = 1.f; mo0 = -(float)s0 * a.svz; mo1 = -(float)s1 * a.svy; mo2 = -(float)s2 * a.svx;
      c00 = c11 = c22 = c01 = c02 = c12 = 0.f;
      o[9] = 1.f;
    } else {
      const float ic = 1.f / cnt;
      mo0 = m0 * ic; mo1 = m1 * ic; mo2 = m2 * ic;
      c00 = c00 * ic - mo0 * mo0; c11 = c11 * ic - mo1 * mo1; c22 = c22 * ic - mo2 * mo2;
      c01 = c01 * ic - mo0 * mo1; c02 = c02 * ic - mo0 * mo2; c12 = c12 * ic - mo1 * mo2;
      o[9] = cnt;
    }
    const float v0 = c00 < 1e-3f ? 1e-3f : c00, v1 = c11 < 1e-3f ? 1e-3f : c11, v2 = c22 < 1e-3f ? 1e-3f : c22;
    o[0] = mo0 / a.sgz * 0.5f + 0.5f; o[1] = mo1 / a.sgy * 0.5f + 0.5f; o[2] = mo2 / a.sgx * 0.5f + 0.5f;
    o[3] = v0 / (a.sgz * a.sgz); o[4] = v1 / (a.sgy * a.sgy); o[5] = v2 / (a.sgx * a.sgx);
    o[6] = c01 / sqrtf(v0 * v1) * 0.5f + 0.5f; o[7] = c02 / sqrtf(v0 * v2) * 0.5f + 0.5f; o[8] = c12 / sqrtf(v1 * v2) * 0.5f + 0.5f;
#pragma unroll
    for (int c = 0; c < 10; ++c) o[c] = o[c] < 0.f ? 0.f : (o[c] > 1.f ? 1.f : o[c]);
  }
#pragma unroll
  for (int c = 0; c < 10; ++c) a.out[(long)c * nvox + i] = o[c];
}
extern "C" int rg_lsd_targets(const int32_t* labels, const int32_t shape[3], const int32_t out_offset[3], const int32_t out_shape[3], int32_t downsample,
                              const float* wz, const float* wy, const float* wx, const int32_t radius[3], const float sub_voxel_size[3],
                              const float sigma[3], float* out, rg_stream s) {
  if (!labels || !out || !wz || !wy || !wx || downsample < 1) RG_FAIL(RG_ERR_BAD_SHAPE, "lsd_targets: null pointer or downsample < 1");
  for (int d = 0; d < 3; ++d)
    if (shape[d] <= 0 || out_shape[d] <= 0 || out_offset[d] < 0 || out_offset[d] + out_shape[d] > shape[d] || radius[d] < 0 || sigma[d] <= 0.f)
      RG_FAIL(RG_ERR_BAD_SHAPE, "lsd_targets: bad ROI / radius / sigma (dim %d)", d);
  LsdArgs a;
  a.seg = labels; a.sz = shape[0]; a.sy = shape[1]; a.sx = shape[2]; a.oz = out_offset[0]; a.oy = out_offset[1]; a.ox = out_offset[2];
  a.nz = out_shape[0]; a.ny = out_shape[1]; a.nx = out_shape[2]; a.df = downsample;
  a.uz = (shape[0] + downsample - 1) / downsample; a.uy = (shape[1] + downsample - 1) / downsample; a.ux = (shape[2] + downsample - 1) / downsample;
  a.wz = wz; a.wy = wy; a.wx = wx; a.rz = radius[0]; a.ry = radius[1]; a.rx = radius[2];
  a.svz = sub_voxel_size[0]; a.svy = sub_voxel_size[1]; a.svx = sub_voxel_size[2];
  a.sgz = sigma[0]; a.sgy = sigma[1]; a.sgx = sigma[2]; a.out = out;
  const long nvox = (long)a.nz * a.ny * a.nx;
  rg_set_kernel_name("lsd_kernel");
  hipLaunchKernelGGL(lsd_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, (hipStream_t)s, a);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
