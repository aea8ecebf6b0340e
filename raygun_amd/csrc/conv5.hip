// halo16 — 3x3x3 stride-1 convolutions of the thin (<= 16 lanes in, <= 16 channels out) U-Net levels with the INPUT HALO TILE
// staged in LDS: Conv3d forward and backward-data of the 12 -> 12 layers at 192^3 / 104^3 (networks/UNet.py:58-69).
//
// igemm_thin re-reads every input voxel 27 times through the texture path (864 B per output voxel; it runs at ~165 TFLOP/s =
// 0.33 ms for 12 -> 12 at 192^3 although the layer has 55 GFLOP and moves 0.46 GB).  Here a workgroup
//   1. loads the (4+2) x (8+2) x (32+2) input voxels around a 4 x 8 x 32 output tile once (32 bytes each; coalesced 16-byte
//      pieces, out-of-volume voxels = zero padding) into LDS, split into the two 8-lane halves so that 16 consecutive voxels of
//      one half are 256 contiguous bytes — a conflict-free ds_read_b128 for the 16 voxel lanes of an MFMA B fragment;
//   2. keeps the whole packed weight matrix (16 rows x 448 = 27 taps x 16 lanes padded to 64) in registers: 14 A fragments per lane;
//   3. per 16-voxel fragment issues 14 LDS reads (lane = (voxel n, k-group g): tap 2s + (g >> 1), half g & 1; the per-lane tap
//      offsets are 14 registers computed once) and 14 chained MFMAs (16x16x32), then the fused bias / activation epilogue.
// LDS read time equals MFMA time (one 1-KiB read per MFMA): the kernel sits at the LDS / MFMA rate instead of the L1 rate.
// Measured (12 -> 12 at 192^3, MI355X): 0.333 ms on igemm_thin -> 0.162 ms.  Ablation of the phases (tools/halo_probe.py history in
// DESIGN.md 5): fragment loop alone 0.114 ms, loads + LDS stores alone 0.055 ms — the two workgroups of a CU overlap little.
// What it took, in order: (1) a loader without per-piece divisions, (2) hand-issued ds_read_b128 (hipcc kept 2 of the 14 reads of
// a fragment in flight and waited a round trip before every MFMA), (3) addresses as immediates (fully unrolled fragment loop, K
// order chosen so that the second tap of a step is a constant away from the first), (4) a branch-free epilogue (the run-time
// activation switch around each of the 4 elements cost more than the 14 MFMAs): 0.267 -> 0.247 -> 0.233 -> 0.162 ms.
// K order = the FLAT packed-weight order (tap-major, 16 lanes per tap), so rg_pack_weights' operand is used as it is.
#include "common.h"
#include <stdlib.h>
#include <utility>

struct Halo16Args {
  const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y;
  int N, ID, IH, IW, RD, RH, RW;
  int lo[3];        // input coordinate of halo voxel 0 relative to the tile's first output coordinate
  int flip;         // 0: tap k reads halo offset k (tstep +1); 1: offset 2 - k (tstep -1, backward-data)
  long xsn, xsd, xsh, xsw, ysn, ysd, ysh, ysw;
  int cout, act; float slope;
  int tiles_d, tiles_h, tiles_w, krow;
};

#define H16_TD 4
#define H16_TH 8
#define H16_TW 32
#define H16_HD (H16_TD + 2)
#define H16_HH (H16_TH + 2)
#define H16_HW 36                                   // 34 voxels + 2 pad entries: the half stride is a multiple of 16 entries
#define H16_HALF (H16_HD * H16_HH * H16_HW)         // 2160 entries = 135 * 16
#define H16_ROWP 68                                 // 16-byte pieces per halo row (34 voxels x 2 halves)

typedef __attribute__((address_space(3))) void* h16_lptr;
// LDS reads the compiler does not reorder or throttle (it kept only two of the 14 fragment reads of a K loop in flight and waited
// a full LDS round trip in front of every MFMA: ~2000 cycles per fragment instead of ~300); waits are counted by hand
template <int N>
__device__ __forceinline__ void h16_wait(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
template <class F, int... Is>
__device__ __forceinline__ void h16_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void h16_for(F&& f) { h16_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- K order of the kernel: 14 steps of two taps (k-groups 0,1 = tap A, 2,3 = tap B; the two halves of a tap's 16 lanes within).
// Steps 0..8: row r = (kd,kh) = s, taps (kw 0, kw 1); steps 9..11: kw 2 of rows (0,1), (3,4), (6,7); step 12: kw 2 of rows (2,5);
// step 13: kw 2 of row 8 and a zero pad.  Within a step tap B's halo entry is tap A's plus a CONSTANT (1, HW, HH*HW entries; negated
// for backward-data), so a lane needs four base addresses per wave — not per fragment: the fragment's and the step's parts of the
// address are compile-time immediates of ds_read_b128 (the fragment loop is fully unrolled) — and no address arithmetic per read.
__host__ __device__ constexpr int h16_row_a(int s) { return s < 9 ? s : (s == 9 ? 0 : s == 10 ? 3 : s == 11 ? 6 : s == 12 ? 2 : 8); }
__host__ __device__ constexpr int h16_row_b(int s) { return s < 9 ? s : (s == 9 ? 1 : s == 10 ? 4 : s == 11 ? 7 : s == 12 ? 5 : -1); }
__host__ __device__ constexpr int h16_kw_a(int s) { return s < 9 ? 0 : 2; }
__host__ __device__ constexpr int h16_kw_b(int s) { return s < 9 ? 1 : 2; }
__host__ __device__ constexpr int h16_cls(int s) { return s < 9 ? 0 : (s < 12 ? 1 : (s == 12 ? 2 : 3)); }   // which base register
__host__ __device__ constexpr int h16_entry(int r, int kw, bool flip) {
  int kd = r / 3, kh = r % 3;
  if (flip) { kd = 2 - kd; kh = 2 - kh; kw = 2 - kw; }
  return (kd * H16_HH + kh) * H16_HW + kw;
}
template <int IMM>
__device__ __forceinline__ void h16_read(bf16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM)); }

// Persistent workgroups (two per CU): weights (14 A fragments per lane) and base addresses are set up once.  Loader geometry
// without divisions: thread t < 204 owns piece t % 68 (voxel (t % 68) >> 1, half (t % 68) & 1) of the halo rows t / 68, t / 68 + 3,
// ... (60 rows = 20 per thread).  No register prefetch of the next tile (its 80 staging registers next to 56 + 56 fragment
// registers spill): the CU's other workgroup computes while this one waits for its loads.
// ACT: RG_ACT_NONE / RG_ACT_RELU compiled in (the U-Net's layers), -1 = the run-time switch of act_fwd (its scalar branches around
// every element cost more than the fragment's 14 MFMAs)
template <bool FLIP, int ACT>
__global__ __launch_bounds__(256, 2) void halo16_kernel(Halo16Args a) {
  __shared__ __attribute__((aligned(256))) bf16x8 tile[2 * H16_HALF];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = a.N * a.tiles_d * a.tiles_h * a.tiles_w;
  const int nlane = lane & 15, g = lane >> 4, gsel = g >> 1;
  // ---- A fragments: lane (row l & 15, k-group g) of step s holds Wp[row][tap * 16 + (g & 1) * 8 .. + 8], tap = A or B by g >> 1
  bf16x8 wf[14];
  {
    const bf16_t* wp = a.w + (long)(lane & 15) * a.krow + (g & 1) * 8;
    h16_for<14>([&](auto S) {
      constexpr int s = decltype(S)::value;
      constexpr int ta = h16_row_a(s) * 3 + h16_kw_a(s), tb = h16_row_b(s) < 0 ? -1 : h16_row_b(s) * 3 + h16_kw_b(s);
      const int t = gsel ? tb : ta;
      wf[s] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (t >= 0) wf[s] = *(const bf16x8*)(wp + t * 16);
    });
  }
  // ---- base addresses (bytes): wave part (fh0 = wave >> 1, fw = wave & 1) + lane part + k-group part of each step class
  // forward: tap B = tap A + D, the immediate is tap A's entry and the B lanes (gsel 1) carry +D; backward-data (FLIP): tap B = tap A - D,
  // the immediate is tap B's entry and the A lanes carry +D — bases never fall below the tile
  const int fw = wave & 1, fh0 = wave >> 1, up = FLIP ? 1 - gsel : gsel;
  const unsigned lds0 = (unsigned)(size_t)(h16_lptr)tile;
  const unsigned common = lds0 + 16u * (unsigned)((g & 1) * H16_HALF + nlane + fh0 * H16_HW + fw * 16);
  const unsigned bases[4] = {common + 16u * (unsigned)up, common + 16u * (unsigned)(up * H16_HW), common + 16u * (unsigned)(up * H16_HH * H16_HW), common};
  float bv[4], cm[4];   // bias and channel mask (1 below cout, 0 for the pad lanes) of this lane's four output channels
#pragma unroll
  for (int r = 0; r < 4; ++r) { cm[r] = g * 4 + r < a.cout ? 1.f : 0.f; bv[r] = (a.bias && g * 4 + r < a.cout) ? a.bias[g * 4 + r] : 0.f; }
  // ---- loader constants
  const bool loader = tid < 3 * H16_ROWP;
  const int rl = tid / H16_ROWP, lp = tid - rl * H16_ROWP, lw = lp >> 1, lhalf = lp & 1;   // (one division per thread, once)
  auto decode = [&](int t, int& n, int& d0, int& h0, int& w0) {
    const int tw = t % a.tiles_w; t /= a.tiles_w; const int th = t % a.tiles_h; t /= a.tiles_h; const int td = t % a.tiles_d; n = t / a.tiles_d;
    d0 = td * H16_TD; h0 = th * H16_TH; w0 = tw * H16_TW;
  };
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int n, d0, h0, w0; decode(t, n, d0, h0, w0);
    {  // halo tile -> registers -> LDS
      bf16x8 v[20];
      const int iw = w0 + a.lo[2] + lw;
      const bool wok = loader && (unsigned)iw < (unsigned)a.IW;
      const bf16_t* xb = a.x + (long)n * a.xsn + (long)iw * a.xsw + lhalf * 8;
      int dd = 0, hh = rl;   // halo row r = rl + 3k
#pragma unroll
      for (int k = 0; k < 20; ++k) {
        const int id = d0 + a.lo[0] + dd, ih = h0 + a.lo[1] + hh;
        v[k] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (wok && (unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH) v[k] = *(const bf16x8*)(xb + (long)id * a.xsd + (long)ih * a.xsh);
        hh += 3; if (hh >= H16_HH) { hh -= H16_HH; ++dd; }
      }
      if (loader) {
        int idx = lhalf * H16_HALF + rl * H16_HW + lw;   // row r -> entry r * HW (r = dd * HH + hh)
#pragma unroll
        for (int k = 0; k < 20; ++k) { tile[idx] = v[k]; idx += 3 * H16_HW; }
      }
    }
    __syncthreads();
    // ---- fragments: 64 per tile; this wave's 16: (fd, fh0 + 2 j, fw), fd, j = 0..3 — fully unrolled, addresses are immediates
    const int w = w0 + fw * 16 + nlane;
    bf16_t* yp = a.y + (long)n * a.ysn + (long)d0 * a.ysd + (long)(h0 + fh0) * a.ysh + (long)w * a.ysw + g * 4;
    const bool wvalid = w < a.RW;
    h16_for<16>([&](auto F) {
      constexpr int fd = decltype(F)::value >> 2, j = decltype(F)::value & 3;
      constexpr int FOFF = 16 * ((fd * H16_HH + 2 * j) * H16_HW);
      bf16x8 xf[14];
      h16_for<14>([&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr int ea = h16_entry(h16_row_a(s), h16_kw_a(s), FLIP);
        constexpr int eb = h16_row_b(s) < 0 ? ea : h16_entry(h16_row_b(s), h16_kw_b(s), FLIP);
        static_assert(h16_row_b(s) < 0 || (FLIP ? ea - eb : eb - ea) == (h16_cls(s) == 0 ? 1 : h16_cls(s) == 1 ? H16_HW : H16_HH * H16_HW), "tap B = tap A +- D");
        h16_read<FOFF + 16 * (ea < eb ? ea : eb)>(xf[s], bases[h16_cls(s)]);   // all 14 in flight
      });
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      h16_for<14>([&](auto S) {
        constexpr int s = decltype(S)::value;
        h16_wait<13 - s>(xf[s]);
        acc = RG_MFMA16(wf[s], xf[s], acc);
      });
      if (wvalid && d0 + fd < a.RD && h0 + fh0 + 2 * j < a.RH) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float val = acc[r] + bv[r];
          if constexpr (ACT == RG_ACT_RELU) val = fmaxf(val, 0.f);
          else if constexpr (ACT != RG_ACT_NONE) val = act_fwd(val, a.act, a.slope);
          o[r] = (short)f2bf(val * cm[r]);   // pad lanes of the voxel record stay zero (sigmoid(0) is not)
        }
        *(bf16x4*)(yp + (long)fd * a.ysd + (long)(2 * j) * a.ysh) = o;
      }
    });
    __syncthreads();   // everybody is done with the tile before the next one is stored over it
  }
}

// eligible: 16 input lanes, <= 16 output channels in 16 lanes, 3x3x3 taps, unit row / output strides, all taps walking the same way
bool rg_halo16_ok(const CL5& X, const CL5& Y, const rg_conv_params* p) {
  if (X.lanes != 16 || Y.lanes != 16 || p->cout > 16) return false;
  for (int d = 0; d < 3; ++d) {
    if (p->T[d] != 3 || p->rstride[d] != 1 || p->ostride[d] != 1 || p->ooffset[d] != 0) return false;
    if (p->tstep[d] != p->tstep[0] || (p->tstep[d] != 1 && p->tstep[d] != -1)) return false;
  }
  if ((long)p->R[0] * p->R[1] * p->R[2] < 32768) return false;   // small problems stay on the persistent thin kernel
  return true;
}
int rg_launch_halo16(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p, int krow, hipStream_t st) {
  Halo16Args a;
  a.x = X.p; a.w = (const bf16_t*)w_packed; a.bias = bias; a.y = Y.p;
  a.N = X.N; a.ID = X.D; a.IH = X.H; a.IW = X.W; a.RD = p->R[0]; a.RH = p->R[1]; a.RW = p->R[2];
  a.flip = p->tstep[0] < 0;
  for (int d = 0; d < 3; ++d) a.lo[d] = p->tbase[d] + (a.flip ? -2 : 0);
  a.xsn = X.sn; a.xsd = X.sd; a.xsh = X.sh; a.xsw = X.sw; a.ysn = Y.sn; a.ysd = Y.sd; a.ysh = Y.sh; a.ysw = Y.sw;
  a.cout = p->cout; a.act = p->act; a.slope = p->slope; a.krow = krow;
  a.tiles_d = (a.RD + H16_TD - 1) / H16_TD; a.tiles_h = (a.RH + H16_TH - 1) / H16_TH; a.tiles_w = (a.RW + H16_TW - 1) / H16_TW;
  const long nblk = (long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
  if (krow != 448 || nblk >= (1L << 31)) RG_FAIL(RG_ERR_UNSUPPORTED, "halo16: unexpected packing (krow %d) or grid", krow);
  rg_set_kernel_name("halo16_kernel<%d>", a.flip);
  const unsigned grid = (unsigned)(nblk < 512 ? nblk : 512);   // two persistent workgroups per CU (69 KB of LDS each)
#define H16_GO(FL_, AC_) hipLaunchKernelGGL((halo16_kernel<FL_, AC_>), dim3(grid), dim3(256), 0, st, a)
  if (a.act == RG_ACT_RELU) { if (a.flip) H16_GO(true, RG_ACT_RELU); else H16_GO(false, RG_ACT_RELU); }
  else if (a.act == RG_ACT_NONE) { if (a.flip) H16_GO(true, RG_ACT_NONE); else H16_GO(false, RG_ACT_NONE); }
  else { if (a.flip) H16_GO(true, -1); else H16_GO(false, -1); }
#undef H16_GO
  return 0;
}
