// Weight-gradient engine for gfx950:  dW[a][tap][b] = sum_rows G[row][a] * X[row @ tap][b]
//
// The reduction runs over voxels, which is the SLOW dimension of both channels-last operands, so
// both MFMA operands need a transpose.  The tiles are staged in LDS in their natural [row][channel]
// layout (coalesced 16-byte global loads) and read back with ds_read_b64_tr_b16, the gfx950
// hardware-transposing LDS read, two per 16x32 fragment.  LDS row strides are kept at an odd
// multiple of 32 bytes so the 8 rows a 32-lane half touches tile a 256-byte bank row.
//
// One workgroup owns a (16*FA x 16*FB) channel tile of ONE tap over a slice of the rows
// (grid.z = row splits).  Partial sums go to a caller-provided fp32 workspace [split][A][taps][B];
// wgrad_finish reduces the splits in a fixed order (bitwise reproducible) and transposes into the
// torch weight layout (A,B,KD,KH,KW).
#include "common.h"

struct WgradArgs {
  const bf16_t* g; const bf16_t* x; float* part;
  int N, ID, IH, IW;
  int RD, RH, RW;
  int TD, TH, TW;
  int rs[3], tb[3], ts[3];
  long gsn, gsd, gsh, gsw;
  long xsn, xsd, xsh, xsw;
  int A, B, alanes, blanes;
  long M; long rows_per_split; int nsplit;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define WG_OOB 0x80000000u  // voffset of a piece that loads nothing: stays >= num_records (< 2^30) after any per-step add
struct RowInfo { long goff; long xoff; };  // -1 = row contributes zeros

constexpr int wg_pad(int t) { return ((t / 16) & 1) ? t : t + 16; }

template <int FA, int FB, int WA, int WB>
__global__ __launch_bounds__(WA* WB * 64) void wgrad_kernel(WgradArgs a) {
  constexpr int NT = WA * WB * 64;
  constexpr int TA = FA * 16, TB = FB * 16;
  constexpr int SA = wg_pad(TA), SB = wg_pad(TB);
  constexpr int CA = TA / 8, CB = TB / 8;       // 16-byte chunks per row
  constexpr int KR = 32;  // rows per barrier step (64 measured slower: 199 vs 326 TFLOP/s on 3024->432, register pressure)
  constexpr int NCH = KR * (CA + CB);
  constexpr int PER_T = (NCH + NT - 1) / NT;
  constexpr int IA = FA / WA, IB = FB / WB;     // fragments per wave
  static_assert(FA % WA == 0 && FB % WB == 0, "wave layout must divide the tile");
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * KR * (SA + SB)];
  __shared__ RowInfo ri[2][KR];
  bf16_t* Gs = lds;                  // [2][KR][SA]
  bf16_t* Xs = lds + 2 * KR * SA;    // [2][KR][SB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave / WB, wb = wave % WB;
  const int ntb = (a.B + TB - 1) / TB;
  const int a0 = (blockIdx.x / ntb) * TA, b0 = (blockIdx.x % ntb) * TB;
  const int tap = blockIdx.y;
  const int kw = tap % a.TW, kh = (tap / a.TW) % a.TH, kd = tap / (a.TW * a.TH);
  const int od = a.tb[0] + kd * a.ts[0], oh = a.tb[1] + kh * a.ts[1], ow = a.tb[2] + kw * a.ts[2];
  const long mbeg = (long)blockIdx.z * a.rows_per_split;
  long mend = mbeg + a.rows_per_split; if (mend > a.M) mend = a.M;
  const int nsteps = (int)((mend - mbeg + KR - 1) / KR);

  f32x4 acc[IA][IB];
#pragma unroll
  for (int i = 0; i < IA; ++i)
#pragma unroll
    for (int j = 0; j < IB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto rowinfo = [&](int step) {
    if (tid < KR) {
      const long m = mbeg + (long)step * KR + tid;
      RowInfo r; r.goff = -1; r.xoff = -1;
      if (m < mend) {
        long t = m;
        const int rw = (int)(t % a.RW); t /= a.RW;
        const int rh = (int)(t % a.RH); t /= a.RH;
        const int rd = (int)(t % a.RD); t /= a.RD;
        r.goff = t * a.gsn + rd * a.gsd + rh * a.gsh + rw * a.gsw;
        const int id = rd * a.rs[0] + od, ih = rh * a.rs[1] + oh, iw = rw * a.rs[2] + ow;
        if ((unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW)
          r.xoff = t * a.xsn + id * a.xsd + ih * a.xsh + iw * a.xsw;
      }
      ri[step & 1][tid] = r;
    }
  };

  bf16x8 regs[PER_T];
  auto gload = [&](int step) {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int q = tid + NT * i;
      bf16x8 v = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (q < KR * CA) {
        const int row = q / CA, c = a0 + (q % CA) * 8;
        const long off = ri[step & 1][row].goff;
        if (off >= 0 && c < a.alanes) v = *(const bf16x8*)(a.g + off + c);
      } else if (q < NCH) {
        const int q2 = q - KR * CA;
        const int row = q2 / CB, c = b0 + (q2 % CB) * 8;
        const RowInfo r = ri[step & 1][row];
        if (r.goff >= 0 && r.xoff >= 0 && c < a.blanes) v = *(const bf16x8*)(a.x + r.xoff + c);
      }
      regs[i] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int q = tid + NT * i;
      if (q < KR * CA) {
        *(bf16x8*)(Gs + (buf * KR + q / CA) * SA + (q % CA) * 8) = regs[i];
      } else if (q < NCH) {
        const int q2 = q - KR * CA;
        *(bf16x8*)(Xs + (buf * KR + q2 / CB) * SB + (q2 % CB) * 8) = regs[i];
      }
    }
  };

  if (nsteps > 0) {
    rowinfo(0);
    __syncthreads();
    gload(0);
  }
  const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
  typedef __attribute__((address_space(3))) bf16x4 lds_b64;
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    lstore(buf);
    if (s + 1 < nsteps) rowinfo(s + 1);
    __syncthreads();
    if (s + 1 < nsteps) gload(s + 1);
#pragma unroll
    for (int sub = 0; sub < KR / 32; ++sub) {
      bf16x8 gf[IA], xf[IB];
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        const bf16_t* base = Gs + (buf * KR + sub * 32 + 8 * fg + fq) * SA + (wa * IA + i) * 16 + 4 * fp;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base + 4 * SA));
        gf[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < IB; ++j) {
        const bf16_t* base = Xs + (buf * KR + sub * 32 + 8 * fg + fq) * SB + (wb * IB + j) * 16 + 4 * fp;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base + 4 * SB));
        xf[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int j = 0; j < IB; ++j)
          acc[i][j] = RG_MFMA16(gf[i], xf[j], acc[i][j]);
    }
  }

  const int ntaps = a.TD * a.TH * a.TW;
  float* out = a.part + (long)blockIdx.z * a.A * ntaps * a.B;
#pragma unroll
  for (int i = 0; i < IA; ++i)
#pragma unroll
    for (int j = 0; j < IB; ++j) {
      const int bb = b0 + (wb * IB + j) * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int aa = a0 + (wa * IA + i) * 16 + fg * 4 + r;
        if (aa < a.A && bb < a.B) out[((long)aa * ntaps + tap) * a.B + bb] = acc[i][j][r];
      }
    }
}

// dw[a][b][tap] (+)= sum_s part[s][a][tap][b].  One workgroup = one `a` x 64 consecutive `b`: the [tap][64] slab is read
// with coalesced 256-byte rows, transposed through LDS and written as one contiguous run of 64*ntaps floats.
// out[j][i] = sum_{s in [j*R, (j+1)*R)} part[s][i]: folds many row splits (thin layers use up to 2048) before the finish
#define WF_FOLD 16
__global__ void wgrad_fold_kernel(const float* part, float* out, long total, int nsplit) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int s0 = blockIdx.y * WF_FOLD;
  const int s1 = s0 + WF_FOLD < nsplit ? s0 + WF_FOLD : nsplit;
  float v = 0.f;
  for (int s = s0; s < s1; ++s) v += part[s * total + i];
  out[blockIdx.y * total + i] = v;
}
#define WF_MAXTAPS 64
// One workgroup per (output channel a, 64 input channels): the source is ntaps runs of 256 B, the destination one run of
// 64*ntaps floats.  Threads walk the DESTINATION (perfectly coalesced stores) and gather their source element: the 6.9 KB
// source footprint of a workgroup stays in the CU's L1, so the strided 4-byte reads cost tag look-ups, not HBM traffic.
// No LDS, no barrier, 8 independent loads in flight per thread (the LDS-transposing variants ran at ~2 TB/s).
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* part, float* dw, int A, int B, int ntaps, int nsplit, int accumulate) {
  const int nbt = (B + 63) / 64;
  const int aa = blockIdx.x / nbt, b0 = (blockIdx.x % nbt) * 64;
  const long total = (long)A * ntaps * B;
  const int nb = B - b0 < 64 ? B - b0 : 64;
  const int nout = nb * ntaps;
  const float* src = part + (long)aa * ntaps * B + b0;
  float* o = dw + ((long)aa * B + b0) * ntaps;
  constexpr int U = 8;
  for (int i0 = threadIdx.x; i0 < nout; i0 += 256 * U) {
    float v[U]; int off[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int i = i0 + 256 * k, bb = i / ntaps, t = i - bb * ntaps;
      off[k] = i < nout ? t * B + bb : -1;
      v[k] = 0.f;
    }
    for (int s = 0; s < nsplit; ++s) {
      const float* ps = src + s * total;
#pragma unroll
      for (int k = 0; k < U; ++k) if (off[k] >= 0) v[k] += ps[off[k]];
    }
#pragma unroll
    for (int k = 0; k < U; ++k) if (off[k] >= 0) { const int i = i0 + 256 * k; o[i] = accumulate ? o[i] + v[k] : v[k]; }
  }
}

// ---- thin layers (A <= 16 and B <= 16 channels: the 1/12-channel U-Net levels and the 1x1x1 heads) -------------
// One tap per workgroup would re-stream both operands 27 times for one MFMA each; here a workgroup owns a slice of the
// rows for ALL taps (up to TMAX per group): per 32-row step the G tile is staged once, every tap's shifted X tile
// next to it, and the four waves take the taps round-robin (2+2 transposing LDS reads and one MFMA per tap).
#define WT_TMAX 28
struct ThinRow { long goff; long xoff; int id, ih, iw, ok; };
__global__ __launch_bounds__(256) void wgrad_thin_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[(WT_TMAX + 1) * 32 * 16];
  __shared__ ThinRow ri[32];
  __shared__ int4 tapinfo[WT_TMAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntaps = a.TD * a.TH * a.TW;
  const int t0 = blockIdx.x * WT_TMAX;
  const int nt = ntaps - t0 < WT_TMAX ? ntaps - t0 : WT_TMAX;
  if (tid < nt) {
    const int tap = t0 + tid;
    const int kw = tap % a.TW, kh = (tap / a.TW) % a.TH, kd = tap / (a.TW * a.TH);
    const int dd = kd * a.ts[0], dh = kh * a.ts[1], dw = kw * a.ts[2];
    tapinfo[tid] = make_int4(dd, dh, dw, (int)(dd * a.xsd + dh * a.xsh + dw * a.xsw));
  }
  const long mbeg = (long)blockIdx.z * a.rows_per_split;
  long mend = mbeg + a.rows_per_split; if (mend > a.M) mend = a.M;
  const int nsteps = (int)((mend - mbeg + 31) / 32);
  f32x4 acc[(WT_TMAX + 3) / 4];
#pragma unroll
  for (int i = 0; i < (WT_TMAX + 3) / 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
  typedef __attribute__((address_space(3))) bf16x4 lds_b64;
  for (int s = 0; s < nsteps; ++s) {
    __syncthreads();  // previous step's fragment reads are done
    if (tid < 32) {
      const long m = mbeg + (long)s * 32 + tid;
      ThinRow r; r.ok = m < mend; r.goff = 0; r.xoff = 0; r.id = r.ih = r.iw = 0;
      if (r.ok) {
        long t = m;
        const int rw = (int)(t % a.RW); t /= a.RW;
        const int rh = (int)(t % a.RH); t /= a.RH;
        const int rd = (int)(t % a.RD); t /= a.RD;
        r.goff = t * a.gsn + rd * a.gsd + rh * a.gsh + rw * a.gsw;
        r.id = rd * a.rs[0] + a.tb[0]; r.ih = rh * a.rs[1] + a.tb[1]; r.iw = rw * a.rs[2] + a.tb[2];
        r.xoff = t * a.xsn + r.id * a.xsd + r.ih * a.xsh + r.iw * a.xsw;
      }
      ri[tid] = r;
    }
    __syncthreads();
    {
      // thread -> fixed (row, 8-lane half); it walks the tiles 4 apart: tile 0 = G, tile 1+t = X at tap t0+t
      const int row = (tid >> 1) & 31, c = (tid & 1) * 8;
      const ThinRow r = ri[row];
      const bool cb = c < a.blanes;
      constexpr int NL = (WT_TMAX + 1 + 3) / 4;
      bf16x8 v[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) {  // issue every load before the first LDS store (one exposed latency, not NL)
        const int tile = (tid >> 6) + 4 * k;
        v[k] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (r.ok && tile <= nt) {
          if (tile == 0) { if (c < a.alanes) v[k] = *(const bf16x8*)(a.g + r.goff + c); }
          else {
            const int4 ti = tapinfo[tile - 1];  // (dd, dh, dw, element offset)
            const int id = r.id + ti.x, ih = r.ih + ti.y, iw = r.iw + ti.z;
            if (cb && (unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW)
              v[k] = *(const bf16x8*)(a.x + r.xoff + ti.w + c);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int tile = (tid >> 6) + 4 * k;
        if (tile <= nt) *(bf16x8*)(lds + (tile * 32 + row) * 16 + c) = v[k];
      }
    }
    __syncthreads();
    const bf16_t* gb = lds + (8 * fg + fq) * 16 + 4 * fp;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(gb));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(gb + 4 * 16));
    const bf16x8 gf = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int i = 0; i < (WT_TMAX + 3) / 4; ++i) {
      const int t = wave + 4 * i;
      if (t < nt) {
        const bf16_t* xb = lds + ((1 + t) * 32 + 8 * fg + fq) * 16 + 4 * fp;
        bf16x4 l2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(xb));
        bf16x4 h2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(xb + 4 * 16));
        const bf16x8 xf = (bf16x8){l2[0], l2[1], l2[2], l2[3], h2[0], h2[1], h2[2], h2[3]};
        acc[i] = RG_MFMA16(gf, xf, acc[i]);
      }
    }
  }
  float* out = a.part + (long)blockIdx.z * a.A * ntaps * a.B;
  const int bb = lane & 15;
#pragma unroll
  for (int i = 0; i < (WT_TMAX + 3) / 4; ++i) {
    const int t = wave + 4 * i;
    if (t >= nt) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int aa = fg * 4 + r;
      if (aa < a.A && bb < a.B) out[((long)aa * ntaps + t0 + t) * a.B + bb] = acc[i][r];
    }
  }
}

// ---- thin layers, line-aligned variant (row stride 1, tap step +1: every stride-1 convolution) ------------------
// K-steps are 32 consecutive voxels of ONE output line, so the three (TW) kw-taps of a (kd,kh) pair read the same
// staged input line segment at row offsets 0..TW-1: 1 + TD*TH tiles are loaded per step instead of 1 + TD*TH*TW
// (10.8 KB instead of 28.7 KB for 3x3x3), and ds_read_b64_tr_b16 still sees 4 consecutive k-rows per block.
#define WL_MAXJ 9    /* (kd,kh) pairs per workgroup */
#define WL_MAXW 8    /* taps along W */
// Same load-path recipe as wgrad_kw_kernel: every 16-byte piece has a per-thread constant byte offset relative to a
// wave-uniform base, buffer descriptors do the bounds checks (G: one output line, rows past RW read zero; X: the whole
// tensor), the unit cursor is incremental, and the loads of unit u+1 fly during the MFMAs of unit u (LDS double buffer,
// one barrier per unit).  CHECK (zero-padded convolutions) adds the per-piece (d,h,w) range test.
struct WgradLineArgs { WgradArgs w; long units; long units_per_split; int wchunks; unsigned xbytes; };
// TWM = taps along W the instantiation is sized for (LDS rows per line, accumulators per wave): 3 or WL_MAXW
template <bool CHECK, int TWM>
__global__ __launch_bounds__(256) void wgrad_line_kernel(WgradLineArgs p) {
  const WgradArgs& a = p.w;
  constexpr int XR = 32 + TWM - 1;
  constexpr int STAGE = (32 + WL_MAXJ * XR) * 16;  // bf16 elements per LDS stage
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntaps = a.TD * a.TH * a.TW;
  const int TDH = a.TD * a.TH;
  const int j0 = blockIdx.x * WL_MAXJ;
  const int nj = TDH - j0 < WL_MAXJ ? TDH - j0 : WL_MAXJ;
  const int xrows = 32 + a.TW - 1;
  const long ubeg = (long)blockIdx.z * p.units_per_split;
  long uend = ubeg + p.units_per_split; if (uend > p.units) uend = p.units;
  constexpr int NACC = (WL_MAXJ * TWM + 3) / 4;
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
  typedef __attribute__((address_space(3))) bf16x4 lds_b64;
  const int nloc = nj * a.TW;  // taps of this workgroup: local index t = jj*TW + kw
  const int gsw2 = (int)a.gsw * 2, xsw2 = (int)a.xsw * 2;
  // per-thread piece list, fixed for the whole kernel.  Piece q < 64: the G tile (row q>>1, half q&1); then the X tiles.
  constexpr int NLD = (64 + WL_MAXJ * XR * 2 + 255) / 256;
  unsigned pv[NLD]; int pl[NLD]; int pd[NLD], ph[NLD], pw[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int q = tid + 256 * k;
    pv[k] = WG_OOB; pl[k] = -1; pd[k] = ph[k] = pw[k] = 0;
    if (q < 64) {
      const int rr = q >> 1, cc = (q & 1) * 8;
      pl[k] = rr * 16 + cc;
      if (cc < a.alanes) pv[k] = (unsigned)(rr * gsw2 + cc * 2);
    } else if (q < 64 + nj * xrows * 2) {
      const int q2 = q - 64, cc = (q2 & 1) * 8, rr = (q2 >> 1) % xrows, jj = (q2 >> 1) / xrows;
      const int kh = (j0 + jj) % a.TH, kd = (j0 + jj) / a.TH;
      pl[k] = (32 + jj * XR + rr) * 16 + cc;
      pd[k] = kd; ph[k] = kh; pw[k] = rr;
      if (cc < a.blanes) pv[k] = (unsigned)((int)(kd * a.xsd + kh * a.xsh) * 2 + rr * xsw2 + cc * 2);
    }
  }
  const bool isg0 = tid < 64;  // only piece 0 of the first 64 threads belongs to the G tile
  int xfo[NACC];               // LDS element offset of tap (wave + 4 i)'s fragment rows inside a stage, -1: no such tap
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    const int tl = wave + 4 * i;
    xfo[i] = tl < nloc ? (32 + (tl / a.TW) * XR + tl % a.TW + 8 * fg + fq) * 16 + 4 * fp : -1;
  }
  int cwc, crh, crd, cn;
  {
    unsigned t = (unsigned)ubeg;
    cwc = (int)(t % (unsigned)p.wchunks); t /= (unsigned)p.wchunks;
    crh = (int)(t % (unsigned)a.RH); t /= (unsigned)a.RH;
    crd = (int)(t % (unsigned)a.RD); cn = (int)(t / (unsigned)a.RD);
  }
  const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)p.xbytes, 0x00020000);
  u32x4 regs[NLD];
  auto gload = [&]() {  // loads the unit under the cursor, then advances the cursor
    const int w0 = cwc * 32;
    const bf16_t* gbase = a.g + cn * a.gsn + crd * a.gsd + crh * a.gsh;
    const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)gbase, (short)0, a.RW * gsw2, 0x00020000);
    const int id0 = crd + a.tb[0], ih0 = crh + a.tb[1], iw0 = w0 + a.tb[2];
    const int xadd = (int)(cn * a.xsn + (long)id0 * a.xsd + (long)ih0 * a.xsh + (long)iw0 * a.xsw) * 2;  // may be negative (padding)
    const int gadd = w0 * gsw2;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      if (k == 0) {
        const u32x4 vg = __builtin_amdgcn_raw_buffer_load_b128(rg, (int)(pv[0] + (unsigned)gadd), 0, 0);
        unsigned vo = pv[0] + (unsigned)xadd;
        if (CHECK) vo = ((unsigned)(id0 + pd[0]) < (unsigned)a.ID && (unsigned)(ih0 + ph[0]) < (unsigned)a.IH && (unsigned)(iw0 + pw[0]) < (unsigned)a.IW) ? vo : WG_OOB;
        const u32x4 vx = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(isg0 ? WG_OOB : vo), 0, 0);
        regs[0] = isg0 ? vg : vx;
      } else {
        unsigned vo = pv[k] + (unsigned)xadd;
        if (CHECK) vo = ((unsigned)(id0 + pd[k]) < (unsigned)a.ID && (unsigned)(ih0 + ph[k]) < (unsigned)a.IH && (unsigned)(iw0 + pw[k]) < (unsigned)a.IW) ? vo : WG_OOB;
        regs[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, 0, 0);
      }
    }
    if (++cwc == p.wchunks) { cwc = 0; if (++crh == a.RH) { crh = 0; if (++crd == a.RD) { crd = 0; ++cn; } } }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) if (pl[k] >= 0) *(u32x4*)(lds + buf * STAGE + pl[k]) = regs[k];
  };
  auto compute = [&](int buf) {
    const bf16_t* Gs = lds + buf * STAGE;
    const bf16_t* gb = Gs + (8 * fg + fq) * 16 + 4 * fp;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(gb));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(gb + 4 * 16));
    const bf16x8 gf = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (wave + 4 * i < nloc) {
        const bf16_t* xb = Gs + xfo[i];
        bf16x4 l2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(xb));
        bf16x4 h2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(xb + 4 * 16));
        const bf16x8 xf = (bf16x8){l2[0], l2[1], l2[2], l2[3], h2[0], h2[1], h2[2], h2[3]};
        acc[i] = RG_MFMA16(gf, xf, acc[i]);
      }
    }
  };
  if (ubeg < uend) gload();
  int buf = 0;
  for (long u = ubeg; u < uend; ++u, buf ^= 1) {
    lstore(buf);
    __syncthreads();
    if (u + 1 < uend) gload();
    compute(buf);
  }
  float* out = a.part + (long)blockIdx.z * a.A * ntaps * a.B;
  const int bb = lane & 15;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    const int tl = wave + 4 * i;
    if (tl >= nloc) continue;
    const int tap = (j0 + tl / a.TW) * a.TW + tl % a.TW;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int aa = fg * 4 + r;
      if (aa < a.A && bb < a.B) out[((long)aa * ntaps + tap) * a.B + bb] = acc[i][r];
    }
  }
}
static long wg_extent(const CL5& T) {
  return (long)(T.N - 1) * T.sn + (long)(T.D - 1) * T.sd + (long)(T.H - 1) * T.sh + (long)(T.W - 1) * T.sw + T.lanes;
}
static bool wgrad_line_ok(const CL5& G, const CL5& X, const rg_conv_params* p, int fa, int fb) {
  if (fa != 1 || fb != 1) return false;
  if (wg_extent(X) * 2 >= (1L << 30) || wg_extent(G) * 2 >= (1L << 30)) return false;  // 32-bit byte offsets
  for (int d = 0; d < 3; ++d) if (p->rstride[d] != 1 || p->tstep[d] != 1) return false;
  return p->T[2] <= WL_MAXW;
}

// ---- thin layers, blocks of lines (3x3xTW valid convolutions of <= 16 lanes on both sides) ------------------------------------
// wgrad_line_kernel stages the nine (kd,kh) input lines of every output line: the input is read 9 times through L1/L2
// (12 -> 12 at 192^3: 2.3 GB per launch, 0.33 ms against a 0.09-ms HBM floor).  Here a unit is a block of LD x LH output lines
// (same 32-voxel chunk along W): the (LD+2) x (LH+2) input lines it needs are staged ONCE — 3 input lines per output line instead
// of 9 — and every (line, tap) pair is one MFMA on fragments read with the transposing LDS reads at line / row offsets.
// Lines past the end of D or H and rows past the end of a line are masked in the G tile (zeros), the input lines they would
// have needed are read from wherever the offsets fall inside the tensor (finite) or beyond it (zeros).
#define WB_LD 2
#define WB_LH 4
struct WgradLinesArgs { WgradArgs w; long units; long units_per_split; int wchunks, dgroups, hgroups; unsigned xbytes, gbytes; };
__global__ __launch_bounds__(256) void wgrad_lines_kernel(WgradLinesArgs p) {
  const WgradArgs& a = p.w;
  constexpr int LD = WB_LD, LH = WB_LH, GL = LD * LH, XH = LH + 2, XL = (LD + 2) * XH, XR = 34;
  constexpr int STAGE = (GL * 32 + XL * XR) * 16;  // bf16 elements per LDS stage (34 KB)
  extern __shared__ __attribute__((aligned(16))) bf16_t wl_lds[];
  bf16_t* lds = wl_lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntaps = 9 * a.TW;
  const long ubeg = (long)blockIdx.z * p.units_per_split;
  long uend = ubeg + p.units_per_split; if (uend > p.units) uend = p.units;
  constexpr int NACC = 7;  // taps wave, wave + 4, ... of at most 27
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
  typedef __attribute__((address_space(3))) bf16x4 lds_b64;
  const int gsw2 = (int)a.gsw * 2, xsw2 = (int)a.xsw * 2;
  // per-thread piece list, fixed for the whole kernel: G pieces (line, row, half) first, then X pieces
  constexpr int NG = GL * 64, NP = NG + XL * XR * 2, NLD = (NP + 255) / 256;
  unsigned pv[NLD]; int pl[NLD]; int pm[NLD];  // byte offset from the unit's base, LDS element offset (-1: none), G mask key
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int q = tid + 256 * k;
    pv[k] = WG_OOB; pl[k] = -1; pm[k] = -1;
    if (q < NG) {
      const int l = q >> 6, rr = (q >> 1) & 31, cc = (q & 1) * 8, ld = l / LH, lh = l % LH;
      pl[k] = (l * 32 + rr) * 16 + cc;
      pm[k] = ld | (lh << 8) | (rr << 16);
      if (cc < a.alanes) pv[k] = (unsigned)((int)(ld * a.gsd + lh * a.gsh) * 2 + rr * gsw2 + cc * 2);
    } else if (q < NP) {
      const int q2 = q - NG, cc = (q2 & 1) * 8, rr = (q2 >> 1) % XR, xl = (q2 >> 1) / XR, xd = xl / XH, xh = xl % XH;
      pl[k] = (GL * 32 + xl * XR + rr) * 16 + cc;
      if (cc < a.blanes) pv[k] = (unsigned)((int)(xd * a.xsd + xh * a.xsh) * 2 + rr * xsw2 + cc * 2);
    }
  }
  int xfo[NACC];  // LDS element offset of tap (wave + 4 i)'s fragment rows relative to an output line's first input line, -1: no tap
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    const int tl = wave + 4 * i;
    const int kw = tl % a.TW, kh = (tl / a.TW) % 3, kd = tl / (3 * a.TW);
    xfo[i] = tl < ntaps ? (GL * 32 + (kd * XH + kh) * XR + kw + 8 * fg + fq) * 16 + 4 * fp : -1;
  }
  int cwc, chg, cdg, cn;
  {
    unsigned t = (unsigned)ubeg;
    cwc = (int)(t % (unsigned)p.wchunks); t /= (unsigned)p.wchunks;
    chg = (int)(t % (unsigned)p.hgroups); t /= (unsigned)p.hgroups;
    cdg = (int)(t % (unsigned)p.dgroups); cn = (int)(t / (unsigned)p.dgroups);
  }
  const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)p.xbytes, 0x00020000);
  const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.g, (short)0, (int)p.gbytes, 0x00020000);
  u32x4 regs[NLD];
  auto gload = [&]() {  // loads the unit under the cursor, then advances the cursor
    const int w0 = cwc * 32, d0 = cdg * LD, h0 = chg * LH;
    const int nld = a.RD - d0, nlh = a.RH - h0, nrw = a.RW - w0;  // valid lines / rows of this unit (may exceed the block)
    const int gadd = (int)(cn * a.gsn + (long)d0 * a.gsd + (long)h0 * a.gsh + (long)w0 * a.gsw) * 2;
    const int xadd = (int)(cn * a.xsn + (long)(d0 + a.tb[0]) * a.xsd + (long)(h0 + a.tb[1]) * a.xsh + (long)(w0 + a.tb[2]) * a.xsw) * 2;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      if ((k + 1) * 256 <= NG) {          // G pieces only
        const bool ok = (pm[k] & 255) < nld && ((pm[k] >> 8) & 255) < nlh && (pm[k] >> 16) < nrw;
        regs[k] = __builtin_amdgcn_raw_buffer_load_b128(rg, (int)(ok ? pv[k] + (unsigned)gadd : WG_OOB), 0, 0);
      } else if (k * 256 >= NG) {         // X pieces only
        regs[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(pv[k] + (unsigned)xadd), 0, 0);
      } else {                            // the straddling slot: per-thread choice
        const bool isg = pm[k] >= 0;
        const bool ok = (pm[k] & 255) < nld && ((pm[k] >> 8) & 255) < nlh && (pm[k] >> 16) < nrw;
        const u32x4 vg = __builtin_amdgcn_raw_buffer_load_b128(rg, (int)(isg && ok ? pv[k] + (unsigned)gadd : WG_OOB), 0, 0);
        const u32x4 vx = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(isg ? WG_OOB : pv[k] + (unsigned)xadd), 0, 0);
        regs[k] = isg ? vg : vx;
      }
    }
    if (++cwc == p.wchunks) { cwc = 0; if (++chg == p.hgroups) { chg = 0; if (++cdg == p.dgroups) { cdg = 0; ++cn; } } }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) if (pl[k] >= 0) *(u32x4*)(lds + buf * STAGE + pl[k]) = regs[k];
  };
  auto compute = [&](int buf) {
    const bf16_t* Gs = lds + buf * STAGE;
#pragma unroll
    for (int l = 0; l < GL; ++l) {
      const bf16_t* gb = Gs + (l * 32 + 8 * fg + fq) * 16 + 4 * fp;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(gb));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(gb + 4 * 16));
      const bf16x8 gf = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      const int loff = ((l / LH) * XH + (l % LH)) * XR * 16;
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (xfo[i] >= 0) {
          const bf16_t* xb = Gs + xfo[i] + loff;
          const bf16x4 l2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(xb));
          const bf16x4 h2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(xb + 4 * 16));
          const bf16x8 xf = (bf16x8){l2[0], l2[1], l2[2], l2[3], h2[0], h2[1], h2[2], h2[3]};
          acc[i] = RG_MFMA16(gf, xf, acc[i]);
        }
      }
    }
  };
  if (ubeg < uend) gload();
  int buf = 0;
  for (long u = ubeg; u < uend; ++u, buf ^= 1) {
    lstore(buf);
    __syncthreads();
    if (u + 1 < uend) gload();
    compute(buf);
  }
  float* out = a.part + (long)blockIdx.z * a.A * ntaps * a.B;
  const int bb = lane & 15;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    const int tap = wave + 4 * i;
    if (tap >= ntaps) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int aa = fg * 4 + r;
      if (aa < a.A && bb < a.B) out[((long)aa * ntaps + tap) * a.B + bb] = acc[i][r];
    }
  }
}
static bool wgrad_lines_ok(const CL5& G, const CL5& X, const rg_conv_params* p, int fa, int fb) {
  if (fa != 1 || fb != 1 || rg_opt(RG_OPT_NO_WLINES)) return false;
  if (wg_extent(X) * 2 >= (1L << 30) || wg_extent(G) * 2 >= (1L << 30)) return false;  // 32-bit byte offsets
  if (p->T[0] != 3 || p->T[1] != 3 || p->T[2] > 3) return false;
  const int I[3] = {X.D, X.H, X.W};
  for (int d = 0; d < 3; ++d)
    if (p->rstride[d] != 1 || p->tstep[d] != 1 || p->tbase[d] < 0 || p->tbase[d] + p->T[d] - 1 + p->R[d] - 1 >= I[d]) return false;
  return (long)G.N * p->R[0] * p->R[1] * p->R[2] >= (1L << 18);  // small problems: the per-line kernel has more units to spread
}

// ---- wide layers, line-aligned K-steps with kw reuse --------------------------------------------------------------
// The flat kernel above is bound by the global->LDS path: a 144x144 tile moves 18.4 KB per 32-row step for ONE tap
// (72 FLOP/B).  Here a K-step is 32 consecutive voxels of one output line and the workgroup owns ALL TW taps of one
// (kd,kh): the X line segment (32+TW-1 rows) is staged once and read at row offsets 0..TW-1 by the transposing LDS reads,
// the G tile once for TW taps: ~TW times the FLOP per byte (216 FLOP/B for 3x3x3 at 144x144, 206 for the 7^3 64x64
// layer).  Wave (wa, wb) owns an IAxIB block of 16x16 fragments for every kw.
//
// Two row orders (host picks the one with fewer padded rows, wgrad_kw_geom):
//  line: a unit is a 32-voxel chunk of one output line; lines are padded to multiples of 32 rows.
//  flat: valid stride-1 convs over dense X planes.  Output voxels are numbered in the INPUT pitch, o = h*IW + w, so
//        tap (kh,kw) is the pure shift o + kh*IW + kw over the whole (h,w) plane: the plane is one long line, padded once
//        (17x17 outputs of a 19x19 plane: 82 % useful rows instead of 53 %).  G rows with w >= RW are masked to zero.
//
// The load path is what bounds this kernel (9 waves x ~300 VALU address instructions per step made SIMD0 the
// bottleneck: measured 4.3 of 5.0 ms with the MFMAs removed), so every piece's byte offset inside its line/plane is a
// per-thread constant, the per-step part is wave-uniform (SGPR base in a buffer descriptor whose num_records does the
// bounds check: out-of-range rows read as zero), and the unit -> (n, d, h, chunk) decode is incremental.
struct WgradKwArgs { WgradArgs w; long units; long units_per_split; int wchunks; int nta, ntb; int flat; unsigned pmagic; int dbg; };

// KHL (1 or TH): how many kh taps ONE workgroup owns.  The thin tiles (16 channels on one side) issue 4 MFMAs per wave between two barriers and are bound by that
// round trip, not by bytes; with KHL = 3 the G tile is staged once for three input lines and every barrier is followed by three times the MFMAs:
// 72 -> 12 (<5,1>: the 16-lane side is X) 0.243 -> 0.148 ms.  84 -> 12 (<1,8>: the X tile is the 128-lane side, three of them are 60 KB of LDS = two workgroups
// per CU) measured 0.46 -> 0.50 ms; its 96-lane form <1,6> (45 KB) 0.456 -> 0.39 ms.
template <int FA, int FB, int WA, int WB, int WT, int TWM, bool DEEP, int KHL = 1>
__global__ __launch_bounds__(WA* WB* WT * 64) void wgrad_kw_kernel(WgradKwArgs p) {
  const WgradArgs& a = p.w;
  constexpr int NT = WA * WB * WT * 64;
  constexpr int TPW = (TWM + WT - 1) / WT;  // taps per wave: wave wt owns taps wt, wt+WT, ...
  constexpr int TA = FA * 16, TB = FB * 16;
  constexpr int SA = wg_pad(TA), SB = wg_pad(TB);
  constexpr int CA = TA / 8, CB = TB / 8;
  constexpr int XR = 32 + TWM - 1;
  constexpr int PER_G = (32 * CA + NT - 1) / NT, PER_X = (XR * CB + NT - 1) / NT;
  constexpr int IA = FA / WA, IB = FB / WB;
  static_assert(FA % WA == 0 && FB % WB == 0, "wave layout must divide the tile");
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * (32 * SA + KHL * XR * SB)];
  bf16_t* Gs = lds;                   // [2][32][SA]
  bf16_t* Xs = lds + 2 * 32 * SA;     // [2][KHL][XR][SB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wt = wave / (WA * WB), wa = (wave / WB) % WA, wb = wave % WB;
  // XCD-aware logical order (as wgrad_kwp_kernel): workgroups are dealt round-robin to the 8 XCDs, so consecutive blockIdx values — the nine (kd, kh) of
  // one split, which read the same input lines — sat behind eight different L2s (rocprofv3 FETCH_SIZE: 3.0 GB per launch of the 84 -> 12 gradient, whose
  // operands are 260 MB); each XCD now owns a contiguous run of (tile, (kd, kh), split) triples
  int a0, b0, kh, kd, zsplit;
  {
    const int ny = gridDim.y, total = gridDim.x * ny * gridDim.z;
    int lin = blockIdx.x + gridDim.x * (blockIdx.y + ny * blockIdx.z);
    { const int q = total / 8, r = total % 8, xcd = lin % 8, idx = lin / 8; lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int t = lin % (int)gridDim.x; lin /= (int)gridDim.x; const int y = lin % ny; zsplit = lin / ny;
    a0 = (t / p.ntb) * TA; b0 = (t % p.ntb) * TB;
    if (KHL > 1) { kh = 0; kd = y; } else { kh = y % a.TH; kd = y / a.TH; }   // (KHL > 1: TH == KHL, grid.y = TD)
  }
  const int xrows = 32 + a.TW - 1;
  const long ubeg = (long)zsplit * p.units_per_split;
  long uend = ubeg + p.units_per_split; if (uend > p.units) uend = p.units;
  const int gsw2 = (int)a.gsw * 2, gsh2 = (int)a.gsh * 2, xsw2 = (int)a.xsw * 2;

  // per-thread piece tables: byte offset inside the line/plane at chunk 0, LDS element offset in buffer 0 (-1: no piece)
  unsigned gv[PER_G], xv[PER_X];
  int gl[PER_G], grow[PER_G], xl[PER_X];
#pragma unroll
  for (int k = 0; k < PER_G; ++k) {
    const int q = tid + NT * k, row = q / CA, ch = q % CA, c = a0 + ch * 8;
    const bool exists = q < 32 * CA;
    gl[k] = exists ? row * SA + ch * 8 : -1;
    grow[k] = row;
    gv[k] = (exists && c < a.alanes) ? (unsigned)((p.flat ? 0 : row * gsw2) + c * 2) : WG_OOB;
  }
#pragma unroll
  for (int k = 0; k < PER_X; ++k) {
    const int q = tid + NT * k, row = q / CB, ch = q % CB, c = b0 + ch * 8;
    const bool exists = q < xrows * CB;
    xl[k] = exists ? row * SB + ch * 8 : -1;
    xv[k] = (exists && c < a.blanes) ? (unsigned)(row * xsw2 + c * 2) : WG_OOB;
  }
  f32x4 acc[KHL][TPW][IA][IB];
#pragma unroll
  for (int h = 0; h < KHL; ++h)
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
      for (int j = 0; j < IB; ++j) acc[h][t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // load cursor (wave-uniform): unit -> (n, rd, rh, wc); flat units have no rh
  const int rhe = p.flat ? 1 : a.RH;
  int cwc, crh, crd, cn;
  {
    unsigned t = (unsigned)ubeg;
    cwc = (int)(t % (unsigned)p.wchunks); t /= (unsigned)p.wchunks;
    crh = (int)(t % (unsigned)rhe); t /= (unsigned)rhe;
    crd = (int)(t % (unsigned)a.RD); cn = (int)(t / (unsigned)a.RD);
  }
  struct Staged { u32x4 g[PER_G], x[KHL][PER_X]; };
  auto gload = [&](Staged& r) {  // loads the unit under the cursor, then advances the cursor
    const int w0 = cwc * 32;
    const int id = crd * a.rs[0] + a.tb[0] + kd * a.ts[0];
    const bf16_t* gbase; int nrecg, gadd;
    if (p.flat) { gbase = a.g + cn * a.gsn + crd * a.gsd; nrecg = a.RH * gsh2; gadd = 0; }
    else { gbase = a.g + cn * a.gsn + crd * a.gsd + crh * a.gsh; nrecg = a.RW * gsw2; gadd = w0 * gsw2; }
    const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)gbase, (short)0, nrecg, 0x00020000);
#pragma unroll
    for (int k = 0; k < PER_G; ++k) {
      unsigned vo = gv[k] + (unsigned)gadd;
      if (p.flat) {  // o = h*IW + w in the input pitch; rows with w >= RW (or past the plane) carry no gradient
        const unsigned o = (unsigned)(w0 + grow[k]), h = __umulhi(o, p.pmagic), w = o - h * (unsigned)a.IW;
        vo = (w < (unsigned)a.RW && h < (unsigned)a.RH) ? gv[k] + h * (unsigned)gsh2 + w * (unsigned)gsw2 : WG_OOB;
      }
      r.g[k] = __builtin_amdgcn_raw_buffer_load_b128(rg, (int)vo, 0, 0);
    }
#pragma unroll
    for (int hh = 0; hh < KHL; ++hh) {
      const int khh = kh + hh;
      const bf16_t* xbase; int nrecx, xadd;
      if (p.flat) {
        const bool ok = (unsigned)id < (unsigned)a.ID;
        xbase = ok ? a.x + cn * a.xsn + (long)id * a.xsd : a.x; nrecx = ok ? a.IH * a.IW * xsw2 : 0;
        xadd = ((a.tb[1] + khh) * a.IW + a.tb[2] + w0) * xsw2;
      } else {
        const int ih = crh * a.rs[1] + a.tb[1] + khh * a.ts[1];
        const bool ok = (unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH;
        xbase = ok ? a.x + cn * a.xsn + (long)id * a.xsd + (long)ih * a.xsh : a.x; nrecx = ok ? a.IW * xsw2 : 0;
        xadd = (w0 + a.tb[2]) * xsw2;
      }
      const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, (short)0, nrecx, 0x00020000);
#pragma unroll
      for (int k = 0; k < PER_X; ++k) r.x[hh][k] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(xv[k] + (unsigned)xadd), 0, 0);
    }
    if (++cwc == p.wchunks) { cwc = 0; if (++crh == rhe) { crh = 0; if (++crd == a.RD) { crd = 0; ++cn; } } }
  };
  auto lstore = [&](const Staged& r, int buf) {
#pragma unroll
    for (int k = 0; k < PER_G; ++k) if (gl[k] >= 0) *(u32x4*)(Gs + buf * 32 * SA + gl[k]) = r.g[k];
#pragma unroll
    for (int hh = 0; hh < KHL; ++hh)
#pragma unroll
      for (int k = 0; k < PER_X; ++k) if (xl[k] >= 0) *(u32x4*)(Xs + (buf * KHL + hh) * XR * SB + xl[k]) = r.x[hh][k];
  };

  const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
  typedef __attribute__((address_space(3))) bf16x4 lds_b64;
  auto compute = [&](int buf) {
    bf16x8 gf[IA];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const bf16_t* base = Gs + (buf * 32 + 8 * fg + fq) * SA + (wa * IA + i) * 16 + 4 * fp;
      bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base));
      bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base + 4 * SA));
      gf[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int hh = 0; hh < KHL; ++hh)
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
      const int t = wt + tt * WT;
      if (t < a.TW) {
#pragma unroll
        for (int j = 0; j < IB; ++j) {
          const bf16_t* base = Xs + ((buf * KHL + hh) * XR + t + 8 * fg + fq) * SB + (wb * IB + j) * 16 + 4 * fp;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base + 4 * SB));
          const bf16x8 xf = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int i = 0; i < IA; ++i) acc[hh][tt][i][j] = RG_MFMA16(gf[i], xf, acc[hh][tt][i][j]);
        }
      }
    }
  };
  if constexpr (DEEP) {
    // step u's operands sit in registers two steps ahead; LDS is double-buffered, one barrier per step
    Staged r0, r1;
    if (ubeg < uend) gload(r0);
    if (ubeg + 1 < uend) gload(r1);
    for (long u = ubeg; u < uend; u += 2) {
      lstore(r0, 0);
      __syncthreads();
      if (u + 2 < uend) gload(r0);
      compute(0);
      if (u + 1 < uend) {
        lstore(r1, 1);
        __syncthreads();
        if (u + 3 < uend) gload(r1);
        compute(1);
      }
    }
  } else {
    Staged r0;
    if (ubeg < uend) gload(r0);
    int buf = 0;
    for (long u = ubeg; u < uend; ++u, buf ^= 1) {
      lstore(r0, buf);
      __syncthreads();
      if (u + 1 < uend) gload(r0);
      compute(buf);
    }
  }

  const int ntaps = a.TD * a.TH * a.TW;
  float* out = a.part + (long)zsplit * a.A * ntaps * a.B;
#pragma unroll
  for (int hh = 0; hh < KHL; ++hh)
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int t = wt + tt * WT;
    if (t >= a.TW) continue;
    const int tap = (kd * a.TH + kh + hh) * a.TW + t;
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
      for (int j = 0; j < IB; ++j) {
        const int bb = b0 + (wb * IB + j) * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int aa = a0 + (wa * IA + i) * 16 + fg * 4 + r;
          if (aa < a.A && bb < a.B) out[((long)aa * ntaps + tap) * a.B + bb] = acc[hh][tt][i][j][r];
        }
      }
  }
}
// ---- wgrad_kw with producer waves ----------------------------------------------------------------------------------
// s_memtime stamps on wgrad_kw_kernel<9,9> (one 9-wave workgroup per CU, 168 registers) showed where a step's ~3000 cycles
// go: ~1100 in the fragment reads + 27 MFMAs, and 1000-1800 in "issuing" three 16-byte loads per thread — at the register
// cap hipcc recycles/spills the staging registers and waits vmcnt(0) BETWEEN the loads of one step (a memory round trip
// per step); an LDS-DMA variant (no staging registers) issues at ~57 cycles per 1-KiB piece and serialises the 27 pieces
// of a step in front of the last wave just the same.  Either way the waves that should feed the MFMA pipe sit in the load
// path.  So the roles are split: the WA*WB consumer waves only read fragments and issue MFMAs; NPW producer waves (they
// fill the SIMDs the 9 consumers leave at two waves: 12 waves x 168 registers = the whole file) own the global loads
// (a third of the tile's pieces each, all in flight at once, loaded one unit ahead of their ds_write) and the LDS stores.
// One barrier per unit for everybody; LDS image conflict-free for the transposing reads (wg_row).
constexpr int wg_row(int row, int S) { return row * S + (row >> 3) * 64; }  // every second 8-row group 128 B further: see below
typedef int i32x4_t __attribute__((ext_vector_type(4)));
// raw buffer descriptor (stride 0) as four SGPR words; dword3 as __builtin_amdgcn_make_buffer_rsrc's 0x00020000
__device__ __forceinline__ i32x4_t wg_rsrc(const void* base, int nrec) {
  const unsigned long b = (unsigned long)base;
  return (i32x4_t){(int)(unsigned)b, (int)((b >> 32) & 0xffffu), nrec, 0x00020000};
}
// A buffer load hipcc does not count: the producer waves keep two units of loads in flight ACROSS barriers and retire them
// with their own counted waits (hipcc's bookkeeping put vmcnt(2..6) waits in front of the next unit's loads).
__device__ __forceinline__ void wg_load16(u32x4& dst, i32x4_t rsrc, unsigned voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
template <int N, int P>
__device__ __forceinline__ void wg_wait_loads(u32x4 (&r)[P]) {  // all but the N youngest loads have landed; names every register of the set
  static_assert(P >= 4 && P <= 8, "producer sets hold 4..8 pieces");
  if constexpr (P == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N) : "memory");
  if constexpr (P == 5) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]) : "n"(N) : "memory");
  if constexpr (P == 6) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "n"(N) : "memory");
  if constexpr (P == 7) asm volatile("s_waitcnt vmcnt(%7)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]) : "n"(N) : "memory");
  if constexpr (P == 8) asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(N) : "memory");
}
template <int FA, int FB, int WA, int WB, int TWM, int NPW>
__global__ __launch_bounds__((WA * WB + NPW) * 64, 3) void wgrad_kwp_kernel(WgradKwArgs p) {
  const WgradArgs& a = p.w;
  constexpr int NC = WA * WB;                 // consumer waves
  constexpr int TA = FA * 16, TB = FB * 16;
  constexpr int SA = wg_pad(TA), SB = wg_pad(TB);
  constexpr int CA = TA / 8, CB = TB / 8;
  constexpr int XR = 32 + TWM - 1;
  constexpr int IA = FA / WA, IB = FB / WB;
  // One ds_read_b64_tr_b16 makes a 32-lane half touch rows {r..r+3} and {r+8..r+11}, 32 bytes each: with a row stride that is
  // an odd multiple of 32 B the first four tile half a 256-byte bank row, the 128-byte push of every second 8-row group
  // gives the other four the other half (checked exhaustively for row offsets 0..7; the plain stride is 2-way conflicted
  // on every fragment read: SQ_LDS_BANK_CONFLICT = 42 % of the LDS cycles of wgrad_kw_kernel).
  constexpr int GST = wg_row(32, SA), XST = wg_row(XR + 7, SB);
  // pieces are dealt to the producer waves in 64-piece chunks that are wholly G or wholly X, so that the descriptor of a load
  // is a wave-uniform choice (a per-lane choice would need both loads and a select on the DATA: an immediate vmcnt wait)
  constexpr int NGC = (32 * CA + 63) / 64, NXC = (XR * CB + 63) / 64, PER_P = (NGC + NXC + NPW - 1) / NPW;
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * (GST + XST)];
  bf16_t* Gs = lds;               // [2][GST]
  bf16_t* Xs = lds + 2 * GST;     // [2][XST]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int a0, b0, kh, kd, zsplit;
  {  // XCD-aware logical order: a-tile fastest, then (kd,kh), then b-tile, then split (89 % L2 hits instead of 44 %)
    const int ny = gridDim.y, total = gridDim.x * ny * gridDim.z;
    int lin = blockIdx.x + gridDim.x * (blockIdx.y + ny * blockIdx.z);
    { const int q = total / 8, r = total % 8, xcd = lin % 8, idx = lin / 8; lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int at = lin % p.nta; lin /= p.nta; const int y = lin % ny; lin /= ny; const int bt = lin % p.ntb; zsplit = lin / p.ntb;
    a0 = at * TA; b0 = bt * TB; kh = y % a.TH; kd = y / a.TH;
  }
  const int xrows = 32 + a.TW - 1;
  const long ubeg = (long)zsplit * p.units_per_split;
  long uend = ubeg + p.units_per_split; if (uend > p.units) uend = p.units;
  const int ntaps = a.TD * a.TH * a.TW;
  typedef __attribute__((address_space(3))) bf16x4 lds_b64;

  if (wave >= NC) {
    // ---------------- producer ----------------
    const int pw = wave - NC;
    const int gsw2 = (int)a.gsw * 2, gsh2 = (int)a.gsh * 2, xsw2 = (int)a.xsw * 2;
    unsigned pv[PER_P]; int pl[PER_P]; int prow[PER_P];  // byte offset at chunk 0 / LDS element offset in stage 0 (X: offset by 2*GST; -1 none) / G row (flat masks)
#pragma unroll
    for (int k = 0; k < PER_P; ++k) {
      const int chunk = pw + NPW * k;  // wave-uniform: chunks [0, NGC) are G pieces, [NGC, NGC + NXC) X pieces
      pv[k] = WG_OOB; pl[k] = -1; prow[k] = 0;
      if (chunk < NGC) {
        const int q = chunk * 64 + lane, row = q / CA, ch = q % CA, c = a0 + ch * 8;
        if (q < 32 * CA) {
          pl[k] = wg_row(row, SA) + ch * 8; prow[k] = row;
          if (c < a.alanes) pv[k] = (unsigned)((p.flat ? 0 : row * gsw2) + c * 2);
        }
      } else if (chunk < NGC + NXC) {
        const int q2 = (chunk - NGC) * 64 + lane, row = q2 / CB, ch = q2 % CB, c = b0 + ch * 8;
        if (q2 < xrows * CB) {
          pl[k] = 2 * GST + wg_row(row, SB) + ch * 8;
          if (c < a.blanes) pv[k] = (unsigned)(row * xsw2 + c * 2);
        }
      }
    }
    const int rhe = p.flat ? 1 : a.RH;
    int cwc, crh, crd, cn;
    {
      unsigned t = (unsigned)ubeg;
      cwc = (int)(t % (unsigned)p.wchunks); t /= (unsigned)p.wchunks;
      crh = (int)(t % (unsigned)rhe); t /= (unsigned)rhe;
      crd = (int)(t % (unsigned)a.RD); cn = (int)(t / (unsigned)a.RD);
    }
    u32x4 regsA[PER_P], regsB[PER_P];  // two units in flight through the registers (memory latency ~ one unit's MFMA time)
    long gu = ubeg;  // unit under the cursor
    auto gload = [&](u32x4 (&regs)[PER_P]) {  // unit under the cursor -> registers, then advance
      const bool live = gu < uend; ++gu;  // past the last unit the loads still issue (counted waits), but read nothing
      const int w0 = cwc * 32;
      const int id = crd * a.rs[0] + a.tb[0] + kd * a.ts[0];
      const bf16_t* gbase; const bf16_t* xbase; int nrecg, nrecx, gadd, xadd;
      if (p.flat) {
        const bool ok = (unsigned)id < (unsigned)a.ID;
        gbase = a.g + cn * a.gsn + crd * a.gsd; nrecg = a.RH * gsh2; gadd = 0;
        xbase = ok ? a.x + cn * a.xsn + (long)id * a.xsd : a.x; nrecx = ok ? a.IH * a.IW * xsw2 : 0;
        xadd = ((a.tb[1] + kh) * a.IW + a.tb[2] + w0) * xsw2;
      } else {
        const int ih = crh * a.rs[1] + a.tb[1] + kh * a.ts[1];
        const bool ok = (unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH;
        gbase = a.g + cn * a.gsn + crd * a.gsd + crh * a.gsh; nrecg = a.RW * gsw2; gadd = w0 * gsw2;
        xbase = ok ? a.x + cn * a.xsn + (long)id * a.xsd + (long)ih * a.xsh : a.x; nrecx = ok ? a.IW * xsw2 : 0;
        xadd = (w0 + a.tb[2]) * xsw2;
      }
      const i32x4_t rg = wg_rsrc(live ? gbase : a.g, live ? nrecg : 0), rx = wg_rsrc(live ? xbase : a.x, live ? nrecx : 0);
#pragma unroll
      for (int k = 0; k < PER_P; ++k) {
        const bool isg = pw + NPW * k < NGC;  // wave-uniform
        unsigned vo = pv[k] + (unsigned)(isg ? gadd : xadd);
        if (p.flat && isg) {
          const unsigned o = (unsigned)(w0 + prow[k]), h = __umulhi(o, p.pmagic), w = o - h * (unsigned)a.IW;
          vo = (w < (unsigned)a.RW && h < (unsigned)a.RH) ? pv[k] + h * (unsigned)gsh2 + w * (unsigned)gsw2 : WG_OOB;
        }
        wg_load16(regs[k], isg ? rg : rx, vo);
      }
      if (++cwc == p.wchunks) { cwc = 0; if (++crh == rhe) { crh = 0; if (++crd == a.RD) { crd = 0; ++cn; } } }
    };
    auto lstore = [&](const u32x4 (&regs)[PER_P], int buf) {
#pragma unroll
      for (int k = 0; k < PER_P; ++k) if (pl[k] >= 0) *(u32x4*)(lds + pl[k] + buf * (pl[k] >= 2 * GST ? XST : GST)) = regs[k];
    };
    // every gload issues exactly PER_P loads (lanes / chunks with nothing to fetch read out of range), so "all but the
    // PER_P youngest" retires the older set
    auto pbar = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my LDS stores are done; the loads of later units stay in flight
      __builtin_amdgcn_s_barrier();                        // unit u is visible to the consumers; the other stage is free
      asm volatile("" ::: "memory");
    };
    gload(regsA); wg_wait_loads<0>(regsA); lstore(regsA, 0);   // unit ubeg (cursor reads past the end are out of range -> zeros)
    gload(regsA);                                              // unit ubeg+1
    gload(regsB);                                              // unit ubeg+2
    for (long u = ubeg; u < uend; u += 2) {
      pbar();
      wg_wait_loads<PER_P>(regsA);                             // unit u+1 landed (unit u+2 may still fly)
      if (u + 1 < uend) lstore(regsA, 1);
      gload(regsA);                                            // unit u+3
      if (u + 1 < uend) {
        pbar();
        wg_wait_loads<PER_P>(regsB);                           // unit u+2
        if (u + 2 < uend) lstore(regsB, 0);
        gload(regsB);                                          // unit u+4
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // nothing of mine is in flight when the wave ends
    return;
  }

  // ---------------- consumer ----------------
  const int wa = wave / WB, wb = wave % WB;
  const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
  f32x4 acc[TWM][IA][IB];
#pragma unroll
  for (int t = 0; t < TWM; ++t)
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
      for (int j = 0; j < IB; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int goff = wg_row(8 * fg + fq, SA) + wa * IA * 16 + 4 * fp;
  int buf = 0;
  for (long u = ubeg; u < uend; ++u, buf ^= 1) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16x8 gf[IA];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const bf16_t* base = Gs + buf * GST + goff + i * 16;
      bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base));
      bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base + 4 * SA));  // rows +4: same 8-row group
      gf[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int t = 0; t < TWM; ++t) {
      if (t < a.TW) {
        const int r0 = t + 8 * fg + fq;
        const bf16_t* base = Xs + buf * XST + wb * IB * 16 + 4 * fp;
        const int o_lo = wg_row(r0, SB), o_hi = wg_row(r0 + 4, SB);
#pragma unroll
        for (int j = 0; j < IB; ++j) {
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base + o_lo + j * 16));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base + o_hi + j * 16));
          const bf16x8 xf = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int i = 0; i < IA; ++i) acc[t][i][j] = RG_MFMA16(gf[i], xf, acc[t][i][j]);
        }
      }
    }
  }
  float* out = a.part + (long)zsplit * a.A * ntaps * a.B;
#pragma unroll
  for (int t = 0; t < TWM; ++t) {
    if (t >= a.TW) continue;
    const int tap = (kd * a.TH + kh) * a.TW + t;
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
      for (int j = 0; j < IB; ++j) {
        const int bb = b0 + (wb * IB + j) * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int aa = a0 + (wa * IA + i) * 16 + fg * 4 + r;
          if (aa < a.A && bb < a.B) out[((long)aa * ntaps + tap) * a.B + bb] = acc[t][i][j][r];
        }
      }
  }
}

// row order and unit count of the kw-reuse kernel for this problem; false = not eligible
struct KwGeom { int flat; int wchunks; long units; };
static bool wgrad_kw_geom(const CL5& G, const CL5& X, const rg_conv_params* p, KwGeom* o) {
  if (p->rstride[2] != 1 || p->tstep[2] != 1 || p->T[2] < 2) return false;
  if ((long)X.H * X.sh * 2 >= (1L << 30) || (long)G.H * G.sh * 2 >= (1L << 30)) return false;  // 32-bit buffer offsets
  const int rw = p->R[2], rh = p->R[1];
  const long line_rows = (long)rh * round_up(rw, 32);
  bool flat_ok = p->rstride[1] == 1 && p->tstep[1] == 1 && X.sh == (long)X.W * X.sw && p->tbase[1] >= 0 &&
                 p->tbase[1] + p->T[1] - 1 + rh - 1 <= X.H - 1 && p->tbase[2] >= 0 && p->tbase[2] + p->T[2] - 1 + rw - 1 <= X.W - 1 &&
                 (double)X.H * X.W * X.W < 4.0e9;  // exact division by IW through the 32-bit magic multiply
  const long flat_rows = ((long)(rh - 1) * X.W + rw + 31) / 32 * 32;
  const bool flat = flat_ok && flat_rows < line_rows;
  const long rows = flat ? flat_rows : line_rows;
  if ((long)rh * rw * 2 < rows) return false;  // padding would waste more than half the MFMA work
  o->flat = flat;
  o->wchunks = flat ? (int)(flat_rows / 32) : (rw + 31) / 32;
  o->units = (long)G.N * p->R[0] * (flat ? 1 : rh) * o->wchunks;
  return true;
}
// which (FA, FB) tiles have a kw-reuse instantiation, and for how many W taps
static int wgrad_kw_kind(const CL5& G, const CL5& X, const rg_conv_params* p, int fa, int fb, KwGeom* geom) {
  if (!wgrad_kw_geom(G, X, p, geom)) return 0;
  if (fa == 9 && fb == 9 && p->T[2] <= 3) return 1;
  if (fa == 4 && fb == 4 && p->T[2] <= 8) return 2;
  if (fa == 8 && fb == 8 && p->T[2] <= 4) return 3;
  if (p->T[2] == 3) {  // the 72/84/504-channel U-Net levels and mixed tiles: one wave group per kw tap
    if (fa == 5 && fb == 5) return 4;
    if (fa == 9 && fb == 5) return 5;
    if (fa == 5 && fb == 8) return 6;
    if (fa == 1 && fb == 8) return 7;
    if (fa == 5 && fb == 1) return 8;
    if (fa == 1 && fb == 6) return 11;
    if (fa == 8 && fb == 4) return 9;
    if (fa == 4 && fb == 8) return 10;
  }
  return 0;
}

static int pick_f(int lanes) {
  // channel-tile menu (fragments of 16): 1, 4, 5, 8, 9
  if (lanes <= 16) return 1;
  if (lanes <= 64) return 4;
  if (lanes <= 80) return 5;
  if (lanes % 144 == 0) return 9;
  if (lanes % 128 == 0) return 8;
  const int w9 = round_up(lanes, 144) - lanes, w8 = round_up(lanes, 128) - lanes;
  return (w9 * 128 <= w8 * 144) ? 9 : 8;
}

static void wgrad_plan(const CL5& G, const CL5& X, const rg_conv_params* p, int* fa, int* fb, int* nsplit, long* rows_per_split) {
  *fa = pick_f(G.lanes); *fb = pick_f(X.lanes);
  // 84 -> 12 (the decoder's first convolution at the top level): a 96-lane X tile instead of 128, so that THREE of them — one workgroup owns the three kh taps, as the
  // <5,1> tile does — stay at 45 KB of LDS: 0.456 -> 0.39 ms (with 128-lane tiles the same scheme measured 0.50)
  if (*fa == 1 && X.lanes > 80 && X.lanes <= 96 && p->T[1] == 3 && p->T[2] == 3) *fb = 6;
  const long M = (long)G.N * p->R[0] * p->R[1] * p->R[2];
  const int ntaps = p->T[0] * p->T[1] * p->T[2];
  long tiles = (long)((G.C + *fa * 16 - 1) / (*fa * 16)) * ((X.C + *fb * 16 - 1) / (*fb * 16)) * ntaps;
  if (*fa == 1 && *fb == 1) tiles = (ntaps + WT_TMAX - 1) / WT_TMAX;  // wgrad_thin_kernel: all taps in one workgroup
  KwGeom kg;
  if (const int kind = wgrad_kw_kind(G, X, p, *fa, *fb, &kg)) {  // wgrad_kw_kernel: split over 32-row units
    const long units = kg.units;
    long tl = (long)((G.C + *fa * 16 - 1) / (*fa * 16)) * ((X.C + *fb * 16 - 1) / (*fb * 16)) * p->T[0] * p->T[1];
    if ((kind == 8 || kind == 11) && p->T[1] == 3) tl /= 3;   // one workgroup owns the three kh taps (KHL = 3): a third of the workgroups per split
    // workgroups aimed at: whole rounds of co-resident workgroups, rounded DOWN so that the grid never spills a few workgroups
    // into an extra round (256 -> 256 at 32^3: 36 tiles x 22 splits = 792 = 3.1 rounds ran at 668 TFLOP/s, x 21 = 756 at 815).
    // Kernels with several workgroups per CU: 768 (measured on the whole U-Net step, so that the partial-sum traffic and the
    // reduction passes count: 1536 -> 29.96 ms, 1024 -> 29.68, 768 -> 29.37, 512 -> 29.46; 256 starves the thin tiles: 29.1 vs 27.8).
    long ns = 768 / tl;
    if ((kind == 1 || kind == 3) && p->T[2] <= 3) {
      // producer-wave kernels, ONE workgroup per CU: 1, 2 or 3 rounds by a cost model — a round costs its units (~0.9 us each)
      // plus ~6 us of prologue / epilogue, every split writes and re-reads one fp32 copy of the gradient tile set (~3.5 TB/s
      // each way).  CycleGAN's 128 -> 128 ResNet-block layers at 32^3 (9 tiles, 1024 units, 1.8 MB of gradient): 85 splits
      // = 150 MB of partial sums, 148 us per call; 28 splits = one round: the step 96.4 -> 94.0 ms.
      const double out_mb = (double)(*fa * 16) * (*fb * 16) * p->T[2] * 4.0 * tl / 1e6;
      double best = 1e30;
      for (int r = 1; r <= 3; ++r) {
        long n = (long)r * 256 / tl; if (n > units / 16) n = units / 16; if (n < 1) n = 1;
        const long u = (units + n - 1) / n, n2 = (units + u - 1) / u;
        const double cost = (double)((tl * n2 + 255) / 256) * (u * 0.9 + 6.0) + (n2 > 1 ? n2 * out_mb * 2.0 / 3.5 : 0.0);
        if (cost < best) { best = cost; ns = n; }
      }
    }
    if (ns > units / 16) ns = units / 16; if (ns < 1) ns = 1;
    const long ups = (units + ns - 1) / ns;
    *nsplit = (int)((units + ups - 1) / ups);
    *rows_per_split = ups;
    return;
  }
  if (wgrad_lines_ok(G, X, p, *fa, *fb)) {  // wgrad_lines_kernel: split over (block of lines, 32-voxel chunk) units
    const long units = (long)G.N * ((p->R[0] + WB_LD - 1) / WB_LD) * ((p->R[1] + WB_LH - 1) / WB_LH) * ((p->R[2] + 31) / 32);
    long ns = 1024; if (ns > units / 4) ns = units / 4; if (ns < 1) ns = 1;   // two workgroups per CU: two rounds
    const long ups = (units + ns - 1) / ns;
    *nsplit = (int)((units + ups - 1) / ups);
    *rows_per_split = ups;
    return;
  }
  if (wgrad_line_ok(G, X, p, *fa, *fb)) {  // wgrad_line_kernel: split over (line, 32-voxel chunk) units
    const long wch = (p->R[2] + 31) / 32;
    const long units = (long)G.N * p->R[0] * p->R[1] * wch;
    const long groups = (p->T[0] * p->T[1] + WL_MAXJ - 1) / WL_MAXJ;
    long ns = 2048 / groups; if (ns > units / 8) ns = units / 8; if (ns < 1) ns = 1;
    const long ups = (units + ns - 1) / ns;
    *nsplit = (int)((units + ups - 1) / ups);
    *rows_per_split = ups;  // units per split
    return;
  }
  long want = 2048 / tiles;                         // aim at one round of 2048 co-resident workgroups, never a few more (rounded down)
  long maxs = (M + 64 * 8 - 1) / (64 * 8);           // at least 8 K-steps per split
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  long rps = ((M + want - 1) / want + 63) / 64 * 64;
  *nsplit = (int)((M + rps - 1) / rps);
  *rows_per_split = rps;
}

size_t rg_wgrad_gemm_bytes(const CL5& G, const CL5& X, const rg_conv_params* p);                                                   // gemm5.hip
int rg_launch_wgrad_gemm(const CL5& G, const CL5& X, const rg_conv_params* p, float* dw, void* workspace, size_t ws_bytes, hipStream_t st);
static int wgrad_check(const rg_tensor* g, const rg_tensor* x, const rg_conv_params* p, CL5* G, CL5* X) {
  int rc;
  if ((rc = rg_as_cl5(g, G, "wgrad.g")) || (rc = rg_as_cl5(x, X, "wgrad.x"))) return rc;
  if (!p) RG_FAIL(RG_ERR_BAD_SHAPE, "wgrad: null params");
  if (G->D != p->R[0] || G->H != p->R[1] || G->W != p->R[2] || G->N != X->N)
    RG_FAIL(RG_ERR_BAD_SHAPE, "wgrad: g spatial dims (%d,%d,%d) must equal the row grid (%d,%d,%d)", G->D, G->H, G->W, p->R[0], p->R[1], p->R[2]);
  return RG_OK;
}

extern "C" int64_t rg_conv3d_wgrad_workspace_bytes(const rg_tensor* g, const rg_tensor* x, const rg_conv_params* p, int32_t* nsplit_out) {
  CL5 G, X;
  if (wgrad_check(g, x, p, &G, &X)) return -1;
  int fa, fb, ns; long rps;
  wgrad_plan(G, X, p, &fa, &fb, &ns, &rps);
  if (nsplit_out) *nsplit_out = ns;
  const int64_t total = (int64_t)G.C * X.C * p->T[0] * p->T[1] * p->T[2];
  const int64_t split_bytes = (ns + (ns > 8 ? (ns + WF_FOLD - 1) / WF_FOLD : 0)) * total * 4;  // (an upper bound of what rg_conv3d_wgrad2 uses)
  const int64_t gemm_bytes = (int64_t)rg_wgrad_gemm_bytes(G, X, p);   // the GEMM form (gemm5.hip): two transposed operand copies instead of partial sums
  return gemm_bytes > split_bytes ? gemm_bytes : split_bytes;
}

template <int FA, int FB, int WA, int WB>
static void launch_wgrad(const WgradArgs& a, hipStream_t s) {
  const int nta = (a.A + FA * 16 - 1) / (FA * 16), ntb = (a.B + FB * 16 - 1) / (FB * 16);
  dim3 grid(nta * ntb, a.TD * a.TH * a.TW, a.nsplit);
  rg_set_kernel_name("wgrad_kernel<%d,%d,%d,%d>", FA, FB, WA, WB);
  hipLaunchKernelGGL((wgrad_kernel<FA, FB, WA, WB>), grid, dim3(WA * WB * 64), 0, s, a);
}

#define WG_ROW(FA_, WA_)                                                        \
  switch (fb) {                                                                 \
    case 1: launch_wgrad<FA_, 1, WA_, 1>(a, st); break;                         \
    case 4: launch_wgrad<FA_, 4, WA_, 2>(a, st); break;                         \
    case 5: launch_wgrad<FA_, 5, WA_, 1>(a, st); break;                         \
    case 8: launch_wgrad<FA_, 8, WA_, 2>(a, st); break;                         \
    default: launch_wgrad<FA_, 9, WA_, 3>(a, st); break;                        \
  }

extern "C" int rg_conv3d_wgrad(const rg_tensor* g, const rg_tensor* x, float* dw, const rg_conv_params* p, int accumulate,
                               void* workspace, size_t ws_bytes, rg_stream s) {
  return rg_conv3d_wgrad2(g, x, dw, p, accumulate ? 1 : 0, workspace, ws_bytes, s);
}
// out[i] = (accumulate ? out[i] : 0) + sum_s part[s][i]: the final pass when dw lives in the kernels' own [a][tap][b] order
__global__ void wgrad_sum_kernel(const float* part, float* out, long total, int nsplit, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float v = accumulate ? out[i] : 0.f;
  for (int s = 0; s < nsplit; ++s) v += part[s * total + i];
  out[i] = v;
}
// The same sum, four elements per thread and four splits per workgroup row: block = 64 element lanes x 4 split lanes, every
// thread keeps up to four 16-byte loads in flight; split lane y sums splits y, y+4, ... in order, lane 0 then adds the four
// partial sums in order — a fixed association, deterministic like every reduction here.  One pass for up to 64 splits.
__global__ __launch_bounds__(256) void wgrad_sum4_kernel(const float4* part, float4* out, long total4, int nsplit, int accumulate) {
  __shared__ float4 red[3][64];
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + lx;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < total4) {
    int sp = ly;
    for (; sp + 12 < nsplit; sp += 16) {
      const float4 a = part[(long)sp * total4 + i], b = part[(long)(sp + 4) * total4 + i], c = part[(long)(sp + 8) * total4 + i],
                   d = part[(long)(sp + 12) * total4 + i];
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w;
      v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
    }
    for (; sp < nsplit; sp += 4) { const float4 a = part[(long)sp * total4 + i]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
  }
  if (ly) red[ly - 1][lx] = v;
  __syncthreads();
  if (ly == 0 && i < total4) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float4 a = red[k][lx]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    if (accumulate) { const float4 o = out[i]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    out[i] = v;
  }
}
extern "C" int rg_conv3d_wgrad2(const rg_tensor* g, const rg_tensor* x, float* dw, const rg_conv_params* p, int flags,
                                void* workspace, size_t ws_bytes, rg_stream s) {
  const int accumulate = flags & 1, dw_cl = (flags >> 1) & 1;
  CL5 G, X;
  int rc;
  if ((rc = wgrad_check(g, x, p, &G, &X))) return rc;
  if (dw_cl && !accumulate && dw && workspace && !rg_opt(RG_OPT_NO_WGEMM)) {   // wide layers: transposed operands + 27 plain GEMMs on igemm5 (gemm5.hip)
    const size_t gb = rg_wgrad_gemm_bytes(G, X, p);
    if (gb && ws_bytes >= gb) { rc = rg_launch_wgrad_gemm(G, X, p, dw, workspace, ws_bytes, (hipStream_t)s); if (rc) return rc; RG_CHECK_LAUNCH(); return RG_OK; }
  }
  int fa, fb, ns; long rps;
  wgrad_plan(G, X, p, &fa, &fb, &ns, &rps);
  const int ntaps = p->T[0] * p->T[1] * p->T[2];
  const size_t total_el = (size_t)G.C * X.C * ntaps;
  // channels-last dw with a multiple of four elements: one vectorised pass sums up to 64 splits (wgrad_sum4_kernel); otherwise,
  // and for the thin layers' hundreds of splits, fold groups of 16 first (serial split loops in the finish are latency-bound)
  const bool sum4 = dw_cl && (total_el & 3) == 0 && (((uintptr_t)dw | (uintptr_t)workspace) & 15) == 0;
  const int nfold = (sum4 ? ns > 64 : ns > 8) ? (ns + WF_FOLD - 1) / WF_FOLD : 0;
  const size_t need = (size_t)(ns + nfold) * total_el * 4;
  if (!workspace || ws_bytes < need) RG_FAIL(RG_ERR_WORKSPACE, "wgrad: workspace %zu < %zu bytes", ws_bytes, need);
  if (!dw) RG_FAIL(RG_ERR_BAD_SHAPE, "wgrad: null dw");
  WgradArgs a;
  const bool direct = dw_cl && ns == 1 && !accumulate;  // the single split's "partial" sums are dw itself
  a.g = G.p; a.x = X.p; a.part = direct ? dw : (float*)workspace;
  a.N = X.N; a.ID = X.D; a.IH = X.H; a.IW = X.W;
  a.RD = p->R[0]; a.RH = p->R[1]; a.RW = p->R[2];
  a.TD = p->T[0]; a.TH = p->T[1]; a.TW = p->T[2];
  for (int d = 0; d < 3; ++d) { a.rs[d] = p->rstride[d]; a.tb[d] = p->tbase[d]; a.ts[d] = p->tstep[d]; }
  a.gsn = G.sn; a.gsd = G.sd; a.gsh = G.sh; a.gsw = G.sw;
  a.xsn = X.sn; a.xsd = X.sd; a.xsh = X.sh; a.xsw = X.sw;
  a.A = G.C; a.B = X.C; a.alanes = G.lanes; a.blanes = X.lanes;
  a.M = (long)G.N * a.RD * a.RH * a.RW; a.rows_per_split = rps; a.nsplit = ns;
  hipStream_t st = (hipStream_t)s;
  rg_prof_mark(st);
  KwGeom kg;
  if (const int kind = wgrad_kw_kind(G, X, p, fa, fb, &kg)) {
    WgradKwArgs ka; ka.w = a;
    ka.wchunks = kg.wchunks; ka.units = kg.units; ka.flat = kg.flat;
    ka.pmagic = (unsigned)((1ULL << 32) / (unsigned)X.W + 1);
    ka.units_per_split = rps;
    ka.dbg = 0;
    ka.nta = (a.A + fa * 16 - 1) / (fa * 16); ka.ntb = (a.B + fb * 16 - 1) / (fb * 16);
    dim3 grid(ka.nta * ka.ntb, a.TD * a.TH, ns);
    rg_set_kernel_name("wgrad_kw_kernel<%d,%d>", fa, fb);
    if (kind == 1 && !rg_opt(RG_OPT_NO_KWP)) {
      rg_set_kernel_name("wgrad_kwp_kernel<%d,%d>", fa, fb);
      hipLaunchKernelGGL((wgrad_kwp_kernel<9, 9, 3, 3, 3, 3>), grid, dim3(768), 0, st, ka);
    } else if (kind == 1) hipLaunchKernelGGL((wgrad_kw_kernel<9, 9, 3, 3, 1, 3, false>), grid, dim3(576), 0, st, ka);
    else if (kind == 2) hipLaunchKernelGGL((wgrad_kw_kernel<4, 4, 2, 2, 2, 8, true>), grid, dim3(512), 0, st, ka);
    else if (kind == 3 && a.TW <= 3 && !rg_opt(RG_OPT_NO_KWP)) {
      rg_set_kernel_name("wgrad_kwp_kernel<%d,%d>", fa, fb);
      hipLaunchKernelGGL((wgrad_kwp_kernel<8, 8, 2, 4, 3, 4>), grid, dim3(768), 0, st, ka);
    } else if (kind == 3) hipLaunchKernelGGL((wgrad_kw_kernel<8, 8, 2, 4, 1, 4, true>), grid, dim3(512), 0, st, ka);
    else if (kind == 4 && !rg_opt(RG_OPT_NO_KWP)) { rg_set_kernel_name("wgrad_kwp_kernel<%d,%d>", fa, fb); hipLaunchKernelGGL((wgrad_kwp_kernel<5, 5, 5, 1, 3, 3>), grid, dim3(512), 0, st, ka); }
    else if (kind == 4) hipLaunchKernelGGL((wgrad_kw_kernel<5, 5, 1, 1, 3, 3, true>), grid, dim3(192), 0, st, ka);
    else if (kind == 5) hipLaunchKernelGGL((wgrad_kw_kernel<9, 5, 3, 1, 3, 3, true>), grid, dim3(576), 0, st, ka);
    else if (kind == 6) hipLaunchKernelGGL((wgrad_kw_kernel<5, 8, 1, 2, 3, 3, true>), grid, dim3(384), 0, st, ka);
    else if (kind == 7) hipLaunchKernelGGL((wgrad_kw_kernel<1, 8, 1, 2, 3, 3, true>), grid, dim3(384), 0, st, ka);
    else if (kind == 11) hipLaunchKernelGGL((wgrad_kw_kernel<1, 6, 1, 2, 3, 3, true, 3>), dim3(grid.x, a.TD, ns), dim3(384), 0, st, ka);
    else if (kind == 8 && a.TH == 3) hipLaunchKernelGGL((wgrad_kw_kernel<5, 1, 1, 1, 3, 3, true, 3>), dim3(grid.x, a.TD, ns), dim3(192), 0, st, ka);
    else if (kind == 8) hipLaunchKernelGGL((wgrad_kw_kernel<5, 1, 1, 1, 3, 3, true>), grid, dim3(192), 0, st, ka);
    else if (kind == 9) hipLaunchKernelGGL((wgrad_kw_kernel<8, 4, 2, 2, 1, 4, true>), grid, dim3(256), 0, st, ka);
    else hipLaunchKernelGGL((wgrad_kw_kernel<4, 8, 2, 2, 1, 4, true>), grid, dim3(256), 0, st, ka);
  } else if (wgrad_lines_ok(G, X, p, fa, fb)) {
    WgradLinesArgs la; la.w = a;
    la.wchunks = (a.RW + 31) / 32; la.dgroups = (a.RD + WB_LD - 1) / WB_LD; la.hgroups = (a.RH + WB_LH - 1) / WB_LH;
    la.units = (long)G.N * la.dgroups * la.hgroups * la.wchunks;
    la.units_per_split = rps;
    la.xbytes = (unsigned)(wg_extent(X) * 2); la.gbytes = (unsigned)(wg_extent(G) * 2);
    constexpr size_t shm = 2 * (size_t)(WB_LD * WB_LH * 32 + (WB_LD + 2) * (WB_LH + 2) * 34) * 16 * sizeof(bf16_t);
    static rg_device_latch attr_lines;
    rg_once_per_device(&attr_lines, [] { hipFuncSetAttribute((const void*)wgrad_lines_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             (int)(2 * (size_t)(WB_LD * WB_LH * 32 + (WB_LD + 2) * (WB_LH + 2) * 34) * 16 * sizeof(bf16_t))); });
    rg_set_kernel_name("wgrad_lines_kernel");
    hipLaunchKernelGGL(wgrad_lines_kernel, dim3(1, 1, ns), dim3(256), shm, st, la);
  } else if (wgrad_line_ok(G, X, p, fa, fb)) {
    WgradLineArgs la; la.w = a;
    la.wchunks = (a.RW + 31) / 32;
    la.units = (long)G.N * a.RD * a.RH * la.wchunks;
    la.units_per_split = rps;
    la.xbytes = (unsigned)(wg_extent(X) * 2);
    bool check = false;
    const int I[3] = {X.D, X.H, X.W};
    for (int d = 0; d < 3; ++d) if (p->tbase[d] < 0 || p->tbase[d] + p->T[d] - 1 + p->R[d] - 1 >= I[d]) check = true;
    rg_set_kernel_name("wgrad_line_kernel<%d>", (int)check);
    const dim3 grid((a.TD * a.TH + WL_MAXJ - 1) / WL_MAXJ, 1, ns);
    if (a.TW <= 3) {
      if (check) hipLaunchKernelGGL((wgrad_line_kernel<true, 3>), grid, dim3(256), 0, st, la);
      else hipLaunchKernelGGL((wgrad_line_kernel<false, 3>), grid, dim3(256), 0, st, la);
    } else {
      if (check) hipLaunchKernelGGL((wgrad_line_kernel<true, WL_MAXW>), grid, dim3(256), 0, st, la);
      else hipLaunchKernelGGL((wgrad_line_kernel<false, WL_MAXW>), grid, dim3(256), 0, st, la);
    }
  } else if (fa == 1 && fb == 1) {
    rg_set_kernel_name("wgrad_thin_kernel");
    hipLaunchKernelGGL(wgrad_thin_kernel, dim3((ntaps + WT_TMAX - 1) / WT_TMAX, 1, ns), dim3(256), 0, st, a);
  } else
  switch (fa) {
    case 1: WG_ROW(1, 1); break;
    case 4: WG_ROW(4, 2); break;
    case 5: WG_ROW(5, 1); break;
    case 8: WG_ROW(8, 2); break;
    default: WG_ROW(9, 1); break;  // 144x144 tile as 1x3 waves of 144x48: 12 fragment reads per 27 MFMAs
  }
  rg_prof_mark(st);
  RG_CHECK_LAUNCH();
  if (direct) return RG_OK;
  const float* fin = (const float*)workspace;
  int nfin = ns;
  if (nfold) {
    float* folded = (float*)workspace + (size_t)ns * total_el;
    hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)((total_el + 255) / 256), nfold), dim3(256), 0, st, (const float*)workspace, folded, (long)total_el, ns);
    RG_CHECK_LAUNCH();
    fin = folded; nfin = nfold;
  }
  if (sum4) hipLaunchKernelGGL(wgrad_sum4_kernel, dim3((unsigned)((total_el / 4 + 63) / 64)), dim3(256), 0, st, (const float4*)fin, (float4*)dw, (long)(total_el / 4), nfin, accumulate);
  else if (dw_cl) hipLaunchKernelGGL(wgrad_sum_kernel, dim3((unsigned)((total_el + 255) / 256)), dim3(256), 0, st, fin, dw, (long)total_el, nfin, accumulate);
  else hipLaunchKernelGGL(wgrad_finish_kernel, dim3(G.C * ((X.C + 63) / 64)), dim3(256), 0, st, fin, dw, G.C, X.C, ntaps, nfin, accumulate);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
