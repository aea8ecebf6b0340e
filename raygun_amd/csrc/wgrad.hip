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

struct RowInfo { long goff; long xoff; };  // -1 = row contributes zeros

constexpr int wg_pad(int t) { return ((t / 16) & 1) ? t : t + 16; }

template <int FA, int FB, int WA, int WB>
__global__ __launch_bounds__(WA* WB * 64) void wgrad_kernel(WgradArgs a) {
  constexpr int NT = WA * WB * 64;
  constexpr int TA = FA * 16, TB = FB * 16;
  constexpr int SA = wg_pad(TA), SB = wg_pad(TB);
  constexpr int CA = TA / 8, CB = TB / 8;       // 16-byte chunks per row
  constexpr int NCH = 32 * (CA + CB);
  constexpr int PER_T = (NCH + NT - 1) / NT;
  constexpr int IA = FA / WA, IB = FB / WB;     // fragments per wave
  static_assert(FA % WA == 0 && FB % WB == 0, "wave layout must divide the tile");
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * 32 * (SA + SB)];
  __shared__ RowInfo ri[2][32];
  bf16_t* Gs = lds;                  // [2][32][SA]
  bf16_t* Xs = lds + 2 * 32 * SA;    // [2][32][SB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave / WB, wb = wave % WB;
  const int ntb = (a.B + TB - 1) / TB;
  const int a0 = (blockIdx.x / ntb) * TA, b0 = (blockIdx.x % ntb) * TB;
  const int tap = blockIdx.y;
  const int kw = tap % a.TW, kh = (tap / a.TW) % a.TH, kd = tap / (a.TW * a.TH);
  const int od = a.tb[0] + kd * a.ts[0], oh = a.tb[1] + kh * a.ts[1], ow = a.tb[2] + kw * a.ts[2];
  const long mbeg = (long)blockIdx.z * a.rows_per_split;
  long mend = mbeg + a.rows_per_split; if (mend > a.M) mend = a.M;
  const int nsteps = (int)((mend - mbeg + 31) / 32);

  f32x4 acc[IA][IB];
#pragma unroll
  for (int i = 0; i < IA; ++i)
#pragma unroll
    for (int j = 0; j < IB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto rowinfo = [&](int step) {
    if (tid < 32) {
      const long m = mbeg + (long)step * 32 + tid;
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
      if (q < 32 * CA) {
        const int row = q / CA, c = a0 + (q % CA) * 8;
        const long off = ri[step & 1][row].goff;
        if (off >= 0 && c < a.alanes) v = *(const bf16x8*)(a.g + off + c);
      } else if (q < NCH) {
        const int q2 = q - 32 * CA;
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
      if (q < 32 * CA) {
        *(bf16x8*)(Gs + (buf * 32 + q / CA) * SA + (q % CA) * 8) = regs[i];
      } else if (q < NCH) {
        const int q2 = q - 32 * CA;
        *(bf16x8*)(Xs + (buf * 32 + q2 / CB) * SB + (q2 % CB) * 8) = regs[i];
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
    bf16x8 gf[IA], xf[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const bf16_t* base = Gs + (buf * 32 + 8 * fg + fq) * SA + (wa * IA + i) * 16 + 4 * fp;
      bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base));
      bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base + 4 * SA));
      gf[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int j = 0; j < IB; ++j) {
      const bf16_t* base = Xs + (buf * 32 + 8 * fg + fq) * SB + (wb * IB + j) * 16 + 4 * fp;
      bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base));
      bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64*)(base + 4 * SB));
      xf[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
      for (int j = 0; j < IB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i], xf[j], acc[i][j], 0, 0, 0);
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

// dw[a][b][tap] (+)= sum_s part[s][a][tap][b]
__global__ void wgrad_finish_kernel(const float* part, float* dw, int A, int B, int ntaps, int nsplit, int accumulate) {
  const long total = (long)A * ntaps * B;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += part[s * total + i];
    const int b = (int)(i % B); long t = i / B;
    const int tap = (int)(t % ntaps); const int aa = (int)(t / ntaps);
    const long o = ((long)aa * B + b) * ntaps + tap;
    dw[o] = accumulate ? dw[o] + v : v;
  }
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
  const long M = (long)G.N * p->R[0] * p->R[1] * p->R[2];
  const int ntaps = p->T[0] * p->T[1] * p->T[2];
  const long tiles = (long)((G.C + *fa * 16 - 1) / (*fa * 16)) * ((X.C + *fb * 16 - 1) / (*fb * 16)) * ntaps;
  long want = (2048 + tiles - 1) / tiles;           // aim at >= 2048 workgroups
  long maxs = (M + 32 * 16 - 1) / (32 * 16);         // at least 16 K-steps per split
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  long rps = ((M + want - 1) / want + 31) / 32 * 32;
  *nsplit = (int)((M + rps - 1) / rps);
  *rows_per_split = rps;
}

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
  return (int64_t)ns * G.C * X.C * p->T[0] * p->T[1] * p->T[2] * 4;
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
  CL5 G, X;
  int rc;
  if ((rc = wgrad_check(g, x, p, &G, &X))) return rc;
  int fa, fb, ns; long rps;
  wgrad_plan(G, X, p, &fa, &fb, &ns, &rps);
  const int ntaps = p->T[0] * p->T[1] * p->T[2];
  const size_t need = (size_t)ns * G.C * X.C * ntaps * 4;
  if (!workspace || ws_bytes < need) RG_FAIL(RG_ERR_WORKSPACE, "wgrad: workspace %zu < %zu bytes", ws_bytes, need);
  if (!dw) RG_FAIL(RG_ERR_BAD_SHAPE, "wgrad: null dw");
  WgradArgs a;
  a.g = G.p; a.x = X.p; a.part = (float*)workspace;
  a.N = X.N; a.ID = X.D; a.IH = X.H; a.IW = X.W;
  a.RD = p->R[0]; a.RH = p->R[1]; a.RW = p->R[2];
  a.TD = p->T[0]; a.TH = p->T[1]; a.TW = p->T[2];
  for (int d = 0; d < 3; ++d) { a.rs[d] = p->rstride[d]; a.tb[d] = p->tbase[d]; a.ts[d] = p->tstep[d]; }
  a.gsn = G.sn; a.gsd = G.sd; a.gsh = G.sh; a.gsw = G.sw;
  a.xsn = X.sn; a.xsd = X.sd; a.xsh = X.sh; a.xsw = X.sw;
  a.A = G.C; a.B = X.C; a.alanes = G.lanes; a.blanes = X.lanes;
  a.M = (long)G.N * a.RD * a.RH * a.RW; a.rows_per_split = rps; a.nsplit = ns;
  hipStream_t st = (hipStream_t)s;
  switch (fa) {
    case 1: WG_ROW(1, 1); break;
    case 4: WG_ROW(4, 2); break;
    case 5: WG_ROW(5, 1); break;
    case 8: WG_ROW(8, 2); break;
    default: WG_ROW(9, 3); break;
  }
  RG_CHECK_LAUNCH();
  const long total = (long)G.C * X.C * ntaps;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, G.C, X.C, ntaps, ns, accumulate);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
