// igemm3 — wide-channel implicit-GEMM convolution with kw REUSE of the activation tile (row stride 1 and tap step
// +-1 along W: every stride-1 Conv3d forward / backward-data and the stride-1 parts of the others).
//
// igemm2 stages a fresh 128-row activation tile for every tap although the kW taps of one (kd,kh) read the same input
// line shifted by one voxel; its global->LDS traffic (measured ~30 GB/s per CU) is what bounds it, not the MFMAs.  Here a
// tile is L whole output-line segments (L*len <= 128 rows); for one (channel chunk, kd, kh) the L input line segments of
// len+kW-1 voxels are staged ONCE ("A' tile") and the kW taps read them at row offset kw through per-lane ds_read
// addresses.  Loads per K-step drop from 4+BPW to 4/kW+BPW pieces per wave (7^3 64->64: 6 -> 2.6; 3^3 144-wide: 9 -> 6.4).
// Everything else follows igemm2: 64-channel K-steps, register-staged operands, double-buffered LDS, XOR-swizzled
// 16-byte slots ((row>>1)&7, keyed on the A' row), hand-counted fragment reads two ahead of the MFMAs, zero page for
// out-of-range lanes, two workgroups per CU.
#include "common.h"
#include <utility>

template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

struct Igemm3Args {
  const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y; const bf16_t* zeros;
  int N, ID, IH, IW;
  int RD, RH, RW;
  int TD, TH, TW;
  int rs[3], tb[3], ts[3], os[3], oo[3];
  long xsn, xsd, xsh, xsw;
  long ysn, ysd, ysh, ysw;
  int cink, cin_lanes, cout, cout_lanes, wrows;
  int act; float slope;
  int L, slen, xw, segs_per_line, nseg;  // lines per tile, segment length, staged rows per segment, segments per line, total
  int wbase;                              // input w of A' row 0 relative to the segment start (tb_w, or tb_w-(TW-1) for ts=-1)
  int tiles_m, tiles_n;
};

typedef __attribute__((address_space(3))) void* lptr3_t;
template <int OFF>
__device__ __forceinline__ void ld16(bf16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF)); }
template <int N>
__device__ __forceinline__ void ldwait(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }

constexpr int AROWS = 160;  // max staged activation rows (128 + 32 halo rows): 2 x 40 KB A + 2 x 40 KB B = 80 KB -> two workgroups per CU

template <int FN, bool CHECK>
__global__ __launch_bounds__(256, 2) void igemm3_kernel(Igemm3Args a) {
  constexpr int BM = 128, BN = FN * 16, FM = 2;
  constexpr int BPIECES = ((BN / 8) + 3) / 4 * 4, BROWS = BPIECES * 8, BPW = BPIECES / 4;
  constexpr int APIECES = AROWS / 8, APW = APIECES / 4;      // 20 pieces, 5 per wave
  constexpr int ABUF = AROWS * 64, BBUF = BROWS * 64;         // bf16 elements
  __shared__ __attribute__((aligned(128))) bf16_t lds[2 * ABUF + 2 * BBUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = a.tiles_m * a.tiles_n;
  int bid = blockIdx.x;
  { const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
  const int tile_n = bid / a.tiles_m, tile_m = bid % a.tiles_m;
  const int n0 = tile_n * BN;
  const int seg0 = tile_m * a.L;
  const int lrow = lane >> 3, lslot = lane & 7;
  const int arows = a.L * a.xw;  // staged rows actually used

  // ---- staged activation rows of this lane: piece p = wave*APW + i -> A' row q = 8p + lrow = (segment l, offset ww)
  int abase[APW], ad0[APW], ah0[APW], aw0[APW]; unsigned avalid = 0;
#pragma unroll
  for (int i = 0; i < APW; ++i) {
    const int q = (wave * APW + i) * 8 + lrow;
    const int l = q / a.xw, ww = q - l * a.xw;
    const int seg = seg0 + l;
    const bool v = q < arows && seg < a.nseg;
    unsigned t = v ? (unsigned)seg : 0u;
    const int sc = (int)(t % (unsigned)a.segs_per_line); t /= (unsigned)a.segs_per_line;
    const int rh = (int)(t % (unsigned)a.RH); t /= (unsigned)a.RH;
    const int rd = (int)(t % (unsigned)a.RD); const int n = (int)(t / (unsigned)a.RD);
    ad0[i] = rd * a.rs[0] + a.tb[0]; ah0[i] = rh * a.rs[1] + a.tb[1]; aw0[i] = sc * a.slen + a.wbase + ww;
    avalid |= (unsigned)(v && (unsigned)aw0[i] < (unsigned)a.IW) << i;  // the W bound does not depend on the tap
    abase[i] = (int)(n * a.xsn + ad0[i] * a.xsd + ah0[i] * a.xsh + aw0[i] * a.xsw);
  }
  const int sw_even = (lslot ^ (lrow >> 1)) * 8, sw_odd = (lslot ^ (4 + (lrow >> 1))) * 8;
  const int ntaps = a.TD * a.TH * a.TW;
  int wb[BPW]; unsigned wvalid = 0;
#pragma unroll
  for (int i = 0; i < BPW; ++i) {
    const int row = (wave * BPW + i) * 8 + lrow;
    const int n = n0 + row;
    const bool v = row < BN && n < a.wrows;
    wvalid |= (unsigned)v << i;
    wb[i] = v ? n * ntaps * a.cink : 0;
  }
  const int nchunk = a.cink >> 6;
  const int ngroups = nchunk * a.TD * a.TH;   // one group = (chunk, kd, kh): kW K-steps sharing one staged A' tile
  const int nk = ngroups * a.TW;

  // ---- producers.  B: one K-step ahead in registers.  A': the next group's pieces, spread over the first TW-1 steps.
  bf16x8 breg[BPW], areg[APW];
  int bchunk = 0, bkd = 0, bkh = 0, bkw = 0;      // K-step whose weights are loaded next
  auto bload = [&]() {
    const int tap = (bkd * a.TH + bkh) * a.TW + bkw;
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
      const bool ok = ((wvalid >> i) & 1) & (bchunk < nchunk);
      const int c = bchunk * 64 + (((wave * BPW + i) & 1) ? sw_odd : sw_even);
      const bf16_t* p = (ok ? a.w : a.zeros) + (ok ? wb[i] + tap * a.cink + c : lslot * 8);
      breg[i] = *(const bf16x8*)p;
    }
    ++bkw;
    const int cw = bkw == a.TW; bkw = cw ? 0 : bkw; bkh += cw;
    const int ch = bkh == a.TH; bkh = ch ? 0 : bkh; bkd += ch;
    const int cd = bkd == a.TD; bkd = cd ? 0 : bkd; bchunk += cd;
  };
  auto bstore = [&](int buf) {
    bf16_t* Bs = lds + 2 * ABUF + buf * BBUF + lane * 8;
#pragma unroll
    for (int i = 0; i < BPW; ++i) *(bf16x8*)(Bs + (wave * BPW + i) * 512) = breg[i];
  };
  int achunk = 0, akd = 0, akh = 0;               // group whose activations are loaded next
  auto aload_piece = [&](int i) {
    const int od = akd * a.ts[0], oh = akh * a.ts[1];
    const int c = achunk * 64 + (((wave * APW + i) & 1) ? sw_odd : sw_even);
    bool ok = ((avalid >> i) & 1) & (c < a.cin_lanes) & (achunk < nchunk);
    if (CHECK) ok &= ((unsigned)(ad0[i] + od) < (unsigned)a.ID) & ((unsigned)(ah0[i] + oh) < (unsigned)a.IH);
    const bf16_t* p = (ok ? a.x : a.zeros) + (ok ? abase[i] + (int)(od * a.xsd + oh * a.xsh) + c : lslot * 8);
    areg[i] = *(const bf16x8*)p;
  };
  auto aadvance = [&]() {
    ++akh;
    const int ch = akh == a.TH; akh = ch ? 0 : akh; akd += ch;
    const int cd = akd == a.TD; akd = cd ? 0 : akd; achunk += cd;
  };
  auto astore_piece = [&](int i, int buf) {
    *(bf16x8*)(lds + buf * ABUF + (wave * APW + i) * 512 + lane * 8) = areg[i];
  };
  // pieces of the next A' tile handled at sub-step kw (kw = 0 .. TW-2): i with i % (TW-1) == kw
  const int nsub = a.TW - 1;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: A'(group 0) and B(step 0) into LDS, B(step 1) into registers
#pragma unroll
  for (int i = 0; i < APW; ++i) aload_piece(i);
  aadvance();
#pragma unroll
  for (int i = 0; i < APW; ++i) astore_piece(i, 0);
  bload(); bstore(0);
  bload();

  // ---- consumer addressing: fragment row r = wave*32 + i*16 + fr -> A' row q = l*xw + w (+ tap shift)
  const int fr = lane & 15, fg = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(lptr3_t)lds;
  int qrow[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int r = wave * 32 + i * 16 + fr;
    const int l = r / a.slen, w = r - l * a.slen;
    qrow[i] = (l < a.L ? l : a.L - 1) * a.xw + w;
    if (l >= a.L) qrow[i] = 0;  // padding rows of the tile: any staged row will do, the result is discarded
  }
  const unsigned b_lane = lds0 + 2 * ABUF * 2 + fr * 128 + ((fg ^ (fr >> 1)) << 4);
  constexpr int T2 = 2 * FN;

  int kw = 0, grp = 0;
  for (int ks = 0; ks < nk; ++ks) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int abuf = grp & 1, bbuf = ks & 1;
    // weights of step ks+1: registers -> LDS; then fetch step ks+2
    bstore(bbuf ^ 1);
    bload();
    // activations of group grp+1: pieces fetched during the previous sub-step -> LDS; fetch this sub-step's pieces
    if (kw > 0) {
#pragma unroll
      for (int i = 0; i < APW; ++i) if (i % nsub == kw - 1) astore_piece(i, abuf ^ 1);
    }
    if (kw < nsub) {
#pragma unroll
      for (int i = 0; i < APW; ++i) if (i % nsub == kw) aload_piece(i);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // fragment addresses for this tap: A' row q + shift, slot swizzled by the A' row
    const int shift = a.ts[2] > 0 ? kw : a.TW - 1 - kw;
    unsigned a0[FM], a1[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int q = qrow[i] + shift;
      const unsigned base = lds0 + abuf * (ABUF * 2) + q * 128;
      const int s = (q >> 1) & 7;
      a0[i] = base + ((fg ^ s) << 4); a1[i] = base + (((4 + fg) ^ s) << 4);
    }
    const unsigned b0 = b_lane + bbuf * (BBUF * 2), b1 = b0 ^ 64u;
    bf16x8 xf0[FM], xf1[FM], wf[3];
    sfor<FM>([&](auto I) { constexpr int i = decltype(I)::value; ld16<0>(xf0[i], a0[i]); });
    sfor<FM>([&](auto I) { constexpr int i = decltype(I)::value; ld16<0>(xf1[i], a1[i]); });
    ld16<0>(wf[0], b0);
    ld16<2048>(wf[1], b0);
    sfor<T2>([&](auto T) {
      constexpr int t = decltype(T)::value;
      constexpr int kk = t / FN, j = t % FN;
      if constexpr (t + 2 < T2) {
        constexpr int k2 = (t + 2) / FN, j2 = (t + 2) % FN;
        ld16<j2 * 2048>(wf[(t + 2) % 3], k2 ? b1 : b0);
      }
      constexpr int NW = (t + 1 < T2) + (t + 2 < T2);
      if constexpr (t == 0) { sfor<FM>([&](auto I) { constexpr int i = decltype(I)::value; ldwait<NW>(xf0[i]); ldwait<NW>(xf1[i]); }); }
      ldwait<NW>(wf[t % 3]);
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i][j] = RG_MFMA16(wf[t % 3], kk ? xf1[i] : xf0[i], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("" ::: "memory");
    if (++kw == a.TW) { kw = 0; ++grp; aadvance(); }
  }

  // ---- epilogue (common.h rg_epilogue_quad: the piecewise-linear activations as straight-line code)
  const bool lin = a.act <= RG_ACT_LEAKY;
  const float nslope = rg_act_nslope(a.act, a.slope);
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int r = wave * 32 + i * 16 + fr;
    const int l = r / a.slen, w = r - l * a.slen;
    const int seg = seg0 + l;
    if (l >= a.L || seg >= a.nseg) continue;
    unsigned t = (unsigned)seg;
    const int sc = (int)(t % (unsigned)a.segs_per_line); t /= (unsigned)a.segs_per_line;
    const int rh = (int)(t % (unsigned)a.RH); t /= (unsigned)a.RH;
    const int rd = (int)(t % (unsigned)a.RD); const long n = t / (unsigned)a.RD;
    const int rw = sc * a.slen + w;
    if (rw >= a.RW) continue;
    bf16_t* yrow = a.y + n * a.ysn + (long)(rd * a.os[0] + a.oo[0]) * a.ysd + (long)(rh * a.os[1] + a.oo[1]) * a.ysh +
                   (long)(rw * a.os[2] + a.oo[2]) * a.ysw;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int c = n0 + j * 16 + fg * 4;
      if (c >= a.cout_lanes) continue;
      float bv[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) bv[rr] = (a.bias && c + rr < a.cout) ? a.bias[c + rr] : 0.f;
      *(uint2*)(yrow + c) = lin ? rg_epilogue_quad<true>(acc[i][j], bv, a.cout - c, nslope, a.act, a.slope)
                                : rg_epilogue_quad<false>(acc[i][j], bv, a.cout - c, nslope, a.act, a.slope);
    }
  }
}

// can this problem run on igemm3?  (W row stride 1, W tap step +-1, at least 2 W taps, staged rows fit)
bool rg_igemm3_ok(const CL5& X, const rg_conv_params* p) {
  if (p->rstride[2] != 1 || (p->tstep[2] != 1 && p->tstep[2] != -1) || p->T[2] < 3 || p->T[2] > 8) return false;
  const int rw = p->R[2];
  const int slen = rw < 128 ? rw : 128;
  const int L = 128 / slen;
  if (L * (slen + p->T[2] - 1) > AROWS) return false;
  // MFMA rows actually used: whole lines per 128-row tile (short lines) or the ragged last segment (long lines, e.g.
  // 134 = 128 + 6).  Below 88 % the flat-row igemm2 measured faster.
  const long used = rw < 128 ? (long)L * slen : rw, alloc = rw < 128 ? 128 : (long)((rw + 127) / 128) * 128;
  if (used * 100 < alloc * 88) return false;
  if ((long)X.N * X.sn >= 0x7fffffffL) return false;
  // zero-padded problems (taps leaving the volume along D/H: backward-data, padded convs) measured slower here than on
  // igemm2 (592 vs 746 TFLOP/s): keep them there
  const int I[2] = {X.D, X.H};
  for (int d = 0; d < 2; ++d) {
    const long lo0 = p->tbase[d], lo1 = p->tbase[d] + (long)p->tstep[d] * (p->T[d] - 1);
    const long mn = lo0 < lo1 ? lo0 : lo1, mx = (lo0 < lo1 ? lo1 : lo0) + (long)(p->R[d] - 1) * p->rstride[d];
    if (mn < 0 || mx >= I[d]) return false;
  }
  return true;
}

template <int FN>
static void launch3(Igemm3Args& a, bool check, hipStream_t st) {
  a.tiles_n = (a.cout_lanes + FN * 16 - 1) / (FN * 16);
  rg_set_kernel_name("igemm3_kernel<%d,%d>", FN, (int)check);
  if (check) hipLaunchKernelGGL((igemm3_kernel<FN, true>), dim3(a.tiles_m * a.tiles_n), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((igemm3_kernel<FN, false>), dim3(a.tiles_m * a.tiles_n), dim3(256), 0, st, a);
}

int rg_launch_igemm3(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p,
                     const void* zeros, hipStream_t st) {
  Igemm3Args a;
  a.x = X.p; a.w = (const bf16_t*)w_packed; a.bias = bias; a.y = Y.p; a.zeros = (const bf16_t*)zeros;
  a.N = X.N; a.ID = X.D; a.IH = X.H; a.IW = X.W;
  a.RD = p->R[0]; a.RH = p->R[1]; a.RW = p->R[2];
  a.TD = p->T[0]; a.TH = p->T[1]; a.TW = p->T[2];
  for (int d = 0; d < 3; ++d) { a.rs[d] = p->rstride[d]; a.tb[d] = p->tbase[d]; a.ts[d] = p->tstep[d]; a.os[d] = p->ostride[d]; a.oo[d] = p->ooffset[d]; }
  a.xsn = X.sn; a.xsd = X.sd; a.xsh = X.sh; a.xsw = X.sw;
  a.ysn = Y.sn; a.ysd = Y.sd; a.ysh = Y.sh; a.ysw = Y.sw;
  a.cink = round_up(X.C, 64); a.cin_lanes = X.lanes;
  a.cout = p->cout; a.cout_lanes = Y.lanes; a.wrows = round_up(p->cout, 16);
  a.act = p->act; a.slope = p->slope;
  a.slen = a.RW < 128 ? a.RW : 128;
  a.L = 128 / a.slen;
  a.xw = a.slen + a.TW - 1;
  a.segs_per_line = (a.RW + a.slen - 1) / a.slen;
  a.nseg = X.N * a.RD * a.RH * a.segs_per_line;
  a.wbase = a.tb[2] + (a.ts[2] < 0 ? -(a.TW - 1) : 0);
  a.tiles_m = (a.nseg + a.L - 1) / a.L;
  bool check = false;  // D/H taps may leave the volume (W is always bounds-checked through avalid)
  const int I[2] = {X.D, X.H};
  for (int d = 0; d < 2; ++d) {
    const long lo0 = p->tbase[d], lo1 = p->tbase[d] + (long)p->tstep[d] * (p->T[d] - 1);
    const long mn = lo0 < lo1 ? lo0 : lo1, mx = (lo0 < lo1 ? lo1 : lo0) + (long)(p->R[d] - 1) * p->rstride[d];
    if (mn < 0 || mx >= I[d]) check = true;
  }
  const int nf = a.wrows / 16;
  int fn;
  if (nf % 9 == 0) fn = 9; else if (nf % 8 == 0) fn = 8; else if (nf <= 5) fn = nf < 4 ? 4 : nf; else if (nf % 5 == 0) fn = 5;
  else fn = (round_up(nf, 9) - nf) * 8 <= (round_up(nf, 8) - nf) * 9 ? 9 : 8;
  switch (fn) {
    case 4: launch3<4>(a, check, st); break;
    case 5: launch3<5>(a, check, st); break;
    case 8: launch3<8>(a, check, st); break;
    default: launch3<9>(a, check, st); break;
  }
  return 0;
}
