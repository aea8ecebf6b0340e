/* Sanitizer fuzz driver for the library's host-side LZ4 block decoder (raygun_amd/csrc/io.hip, built alone with -DRG_IO_STANDALONE): valid streams
 * from a tiny greedy encoder must round-trip; truncated, bit-flipped and random streams, and too-small destination buffers, must return -1 or a
 * length within the buffer — never touch memory outside src / dst (ASan + UBSan watch). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef __cplusplus
extern "C"
#endif
int64_t rg_lz4_decompress_block(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap);
#ifdef __cplusplus
extern "C" {
#endif
int64_t rg_lz4_compress_bound(int64_t n);
int64_t rg_lz4_compress_block(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap);
int64_t rg_blosc_compress_bound(int64_t n);
int64_t rg_blosc_compress(const uint8_t* src, int64_t n, int32_t typesize, int32_t shuffle, int32_t blocksize, uint8_t* dst, int64_t dst_cap);
int64_t rg_blosc_nbytes(const uint8_t* src, int64_t src_len);
int64_t rg_blosc_decompress(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap);
#ifdef __cplusplus
}
#endif
static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 13); }
static size_t put_len(uint8_t* o, size_t n) { size_t k = 0; while (n >= 255) { o[k++] = 255; n -= 255; } o[k++] = (uint8_t)n; return k; }
/* greedy encoder: matches of length >= 4 found by brute force in a 64-byte window (enough to produce every token form) */
static size_t encode(const uint8_t* in, size_t n, uint8_t* out) {
  size_t ip = 0, op = 0, lit0 = 0;
  while (ip < n) {
    size_t best = 0, boff = 0;
    if (ip + 12 < n)
      for (size_t off = 1; off <= 64 && off <= ip; ++off) {
        size_t l = 0; while (ip + l + 5 < n && in[ip + l] == in[ip + l - off] && l < 300) ++l;
        if (l > best) { best = l; boff = off; }
      }
    if (best >= 4) {
      const size_t lit = ip - lit0, ml = best - 4;
      uint8_t* tok = out + op++; *tok = (uint8_t)(((lit < 15 ? lit : 15) << 4) | (ml < 15 ? ml : 15));
      if (lit >= 15) op += put_len(out + op, lit - 15);
      memcpy(out + op, in + lit0, lit); op += lit;
      out[op++] = (uint8_t)boff; out[op++] = 0;
      if (ml >= 15) op += put_len(out + op, ml - 15);
      ip += best; lit0 = ip;
    } else ++ip;
  }
  { const size_t lit = n - lit0; uint8_t* tok = out + op++; *tok = (uint8_t)((lit < 15 ? lit : 15) << 4);
    if (lit >= 15) op += put_len(out + op, lit - 15);
    memcpy(out + op, in + lit0, lit); op += lit; }
  return op;
}
int main(void) {
  for (int rep = 0; rep < 300; ++rep) {
    const size_t n = 1 + rnd() % 3000;
    uint8_t* in = (uint8_t*)malloc(n); uint8_t* enc = (uint8_t*)malloc(2 * n + 64); uint8_t* out = (uint8_t*)malloc(n + 8);
    for (size_t i = 0; i < n; ++i) in[i] = (rep & 1) ? (uint8_t)(rnd() % 3) : (uint8_t)((i / 7) % 5 + (rnd() % 16 == 0));
    const size_t m = encode(in, n, enc);
    if (rg_lz4_decompress_block(enc, (int64_t)m, out, (int64_t)n) != (int64_t)n || memcmp(in, out, n)) { printf("round trip failed (rep %d)\n", rep); return 2; }
    /* exact-size source copies so that any over-read trips ASan */
    for (int k = 0; k < 40; ++k) {
      const size_t cut = rnd() % (m + 1);
      uint8_t* t = (uint8_t*)malloc(cut ? cut : 1); memcpy(t, enc, cut);
      if (cut && (k & 1)) t[rnd() % cut] ^= (uint8_t)(1u << (rnd() % 8));
      const size_t cap = (k % 3 == 0) ? rnd() % (n + 1) : n;
      uint8_t* d = (uint8_t*)malloc(cap ? cap : 1);
      const int64_t r = rg_lz4_decompress_block(t, (int64_t)cut, d, (int64_t)cap);
      if (r < -1 || r > (int64_t)cap) { printf("bad return %lld\n", (long long)r); return 3; }
      free(t); free(d);
    }
    { const size_t q = 1 + rnd() % 200; uint8_t* t = (uint8_t*)malloc(q); for (size_t i = 0; i < q; ++i) t[i] = (uint8_t)rnd();
      uint8_t* d = (uint8_t*)malloc(64); const int64_t r = rg_lz4_decompress_block(t, (int64_t)q, d, 64);
      if (r < -1 || r > 64) return 4; free(t); free(d); }
    free(in); free(enc); free(out);
  }
  if (rg_lz4_decompress_block(NULL, 3, NULL, 3) != -1) return 5;
  /* the library's own compressor and the whole-frame Blosc codec: round trips over sizes around the end-of-block rules, type sizes with and without
   * shuffle, small block sizes (many blocks, a short last block), then truncated / bit-flipped frames and short destinations */
  for (int rep = 0; rep < 400; ++rep) {
    const size_t n = rep < 40 ? (size_t)rep : 1 + rnd() % 20000;
    uint8_t* in = (uint8_t*)malloc(n ? n : 1);
    for (size_t i = 0; i < n; ++i) in[i] = (rep % 3 == 0) ? (uint8_t)rnd() : (uint8_t)((i / 5) % 7 + (rnd() % 11 == 0));
    const int64_t cap = rg_lz4_compress_bound((int64_t)n);
    uint8_t* enc = (uint8_t*)malloc((size_t)cap); uint8_t* out = (uint8_t*)malloc(n ? n : 1);
    const int64_t m = rg_lz4_compress_block(in, (int64_t)n, enc, cap);
    if (m < 0 || rg_lz4_decompress_block(enc, m, out, (int64_t)n) != (int64_t)n || memcmp(in, out, n)) { printf("compress round trip failed (n %zu)\n", n); return 6; }
    if (n > 64) { uint8_t* tight = (uint8_t*)malloc(n / 4); const int64_t r = rg_lz4_compress_block(in, (int64_t)n, tight, (int64_t)(n / 4)); if (r > (int64_t)(n / 4)) return 7; free(tight); }
    const int ts = 1 + (int)(rnd() % 8), shuf = (int)(rnd() & 1), bs = (rnd() & 1) ? 0 : 16 + (int)(rnd() % 3000);
    const int64_t fcap = rg_blosc_compress_bound((int64_t)n);
    uint8_t* fr = (uint8_t*)malloc((size_t)fcap);
    const int64_t fl = rg_blosc_compress(in, (int64_t)n, ts, shuf, bs, fr, fcap);
    if (fl < 16 || fl > fcap || rg_blosc_nbytes(fr, fl) != (int64_t)n) { printf("blosc compress failed (n %zu)\n", n); return 8; }
    if (rg_blosc_decompress(fr, fl, out, (int64_t)n) != (int64_t)n || memcmp(in, out, n)) { printf("blosc round trip failed (n %zu ts %d shuffle %d bs %d)\n", n, ts, shuf, bs); return 9; }
    for (int k = 0; k < 12; ++k) {
      const size_t cut = 16 + rnd() % (size_t)(fl - 16 + 1);
      uint8_t* t = (uint8_t*)malloc(cut); memcpy(t, fr, cut);
      if (k & 1) t[rnd() % cut] ^= (uint8_t)(1u << (rnd() % 8));
      const size_t dcap = (k % 3 == 0) ? rnd() % (n + 1) : n;
      uint8_t* d = (uint8_t*)malloc(dcap ? dcap : 1);
      const int64_t r = rg_blosc_decompress(t, (int64_t)cut, d, (int64_t)dcap);
      if (r < -2 || r > (int64_t)dcap) { printf("blosc bad return %lld\n", (long long)r); return 10; }
      free(t); free(d);
    }
    free(in); free(enc); free(out); free(fr);
  }
  printf("lz4 fuzz ok\n");
  return 0;
}
