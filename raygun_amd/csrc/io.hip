// Host-side codecs for the chunk reader (raygun_amd/zarrio.py): LZ4 block decompression, the stream format inside zarr's default
// Blosc compressor (cname "lz4" / "lz4hc").  Restated from the published LZ4 block format description (lz4_Block_format.md):
// a block is a series of sequences [token][literal length+][literals][offset LE16][match length+]; the last sequence ends after
// its literals.  No device code here; it lives in the library so that the Python reader has a native path for multi-megabyte chunks.
#ifdef RG_IO_STANDALONE   // tests/test_oracle_sanitize.py builds this file alone with g++ -fsanitize=address,undefined (no HIP headers needed: host code only)
#include <stdint.h>
#include <string.h>
#else
#include "common.h"
#endif

// returns the number of bytes written to dst, or -1 on malformed input / overflow of either buffer
extern "C" int64_t rg_lz4_decompress_block(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap) {
  if (!src || !dst || src_len < 0 || dst_cap < 0) return -1;
  int64_t ip = 0, op = 0;
  while (ip < src_len) {
    const unsigned token = src[ip++];
    int64_t lit = token >> 4;
    if (lit == 15) { unsigned b; do { if (ip >= src_len) return -1; b = src[ip++]; lit += b; } while (b == 255); }
    if (lit > src_len - ip || lit > dst_cap - op) return -1;
    memcpy(dst + op, src + ip, (size_t)lit); ip += lit; op += lit;
    if (ip >= src_len) break;                      // last sequence: literals only
    if (ip + 2 > src_len) return -1;
    const int64_t off = (int64_t)src[ip] | ((int64_t)src[ip + 1] << 8); ip += 2;
    if (off == 0 || off > op) return -1;
    int64_t ml = token & 15;
    if (ml == 15) { unsigned b; do { if (ip >= src_len) return -1; b = src[ip++]; ml += b; } while (b == 255); }
    ml += 4;
    if (ml > dst_cap - op) return -1;
    const uint8_t* m = dst + op - off;
    if (off >= ml) memcpy(dst + op, m, (size_t)ml);
    else for (int64_t i = 0; i < ml; ++i) dst[op + i] = m[i];   // overlapping match: byte by byte (run-length semantics)
    op += ml;
  }
  return op;
}

// LZ4 block COMPRESSION (greedy, one 8 Ki-entry hash table of 4-byte sequences, 64 KiB window) — what the chunk writer needs to lay down Blosc / lz4
// frames (zarr's and daisy.prepare_ds's default compressor) at memory speed instead of python-zlib speed (predict's zarr destinations: 34 MB per block).
// Follows the block format's end-of-block rules: the last match starts at least 12 bytes before the end, the last 5 bytes are literals.
// Returns the compressed size, or -1 when dst_cap is too small (callers then store the block raw).  Any valid LZ4 decoder reads the output.
static inline uint32_t lz4_read32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
extern "C" int64_t rg_lz4_compress_bound(int64_t n) { return n < 0 ? -1 : n + n / 255 + 16; }
extern "C" int64_t rg_lz4_compress_block(const uint8_t* src, int64_t n, uint8_t* dst, int64_t dst_cap) {
  if (!src || !dst || n < 0 || dst_cap < 0 || n > 0x7E000000) return -1;
  enum { HLOG = 13 };                                        // 8 Ki entries = 32 KiB: stays in L1/L2 and costs little to clear per 256-KiB Blosc block
  static thread_local uint32_t table[1 << HLOG];             // position + 1 (0 = empty)
  memset(table, 0, sizeof(table));
  int64_t ip = 0, anchor = 0, op = 0;
  auto emit = [&](int64_t lit, int64_t ml, int64_t off) -> bool {   // ml < 0: the closing literal run
    const int64_t need = 1 + lit / 255 + 1 + lit + (ml >= 0 ? 2 + (ml - 4) / 255 + 1 : 0);
    if (op + need > dst_cap) return false;
    uint8_t* tok = dst + op++;
    if (lit >= 15) { *tok = 0xF0; int64_t r = lit - 15; while (r >= 255) { dst[op++] = 255; r -= 255; } dst[op++] = (uint8_t)r; }
    else *tok = (uint8_t)(lit << 4);
    memcpy(dst + op, src + anchor, (size_t)lit); op += lit;
    if (ml < 0) return true;
    dst[op++] = (uint8_t)(off & 255); dst[op++] = (uint8_t)(off >> 8);
    int64_t m = ml - 4;
    if (m >= 15) { *tok |= 15; m -= 15; while (m >= 255) { dst[op++] = 255; m -= 255; } dst[op++] = (uint8_t)m; }
    else *tok |= (uint8_t)m;
    return true;
  };
  if (n >= 13) {
    const int64_t mflimit = n - 12, matchlimit = n - 5;
    unsigned misses = 0;
    while (ip < mflimit) {
      const uint32_t seq = lz4_read32(src + ip);
      const uint32_t h = (seq * 2654435761u) >> (32 - HLOG);
      const int64_t ref = (int64_t)table[h] - 1;
      table[h] = (uint32_t)(ip + 1);
      if (ref >= 0 && ip - ref <= 65535 && lz4_read32(src + ref) == seq) {
        int64_t ml = 4;
        while (ip + ml < matchlimit && src[ref + ml] == src[ip + ml]) ++ml;
        if (!emit(ip - anchor, ml, ip - ref)) return -1;
        ip += ml; anchor = ip; misses = 0;
        if (ip - 2 > ref && ip - 2 < mflimit) table[(lz4_read32(src + ip - 2) * 2654435761u) >> (32 - HLOG)] = (uint32_t)(ip - 2 + 1);
      } else {
        ip += 1 + (misses++ >> 6);                            // incompressible stretches are skipped faster and faster
      }
    }
  }
  if (!emit(n - anchor, -1, 0)) return -1;
  return op;
}

// ---- whole Blosc (c-blosc 1.x) frames in one call each, so that the chunk reader / writer holds Python's GIL for microseconds per 34-MB chunk instead
// of for a loop over its 256-KiB blocks.  Frame layout as published (blosc.h / README_HEADER.rst): 16-byte header { version 2, versionlz 1, flags,
// typesize, nbytes, blocksize, cbytes (LE32) }, flags: 0x1 byte shuffle, 0x2 memcpy'd (stored), 0x4 bit shuffle, 0x10 blocks NOT split into
// `typesize` streams, bits 5-7 the inner codec (1 = lz4 / lz4hc, 3 = zlib, ...); then one LE32 start offset per block; a block = [LE32 stream size][stream]
// per stream; a stream whose size equals its uncompressed size is stored raw.
static inline uint32_t io_le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline void io_put32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
extern "C" int64_t rg_blosc_compress_bound(int64_t n) { return n < 0 ? -1 : n + 16 + 8 * (n / (1 << 18) + 2) + 64; }
// lz4 inner codec, byte shuffle when typesize > 1 and shuffle != 0, unsplit blocks of `blocksize` (0: 256 KiB; clamped to the buffer like c-blosc).
// Returns the frame length, or -1 (bad arguments / dst_cap below rg_blosc_compress_bound(n)).  `scratch`: 2 * blocksize + 64 bytes or NULL (allocated).
extern "C" int64_t rg_blosc_compress(const uint8_t* src, int64_t n, int32_t typesize, int32_t shuffle, int32_t blocksize, uint8_t* dst, int64_t dst_cap) {
  if ((!src && n) || !dst || n < 0 || n >= 0x7E000000 || typesize < 1 || typesize > 255 || dst_cap < rg_blosc_compress_bound(n)) return -1;
  int64_t bs = blocksize > 0 ? blocksize : (1 << 18);
  bs = bs / typesize * typesize; if (bs < typesize) bs = typesize;
  if (n > 0) { int64_t lim = n / typesize * typesize; if (lim < typesize) lim = typesize; if (bs > lim) bs = lim; }
  const bool shuf = typesize > 1 && shuffle != 0;
  const int64_t nblocks = n ? (n + bs - 1) / bs : 0;
  const uint8_t flags = (uint8_t)((shuf ? 0x1 : 0) | 0x10 | (1 << 5));
  uint8_t* tmp = shuf ? new uint8_t[(size_t)bs] : nullptr;
  int64_t pos = 16 + 4 * nblocks;
  bool stored = false;
  for (int64_t b = 0; b < nblocks && !stored; ++b) {
    const int64_t len = (b + 1) * bs <= n ? bs : n - b * bs;
    const uint8_t* blk = src + b * bs;
    if (shuf) {                                               // byte k of every element together; a ragged tail (len % typesize) stays in place
      const int64_t nel = len / typesize;
      for (int k = 0; k < typesize; ++k) for (int64_t e = 0; e < nel; ++e) tmp[k * nel + e] = blk[e * typesize + k];
      memcpy(tmp + nel * typesize, blk + nel * typesize, (size_t)(len - nel * typesize));
      blk = tmp;
    }
    if (pos + 4 + len > dst_cap) { stored = true; break; }
    io_put32(dst + 16 + 4 * b, (uint32_t)pos);
    int64_t c = rg_lz4_compress_block(blk, len, dst + pos + 4, len - 1 > 0 ? len - 1 : 0);   // must shrink, else the stream is stored raw
    if (c < 0) { memcpy(dst + pos + 4, blk, (size_t)len); c = len; }
    io_put32(dst + pos, (uint32_t)c);
    pos += 4 + c;
    if (pos >= n + 16) stored = true;                         // nothing gained: the memcpy form below
  }
  delete[] tmp;
  dst[0] = 2; dst[1] = 1; dst[3] = (uint8_t)typesize;
  io_put32(dst + 4, (uint32_t)n); io_put32(dst + 8, (uint32_t)bs);
  if (stored || pos >= n + 16) {
    dst[2] = flags | 0x2; io_put32(dst + 12, (uint32_t)(n + 16));
    if (n) memcpy(dst + 16, src, (size_t)n);
    return n + 16;
  }
  dst[2] = flags; io_put32(dst + 12, (uint32_t)pos);
  return pos;
}
// nbytes of a frame (what the caller must provide to rg_blosc_decompress), or -1
extern "C" int64_t rg_blosc_nbytes(const uint8_t* src, int64_t src_len) { return (!src || src_len < 16) ? -1 : (int64_t)io_le32(src + 4); }
// lz4 / lz4hc frames and stored frames, byte shuffle, split or unsplit blocks.  Returns nbytes, -2 for a frame this decoder does not handle (bit shuffle,
// other inner codecs: the Python reader takes over), -1 for a malformed frame.
extern "C" int64_t rg_blosc_decompress(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap) {
  if (!src || !dst || src_len < 16) return -1;
  const unsigned flags = src[2], ts = src[3] ? src[3] : 1;
  const int64_t n = io_le32(src + 4), bs = io_le32(src + 8), cbytes = io_le32(src + 12);
  if (n > dst_cap || cbytes > src_len) return -1;
  if (flags & 0x2) { if (src_len < 16 + n) return -1; memcpy(dst, src + 16, (size_t)n); return n; }
  if ((flags & 0x4) || ((flags >> 5) != 1)) return -2;
  if (n == 0) return 0;
  if (bs <= 0) return -1;
  const int64_t nblocks = (n + bs - 1) / bs;
  if (16 + 4 * nblocks > src_len) return -1;
  const bool shuf = (flags & 0x1) && ts > 1;
  uint8_t* tmp = shuf ? new uint8_t[(size_t)(bs < n ? bs : n)] : nullptr;   // (a block is never longer than the buffer, whatever the header claims)
  int64_t rc = n;
  for (int64_t b = 0; b < nblocks && rc >= 0; ++b) {
    const int64_t len = (b + 1) * bs <= n ? bs : n - b * bs;
    int64_t p = io_le32(src + 16 + 4 * b);
    const bool split = !(flags & 0x10) && ts > 1 && ts <= 16 && bs / (int64_t)ts >= 128 && len == bs;   // c-blosc 1.x: never the short last block
    const int nstreams = split ? (int)ts : 1;
    const int64_t slen = len / nstreams;
    uint8_t* out = shuf ? tmp : dst + b * bs;
    for (int s = 0; s < nstreams && rc >= 0; ++s) {
      if (p + 4 > src_len) { rc = -1; break; }
      const int64_t cs = io_le32(src + p); p += 4;
      if (cs < 0 || p + cs > src_len) { rc = -1; break; }
      if (cs == slen) memcpy(out + s * slen, src + p, (size_t)slen);
      else if (rg_lz4_decompress_block(src + p, cs, out + s * slen, slen) != slen) rc = -1;
      p += cs;
    }
    if (shuf && rc >= 0) {
      uint8_t* o = dst + b * bs;
      const int64_t nel = len / ts;
      for (unsigned k = 0; k < ts; ++k) for (int64_t e = 0; e < nel; ++e) o[e * ts + k] = tmp[k * nel + e];
      memcpy(o + nel * ts, tmp + nel * ts, (size_t)(len - nel * ts));
    }
  }
  delete[] tmp;
  return rc;
}
