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

// LZ4 block COMPRESSION (greedy, one 64 Ki-entry hash table of 4-byte sequences, 64 KiB window) — what the chunk writer needs to lay down Blosc / lz4
// frames (zarr's and daisy.prepare_ds's default compressor) at memory speed instead of python-zlib speed (predict's zarr destinations: 34 MB per block).
// Follows the block format's end-of-block rules: the last match starts at least 12 bytes before the end, the last 5 bytes are literals.
// Returns the compressed size, or -1 when dst_cap is too small (callers then store the block raw).  Any valid LZ4 decoder reads the output.
static inline uint32_t lz4_read32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
extern "C" int64_t rg_lz4_compress_bound(int64_t n) { return n < 0 ? -1 : n + n / 255 + 16; }
extern "C" int64_t rg_lz4_compress_block(const uint8_t* src, int64_t n, uint8_t* dst, int64_t dst_cap) {
  if (!src || !dst || n < 0 || dst_cap < 0 || n > 0x7E000000) return -1;
  enum { HLOG = 16 };
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
