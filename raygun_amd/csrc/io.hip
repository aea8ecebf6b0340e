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
