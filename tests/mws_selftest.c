/* Sanitizer driver for the CPU oracle oracle/mws.c (tests/test_oracle_sanitize.py builds both with -fsanitize=address,undefined):
 * random volumes through mws_oracle2 (mask, stride grid with an origin, offsets that leave the volume) and random edge lists through
 * mws_graph_oracle; checks the label convention (0 exactly on masked-out voxels, labels 1..n without gaps in raster order). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
int mws_oracle2(const float* affs, const int32_t* offsets, int32_t n_off, int32_t n_attr, const int32_t shape[3], const int32_t strides[3],
                const int32_t origin[3], const uint8_t* mask, uint64_t* labels);
int mws_graph_oracle(int64_t n_nodes, int64_t n_edges, const uint32_t* wkey, const uint64_t* gid, const int64_t* eu, const int64_t* ev,
                     const uint8_t* attr, int64_t* rep);
static uint64_t s = 88172645463325252ull;
static uint32_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }
int main(void) {
  const int32_t offsets[7 * 3] = {-1, 0, 0, 0, -1, 0, 0, 0, -1, -2, 0, 0, 0, -3, 1, 0, 2, -3, -9, 0, 0};
  for (int rep = 0; rep < 6; ++rep) {
    const int32_t shape[3] = {3 + (int)(rnd() % 6), 2 + (int)(rnd() % 9), 1 + (int)(rnd() % 11)};
    const int32_t strides[3] = {1 + (int)(rnd() % 3), 1 + (int)(rnd() % 3), 1 + (int)(rnd() % 3)};
    const int32_t origin[3] = {(int)(rnd() % 5), (int)(rnd() % 5), (int)(rnd() % 5)};
    const int64_t N = (int64_t)shape[0] * shape[1] * shape[2];
    float* affs = (float*)malloc((size_t)N * 7 * sizeof(float));
    uint8_t* mask = (uint8_t*)malloc((size_t)N);
    uint64_t* lab = (uint64_t*)malloc((size_t)N * sizeof(uint64_t));
    for (int64_t i = 0; i < N * 7; ++i) affs[i] = (float)(rnd() % 64) / 63.f;   /* many ties */
    for (int64_t i = 0; i < N; ++i) mask[i] = (rnd() % 5) != 0;
    if (mws_oracle2(affs, offsets, 7, 3, shape, strides, origin, rep & 1 ? mask : NULL, lab)) return 2;
    uint64_t next = 1;
    for (int64_t i = 0; i < N; ++i) {
      if ((rep & 1) && !mask[i]) { if (lab[i] != 0) return 3; continue; }
      if (lab[i] == 0 || lab[i] > next) return 4;
      if (lab[i] == next) ++next;
    }
    free(affs); free(mask); free(lab);
  }
  for (int rep = 0; rep < 4; ++rep) {
    const int64_t n = 1 + rnd() % 40, m = rnd() % 200;
    uint32_t* wkey = (uint32_t*)malloc((size_t)(m + 1) * 4); uint64_t* gid = (uint64_t*)malloc((size_t)(m + 1) * 8);
    int64_t* eu = (int64_t*)malloc((size_t)(m + 1) * 8); int64_t* ev = (int64_t*)malloc((size_t)(m + 1) * 8);
    uint8_t* at = (uint8_t*)malloc((size_t)m + 1); int64_t* out = (int64_t*)malloc((size_t)n * 8);
    for (int64_t i = 0; i < m; ++i) { wkey[i] = rnd() % 16; gid[i] = rnd(); eu[i] = rnd() % n; ev[i] = rnd() % n; at[i] = rnd() & 1; }
    if (mws_graph_oracle(n, m, wkey, gid, eu, ev, at, out)) return 5;
    for (int64_t i = 0; i < n; ++i) if (out[i] < 0 || out[i] > i || out[out[i]] != out[i]) return 6;
    free(wkey); free(gid); free(eu); free(ev); free(at); free(out);
  }
  printf("mws oracle selftest ok\n");
  return 0;
}
