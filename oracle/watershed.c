/* ORACLE (test infrastructure — never linked or called by the product path): C helpers of oracle/watershed.py, the CPU restatement of
 * the reference's DEFAULT `raygun-segment` route (src/raygun/segment.py:15-136: seeded watershed on the boundary-distance transform, then
 * waterz agglomeration).
 *
 * PARITY UNPINNED for everything in this file: scikit-image and waterz are neither installed in the image nor vendored under
 * /root/reference, and no reference test touches segment.py.  What is restated:
 *
 *   ws_skimage   skimage.segmentation.watershed(image, markers, mask=mask) with connectivity 1, compactness 0, no watershed line
 *                (scikit-image >= 0.19, _watershed.py / _watershed_cython.pyx / heap_general.pxi as published): markers * mask, every marker
 *                voxel pushed in raster order with age 0, then pop the smallest (value, age); every unlabelled in-mask neighbour — order -z, -y, -x,
 *                +x, +y, +z — takes the popped voxel's label AT PUSH TIME and is pushed with ITS OWN image value and the next age.  The heap is
 *                CPython's heapq (which heap_general.pxi transcribes), so equal (value, age) pairs — only the age-0 markers — leave in heapq's order.
 *   ws_msf       the build-defined seeded watershed the GPU path implements (raygun_amd/csrc/watershed.hip): the minimum spanning forest of the
 *                6-connected in-mask grid graph rooted at the seeds, under the strict total edge order
 *                    key(u, v) = (min(d2[u], d2[v]) DESCENDING, max(d2[u], d2[v]) DESCENDING, edge id ASCENDING),   edge id = 3 * min(u, v) + axis
 *                (d2 = squared boundary distance: the higher end of an edge in the flooded image Dmax - D is the one with the SMALLER d2; among the edges
 *                that share their higher end the steepest descent comes first) — Kruskal with markers: edges in that order, two clusters are joined
 *                unless BOTH already hold a seed.  Same minimax-path assignment as the priority flood; it differs from skimage's where skimage's result
 *                hangs on insertion order (plateaus of the integer-valued squared distance, passes reached at equal level): a few per cent of the voxels
 *                on the synthetic test volumes, along the surfaces where two floods meet.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------------------------- skimage restatement */
typedef struct { double value; int64_t age; int64_t index; } Item;
static int smaller(const Item* a, const Item* b) { return a->value != b->value ? a->value < b->value : a->age < b->age; }
typedef struct { Item* d; int64_t n, cap; } Heap;
static int hpush(Heap* h, Item it) {                       /* heapq.heappush: append, _siftdown(heap, 0, len - 1) */
  if (h->n == h->cap) { h->cap = h->cap ? 2 * h->cap : 1024; Item* p = (Item*)realloc(h->d, (size_t)h->cap * sizeof(Item)); if (!p) return -1; h->d = p; }
  int64_t pos = h->n++;
  while (pos > 0) { int64_t par = (pos - 1) >> 1; if (smaller(&it, &h->d[par])) { h->d[pos] = h->d[par]; pos = par; } else break; }
  h->d[pos] = it;
  return 0;
}
static Item hpop(Heap* h) {                                /* heapq.heappop: last element to the root, _siftup (to a leaf, then _siftdown) */
  Item last = h->d[--h->n];
  if (h->n == 0) return last;
  Item ret = h->d[0];
  int64_t pos = 0, end = h->n, child = 1;
  while (child < end) {
    int64_t right = child + 1;
    if (right < end && !smaller(&h->d[child], &h->d[right])) child = right;
    h->d[pos] = h->d[child]; pos = child; child = 2 * pos + 1;
  }
  while (pos > 0) { int64_t par = (pos - 1) >> 1; if (smaller(&last, &h->d[par])) { h->d[pos] = h->d[par]; pos = par; } else break; }
  h->d[pos] = last;
  return ret;
}

int ws_skimage(const double* image, const int32_t* markers, const uint8_t* mask, const int32_t* shape, int32_t* out) {
  const int64_t Z = shape[0], Y = shape[1], X = shape[2], N = Z * Y * X;
  Heap h = {0, 0, 0};
  for (int64_t i = 0; i < N; ++i) out[i] = (!mask || mask[i]) ? markers[i] : 0;
  for (int64_t i = 0; i < N; ++i)
    if (out[i]) { Item it = {image[i], 0, i}; if (hpush(&h, it)) { free(h.d); return -1; } }
  int64_t age = 1;
  const int64_t step[6] = {-Y * X, -X, -1, 1, X, Y * X};
  while (h.n) {
    const Item e = hpop(&h);
    const int64_t z = e.index / (Y * X), y = (e.index / X) % Y, x = e.index % X;
    const int ok[6] = {z > 0, y > 0, x > 0, x < X - 1, y < Y - 1, z < Z - 1};
    for (int k = 0; k < 6; ++k) {
      if (!ok[k]) continue;
      const int64_t nb = e.index + step[k];
      if (mask && !mask[nb]) continue;
      if (out[nb]) continue;
      age += 1;
      out[nb] = out[e.index];
      Item it = {image[nb], age, nb};
      if (hpush(&h, it)) { free(h.d); return -1; }
    }
  }
  free(h.d);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ build-defined order: Kruskal with markers */
typedef struct { uint32_t k1, k2; uint64_t id; } EdgeKey;
static int cmp_key(const void* a, const void* b) {
  const EdgeKey* x = (const EdgeKey*)a; const EdgeKey* y = (const EdgeKey*)b;
  if (x->k1 != y->k1) return x->k1 < y->k1 ? -1 : 1;
  if (x->k2 != y->k2) return x->k2 < y->k2 ? -1 : 1;
  return x->id < y->id ? -1 : x->id > y->id;
}
static int64_t uf_find(int64_t* p, int64_t v) { while (p[v] != v) { p[v] = p[p[v]]; v = p[v]; } return v; }

#define WS_D2_CAP ((1 << 30) - 1)
int ws_msf(const int32_t* d2, const int32_t* markers, const uint8_t* mask, const int32_t* shape, int32_t* out) {
  const int64_t Z = shape[0], Y = shape[1], X = shape[2], N = Z * Y * X;
  int64_t* parent = (int64_t*)malloc((size_t)N * sizeof(int64_t));
  int32_t* seed = (int32_t*)calloc((size_t)N, sizeof(int32_t));        /* label held by the cluster whose root this is (0: none) */
  EdgeKey* keys = (EdgeKey*)malloc((size_t)(3 * N) * sizeof(EdgeKey));
  if (!parent || !seed || !keys) { free(parent); free(seed); free(keys); return -1; }
  for (int64_t i = 0; i < N; ++i) { parent[i] = i; seed[i] = (!mask || mask[i]) ? markers[i] : 0; }
  const int64_t step[3] = {Y * X, X, 1};
  int64_t ne = 0;
  for (int64_t i = 0; i < N; ++i) {
    if (mask && !mask[i]) continue;
    const int64_t z = i / (Y * X), y = (i / X) % Y, x = i % X;
    const int ok[3] = {z < Z - 1, y < Y - 1, x < X - 1};
    for (int a = 0; a < 3; ++a) {
      if (!ok[a]) continue;
      const int64_t j = i + step[a];
      if (mask && !mask[j]) continue;
      const int32_t lo = d2[i] < d2[j] ? d2[i] : d2[j], hi = d2[i] < d2[j] ? d2[j] : d2[i];
      if (lo < 0 || hi > WS_D2_CAP) { free(parent); free(seed); free(keys); return -3; }
      keys[ne].k1 = (uint32_t)(WS_D2_CAP - lo);           /* the edge's HIGHER end in the flooded image (smaller distance): later = larger k1 */
      keys[ne].k2 = (uint32_t)(WS_D2_CAP - hi);           /* then its lower end: steepest descent first */
      keys[ne].id = (uint64_t)(3 * i + a);
      ++ne;
    }
  }
  qsort(keys, (size_t)ne, sizeof(EdgeKey), cmp_key);
  for (int64_t e = 0; e < ne; ++e) {
    const uint64_t id = keys[e].id;
    const int64_t i = (int64_t)(id / 3), j = i + step[id % 3];
    int64_t ri = uf_find(parent, i), rj = uf_find(parent, j);
    if (ri == rj) continue;
    if (seed[ri] && seed[rj]) continue;                     /* two seeded clusters never join */
    if (seed[ri]) { parent[rj] = ri; } else { parent[ri] = rj; }
  }
  for (int64_t i = 0; i < N; ++i) out[i] = (!mask || mask[i]) ? seed[uf_find(parent, i)] : 0;
  free(parent); free(seed); free(keys);
  return 0;
}
