/* ORACLE — CPU restatement of the mutex watershed that raygun.segment.mutex_segment obtains from
 * affogato.segmentation.compute_mws_segmentation (reference call site: src/raygun/segment.py:197-199).
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * PARITY UNPINNED: affogato (github.com/constantinpape/affogato, conda-forge, no version pin in the reference:
 * README.md:13) is not vendored and not installed here, and no reference test touches segment.py, so this file
 * restates the PUBLISHED algorithm (Wolf et al., "The Mutex Watershed", ECCV 2018, Alg. 1) with the edge set of
 * affogato's grid graph as recorded in SURVEY.md §8c:
 *   - for every voxel u and offset k an edge (u, u+offset_k) if the target is inside the volume;
 *   - offsets k < n_attractive are attractive, the others repulsive;
 *   - repulsive edges are kept only where u lies on the stride grid (every coordinate % stride == 0);
 *   - edges touching a masked-out voxel (mask == 0) are dropped;
 *   - edges are processed by DESCENDING weight, ties by ASCENDING edge id (k*N + u)  [build-defined total order];
 *   - attractive edge: merge the two clusters unless a mutex constraint separates them;
 *     repulsive edge: add a mutex constraint between the two clusters if they differ;
 *   - labels: 1 + rank of the cluster's first voxel in raster order; masked-out voxels get 0.
 * Weights are ranked as given: affogato's wrapper-level transform (invert_repulsive_weights, bias_cut) is applied by
 * oracle/mws.py:affogato_weights before the call; `affs` is (n_offsets, Z, Y, X) float32.
 * Labels: affogato returns union-find root ids (+1 with a mask), which depend on its internals; the canonical numbering
 * here is build-defined, the PARTITION is the portable claim (oracle/mws.py:same_partition).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float w; uint64_t id; } edge_t;

static int cmp_edge(const void* a, const void* b) {
  const edge_t* x = (const edge_t*)a; const edge_t* y = (const edge_t*)b;
  if (x->w > y->w) return -1;
  if (x->w < y->w) return 1;
  return (x->id > y->id) - (x->id < y->id);
}

typedef struct { int64_t* v; int64_t n, cap; } list_t;

static int64_t find(int64_t* parent, int64_t x) {
  while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
  return x;
}
static void push(list_t* l, int64_t x) {
  if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 4; l->v = (int64_t*)realloc(l->v, (size_t)l->cap * sizeof(int64_t)); }
  l->v[l->n++] = x;
}
/* is there a mutex between clusters ra and rb?  mutex lists hold (possibly stale) members of the partner cluster */
static int has_mutex(int64_t* parent, list_t* mtx, int64_t ra, int64_t rb) {
  if (mtx[ra].n > mtx[rb].n) { int64_t t = ra; ra = rb; rb = t; }
  for (int64_t i = 0; i < mtx[ra].n; ++i) if (find(parent, mtx[ra].v[i]) == rb) return 1;
  return 0;
}

/* origin: global coordinate of voxel (0,0,0) — the stride grid is (coordinate + origin) % stride == 0 (a slab of a larger volume) */
int mws_oracle2(const float* affs, const int32_t* offsets, int32_t n_off, int32_t n_attr, const int32_t shape[3],
                const int32_t strides[3], const int32_t origin[3], const uint8_t* mask, uint64_t* labels) {
  const int64_t Z = shape[0], Y = shape[1], X = shape[2], N = Z * Y * X;
  if (N <= 0) return 0;
  edge_t* edges = (edge_t*)malloc((size_t)N * (size_t)n_off * sizeof(edge_t));
  if (!edges) return -1;
  int64_t ne = 0;
  for (int32_t k = 0; k < n_off; ++k) {
    const int32_t oz = offsets[3 * k], oy = offsets[3 * k + 1], ox = offsets[3 * k + 2];
    for (int64_t z = 0; z < Z; ++z) for (int64_t y = 0; y < Y; ++y) for (int64_t x = 0; x < X; ++x) {
      const int64_t z2 = z + oz, y2 = y + oy, x2 = x + ox;
      if (z2 < 0 || z2 >= Z || y2 < 0 || y2 >= Y || x2 < 0 || x2 >= X) continue;
      if (k >= n_attr && (((z + origin[0]) % strides[0]) || ((y + origin[1]) % strides[1]) || ((x + origin[2]) % strides[2]))) continue;
      const int64_t u = (z * Y + y) * X + x, v = (z2 * Y + y2) * X + x2;
      if (mask && (!mask[u] || !mask[v])) continue;
      edges[ne].w = affs[(int64_t)k * N + u];
      edges[ne].id = (uint64_t)k * (uint64_t)N + (uint64_t)u;
      ++ne;
    }
  }
  qsort(edges, (size_t)ne, sizeof(edge_t), cmp_edge);
  int64_t* parent = (int64_t*)malloc((size_t)N * sizeof(int64_t));
  int64_t* size = (int64_t*)malloc((size_t)N * sizeof(int64_t));
  list_t* mtx = (list_t*)calloc((size_t)N, sizeof(list_t));
  for (int64_t i = 0; i < N; ++i) { parent[i] = i; size[i] = 1; }
  for (int64_t e = 0; e < ne; ++e) {
    const int64_t k = (int64_t)(edges[e].id / (uint64_t)N), u = (int64_t)(edges[e].id % (uint64_t)N);
    const int64_t z = u / (Y * X), y = (u / X) % Y, x = u % X;
    const int64_t v = ((z + offsets[3 * k]) * Y + (y + offsets[3 * k + 1])) * X + (x + offsets[3 * k + 2]);
    int64_t ru = find(parent, u), rv = find(parent, v);
    if (ru == rv) continue;
    if (k < n_attr) {
      if (has_mutex(parent, mtx, ru, rv)) continue;
      if (size[ru] < size[rv]) { int64_t t = ru; ru = rv; rv = t; }
      parent[rv] = ru; size[ru] += size[rv];
      for (int64_t i = 0; i < mtx[rv].n; ++i) push(&mtx[ru], mtx[rv].v[i]);
      free(mtx[rv].v); mtx[rv].v = NULL; mtx[rv].n = mtx[rv].cap = 0;
    } else {
      push(&mtx[ru], rv); push(&mtx[rv], ru);
    }
  }
  /* canonical labels: 1 + rank of first occurrence in raster order, 0 for masked-out voxels */
  int64_t* lab = size;  /* reuse */
  for (int64_t i = 0; i < N; ++i) lab[i] = 0;
  uint64_t next = 1;
  for (int64_t i = 0; i < N; ++i) {
    if (mask && !mask[i]) { labels[i] = 0; continue; }
    const int64_t r = find(parent, i);
    if (!lab[r]) lab[r] = (int64_t)next++;
    labels[i] = (uint64_t)lab[r];
  }
  for (int64_t i = 0; i < N; ++i) free(mtx[i].v);
  free(mtx); free(parent); free(size); free(edges);
  return 0;
}

int mws_oracle(const float* affs, const int32_t* offsets, int32_t n_off, int32_t n_attr, const int32_t shape[3],
               const int32_t strides[3], const uint8_t* mask, uint64_t* labels) {
  const int32_t origin[3] = {0, 0, 0};
  return mws_oracle2(affs, offsets, n_off, n_attr, shape, strides, origin, mask, labels);
}

/* Second level of the slab decomposition (build-defined): Kruskal mutex watershed over an explicit edge list.
 * Order: ascending wkey (the 32-bit descending-weight key of the grid pass; 0 = a mutex that exists before every edge), ties by
 * ascending 64-bit global edge id.  rep[i] = smallest node id in i's cluster. */
typedef struct { uint32_t wkey; uint64_t gid; int64_t idx; } gedge_t;
static int cmp_gedge(const void* a, const void* b) {
  const gedge_t* x = (const gedge_t*)a; const gedge_t* y = (const gedge_t*)b;
  if (x->wkey != y->wkey) return x->wkey < y->wkey ? -1 : 1;
  return (x->gid > y->gid) - (x->gid < y->gid);
}
int mws_graph_oracle(int64_t n_nodes, int64_t n_edges, const uint32_t* wkey, const uint64_t* gid, const int64_t* eu, const int64_t* ev,
                     const uint8_t* attr, int64_t* rep) {
  gedge_t* e = (gedge_t*)malloc((size_t)(n_edges > 0 ? n_edges : 1) * sizeof(gedge_t));
  int64_t* parent = (int64_t*)malloc((size_t)n_nodes * sizeof(int64_t));
  int64_t* size = (int64_t*)malloc((size_t)n_nodes * sizeof(int64_t));
  list_t* mtx = (list_t*)calloc((size_t)n_nodes, sizeof(list_t));
  if (!e || !parent || !size || !mtx) return -1;
  for (int64_t i = 0; i < n_edges; ++i) { e[i].wkey = wkey[i]; e[i].gid = gid[i]; e[i].idx = i; }
  qsort(e, (size_t)n_edges, sizeof(gedge_t), cmp_gedge);
  for (int64_t i = 0; i < n_nodes; ++i) { parent[i] = i; size[i] = 1; }
  for (int64_t j = 0; j < n_edges; ++j) {
    const int64_t i = e[j].idx;
    int64_t ru = find(parent, eu[i]), rv = find(parent, ev[i]);
    if (ru == rv) continue;
    if (attr[i]) {
      if (has_mutex(parent, mtx, ru, rv)) continue;
      if (size[ru] < size[rv]) { int64_t t = ru; ru = rv; rv = t; }
      parent[rv] = ru; size[ru] += size[rv];
      for (int64_t q = 0; q < mtx[rv].n; ++q) push(&mtx[ru], mtx[rv].v[q]);
      free(mtx[rv].v); mtx[rv].v = NULL; mtx[rv].n = mtx[rv].cap = 0;
    } else {
      push(&mtx[ru], rv); push(&mtx[rv], ru);
    }
  }
  for (int64_t i = 0; i < n_nodes; ++i) size[i] = -1;   /* reuse: smallest member per root */
  for (int64_t i = 0; i < n_nodes; ++i) { const int64_t r = find(parent, i); if (size[r] < 0) size[r] = i; rep[i] = size[r]; }
  for (int64_t i = 0; i < n_nodes; ++i) free(mtx[i].v);
  free(mtx); free(parent); free(size); free(e);
  return 0;
}
