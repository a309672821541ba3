// Host side of K1 for K8: a locality ORDER of the destination rows of a graph whose own numbering has none.
//
// K8 (wide.hip) cuts the rows into tiles of 64 and stages the union of the source rows a tile names; that union
// must stay within GWEN_TILE_UNION rows, which holds when 64 consecutive rows are a compact patch of the mesh --
// true for the locality-ordered numberings of gwen_amd/mesh.py, not for a caller's arbitrary edge_index (the
// reference takes whatever the dataset's edge_index says: /root/reference/src/gwen/utils.py:175-176,
// models_gnn.py:147-149).  This pass GROWS the patches itself: breadth-first balls of 64 still-unassigned rows,
// each started next to the patches already cut, over the in-neighbour lists of the prepared CSR.  The result is a
// permutation (new position -> old row); the caller relabels the CSR with it, runs the stack in the permuted
// numbering and permutes the output back -- every row still sums its entries in stored order, so results are
// bitwise those of the unpermuted kernels.  O(N + E) on the host, once per graph.
#include "common.h"
#include <vector>

extern "C" int gwen_cluster_rows64_host(const int32_t *rowptr, const int32_t *col, int64_t N, int64_t N_src,
                                        int32_t *perm) {
  if (N < 0 || (N > 0 && (!rowptr || !col || !perm)) || N >= (int64_t(1) << 31)) return GWEN_EINVAL;
  if (N_src != N) return GWEN_EINVAL;                       // square graphs: rows and sources share one numbering
  const int32_t n = (int32_t)N;
  std::vector<uint8_t> state(n, 0);                         // 0 free, 1 queued as a seed candidate, 2 assigned
  std::vector<int32_t> frontier, ball;                      // global seed candidates (FIFO), the patch being grown
  frontier.reserve(n);
  ball.reserve(GWEN_TILE_ROWS);
  size_t fhead = 0;
  int32_t next_free = 0, out = 0;
  while (out < n) {
    // seed: the oldest candidate next to an earlier patch, else the lowest free row (a new component)
    int32_t seed = -1;
    while (fhead < frontier.size()) {
      const int32_t c = frontier[fhead++];
      if (state[c] != 2) { seed = c; break; }
    }
    if (seed < 0) {
      while (state[next_free] == 2) ++next_free;
      seed = next_free;
    }
    // breadth-first ball of up to 64 free rows around the seed
    ball.clear();
    ball.push_back(seed);
    state[seed] = 2;
    for (size_t head = 0; head < ball.size() && (int)ball.size() < GWEN_TILE_ROWS; ++head) {
      const int32_t r = ball[head];
      for (int32_t s = rowptr[r]; s < rowptr[r + 1] && (int)ball.size() < GWEN_TILE_ROWS; ++s) {
        const int32_t c = col[s];
        if (c < 0 || c >= n) return GWEN_ERANGE;
        if (state[c] != 2) {
          state[c] = 2;
          ball.push_back(c);
        }
      }
    }
    for (const int32_t r : ball) perm[out++] = r;
    // the free neighbours of the patch become seed candidates
    for (const int32_t r : ball)
      for (int32_t s = rowptr[r]; s < rowptr[r + 1]; ++s) {
        const int32_t c = col[s];
        if (c < 0 || c >= n) return GWEN_ERANGE;
        if (state[c] == 0) {
          state[c] = 1;
          frontier.push_back(c);
        }
      }
  }
  return GWEN_OK;
}
