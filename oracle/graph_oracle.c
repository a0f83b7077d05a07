/*
 * graph_oracle.c -- plain-C restatement of the integer / index work on the CGVAE hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle/cgvae_oracle.py for who may use the oracle).
 *
 * Pinned by tests/test_oracle_c.py against the golden vectors produced by the reference itself
 * (tests/golden/g3_radius_graph.npz, g4_make_directed.npz, g5_scatter.npz).
 * Citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* get_neighbor_list, CoarseGrainingVAE/data.py:65-82.
 * dist = sqrt(((x_j - x_i)^2).sum over xyz) in fp32, sum order (x+y)+z (what torch's size-3
 * reduction produces); mask = dist <= cutoff (fp32 compare), diagonal cleared, nonzero in
 * row-major order, `undirected` keeps j > i.  Returns the number of pairs; `out` may be NULL
 * to count only.  Compile with -ffp-contract=off so no fma sneaks into the squared sum. */
int64_t orc_radius_graph(const float* xyz, int n, float cutoff, int undirected, int64_t* out) {
  int64_t count = 0;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) {
      if (j == i) continue;                                   /* data.py:76 */
      if (undirected && j <= i) continue;                     /* data.py:79-80 */
      volatile float dx = xyz[3 * j + 0] - xyz[3 * i + 0];
      volatile float dy = xyz[3 * j + 1] - xyz[3 * i + 1];
      volatile float dz = xyz[3 * j + 2] - xyz[3 * i + 2];
      volatile float sx = dx * dx, sy = dy * dy, sz = dz * dz;
      volatile float s = sx + sy;
      s = s + sz;
      float dist = sqrtf(s);                                  /* data.py:71-72 */
      if (dist <= cutoff) {                                   /* data.py:75 */
        if (out) { out[2 * count] = i; out[2 * count + 1] = j; }
        ++count;
      }
    }
  }
  return count;
}

/* make_directed, CoarseGrainingVAE/conv.py:10-20.  Returns the output length (e or 2e) and
 * sets *directed; out must hold 2e pairs. */
int64_t orc_make_directed(const int64_t* nbrs, int64_t e, int64_t* out, int* directed) {
  int gt = 0, lt = 0;
  for (int64_t k = 0; k < e; ++k) {
    if (nbrs[2 * k] > nbrs[2 * k + 1]) gt = 1;
    if (nbrs[2 * k + 1] > nbrs[2 * k]) lt = 1;
  }
  *directed = gt && lt;
  memcpy(out, nbrs, sizeof(int64_t) * 2 * (size_t)e);
  if (*directed) return e;
  for (int64_t k = 0; k < e; ++k) {                           /* cat([nbrs, nbrs.flip(1)]) */
    out[2 * (e + k)] = nbrs[2 * k + 1];
    out[2 * (e + k) + 1] = nbrs[2 * k];
  }
  return 2 * e;
}

/* Edge ids ordered by (key, other, original index) -- what the CSR plan K7 must reproduce bit for bit:
 * rowptr[n_rows+1], perm[e].  other may be NULL (order by key, then original index).  Two stable counting
 * sorts, least significant key first. */
static void counting_pass(const int64_t* key, int64_t stride, int64_t e, int n_rows, const int32_t* in, int32_t* out,
                          int32_t* rowptr) {
  memset(rowptr, 0, sizeof(int32_t) * (size_t)(n_rows + 1));
  for (int64_t k = 0; k < e; ++k) rowptr[key[k * stride] + 1]++;
  for (int r = 0; r < n_rows; ++r) rowptr[r + 1] += rowptr[r];
  int32_t* cursor = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_rows > 0 ? n_rows : 1));
  memcpy(cursor, rowptr, sizeof(int32_t) * (size_t)n_rows);
  for (int64_t p = 0; p < e; ++p) {
    const int32_t id = in ? in[p] : (int32_t)p;
    out[cursor[key[(int64_t)id * stride]]++] = id;
  }
  free(cursor);
}

void orc_csr_sorted(const int64_t* key, const int64_t* other, int64_t stride, int64_t e, int n_rows, int n_other,
                    int32_t* rowptr, int32_t* perm) {
  if (!other) { counting_pass(key, stride, e, n_rows, NULL, perm, rowptr); return; }
  int32_t* first = (int32_t*)malloc(sizeof(int32_t) * (size_t)(e > 0 ? e : 1));
  int32_t* rp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_other + 1));
  counting_pass(other, stride, e, n_other, NULL, first, rp);
  counting_pass(key, stride, e, n_rows, first, perm, rowptr);
  free(first); free(rp);
}

/* torch_scatter.scatter_add / scatter_mean semantics (requirements.txt:18; third party, restated)
 * accumulated in fp64 as an independent cross-check of the fp32 paths. */
void orc_scatter_f64(const float* src, const int64_t* index, int64_t e, int64_t c, int64_t n_out, int mean, double* out) {
  memset(out, 0, sizeof(double) * (size_t)(n_out * c));
  int64_t* cnt = (int64_t*)calloc((size_t)(n_out > 0 ? n_out : 1), sizeof(int64_t));
  for (int64_t k = 0; k < e; ++k) {
    double* row = out + index[k] * c;
    for (int64_t q = 0; q < c; ++q) row[q] += (double)src[k * c + q];
    cnt[index[k]]++;
  }
  if (mean)
    for (int64_t r = 0; r < n_out; ++r) {
      double d = (double)(cnt[r] > 1 ? cnt[r] : 1);            /* count clamped to >= 1 */
      for (int64_t q = 0; q < c; ++q) out[r * c + q] /= d;
    }
  free(cnt);
}

/* CG2ChannelIdx, CoarseGrainingVAE/cgvae.py:451-460: rank of each atom inside its bead. */
void orc_channel_index(const int64_t* mapping, int64_t n, int64_t n_beads, int64_t* out) {
  int64_t* seen = (int64_t*)calloc((size_t)(n_beads > 0 ? n_beads : 1), sizeof(int64_t));
  for (int64_t a = 0; a < n; ++a) out[a] = seen[mapping[a]]++;
  free(seen);
}
