// K0 radius graph and K7 CSR plan (see include/cgvae_hip.h).
//
// K0 restates get_neighbor_list (reference CoarseGrainingVAE/data.py:65-82) for a batch of
// frames: same membership rule bit for bit, same output order, no O(n^2) host loop.
// K7 turns the reference's unsorted scatter index (conv.py:10-20 make_directed output) into
// destination- and source-sorted CSR views so the fused kernels reduce without atomics.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdlib.h>
#include <type_traits>
#include "cgv_common.h"

namespace cgv {

// ------------------------------------------------------------------ K0 radius graph
// One wave per row i.  Lanes test 64 candidate j at a time; __ballot + popcount give each
// hit its rank, so a row's edges come out in ascending j exactly like torch.nonzero.
__device__ inline bool pair_hit(const float* __restrict__ xyz, int i, int j, float s_star) {
  // (dx*dx + dy*dy) + dz*dz with every operation individually rounded (no fma contraction):
  // bitwise what `.pow(2).sum(dim=2)` produces on the host.
  float dx = __fsub_rn(xyz[3 * j + 0], xyz[3 * i + 0]);
  float dy = __fsub_rn(xyz[3 * j + 1], xyz[3 * i + 1]);
  float dz = __fsub_rn(xyz[3 * j + 2], xyz[3 * i + 2]);
  float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
  return s <= s_star;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void radius_rows(const float* __restrict__ xyz,
                                                   const int* __restrict__ frame_of_node_ptr,  // frame_ptr
                                                   int n_frames, int n_nodes, float s_star, int undirected,
                                                   int* __restrict__ counts, const int* __restrict__ offsets,
                                                   int64_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  // locate the frame of this row (binary search in frame_ptr; uniform per wave)
  int lo = 0, hi = n_frames;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (frame_of_node_ptr[mid] <= row) lo = mid; else hi = mid;
  }
  const int f_beg = frame_of_node_ptr[lo], f_end = frame_of_node_ptr[lo + 1];
  const int j_beg = undirected ? row + 1 : f_beg;
  int total = 0;
  int64_t* dst = EMIT ? out + 2 * (int64_t)offsets[row] : nullptr;
  for (int j0 = j_beg; j0 < f_end; j0 += 64) {
    const int j = j0 + lane;
    bool hit = (j < f_end) && (j != row) && pair_hit(xyz, row, j, s_star);
    unsigned long long m = __ballot(hit);
    if (EMIT && hit) {
      int rank = total + __popcll(m & ((1ull << lane) - 1ull));
      dst[2 * rank + 0] = row;
      dst[2 * rank + 1] = j;
    }
    total += __popcll(m);
  }
  if (!EMIT && lane == 0) counts[row] = total;
}

// single-block exclusive scan (n up to a few million is fine: it runs once per dataset batch)
__global__ __launch_bounds__(1024) void exclusive_scan_i32(const int* __restrict__ in, int* __restrict__ out, int n) {
  __shared__ int wave_tot[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int i = base + tid;
    int x = (i < n) ? in[i] : 0;
    int incl = x;
    for (int d = 1; d < 64; d <<= 1) {
      int y = __shfl_up(incl, d);
      if (lane >= d) incl += y;
    }
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int k = 0; k < w; ++k) wave_off += wave_tot[k];
    int carry = carry_s;
    if (i < n) out[i] = carry + wave_off + incl - x;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wave_off + incl;
    __syncthreads();
  }
  if (tid == 0) out[n] = carry_s;
}

// ------------------------------------------------------------------ K7 CSR plan
__global__ void csr_keys(const int64_t* __restrict__ key, int stride, int n, int* __restrict__ keys,
                         int* __restrict__ vals) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  keys[e] = key ? (int)key[(int64_t)e * stride] : e;
  vals[e] = e;
}

__global__ void csr_gather(const int64_t* __restrict__ other, int stride, const int* __restrict__ eid, int n,
                           int* __restrict__ out) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int e = eid[p];
  out[p] = other ? (int)other[(int64_t)e * stride] : e;
}

// rowptr[i] = first sorted position whose key >= i  (i in [0, n_rows])
__global__ void csr_rowptr(const int* __restrict__ sorted_keys, int n, int n_rows, int* __restrict__ rowptr) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_rows) return;
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (sorted_keys[mid] < i) lo = mid + 1; else hi = mid;
  }
  rowptr[i] = lo;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static hipError_t sort_temp_bytes(int n, size_t* bytes) {
  int* k = nullptr;
  *bytes = 0;
  return rocprim::radix_sort_pairs(nullptr, *bytes, k, k, k, k, (size_t)(n > 0 ? n : 1), 0, 32, (hipStream_t)0);
}

// keys[p] = key[eid[p]] for the second pass of the two-key sort
__global__ void csr_regather(const int64_t* __restrict__ key, int stride, const int* __restrict__ eid, int n,
                             int* __restrict__ keys, int* __restrict__ vals) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int e = eid[p];
  keys[p] = (int)key[(int64_t)e * stride];
  vals[p] = e;
}

// One sorted view: edges ordered by (key, other, original index).  Two stable LSD passes of the 32-bit radix sort
// (by ``other`` first, then by ``key``): within a row the partner rows are visited in ascending order.
static int sorted_view(const int64_t* key, const int64_t* other, int stride, int E, int n_rows, int n_other, int* rowptr,
                       int* eid, int* key_sorted, int* other_sorted, char* ws, size_t ws_bytes, hipStream_t st) {
  int* keys_in = reinterpret_cast<int*>(ws);
  int* vals_in = reinterpret_cast<int*>(ws + align256(sizeof(int) * (size_t)E));
  char* temp = ws + 2 * align256(sizeof(int) * (size_t)E);
  size_t temp_bytes = ws_bytes - 2 * align256(sizeof(int) * (size_t)E);
  const int T = 256, B = (E + T - 1) / T;
  auto bits_for = [](int n) { int b = 1; while ((1ll << b) < (long long)(n > 1 ? n : 2)) ++b; return b; };
  if (E > 0) {
    hipError_t e = hipSuccess;
    if (other && key) {                            // pass 1: by the partner index (eid <- ids in that order)
      hipLaunchKernelGGL(csr_keys, dim3(B), dim3(T), 0, st, other, stride, E, keys_in, vals_in);
      e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, key_sorted, vals_in, eid, (size_t)E, 0, bits_for(n_other), st);
      if (e == hipSuccess) hipLaunchKernelGGL(csr_regather, dim3(B), dim3(T), 0, st, key, stride, eid, E, keys_in, vals_in);
    } else {
      hipLaunchKernelGGL(csr_keys, dim3(B), dim3(T), 0, st, key, stride, E, keys_in, vals_in);
    }
    if (e == hipSuccess)                           // pass 2 (stable): by the row key
      e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, key_sorted, vals_in, eid, (size_t)E, 0, bits_for(n_rows), st);
    if (e != hipSuccess) {
      set_error("cgv_csr_build: radix sort failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    hipLaunchKernelGGL(csr_gather, dim3(B), dim3(T), 0, st, other, stride, eid, E, other_sorted);
  }
  hipLaunchKernelGGL(csr_rowptr, dim3((n_rows + 1 + T - 1) / T), dim3(T), 0, st, key_sorted, E, n_rows, rowptr);
  return check_launch("cgv_csr_build");
}


// ------------------------------------------------------------------ K7b receiver-group order
// Shared-source walk of the fused message forward (equi_msg_grp.hip): RB consecutive receivers form a group
// whose edges -- one contiguous range of the destination-sorted view -- are re-ordered by (source, receiver),
// so that a wave gathers each source row ONCE for all the receivers of the group that see it.
__global__ void grp_keys_src(const int* __restrict__ src_d, int n, int* __restrict__ keys, int* __restrict__ vals) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  keys[p] = src_d[p];
  vals[p] = p;
}
__global__ void grp_keys_group(const int* __restrict__ dst_d, const int* __restrict__ perm, int n, int rb,
                               int* __restrict__ keys, int* __restrict__ vals) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int p = perm[q];
  keys[q] = dst_d[p] / rb;
  vals[q] = p;
}
__global__ void grp_gather(const int* __restrict__ dst_d, const int* __restrict__ src_d, const int* __restrict__ perm, int n,
                           int* __restrict__ dst_g, int* __restrict__ src_g, int* __restrict__ pos_g) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int p = perm[q];
  dst_g[q] = dst_d[p];
  src_g[q] = src_d[p];
  if (pos_g) pos_g[q] = p;
}
// meta[q] = { slot | head << 8 | last << 9 | mask << 16 , source of the NEXT step of this group (own source at the last step) }
// slot = receiver - group base; a step = a maximal run of edges of one (group, source) pair with strictly increasing
// receivers (at most RB edges; a duplicated edge opens a new step on the same source), mask = its slots,
// head = first edge of the step, last = the step is the group's last one.
__global__ void grp_meta(const int* __restrict__ dst_g, const int* __restrict__ src_g, int n, int rb,
                         int2* __restrict__ meta) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int d = dst_g[q], g = d / rb, s = src_g[q];
  // same step as the edge before it?
  auto joins = [&](int t) { return t > 0 && src_g[t - 1] == src_g[t] && dst_g[t - 1] / rb == dst_g[t] / rb && dst_g[t - 1] < dst_g[t]; };
  int mask = 1 << (d - g * rb);
  for (int t = q; joins(t); --t) mask |= 1 << (dst_g[t - 1] - g * rb);
  int t = q + 1;
  for (; t < n && joins(t); ++t) mask |= 1 << (dst_g[t] - g * rb);
  const bool more = t < n && dst_g[t] / rb == g;           // the group has another step behind this one
  const int next = more ? src_g[t] : s;
  meta[q] = make_int2((d - g * rb) | (joins(q) ? 0 : 0x100) | (more ? 0 : 0x200) | (mask << 16), next);
}


// One launch for the whole receiver-group order (the two radix passes above cost ~20 launches per batch, and the CLI's
// per-batch preparation is host-launch bound): a group's edges are a contiguous range of the dst-sorted view, so a
// block ranks them by (source, receiver, position) -- keys are unique, rank = number of smaller keys -- and writes
// dst_g / src_g / pos_g / meta_g of its range.  Keys sit in LDS up to GRP_LDS_KEYS edges per group (850 on the
// 2000-atom graph at rb = 2), beyond that they are re-read from global memory.
constexpr int GRP_LDS_KEYS = 2816;        // (with the 16 KB partner histogram: 61 KB of LDS)
constexpr int GRP_HIST_MAX = 4096, GRP_COUNT_FROM = 96;

// slot / head / mask / next source of position q of a group's sorted (source * rb + slot) sequence hi[0..L), as in grp_meta
template <typename HiAt>
__device__ __forceinline__ int2 grp_meta_of(HiAt hi_at, int q, int L, int rb) {
  auto src_of = [&](int t) { return (int)(hi_at(t) / (unsigned)rb); };
  auto slot_of = [&](int t) { const unsigned h = hi_at(t); return (int)(h - (h / (unsigned)rb) * (unsigned)rb); };
  auto joins = [&](int t) { return t > 0 && src_of(t - 1) == src_of(t) && slot_of(t - 1) < slot_of(t); };
  const int s = src_of(q), sl = slot_of(q);
  int mask = 1 << sl;
  for (int t = q; joins(t); --t) mask |= 1 << slot_of(t - 1);
  int t = q + 1;
  for (; t < L && joins(t); ++t) mask |= 1 << slot_of(t);
  const int next = t < L ? src_of(t) : s;
  return make_int2(sl | (joins(q) ? 0 : 0x100) | (t < L ? 0 : 0x200) | (mask << 16), next);
}

__global__ __launch_bounds__(256) void grp_build_k(const int* __restrict__ rowptr_d, const int* __restrict__ dst_d,
                                                   const int* __restrict__ src_d, int n_dst, int rb,
                                                   int* dst_g, int* src_g, int* pos_g, int2* __restrict__ meta,
                                                   unsigned long long* spill, int n_bins /* n_src * rb */) {
  // (dst_g / src_g are written and read back by other threads of the block between barriers on the spill path: not __restrict__)
  __shared__ unsigned long long keys[GRP_LDS_KEYS];
  __shared__ unsigned long long sorted[GRP_LDS_KEYS];
  const int g = blockIdx.x, node0 = g * rb;
  const int beg = rowptr_d[node0], end = rowptr_d[min(node0 + rb, n_dst)];
  const int L = end - beg;
  if (L <= 0) return;
  const bool in_lds = L <= GRP_LDS_KEYS;
  // The fill + rank passes are instantiated for each home of the keys (LDS, or the global spill range of a longer group): a
  // pointer chosen between the two at run time is generic, and the rank loop's key reads then are flat loads -- through
  // the vector memory path even when they land in LDS, and counted on both wait counters -- instead of ds_reads.
  auto fill_and_rank = [&](unsigned long long* __restrict__ k, auto lds_tag) {
    constexpr bool IN_LDS = decltype(lds_tag)::value;
    for (int t = threadIdx.x; t < L; t += blockDim.x) {
      const int p = beg + t;
      const unsigned hi = (unsigned)src_d[p] * (unsigned)rb + (unsigned)(dst_d[p] - node0);
      k[t] = ((unsigned long long)hi << 32) | (unsigned)p;
    }
    __syncthreads();
    // sorted position of every element: four keys per thread are ranked against every key read (850 keys per group on
    // the 2000-atom graph: 666 us per batch with one key per pass)
    for (int t0 = threadIdx.x; t0 < L; t0 += 4 * blockDim.x) {
      unsigned long long mine[4];
      int rank[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q) mine[q] = k[min(t0 + q * (int)blockDim.x, L - 1)];
      // (eight keys per trip: one key per trip waits a full LDS round trip for each of them)
      int u = 0;
      for (; u + 8 <= L; u += 8) {
        unsigned long long other[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) other[w] = k[u + w];
#pragma unroll
        for (int w = 0; w < 8; ++w)
#pragma unroll
          for (int q = 0; q < 4; ++q) rank[q] += other[w] < mine[q];
      }
      for (; u < L; ++u) {
        const unsigned long long other = k[u];
#pragma unroll
        for (int q = 0; q < 4; ++q) rank[q] += other < mine[q];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (t0 + q * (int)blockDim.x < L) {
          if (IN_LDS) {
            sorted[rank[q]] = mine[q];
          } else {
            pos_g[beg + rank[q]] = (int)(unsigned)(mine[q] & 0xffffffffull);  // scratch: the key's low word, in sorted order
            dst_g[beg + rank[q]] = (int)(mine[q] >> 32);                      // scratch: the key's high word
          }
        }
    }
  };
  // Long groups whose (source, slot) words fit the LDS histogram are placed by counting (as pj_row_sort_k's long rows):
  // offset of the word + rank by position among equal words; 850 keys per group on the 2000-atom graph made the comparison
  // ranking 136 us per batch.
  __shared__ int hist[GRP_HIST_MAX + 1];
  __shared__ int wave_tot[4];
  if (in_lds && L >= GRP_COUNT_FROM && n_bins <= GRP_HIST_MAX) {
    const int tid = threadIdx.x;
    for (int b = tid; b <= n_bins; b += 256) hist[b] = 0;
    __syncthreads();
    for (int t = tid; t < L; t += 256) {
      const int p = beg + t;
      const unsigned hi = min((unsigned)src_d[p] * (unsigned)rb + (unsigned)(dst_d[p] - node0), (unsigned)n_bins - 1u);
      keys[t] = ((unsigned long long)hi << 32) | (unsigned)p;
      atomicAdd(&hist[hi], 1);
    }
    __syncthreads();
    const int C = (n_bins + 255) / 256;
    int sum = 0;
    for (int b = tid * C; b < min((tid + 1) * C, n_bins); ++b) sum += hist[b];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if ((tid & 63) >= d) incl += up;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < (tid >> 6); ++w) run += wave_tot[w];
    for (int b = tid * C; b < min((tid + 1) * C, n_bins); ++b) { const int c = hist[b]; hist[b] = run; run += c; }
    if (tid == 255) hist[n_bins] = L;
    __syncthreads();
    for (int t = tid; t < L; t += 256) {
      const unsigned long long mine = keys[t];
      const unsigned hi = (unsigned)(mine >> 32);
      int pos = hist[hi];
      if (hist[hi + 1] - pos > 1)
        for (int u = 0; u < L; ++u) pos += (keys[u] >> 32) == hi && keys[u] < mine;
      sorted[pos] = mine;
    }
  } else if (in_lds) fill_and_rank(keys, std::true_type{});
  else fill_and_rank(spill + beg, std::false_type{});
  if (in_lds) {
    // the sorted keys stay in LDS: every output of a position is written once, no round trip through global memory
    __syncthreads();
    for (int q = threadIdx.x; q < L; q += blockDim.x) {
      const unsigned long long key = sorted[q];
      const unsigned hi = (unsigned)(key >> 32);
      const int s = (int)(hi / (unsigned)rb);
      pos_g[beg + q] = (int)(unsigned)(key & 0xffffffffull);
      src_g[beg + q] = s;
      dst_g[beg + q] = node0 + (int)(hi - (unsigned)s * (unsigned)rb);
      meta[beg + q] = grp_meta_of([&](int t) { return (unsigned)(sorted[t] >> 32); }, q, L, rb);
    }
    return;
  }
  __threadfence_block();
  __syncthreads();
  for (int q = beg + threadIdx.x; q < end; q += blockDim.x) src_g[q] = (int)((unsigned)dst_g[q] / (unsigned)rb);
  __threadfence_block();
  __syncthreads();
  // slot / head / mask / next source from the sorted (source, slot) sequence
  for (int q = beg + threadIdx.x; q < end; q += blockDim.x)
    meta[q] = grp_meta_of([&](int t) { return (unsigned)dst_g[beg + t]; }, q - beg, L, rb);
  __syncthreads();                                                         // every reader of the key words is done
  for (int q = beg + threadIdx.x; q < end; q += blockDim.x) {
    const int sl = (int)((unsigned)dst_g[q] - (unsigned)src_g[q] * (unsigned)rb);
    dst_g[q] = node0 + sl;
  }
}


// ------------------------------------------------------------------ K7 by rows (few launches)
// The same sorted view -- edges ordered by (row, partner, edge id) -- from five launches instead of the ~23 of two radix
// passes (the CLI's per-batch preparation is launch bound): count the rows' lengths, scan them into rowptr, drop every
// edge into its row in arbitrary order, then a block per row ranks the row's unique (partner, edge id) keys in LDS.
// Deterministic: the final position of an edge depends on the keys only.
__device__ __forceinline__ int csr_key_of(const int64_t* key, int stride, int e) { return key ? (int)key[(int64_t)e * stride] : e; }

__global__ void csr_count_k(const int64_t* __restrict__ key, int stride, int n, int* __restrict__ count) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) atomicAdd(count + csr_key_of(key, stride, e), 1);
}
__global__ void csr_drop_k(const int64_t* __restrict__ key, int stride, int n, const int* __restrict__ rowptr,
                           int* __restrict__ count, int* __restrict__ tmp) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int r = csr_key_of(key, stride, e);
  tmp[rowptr[r] + atomicSub(count + r, 1) - 1] = e;            // the row's slots are handed out last to first
}
constexpr int CSR_LDS_KEYS = 2048;
__global__ __launch_bounds__(128) void csr_row_sort_k(const int64_t* __restrict__ other, int stride,
                                                      const int* __restrict__ rowptr, const int* __restrict__ tmp,
                                                      int* __restrict__ eid, int* __restrict__ key_sorted,
                                                      int* __restrict__ other_sorted, unsigned long long* spill) {
  __shared__ unsigned long long keys[CSR_LDS_KEYS];
  const int r = blockIdx.x;
  const int beg = rowptr[r], L = rowptr[r + 1] - beg;
  if (L <= 0) return;
  // (instantiated per home of the keys: see grp_build_k)
  auto rank_row = [&](unsigned long long* __restrict__ k) {
    for (int t = threadIdx.x; t < L; t += blockDim.x) {
      const int e = tmp[beg + t];
      const unsigned partner = (unsigned)(other ? (int)other[(int64_t)e * stride] : e);
      k[t] = ((unsigned long long)partner << 32) | (unsigned)e;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < L; t += blockDim.x) {
      const unsigned long long mine = k[t];
      int rank = 0;
      for (int u = 0; u < L; ++u) rank += k[u] < mine;
      eid[beg + rank] = (int)(unsigned)(mine & 0xffffffffull);
      other_sorted[beg + rank] = (int)(mine >> 32);
      key_sorted[beg + rank] = r;
    }
  };
  if (L <= CSR_LDS_KEYS) rank_row(keys);
  else rank_row(spill + beg);
}

static size_t rows_view_bytes(int E, int n_rows_max) {
  return align256(sizeof(int) * ((size_t)n_rows_max + 1)) + align256(sizeof(int) * (size_t)(E > 0 ? E : 1)) +
         align256(sizeof(unsigned long long) * (size_t)(E > 0 ? E : 1));
}

static int rows_view(const int64_t* key, const int64_t* other, int stride, int E, int n_rows, int* rowptr, int* eid,
                     int* key_sorted, int* other_sorted, char* ws, hipStream_t st) {
  int* count = reinterpret_cast<int*>(ws);
  int* tmp = reinterpret_cast<int*>(ws + align256(sizeof(int) * ((size_t)n_rows + 1)));
  unsigned long long* spill = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(tmp) + align256(sizeof(int) * (size_t)(E > 0 ? E : 1)));
  const int T = 256, B = (E + T - 1) / T;
  hipError_t err = hipMemsetAsync(count, 0, sizeof(int) * ((size_t)n_rows + 1), st);
  if (err != hipSuccess) {
    set_error("cgv_csr_build: memset failed: %s", hipGetErrorString(err));
    return (int)err;
  }
  if (E > 0) hipLaunchKernelGGL(csr_count_k, dim3(B), dim3(T), 0, st, key, stride, E, count);
  hipLaunchKernelGGL(exclusive_scan_i32, dim3(1), dim3(1024), 0, st, count, rowptr, n_rows);
  if (E > 0) {
    hipLaunchKernelGGL(csr_drop_k, dim3(B), dim3(T), 0, st, key, stride, E, rowptr, count, tmp);
    hipLaunchKernelGGL(csr_row_sort_k, dim3(n_rows), dim3(128), 0, st, other, stride, rowptr, tmp, eid, key_sorted,
                       other_sorted, spill);
  }
  return check_launch("cgv_csr_build");
}

}  // namespace cgv

extern "C" {

int cgv_radius_graph_count(const float* xyz, const int32_t* frame_ptr, int n_frames, int n_nodes, float s_star,
                           int undirected, int32_t* counts, int32_t* offsets, void* stream) {
  CGV_REQUIRE(xyz && frame_ptr && counts && offsets && n_frames >= 0 && n_nodes >= 0, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (n_nodes > 0)
    hipLaunchKernelGGL(cgv::radius_rows<false>, dim3((n_nodes + 3) / 4), dim3(256), 0, st, xyz, frame_ptr, n_frames,
                       n_nodes, s_star, undirected, counts, (const int*)nullptr, (int64_t*)nullptr);
  hipLaunchKernelGGL(cgv::exclusive_scan_i32, dim3(1), dim3(1024), 0, st, counts, offsets, n_nodes);
  return cgv::check_launch("cgv_radius_graph_count");
}

int cgv_radius_graph_emit(const float* xyz, const int32_t* frame_ptr, int n_frames, int n_nodes, float s_star,
                          int undirected, const int32_t* offsets, int64_t* nbr_out, void* stream) {
  CGV_REQUIRE(xyz && frame_ptr && offsets && nbr_out && n_frames >= 0 && n_nodes >= 0, "bad argument");
  if (n_nodes == 0) return 0;
  hipLaunchKernelGGL(cgv::radius_rows<true>, dim3((n_nodes + 3) / 4), dim3(256), 0, (hipStream_t)stream, xyz, frame_ptr,
                     n_frames, n_nodes, s_star, undirected, (int*)nullptr, offsets, nbr_out);
  return cgv::check_launch("cgv_radius_graph_emit");
}

size_t cgv_csr_workspace_bytes(int n_edges) {
  // n_edges: pass max(edges, destination rows, source rows) -- the by-rows construction keeps one counter per row
  size_t temp = 0;
  if (cgv::sort_temp_bytes(n_edges, &temp) != hipSuccess) temp = (size_t)n_edges * 16 + (1 << 20);
  const size_t radix = 2 * cgv::align256(sizeof(int) * (size_t)(n_edges > 0 ? n_edges : 1)) + cgv::align256(temp) + 256;
  const size_t rows = cgv::rows_view_bytes(n_edges, n_edges) + 256;
  return radix > rows ? radix : rows;
}

int cgv_csr_build(const int64_t* dst, const int64_t* src, int stride, int n_edges, int n_dst, int n_src,
                  int32_t* rowptr_d, int32_t* eid_d, int32_t* dst_d, int32_t* src_d, int32_t* rowptr_s, int32_t* eid_s,
                  int32_t* dst_s, int32_t* src_s, void* workspace, size_t workspace_bytes, void* stream) {
  CGV_REQUIRE(n_edges >= 0 && n_dst >= 0 && n_src >= 0 && stride >= 1, "bad size");
  CGV_REQUIRE(rowptr_d && rowptr_s && workspace, "null output");
  CGV_REQUIRE(n_edges == 0 || (dst && eid_d && dst_d && src_d && eid_s && dst_s && src_s), "null edge array");
  hipStream_t st = (hipStream_t)stream;
  char* ws = reinterpret_cast<char*>(workspace);
  // by rows (5 launches per view) when the workspace has room for the row counters; cgv_set_option(CGV_OPT_CSR_BUILD, 1) forces the
  // two-radix-pass construction (tests compare the two)
  const int rows_max = n_dst > n_src ? n_dst : n_src;
  bool by_rows = workspace_bytes >= cgv::rows_view_bytes(n_edges, rows_max) && (((uintptr_t)workspace) & 7) == 0;
  if (cgv::option(CGV_OPT_CSR_BUILD) == 1) by_rows = false;
  if (by_rows) {
    int rc = cgv::rows_view(dst, src, stride, n_edges, n_dst, rowptr_d, eid_d, dst_d, src_d, ws, st);
    if (rc) return rc;
    return cgv::rows_view(src, dst, stride, n_edges, n_src, rowptr_s, eid_s, src_s, dst_s, ws, st);
  }
  if (workspace_bytes < cgv_csr_workspace_bytes(n_edges)) {
    cgv::set_error("cgv_csr_build: workspace too small");
    return CGV_E_WORKSPACE;
  }
  int rc = cgv::sorted_view(dst, src, stride, n_edges, n_dst, n_src, rowptr_d, eid_d, dst_d, src_d, ws, workspace_bytes, st);
  if (rc) return rc;
  return cgv::sorted_view(src, dst, stride, n_edges, n_src, n_dst, rowptr_s, eid_s, src_s, dst_s, ws, workspace_bytes, st);
}

size_t cgv_group_plan_radix_workspace_bytes(int n_edges) {
  size_t temp = 0;
  if (cgv::sort_temp_bytes(n_edges, &temp) != hipSuccess) temp = (size_t)n_edges * 16 + (1 << 20);
  return 4 * cgv::align256(sizeof(int) * (size_t)(n_edges > 0 ? n_edges : 1)) + cgv::align256(temp) + 256;
}

int cgv_group_plan_build_radix(const int32_t* dst_d, const int32_t* src_d, int n_edges, int n_dst, int n_src, int rb,
                         int32_t* dst_g, int32_t* src_g, int32_t* pos_g, int32_t* meta_g, void* workspace,
                         size_t workspace_bytes, void* stream) {
  CGV_REQUIRE(n_edges >= 0 && n_dst >= 0 && n_src >= 0 && rb >= 1 && rb <= 8, "bad size");
  if (n_edges == 0) return 0;
  CGV_REQUIRE(dst_d && src_d && dst_g && src_g && meta_g && workspace, "null pointer");
  CGV_REQUIRE((((uintptr_t)meta_g) & 7) == 0, "meta_g must be 8-byte aligned");
  if (workspace_bytes < cgv_group_plan_radix_workspace_bytes(n_edges)) {
    cgv::set_error("cgv_group_plan_build_radix: workspace too small");
    return CGV_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t E = (size_t)n_edges, slot = cgv::align256(sizeof(int) * E);
  char* ws = reinterpret_cast<char*>(workspace);
  int* k_in = reinterpret_cast<int*>(ws);
  int* v_in = reinterpret_cast<int*>(ws + slot);
  int* k_out = reinterpret_cast<int*>(ws + 2 * slot);
  int* v_out = reinterpret_cast<int*>(ws + 3 * slot);
  char* temp = ws + 4 * slot;
  size_t temp_bytes = workspace_bytes - 4 * slot;
  auto bits_for = [](int n) { int b = 1; while ((1ll << b) < (long long)(n > 1 ? n : 2)) ++b; return b; };
  const int T = 256, B = (n_edges + T - 1) / T;
  // the destination-sorted view is ordered by (receiver, source, edge id): a stable pass by source, then a stable
  // pass by group, leaves (group, source, receiver)
  hipLaunchKernelGGL(cgv::grp_keys_src, dim3(B), dim3(T), 0, st, src_d, n_edges, k_in, v_in);
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, k_in, k_out, v_in, v_out, E, 0, bits_for(n_src), st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(cgv::grp_keys_group, dim3(B), dim3(T), 0, st, dst_d, v_out, n_edges, rb, k_in, v_in);
    e = rocprim::radix_sort_pairs(temp, temp_bytes, k_in, k_out, v_in, v_out, E, 0, bits_for((n_dst + rb - 1) / rb), st);
  }
  if (e != hipSuccess) {
    cgv::set_error("cgv_group_plan_build_radix: radix sort failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(cgv::grp_gather, dim3(B), dim3(T), 0, st, dst_d, src_d, v_out, n_edges, dst_g, src_g, pos_g);
  hipLaunchKernelGGL(cgv::grp_meta, dim3(B), dim3(T), 0, st, dst_g, src_g, n_edges, rb, reinterpret_cast<int2*>(meta_g));
  return cgv::check_launch("cgv_group_plan_build_radix");
}

size_t cgv_group_plan_workspace_bytes(int n_edges) { return sizeof(unsigned long long) * (size_t)(n_edges > 0 ? n_edges : 1); }

int cgv_group_plan_build(const int32_t* rowptr_d, const int32_t* dst_d, const int32_t* src_d, int n_edges, int n_dst,
                              int n_src, int rb, int32_t* dst_g, int32_t* src_g, int32_t* pos_g, int32_t* meta_g,
                              void* workspace, size_t workspace_bytes, void* stream) {
  CGV_REQUIRE(n_edges >= 0 && n_dst >= 0 && n_src >= 0 && rb >= 1 && rb <= 8, "bad size");
  if (n_edges == 0 || n_dst == 0) return 0;
  CGV_REQUIRE(rowptr_d && dst_d && src_d && dst_g && src_g && pos_g && meta_g && workspace, "null pointer");
  CGV_REQUIRE((((uintptr_t)meta_g | (uintptr_t)workspace) & 7) == 0, "meta_g / workspace must be 8-byte aligned");
  CGV_REQUIRE((uint64_t)n_src * (uint64_t)rb < (1ull << 32), "n_src * rb must fit 32 bits");
  if (workspace_bytes < cgv_group_plan_workspace_bytes(n_edges)) {
    cgv::set_error("cgv_group_plan_build: workspace too small");
    return CGV_E_WORKSPACE;
  }
  const int groups = (n_dst + rb - 1) / rb;
  hipLaunchKernelGGL(cgv::grp_build_k, dim3(groups), dim3(256), 0, (hipStream_t)stream, rowptr_d, dst_d, src_d, n_dst, rb,
                     dst_g, src_g, pos_g, reinterpret_cast<int2*>(meta_g), reinterpret_cast<unsigned long long*>(workspace),
                     (int)((uint64_t)n_src * (uint64_t)rb > 0x7fffffffull ? 0x7fffffff : (uint64_t)n_src * (uint64_t)rb));
  return cgv::check_launch("cgv_group_plan_build");
}

}  // extern "C"
