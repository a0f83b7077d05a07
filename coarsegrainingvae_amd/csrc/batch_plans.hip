// Per-batch graph work in a handful of launches (K7 / K6 over job tables).
//
// What the reference does inside every forward -- make_directed (conv.py:10-20), preprocess_r / PainnRadialBasis /
// CosineEnvelope per block (conv.py:25-29, modules.py:148-172, 52-58) -- is, here, the in-place re-plan of a batch's
// sorted edge views and the recomputation of its edge records (graph.py: BatchGraph.update).  As separate calls that is
// 5 launches per sorted view (6 views per batch) + one per record array (9): ~56 launches whose ISSUE alone costs the
// host 0.33 ms per step on the chignolin config (tools/prefetch_probe.py).  Here every view / record array is a JOB in a
// table passed by value: count, scan, drop, row-sort run once over all views, the records once over all arrays:
// 5 launches per batch.  The kernels are those of graph.hip / geometry.hip with a job lookup in front.
#include <cstring>
#include "cgv_common.h"

namespace cgv {

constexpr int PJ_MAX = 8;
struct PlanJob {            // one sorted view: edges ordered by (key, other, edge id)
  const int64_t* key;       // NULL: key = edge id (the source side of a mapping plan)
  const int64_t* other;     // NULL: partner = edge id
  const float* ids_f32;     // non-NULL: the plan's index column is (int) ids_f32[e * stride] with `pad` mapped to `pad_to`
                            // (type ids of an nn.Embedding with padding_idx, read straight from nxyz[:, 0]); it replaces
                            // whichever of key / other is marked by ids_is_key
  int* rowptr;              // [n_rows + 1]
  int* eid;
  int* key_sorted;
  int* other_sorted;
  int* count;               // [n_rows + 1] scratch, ZERO on entry and on exit (every edge is counted once, dropped once)
  int* tmp;                 // [E] scratch
  unsigned long long* spill;  // [E] scratch (rows longer than the LDS key budget)
  int stride, E, n_rows;
  int edge_begin, row_begin;  // prefixes over the jobs
  int ids_is_key, pad, pad_to;
  int n_other;              // partners are in [0, n_other) (0: unknown -- rows are ranked by comparisons only)
};
struct PlanJobs { int n, total_edges, total_rows, pad; PlanJob job[PJ_MAX]; };

__device__ __forceinline__ int pj_ids(const PlanJob& p, int e) {
  const int t = (int)p.ids_f32[(int64_t)e * p.stride];
  return t == p.pad ? p.pad_to : t;
}
__device__ __forceinline__ int pj_key(const PlanJob& p, int e) {
  if (p.ids_f32 && p.ids_is_key) return pj_ids(p, e);
  return p.key ? (int)p.key[(int64_t)e * p.stride] : e;
}
__device__ __forceinline__ int pj_other(const PlanJob& p, int e) {
  if (p.ids_f32 && !p.ids_is_key) return pj_ids(p, e);
  return p.other ? (int)p.other[(int64_t)e * p.stride] : e;
}

__device__ __forceinline__ int pj_find_edge(const PlanJobs& J, int e) {
  int j = 0;
#pragma unroll
  for (int t = 1; t < PJ_MAX; ++t) if (t < J.n && J.job[t].edge_begin <= e) j = t;
  return j;
}

// Run-aggregated counter updates.  The edge lists of a radius graph arrive (mostly) sorted by one end, so in the view
// keyed by that end the 64 lanes of a wave address a handful of counters (125 edges per atom on the chignolin graph: 43 /
// 49 us for the two kernels with one atomic per edge) -- and in the view keyed by the OTHER end 64 different ones.  Lanes
// are grouped into RUNS of consecutive equal (job, key): a run's first lane issues one atomic for the whole run, and the
// runs are found with one shuffle and one ballot, no loop (a leader-election loop ran once per distinct counter of the
// wave: 64 rounds in the other-end view).  Equal keys in separate runs are simply two atomics.
struct PjRun { int head_lane, len; bool head; };
__device__ __forceinline__ PjRun pj_run(int tag /* >= 0 for an active lane, -1 otherwise */) {
  const int lane = threadIdx.x & 63;
  const int prev = __shfl_up(tag, 1);
  const bool head = lane == 0 || tag != prev;
  const unsigned long long heads = __ballot(head);
  const unsigned long long below = heads & ((2ull << lane) - 1ull);        // lane 63: 2 << 63 wraps to 0, mask = all ones
  const unsigned long long above = lane == 63 ? 0ull : heads & ~((2ull << lane) - 1ull);
  PjRun r;
  r.head = head;
  r.head_lane = 63 - __clzll((long long)below);
  const int end = above ? __ffsll((long long)above) - 1 : 64;
  r.len = end - r.head_lane;
  return r;
}

__global__ __launch_bounds__(256) void pj_count_k(PlanJobs J) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const bool active = e < J.total_edges;
  int* addr = nullptr;
  int tag = -1;
  if (active) {
    const int j = pj_find_edge(J, e);
    const PlanJob& p = J.job[j];
    const int r = pj_key(p, e - p.edge_begin);
    addr = p.count + r;
    tag = (j << 27) | r;                                                   // PJ_MAX = 8 jobs, rows < 2^27
  }
  const PjRun run = pj_run(tag);
  if (active && run.head) atomicAdd(addr, run.len);
}

__global__ __launch_bounds__(1024) void pj_scan_k(PlanJobs J) {        // block b: exclusive scan of job b's counts
  const PlanJob& p = J.job[blockIdx.x];
  const int* in = p.count;
  int* out = p.rowptr;
  const int n = p.n_rows;
  __shared__ int wave_tot[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int x = (i < n) ? in[i] : 0;
    int incl = x;
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(incl, d);
      if (lane >= d) incl += y;
    }
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int k = 0; k < w; ++k) wave_off += wave_tot[k];
    const int carry = carry_s;
    if (i < n) out[i] = carry + wave_off + incl - x;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wave_off + incl;
    __syncthreads();
  }
  if (tid == 0) out[n] = carry_s;
}

__global__ __launch_bounds__(256) void pj_drop_k(PlanJobs J) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const bool active = e < J.total_edges;
  int* addr = nullptr;
  int* tmp = nullptr;
  int le = 0, row_begin = 0, tag = -1;
  if (active) {
    const int j = pj_find_edge(J, e);
    const PlanJob& p = J.job[j];
    le = e - p.edge_begin;
    const int r = pj_key(p, le);
    addr = p.count + r;
    tmp = p.tmp;
    row_begin = p.rowptr[r];
    tag = (j << 27) | r;
  }
  const PjRun run = pj_run(tag);
  int base = 0;
  if (active && run.head) base = atomicSub(addr, run.len);              // the row's slots are handed out last to first
  base = __shfl(base, run.head_lane);
  if (active) tmp[row_begin + base - 1 - ((int)(threadIdx.x & 63) - run.head_lane)] = le;     // any order inside the row: the row sort fixes it
}

constexpr int PJ_LDS_KEYS = 2048;
// The body takes the key array as its own argument and is instantiated twice -- keys in LDS (every row that fits) or in
// the job's global spill array: one pointer selected at run time between the two is a GENERIC pointer, and every key
// read of the rank loop then is a flat_load (routed through the vector memory path even when it lands in LDS, and
// counted on both wait counters) instead of a ds_read.
__device__ __forceinline__ void pj_rank_row(unsigned long long* __restrict__ k, const PlanJob& p, int r, int beg, int L) {
  for (int t = threadIdx.x; t < L; t += blockDim.x) {
    const int e = p.tmp[beg + t];
    const unsigned partner = (unsigned)pj_other(p, e);
    k[t] = ((unsigned long long)partner << 32) | (unsigned)e;
  }
  __syncthreads();
  // rank = number of smaller keys (keys are unique).  A thread ranks FOUR of its keys against every key it reads: the
  // 2000-atom graph has 425 edges per row, and one key per pass made this kernel 506 us per batch there.
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
        p.eid[beg + rank[q]] = (int)(unsigned)(mine[q] & 0xffffffffull);
        p.other_sorted[beg + rank[q]] = (int)(mine[q] >> 32);
        p.key_sorted[beg + rank[q]] = r;
      }
  }
}

// Long rows whose partners come from a small range (the atom graph of the 2000-atom config: 425 edges per row, 2000
// nodes) are placed by COUNTING instead: a histogram of the row's partners in LDS, its exclusive scan, and an edge's
// position is the offset of its partner -- plus, only where a partner occurs more than once in the row, its rank by
// edge id among those (exactly the (partner, edge id) order of the comparison ranking, which costs L^2 64-bit compares
// per row: 118 us per batch there).
constexpr int PJ_HIST_MAX = 4096, PJ_COUNT_FROM = 96;
__device__ __forceinline__ void pj_count_row(unsigned long long* __restrict__ k, int* __restrict__ hist, const PlanJob& p, int r,
                                             int beg, int L) {
  const int nb = p.n_other, tid = threadIdx.x;
  __shared__ int wave_tot[2];
  for (int b = tid; b <= nb; b += 128) hist[b] = 0;
  __syncthreads();
  for (int t = tid; t < L; t += 128) {
    const int e = p.tmp[beg + t];
    const unsigned partner = min((unsigned)pj_other(p, e), (unsigned)nb - 1u);     // (in range by construction; an LDS guard)
    k[t] = ((unsigned long long)partner << 32) | (unsigned)e;
    atomicAdd(&hist[partner], 1);
  }
  __syncthreads();
  // exclusive scan of hist[0 .. nb): thread t owns the bins [t C, (t + 1) C)
  const int C = (nb + 127) / 128;
  int sum = 0;
  for (int b = tid * C; b < min((tid + 1) * C, nb); ++b) sum += hist[b];
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if ((tid & 63) >= d) incl += up;
  }
  if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
  __syncthreads();
  int run = incl - sum + (tid >= 64 ? wave_tot[0] : 0);
  for (int b = tid * C; b < min((tid + 1) * C, nb); ++b) { const int c = hist[b]; hist[b] = run; run += c; }
  if (tid == 127) hist[nb] = L;
  __syncthreads();
  for (int t = tid; t < L; t += 128) {
    const unsigned long long mine = k[t];
    const unsigned partner = (unsigned)(mine >> 32);
    int pos = hist[partner];
    if (hist[partner + 1] - pos > 1)                                 // the partner occurs again in this row: order by edge id
      for (int u = 0; u < L; ++u) pos += (k[u] >> 32) == partner && k[u] < mine;
    p.eid[beg + pos] = (int)(unsigned)(mine & 0xffffffffull);
    p.other_sorted[beg + pos] = (int)partner;
    p.key_sorted[beg + pos] = r;
  }
}

__global__ __launch_bounds__(128) void pj_row_sort_k(PlanJobs J) {   // block: one row of one job
  __shared__ unsigned long long keys[PJ_LDS_KEYS];
  __shared__ int hist[PJ_HIST_MAX + 1];
  int j = 0;
#pragma unroll
  for (int t = 1; t < PJ_MAX; ++t) if (t < J.n && J.job[t].row_begin <= (int)blockIdx.x) j = t;
  const PlanJob& p = J.job[j];
  const int r = blockIdx.x - p.row_begin;
  const int beg = p.rowptr[r], L = p.rowptr[r + 1] - beg;
  if (L <= 0) return;
  if (L <= PJ_LDS_KEYS && L >= PJ_COUNT_FROM && p.n_other > 0 && p.n_other <= PJ_HIST_MAX) pj_count_row(keys, hist, p, r, beg, L);
  else if (L <= PJ_LDS_KEYS) pj_rank_row(keys, p, r, beg, L);
  else pj_rank_row(p.spill + beg, p, r, beg, L);
}

// ------------------------------------------------------------------ edge records of several (view, cutoff) pairs
constexpr int GJ_MAX = 12;
struct GeomJob {
  const float* pos_dst; const float* pos_src;
  const int* dst; const int* src;
  const int2* meta;          // receiver-group records when non-NULL
  const float* coef;         // n pi / cutoff, n = 1..R
  float* geom;
  float cutoff;
  int E, R, GS, edge_begin;
};
struct GeomJobs { int n, total_edges; GeomJob job[GJ_MAX]; };

// A record is built in LDS (its length depends on the job's n_rbf) and leaves as 16-byte stores: 16 - 28 separate
// 4-byte stores per thread, 80+ bytes apart across the lanes, made this kernel 309 us per batch on the 2000-atom graph.
constexpr int GJ_REC_MAX = 28;           // geom_stride(20)
__global__ __launch_bounds__(256) void gj_records_k(GeomJobs J) {
  __shared__ __attribute__((aligned(16))) float recs[256][GJ_REC_MAX];
  const int e_first = blockIdx.x * 256;
  const int e_raw = e_first + threadIdx.x;
  const bool valid = e_raw < J.total_edges;
  const int e = valid ? e_raw : J.total_edges - 1;                  // (surplus threads rebuild the last record: no divergent barrier)
  int j = 0;
#pragma unroll
  for (int t = 1; t < GJ_MAX; ++t) if (t < J.n && J.job[t].edge_begin <= e) j = t;
  // a block whose 256 edges belong to ONE job (all but a handful) writes its records -- one contiguous range of the job's
  // array -- as consecutive 16-byte pieces over the lanes (whole lines); a block across a job border keeps a thread's own stores
  int j_first = 0, j_last = 0;
  {
    const int e_last = min(e_first + 255, J.total_edges - 1);
#pragma unroll
    for (int t = 1; t < GJ_MAX; ++t) {
      if (t < J.n && J.job[t].edge_begin <= e_first) j_first = t;
      if (t < J.n && J.job[t].edge_begin <= e_last) j_last = t;
    }
  }
  const bool one_job = j_first == j_last;
  const GeomJob& q = J.job[j];
  const int p = e - q.edge_begin;
  const int R = q.R, GS = q.GS;
  const float cutoff = q.cutoff;
  const float pi_f = 3.14159265358979323846f;
  // geometry.hip: edge_geometry, same expressions in the same order (conv.py:25-29, modules.py:148-172, 52-58)
  const float* a = q.pos_src + 3 * (size_t)q.src[p];
  const float* b = q.pos_dst + 3 * (size_t)q.dst[p];
  const float rx = __fsub_rn(a[0], b[0]), ry = __fsub_rn(a[1], b[1]), rz = __fsub_rn(a[2], b[2]);
  const float sx = __fadd_rn(__fmul_rn(rx, rx), 1e-8f);
  const float sy = __fadd_rn(__fmul_rn(ry, ry), 1e-8f);
  const float sz = __fadd_rn(__fmul_rn(rz, rz), 1e-8f);
  const float d = __fsqrt_rn(__fadd_rn(__fadd_rn(sx, sy), sz));
  float* g = recs[threadIdx.x];
  float4* dst4 = reinterpret_cast<float4*>(q.geom + (size_t)p * GS);      // GS % 4 == 0
  float env = 0.5f * (cosf(__fdiv_rn(__fmul_rn(pi_f, d), cutoff)) + 1.0f);
  const bool outside = d >= cutoff;
  if (outside) env = 0.0f;
  for (int n = 0; n < R; ++n) {
    const float c = q.coef[n];
    float val = (d == 0.0f) ? c : __fdiv_rn(sinf(__fmul_rn(c, d)), d);
    if (outside) val = 0.0f;
    g[n] = val * env;
  }
  g[R] = env;
  const float ux = __fdiv_rn(rx, d), uy = __fdiv_rn(ry, d), uz = __fdiv_rn(rz, d);
  if (q.meta) {
    const int2 m = q.meta[p];
    g[R + 1] = __int_as_float(m.x);
    g[R + 2] = ux; g[R + 3] = uy; g[R + 4] = uz;
    g[R + 5] = __int_as_float(m.y);
    for (int k = R + 6; k < GS; ++k) g[k] = 0.0f;
  } else {
    const int U = geom_unit_offset(R);
    for (int k = R + 1; k < U; ++k) g[k] = 0.0f;
    g[U + 0] = ux; g[U + 1] = uy; g[U + 2] = uz;
    g[U + 3] = ux; g[U + 4] = uy; g[U + 5] = uz;
    for (int k = U + 6; k < GS; ++k) g[k] = 0.0f;
  }
  if (!one_job) {
    if (valid)
      for (int k4 = 0; k4 < GS / 4; ++k4) dst4[k4] = reinterpret_cast<const float4*>(g)[k4];     // own record: no barrier needed
    return;
  }
  __syncthreads();
  const int n_rec = min(256, J.total_edges - e_first), GS4 = GS >> 2;
  float4* out4 = reinterpret_cast<float4*>(q.geom + (size_t)(e_first - q.edge_begin) * GS);
  for (int idx = threadIdx.x; idx < n_rec * GS4; idx += 256) {
    const int r = idx / GS4, k4 = idx - r * GS4;
    out4[idx] = reinterpret_cast<const float4*>(recs[r])[k4];
  }
}

// ------------------------------------------------------------------ loading a prepared batch into the captured buffers
// [n][4] rows (type id, x, y, z) of the atoms and of the beads: copied into the step's batch tensors and, as contiguous
// [n][3] coordinates, into the graph bundle -- one launch instead of four copies and two strided extractions.
__global__ __launch_bounds__(256) void batch_rows_k(const float4* __restrict__ src_a, float4* __restrict__ dst_a,
                                                    float* __restrict__ xyz_a, int n_a, const float4* __restrict__ src_b,
                                                    float4* __restrict__ dst_b, float* __restrict__ xyz_b, int n_b) {
  int i = blockIdx.x * 256 + threadIdx.x;
  const float4* src = src_a; float4* dst = dst_a; float* xyz = xyz_a;
  if (i >= n_a) { i -= n_a; src = src_b; dst = dst_b; xyz = xyz_b; if (i >= n_b) return; }
  const float4 r = src[i];
  dst[i] = r;
  xyz[3 * (size_t)i + 0] = r.y; xyz[3 * (size_t)i + 1] = r.z; xyz[3 * (size_t)i + 2] = r.w;
}

}  // namespace cgv

extern "C" {

/* nxyz / CG_nxyz of a prepared batch -> the captured step's tensors + the bundle's coordinate arrays (data.py:
 * copy_batch_into; reference: the DataLoader handing the next batch to cgvae.py:486). */
int cgv_batch_load_rows(const float* src_atoms, float* dst_atoms, float* xyz_atoms, int n_atoms, const float* src_beads,
                        float* dst_beads, float* xyz_beads, int n_beads, void* stream) {
  CGV_REQUIRE(src_atoms && dst_atoms && xyz_atoms && src_beads && dst_beads && xyz_beads && n_atoms >= 0 && n_beads >= 0, "bad argument");
  const int n = n_atoms + n_beads;
  if (n == 0) return 0;
  hipLaunchKernelGGL(cgv::batch_rows_k, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(src_atoms), reinterpret_cast<float4*>(dst_atoms), xyz_atoms, n_atoms,
                     reinterpret_cast<const float4*>(src_beads), reinterpret_cast<float4*>(dst_beads), xyz_beads, n_beads);
  return cgv::check_launch("cgv_batch_load_rows");
}

int cgv_plan_jobs_max(void) { return cgv::PJ_MAX; }
int cgv_plan_job_bytes(void) { return (int)sizeof(cgv::PlanJob); }
int cgv_geom_jobs_max(void) { return cgv::GJ_MAX; }
int cgv_geom_job_bytes(void) { return (int)sizeof(cgv::GeomJob); }

/* jobs_host: n records laid out as cgv::PlanJob (edge_begin / row_begin are filled in here). */
int cgv_plan_jobs_build(const void* jobs_host, int n_jobs, void* stream) {
  CGV_REQUIRE(jobs_host && n_jobs >= 1 && n_jobs <= cgv::PJ_MAX, "bad job table");
  cgv::PlanJobs J;
  std::memset(static_cast<void*>(&J), 0, sizeof(J));
  std::memcpy(static_cast<void*>(J.job), jobs_host, sizeof(cgv::PlanJob) * (size_t)n_jobs);
  J.n = n_jobs;
  int e = 0, r = 0;
  for (int j = 0; j < n_jobs; ++j) {
    cgv::PlanJob& p = J.job[j];
    CGV_REQUIRE(p.E >= 0 && p.n_rows >= 0 && p.n_rows < (1 << 27) && p.stride >= 1 && p.rowptr && p.count, "bad job");
    CGV_REQUIRE(p.E == 0 || (p.eid && p.key_sorted && p.other_sorted && p.tmp && p.spill), "null edge array");
    CGV_REQUIRE(p.n_other >= 0, "bad partner range");
    p.edge_begin = e; p.row_begin = r;
    e += p.E; r += p.n_rows;
  }
  J.total_edges = e; J.total_rows = r;
  hipStream_t st = (hipStream_t)stream;
  if (e > 0) hipLaunchKernelGGL(cgv::pj_count_k, dim3((e + 255) / 256), dim3(256), 0, st, J);
  hipLaunchKernelGGL(cgv::pj_scan_k, dim3(n_jobs), dim3(1024), 0, st, J);
  if (e > 0) {
    hipLaunchKernelGGL(cgv::pj_drop_k, dim3((e + 255) / 256), dim3(256), 0, st, J);
    if (r > 0) hipLaunchKernelGGL(cgv::pj_row_sort_k, dim3(r), dim3(128), 0, st, J);
  }
  return cgv::check_launch("cgv_plan_jobs_build");
}

int cgv_geom_jobs_build(const void* jobs_host, int n_jobs, void* stream) {
  CGV_REQUIRE(jobs_host && n_jobs >= 1 && n_jobs <= cgv::GJ_MAX, "bad job table");
  cgv::GeomJobs J;
  std::memset(static_cast<void*>(&J), 0, sizeof(J));
  std::memcpy(static_cast<void*>(J.job), jobs_host, sizeof(cgv::GeomJob) * (size_t)n_jobs);
  J.n = n_jobs;
  int e = 0;
  for (int j = 0; j < n_jobs; ++j) {
    cgv::GeomJob& q = J.job[j];
    CGV_REQUIRE(q.E >= 0 && q.R > 0 && cgv_rbf_supported(q.R) && q.coef && q.geom && q.pos_dst && q.pos_src && q.dst && q.src, "bad job");
    q.GS = q.meta ? cgv::geom_group_stride(q.R) : cgv::geom_stride(q.R);
    q.edge_begin = e;
    e += q.E;
  }
  J.total_edges = e;
  if (e == 0) return 0;
  hipLaunchKernelGGL(cgv::gj_records_k, dim3((e + 255) / 256), dim3(256), 0, (hipStream_t)stream, J);
  return cgv::check_launch("cgv_geom_jobs_build");
}

}  // extern "C"
