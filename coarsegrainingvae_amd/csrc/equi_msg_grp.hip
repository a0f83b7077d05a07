// K2g: fused EquiMessageBlock forward, shared-source walk.
//
// Same math as equi_msg.hip (reference conv.py:505-563, InvariantMessage 63-75, DistanceEmbed
// modules.py:192-197).  What changes is who shares what.  equi_msg_fwd_k gathers phi[j] / v[j] of the
// source once per EDGE: 3 KB per (edge, 128-channel wave), 13 GB per layer on the 2000-atom graph, and
// the kernel sits at the ~11-15 TB/s the L1/L2 gather path delivers for 8- and 12-byte lane fragments,
// with the packed-FMA pipe 75 % idle (profiles/).  Here RB consecutive receivers (atoms that
// follow each other in a molecule are neighbours in space, so their neighbour lists overlap almost
// completely) form a group; the group's edges are walked in (source, receiver) order
// (cgv_group_plan_build), and a wave keeps RB accumulator sets in registers: a source row is gathered
// ONCE per group and used for every receiver of the group that sees it -- gather bytes per edge fall by
// the mean multiplicity (3.0-3.6 of RB = 4 on the dense graphs), the filter FMAs are untouched.
//
// Pipeline of one wave: the rows of step t+1 (the next distinct source) are requested before the edges of
// step t are evaluated (two register buffers that swap roles, no copies), the scalar edge records and
// the per-edge (slot, step mask, next source) words -- folded into the record, one aligned 64-byte scalar load
// per edge at n_rbf = 10 -- run two edges ahead.  A step is a statically unrolled
// sequence of RB slots, each guarded by a wave-uniform test of the step's mask (scalar branch), so every
// accumulator set has its own code and stays in its registers.
#include <stdlib.h>
#include "cgv_common.h"
#include "equi_msg_dev.h"

namespace cgv {

typedef float q4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4v __attribute__((__vector_size__(16)));
constexpr int GRP_SC1 = 16;                          // buffer cache policy: agent scope (sc1) -- past the XCD's own L2

// Wave timeline of a few blocks (measurement only, compiled in with -DCGV_K2G_CLOCK=1; tools/k2g_clock_probe.py): slots
// [(sample * SPLIT + wave) * 8 + i], sample = one of 8 blocks spread over the grid; i: 0 entry, 1 first records + rows
// requested, 2 filter tile staged + rows in registers, 3 edge loop done, 4 partials exchanged, 5 stores issued, 6 edges walked, 7 shader cycles of the edge loop
#ifndef CGV_K2G_CLOCK
#define CGV_K2G_CLOCK 0
#endif
#if CGV_K2G_CLOCK
__constant__ unsigned long long* g_k2g_clock = nullptr;
#define K2G_TICK(i, val)                                                                                      \
  do {                                                                                                        \
    if (g_k2g_clock && (threadIdx.x & 63) == 0 && (blockIdx.x % (gridDim.x / 8)) == 3 && blockIdx.x / (gridDim.x / 8) < 8) \
      g_k2g_clock[((blockIdx.x / (gridDim.x / 8)) * SPLIT + (threadIdx.x >> 6)) * 8 + (i)] = (val);             \
  } while (0)
#else
#define K2G_TICK(i, val) do { } while (0)
#endif
// grid = 8 * items_per_xcd blocks (items = groups x channel tiles of 128), block = 64 * SPLIT threads: the SPLIT waves
// of a block take contiguous slices of the group's edge range and meet in LDS.
template <int R, int RB, int SPLIT, int PARTS>
__global__ __launch_bounds__(64 * SPLIT) void equi_msg_fwd_grp_k(
    const float* __restrict__ phi, const float* __restrict__ v, const float* __restrict__ geom /* group order */,
    const int* __restrict__ rowptr /* destination CSR */, const int* __restrict__ src_g,
    const float* __restrict__ Wd, const float* __restrict__ bd, float* __restrict__ ds, float* __restrict__ dv, int F,
    int n_dst, int items_per_xcd, int tiles, const float* __restrict__ s_res, const float* __restrict__ v_res,
    float* __restrict__ ws /* parts > 1: tickets + partial sums, cgv_equi_msg_grouped_workspace_bytes */) {
  constexpr int GS = geom_group_stride(R), NG = R + 6, MX = R + 1, MY = R + 5;   // record floats used; meta words
  constexpr int FILT = 3 * 128 * R, RED = SPLIT * RB * 8 * 64;
  __shared__ __attribute__((aligned(16))) float smem[FILT > RED ? FILT : RED];
  // work item = (channel tile, group, part), tile-major; XCD x (= blockIdx % 8) takes items [x * items_per_xcd, ...): an XCD
  // sweeps consecutive groups of ONE channel tile (at most two), so the rows its L2 must hold are one 3 KB tile slice of
  // the sources around ~100 consecutive groups, not all five slices (2000-atom graph: L2 hit rate 46 % -> see DESIGN.md).
  // PARTS blocks share a group's edge range (cut into PARTS x SPLIT wave slices): shorter
  // blocks in larger number spread evenly over the CUs where one block per (group, tile) is 3.25 blocks per CU
  // (chignolin) or lives 20-145 us depending on the group's degree (2000 atoms); their sums meet in ws (see the end).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n_groups = (n_dst + RB - 1) / RB;
  int item = xcd * items_per_xcd + slot;
  if (slot >= items_per_xcd || item >= n_groups * tiles * PARTS) return;
  const int part = item % PARTS;
  item /= PARTS;
  const int tile = item / n_groups;
  const int group = item - tile * n_groups;
  const int node0 = group * RB;
  const int node1 = min(node0 + RB, n_dst);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform -> record loads stay scalar
  const ChanPair cp = chan_pair(tile, lane, F);
  K2G_TICK(0, wall_clock64());

  int beg = rowptr[node0], end = rowptr[node1];
  {
    const int len = (end - beg + SPLIT * PARTS - 1) / (SPLIT * PARTS);
    beg = min(beg + (part * SPLIT + wave) * len, end);
    end = min(beg + len, end);
  }
  K2G_TICK(6, (unsigned long long)(end - beg));
  const unsigned row_bytes = 12u * (unsigned)F;              // bytes per node row of phi [3F] AND of v [F,3]
  const unsigned oc = 4u * (unsigned)cp.c, oF = 4u * (unsigned)F, ov = 12u * (unsigned)cp.c;
  const rsrc_t r_phi = make_rsrc(phi), r_v = make_rsrc(v);

  // The slice's first two records and the rows of its first source are requested BEFORE the filter tile is staged:
  // a block lives for only ~60 edges per wave on the chignolin graph, and rowptr -> record -> source id -> rows ->
  // filter tile as a serial chain of memory round trips was a third of its life.
  // Records run two edges ahead of the math (measured against one: 654 vs 719 us on the 2000-atom graph).  Scalar
  // loads return out of order, so every use waits for ALL of them (lgkmcnt(0)): what the second buffer buys is
  // that the record needed next has already landed when the wait for the newest request starts.
  const bool work = beg < end;
  const int last = end - 1;
  float gc[NG], g1[NG], g2[NG];
  RowBuf bufA, bufB;
  bufA.p0 = bufA.p1 = bufA.p2 = bufA.A = bufA.B = bufA.C = splat(0.f);
#pragma unroll
  for (int t = 0; t < NG; ++t) gc[t] = g1[t] = 0.f;
  if (work) {
#pragma unroll
    for (int t = 0; t < NG; ++t) gc[t] = geom[(size_t)beg * GS + t];
#pragma unroll
    for (int t = 0; t < NG; ++t) g1[t] = geom[(size_t)min(beg + 1, last) * GS + t];
    gather_row(bufA, r_phi, r_v, oc, oF, ov, (unsigned)src_g[beg] * row_bytes);
  }
  K2G_TICK(1, wall_clock64());

  f2 W0[R + 1], W1[R + 1], W2[R + 1];
  {
    const int sl[3] = {0, 1, 2};
    stage_filter_tile<R, 3>(smem, Wd, sl, F, tile * 128);
    const int cl = cp.c - tile * 128;                // even; clamped lanes stay inside the staged tile
    read_filter_rows2<R>(W0, smem, bd, cl, cp.c);
    read_filter_rows2<R>(W1, smem + 128 * R, bd, cl, F + cp.c);
    read_filter_rows2<R>(W2, smem + 2 * 128 * R, bd, cl, 2 * F + cp.c);
  }
  K2G_TICK(2, wall_clock64());
#if CGV_K2G_CLOCK
  const unsigned long long cyc0 = __builtin_readcyclecounter();      // shader clock (s_memtime)
#endif

  Acc acc[RB];
#pragma unroll
  for (int k = 0; k < RB; ++k) acc[k].s = acc[k].A = acc[k].B = acc[k].C = splat(0.f);

  if (work) {
    int e = beg;

    // one edge of the current step into the accumulators of receiver slot K (a wave-uniform test: scalar branch)
#define CGV_GRP_EDGE(BUF, K)                                                                      \
    if (((m >> (K)) & 1) && e < end) {                                                            \
      const int e2 = min(e + 2, last);                                                            \
      _Pragma("unroll") for (int t = 0; t < NG; ++t) g2[t] = geom[(size_t)e2 * GS + t];           \
      edge_math<R>(W0, W1, W2, gc, BUF, acc[(K) % RB]);                                           \
      _Pragma("unroll") for (int t = 0; t < NG; ++t) { gc[t] = g1[t]; g1[t] = g2[t]; }            \
      ++e;                                                                                        \
    }
#define CGV_GRP_STEP(BUF)                                      \
    {                                                          \
      CGV_GRP_EDGE(BUF, 0)                                     \
      if (RB > 1) { CGV_GRP_EDGE(BUF, 1) }                     \
      if (RB > 2) { CGV_GRP_EDGE(BUF, 2) CGV_GRP_EDGE(BUF, 3) } \
    }
#define CGV_GRP_META_X __float_as_int(gc[MX])
#define CGV_GRP_META_Y __float_as_int(gc[MY])

    // (An L2-warming vector load ahead of the scalar record stream, which pays off in the backward kernel, measured
    // neutral to slightly negative here -- 41.9 vs 40.8 us, 23.0 vs 20.3 us on the dipeptide graph -- and was dropped.)
    int m = (CGV_GRP_META_X >> 16) & ~((1 << (CGV_GRP_META_X & 0xff)) - 1);      // a slice may start inside a step
    while (true) {
      gather_row(bufB, r_phi, r_v, oc, oF, ov, (unsigned)CGV_GRP_META_Y * row_bytes);     // rows of the NEXT step
      CGV_GRP_STEP(bufA)
      if (e >= end) break;
      m = CGV_GRP_META_X >> 16;
      gather_row(bufA, r_phi, r_v, oc, oF, ov, (unsigned)CGV_GRP_META_Y * row_bytes);
      CGV_GRP_STEP(bufB)
      if (e >= end) break;
      m = CGV_GRP_META_X >> 16;
    }
#undef CGV_GRP_EDGE
#undef CGV_GRP_STEP
#undef CGV_GRP_META_X
#undef CGV_GRP_META_Y
  }

  // every wave deposits its partial sums; wave w then finishes receivers w, w + SPLIT, ... in a fixed order
  K2G_TICK(3, wall_clock64());
#if CGV_K2G_CLOCK
  K2G_TICK(7, __builtin_readcyclecounter() - cyc0);
#endif
  __syncthreads();                                   // all filter reads of the shared buffer are done
#pragma unroll
  for (int k = 0; k < RB; ++k) {
    float* r = smem + ((wave * RB + k) * 8) * 64 + lane;
    r[0] = acc[k].s.x; r[64] = acc[k].s.y; r[128] = acc[k].A.x; r[192] = acc[k].A.y;
    r[256] = acc[k].B.x; r[320] = acc[k].B.y; r[384] = acc[k].C.x; r[448] = acc[k].C.y;
  }
  __syncthreads();
  K2G_TICK(4, wall_clock64());
  if constexpr (PARTS == 1) {
    for (int k = wave; k < node1 - node0; k += SPLIT) {
      f2 s = splat(0.f), A = splat(0.f), B = splat(0.f), C = splat(0.f);
#pragma unroll
      for (int w = 0; w < SPLIT; ++w) {
        const float* r = smem + ((w * RB + k) * 8) * 64 + lane;
        s += f2{r[0], r[64]}; A += f2{r[128], r[192]}; B += f2{r[256], r[320]}; C += f2{r[384], r[448]};
      }
      if (cp.live) {
        const int node = node0 + k;
        if (s_res) s += ldpair<true>(s_res + (size_t)node * F, cp);          // emit h + ds (cgvae.py:287, 309, 391)
        stpair<true>(ds + (size_t)node * F, cp, s);
        if (v_res) {
          f2 rA, rB, rC;
          ldvec<true>(v_res + (size_t)node * F * 3, cp, rA, rB, rC);
          A += rA; B += rB; C += rC;
        }
        stvec<true>(dv + (size_t)node * F * 3, cp, A, B, C);
      }
    }
  } else {
    // PARTS blocks hold the group's sums between them: each leaves its own in its slot of the workspace -- agent-scope
    // write-through stores, acknowledged (vmcnt(0)) before its ticket is drawn: the hand-over of loss_tail.hip, valid
    // where stores count on vmcnt -- and the block that draws the group's LAST ticket adds the slots in part order
    // (the result does not depend on who arrives last) and stores.  Tickets reset themselves.
    // ws: [tiles * n_groups] tickets (padded to 256 bytes), then [tiles * n_groups][PARTS][RB][2][64][4] floats
    unsigned* ticket = reinterpret_cast<unsigned*>(ws) + (size_t)tile * n_groups + group;
    const size_t head = (((size_t)tiles * n_groups * sizeof(unsigned) + 255) & ~(size_t)255) / sizeof(float);
    const rsrc_t r_part = make_rsrc(ws + head + ((size_t)tile * n_groups + group) * (PARTS * RB * 512));
    for (int k = wave; k < RB; k += SPLIT) {
      f2 s = splat(0.f), A = splat(0.f), B = splat(0.f), C = splat(0.f);
#pragma unroll
      for (int w = 0; w < SPLIT; ++w) {
        const float* r = smem + ((w * RB + k) * 8) * 64 + lane;
        s += f2{r[0], r[64]}; A += f2{r[128], r[192]}; B += f2{r[256], r[320]}; C += f2{r[384], r[448]};
      }
      const unsigned at = (unsigned)((part * RB + k) * 2048) + 16u * (unsigned)lane;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, q4{s.x, s.y, A.x, A.y}), r_part, at, 0, GRP_SC1);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, q4{B.x, B.y, C.x, C.y}), r_part, at + 1024u, 0, GRP_SC1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // every wave's slot stores are acknowledged
    __shared__ unsigned s_last;
    if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == (unsigned)(PARTS - 1) ? 1u : 0u;
    __syncthreads();
    if (s_last) {
      if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int k = wave; k < node1 - node0; k += SPLIT) {
        f2 s = splat(0.f), A = splat(0.f), B = splat(0.f), C = splat(0.f);
        q4 x[PARTS], y[PARTS];
#pragma unroll
        for (int p = 0; p < PARTS; ++p) {
          const unsigned at = (unsigned)((p * RB + k) * 2048) + 16u * (unsigned)lane;
          x[p] = __builtin_bit_cast(q4, __builtin_amdgcn_raw_buffer_load_b128(r_part, at, 0, GRP_SC1));
          y[p] = __builtin_bit_cast(q4, __builtin_amdgcn_raw_buffer_load_b128(r_part, at + 1024u, 0, GRP_SC1));
        }
#pragma unroll
        for (int p = 0; p < PARTS; ++p) {
          s += f2{x[p].x, x[p].y}; A += f2{x[p].z, x[p].w}; B += f2{y[p].x, y[p].y}; C += f2{y[p].z, y[p].w};
        }
        if (cp.live) {
          const int node = node0 + k;
          if (s_res) s += ldpair<true>(s_res + (size_t)node * F, cp);
          stpair<true>(ds + (size_t)node * F, cp, s);
          if (v_res) {
            f2 rA, rB, rC;
            ldvec<true>(v_res + (size_t)node * F * 3, cp, rA, rB, rC);
            A += rA; B += rB; C += rC;
          }
          stvec<true>(dv + (size_t)node * F * 3, cp, A, B, C);
        }
      }
    }
  }
  K2G_TICK(5, wall_clock64());
}

// ------------------------------------------------------------------ records through LDS
// Same walk, but the per-edge records reach the wave through the VECTOR memory path and LDS instead of scalar loads.
// Why: scalar loads return out of order, so the compiler must wait for ALL of them before any use (lgkmcnt(0)) -- a
// record requested one edge ahead is the deepest prefetch that pays, and most of these loads miss the scalar cache (a
// record is touched once per (group, channel tile)).  Vector loads return in order: each wave streams its slice of the
// record array 32 edges at a time (two 16-byte-per-lane loads = 2 KiB, issued one piece ahead and parked in 8 VGPRs),
// drops it into a 64-record LDS ring, and reads an edge's record back with four broadcast ds_read_b128 (every lane
// the same address).  The FMAs then take the record from VGPRs (op_sel picks the half of a register pair), the
// step's mask / next source go to SGPRs with two v_readfirstlane per step.  Needs 16-float records (n_rbf 8 or 10).
constexpr int GRP_PIECE = 32;                 // edges per staged piece
constexpr int GRP_RING = 2 * GRP_PIECE;       // records per wave in LDS

template <int R, int RB, int SPLIT>
__global__ __launch_bounds__(64 * SPLIT) void equi_msg_fwd_grp_lds_k(
    const float* __restrict__ phi, const float* __restrict__ v, const float* __restrict__ geom /* group order */,
    const int* __restrict__ rowptr /* destination CSR */, const int* __restrict__ src_g,
    const float* __restrict__ Wd, const float* __restrict__ bd, float* __restrict__ ds, float* __restrict__ dv, int F,
    int n_dst, int n_edges, int items_per_xcd, int tiles, const float* __restrict__ s_res,
    const float* __restrict__ v_res) {
  constexpr int GS = geom_group_stride(R), MX = R + 1, MY = R + 5;
  static_assert(GS == 16, "the LDS ring holds 16-float records");
  constexpr int FILT = 3 * 128 * R, RED = SPLIT * RB * 8 * 64;
  __shared__ __attribute__((aligned(16))) float smem[FILT > RED ? FILT : RED];
  __shared__ __attribute__((aligned(16))) float recs[SPLIT][GRP_RING * 16];
  // work item = (channel tile, group), tile-major; XCD x (= blockIdx % 8) takes items [x * items_per_xcd, ...): an XCD
  // sweeps consecutive groups of ONE channel tile (at most two), so the rows its L2 must hold are one 3 KB tile slice of
  // the sources around ~100 consecutive groups, not all five slices (2000-atom graph: L2 hit rate 46 % -> see DESIGN.md)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n_groups = (n_dst + RB - 1) / RB;
  const int item = xcd * items_per_xcd + slot;
  if (slot >= items_per_xcd || item >= n_groups * tiles) return;
  const int tile = item / n_groups;
  const int group = item - tile * n_groups;
  const int node0 = group * RB;
  const int node1 = min(node0 + RB, n_dst);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const ChanPair cp = chan_pair(tile, lane, F);

  f2 W0[R + 1], W1[R + 1], W2[R + 1];
  {
    const int sl[3] = {0, 1, 2};
    stage_filter_tile<R, 3>(smem, Wd, sl, F, tile * 128);
    const int cl = cp.c - tile * 128;
    read_filter_rows2<R>(W0, smem, bd, cl, cp.c);
    read_filter_rows2<R>(W1, smem + 128 * R, bd, cl, F + cp.c);
    read_filter_rows2<R>(W2, smem + 2 * 128 * R, bd, cl, 2 * F + cp.c);
  }
  Acc acc[RB];
#pragma unroll
  for (int k = 0; k < RB; ++k) acc[k].s = acc[k].A = acc[k].B = acc[k].C = splat(0.f);

  int beg = rowptr[node0], end = rowptr[node1];
  {
    const int len = (end - beg + SPLIT - 1) / SPLIT;
    beg = min(beg + wave * len, end);
    end = min(beg + len, end);
  }
  const unsigned row_bytes = 12u * (unsigned)F;
  const unsigned oc = 4u * (unsigned)cp.c, oF = 4u * (unsigned)F, ov = 12u * (unsigned)cp.c;
  const rsrc_t r_phi = make_rsrc(phi), r_v = make_rsrc(v);

  if (beg < end) {
    float* ring = recs[wave];
    const float4* g4 = reinterpret_cast<const float4*>(geom);
    const int lim4 = n_edges * 4 - 1;                                  // last float4 of the record array
    // piece p = edges [beg + 32 p, beg + 32 p + 32) = 128 float4; lane l takes float4 l and 64 + l of it
    auto piece_ld = [&](int first_edge, float4& a, float4& b) {
      const int at = first_edge * 4 + lane;
      a = g4[min(at, lim4)];
      b = g4[min(at + 64, lim4)];
    };
    auto piece_st = [&](int first_edge, const float4& a, const float4& b) {
      float4* dst = reinterpret_cast<float4*>(ring + ((first_edge - beg) & (GRP_RING - 1)) * 16);
      dst[lane] = a;
      dst[64 + lane] = b;
    };
    float4 na, nb;
    piece_ld(beg, na, nb);
    piece_st(beg, na, nb);
    int staged_end = beg + GRP_PIECE;
    piece_ld(staged_end, na, nb);                                      // parked until the ring half is free

    float rec[16];
    auto fetch = [&](int e) {
      const float4* r4 = reinterpret_cast<const float4*>(ring + ((e - beg) & (GRP_RING - 1)) * 16);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 x = r4[t];
        rec[4 * t] = x.x; rec[4 * t + 1] = x.y; rec[4 * t + 2] = x.z; rec[4 * t + 3] = x.w;
      }
    };
    int e = beg;
    fetch(e);
    RowBuf bufA, bufB;
    gather_row(bufA, r_phi, r_v, oc, oF, ov, (unsigned)src_g[beg] * row_bytes);

#define CGV_GRP_EDGE(BUF, K)                                                                      \
    if (((m >> (K)) & 1) && e < end) {                                                            \
      edge_math<R>(W0, W1, W2, rec, BUF, acc[(K) % RB]);                                          \
      ++e;                                                                                        \
      fetch(min(e, end - 1));                                                                     \
    }
#define CGV_GRP_STEP(BUF)                                      \
    {                                                          \
      CGV_GRP_EDGE(BUF, 0)                                     \
      if (RB > 1) { CGV_GRP_EDGE(BUF, 1) }                     \
      if (RB > 2) { CGV_GRP_EDGE(BUF, 2) CGV_GRP_EDGE(BUF, 3) } \
    }
    // start of a step: refill the ring half whose edges are all consumed, then the step's mask / next source
#define CGV_GRP_HEAD(FIRST)                                                                                      \
    if (e >= staged_end - GRP_PIECE) {                                                                           \
      piece_st(staged_end, na, nb);                                                                              \
      staged_end += GRP_PIECE;                                                                                   \
      piece_ld(staged_end, na, nb);                                                                              \
    }                                                                                                            \
    {                                                                                                            \
      const int mx = __builtin_amdgcn_readfirstlane(__float_as_int(rec[MX]));                                    \
      nsrc = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(rec[MY]));                                  \
      m = (FIRST) ? ((mx >> 16) & ~((1 << (mx & 0xff)) - 1)) : (mx >> 16);                                       \
    }

    int m;
    unsigned nsrc;
    CGV_GRP_HEAD(true)
    while (true) {
      gather_row(bufB, r_phi, r_v, oc, oF, ov, nsrc * row_bytes);     // rows of the NEXT step
      CGV_GRP_STEP(bufA)
      if (e >= end) break;
      CGV_GRP_HEAD(false)
      gather_row(bufA, r_phi, r_v, oc, oF, ov, nsrc * row_bytes);
      CGV_GRP_STEP(bufB)
      if (e >= end) break;
      CGV_GRP_HEAD(false)
    }
#undef CGV_GRP_EDGE
#undef CGV_GRP_STEP
#undef CGV_GRP_HEAD
  }

  __syncthreads();
#pragma unroll
  for (int k = 0; k < RB; ++k) {
    float* r = smem + ((wave * RB + k) * 8) * 64 + lane;
    r[0] = acc[k].s.x; r[64] = acc[k].s.y; r[128] = acc[k].A.x; r[192] = acc[k].A.y;
    r[256] = acc[k].B.x; r[320] = acc[k].B.y; r[384] = acc[k].C.x; r[448] = acc[k].C.y;
  }
  __syncthreads();
  for (int k = wave; k < node1 - node0; k += SPLIT) {
    f2 s = splat(0.f), A = splat(0.f), B = splat(0.f), C = splat(0.f);
#pragma unroll
    for (int w = 0; w < SPLIT; ++w) {
      const float* r = smem + ((w * RB + k) * 8) * 64 + lane;
      s += f2{r[0], r[64]}; A += f2{r[128], r[192]}; B += f2{r[256], r[320]}; C += f2{r[384], r[448]};
    }
    if (cp.live) {
      const int node = node0 + k;
      if (s_res) s += ldpair<true>(s_res + (size_t)node * F, cp);
      stpair<true>(ds + (size_t)node * F, cp, s);
      if (v_res) {
        f2 rA, rB, rC;
        ldvec<true>(v_res + (size_t)node * F * 3, cp, rA, rB, rC);
        A += rA; B += rB; C += rC;
      }
      stvec<true>(dv + (size_t)node * F * 3, cp, A, B, C);
    }
  }
}

}  // namespace cgv

extern "C" {

int cgv_equi_msg_grouped_supported(int n_feat, int n_rbf, int rb) {
  bool r_ok = false;
#define CGV_X(r) r_ok = r_ok || n_rbf == r;
  CGV_RBF_LIST(CGV_X)
#undef CGV_X
  return r_ok && (n_feat % 2 == 0) && (n_rbf % 2 == 0) && (rb == 2 || rb == 4);
}

constexpr int CGV_GRP_PARTS_MAX = 4;

size_t cgv_equi_msg_grouped_workspace_bytes(int n_dst, int n_feat, int rb, int parts) {
  if (n_dst <= 0 || n_feat <= 0 || rb <= 0 || parts <= 1) return 0;
  const size_t items = (((size_t)n_feat + 127) / 128) * (((size_t)n_dst + rb - 1) / rb);
  return ((items * sizeof(unsigned) + 255) & ~(size_t)255) + items * (size_t)parts * (size_t)rb * 2048;
}

int cgv_equi_msg_fwd_grouped_parts(const float* phi, const float* v, const float* geom_g, const int32_t* rowptr_d,
                                   const int32_t* src_g, const float* Wd, const float* bd, float* ds,
                                   float* dv, int n_dst, int n_feat, int n_rbf, int rb, int64_t n_rows, int64_t n_edges,
                                   const float* s_res, const float* v_res, int parts, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  CGV_REQUIRE(n_dst >= 0 && n_feat > 0, "bad size");
  if (n_dst == 0) return 0;
  CGV_REQUIRE(phi && v && geom_g && rowptr_d && src_g && Wd && bd && ds && dv, "null pointer");
  CGV_REQUIRE(cgv_equi_msg_grouped_supported(n_feat, n_rbf, rb), "unsupported shape (need even n_feat / n_rbf, rb in {2,4})");
  CGV_REQUIRE(n_rows > 0 && (uint64_t)n_rows * 12u * (uint64_t)n_feat < 0x7fffffffull, "rows must lie within 2 GiB");
  CGV_REQUIRE((((uintptr_t)phi | (uintptr_t)v | (uintptr_t)ds | (uintptr_t)dv | (uintptr_t)s_res | (uintptr_t)v_res |
                (uintptr_t)bd) & 7) == 0 && ((((uintptr_t)Wd) | ((uintptr_t)geom_g)) & 15) == 0,
              "operands must be 8-byte (Wd, geom_g: 16-byte) aligned");
  CGV_REQUIRE(parts >= 1 && parts <= CGV_GRP_PARTS_MAX, "parts must be 1..4");
  hipStream_t st = (hipStream_t)stream;
  // waves per block: 4 (measured on the 2000-atom graph: 4 -> 654 us, 8 -> 942 us: one 8-wave block per CU leaves two
  // waves per SIMD; 6 -> 1194 us: a wave count that is not a multiple of the 4 SIMDs loads them unevenly).
  // cgv_set_option(CGV_OPT_GRP_WAVES, 8) is kept for A/B runs.
  const int split = cgv::option(CGV_OPT_GRP_WAVES);
  // record stream: "lds" (vector loads -> LDS ring -> VGPR operands; 16-float records: n_rbf 8 / 10) or "scalar"
  const bool lds = cgv::option(CGV_OPT_GRP_RECORDS) == 1;
  if (rb != 2 || split == 8 || lds) parts = 1;        // (several blocks per group: the default kernel only)
  float* ws = reinterpret_cast<float*>(workspace);
  if (parts > 1) {
    CGV_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 &&
                    workspace_bytes >= cgv_equi_msg_grouped_workspace_bytes(n_dst, n_feat, rb, parts),
                "parts > 1 needs a zero-filled workspace of cgv_equi_msg_grouped_workspace_bytes bytes");
  }
  const int tiles = (n_feat + 127) / 128;
  const int groups = (n_dst + rb - 1) / rb;
  const int gpx = (groups * tiles * parts + 7) / 8;   // work items per XCD
  const dim3 grid(8 * gpx);
#define CGV_GRP_LAUNCH(RBV, SP, PT)                                                                              \
  hipLaunchKernelGGL((cgv::equi_msg_fwd_grp_k<RBF, RBV, SP, PT>), grid, dim3(64 * SP), 0, st, phi, v, geom_g, rowptr_d, src_g, \
                     Wd, bd, ds, dv, n_feat, n_dst, gpx, tiles, s_res, v_res, ws)
#define CGV_GRP_PICK(RBV) \
  if (split == 8) CGV_GRP_LAUNCH(RBV, 8, 1); else CGV_GRP_LAUNCH(RBV, 4, 1)
  if (lds && (n_rbf == 8 || n_rbf == 10) && n_edges > 0 && n_edges < (1ll << 28)) {
    const int ne = (int)n_edges;
#define CGV_GRP_LDS(RV, RBV)                                                                                       \
  hipLaunchKernelGGL((cgv::equi_msg_fwd_grp_lds_k<RV, RBV, 4>), grid, dim3(256), 0, st, phi, v, geom_g, rowptr_d, src_g, \
                     Wd, bd, ds, dv, n_feat, n_dst, ne, gpx, tiles, s_res, v_res)
    if (n_rbf == 10) { if (rb == 2) CGV_GRP_LDS(10, 2); else CGV_GRP_LDS(10, 4); }
    else             { if (rb == 2) CGV_GRP_LDS(8, 2);  else CGV_GRP_LDS(8, 4); }
#undef CGV_GRP_LDS
    return cgv::check_launch("cgv_equi_msg_fwd_grouped");
  }
  CGV_DISPATCH_RBF(n_rbf, {
    if (parts == 2) { CGV_GRP_LAUNCH(2, 4, 2); }
    else if (parts == 3) { CGV_GRP_LAUNCH(2, 4, 3); }
    else if (parts == 4) { CGV_GRP_LAUNCH(2, 4, 4); }
    else if (rb == 2) { CGV_GRP_PICK(2); }
    else { CGV_GRP_PICK(4); }
  });
#undef CGV_GRP_PICK
#undef CGV_GRP_LAUNCH
  return cgv::check_launch("cgv_equi_msg_fwd_grouped");
}

int cgv_equi_msg_fwd_grouped(const float* phi, const float* v, const float* geom_g, const int32_t* rowptr_d,
                             const int32_t* src_g, const float* Wd, const float* bd, float* ds,
                             float* dv, int n_dst, int n_feat, int n_rbf, int rb, int64_t n_rows, int64_t n_edges,
                             const float* s_res, const float* v_res, void* stream) {
  return cgv_equi_msg_fwd_grouped_parts(phi, v, geom_g, rowptr_d, src_g, Wd, bd, ds, dv, n_dst, n_feat, n_rbf, rb, n_rows,
                                        n_edges, s_res, v_res, 1, nullptr, 0, stream);
}

}  // extern "C"

#if CGV_K2G_CLOCK
/* measurement builds only (tools/build_variant.sh equi_msg_grp -DCGV_K2G_CLOCK=1): wave timeline buffer, 8 x 4 x 8 uint64 */
extern "C" int cgv_k2g_debug_clock(uint64_t* buf) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(cgv::g_k2g_clock), &p, sizeof(p));
  return e == hipSuccess ? 0 : (int)e;
}
#endif
