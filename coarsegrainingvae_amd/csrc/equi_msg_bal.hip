// K2e: fused EquiMessageBlock forward, shared-source walk over EQUAL EDGE RANGES (a grid that is resident at once).
//
// Same math and the same walk as equi_msg_grp.hip (reference conv.py:505-563, InvariantMessage 63-75, DistanceEmbed
// modules.py:192-197): RB consecutive receivers form a group, a group's edges are walked in (source, receiver) order,
// a source row is gathered once per group.  What changes is WHO walks WHAT.  There a block is one (group, channel
// tile): 830 blocks of 18-31 us on the chignolin graph = 3.25 per CU, so a quarter of the CUs carry a fourth block
// while the others idle (wave timeline, profiles/r05_k2g_wave_timeline.txt: the vector pipe is saturated while the
// blocks run -- ~3 waves x 188 issue cycles per edge -- and the launch is 42 us for 16 us of issue); every block
// stages its own 15 KB filter tile (49 KB per CU in one burst), and on the 2000-atom graph a block lives 20-145 us
// depending on its group's degree.  Here the launch is 3 four-wave blocks per CU, all resident; the channel tile's edge array
// (all groups back to back, the plan's group order) is cut into equal ranges, one per wave, regardless of group
// boundaries: every SIMD carries the same number of waves with the same number of edges.
//
// A wave walks its range group by group.  A group that lies entirely inside the range is finished and stored by the
// wave alone.  A group cut by a range boundary is finished by the LAST of its contributing waves to arrive: each
// contributor leaves its partial sums in its own workspace slot (agent-scope write-through stores, acknowledged before
// the ticket: `s_waitcnt vmcnt(0)`, then a relaxed agent-scope atomic -- the hand-over of loss_tail.hip / optim.hip,
// valid on gfx9 where stores count on vmcnt), and the wave that draws the last ticket of the (tile, group) adds the
// slots in RANGE order -- the result does not depend on who arrives last.  Tickets reset themselves.
#include <stdlib.h>
#include "cgv_common.h"
#include "equi_msg_dev.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "ticket hand-over ordered by s_waitcnt vmcnt(0): gfx942 / gfx950 only"
#endif

namespace cgv {

// measurement builds (-DCGV_K2E_CLOCK=1, tools/k2g_clock_probe.py --balanced): per wave of 8 sampled blocks
// [(sample * 16 + wave) * 8 + i] (sample = one block per eighth of the grid); i: 0 entry, 1 first records + rows requested, 2 filter rows in registers, 3 edge loop
// done (incl. hand-overs), 4 ticks inside the hand-overs that were followed by another segment, 5 -, 6 edges walked, 7 segments
#ifndef CGV_K2E_CLOCK
#define CGV_K2E_CLOCK 0
#endif
#if CGV_K2E_CLOCK
__constant__ unsigned long long* g_k2e_clock = nullptr;
#define K2E_TICK(i, val)                                                                                       \
  do {                                                                                                         \
    if (g_k2e_clock && (threadIdx.x & 63) == 0 && (blockIdx.x % (gridDim.x / 8)) == 3)                          \
      g_k2e_clock[((blockIdx.x / (gridDim.x / 8)) * 16 + (threadIdx.x >> 6)) * 8 + (i)] = (val);                \
  } while (0)
#define K2E_ONLY(x) x
#else
#define K2E_TICK(i, val) do { } while (0)
#define K2E_ONLY(x)
#endif

typedef float q4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4v __attribute__((__vector_size__(16)));
static inline size_t bal_align256(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr int SC1 = 16;                              // buffer cache policy: agent scope (sc1) -- past the XCD's own L2

// A wave's partial sums of one group: [2 RB][64 lanes][4 floats] = RB x 2 KB; slot s of wave w at (w * 2 + s) * RB * 2 KB
template <int RB>
__device__ __forceinline__ void part_store(rsrc_t rp, unsigned slot, int lane, const Acc (&a)[RB]) {
  const unsigned base = slot * (unsigned)(RB * 2048) + 16u * (unsigned)lane;
#pragma unroll
  for (int k = 0; k < RB; ++k) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, q4{a[k].s.x, a[k].s.y, a[k].A.x, a[k].A.y}),
                                           rp, base + (unsigned)(2 * k) * 1024u, 0, SC1);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, q4{a[k].B.x, a[k].B.y, a[k].C.x, a[k].C.y}),
                                           rp, base + (unsigned)(2 * k + 1) * 1024u, 0, SC1);
  }
}
template <int RB>
__device__ __forceinline__ void part_add(rsrc_t rp, unsigned slot, int lane, Acc (&t)[RB]) {
  const unsigned base = slot * (unsigned)(RB * 2048) + 16u * (unsigned)lane;
  q4 x[2 * RB];
#pragma unroll
  for (int k = 0; k < 2 * RB; ++k)
    x[k] = __builtin_bit_cast(q4, __builtin_amdgcn_raw_buffer_load_b128(rp, base + (unsigned)k * 1024u, 0, SC1));
#pragma unroll
  for (int k = 0; k < RB; ++k) {
    t[k].s += f2{x[2 * k].x, x[2 * k].y};
    t[k].A += f2{x[2 * k].z, x[2 * k].w};
    t[k].B += f2{x[2 * k + 1].x, x[2 * k + 1].y};
    t[k].C += f2{x[2 * k + 1].z, x[2 * k + 1].w};
  }
}

// What the rare paths need (group hand-over, moving to the next group) travels in the lanes of ONE vector register
// instead of ~30 scalar registers: the walk itself keeps three 16-float edge records, two buffer descriptors and its
// counters in SGPRs (~70 of the 100 there are), and with the hand-over's pointers and bounds alive across it the
// compiler spills scalars to vector lanes INSIDE the edge loop (measured in the ISA: up to 36 v_readlane / v_writelane
// per edge beside its 42 packed FMAs).  Written once before the walk, read back where needed.
enum { CX_DS = 0, CX_DV = 2, CX_SRES = 4, CX_VRES = 6, CX_ROWPTR = 8, CX_TICKET = 10, CX_PART = 12, CX_SRCG = 14,
       CX_NDST = 16, CX_F = 17, CX_TILE = 18, CX_JW = 19, CX_WT = 20, CX_E = 21, CX_BEG0 = 22, CX_END0 = 23,
       CX_G = 24, CX_GE = 25, CX_FIRST = 26, CX_NFS = 27 };
__device__ __forceinline__ void cx_put(int& cx, int k, int val) { cx = (int)(threadIdx.x & 63) == k ? val : cx; }
__device__ __forceinline__ void cx_put_ptr(int& cx, int k, const void* p) {
  const unsigned long long u = (unsigned long long)p;
  cx_put(cx, k, (int)(unsigned)u);
  cx_put(cx, k + 1, (int)(unsigned)(u >> 32));
}
__device__ __forceinline__ int cx_get(int cx, int k) { return __builtin_amdgcn_readlane(cx, k); }
// (global address space in the TYPE: a generic pointer made from integers compiles to flat loads, whose results count as
// divergent -- they could come from private memory -- and drag the whole walk off the scalar path)
#define CGV_GLOBAL_AS __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ CGV_GLOBAL_AS T* cx_get_ptr(int cx, int k) {
  return reinterpret_cast<CGV_GLOBAL_AS T*>(((unsigned long long)(unsigned)cx_get(cx, k + 1) << 32) | (unsigned)cx_get(cx, k));
}
// uniform load of a value that must end up in a SCALAR register (the builtin readfirstlane of a provably uniform value is
// folded away, and the value -- loaded through the vector path -- then feeds scalar operands through a waterfall loop)
__device__ __forceinline__ int ldu(const CGV_GLOBAL_AS int* p) {
  const int x = *p;
  int r;
  asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(r) : "v"(x));
  return r;
}

// sums of receiver `node` -> ds / dv (+ the residual: emit h + ds, cgvae.py:287, 309, 391)
__device__ __forceinline__ void bal_store_out(int cx, const ChanPair& cp, int node, const Acc& a) {
  if (!cp.live) return;
  const int F = cx_get(cx, CX_F);
  const float* s_res = (const float*)cx_get_ptr<const float>(cx, CX_SRES);
  const float* v_res = (const float*)cx_get_ptr<const float>(cx, CX_VRES);
  f2 s = a.s, A = a.A, B = a.B, C = a.C;
  if (s_res) s += ldpair<true>(s_res + (size_t)node * F, cp);
  stpair<true>((float*)cx_get_ptr<float>(cx, CX_DS) + (size_t)node * F, cp, s);
  if (v_res) {
    f2 rA, rB, rC;
    ldvec<true>(v_res + (size_t)node * F * 3, cp, rA, rB, rC);
    A += rA; B += rB; C += rC;
  }
  stvec<true>((float*)cx_get_ptr<float>(cx, CX_DV) + (size_t)node * F * 3, cp, A, B, C);
}
template <int RB>
__device__ __forceinline__ void bal_store_group(int cx, const ChanPair& cp, int grp, const Acc (&a)[RB]) {
  const int n_dst = cx_get(cx, CX_NDST);
#pragma unroll
  for (int k = 0; k < RB; ++k)
    if (grp * RB + k < n_dst) bal_store_out(cx, cp, grp * RB + k, a[k]);
}
template <int RB>
__device__ __forceinline__ void bal_store_empty(int cx, const ChanPair& cp, int grp) {   // no edges: the residual passes through
  Acc z[RB];
#pragma unroll
  for (int k = 0; k < RB; ++k) z[k].s = z[k].A = z[k].B = z[k].C = splat(0.f);
  bal_store_group<RB>(cx, cp, grp, z);
}

// The sums of this wave's segment of group grp = edges [gb, ge) are complete (first: the segment opens the wave's range).
template <int RB>
__device__ __forceinline__ void bal_flush(int cx, const ChanPair& cp, int lane, int grp, int gb, int ge, bool first,
                                          const Acc (&acc)[RB]) {
  const int beg0 = cx_get(cx, CX_BEG0), end0 = cx_get(cx, CX_END0);
  if (gb >= beg0 && ge <= end0) { bal_store_group<RB>(cx, cp, grp, acc); return; }   // nobody else holds a part of it
  const unsigned E = (unsigned)cx_get(cx, CX_E), Wt = (unsigned)cx_get(cx, CX_WT);
  const int tile = cx_get(cx, CX_TILE), jw = cx_get(cx, CX_JW), n_dst = cx_get(cx, CX_NDST);
  const rsrc_t r_part = make_rsrc((const float*)cx_get_ptr<const float>(cx, CX_PART));
  const unsigned w0 = (unsigned)tile * Wt;
  part_store<RB>(r_part, (w0 + (unsigned)jw) * 2u + (first ? 0u : 1u), lane, acc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the slot is acknowledged before the ticket is drawn
  // the ranges that hold the group's first and last edge (range j = edges [j E / Wt, (j + 1) E / Wt))
  const int j_lo = (int)(((unsigned long long)(gb + 1) * Wt - 1ull) / E), j_hi = (int)(((unsigned long long)ge * Wt - 1ull) / E);
  unsigned* tk = (unsigned*)cx_get_ptr<unsigned>(cx, CX_TICKET) + (size_t)tile * ((n_dst + RB - 1) / RB) + grp;
  unsigned old = 0;
  if (lane == 0) old = atomicAdd(tk, 1u);
  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  // contributors = the NON-EMPTY ranges j_lo .. j_hi (with fewer edges than ranges a range holds one edge or none)
  const unsigned n_contrib = E >= Wt ? (unsigned)(j_hi - j_lo + 1) : (unsigned)(ge - gb);
  if (old != n_contrib - 1u) return;
  if (lane == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  Acc tot[RB];
#pragma unroll
  for (int k = 0; k < RB; ++k) tot[k].s = tot[k].A = tot[k].B = tot[k].C = splat(0.f);
  for (int j = j_lo; j <= j_hi; ++j) {
    // range j's part of the group is its LAST segment (slot 1) only if the range began in an earlier group
    const int bj = (int)((unsigned long long)j * E / Wt), bj1 = (int)((unsigned long long)(j + 1) * E / Wt);
    if (bj == bj1) continue;                                           // an empty range left nothing
    const bool later = j == j_lo && bj < gb;
    part_add<RB>(r_part, (w0 + (unsigned)j) * 2u + (later ? 1u : 0u), lane, tot);
  }
  bal_store_group<RB>(cx, cp, grp, tot);
}

// grid = (blocks per CU) x CUs (a multiple of 8), block = 64 WPB threads.  Block b sits on XCD b % 8; XCD x takes the
// contiguous run x of the (tile-major) block sequence, so an XCD's L2 holds one channel slice of the rows around its range.
template <int R, int RB, int WPB>
__global__ __launch_bounds__(64 * WPB) void equi_msg_fwd_bal_k(
    const float* __restrict__ phi, const float* __restrict__ v, const float* __restrict__ geom /* group order */,
    const int* __restrict__ rowptr /* destination CSR */, const int* __restrict__ src_g, const int* __restrict__ dst_g,
    const float* __restrict__ Wd, const float* __restrict__ bd, float* __restrict__ ds, float* __restrict__ dv, int F,
    int n_dst, int tiles, int bpt /* blocks per channel tile */, const float* __restrict__ s_res,
    const float* __restrict__ v_res, float* part, unsigned* ticket) {
  constexpr int GS = geom_group_stride(R), NG = R + 6, MX = R + 1, MY = R + 5;   // record floats used; meta words
  __shared__ __attribute__((aligned(16))) float smem[3 * 128 * R];
  const int nb8 = gridDim.x >> 3;
  const int idx = (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3);
  const int tile = idx / bpt;
  if (tile >= tiles) return;                                            // block-uniform: surplus blocks
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform -> record loads stay scalar
  const int Wt = bpt * WPB, jw = (idx - tile * bpt) * WPB + wave;      // ranges per tile; this wave's
  const ChanPair cp = chan_pair(tile, lane, F);
  K2E_TICK(0, wall_clock64());

  const int E = rowptr[n_dst];                                          // (read on the device: a replayed graph takes other edge counts)
  const int beg0 = (int)((unsigned long long)jw * (unsigned)E / (unsigned)Wt);
  const int end0 = (int)((unsigned long long)(jw + 1) * (unsigned)E / (unsigned)Wt);
  const bool work = beg0 < end0;
  const int last = end0 - 1;
  const unsigned row_bytes = 12u * (unsigned)F;              // bytes per node row of phi [3F] AND of v [F,3]
  const unsigned oc = 4u * (unsigned)cp.c, oF = 4u * (unsigned)F, ov = 12u * (unsigned)cp.c;
  const rsrc_t r_phi = make_rsrc(phi), r_v = make_rsrc(v);

  int cx = 0;
  cx_put_ptr(cx, CX_DS, ds); cx_put_ptr(cx, CX_DV, dv); cx_put_ptr(cx, CX_SRES, s_res); cx_put_ptr(cx, CX_VRES, v_res);
  cx_put_ptr(cx, CX_ROWPTR, rowptr); cx_put_ptr(cx, CX_TICKET, ticket); cx_put_ptr(cx, CX_PART, part);
  cx_put_ptr(cx, CX_SRCG, src_g);
  cx_put(cx, CX_NDST, n_dst); cx_put(cx, CX_F, F); cx_put(cx, CX_TILE, tile); cx_put(cx, CX_JW, jw); cx_put(cx, CX_WT, Wt);
  cx_put(cx, CX_E, E); cx_put(cx, CX_BEG0, beg0); cx_put(cx, CX_END0, end0);

  float gc[NG], g1[NG], g2[NG];
  RowBuf bufA, bufB;
  int g = 0;
  if (work) g = dst_g[beg0] / RB;
  K2E_TICK(1, wall_clock64());

  f2 W0[R + 1], W1[R + 1], W2[R + 1];
  {
    const int sl[3] = {0, 1, 2};
    stage_filter_tile<R, 3>(smem, Wd, sl, F, tile * 128);
    const int cl = cp.c - tile * 128;                // even; clamped lanes stay inside the staged tile
    read_filter_rows2<R>(W0, smem, bd, cl, cp.c);
    read_filter_rows2<R>(W1, smem + 128 * R, bd, cl, F + cp.c);
    read_filter_rows2<R>(W2, smem + 2 * 128 * R, bd, cl, 2 * F + cp.c);
  }
  K2E_TICK(2, wall_clock64());

  Acc acc[RB];
#pragma unroll
  for (int k = 0; k < RB; ++k) acc[k].s = acc[k].A = acc[k].B = acc[k].C = splat(0.f);

  int n_seg = 0;
  K2E_ONLY(unsigned long long t_flush = 0;)
  if (work) {
    int e = beg0;
    int seg_end, ge;
    {
      // (scalar registers by construction: behind stores through the context's pointers the compiler no longer proves
      // these arrays unclobbered, loads them through the vector path and keeps what depends on them in VGPRs)
      ge = ldu((const CGV_GLOBAL_AS int*)rowptr + min(g * RB + RB, n_dst));
      seg_end = min(end0, ge);
      if (beg0 == 0)
        for (int h = 0; h < g; ++h) bal_store_empty<RB>(cx, cp, h);     // groups without edges ahead of the first edge
    }

    // one edge of the current step into the accumulators of receiver slot K (a wave-uniform test: scalar branch)
#define CGV_BAL_EDGE(BUF, K)                                                                      \
    if (((m >> (K)) & 1) && e < seg_end) {                                                        \
      const int e2 = min(e + 2, last);                                                            \
      _Pragma("unroll") for (int t = 0; t < NG; ++t) g2[t] = geom[(size_t)e2 * GS + t];           \
      edge_math<R>(W0, W1, W2, gc, BUF, acc[(K) % RB]);                                           \
      _Pragma("unroll") for (int t = 0; t < NG; ++t) { gc[t] = g1[t]; g1[t] = g2[t]; }            \
      ++e;                                                                                        \
    }
#define CGV_BAL_STEP(BUF)                                      \
    {                                                          \
      CGV_BAL_EDGE(BUF, 0)                                     \
      if (RB > 1) { CGV_BAL_EDGE(BUF, 1) }                     \
      if (RB > 2) { CGV_BAL_EDGE(BUF, 2) CGV_BAL_EDGE(BUF, 3) } \
    }
#define CGV_BAL_META_X __float_as_int(gc[MX])
#define CGV_BAL_META_Y ((unsigned)__float_as_int(gc[MY]))
    // rows of the NEXT step (the plan's `next source`; a group's last step names its own source again)
#define CGV_BAL_NEXT(BUF) gather_row(BUF, r_phi, r_v, oc, oF, ov, CGV_BAL_META_Y * row_bytes);

    cx_put(cx, CX_G, g); cx_put(cx, CX_GE, ge); cx_put(cx, CX_FIRST, 1);
    while (true) {
      // A segment = this wave's part [e, seg_end) of group g.  Its first two records and first source's rows are
      // requested HERE for every segment, the range's first one included: one site defines what the walk starts from
      // (requested ahead of the filter staging for the first segment and afresh behind a hand-over for the others, the
      // two sets of definitions meet in copies / a spilled record set INSIDE the edge loop).
#pragma unroll
      for (int t = 0; t < NG; ++t) gc[t] = geom[(size_t)e * GS + t];
#pragma unroll
      for (int t = 0; t < NG; ++t) g1[t] = geom[(size_t)min(e + 1, last) * GS + t];
      gather_row(bufA, r_phi, r_v, oc, oF, ov, (unsigned)ldu((const CGV_GLOBAL_AS int*)src_g + e) * row_bytes);
      int m = (CGV_BAL_META_X >> 16) & ~((1 << (CGV_BAL_META_X & 0xff)) - 1);      // a range may start inside a step
      // the walk of equi_msg_fwd_grp_k, unchanged
      while (true) {
        CGV_BAL_NEXT(bufB)
        CGV_BAL_STEP(bufA)
        if (e >= seg_end) break;
        m = CGV_BAL_META_X >> 16;
        CGV_BAL_NEXT(bufA)
        CGV_BAL_STEP(bufB)
        if (e >= seg_end) break;
        m = CGV_BAL_META_X >> 16;
      }
      // hand the segment's sums over and move on to the next group with edges (what the walk does not need was parked
      // in the context)
      K2E_ONLY(const unsigned long long tf0 = wall_clock64();)
      int c2 = cx;
      asm volatile("" : "+v"(c2));                  // (the context is read HERE, not ahead of the walk)
      const CGV_GLOBAL_AS int* rp = cx_get_ptr<const int>(c2, CX_ROWPTR);
      const int nd = cx_get(c2, CX_NDST), e0 = cx_get(c2, CX_END0);
      g = cx_get(c2, CX_G); ge = cx_get(c2, CX_GE);
      bal_flush<RB>(c2, cp, lane, g, ldu(rp + g * RB), ge, cx_get(c2, CX_FIRST) != 0, acc);
      ++n_seg;
      if (e >= e0) {
        if (ge == e0) {    // the group ends with the range: groups without edges behind it belong to this wave
          const int n_groups = (nd + RB - 1) / RB;
          for (int h = g + 1; h < n_groups && ldu(rp + min(h * RB + RB, nd)) == e0; ++h) bal_store_empty<RB>(c2, cp, h);
        }
        break;
      }
#pragma unroll
      for (int k = 0; k < RB; ++k) acc[k].s = acc[k].A = acc[k].B = acc[k].C = splat(0.f);
      const int gb = ge;
      ++g; ge = ldu(rp + min(g * RB + RB, nd));
      while (ge == gb) { bal_store_empty<RB>(c2, cp, g); ++g; ge = ldu(rp + min(g * RB + RB, nd)); }
      seg_end = min(e0, ge);
      cx_put(cx, CX_G, g); cx_put(cx, CX_GE, ge); cx_put(cx, CX_FIRST, 0);
      K2E_ONLY(t_flush += wall_clock64() - tf0;)
    }
#undef CGV_BAL_EDGE
#undef CGV_BAL_STEP
#undef CGV_BAL_META_X
#undef CGV_BAL_META_Y
#undef CGV_BAL_NEXT
  } else if (jw == 0 && E == 0) {
    const int n_groups = (n_dst + RB - 1) / RB;
    for (int h = 0; h < n_groups; ++h) bal_store_empty<RB>(cx, cp, h);  // a graph without edges
  }
  K2E_TICK(3, wall_clock64());
  K2E_TICK(6, (unsigned long long)(end0 - beg0));
  K2E_TICK(7, (unsigned long long)n_seg);
  K2E_TICK(4, t_flush);
}

static int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && c > 0)
      n = c;
    else
      n = 256;
  }
  return n;
}

constexpr int BAL_WPB = 4;               // waves per block: with 12- or 16-wave blocks (launch bounds 768 / 1024) the compiler rotates
                                         // the walk's record sets through copies and spills one to vector lanes inside the edge loop
constexpr int BAL_BPC_MAX = 4;           // blocks per CU the workspace is sized for

}  // namespace cgv

extern "C" {

int cgv_equi_msg_balanced_supported(int n_feat, int n_rbf, int rb) { return cgv_equi_msg_grouped_supported(n_feat, n_rbf, rb) && rb == 2; }

size_t cgv_equi_msg_balanced_workspace_bytes(int n_dst, int n_feat, int rb) {
  if (n_dst <= 0 || n_feat <= 0 || rb <= 0) return 0;
  const size_t tiles = ((size_t)n_feat + 127) / 128, groups = ((size_t)n_dst + rb - 1) / rb;
  const size_t tickets = cgv::bal_align256(tiles * groups * sizeof(unsigned));
  const size_t waves = (size_t)(cgv::cu_count() & ~7) * cgv::BAL_BPC_MAX * cgv::BAL_WPB;
  return tickets + waves * 2 * (size_t)rb * 2048;
}

int cgv_equi_msg_fwd_balanced(const float* phi, const float* v, const float* geom_g, const int32_t* rowptr_d,
                              const int32_t* src_g, const int32_t* dst_g, const float* Wd, const float* bd, float* ds,
                              float* dv, int n_dst, int n_feat, int n_rbf, int rb, int64_t n_rows, const float* s_res,
                              const float* v_res, void* workspace, size_t workspace_bytes, void* stream) {
  CGV_REQUIRE(n_dst >= 0 && n_feat > 0, "bad size");
  if (n_dst == 0) return 0;
  CGV_REQUIRE(phi && v && geom_g && rowptr_d && src_g && dst_g && Wd && bd && ds && dv && workspace, "null pointer");
  CGV_REQUIRE(cgv_equi_msg_balanced_supported(n_feat, n_rbf, rb), "unsupported shape (need even n_feat / n_rbf, rb = 2)");
  CGV_REQUIRE(n_rows > 0 && (uint64_t)n_rows * 12u * (uint64_t)n_feat < 0x7fffffffull, "rows must lie within 2 GiB");
  CGV_REQUIRE((((uintptr_t)phi | (uintptr_t)v | (uintptr_t)ds | (uintptr_t)dv | (uintptr_t)s_res | (uintptr_t)v_res |
                (uintptr_t)bd) & 7) == 0 && ((((uintptr_t)Wd) | ((uintptr_t)geom_g) | (uintptr_t)workspace) & 15) == 0,
              "operands must be 8-byte (Wd, geom_g, workspace: 16-byte) aligned");
  CGV_REQUIRE(workspace_bytes >= cgv_equi_msg_balanced_workspace_bytes(n_dst, n_feat, rb), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int tiles = (n_feat + 127) / 128;
  const int groups = (n_dst + rb - 1) / rb;
  // blocks per CU: 3 (default: the kernel's 134 VGPRs admit three 4-wave blocks per CU, so a grid of 3 x CUs blocks is
  // resident at once with the same number of waves on every SIMD), 1..4 for A/B runs
  int bpc = cgv::option(CGV_OPT_MSG_FWD_BALANCED);
  bpc = bpc < 1 ? 1 : (bpc > cgv::BAL_BPC_MAX ? cgv::BAL_BPC_MAX : bpc);
  const int blocks = (cgv::cu_count() & ~7) * bpc;
  CGV_REQUIRE(tiles <= blocks, "more channel tiles than blocks");
  const int bpt = blocks / tiles;
  unsigned* ticket = reinterpret_cast<unsigned*>(workspace);
  float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + cgv::bal_align256((size_t)tiles * groups * sizeof(unsigned)));
  CGV_DISPATCH_RBF(n_rbf, {
    hipLaunchKernelGGL((cgv::equi_msg_fwd_bal_k<RBF, 2, cgv::BAL_WPB>), dim3(blocks), dim3(64 * cgv::BAL_WPB), 0, st, phi, v, geom_g,
                       rowptr_d, src_g, dst_g, Wd, bd, ds, dv, n_feat, n_dst, tiles, bpt, s_res, v_res, part, ticket);
  });
  return cgv::check_launch("cgv_equi_msg_fwd_balanced");
}

}  // extern "C"

#if CGV_K2E_CLOCK
/* measurement builds only (tools/build_variant.sh equi_msg_bal -DCGV_K2E_CLOCK=1): wave timeline buffer, 8 x 16 x 8 uint64 */
extern "C" int cgv_k2e_debug_clock(uint64_t* buf) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(cgv::g_k2e_clock), &p, sizeof(p));
  return e == hipSuccess ? 0 : (int)e;
}
#endif
