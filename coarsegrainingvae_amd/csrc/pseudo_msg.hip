// K3: fused EquiMessagePsuedo (reference CoarseGrainingVAE/conv.py:180-242), forward + backward.
//
// With i = dst (receiver), j = src, u = unit_e, q_k = phi[j,kF+f] * w_k(e,f), k = 0..8
// (w_k = distance filter, modules.py:192-197), per edge and channel f (conv.py:205-217):
//     T_h   = q0 s_i                                  T_hbar = v_i . vbar_j      (no filter)
//     T_v   = q1 u + q2 v_j + q3 (v_i x vbar_j) + q4 sbar_i vbar_j
//     T_vb  = q5 vbar_j + q6 sbar_i v_j + q7 (v_i x v_j) + q8 (vbar_i x vbar_j)
// and dh, dhbar, dv, dvbar are their sums over the edges of receiver i (conv.py:221-240).
// The reference's cross products run along the last (xyz) axis whenever E != 3 and F != 3.
//
// Backward (autograd of the above; upstream gh, ghb, gv, gvb at receiver i):
//   filter side   gq0 = gh s_i, gq1 = gv.u, gq2 = gv.v_j, gq3 = gv.(v_i x vbar_j), gq4 = sbar_i (gv.vbar_j),
//                 gq5 = gvb.vbar_j, gq6 = sbar_i (gvb.v_j), gq7 = gvb.(v_i x v_j), gq8 = gvb.(vbar_i x vbar_j)
//                 g_phi[j,k] += gq_k w_k ;  gWd[k][n] += gq_k phi_k a_n ;  gbd[k] += gq_k phi_k env
//   receiver side g_s[i] += gh q0 ; g_sbar[i] += q4 (gv.vbar_j) + q6 (gvb.v_j)
//                 g_v[i] += ghb vbar_j + q3 (vbar_j x gv) + q7 (v_j x gvb) ; g_vbar[i] += q8 (vbar_j x gvb)
//   source side   g_v[j] += q2 gv + q6 sbar_i gvb + q7 (gvb x v_i)
//                 g_vbar[j] += ghb v_i + q3 (gv x v_i) + q4 sbar_i gv + q5 gvb + q8 (gvb x vbar_i)
// Pass A walks the dst-sorted view (receiver-side sums, plain stores); pass B walks the
// src-sorted view (g_phi, filter-weight partials, and source-side sums added onto pass A's
// output -- one block per node, same stream, so the read-modify-write is race free and the
// summation order is fixed).  The bead graph is tiny (60..4k edges): the point of fusing is
// launch count (1 + 3 launches instead of ~120 ATen kernels per layer), not bandwidth.
#include <stdlib.h>
#include <type_traits>
#include "cgv_common.h"
#include "equi_msg_dev.h"

namespace cgv {

struct v3 {
  float x, y, z;
};
__device__ __forceinline__ v3 ldv(const float* p) { f3 t = ld3(p); return v3{t.x, t.y, t.z}; }
__device__ __forceinline__ v3 cross(const v3& a, const v3& b) {
  return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float dot(const v3& a, const v3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ void axpy(v3& acc, float a, const v3& x) {
  acc.x = fmaf(a, x.x, acc.x); acc.y = fmaf(a, x.y, acc.y); acc.z = fmaf(a, x.z, acc.z);
}

// Edges whose source indices / gathers are issued together in pseudo_fwd_k / pseudo_bwd_recv_k: 8 when the launch has
// at most one block per CU (the 12-bead chignolin graph: one round trip for a bead's 5 edges), 2 otherwise -- the wider
// batch costs registers, and a 9-wave block that needs 98 of them runs alone on its CU (96-bead dipeptide graph,
// 960 blocks: 7.7 -> 12.6 us per call with 8).
constexpr int PSEUDO_EB_WIDE = 8, PSEUDO_EB_NARROW = 2;

template <int R>
__device__ __forceinline__ float filt(const float (&W)[R + 1], const float* __restrict__ g) {
  float w = W[R] * g[R];
#pragma unroll
  for (int n = 0; n < R; ++n) w = fmaf(W[n], g[n], w);
  return w;
}
template <int R>
__device__ __forceinline__ void load_row(float (&W)[R + 1], const float* __restrict__ Wd, const float* __restrict__ bd,
                                         int c) {
#pragma unroll
  for (int n = 0; n < R; ++n) W[n] = Wd[(size_t)c * R + n];
  W[R] = bd[c];
}

// Filter rows of the block's 64 channels through LDS: Wd[(k F + f0 + c) R + n] is contiguous over (c, n), so a
// row set is 64 R floats read with coalesced float4 loads (all in flight at once) instead of one 40-byte-
// strided load per (k, n) -- 99 such loads, 20 cache lines each, were ~1/3 of these kernels' time.
template <int R>
__device__ __forceinline__ bool filter_rows_stageable(const float* Wd, int F) {
  return ((F * R) & 3) == 0 && ((64 * R) & 3) == 0 && (reinterpret_cast<uintptr_t>(Wd) & 15) == 0;
}
template <int R, int NK>
__device__ __forceinline__ void stage_filter_rows(float* __restrict__ wt, const float* __restrict__ Wd, const int (&ks)[NK],
                                                  int F, int f0, int lane) {
  constexpr int NT = (64 * R / 4 + 63) / 64;
  const int n4 = min(64, F - f0) * R / 4;
  float4 tmp[NK][NT];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const float4* src = reinterpret_cast<const float4*>(Wd + ((size_t)ks[kk] * F + f0) * R);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int idx = lane + 64 * t;
      tmp[kk][t] = idx < n4 ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int kk = 0; kk < NK; ++kk)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int idx = lane + 64 * t;
      if (idx < 64 * R / 4) reinterpret_cast<float4*>(wt + kk * 64 * R)[idx] = tmp[kk][t];
    }
  __syncthreads();
}
template <int R>
__device__ __forceinline__ void read_row(float (&W)[R + 1], const float* __restrict__ wt_k, int cl, float bias) {
#pragma unroll
  for (int n = 0; n < R; ++n) W[n] = wt_k[cl * R + n];
  W[R] = bias;
}

// ------------------------------------------------------------------ forward: grid (N, ceil(F/64)), 9 waves per block
// Wave k owns filter k: q_k = phi[j, kF+f] w_k and the one output term that carries it (k = 0: T_h and the
// filter-free T_hbar; k = 1..4: the four terms of T_v; k = 5..8: those of T_vb).  The waves' vectors meet in LDS
// in wave order.  (One wave doing all nine filters: 99 weights per lane, 10.4 us; this: see DESIGN.md.)
// STAGE (dense bead graphs: 61 edges per node on the 2000-atom config): the receiver's edge records and source indices
// are copied to LDS in chunks of SEG_LDS edges by all nine waves (coalesced 16-byte loads) before they are walked -- per
// edge they were a dependent scalar load of 80 bytes (`s_waitcnt lgkmcnt(0)` before every use): 57 us per layer there.
constexpr int SEG_LDS = 128;
template <int R, int PSEUDO_EB, bool STAGE = false>
__global__ __launch_bounds__(576) void pseudo_fwd_k(const float* __restrict__ phi, const float* __restrict__ s,
                                                    const float* __restrict__ sbar, const float* __restrict__ v,
                                                    const float* __restrict__ vbar, const float* __restrict__ geom,
                                                    const int* __restrict__ rowptr, const int* __restrict__ src,
                                                    const float* __restrict__ Wd, const float* __restrict__ bd,
                                                    float* __restrict__ dh, float* __restrict__ dhbar,
                                                    float* __restrict__ dv, float* __restrict__ dvbar, int F,
                                                    int residual, float* __restrict__ dv_rows) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  __shared__ float red[8][3][64];
  __shared__ __attribute__((aligned(16))) float seg_geom[STAGE ? SEG_LDS * GS : 4];
  __shared__ int seg_src[STAGE ? SEG_LDS : 1];
  const int i = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int f_raw = blockIdx.y * 64 + lane;
  const bool live = f_raw < F;
  const int f = live ? f_raw : F - 1;
  float W[R + 1];
  load_row<R>(W, Wd, bd, k * F + f);
  const size_t nf = (size_t)i * F + f;
  const float s_i = s[nf], sb_i = sbar[nf];
  const v3 v_i = ldv(v + nf * 3), vb_i = ldv(vbar + nf * 3);
  float ah = 0.f, ahb = 0.f;
  v3 acc{0.f, 0.f, 0.f};
  // the bead graph's segments are a handful of edges (5 on the chignolin config) and every load of an edge hangs on
  // its source index: indices, then gathers, are issued for PSEUDO_EB edges at once (one round trip each instead of
  // one per edge pair); the terms are still added in edge order
  const float* __restrict__ vsrc = (k == 2 || k == 6 || k == 7) ? v : vbar;      // the one vector this wave's term reads
  const int e_beg = rowptr[i], e_end = rowptr[i + 1];
  for (int c_beg = e_beg; c_beg < e_end; c_beg += STAGE ? SEG_LDS : (e_end - e_beg)) {
    const int c_end = STAGE ? min(c_beg + SEG_LDS, e_end) : e_end;
    if (STAGE) {
      if (c_beg != e_beg) __syncthreads();                               // readers of the previous chunk
      const float4* gsrc = reinterpret_cast<const float4*>(geom + (size_t)c_beg * GS);
      for (int t = threadIdx.x; t < (c_end - c_beg) * (GS / 4); t += 576) reinterpret_cast<float4*>(seg_geom)[t] = gsrc[t];
      for (int t = threadIdx.x; t < c_end - c_beg; t += 576) seg_src[t] = src[c_beg + t];
      __syncthreads();
    }
    for (int eb = c_beg; eb < c_end; eb += PSEUDO_EB) {
      int jj[PSEUDO_EB];
      float ph[PSEUDO_EB];
      v3 vj[PSEUDO_EB];
#pragma unroll
      for (int u = 0; u < PSEUDO_EB; ++u) {
        const int e = min(eb + u, c_end - 1);
        jj[u] = STAGE ? seg_src[e - c_beg] : src[e];
      }
#pragma unroll
      for (int u = 0; u < PSEUDO_EB; ++u) {
        ph[u] = phi[(size_t)jj[u] * 9 * F + (size_t)k * F + f];
        vj[u] = ldv(vsrc + ((size_t)jj[u] * F + f) * 3);
      }
#pragma unroll
      for (int u = 0; u < PSEUDO_EB; ++u) {
        if (eb + u < c_end) {
          const float* __restrict__ g = STAGE ? seg_geom + (size_t)(eb + u - c_beg) * GS : geom + (size_t)(eb + u) * GS;
          const float q = ph[u] * filt<R>(W, g);
          switch (k) {                                   // wave-uniform
            case 0: ah = fmaf(q, s_i, ah); ahb += dot(v_i, vj[u]); break;
            case 1: axpy(acc, q, v3{g[U], g[U + 1], g[U + 2]}); break;
            case 2: axpy(acc, q, vj[u]); break;
            case 3: axpy(acc, q, cross(v_i, vj[u])); break;
            case 4: axpy(acc, q * sb_i, vj[u]); break;
            case 5: axpy(acc, q, vj[u]); break;
            case 6: axpy(acc, q * sb_i, vj[u]); break;
            case 7: axpy(acc, q, cross(v_i, vj[u])); break;
            default: axpy(acc, q, cross(vb_i, vj[u])); break;
          }
        }
      }
    }
  }
  if (k > 0) { red[k - 1][0][lane] = acc.x; red[k - 1][1][lane] = acc.y; red[k - 1][2][lane] = acc.z; }
  __syncthreads();
  if (k != 0 || !live) return;
  v3 av{0.f, 0.f, 0.f}, avb{0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 4; ++w) { av.x += red[w][0][lane]; av.y += red[w][1][lane]; av.z += red[w][2][lane]; }
#pragma unroll
  for (int w = 4; w < 8; ++w) { avb.x += red[w][0][lane]; avb.y += red[w][1][lane]; avb.z += red[w][2][lane]; }
  if (residual) {      // emit the updated state S + dS, ... directly (cgvae.py:108-111): the receiver's values are in registers
    ah += s_i; ahb += sb_i;
    av.x += v_i.x; av.y += v_i.y; av.z += v_i.z;
    avb.x += vb_i.x; avb.y += vb_i.y; avb.z += vb_i.z;
  }
  dh[nf] = ah;
  dhbar[nf] = ahb;
  st3(dv + nf * 3, av.x, av.y, av.z);
  st3(dvbar + nf * 3, avb.x, avb.y, avb.z);
  if (dv_rows) {       // the same vector as rows [3 i + xyz][f]: the layout the update block's channel mixing reads
    dv_rows[((size_t)3 * i + 0) * F + f] = av.x;
    dv_rows[((size_t)3 * i + 1) * F + f] = av.y;
    dv_rows[((size_t)3 * i + 2) * F + f] = av.z;
  }
}

// ------------------------------------------------------------------ dense bead graphs (>= 16 edges per node): forward
// The kernels above walk a node's edges with one code path for all nine waves: per edge a ladder of wave-uniform
// branches (the switch over k), 64-bit address arithmetic per gather and a wait right behind every request -- ~100
// instructions and two exposed round trips per edge and wave.  With 61 edges per node (64 beads, 2000-atom config) that
// walk is the whole kernel (54 us per layer against ~10 us of FMAs).  Here every wave runs a body compiled for ITS
// filter (template K), gathers go out as base + 32-bit offset four edges at a time with the next four already in flight
// (two register sets), and the staged segment is padded to whole batches (last source index repeated, zero records:
// w = 0, so a padded edge adds 0) so the walk has no tail code.  Same per-edge operations in the same order as
// pseudo_fwd_k: results are identical.
constexpr int PD_EB = 4;
// Gather bases are laundered through an empty asm (address space 1 kept in the type): loads through a `const __restrict__`
// kernel argument are free to sink below pd_pin() to their first use, which puts every round trip back in front of its FMAs.
#define CGV_PD_GLOBAL __attribute__((address_space(1)))
typedef const CGV_PD_GLOBAL char* pd_base;
__device__ __forceinline__ pd_base pd_launder(const float* p) {
  unsigned long long a = reinterpret_cast<unsigned long long>(p);
  asm volatile("" : "+s"(a));
  return reinterpret_cast<pd_base>(a);
}
__device__ __forceinline__ v3 ldv_at(pd_base base, unsigned byte_off) {
  const CGV_PD_GLOBAL float* q = reinterpret_cast<const CGV_PD_GLOBAL float*>(base + byte_off);
  return v3{q[0], q[1], q[2]};
}
__device__ __forceinline__ float ldf_at(pd_base base, unsigned byte_off) {
  return *reinterpret_cast<const CGV_PD_GLOBAL float*>(base + byte_off);
}
__device__ __forceinline__ void pd_pin() { asm volatile("" ::: "memory"); }

// Staged: the segment's node indices, padded to whole double batches with the last one repeated (a padded edge gathers
// valid rows and is skipped by a wave-uniform test).  The edge RECORDS are not staged: every lane reading the same LDS
// words costs the LDS pipe a full 64-lane access (3 x ds_read_b128 per edge and wave: 432 pipe cycles per edge of a
// 9-wave block, which was the pace of the first version of this kernel); addressed wave-uniformly in global memory they
// come through the scalar cache into SGPRs and cost the vector side nothing.
__device__ __forceinline__ int pd_stage_indices(int* __restrict__ seg_idx, const int* __restrict__ idx, int c_beg, int n, int threads) {
  const int n_pad = (n + 2 * PD_EB - 1) / (2 * PD_EB) * (2 * PD_EB);
  for (int t = threadIdx.x; t < n_pad; t += threads) seg_idx[t] = idx[c_beg + min(t, n - 1)];
  return n_pad;
}

// The filter value as two interleaved partial sums (even / odd n) in one register pair: R / 2 packed FMAs + 2 instead of
// R + 1 (the walk is VALU-issue bound once its loads are out of the way).  Differs from filt() by summation order only.
typedef float pd_f2 __attribute__((ext_vector_type(2)));
template <int R>
__device__ __forceinline__ float filt_pk(const float (&W)[R + 1], const float* __restrict__ g) {
  static_assert((R & 1) == 0, "pairs");
  pd_f2 w2 = pd_f2{W[R] * g[R], 0.f};
#pragma unroll
  for (int n = 0; n < R; n += 2) w2 = __builtin_elementwise_fma(pd_f2{W[n], W[n + 1]}, pd_f2{g[n], g[n + 1]}, w2);
  return w2.x + w2.y;
}

struct PdFwdBatch {
  float ph[PD_EB];
  v3 vj[PD_EB];
};

// Gathers as buffer loads: address = descriptor base + lane byte offset (VGPR, loop invariant) + row byte offset (SGPR, one
// s_mul per gather) -- the node index comes from a wave-uniform scalar load of the index array, so a gather costs the
// vector unit nothing but the load itself (the walk is bound by VALU issue: ~15 -> ~11 instructions per edge and wave).
// The indices of a batch are requested one batch ahead (the scalar unit's loads return out of order: every wait is for
// all of them, so nothing may be requested right in front of its use).
__device__ __forceinline__ float pd_ldf_buf(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ v3 pd_ldv_buf(rsrc_t r, unsigned voff, unsigned soff) {
  const f3v x = __builtin_bit_cast(f3v, __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, 0));
  return v3{x.x, x.y, x.z};
}
struct PdIdx { int j[PD_EB]; };
__device__ __forceinline__ PdIdx pd_indices(const int* __restrict__ idx_c, int e0, int n) {   // wave-uniform: scalar loads
  PdIdx r;
#pragma unroll
  for (int u = 0; u < PD_EB; ++u) r.j[u] = idx_c[min(e0 + u, n - 1)];
  return r;
}

template <int R, int K>
__device__ __forceinline__ void pseudo_fwd_dense_walk(const float* __restrict__ phi, const float* __restrict__ v,
                                                      const float* __restrict__ vbar, const float* __restrict__ geom_c,
                                                      const int* __restrict__ src_c, int n, int F, int f,
                                                      const float (&W)[R + 1], float& ah, v3& acc) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  // Every term is linear in the receiver's own values (s_i q, v_i . vbar_j, q v_i x vbar_j, q sbar_i vbar_j, ...): the walk
  // only sums q and q * (the gathered vector); the receiver's factor -- a product, a dot or ONE cross product -- is applied
  // by the kernel when the segment is done (6 instructions per edge less in the three cross-product bodies, and the
  // receiver's values are not live across the walk).  ah: sum of q; acc: sum of q x_j (k = 0: sum of vbar_j for hbar).
  const rsrc_t r_vec = make_rsrc((K == 2 || K == 6 || K == 7) ? v : vbar), r_phi = make_rsrc(phi);
  const unsigned phi_c = (unsigned)(K * F + f) * 4u, phi_s = 9u * (unsigned)F * 4u;
  const unsigned vec_c = (unsigned)f * 12u, vec_s = (unsigned)F * 12u;
  auto issue = [&](PdFwdBatch& b, const PdIdx& ix) {
#pragma unroll
    for (int u = 0; u < PD_EB; ++u) {
      const unsigned j = (unsigned)ix.j[u];
      b.ph[u] = pd_ldf_buf(r_phi, phi_c, j * phi_s);
      if (K != 1) b.vj[u] = pd_ldv_buf(r_vec, vec_c, j * vec_s);
    }
  };
  auto compute = [&](const PdFwdBatch& b, int e0, auto tail) {
    const float* __restrict__ gb = geom_c + (size_t)e0 * GS;
#pragma unroll
    for (int u = 0; u < PD_EB; ++u) {
      if (decltype(tail)::value && e0 + u >= n) break;                 // wave-uniform (the last trip only: a test per edge
                                                                       // keeps each record load behind its branch)
      const float* __restrict__ g = gb + u * GS;                      // wave-uniform address: scalar loads, constant offsets
      const float q = b.ph[u] * filt_pk<R>(W, g);
      if (K == 0) { ah += q; acc.x += b.vj[u].x; acc.y += b.vj[u].y; acc.z += b.vj[u].z; }
      else if (K == 1) axpy(acc, q, v3{g[U], g[U + 1], g[U + 2]});
      else axpy(acc, q, b.vj[u]);
    }
  };
  PdFwdBatch b0, b1;
  PdIdx ix = pd_indices(src_c, 0, n);
  issue(b0, ix);
  ix = pd_indices(src_c, PD_EB, n);
  int e0 = 0;
  for (; e0 + 2 * PD_EB <= n; e0 += 2 * PD_EB) {
    issue(b1, ix);
    ix = pd_indices(src_c, e0 + 2 * PD_EB, n);
    pd_pin();
    compute(b0, e0, std::false_type{});
    issue(b0, ix);
    ix = pd_indices(src_c, e0 + 3 * PD_EB, n);
    pd_pin();
    compute(b1, e0 + PD_EB, std::false_type{});
  }
  if (e0 < n) {
    issue(b1, ix);
    pd_pin();
    compute(b0, e0, std::true_type{});
    compute(b1, e0 + PD_EB, std::true_type{});
  }
}

__device__ __forceinline__ int pd_filter_of_wave(int wave) {
  //                     wave: 0  1  2  3  4  5  6  7  8
  constexpr unsigned table = (1u << 0) | (3u << 4) | (7u << 8) | (8u << 12) | (2u << 16) | (0u << 20) | (4u << 24) | (6u << 28);
  return wave == 8 ? 5 : (int)((table >> (4 * wave)) & 15u);
}

template <int R>
__global__ __launch_bounds__(576) void pseudo_fwd_dense_k(const float* __restrict__ phi, const float* __restrict__ s,
                                                          const float* __restrict__ sbar, const float* __restrict__ v,
                                                          const float* __restrict__ vbar, const float* __restrict__ geom,
                                                          const int* __restrict__ rowptr, const int* __restrict__ src,
                                                          const float* __restrict__ Wd, const float* __restrict__ bd,
                                                          float* __restrict__ dh, float* __restrict__ dhbar,
                                                          float* __restrict__ dv, float* __restrict__ dvbar, int F,
                                                          int residual, float* __restrict__ dv_rows) {
  constexpr int GS = geom_stride(R);
  __shared__ float red[8][3][64];
  const int i = blockIdx.x;
  const int lane = threadIdx.x & 63;
  // Nine waves on four SIMDs (wave w on SIMD w % 4): the SIMD that gets three is given the three cheapest terms
  // (k = 1, 2, 5: one axpy), the others a cross-product term and a scaled one each -- per-edge instruction totals
  // 42 / 35 / 35 / 35 instead of 50 / 28 / 29 / 40 with wave w = filter w.
  const int k = pd_filter_of_wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
  const int f_raw = blockIdx.y * 64 + lane;
  const bool live = f_raw < F;
  const int f = live ? f_raw : F - 1;
  float W[R + 1];
  load_row<R>(W, Wd, bd, k * F + f);
  const size_t nf = (size_t)i * F + f;
  float ah = 0.f, ahb = 0.f;
  v3 acc{0.f, 0.f, 0.f};
  const int e_beg = rowptr[i], e_end = rowptr[i + 1];
  if (e_end > e_beg) {
    const int n = e_end - e_beg;
    const float* __restrict__ geom_c = geom + (size_t)e_beg * GS;
    const int* __restrict__ src_c = src + e_beg;
    switch (k) {                                                         // wave-uniform: each wave runs the body of its filter
#define CGV_PD_FWD(KV) case KV: pseudo_fwd_dense_walk<R, KV>(phi, v, vbar, geom_c, src_c, n, F, f, W, ah, acc); break
      CGV_PD_FWD(0); CGV_PD_FWD(1); CGV_PD_FWD(2); CGV_PD_FWD(3); CGV_PD_FWD(4);
      CGV_PD_FWD(5); CGV_PD_FWD(6); CGV_PD_FWD(7);
      default: pseudo_fwd_dense_walk<R, 8>(phi, v, vbar, geom_c, src_c, n, F, f, W, ah, acc); break;
#undef CGV_PD_FWD
    }
  }
  {                                                                      // the receiver's factor of this wave's term
    if (k == 0) { ahb = dot(ldv(v + nf * 3), acc); ah *= s[nf]; }
    else if (k == 3 || k == 7) acc = cross(ldv(v + nf * 3), acc);
    else if (k == 8) acc = cross(ldv(vbar + nf * 3), acc);
    else if (k == 4 || k == 6) { const float sb_i = sbar[nf]; acc.x *= sb_i; acc.y *= sb_i; acc.z *= sb_i; }
  }
  if (k > 0) { red[k - 1][0][lane] = acc.x; red[k - 1][1][lane] = acc.y; red[k - 1][2][lane] = acc.z; }
  __syncthreads();
  if (k != 0 || !live) return;
  v3 av{0.f, 0.f, 0.f}, avb{0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 4; ++w) { av.x += red[w][0][lane]; av.y += red[w][1][lane]; av.z += red[w][2][lane]; }
#pragma unroll
  for (int w = 4; w < 8; ++w) { avb.x += red[w][0][lane]; avb.y += red[w][1][lane]; avb.z += red[w][2][lane]; }
  if (residual) {
    const v3 v_i = ldv(v + nf * 3), vb_i = ldv(vbar + nf * 3);
    ah += s[nf]; ahb += sbar[nf];
    av.x += v_i.x; av.y += v_i.y; av.z += v_i.z;
    avb.x += vb_i.x; avb.y += vb_i.y; avb.z += vb_i.z;
  }
  dh[nf] = ah;
  dhbar[nf] = ahb;
  st3(dv + nf * 3, av.x, av.y, av.z);
  st3(dvbar + nf * 3, avb.x, avb.y, avb.z);
  if (dv_rows) {
    dv_rows[((size_t)3 * i + 0) * F + f] = av.x;
    dv_rows[((size_t)3 * i + 1) * F + f] = av.y;
    dv_rows[((size_t)3 * i + 2) * F + f] = av.z;
  }
}

// ------------------------------------------------------------------ backward pass A (receiver side)
template <int R, int PSEUDO_EB>
__global__ __launch_bounds__(64) void pseudo_bwd_recv_k(const float* __restrict__ phi, const float* __restrict__ v,
                                                        const float* __restrict__ vbar, const float* __restrict__ geom,
                                                        const int* __restrict__ rowptr, const int* __restrict__ src,
                                                        const float* __restrict__ Wd, const float* __restrict__ bd,
                                                        const float* __restrict__ gh, const float* __restrict__ ghb,
                                                        const float* __restrict__ gv, const float* __restrict__ gvb,
                                                        float* __restrict__ g_s, float* __restrict__ g_sbar,
                                                        float* __restrict__ g_v, float* __restrict__ g_vbar, int F,
                                                        int residual) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  const int i = blockIdx.x;
  const int f_raw = blockIdx.y * 64 + threadIdx.x;
  const bool live = f_raw < F;
  const int f = live ? f_raw : F - 1;
  // filters needed on the receiver side: k = 0, 3, 4, 6, 7, 8
  float W0[R + 1], W3[R + 1], W4[R + 1], W6[R + 1], W7[R + 1], W8[R + 1];
  __shared__ __attribute__((aligned(16))) float wt[6 * 64 * R];
  if (filter_rows_stageable<R>(Wd, F)) {
    const int ks[6] = {0, 3, 4, 6, 7, 8};
    stage_filter_rows<R, 6>(wt, Wd, ks, F, blockIdx.y * 64, threadIdx.x);
    const int cl = f - blockIdx.y * 64;
    read_row<R>(W0, wt + 0 * 64 * R, cl, bd[0 * F + f]);
    read_row<R>(W3, wt + 1 * 64 * R, cl, bd[3 * F + f]);
    read_row<R>(W4, wt + 2 * 64 * R, cl, bd[4 * F + f]);
    read_row<R>(W6, wt + 3 * 64 * R, cl, bd[6 * F + f]);
    read_row<R>(W7, wt + 4 * 64 * R, cl, bd[7 * F + f]);
    read_row<R>(W8, wt + 5 * 64 * R, cl, bd[8 * F + f]);
  } else {
    load_row<R>(W0, Wd, bd, 0 * F + f);
    load_row<R>(W3, Wd, bd, 3 * F + f);
    load_row<R>(W4, Wd, bd, 4 * F + f);
    load_row<R>(W6, Wd, bd, 6 * F + f);
    load_row<R>(W7, Wd, bd, 7 * F + f);
    load_row<R>(W8, Wd, bd, 8 * F + f);
  }
  const size_t nf = (size_t)i * F + f;
  const float gh_i = gh ? gh[nf] : 0.f, ghb_i = ghb ? ghb[nf] : 0.f;
  const v3 gv_i = gv ? ldv(gv + nf * 3) : v3{0.f, 0.f, 0.f};
  const v3 gvb_i = gvb ? ldv(gvb + nf * 3) : v3{0.f, 0.f, 0.f};
  float as = 0.f, asb = 0.f;
  v3 av{0.f, 0.f, 0.f}, avb{0.f, 0.f, 0.f};
  const int e_beg = rowptr[i], e_end = rowptr[i + 1];
  constexpr int EB = PSEUDO_EB > 2 ? PSEUDO_EB / 2 : 2;          // 14 gathered values per edge here
  for (int eb = e_beg; eb < e_end; eb += EB) {       // indices, then gathers, for EB edges at once (see pseudo_fwd_k)
    int jj[EB];
    float p0[EB], p3[EB], p4[EB], p6[EB], p7[EB], p8[EB];
    v3 vjs[EB], vbs[EB];
#pragma unroll
    for (int u = 0; u < EB; ++u) jj[u] = src[min(eb + u, e_end - 1)];
#pragma unroll
    for (int u = 0; u < EB; ++u) {
      const float* __restrict__ pr = phi + (size_t)jj[u] * 9 * F + f;
      p0[u] = pr[0]; p3[u] = pr[(size_t)3 * F]; p4[u] = pr[(size_t)4 * F];
      p6[u] = pr[(size_t)6 * F]; p7[u] = pr[(size_t)7 * F]; p8[u] = pr[(size_t)8 * F];
      vjs[u] = ldv(v + ((size_t)jj[u] * F + f) * 3);
      vbs[u] = ldv(vbar + ((size_t)jj[u] * F + f) * 3);
    }
#pragma unroll
    for (int u = 0; u < EB; ++u) {
      if (eb + u < e_end) {
        const float* __restrict__ g = geom + (size_t)(eb + u) * GS;
        const v3 v_j = vjs[u], vb_j = vbs[u];
        const float q0 = p0[u] * filt<R>(W0, g);
        const float q3 = p3[u] * filt<R>(W3, g);
        const float q4 = p4[u] * filt<R>(W4, g);
        const float q6 = p6[u] * filt<R>(W6, g);
        const float q7 = p7[u] * filt<R>(W7, g);
        const float q8 = p8[u] * filt<R>(W8, g);
        as = fmaf(gh_i, q0, as);
        asb = fmaf(q4, dot(gv_i, vb_j), fmaf(q6, dot(gvb_i, v_j), asb));
        axpy(av, ghb_i, vb_j);
        axpy(av, q3, cross(vb_j, gv_i));
        axpy(av, q7, cross(v_j, gvb_i));
        axpy(avb, q8, cross(vb_j, gvb_i));
      }
    }
  }
  if (residual) {      // outputs were state + delta: the upstream gradient also flows straight through
    as += gh_i; asb += ghb_i;
    av.x += gv_i.x; av.y += gv_i.y; av.z += gv_i.z;
    avb.x += gvb_i.x; avb.y += gvb_i.y; avb.z += gvb_i.z;
  }
  if (live) {
    g_s[nf] = as;
    g_sbar[nf] = asb;
    st3(g_v + nf * 3, av.x, av.y, av.z);
    st3(g_vbar + nf * 3, avb.x, avb.y, avb.z);
  }
}

// ------------------------------------------------------------------ backward pass B (source side + filter grads)
// grid (n_chunks, ceil(F/64)), 9 waves per block: wave k owns filter k -- its weight row, its g_phi slot, its
// (R+1) filter-gradient accumulators and the terms of the source-side vector gradients that carry q_k.  The
// one-wave version kept 99 accumulators + 99 weights per lane and walked the edges alone (14 us per call on
// the 60-edge bead graph, the longest kernel of a decoder layer's backward); here a lane holds 2(R+1) values
// and nine waves hide each other's gather latency.  The waves' partial vector sums meet in LDS in wave order.
// STAGE: as in pseudo_fwd_k -- the node's edge records and receiver indices go to LDS in chunks of SEG_LDS edges (the
// per-edge record was a dependent 80-byte scalar load in front of every edge of every wave).
template <int R, int SRC_EB, bool STAGE = false>
__global__ __launch_bounds__(576) void pseudo_bwd_src_k(
    const float* __restrict__ phi, const float* __restrict__ s, const float* __restrict__ sbar,
    const float* __restrict__ v, const float* __restrict__ vbar, const float* __restrict__ geom,
    const int* __restrict__ rowptr, const int* __restrict__ dst, const float* __restrict__ Wd,
    const float* __restrict__ bd, const float* __restrict__ gh, const float* __restrict__ ghb,
    const float* __restrict__ gv, const float* __restrict__ gvb, float* __restrict__ g_phi,
    float* __restrict__ g_v, float* __restrict__ g_vbar, float* __restrict__ part, int F, int N, int nodes_per_chunk) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  __shared__ float red[8][6][64];                    // waves 1..8: (av, avb) partials of the current node
  __shared__ __attribute__((aligned(16))) float seg_geom[STAGE ? SEG_LDS * GS : 4];
  __shared__ int seg_dst[STAGE ? SEG_LDS : 1];
  const int lane = threadIdx.x & 63;
  const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // this wave's filter
  const int f_raw = blockIdx.y * 64 + lane;
  const bool live = f_raw < F;
  const int f = live ? f_raw : F - 1;
  float W[R + 1], G[R + 1];
  load_row<R>(W, Wd, bd, k * F + f);
#pragma unroll
  for (int n = 0; n <= R; ++n) G[n] = 0.f;

  const int n_beg = blockIdx.x * nodes_per_chunk, n_end = min(n_beg + nodes_per_chunk, N);
  for (int j = n_beg; j < n_end; ++j) {
    const float p = phi[(size_t)j * 9 * F + (size_t)k * F + f];
    const size_t jf = (size_t)j * F + f;
    const v3 v_j = ldv(v + jf * 3), vb_j = ldv(vbar + jf * 3);
    float a = 0.f;
    v3 av{0.f, 0.f, 0.f}, avb{0.f, 0.f, 0.f};
    // The receiver-side operands of an edge hang on its receiver index: indices, then gathers, are issued for SRC_EB
    // edges at once (the 64-bead graph of the 2000-atom config has 61 edges per node: one dependent round trip per edge
    // made this the longest kernel of a decoder layer there); the terms are still added in edge order.
    const float* __restrict__ vecA = (k >= 1 && k <= 4) ? gv : (k >= 5 ? gvb : nullptr);          // upstream vector of this filter
    const float* __restrict__ vecB = (k == 3 || k == 7 || k == 0) ? v : (k == 8 ? vbar : nullptr);  // receiver state vector
    const float* __restrict__ scaA = (k == 0) ? gh : ((k == 4 || k == 6) ? sbar : nullptr);
    const int e_beg = rowptr[j], e_end = rowptr[j + 1];
    for (int c_beg = e_beg; c_beg < e_end; c_beg += STAGE ? SEG_LDS : (e_end - e_beg)) {
    const int c_end = STAGE ? min(c_beg + SEG_LDS, e_end) : e_end;
    if (STAGE) {
      if (c_beg != e_beg) __syncthreads();                               // readers of the previous chunk (the previous node's
                                                                         // are behind the barrier that closes its turn)
      const float4* gsrc = reinterpret_cast<const float4*>(geom + (size_t)c_beg * GS);
      for (int t = threadIdx.x; t < (c_end - c_beg) * (GS / 4); t += 576) reinterpret_cast<float4*>(seg_geom)[t] = gsrc[t];
      for (int t = threadIdx.x; t < c_end - c_beg; t += 576) seg_dst[t] = dst[c_beg + t];
      __syncthreads();
    }
    for (int eb = c_beg; eb < c_end; eb += SRC_EB) {
      int ii[SRC_EB];
      v3 A[SRC_EB], B[SRC_EB];
      float sA[SRC_EB], sB[SRC_EB], hb[SRC_EB];
#pragma unroll
      for (int u = 0; u < SRC_EB; ++u) ii[u] = STAGE ? seg_dst[min(eb + u, c_end - 1) - c_beg] : dst[min(eb + u, e_end - 1)];
#pragma unroll
      for (int u = 0; u < SRC_EB; ++u) {
        const size_t nf = (size_t)ii[u] * F + f;
        const v3 zero{0.f, 0.f, 0.f};
        A[u] = vecA ? ldv(vecA + nf * 3) : zero;
        B[u] = vecB ? ldv(vecB + nf * 3) : zero;
        sA[u] = scaA ? scaA[nf] : 0.f;
        sB[u] = (k == 0) ? s[nf] : 0.f;
        hb[u] = (k == 0 && ghb) ? ghb[nf] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < SRC_EB; ++u) {
        if (eb + u < c_end) {
          const float* __restrict__ g = STAGE ? seg_geom + (size_t)(eb + u - c_beg) * GS : geom + (size_t)(eb + u) * GS;
          const v3 zero{0.f, 0.f, 0.f};
          float gq = 0.f;
          v3 cav = zero, cavb = zero;                    // source-side vectors that get multiplied by q_k
          switch (k) {                                   // wave-uniform
            case 0: gq = sA[u] * sB[u]; break;                                             // gh_i s_i
            case 1: gq = dot(A[u], v3{g[U], g[U + 1], g[U + 2]}); break;
            case 2: gq = dot(A[u], v_j); cav = A[u]; break;
            case 3: gq = dot(A[u], cross(B[u], vb_j)); cavb = cross(A[u], B[u]); break;
            case 4: gq = sA[u] * dot(A[u], vb_j); cavb = v3{sA[u] * A[u].x, sA[u] * A[u].y, sA[u] * A[u].z}; break;
            case 5: gq = dot(A[u], vb_j); cavb = A[u]; break;
            case 6: gq = sA[u] * dot(A[u], v_j); cav = v3{sA[u] * A[u].x, sA[u] * A[u].y, sA[u] * A[u].z}; break;
            case 7: gq = dot(A[u], cross(B[u], v_j)); cav = cross(A[u], B[u]); break;
            default: gq = dot(A[u], cross(B[u], vb_j)); cavb = cross(A[u], B[u]); break;
          }
          const float w = filt<R>(W, g);
          a = fmaf(gq, w, a);
          const float t = gq * p;
#pragma unroll
          for (int n = 0; n <= R; ++n) G[n] = fmaf(t, g[n], G[n]);
          const float q = p * w;
          axpy(av, q, cav);
          axpy(avb, q, cavb);
          if (k == 0 && ghb) axpy(avb, hb[u], B[u]);                                      // the filter-free term ghb_i v_i
        }
      }
    }
    }
    if (k > 0) {
      float* r = &red[k - 1][0][lane];
      r[0] = av.x; r[64] = av.y; r[128] = av.z; r[192] = avb.x; r[256] = avb.y; r[320] = avb.z;
    }
    __syncthreads();
    if (live) g_phi[(size_t)j * 9 * F + (size_t)k * F + f] = a;
    if (k == 0) {
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) {
        const float* r = &red[w8][0][lane];
        av.x += r[0]; av.y += r[64]; av.z += r[128]; avb.x += r[192]; avb.y += r[256]; avb.z += r[320];
      }
      if (live) {
        const v3 r0 = ldv(g_v + jf * 3), r1 = ldv(g_vbar + jf * 3);        // pass A's receiver-side part
        st3(g_v + jf * 3, r0.x + av.x, r0.y + av.y, r0.z + av.z);
        st3(g_vbar + jf * 3, r1.x + avb.x, r1.y + avb.y, r1.z + avb.z);
      }
    }
    __syncthreads();                                 // red is reused by the next node
  }
  if (live) {
    float* __restrict__ out = part + (size_t)blockIdx.x * 9 * (R + 1) * F;
#pragma unroll
    for (int n = 0; n <= R; ++n) out[(size_t)(k * (R + 1) + n) * F + f] = G[n];
  }
}

// ------------------------------------------------------------------ dense bead graphs: backward
// Same recipe as pseudo_fwd_dense_k (a body per filter, pipelined 32-bit-offset gathers, records through the scalar cache,
// packed filter sums).  Both passes require all four upstream gradients (the launcher falls back otherwise).
//
// Pass A (receiver side): the general kernel gives a receiver's 64 channels to ONE wave that evaluates six filters per edge
// (66 weights per lane; 640 one-wave blocks on the 64-bead graph = 0.6 waves per SIMD).  Here six waves take one filter
// each (k = 0, 3, 4, 6, 7, 8) and their partial sums meet in LDS in filter order.
struct PdRecvBatch {
  float ph[PD_EB];
  v3 vj[PD_EB];
};
template <int R, int K>
__device__ __forceinline__ void pseudo_bwd_recv_dense_walk(const float* __restrict__ phi, const float* __restrict__ v,
                                                           const float* __restrict__ vbar, const float* __restrict__ geom_c,
                                                           const int* __restrict__ seg_src, int n, int n_pad, int F, int f,
                                                           const float (&W)[R + 1], float& sc, v3& vec) {
  constexpr int GS = geom_stride(R);
  const pd_base vsrc = pd_launder((K == 6 || K == 7) ? v : vbar), phib = pd_launder(phi);
  const unsigned phi_c = (unsigned)(K * F + f) * 4u, phi_s = 9u * (unsigned)F * 4u;
  const unsigned vec_c = (unsigned)f * 12u, vec_s = (unsigned)F * 12u;
  auto issue = [&](PdRecvBatch& b, int e0) {
    const int4 j4 = *reinterpret_cast<const int4*>(seg_src + e0);
    const int js[4] = {j4.x, j4.y, j4.z, j4.w};
#pragma unroll
    for (int u = 0; u < PD_EB; ++u) {
      unsigned j = (unsigned)js[u];
      asm volatile("" : "+v"(j));
      b.ph[u] = ldf_at(phib, __umul24(j, phi_s) + phi_c);
      b.vj[u] = ldv_at(vsrc, __umul24(j, vec_s) + vec_c);
    }
  };
  auto compute = [&](const PdRecvBatch& b, int e0, auto tail) {
    const float* __restrict__ gb = geom_c + (size_t)e0 * GS;
#pragma unroll
    for (int u = 0; u < PD_EB; ++u) {
      if (decltype(tail)::value && e0 + u >= n) break;
      const float q = b.ph[u] * filt_pk<R>(W, gb + u * GS);
      // linear in the receiver's upstream gradients: the walk sums q (k = 0, with the plain sum of vbar_j) or q x_j; the
      // kernel applies gh_i / ghb_i, the dot or the cross product with gv_i / gvb_i once, behind the segment
      if (K == 0) { sc += q; vec.x += b.vj[u].x; vec.y += b.vj[u].y; vec.z += b.vj[u].z; }
      else axpy(vec, q, b.vj[u]);
    }
  };
  PdRecvBatch b0, b1;
  issue(b0, 0);
  int e0 = 0;
  for (; e0 + 2 * PD_EB <= n; e0 += 2 * PD_EB) {
    issue(b1, e0 + PD_EB);
    pd_pin();
    compute(b0, e0, std::false_type{});
    issue(b0, min(e0 + 2 * PD_EB, n_pad - PD_EB));
    pd_pin();
    compute(b1, e0 + PD_EB, std::false_type{});
  }
  if (e0 < n) {
    issue(b1, e0 + PD_EB);
    pd_pin();
    compute(b0, e0, std::true_type{});
    compute(b1, e0 + PD_EB, std::true_type{});
  }
}

template <int R>
__global__ __launch_bounds__(384) void pseudo_bwd_recv_dense_k(const float* __restrict__ phi, const float* __restrict__ v,
                                                               const float* __restrict__ vbar, const float* __restrict__ geom,
                                                               const int* __restrict__ rowptr, const int* __restrict__ src,
                                                               const float* __restrict__ Wd, const float* __restrict__ bd,
                                                               const float* __restrict__ gh, const float* __restrict__ ghb,
                                                               const float* __restrict__ gv, const float* __restrict__ gvb,
                                                               float* __restrict__ g_s, float* __restrict__ g_sbar,
                                                               float* __restrict__ g_v, float* __restrict__ g_vbar, int F,
                                                               int residual) {
  constexpr int GS = geom_stride(R);
  __shared__ float red[6][4][64];                       // per wave: scalar sum, vector sum
  __shared__ __attribute__((aligned(16))) int seg_src[SEG_LDS];
  const int i = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // waves 0..5 -> filters 0, 3, 4, 6, 7, 8
  const int k = wave == 0 ? 0 : (wave == 1 ? 3 : (wave == 2 ? 4 : wave + 3));
  const int f_raw = blockIdx.y * 64 + lane;
  const bool live = f_raw < F;
  const int f = live ? f_raw : F - 1;
  float W[R + 1];
  load_row<R>(W, Wd, bd, k * F + f);
  const size_t nf = (size_t)i * F + f;
  float sc = 0.f;
  v3 vec{0.f, 0.f, 0.f};
  const int e_beg = rowptr[i], e_end = rowptr[i + 1];
  for (int c_beg = e_beg; c_beg < e_end; c_beg += SEG_LDS) {
    const int n = min(SEG_LDS, e_end - c_beg);
    if (c_beg != e_beg) __syncthreads();
    const int n_pad = pd_stage_indices(seg_src, src, c_beg, n, 384);
    const float* __restrict__ geom_c = geom + (size_t)c_beg * GS;
    __syncthreads();
    switch (k) {
#define CGV_PD_RECV(KV) case KV: pseudo_bwd_recv_dense_walk<R, KV>(phi, v, vbar, geom_c, seg_src, n, n_pad, F, f, W, sc, vec); break
      CGV_PD_RECV(0); CGV_PD_RECV(3); CGV_PD_RECV(4); CGV_PD_RECV(6); CGV_PD_RECV(7);
      default: pseudo_bwd_recv_dense_walk<R, 8>(phi, v, vbar, geom_c, seg_src, n, n_pad, F, f, W, sc, vec); break;
#undef CGV_PD_RECV
    }
  }
  const float gh_i = gh[nf], ghb_i = ghb[nf];
  const v3 gv_i = ldv(gv + nf * 3), gvb_i = ldv(gvb + nf * 3);
  switch (k) {                                                           // the receiver's factor of this wave's term
    case 0: sc *= gh_i; vec.x *= ghb_i; vec.y *= ghb_i; vec.z *= ghb_i; break;          // g_s ; the filter-free ghb_i vbar_j of g_v
    case 3: vec = cross(vec, gv_i); break;                                              // g_v
    case 4: sc = dot(gv_i, vec); break;                                                 // g_sbar
    case 6: sc = dot(gvb_i, vec); break;                                                // g_sbar
    default: vec = cross(vec, gvb_i); break;                                            // g_v (k = 7), g_vbar (k = 8)
  }
  red[wave][0][lane] = sc; red[wave][1][lane] = vec.x; red[wave][2][lane] = vec.y; red[wave][3][lane] = vec.z;
  __syncthreads();
  if (wave != 0 || !live) return;
  // wave order: 0 (k 0: g_s, ghb part of g_v), 1 (k 3: g_v), 2 (k 4: g_sbar), 3 (k 6: g_sbar), 4 (k 7: g_v), 5 (k 8: g_vbar)
  float as = sc, asb = red[2][0][lane] + red[3][0][lane];
  v3 av{vec.x + red[1][1][lane] + red[4][1][lane], vec.y + red[1][2][lane] + red[4][2][lane], vec.z + red[1][3][lane] + red[4][3][lane]};
  v3 avb{red[5][1][lane], red[5][2][lane], red[5][3][lane]};
  if (residual) {
    as += gh_i; asb += ghb_i;
    av.x += gv_i.x; av.y += gv_i.y; av.z += gv_i.z;
    avb.x += gvb_i.x; avb.y += gvb_i.y; avb.z += gvb_i.z;
  }
  g_s[nf] = as;
  g_sbar[nf] = asb;
  st3(g_v + nf * 3, av.x, av.y, av.z);
  st3(g_vbar + nf * 3, avb.x, avb.y, avb.z);
}

// Pass B (source side + filter gradients): as pseudo_bwd_src_k, one body per filter.  The (R + 1) filter-gradient
// accumulators of a lane are kept as pairs and take packed FMAs; the gathers of a wave are exactly the receiver-side
// operands its filter's terms read (3 - 7 values per edge).
constexpr int PD_SRC_EB = 2;          // (four edges' operands in flight: with PD_EB = 4 the cross-product bodies spill)
struct PdSrcBatch {
  v3 A[PD_SRC_EB], B[PD_SRC_EB];
  float sA[PD_SRC_EB], sB[PD_SRC_EB], hb[PD_SRC_EB];
};
template <int R, int K>
__device__ __forceinline__ void pseudo_bwd_src_dense_walk(const float* __restrict__ s, const float* __restrict__ sbar,
                                                          const float* __restrict__ v, const float* __restrict__ vbar,
                                                          const float* __restrict__ gh, const float* __restrict__ ghb,
                                                          const float* __restrict__ gv, const float* __restrict__ gvb,
                                                          const float* __restrict__ geom_c, const int* __restrict__ seg_dst,
                                                          int n, int n_pad, int F, int f, const float (&W)[R + 1], float p,
                                                          const v3& v_j, const v3& vb_j, float& a, pd_f2 (&G2)[R / 2], float& GR,
                                                          v3& av, v3& avb) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  const pd_base vecA = pd_launder(K <= 4 ? gv : gvb);                                 // K >= 1
  const pd_base vecB = pd_launder(K == 8 ? vbar : v);                                 // K = 0, 3, 7, 8
  const pd_base scaA = pd_launder(K == 0 ? gh : sbar);                                // K = 0, 4, 6
  const pd_base sBb = pd_launder(s), hbb = pd_launder(ghb);                           // K = 0
  const unsigned sc_c = (unsigned)f * 4u, sc_s = (unsigned)F * 4u;
  const unsigned vec_c = (unsigned)f * 12u, vec_s = (unsigned)F * 12u;
  constexpr bool hasA = K >= 1, hasB = K == 0 || K == 3 || K == 7 || K == 8, hasS = K == 0 || K == 4 || K == 6;
  auto issue = [&](PdSrcBatch& b, int e0) {
    const int2 i2 = *reinterpret_cast<const int2*>(seg_dst + e0);
    const int is[2] = {i2.x, i2.y};
#pragma unroll
    for (int u = 0; u < PD_SRC_EB; ++u) {
      unsigned ii = (unsigned)is[u];
      asm volatile("" : "+v"(ii));
      const unsigned vo = __umul24(ii, vec_s) + vec_c, so = __umul24(ii, sc_s) + sc_c;
      if (hasA) b.A[u] = ldv_at(vecA, vo);
      if (hasB) b.B[u] = ldv_at(vecB, vo);
      if (hasS) b.sA[u] = ldf_at(scaA, so);
      if (K == 0) { b.sB[u] = ldf_at(sBb, so); b.hb[u] = ldf_at(hbb, so); }
    }
  };
  auto compute = [&](const PdSrcBatch& b, int e0, auto tail) {
    const float* __restrict__ gb = geom_c + (size_t)e0 * GS;
#pragma unroll
    for (int u = 0; u < PD_SRC_EB; ++u) {
      if (decltype(tail)::value && e0 + u >= n) break;
      const float* __restrict__ g = gb + u * GS;
      const v3 zero{0.f, 0.f, 0.f};
      float gq;
      v3 cav = zero, cavb = zero;
      switch (K) {
        case 0: gq = b.sA[u] * b.sB[u]; break;
        case 1: gq = dot(b.A[u], v3{g[U], g[U + 1], g[U + 2]}); break;
        case 2: gq = dot(b.A[u], v_j); cav = b.A[u]; break;
        // cross-product bodies: a . (b x c) = c . (a x b), so the cross product the source-side sum needs anyway also gives gq
        case 3: cavb = cross(b.A[u], b.B[u]); gq = dot(vb_j, cavb); break;
        case 4: gq = b.sA[u] * dot(b.A[u], vb_j); cavb = b.A[u]; break;                 // (sbar_i joins q below)
        case 5: gq = dot(b.A[u], vb_j); cavb = b.A[u]; break;
        case 6: gq = b.sA[u] * dot(b.A[u], v_j); cav = b.A[u]; break;
        case 7: cav = cross(b.A[u], b.B[u]); gq = dot(v_j, cav); break;
        default: cavb = cross(b.A[u], b.B[u]); gq = dot(vb_j, cavb); break;
      }
      // (the filter-gradient sums leave out the node's phi: it multiplies them once, when the node is done)
#pragma unroll
      for (int m = 0; m < R / 2; ++m) G2[m] = __builtin_elementwise_fma(pd_f2{gq, gq}, pd_f2{g[2 * m], g[2 * m + 1]}, G2[m]);
      GR = fmaf(gq, g[R], GR);
      // (g_phi = sum_e gq_e w_e = W . (sum_e gq_e g_e): from the node's sums, once per node -- the bodies without a
      //  source-side vector term, k = 0 and 1, never evaluate their filter)
      if (K >= 2) {
        const float w = filt_pk<R>(W, g);
        const float q = (K == 4 || K == 6) ? p * w * b.sA[u] : p * w;
        if (K == 2 || K == 6 || K == 7) axpy(av, q, cav);
        if (K == 3 || K == 4 || K == 5 || K == 8) axpy(avb, q, cavb);
      }
      if (K == 0) axpy(avb, b.hb[u], b.B[u]);                                          // the filter-free term ghb_i v_i
    }
  };
  PdSrcBatch b0, b1;
  issue(b0, 0);
  int e0 = 0;
  for (; e0 + 2 * PD_SRC_EB <= n; e0 += 2 * PD_SRC_EB) {
    issue(b1, e0 + PD_SRC_EB);
    pd_pin();
    compute(b0, e0, std::false_type{});
    issue(b0, min(e0 + 2 * PD_SRC_EB, n_pad - PD_SRC_EB));
    pd_pin();
    compute(b1, e0 + PD_SRC_EB, std::false_type{});
  }
  if (e0 < n) {
    issue(b1, e0 + PD_SRC_EB);
    pd_pin();
    compute(b0, e0, std::true_type{});
    compute(b1, e0 + PD_SRC_EB, std::true_type{});
  }
}

__device__ __forceinline__ int pd_src_filter_of_wave(int wave) {
  // the SIMD with three waves (0, 4, 8) gets the three cheapest bodies (k = 1, 0, 2); the others a cross-product body
  // (k = 3, 7, 8) and a lighter one (k = 4, 6, 5) each.      wave: 0  1  2  3  4  5  6  7  (8 -> 2)
  constexpr unsigned table = (1u << 0) | (3u << 4) | (7u << 8) | (8u << 12) | (0u << 16) | (4u << 20) | (6u << 24) | (5u << 28);
  return wave == 8 ? 2 : (int)((table >> (4 * wave)) & 15u);
}

template <int R>
__global__ __launch_bounds__(576) void pseudo_bwd_src_dense_k(
    const float* __restrict__ phi, const float* __restrict__ s, const float* __restrict__ sbar,
    const float* __restrict__ v, const float* __restrict__ vbar, const float* __restrict__ geom,
    const int* __restrict__ rowptr, const int* __restrict__ dst, const float* __restrict__ Wd,
    const float* __restrict__ bd, const float* __restrict__ gh, const float* __restrict__ ghb,
    const float* __restrict__ gv, const float* __restrict__ gvb, float* __restrict__ g_phi,
    float* __restrict__ g_v, float* __restrict__ g_vbar, float* __restrict__ part, int F, int N, int nodes_per_chunk) {
  constexpr int GS = geom_stride(R);
  __shared__ float red[8][6][64];
  __shared__ __attribute__((aligned(16))) int seg_dst[SEG_LDS];
  const int lane = threadIdx.x & 63;
  const int k = pd_src_filter_of_wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
  const int f_raw = blockIdx.y * 64 + lane;
  const bool live = f_raw < F;
  const int f = live ? f_raw : F - 1;
  float W[R + 1], GR = 0.f;
  pd_f2 G2[R / 2];
  load_row<R>(W, Wd, bd, k * F + f);
#pragma unroll
  for (int m = 0; m < R / 2; ++m) G2[m] = pd_f2{0.f, 0.f};
  const int n_beg = blockIdx.x * nodes_per_chunk, n_end = min(n_beg + nodes_per_chunk, N);
  for (int j = n_beg; j < n_end; ++j) {
    const float p = phi[(size_t)j * 9 * F + (size_t)k * F + f];
    const size_t jf = (size_t)j * F + f;
    const v3 v_j = ldv(v + jf * 3), vb_j = ldv(vbar + jf * 3);
    float a = 0.f;
    v3 av{0.f, 0.f, 0.f}, avb{0.f, 0.f, 0.f};
    pd_f2 N2[R / 2];                                                     // this node's filter-gradient sums, without phi
    float NR = 0.f;
#pragma unroll
    for (int m = 0; m < R / 2; ++m) N2[m] = pd_f2{0.f, 0.f};
    const int e_beg = rowptr[j], e_end = rowptr[j + 1];
    for (int c_beg = e_beg; c_beg < e_end; c_beg += SEG_LDS) {
      const int n = min(SEG_LDS, e_end - c_beg);
      __syncthreads();                                                   // readers of the previous chunk / node
      const int n_pad = pd_stage_indices(seg_dst, dst, c_beg, n, 576);
      const float* __restrict__ geom_c = geom + (size_t)c_beg * GS;
      __syncthreads();
      switch (k) {
#define CGV_PD_SRC(KV) case KV: pseudo_bwd_src_dense_walk<R, KV>(s, sbar, v, vbar, gh, ghb, gv, gvb, geom_c, seg_dst, n, n_pad, F, f, W, p, v_j, vb_j, a, N2, NR, av, avb); break
        CGV_PD_SRC(0); CGV_PD_SRC(1); CGV_PD_SRC(2); CGV_PD_SRC(3); CGV_PD_SRC(4); CGV_PD_SRC(5); CGV_PD_SRC(6); CGV_PD_SRC(7);
        default: pseudo_bwd_src_dense_walk<R, 8>(s, sbar, v, vbar, gh, ghb, gv, gvb, geom_c, seg_dst, n, n_pad, F, f, W, p, v_j, vb_j, a, N2, NR, av, avb); break;
#undef CGV_PD_SRC
      }
    }
    {                                                                    // g_phi[j, k, f] = W_k[f] . N (env term first, as filt())
      pd_f2 a2 = pd_f2{W[R] * NR, 0.f};
#pragma unroll
      for (int m = 0; m < R / 2; ++m) a2 = __builtin_elementwise_fma(pd_f2{W[2 * m], W[2 * m + 1]}, N2[m], a2);
      a = a2.x + a2.y;
    }
#pragma unroll
    for (int m = 0; m < R / 2; ++m) G2[m] = __builtin_elementwise_fma(pd_f2{p, p}, N2[m], G2[m]);
    GR = fmaf(p, NR, GR);
    if (k > 0) {
      float* r = &red[k - 1][0][lane];
      r[0] = av.x; r[64] = av.y; r[128] = av.z; r[192] = avb.x; r[256] = avb.y; r[320] = avb.z;
    }
    __syncthreads();
    if (live) g_phi[(size_t)j * 9 * F + (size_t)k * F + f] = a;
    if (k == 0) {
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) {
        const float* r = &red[w8][0][lane];
        av.x += r[0]; av.y += r[64]; av.z += r[128]; avb.x += r[192]; avb.y += r[256]; avb.z += r[320];
      }
      if (live) {
        const v3 r0 = ldv(g_v + jf * 3), r1 = ldv(g_vbar + jf * 3);        // pass A's receiver-side part
        st3(g_v + jf * 3, r0.x + av.x, r0.y + av.y, r0.z + av.z);
        st3(g_vbar + jf * 3, r1.x + avb.x, r1.y + avb.y, r1.z + avb.z);
      }
    }
    __syncthreads();                                 // red is reused by the next node
  }
  if (live) {
    float* __restrict__ out = part + (size_t)blockIdx.x * 9 * (R + 1) * F;
#pragma unroll
    for (int m = 0; m < R / 2; ++m) {
      out[(size_t)(k * (R + 1) + 2 * m) * F + f] = G2[m].x;
      out[(size_t)(k * (R + 1) + 2 * m + 1) * F + f] = G2[m].y;
    }
    out[(size_t)(k * (R + 1) + R) * F + f] = GR;
  }
}

// gWd[c][n] = sum over chunks of part[chunk][k][n][f] (c = k F + f), gbd likewise for n = R.  Block = 64 channels x 4
// chunk slices: slice s sums chunks s, s+4, ... and the four partial sums meet in LDS in slice order (deterministic);
// the serial version (one thread per output walking all 64 chunks of a 96-bead batch) took 17.8 us per layer.
constexpr int PRED_SLICES = 4;
__global__ __launch_bounds__(64 * PRED_SLICES) void pseudo_bwd_reduce(const float* __restrict__ part, int n_chunks, int R,
                                                                      int F, float* __restrict__ gWd,
                                                                      float* __restrict__ gbd) {
  __shared__ float red[PRED_SLICES][64];
  const int f = blockIdx.x * 64 + threadIdx.x;
  const int sl = threadIdx.y;
  const int n = blockIdx.y, k = blockIdx.z;
  float acc = 0.f;
  if (f < F) {
    const size_t stride = (size_t)9 * (R + 1) * F;
    const float* p = part + ((size_t)k * (R + 1) + n) * F + f;
    constexpr int CB = 4;                            // chunks loaded together (clamped index, surplus zeroed; same order)
    for (int c0 = sl; c0 < n_chunks; c0 += PRED_SLICES * CB) {
      float v[CB];
#pragma unroll
      for (int u = 0; u < CB; ++u) v[u] = p[(size_t)min(c0 + PRED_SLICES * u, n_chunks - 1) * stride];
#pragma unroll
      for (int u = 0; u < CB; ++u) acc += c0 + PRED_SLICES * u < n_chunks ? v[u] : 0.f;
    }
  }
  red[sl][threadIdx.x] = acc;
  __syncthreads();
  if (sl != 0 || f >= F) return;
  float tot = 0.f;
#pragma unroll
  for (int t = 0; t < PRED_SLICES; ++t) tot += red[t][threadIdx.x];
  const int c_out = k * F + f;
  if (n < R) gWd[(size_t)c_out * R + n] = tot; else gbd[c_out] = tot;
}

// Source-node chunks of pass B (= partial filter-gradient sets): one node per chunk up to 64 nodes (12-bead chignolin
// batch; 64 beads x 61 edges of the 2000-atom config: 119 us with 64 chunks, 150 with 32), a quarter of the nodes per
// chunk count above that -- a 96-bead dipeptide batch has 2 edges per node, and 24 chunks of 4 nodes (13.0 us) beat 64
// chunks of 1-2 (19.5 us: the per-block filter staging dominates).  cgv_set_option(CGV_OPT_PSEUDO_CHUNKS, cap) overrides the cap (experiments).
static inline int pseudo_chunks(int n) {
  if (const int cap = cgv::option(CGV_OPT_PSEUDO_CHUNKS); cap > 0) return n < cap ? (n > 0 ? n : 1) : cap;
  if (n <= 64) return n > 0 ? n : 1;
  const int c = n / 4 > 24 ? n / 4 : 24;
  return c < 64 ? c : 64;
}

}  // namespace cgv

extern "C" {

int cgv_pseudo_msg_fwd(const float* phi, const float* s, const float* sbar, const float* v, const float* vbar,
                       const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* Wd,
                       const float* bd, float* dh, float* dhbar, float* dv, float* dvbar, int n_nodes, int n_feat,
                       int n_rbf, int residual, void* stream) {
  return cgv_pseudo_msg_fwd_rows(phi, s, sbar, v, vbar, geom_d, rowptr_d, src_d, Wd, bd, dh, dhbar, dv, dvbar, nullptr,
                                 n_nodes, n_feat, n_rbf, residual, 0, stream);
}

/* As cgv_pseudo_msg_fwd; dv_rows (or NULL) [3 N, F] additionally receives dv as rows 3 i + xyz -- the operand layout
 * of UpdateBlock's u_mat / v_mat products (conv.py:591), which otherwise costs a transpose launch per decoder layer. */
int cgv_pseudo_msg_fwd_rows(const float* phi, const float* s, const float* sbar, const float* v, const float* vbar,
                            const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* Wd,
                            const float* bd, float* dh, float* dhbar, float* dv, float* dvbar, float* dv_rows,
                            int n_nodes, int n_feat, int n_rbf, int residual, int64_t n_edges_hint, void* stream) {
  CGV_REQUIRE(n_nodes >= 0 && n_feat > 0, "bad size");
  if (n_nodes == 0) return 0;
  CGV_REQUIRE(phi && s && sbar && v && vbar && rowptr_d && Wd && bd && dh && dhbar && dv && dvbar, "null pointer");
  dim3 grid(n_nodes, (n_feat + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  // n_edges_hint (0 = unknown) only picks the variant: >= 16 edges per node -> 8 edges' gathers in flight (long segments:
  // gather latency, not registers, sets the pace); results do not depend on it
  const bool dense = n_edges_hint >= 16LL * n_nodes;
  CGV_DISPATCH_RBF(n_rbf, {
#define CGV_PF(EBV, STV) hipLaunchKernelGGL((cgv::pseudo_fwd_k<RBF, EBV, STV>), grid, dim3(576), 0, st, phi, s, sbar, v, vbar, geom_d, \
                         rowptr_d, src_d, Wd, bd, dh, dhbar, dv, dvbar, n_feat, residual, dv_rows)
    const int variant = cgv::option(CGV_OPT_PSEUDO_FWD);
    if (variant == 1) CGV_PF(2, false);
    else if (variant == 2) CGV_PF(2, true);
    else if (variant == 3) CGV_PF(4, true);
    else if (variant == 4) CGV_PF(8, true);
    else if (variant == 5) CGV_PF(8, false);
    else if (variant == 6) CGV_PF(1, true);
    else if (dense && n_nodes < (1 << 24) && 9L * n_feat * 4 < (1L << 24) && (long)n_nodes * 9 * n_feat * 4 < (1L << 31))   // (32-bit gather offsets from 24-bit factors)
      hipLaunchKernelGGL((cgv::pseudo_fwd_dense_k<RBF>), grid, dim3(576), 0, st, phi, s, sbar, v, vbar, geom_d, rowptr_d, src_d,
                         Wd, bd, dh, dhbar, dv, dvbar, n_feat, residual, dv_rows);
    else if (dense)        // 2000-atom config (64 beads, 61 edges each), per layer: EB 2 staged 56 us, EB 2 59, EB 4 / 8 staged 73 / 71, EB 8 80
      CGV_PF(2, true);
    else if ((long)grid.x * grid.y <= 256)
      hipLaunchKernelGGL((cgv::pseudo_fwd_k<RBF, cgv::PSEUDO_EB_WIDE>), grid, dim3(576), 0, st, phi, s, sbar, v, vbar, geom_d,
                         rowptr_d, src_d, Wd, bd, dh, dhbar, dv, dvbar, n_feat, residual, dv_rows);
    else
      hipLaunchKernelGGL((cgv::pseudo_fwd_k<RBF, cgv::PSEUDO_EB_NARROW>), grid, dim3(576), 0, st, phi, s, sbar, v, vbar, geom_d,
                         rowptr_d, src_d, Wd, bd, dh, dhbar, dv, dvbar, n_feat, residual, dv_rows);
  });
  return cgv::check_launch("cgv_pseudo_msg_fwd");
}

size_t cgv_pseudo_msg_bwd_workspace_bytes(int n_nodes, int n_feat, int n_rbf) {
  return sizeof(float) * (size_t)cgv::pseudo_chunks(n_nodes) * 9 * (n_rbf + 1) * n_feat + 256;
}

static int pseudo_msg_bwd_impl(const float* phi, const float* s, const float* sbar, const float* v, const float* vbar,
                       const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* geom_s,
                       const int32_t* rowptr_s, const int32_t* dst_s, const float* Wd, const float* bd, const float* gh,
                       const float* ghbar, const float* gv, const float* gvbar, float* g_phi, float* g_s, float* g_sbar,
                       float* g_v, float* g_vbar, float* gWd, float* gbd, int n_nodes, int n_feat, int n_rbf,
                       int residual, int64_t n_edges_hint, void* workspace, size_t workspace_bytes, void* stream,
                       int* n_chunks_out) {
  CGV_REQUIRE(n_nodes >= 0 && n_feat > 0, "bad size");
  CGV_REQUIRE(phi && s && sbar && v && vbar && rowptr_d && rowptr_s && Wd && bd, "null input");
  CGV_REQUIRE(g_phi && g_s && g_sbar && g_v && g_vbar && (n_chunks_out || (gWd && gbd)) && workspace, "null output");
  if (workspace_bytes < cgv_pseudo_msg_bwd_workspace_bytes(n_nodes, n_feat, n_rbf)) {
    cgv::set_error("cgv_pseudo_msg_bwd: workspace too small");
    return CGV_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int chunks = cgv::pseudo_chunks(n_nodes);
  const int npc = n_nodes > 0 ? (n_nodes + chunks - 1) / chunks : 1;
  float* part = reinterpret_cast<float*>(workspace);
  dim3 gridA(n_nodes > 0 ? n_nodes : 1, (n_feat + 63) / 64), gridB(chunks, (n_feat + 63) / 64);
  const bool dense = n_edges_hint >= 16LL * n_nodes;
  // dense bead graph, all four upstream gradients present, 32-bit gather offsets from 24-bit factors: the per-filter kernels
  const bool lean = dense && gh && ghbar && gv && gvbar && cgv::option(CGV_OPT_PSEUDO_FWD) == 0 && n_nodes < (1 << 24) &&
                    9L * n_feat * 4 < (1L << 24) && (long)n_nodes * 9 * n_feat * 4 < (1L << 31);
  CGV_DISPATCH_RBF(n_rbf, {
    if (lean) {
      hipLaunchKernelGGL((cgv::pseudo_bwd_recv_dense_k<RBF>), gridA, dim3(384), 0, st, phi, v, vbar, geom_d, rowptr_d, src_d, Wd, bd,
                         gh, ghbar, gv, gvbar, g_s, g_sbar, g_v, g_vbar, n_feat, residual);
      hipLaunchKernelGGL((cgv::pseudo_bwd_src_dense_k<RBF>), gridB, dim3(576), 0, st, phi, s, sbar, v, vbar, geom_s, rowptr_s, dst_s,
                         Wd, bd, gh, ghbar, gv, gvbar, g_phi, g_v, g_vbar, part, n_feat, n_nodes, npc);
    } else {
    if (n_nodes > 0 && (long)gridA.x * gridA.y <= 256)
      hipLaunchKernelGGL((cgv::pseudo_bwd_recv_k<RBF, cgv::PSEUDO_EB_WIDE>), gridA, dim3(64), 0, st, phi, v, vbar, geom_d,
                         rowptr_d, src_d, Wd, bd, gh, ghbar, gv, gvbar, g_s, g_sbar, g_v, g_vbar, n_feat, residual);
    else if (n_nodes > 0)
      hipLaunchKernelGGL((cgv::pseudo_bwd_recv_k<RBF, cgv::PSEUDO_EB_NARROW>), gridA, dim3(64), 0, st, phi, v, vbar, geom_d,
                         rowptr_d, src_d, Wd, bd, gh, ghbar, gv, gvbar, g_s, g_sbar, g_v, g_vbar, n_feat, residual);
    if (dense && cgv::option(CGV_OPT_PSEUDO_FWD) == 4)     // (8 edges' gathers in flight: 119.7 us against 119.1 with 2 -- not the limiter)
      hipLaunchKernelGGL((cgv::pseudo_bwd_src_k<RBF, 8>), gridB, dim3(576), 0, st, phi, s, sbar, v, vbar, geom_s, rowptr_s,
                         dst_s, Wd, bd, gh, ghbar, gv, gvbar, g_phi, g_v, g_vbar, part, n_feat, n_nodes, npc);
    else if (dense && cgv::option(CGV_OPT_PSEUDO_FWD) != 5)           // records + receiver indices staged in LDS (5: the plain walk, A/B)
      hipLaunchKernelGGL((cgv::pseudo_bwd_src_k<RBF, 2, true>), gridB, dim3(576), 0, st, phi, s, sbar, v, vbar, geom_s, rowptr_s,
                         dst_s, Wd, bd, gh, ghbar, gv, gvbar, g_phi, g_v, g_vbar, part, n_feat, n_nodes, npc);
    else
      hipLaunchKernelGGL((cgv::pseudo_bwd_src_k<RBF, 2>), gridB, dim3(576), 0, st, phi, s, sbar, v, vbar, geom_s, rowptr_s,
                         dst_s, Wd, bd, gh, ghbar, gv, gvbar, g_phi, g_v, g_vbar, part, n_feat, n_nodes, npc);
    }
  });
  if (n_chunks_out) {               // the caller finishes the filter gradients (cgv_filter_reduce_jobs, K = 9)
    *n_chunks_out = chunks;
    return cgv::check_launch("cgv_pseudo_msg_bwd_deferred");
  }
  dim3 rgrid((n_feat + 63) / 64, n_rbf + 1, 9);
  hipLaunchKernelGGL(cgv::pseudo_bwd_reduce, rgrid, dim3(64, cgv::PRED_SLICES), 0, st, part, chunks, n_rbf, n_feat, gWd, gbd);
  return cgv::check_launch("cgv_pseudo_msg_bwd");
}

int cgv_pseudo_msg_bwd(const float* phi, const float* s, const float* sbar, const float* v, const float* vbar,
                       const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* geom_s,
                       const int32_t* rowptr_s, const int32_t* dst_s, const float* Wd, const float* bd, const float* gh,
                       const float* ghbar, const float* gv, const float* gvbar, float* g_phi, float* g_s, float* g_sbar,
                       float* g_v, float* g_vbar, float* gWd, float* gbd, int n_nodes, int n_feat, int n_rbf,
                       int residual, int64_t n_edges_hint, void* workspace, size_t workspace_bytes, void* stream) {
  return pseudo_msg_bwd_impl(phi, s, sbar, v, vbar, geom_d, rowptr_d, src_d, geom_s, rowptr_s, dst_s, Wd, bd, gh, ghbar, gv, gvbar,
                             g_phi, g_s, g_sbar, g_v, g_vbar, gWd, gbd, n_nodes, n_feat, n_rbf, residual, n_edges_hint, workspace,
                             workspace_bytes, stream, nullptr);
}

/* As cgv_pseudo_msg_bwd without its last launch: the per-chunk partial sums of the filter gradients stay in `workspace`
 * ([n_chunks][9][n_rbf + 1][F]); a training step hands them, with those of its other message blocks, to ONE
 * cgv_filter_reduce_jobs launch at the end of backward (job {workspace, gWd, gbd, *n_chunks, K = 9, n_rbf, F}). */
int cgv_pseudo_msg_bwd_deferred(const float* phi, const float* s, const float* sbar, const float* v, const float* vbar,
                                const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* geom_s,
                                const int32_t* rowptr_s, const int32_t* dst_s, const float* Wd, const float* bd, const float* gh,
                                const float* ghbar, const float* gv, const float* gvbar, float* g_phi, float* g_s, float* g_sbar,
                                float* g_v, float* g_vbar, int n_nodes, int n_feat, int n_rbf, int residual,
                                int64_t n_edges_hint, void* workspace, size_t workspace_bytes, int* n_chunks, void* stream) {
  CGV_REQUIRE(n_chunks, "null pointer");
  return pseudo_msg_bwd_impl(phi, s, sbar, v, vbar, geom_d, rowptr_d, src_d, geom_s, rowptr_s, dst_s, Wd, bd, gh, ghbar, gv, gvbar,
                             g_phi, g_s, g_sbar, g_v, g_vbar, nullptr, nullptr, n_nodes, n_feat, n_rbf, residual, n_edges_hint,
                             workspace, workspace_bytes, stream, n_chunks);
}

}  // extern "C"
