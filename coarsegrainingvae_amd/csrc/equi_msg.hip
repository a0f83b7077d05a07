// K2 / K4: fused EquiMessageBlock and ContractiveMessageBlock (forward + backward).
//
// Reference math (CoarseGrainingVAE/conv.py:505-563, InvariantMessage 63-75, DistanceEmbed
// modules.py:192-197; contraction variant conv.py:703-733), per directed edge e = (i <- j):
//     w_k(e,f) = sum_n Wd[kF+f][n] * a_n(e) + bd[kF+f] * env(e)            k = 0,1,2
//     m_k(e,f) = phi[j, kF+f] * w_k(e,f)
//     ds[i,f]   = sum_e m_1                dv[i,f,:] = sum_e ( m_2 * unit_e + m_0 * v[j,f,:] )
// The reference materialises ~10 tensors of shape [E,3F] / [E,F,3] per layer and reduces them
// with an unsorted scatter_add.  Here a wave owns (receiver node, 128 channels): it walks the
// node's CSR segment, gathers phi[j] / v[j] rows coalesced along channels (L2 resident),
// rebuilds the filter from the per-edge geometry record held in SGPRs (the record address is
// wave-uniform -> scalar loads) and accumulates in VGPRs.  Nothing of size E*F is written.
//
// What bounds it (rocprofv3 PMC, chignolin layer, E = 41.5k, F = 600): the kernel is VALU-issue
// bound, not HBM bound -- a plain wave64 v_fma_f32 occupies its SIMD for 4 cycles, and the
// filter rebuild is 3(R+1) FMAs per (edge, channel).  So each lane carries TWO adjacent channels
// as 2-wide vectors: every FMA below is a v_pk_fma_f32 (same issue cost, two channels), loads are
// 8 B / 24 B per lane, and the (x0,y0)(z0,x1)(y1,z1) layout of two channels' xyz triples lines up
// with the doubled unit vector in the geometry record (cgv_common.h), so the vector channel needs
// no register shuffles.  MFMA was considered and rejected: the contraction is K = n_rbf+1 = 9..11
// deep and fp32; v_mfma_f32_16x16x4_f32 runs at exactly the packed-VALU rate (MI355X_MICROARCH.md),
// so it could not beat this formulation.
//
// Scheduling: 1-D grid, XCD-aware (block b -> XCD b % 8 gets a contiguous node range, so an XCD's
// gathers stay inside a few frames and its phi/v working set fits the 4 MiB L2); on high-degree
// graphs the 4 waves of a block split one node's segment and meet in LDS.
#include <stdlib.h>
#include <cstring>
#include "cgv_common.h"
#include "equi_msg_dev.h"

namespace cgv {

// ------------------------------------------------------------------ forward
// grid = 8 * nodes_per_xcd * tiles blocks (tiles = ceil(F/128)), block = 64 * SPLIT threads.
template <int R, bool WITH_DV, int SPLIT, bool PAIR>
__global__ __launch_bounds__(64 * SPLIT) void equi_msg_fwd_k(const float* __restrict__ phi, const float* __restrict__ v,
                                                             const float* __restrict__ geom,
                                                             const int* __restrict__ rowptr, const int* __restrict__ src,
                                                             const float* __restrict__ Wd, const float* __restrict__ bd,
                                                             float* __restrict__ ds, float* __restrict__ dv, int F,
                                                             int n_dst, int nodes_per_xcd, int tiles,
                                                             const float* __restrict__ s_res,
                                                             const float* __restrict__ v_res) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int local = slot / tiles;
  const int node = xcd * nodes_per_xcd + local;
  const int tile = slot - local * tiles;
  if (node >= n_dst || local >= nodes_per_xcd) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform -> record loads stay scalar
  const ChanPair cp = chan_pair(tile, lane, F);

  f2 W0[R + 1], W1[R + 1], W2[R + 1];
  if constexpr (PAIR) {
    constexpr int NSL = WITH_DV ? 3 : 1;
    __shared__ __attribute__((aligned(16))) float wt[NSL * 128 * R];
    int sl[NSL];
#pragma unroll
    for (int k = 0; k < NSL; ++k) sl[k] = WITH_DV ? k : 1;
    stage_filter_tile<R, NSL>(wt, Wd, sl, F, tile * 128);
    const int cl = cp.c - tile * 128;              // even; clamped lanes stay inside the staged tile
    if constexpr (WITH_DV) {
      read_filter_rows2<R>(W0, wt, bd, cl, cp.c);
      read_filter_rows2<R>(W1, wt + 128 * R, bd, cl, F + cp.c);
      read_filter_rows2<R>(W2, wt + 2 * 128 * R, bd, cl, 2 * F + cp.c);
    } else {
      read_filter_rows2<R>(W1, wt, bd, cl, F + cp.c);
    }
  } else {
    load_filter_rows2<R>(W1, Wd, bd, F + cp.c, F + cp.c1);
    if constexpr (WITH_DV) {
      load_filter_rows2<R>(W0, Wd, bd, cp.c, cp.c1);
      load_filter_rows2<R>(W2, Wd, bd, 2 * F + cp.c, 2 * F + cp.c1);
    }
  }

  f2 acc_s = splat(0.f), accA = splat(0.f), accB = splat(0.f), accC = splat(0.f);
  int beg = rowptr[node], end = rowptr[node + 1];
  if constexpr (SPLIT > 1) {
    const int len = (end - beg + SPLIT - 1) / SPLIT;
    beg = min(beg + wave * len, end);
    end = min(beg + len, end);
  }
  const unsigned row_bytes = 12u * (unsigned)F;              // bytes per node row of phi [3F] AND of v [F,3]
  const unsigned oc = 4u * (unsigned)cp.c, oF = 4u * (unsigned)F, ov = 12u * (unsigned)cp.c;
  const rsrc_t r_phi = make_rsrc(phi), r_v = make_rsrc(WITH_DV ? v : phi);
  if constexpr (PAIR) {
    // Software-pipelined walk (the probe in tools/gather_probe.py showed the time does not depend on WHERE
    // the gathers hit -- the waves were serialising scalar-load latency, gather latency and FMAs per edge):
    // record e+1 and source index e+2 are requested, and the gathers of edge e+1 issued, before the FMAs of
    // edge e run, so both latencies hide behind ~200 cycles of packed math of this very wave.
    constexpr int NG = U + 6;                        // record floats actually used
    if (beg < end) {
      float gc[NG], gn[NG];
#pragma unroll
      for (int t = 0; t < NG; ++t) gc[t] = geom[(size_t)beg * GS + t];
      unsigned so_c = (unsigned)src[beg] * row_bytes;
      unsigned so_n = (unsigned)src[min(beg + 1, end - 1)] * row_bytes;
      f2 c_p0 = splat(0.f), c_p1, c_p2 = splat(0.f), c_A = splat(0.f), c_B = splat(0.f), c_C = splat(0.f);
      c_p1 = ld2_buf(r_phi, oc + oF, so_c);
      if constexpr (WITH_DV) {
        c_p0 = ld2_buf(r_phi, oc, so_c); c_p2 = ld2_buf(r_phi, oc + 2u * oF, so_c);
        ldvec_buf(r_v, ov, so_c, c_A, c_B, c_C);
      }
#pragma unroll 2
      for (int e = beg; e < end; ++e) {
        const int e1 = min(e + 1, end - 1), e2 = min(e + 2, end - 1);
#pragma unroll
        for (int t = 0; t < NG; ++t) gn[t] = geom[(size_t)e1 * GS + t];          // scalar prefetch: record e+1
        const unsigned so_nn = (unsigned)src[e2] * row_bytes;                     // and source index e+2
        f2 n_p0 = splat(0.f), n_p1, n_p2 = splat(0.f), n_A = splat(0.f), n_B = splat(0.f), n_C = splat(0.f);
        n_p1 = ld2_buf(r_phi, oc + oF, so_n);                                     // gathers of edge e+1
        if constexpr (WITH_DV) {
          n_p0 = ld2_buf(r_phi, oc, so_n); n_p2 = ld2_buf(r_phi, oc + 2u * oF, so_n);
          ldvec_buf(r_v, ov, so_n, n_A, n_B, n_C);
        }
        acc_s = fma2(c_p1, filter2<R>(W1, gc), acc_s);
        if constexpr (WITH_DV) {
          const f2 m0 = c_p0 * filter2<R>(W0, gc);
          const f2 m2 = c_p2 * filter2<R>(W2, gc);
          const f2 u01 = f2{gc[U], gc[U + 1]}, u20 = f2{gc[U + 2], gc[U + 3]}, u12 = f2{gc[U + 4], gc[U + 5]};
          accA = fma2(lo2(m2), u01, fma2(lo2(m0), c_A, accA));
          accB = fma2(m2, u20, fma2(m0, c_B, accB));
          accC = fma2(hi2(m2), u12, fma2(hi2(m0), c_C, accC));
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) gc[t] = gn[t];
        so_n = so_nn;
        c_p0 = n_p0; c_p1 = n_p1; c_p2 = n_p2; c_A = n_A; c_B = n_B; c_C = n_C;
      }
    }
  } else {
  for (int e = beg; e < end; ++e) {
    const float* __restrict__ g = geom + (size_t)e * GS;   // wave-uniform -> s_load
    const int j = src[e];
    const float* __restrict__ prow = phi + (size_t)j * 3 * F;
    const f2 p1 = ldpair<PAIR>(prow + F, cp);
    acc_s = fma2(p1, filter2<R>(W1, g), acc_s);
    if constexpr (WITH_DV) {
      f2 A, B, C;
      const f2 p0 = ldpair<PAIR>(prow, cp);
      const f2 p2 = ldpair<PAIR>(prow + 2 * F, cp);
      ldvec<PAIR>(v + (size_t)j * F * 3, cp, A, B, C);
      const f2 m0 = p0 * filter2<R>(W0, g);
      const f2 m2 = p2 * filter2<R>(W2, g);
      const f2 u01 = f2{g[U], g[U + 1]}, u20 = f2{g[U + 2], g[U + 3]}, u12 = f2{g[U + 4], g[U + 5]};
      accA = fma2(lo2(m2), u01, fma2(lo2(m0), A, accA));
      accB = fma2(m2, u20, fma2(m0, B, accB));
      accC = fma2(hi2(m2), u12, fma2(hi2(m0), C, accC));
    }
  }
  }
  if constexpr (SPLIT > 1) {
    __shared__ float red[(SPLIT - 1) * 8 * 64];
    if (wave > 0) {
      float* r = red + (wave - 1) * 8 * 64 + lane;
      r[0] = acc_s.x; r[64] = acc_s.y; r[128] = accA.x; r[192] = accA.y;
      r[256] = accB.x; r[320] = accB.y; r[384] = accC.x; r[448] = accC.y;
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < SPLIT - 1; ++w) {
      const float* r = red + w * 8 * 64 + lane;
      acc_s += f2{r[0], r[64]}; accA += f2{r[128], r[192]}; accB += f2{r[256], r[320]}; accC += f2{r[384], r[448]};
    }
  }
  if (cp.live) {
    if (s_res) acc_s += ldpair<PAIR>(s_res + (size_t)node * F, cp);      // emit h + ds (cgvae.py:287, 309, 391)
    stpair<PAIR>(ds + (size_t)node * F, cp, acc_s);
    if constexpr (WITH_DV) {
      if (v_res) {
        f2 rA, rB, rC;
        ldvec<PAIR>(v_res + (size_t)node * F * 3, cp, rA, rB, rC);
        accA += rA; accB += rB; accC += rC;
      }
      stvec<PAIR>(dv + (size_t)node * F * 3, cp, accA, accB, accC);
    }
  }
}

// ------------------------------------------------------------------ forward, matrix-core variant
// The filter rebuild  w_k(e, c) = sum_n a_n(e) Wd[kF+c][n] + env(e) bd[kF+c]  is a [edges x (R+1)] x
// [(R+1) x channels] product: 3(R+1) of the ~45 FMAs per (edge, channel).  f32 MFMA has the SAME peak as
// packed VALU (MI355X_MICROARCH.md), so it cannot make the product itself faster -- but it runs on the
// separate matrix pipe, concurrently with the VALU.  Here a wave takes 16 edges of its receiver and
// 64 channels at a time:  A[i = edge][k = rbf] = geometry record (one dword per lane and k-step),
// B[k = rbf][j = channel] = filter weights (loop-invariant registers), D[edge][channel] = w.
// v_mfma_f32_16x16x4_f32 leaves lane (j = l&15, q = l>>4) with w of channel j for edges 4q..4q+3, so
// the VALU part (gather phi / v of those 4 edges' sources, multiply, accumulate) follows in the same
// lane without any shuffle; the 4 edge groups q meet through two __shfl_xor at the end of the segment.
// VALU work drops from ~45 to ~10 instructions per (edge, channel); the matrix pipe carries the rest.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ld1_buf(rsrc_t r, unsigned voff_bytes) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_bytes, 0u, 0));
}

template <int R, bool WITH_DV, int SPLIT>
__global__ __launch_bounds__(64 * SPLIT) void equi_msg_fwd_mfma_k(
    const float* __restrict__ phi, const float* __restrict__ v, const float* __restrict__ geom,
    const int* __restrict__ rowptr, const int* __restrict__ src, const float* __restrict__ Wd,
    const float* __restrict__ bd, float* __restrict__ ds, float* __restrict__ dv, int F, int n_dst, int nodes_per_xcd,
    int tiles, const float* __restrict__ s_res, const float* __restrict__ v_res) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  constexpr int KS = (R + 1 + 3) / 4;           // k-steps of 4 over the R+1 filter terms
  constexpr int NK = WITH_DV ? 3 : 1;           // filter slices evaluated
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int local = slot / tiles;
  const int node = xcd * nodes_per_xcd + local;
  const int tile = slot - local * tiles;
  if (node >= n_dst || local >= nodes_per_xcd) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int c0 = tile * 64;

  // B operands: Wb[kk][nb][ks] = filter term (4 ks + q) of channel c0 + 16 nb + j in slice k
  float Wb[NK][4][KS];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int k = WITH_DV ? kk : 1;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int c = c0 + 16 * nb + j;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int n = 4 * ks + q;
        float w = 0.f;
        if (c < F) {
          if (n < R) w = Wd[(size_t)(k * F + c) * R + n];
          else if (n == R) w = bd[k * F + c];
        }
        Wb[kk][nb][ks] = w;
      }
    }
  }
  // per-lane byte offsets of its 4 channels (clamped: lanes beyond F read a valid channel, their B is 0)
  unsigned cb[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) cb[nb] = (unsigned)min(c0 + 16 * nb + j, F - 1);
  const unsigned row_bytes = 12u * (unsigned)F;
  const rsrc_t r_phi = make_rsrc(phi), r_v = make_rsrc(WITH_DV ? v : phi);

  float as[4] = {0.f, 0.f, 0.f, 0.f};
  float ax[4] = {0.f, 0.f, 0.f, 0.f}, ay[4] = {0.f, 0.f, 0.f, 0.f}, az[4] = {0.f, 0.f, 0.f, 0.f};
  int beg = rowptr[node], end = rowptr[node + 1];
  if constexpr (SPLIT > 1) {
    const int len = (((end - beg + SPLIT - 1) / SPLIT) + 15) & ~15;      // whole 16-edge tiles per wave
    beg = min(beg + wave * len, end);
    end = min(beg + len, end);
  }
  for (int t0 = beg; t0 < end; t0 += 16) {
    // A operand: lane (i = l&15, kq = l>>4) supplies record entry 4 ks + kq of edge t0 + i (0 past the end)
    const int eA = t0 + j;
    const bool okA = eA < end;
    float a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = (okA && 4 * ks + q <= R) ? geom[(size_t)eA * GS + 4 * ks + q] : 0.f;
    f32x4 D[NK][4];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        D[kk][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          D[kk][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], Wb[kk][nb][ks], D[kk][nb], 0, 0, 0);
      }
    // VALU part: this lane's 4 edges t0 + 4 q + r (rows past the end have w == 0: only addresses are clamped)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = min(t0 + 4 * q + r, end - 1);
      const unsigned base = (unsigned)src[e] * row_bytes;
      float ux = 0.f, uy = 0.f, uz = 0.f;
      if constexpr (WITH_DV) {
        const f3 u = ld3(geom + (size_t)e * GS + U);
        ux = u.x; uy = u.y; uz = u.z;
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const unsigned oc = base + 4u * cb[nb];
        const float p1 = ld1_buf(r_phi, oc + 4u * (unsigned)F);
        as[nb] = fmaf(p1, D[WITH_DV ? 1 : 0][nb][r], as[nb]);
        if constexpr (WITH_DV) {
          const float p0 = ld1_buf(r_phi, oc);
          const float p2 = ld1_buf(r_phi, oc + 8u * (unsigned)F);
          const unsigned ovv = base + 12u * cb[nb];
          const float vx = ld1_buf(r_v, ovv), vy = ld1_buf(r_v, ovv + 4u), vz = ld1_buf(r_v, ovv + 8u);
          const float m0 = p0 * D[0][nb][r], m2 = p2 * D[2][nb][r];
          ax[nb] = fmaf(m2, ux, fmaf(m0, vx, ax[nb]));
          ay[nb] = fmaf(m2, uy, fmaf(m0, vy, ay[nb]));
          az[nb] = fmaf(m2, uz, fmaf(m0, vz, az[nb]));
        }
      }
    }
  }
  // the 4 edge groups (lanes l, l^16, l^32, l^48) hold partial sums of the same channels
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    as[nb] += __shfl_xor(as[nb], 16); as[nb] += __shfl_xor(as[nb], 32);
    if constexpr (WITH_DV) {
      ax[nb] += __shfl_xor(ax[nb], 16); ax[nb] += __shfl_xor(ax[nb], 32);
      ay[nb] += __shfl_xor(ay[nb], 16); ay[nb] += __shfl_xor(ay[nb], 32);
      az[nb] += __shfl_xor(az[nb], 16); az[nb] += __shfl_xor(az[nb], 32);
    }
  }
  // lane (j, q) finishes channel c0 + 16 q + j  (nb = q): 64 consecutive channels per wave store
  float os = q == 0 ? as[0] : (q == 1 ? as[1] : (q == 2 ? as[2] : as[3]));
  float ox = q == 0 ? ax[0] : (q == 1 ? ax[1] : (q == 2 ? ax[2] : ax[3]));
  float oy = q == 0 ? ay[0] : (q == 1 ? ay[1] : (q == 2 ? ay[2] : ay[3]));
  float oz = q == 0 ? az[0] : (q == 1 ? az[1] : (q == 2 ? az[2] : az[3]));
  if constexpr (SPLIT > 1) {
    __shared__ float red[(SPLIT - 1) * 4 * 64];
    if (wave > 0) {
      float* rr = red + (wave - 1) * 4 * 64 + lane;
      rr[0] = os; rr[64] = ox; rr[128] = oy; rr[192] = oz;
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < SPLIT - 1; ++w) {
      const float* rr = red + w * 4 * 64 + lane;
      os += rr[0]; ox += rr[64]; oy += rr[128]; oz += rr[192];
    }
  }
  const int c = c0 + 16 * q + j;
  if (c < F) {
    const size_t o = (size_t)node * F + c;
    if (s_res) os += s_res[o];
    ds[o] = os;
    if constexpr (WITH_DV) {
      if (v_res) { const f3 t = ld3(v_res + o * 3); ox += t.x; oy += t.y; oz += t.z; }
      st3(dv + o * 3, ox, oy, oz);
    }
  }
}

// ------------------------------------------------------------------ backward
// Upstream gs[i,f], gv[i,f,:] at the receivers.  With gq_1 = gs, gq_2 = gv.unit, gq_0 = gv.v_j:
//     g_phi[j,kF+f] = sum_{e: src(e)=j} gq_k * w_k          g_v[j,f,:] = sum_e m_0 * gv_i
//     gWd[kF+f][n]  = sum_e gq_k * phi[j,kF+f] * a_n(e)      gbd[kF+f]  = sum_e gq_k * phi * env
// A block owns (chunk of source nodes, 128 channels) and walks the SRC-sorted view, so g_phi and
// g_v are plain stores; gWd/gbd accumulate in registers over the whole chunk and leave as ONE
// partial per block (LDS reduction over its 4 waves), summed by a second kernel in a fixed order:
// deterministic, no atomics.  SPLIT: the 4 waves share one node (slices of its segment) instead
// of taking different nodes -- for high-degree graphs.
// grid = 8 * chunks_per_xcd * tiles (tile-major work items per XCD), block = 256.
constexpr int BWD_WAVES = 4;

template <int R, bool HAS_GV, bool SPLIT, bool PAIR>
__global__ __launch_bounds__(64 * BWD_WAVES) void equi_msg_bwd_k(
    const float* __restrict__ phi, const float* __restrict__ v, const float* __restrict__ geom,
    const int* __restrict__ rowptr, const int* __restrict__ dst, const float* __restrict__ Wd,
    const float* __restrict__ bd, const float* __restrict__ gs, const float* __restrict__ gv,
    float* __restrict__ g_phi, float* __restrict__ g_v, float* __restrict__ part, int F, int n_src,
    int nodes_per_chunk, int chunks_per_xcd, int tiles) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  constexpr int K = HAS_GV ? 3 : 1;           // live filter slices (k = 1 only without gv)
  constexpr int NRED = K * (R + 1) * 2;       // filter-gradient floats per lane
  constexpr int NACC = SPLIT ? (HAS_GV ? 12 : 2) : 0;
  __shared__ float red[(BWD_WAVES - 1) * (NRED + NACC) * 64];

  // work item = (channel tile, chunk), tile-major; XCD x (= blockIdx % 8) takes items [x * chunks_per_xcd * tiles, ...):
  // it sweeps consecutive chunks of one channel tile (as the forward does, equi_msg_grp.hip), so the gs / gv / phi
  // slices its L2 must hold belong to one 128-channel tile, not to all of them
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n_chunks = 8 * chunks_per_xcd;
  const int item = xcd * chunks_per_xcd * tiles + slot;
  const int tile = item / n_chunks;
  const int chunk = item - tile * n_chunks;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform -> record loads stay scalar
  const ChanPair cp = chan_pair(tile, lane, F);

  f2 W[K][R + 1], G[K][R + 1];
  if constexpr (PAIR) {
    __shared__ __attribute__((aligned(16))) float wt[K * 128 * R];
    int sl[K];
#pragma unroll
    for (int k = 0; k < K; ++k) sl[k] = HAS_GV ? k : 1;
    stage_filter_tile<R, K>(wt, Wd, sl, F, tile * 128);
    const int cl = cp.c - tile * 128;
#pragma unroll
    for (int k = 0; k < K; ++k) read_filter_rows2<R>(W[k], wt + k * 128 * R, bd, cl, sl[k] * F + cp.c);
  } else if constexpr (HAS_GV) {
    load_filter_rows2<R>(W[0], Wd, bd, cp.c, cp.c1);
    load_filter_rows2<R>(W[1], Wd, bd, F + cp.c, F + cp.c1);
    load_filter_rows2<R>(W[2], Wd, bd, 2 * F + cp.c, 2 * F + cp.c1);
  } else {
    load_filter_rows2<R>(W[0], Wd, bd, F + cp.c, F + cp.c1);
  }
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int n = 0; n <= R; ++n) G[k][n] = splat(0.f);

  const rsrc_t r_gs = make_rsrc(gs ? gs : phi), r_gv = make_rsrc(HAS_GV ? gv : phi);
  const int n_beg = chunk * nodes_per_chunk;
  const int n_end = min(n_beg + nodes_per_chunk, n_src);
  const int j0 = SPLIT ? n_beg : n_beg + wave;
  const int jstep = SPLIT ? 1 : BWD_WAVES;
  const int n_edges_total = rowptr[n_src];
  float warm = 0.f, pend_g = 0.f;                    // cache-warming loads of the scalar-only walk (see there)
  int pend_i = 0;
  for (int j = j0; j < n_end; j += jstep) {
    const float* __restrict__ prow = phi + (size_t)j * 3 * F;
    const f2 p1 = ldpair<PAIR>(prow + F, cp);
    f2 p0 = splat(0.f), p2 = splat(0.f), vA = splat(0.f), vB = splat(0.f), vC = splat(0.f);
    if constexpr (HAS_GV) {
      p0 = ldpair<PAIR>(prow, cp);
      p2 = ldpair<PAIR>(prow + 2 * F, cp);
      ldvec<PAIR>(v + (size_t)j * F * 3, cp, vA, vB, vC);
    }
    f2 a0 = splat(0.f), a1 = splat(0.f), a2 = splat(0.f), bA = splat(0.f), bB = splat(0.f), bC = splat(0.f);
    int beg = rowptr[j], end = rowptr[j + 1];
    if constexpr (SPLIT) {
      const int len = (end - beg + BWD_WAVES - 1) / BWD_WAVES;
      beg = min(beg + wave * len, end);
      end = min(beg + len, end);
    }
    if constexpr (PAIR && !HAS_GV) {
      // scalar-only upstream (the model's case: the encoder's vector channel feeds nothing): software-pipelined like
      // the forward -- record e+1 and receiver index e+2 are requested, and the gs gather of edge e+1 issued, before
      // the FMAs of edge e run
      if (gs != nullptr && beg < end) {
        constexpr int NGB = R + 1;                                     // a_n and env; the unit vector is not needed
        float gc[NGB], gn[NGB];
        // Per edge only  N[n] += gs_i a_n(e)  is accumulated (R + 1 packed FMAs): both sums of the node follow from it
        // when its segment is done --  g_phi[j] = sum_e gs_i w(e) = W . N  and  gWd += phi[j] N  -- instead of a filter
        // evaluation AND a filter-gradient update per edge (2 (R + 1) + 2 packed FMAs).
        f2 Nn[R + 1];
#pragma unroll
        for (int n = 0; n <= R; ++n) Nn[n] = splat(0.f);
        const unsigned og = 4u * (unsigned)cp.c, rowb = 4u * (unsigned)F;
        // Batches of EB edges: the scalar unit's loads return out of order, so every use waits for all of them -- one
        // wait per BATCH (its EB records and the receiver indices of the next batch) instead of one per edge; the gs
        // gathers of a batch are issued one batch ahead (vector loads: in order, waited for individually).
        // The records and receiver indices are streamed exactly once per (node, channel tile); on graphs whose record
        // array is far beyond L2 (2000 atoms: 68 MB) each scalar load went to HBM: every 64 edges the lanes touch one dword
        // of the 64 records (and receiver indices) that follow the next 64 -- vector loads nobody waits for -- so the
        // scalar loads find their lines in L2.
        constexpr int EB = 4;
        const int e_last = n_edges_total - 1, seg_last = end - 1;
        f2 q_cur[EB], q_nxt[EB];
        int i_nxt[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) q_cur[u] = ld2_buf(r_gs, og, (unsigned)dst[min(beg + u, seg_last)] * rowb);
#pragma unroll
        for (int u = 0; u < EB; ++u) i_nxt[u] = dst[min(beg + EB + u, seg_last)];
        (void)gc; (void)gn;
        for (int eb = beg; eb < end; eb += 64) {
          const int ew = min(eb + 64 + lane, e_last);
          warm = fmaf(pend_g, 0.f, warm) + (float)(pend_i & 0);
          pend_g = geom[(size_t)ew * GS];
          pend_i = dst[ew];
          const int ee = min(eb + 64, end);
          for (int e = eb; e < ee; e += EB) {
            float rec[EB][NGB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
              const float* __restrict__ gp = geom + (size_t)min(e + u, seg_last) * GS;
#pragma unroll
              for (int t = 0; t < NGB; ++t) rec[u][t] = gp[t];
            }
#pragma unroll
            for (int u = 0; u < EB; ++u) q_nxt[u] = ld2_buf(r_gs, og, (unsigned)i_nxt[u] * rowb);
#pragma unroll
            for (int u = 0; u < EB; ++u) i_nxt[u] = dst[min(e + 2 * EB + u, seg_last)];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
              const f2 cq = e + u < ee ? q_cur[u] : splat(0.f);          // (wave-uniform: the last batch of a 64-edge piece)
#pragma unroll
              for (int n = 0; n <= R; ++n) Nn[n] = fma2(cq, splat(rec[u][n]), Nn[n]);
            }
#pragma unroll
            for (int u = 0; u < EB; ++u) q_cur[u] = q_nxt[u];
          }
        }
        a1 = W[0][R] * Nn[R];                                          // (env term first, as filter2)
#pragma unroll
        for (int n = 0; n < R; ++n) a1 = fma2(W[0][n], Nn[n], a1);
#pragma unroll
        for (int n = 0; n <= R; ++n) G[0][n] = fma2(p1, Nn[n], G[0][n]);
      }
    } else {
#pragma unroll 2
    for (int e = beg; e < end; ++e) {
      const float* __restrict__ g = geom + (size_t)e * GS;
      const int i = dst[e];
      f2 gq1 = splat(0.f);
      if (gs) {
        if constexpr (PAIR) gq1 = ld2_buf(r_gs, 4u * (unsigned)cp.c, (unsigned)i * 4u * (unsigned)F);
        else gq1 = ldpair<PAIR>(gs + (size_t)i * F, cp);
      }
      if constexpr (HAS_GV) {
        f2 gA, gB, gC;
        if constexpr (PAIR) {
          const unsigned ov = 12u * (unsigned)cp.c, so = (unsigned)i * 12u * (unsigned)F;
          ldvec_buf(r_gv, ov, so, gA, gB, gC);
        } else {
          ldvec<PAIR>(gv + (size_t)i * F * 3, cp, gA, gB, gC);
        }
        const f2 w0 = filter2<R>(W[0], g), w1 = filter2<R>(W[1], g), w2 = filter2<R>(W[2], g);
        const f2 u01 = f2{g[U], g[U + 1]}, u20 = f2{g[U + 2], g[U + 3]}, u12 = f2{g[U + 4], g[U + 5]};
        const f2 PA = gA * vA, PB = gB * vB, PC = gC * vC;          // (x0x0,y0y0)(z0z0,x1x1)(y1y1,z1z1)
        const f2 QA = gA * u01, QB = gB * u20, QC = gC * u12;
        const f2 gq0 = f2{PA.x + PA.y + PB.x, PB.y + PC.x + PC.y};  // gv_i . v_j per channel
        const f2 gq2 = f2{QA.x + QA.y + QB.x, QB.y + QC.x + QC.y};  // gv_i . unit
        a0 = fma2(gq0, w0, a0);
        a1 = fma2(gq1, w1, a1);
        a2 = fma2(gq2, w2, a2);
        const f2 m0 = p0 * w0;
        bA = fma2(lo2(m0), gA, bA);
        bB = fma2(m0, gB, bB);
        bC = fma2(hi2(m0), gC, bC);
        const f2 t0 = gq0 * p0, t1 = gq1 * p1, t2 = gq2 * p2;
#pragma unroll
        for (int n = 0; n <= R; ++n) {
          const f2 gn = splat(g[n]);
          G[0][n] = fma2(t0, gn, G[0][n]);
          G[1][n] = fma2(t1, gn, G[1][n]);
          G[2][n] = fma2(t2, gn, G[2][n]);
        }
      } else {
        a1 = fma2(gq1, filter2<R>(W[0], g), a1);
        const f2 t1 = gq1 * p1;
#pragma unroll
        for (int n = 0; n <= R; ++n) G[0][n] = fma2(t1, splat(g[n]), G[0][n]);
      }
    }
    }
    if constexpr (SPLIT) {       // the 4 waves hold partial sums of the SAME node: combine in LDS
      float* acc = red + (BWD_WAVES - 1) * NRED * 64;
      if (wave > 0) {
        float* r = acc + (wave - 1) * NACC * 64 + lane;
        r[0] = a1.x; r[64] = a1.y;
        if constexpr (HAS_GV) {
          r[128] = a0.x; r[192] = a0.y; r[256] = a2.x; r[320] = a2.y;
          r[384] = bA.x; r[448] = bA.y; r[512] = bB.x; r[576] = bB.y; r[640] = bC.x; r[704] = bC.y;
        }
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int w = 0; w < BWD_WAVES - 1; ++w) {
          const float* r = acc + w * NACC * 64 + lane;
          a1 += f2{r[0], r[64]};
          if constexpr (HAS_GV) {
            a0 += f2{r[128], r[192]}; a2 += f2{r[256], r[320]};
            bA += f2{r[384], r[448]}; bB += f2{r[512], r[576]}; bC += f2{r[640], r[704]};
          }
        }
      }
      __syncthreads();           // acc region is reused by the next node
    }
    if (cp.live && (!SPLIT || wave == 0)) {
      float* __restrict__ grow = g_phi + (size_t)j * 3 * F;
      stpair<PAIR>(grow, cp, a0);
      stpair<PAIR>(grow + F, cp, a1);
      stpair<PAIR>(grow + 2 * F, cp, a2);
      if constexpr (HAS_GV) stvec<PAIR>(g_v + (size_t)j * F * 3, cp, bA, bB, bC);
    }
  }

  asm volatile("" ::"v"(warm), "v"(pend_g), "v"(pend_i));      // the warming loads must not be optimised away
  // block partial of the filter-weight gradient: part[chunk][k][n][F]  (channel fastest -> coalesced)
  if (wave > 0) {
    float* r = red + (wave - 1) * NRED * 64 + lane;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int n = 0; n <= R; ++n) {
        r[((k * (R + 1) + n) * 2 + 0) * 64] = G[k][n].x;
        r[((k * (R + 1) + n) * 2 + 1) * 64] = G[k][n].y;
      }
  }
  __syncthreads();
  if (wave == 0) {
    for (int w = 0; w < BWD_WAVES - 1; ++w) {
      const float* r = red + w * NRED * 64 + lane;
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int n = 0; n <= R; ++n)
          G[k][n] += f2{r[((k * (R + 1) + n) * 2 + 0) * 64], r[((k * (R + 1) + n) * 2 + 1) * 64]};
    }
    if (cp.live) {
      float* __restrict__ out = part + (size_t)chunk * K * (R + 1) * F;
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int n = 0; n <= R; ++n) stpair<PAIR>(out + ((size_t)k * (R + 1) + n) * F, cp, G[k][n]);
    }
  }
}

// ------------------------------------------------------------------ backward, scalar-only upstream, on the matrix cores
// Without a vector gradient the edge loop of equi_msg_bwd_k accumulates only  N[n][c] += a_n(e) gs[dst(e)][c]  per source
// node (R + 1 packed FMAs per edge and 128-channel wave) -- but each edge costs that wave a record through the scalar cache
// (three loads, ~20 scalar instructions of addressing) and a row gather, and the loop runs at 187 cycles per edge where
// its FMAs need 44 (2000-atom graph: 390 us per launch).  N is a product: A^T [16 x edges] (the record's first 16 floats:
// a_0 .. a_{R-1}, env; the rows beyond R are never read) times the gathered rows [edges x channels], summed over the node's
// edges -- v_mfma_f32_16x16x4_f32 takes 4 edges per instruction, its A operand is ONE coalesced dword load per lane (lane
// (k, n): float n of edge k's record), its B operand the receivers' rows as 16-byte pieces (lane (k, col): channels
// c0 + 64 h + 4 col .. + 3 of edge k's receiver; a tile T = 4 h + t holds channel 4 col + t of half h), no scalar stream, no
// broadcast.  The fp32 MFMA is an fmaf chain over k: the edge order of the sums is the one of the FMA loop.
// A block = (chunk of source nodes, 128 channels), its 4 waves share each node's segment; every wave deposits its 8 tiles
// and wave w finishes tiles 2 w, 2 w + 1 (g_phi = W . N over the rows it holds, then across the four row groups; the filter
// gradient p1 N accumulated over the chunk), so no wave runs a tail alone.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R>
__global__ __launch_bounds__(64 * BWD_WAVES) void equi_msg_bwd_mfma_k(
    const float* __restrict__ phi, const float* __restrict__ geom, const int* __restrict__ rowptr, const int* __restrict__ dst,
    const float* __restrict__ Wd, const float* __restrict__ bd, const float* __restrict__ gs, float* __restrict__ g_phi,
    float* __restrict__ part, int F, int n_src, int nodes_per_chunk, int chunks_per_xcd, int tiles) {
  constexpr int GS = geom_stride(R);
  __shared__ float red[BWD_WAVES * 32 * 64];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n_chunks = 8 * chunks_per_xcd;
  const int item = xcd * chunks_per_xcd * tiles + slot;
  const int tile = item / n_chunks;
  const int chunk = item - tile * n_chunks;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, col = lane & 15;
  const int c0 = tile * 128;
  // columns gathered (clamped into the row: surplus columns are computed and never stored)
  const int cb0 = min(c0 + 4 * col, F - 4), cb1 = min(c0 + 64 + 4 * col, F - 4);
  // the two channels this lane finishes: tiles 2 wave, 2 wave + 1 of its column
  const int cf = c0 + 64 * (wave >> 1) + 4 * col + 2 * (wave & 1);
  const bool live = cf < F;                                         // (F is even: cf + 1 < F as well)
  float Wl[2][4], Gl[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * q + r;
      Gl[u][r] = 0.f;
      Wl[u][r] = (live && n <= R) ? (n < R ? Wd[(size_t)(F + cf + u) * R + n] : bd[F + cf + u]) : 0.f;
    }
  const int n_beg = chunk * nodes_per_chunk;
  const int n_end = min(n_beg + nodes_per_chunk, n_src);
#ifndef CGV_K2BM_DEPTH
#define CGV_K2BM_DEPTH 1
#endif
#ifndef CGV_K2BM_RDEPTH
#define CGV_K2BM_RDEPTH (CGV_K2BM_DEPTH + 2)
#endif
  constexpr int DG = CGV_K2BM_DEPTH, DR = CGV_K2BM_RDEPTH;         // rows DG batches ahead of the products, records DR
  // lane (k = q, n = col) of a batch of 4 edges: the receiver of edge k and float n of its record; edges beyond the slice
  // [b, e_end) contribute a zero row of A (their loads go to the slice's first edge)
  auto rec = [&](int e0, int b, int e_end, int& idx, float& a) {
    const int e = e0 + q;
    const bool ok = e < e_end;
    const int ec = ok ? e : b;
    idx = dst[ec];
    const float t = geom[(size_t)ec * GS + (col < GS ? col : 0)];     // (n_rbf = 4: a record has 12 floats)
    a = ok ? t : 0.f;
  };
  auto gather = [&](int idx, f32x4& B0, f32x4& B1) {
    const float* __restrict__ row = gs + (size_t)idx * F;
    B0 = *reinterpret_cast<const f32x4*>(row + cb0);
    B1 = *reinterpret_cast<const f32x4*>(row + cb1);
  };
  auto slice_of = [&](int j, int& b, int& e_end) {
    const int beg = rowptr[j], end = rowptr[j + 1];
    const int len = (end - beg + BWD_WAVES - 1) / BWD_WAVES;
    b = min(beg + wave * len, end);
    e_end = min(b + len, end);
  };
  int ii[DR];
  float aa[DR];
  f32x4 Bq0[DG + 1], Bq1[DG + 1];
  // first records and rows of a slice: requested before the PREVIOUS node's sums are exchanged and finished, so the two
  // dependent round trips (receiver index -> row) of a node's start run under that tail
  auto open_slice = [&](int b, int e_end) {
    if (b < e_end) {
#pragma unroll
      for (int k = 0; k < DR; ++k) rec(b + 4 * k, b, e_end, ii[k], aa[k]);
#pragma unroll
      for (int k = 0; k < DG; ++k) gather(ii[k], Bq0[k], Bq1[k]);
    }
  };
  int b = 0, e_end = 0;
  if (n_beg < n_end) {
    slice_of(n_beg, b, e_end);
    open_slice(b, e_end);
  }
  for (int j = n_beg; j < n_end; ++j) {
    f32x4 acc[8];
#pragma unroll
    for (int T = 0; T < 8; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int e0 = b; e0 < e_end; e0 += 4) {
      int i_new;
      float a_new;
      rec(e0 + 4 * DR, b, e_end, i_new, a_new);
      gather(ii[DG], Bq0[DG], Bq1[DG]);
      const float a0 = aa[0];
      const f32x4 B0 = Bq0[0], B1 = Bq1[0];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, B0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, B0.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, B0.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, B0.w, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, B1.x, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, B1.y, acc[5], 0, 0, 0);
      acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, B1.z, acc[6], 0, 0, 0);
      acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, B1.w, acc[7], 0, 0, 0);
#pragma unroll
      for (int k = 0; k + 1 < DR; ++k) { ii[k] = ii[k + 1]; aa[k] = aa[k + 1]; }
      ii[DR - 1] = i_new; aa[DR - 1] = a_new;
#pragma unroll
      for (int k = 0; k < DG; ++k) { Bq0[k] = Bq0[k + 1]; Bq1[k] = Bq1[k + 1]; }
    }
    if (j + 1 < n_end) {
      slice_of(j + 1, b, e_end);
      open_slice(b, e_end);
    }
    // every wave deposits its tiles (register r of tile T: row 4 q + r, column col); wave w sums tiles 2 w, 2 w + 1 in wave order
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * 8 + T) * 4 + r) * 64 + lane] = acc[T][r];
    // p1 of the finished channels: requested ahead of the barrier
    float2 p1 = make_float2(0.f, 0.f);
    if (live) p1 = *reinterpret_cast<const float2*>(phi + (size_t)j * 3 * F + F + cf);
    __syncthreads();
    float Nn[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < BWD_WAVES; ++w) sum += red[((w * 8 + 2 * wave + u) * 4 + r) * 64 + lane];
        Nn[u][r] = sum;
      }
    __syncthreads();                                   // (the deposits of the next node overwrite red)
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s0 = fmaf(Wl[0][r], Nn[0][r], s0);
      s1 = fmaf(Wl[1][r], Nn[1][r], s1);
      Gl[0][r] = fmaf(p1.x, Nn[0][r], Gl[0][r]);
      Gl[1][r] = fmaf(p1.y, Nn[1][r], Gl[1][r]);
    }
    s0 += __shfl_xor(s0, 16); s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
    if (live) {
      float* __restrict__ grow = g_phi + (size_t)j * 3 * F + cf;
      if (q == 0) *reinterpret_cast<float2*>(grow + F) = make_float2(s0, s1);
      else if (q == 1) *reinterpret_cast<float2*>(grow) = make_float2(0.f, 0.f);            // slices 0 and 2: no vector gradient
      else if (q == 2) *reinterpret_cast<float2*>(grow + 2 * F) = make_float2(0.f, 0.f);
    }
  }
  // block partial of the filter-weight gradient: part[chunk][n][F]
  if (live) {
    float* __restrict__ out = part + (size_t)chunk * (R + 1) * F + cf;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * q + r;
      if (n <= R) *reinterpret_cast<float2*>(out + (size_t)n * F) = make_float2(Gl[0][r], Gl[1][r]);
    }
  }
}

// second stage: gWd[c][n] = sum_chunk part[chunk][k][n][f]; c = kk*F + f over all 3 slices.
// K live slices: K == 3 -> kk = k; K == 1 -> only kk == 1 is live, the rest is written as 0.
// block = (64 channels, 16 chunk slices): slice s sums chunks s, s+16, ... and the 16 partial sums
// meet in LDS in a fixed order (deterministic).  grid = (ceil(F/64), R+1, 3).
constexpr int RED_SLICES = 16;
__global__ __launch_bounds__(64 * RED_SLICES) void equi_msg_bwd_reduce(const float* __restrict__ part, int n_chunks,
                                                                       int K, int R, int F, float* __restrict__ gWd,
                                                                       float* __restrict__ gbd) {
  __shared__ float red[RED_SLICES][64];
  const int f = blockIdx.x * 64 + threadIdx.x;
  const int s = threadIdx.y;
  const int n = blockIdx.y;        // 0..R  (R = bias)
  const int kk = blockIdx.z;       // 0..2
  const int k = (K == 3) ? kk : (kk == 1 ? 0 : -1);
  float acc = 0.f;
  if (k >= 0 && f < F) {
    const size_t stride = (size_t)K * (R + 1) * F;
    const float* p = part + ((size_t)k * (R + 1) + n) * F + f;
    constexpr int CB = 8;                            // chunks loaded together (clamped index, surplus zeroed; same order):
    for (int c0 = s; c0 < n_chunks; c0 += RED_SLICES * CB) {       // the plain loop was one memory round trip per chunk
      float v[CB];
#pragma unroll
      for (int u = 0; u < CB; ++u) v[u] = p[(size_t)min(c0 + RED_SLICES * u, n_chunks - 1) * stride];
#pragma unroll
      for (int u = 0; u < CB; ++u) acc += c0 + RED_SLICES * u < n_chunks ? v[u] : 0.f;
    }
  }
  red[s][threadIdx.x] = acc;
  __syncthreads();
  if (s != 0 || f >= F) return;
  float tot = 0.f;
#pragma unroll
  for (int q = 0; q < RED_SLICES; ++q) tot += red[q][threadIdx.x];
  const int c_out = kk * F + f;
  if (n < R) gWd[(size_t)c_out * R + n] = tot; else gbd[c_out] = tot;
}

// The same second stage for SEVERAL backward launches at once (blockIdx.z = 3 job + kk): the filter gradients are wanted
// by the optimiser only, so a training step defers the second stages of all its message blocks (7 on the chignolin
// config, one ~3 us link each in the backward chain) to one launch at the end of backward.
constexpr int FR_MAX = 16;
struct FilterReduceJob { const float* part; float* gWd; float* gbd; int n_chunks, K, R, F; };
struct FilterReduceJobs { int n, pad; FilterReduceJob job[FR_MAX]; };
__global__ __launch_bounds__(64 * RED_SLICES) void equi_msg_bwd_reduce_jobs_k(FilterReduceJobs J) {
  __shared__ float red[RED_SLICES][64];
  const int KZ = J.pad;            // z blocks per job: 3, or 9 when the table holds a nine-filter (EquiMessagePsuedo) job
  const int jz = blockIdx.z / KZ, kk = blockIdx.z - KZ * jz;
  const FilterReduceJob& q = J.job[jz];
  const int F = q.F, R = q.R, K = q.K, n_chunks = q.n_chunks;
  const int f = blockIdx.x * 64 + threadIdx.x;
  const int s = threadIdx.y;
  const int n = blockIdx.y;        // 0..R  (R = bias)
  if (n > R || kk >= (K == 9 ? 9 : 3)) return;               // (block-uniform)
  const int k = (K != 1) ? kk : (kk == 1 ? 0 : -1);
  float acc = 0.f;
  if (k >= 0 && f < F) {
    const size_t stride = (size_t)K * (R + 1) * F;
    const float* p = q.part + ((size_t)k * (R + 1) + n) * F + f;
    constexpr int CB = 8;
    for (int c0 = s; c0 < n_chunks; c0 += RED_SLICES * CB) {
      float v[CB];
#pragma unroll
      for (int u = 0; u < CB; ++u) v[u] = p[(size_t)min(c0 + RED_SLICES * u, n_chunks - 1) * stride];
#pragma unroll
      for (int u = 0; u < CB; ++u) acc += c0 + RED_SLICES * u < n_chunks ? v[u] : 0.f;
    }
  }
  red[s][threadIdx.x] = acc;
  __syncthreads();
  if (s != 0 || f >= F) return;
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < RED_SLICES; ++w) tot += red[w][threadIdx.x];
  const int c_out = kk * F + f;
  if (n < R) q.gWd[(size_t)c_out * R + n] = tot; else q.gbd[c_out] = tot;
}

constexpr int BWD_MAX_CHUNKS = 384;

struct BwdShape {
  bool split;
  int npc, chunks, cpx;
};
static inline BwdShape bwd_shape(int n_src, long long n_edges_hint) {
  BwdShape s;
  s.split = n_edges_hint >= 48LL * (n_src > 0 ? n_src : 1);
  if (const int o = cgv::option(CGV_OPT_MSG_BWD_SPLIT); o >= 0) s.split = o != 0;   // experiments only
  const int min_npc = s.split ? 1 : BWD_WAVES;
  int npc = (n_src + BWD_MAX_CHUNKS - 1) / BWD_MAX_CHUNKS;
  if (npc < min_npc) npc = min_npc;
  s.npc = npc;
  s.chunks = n_src > 0 ? (n_src + npc - 1) / npc : 1;
  s.cpx = (s.chunks + 7) / 8;
  return s;
}

}  // namespace cgv

extern "C" {

int cgv_equi_msg_fwd(const float* phi, const float* v, const float* geom_d, const int32_t* rowptr_d,
                     const int32_t* src_d, const float* Wd, const float* bd, float* ds, float* dv, int n_dst,
                     int n_feat, int n_rbf, int with_dv, int64_t n_edges_hint, int64_t n_rows_hint, const float* s_res,
                     const float* v_res, void* stream) {
  CGV_REQUIRE(n_dst >= 0 && n_feat > 0, "bad size");
  if (n_dst == 0) return 0;
  CGV_REQUIRE(phi && rowptr_d && Wd && bd && ds, "null pointer");
  CGV_REQUIRE(!with_dv || (v && dv), "with_dv needs v and dv");
  hipStream_t st = (hipStream_t)stream;
  const int tiles = (n_feat + 127) / 128;
  const int npx = (n_dst + 7) / 8;
  const dim3 grid(8 * npx * tiles);
  // >= 16 edges per receiver on average: 4 waves share a (node, tile) (the atom graph: 125; atom -> bead: 28).  Also on
  // tiny graphs with a few edges per receiver (the prior's 12 beads x 5): the launch is a chain of dependent round trips
  // per edge, and four waves walk a row of 5 in two trips instead of five (9.6 -> 6.4 us per launch)
  bool split = n_edges_hint >= 16LL * n_dst || (n_dst <= 64 && n_edges_hint >= 3LL * n_dst);
  if (const int o = cgv::option(CGV_OPT_MSG_FWD_SPLIT); o >= 0) split = o != 0;   // experiments only
  // 8-byte vector accesses need an even channel count and 8-byte aligned bases; the buffer-descriptor
  // gathers need every row within 2 GiB of the base (n_rows_hint = rows of phi / v, 0 = unknown)
  const bool pair = (n_feat % 2 == 0) && (n_rbf % 2 == 0) && n_rows_hint > 0 &&
                    (uint64_t)n_rows_hint * 12u * (uint64_t)n_feat < 0x7fffffffull &&
                    ((((uintptr_t)phi | (uintptr_t)v | (uintptr_t)ds | (uintptr_t)dv | (uintptr_t)s_res | (uintptr_t)v_res) & 7) == 0) &&
                    ((((uintptr_t)Wd) & 15) == 0) && ((((uintptr_t)bd) & 7) == 0);
#define CGV_FWD_LAUNCH(DV, SP, PR)                                                                                   \
  hipLaunchKernelGGL((cgv::equi_msg_fwd_k<RBF, DV, SP, PR>), grid, dim3(64 * SP), 0, st, phi, v, geom_d, rowptr_d, src_d, \
                     Wd, bd, ds, dv, n_feat, n_dst, npx, tiles, s_res, v_res)
#define CGV_FWD_PICK(DV)                                                 \
  if (pair) { if (split) CGV_FWD_LAUNCH(DV, 4, true); else CGV_FWD_LAUNCH(DV, 1, true); } \
  else      { if (split) CGV_FWD_LAUNCH(DV, 4, false); else CGV_FWD_LAUNCH(DV, 1, false); }
  // matrix-core variant: segments long enough to fill 16-edge tiles, rows addressable through a buffer
  // descriptor.  cgv_set_option(CGV_OPT_MSG_FWD_KERNEL, 0 | 1) overrides (A/B measurements only).
  // Measured (chignolin layer): 71 us against 55 us for the packed-VALU kernel -- with 173 VGPRs only two
  // waves fit a SIMD and the per-lane dword gathers (4 source rows per instruction) load the texture path
  // more than the VALU kernel's 8-byte row gathers; the kernel is gather-latency bound, not FMA bound.
  // Kept as an opt-in A/B variant: cgv_set_option(CGV_OPT_MSG_FWD_KERNEL, 1).
  const bool use_mfma = cgv::option(CGV_OPT_MSG_FWD_KERNEL) == 1 && n_rows_hint > 0 &&
                        (uint64_t)n_rows_hint * 12u * (uint64_t)n_feat < 0x7fffffffull;
  if (use_mfma) {
    const int tiles64 = (n_feat + 63) / 64;
    const dim3 grid64(8 * npx * tiles64);
    CGV_DISPATCH_RBF(n_rbf, {
      if (with_dv)
        hipLaunchKernelGGL((cgv::equi_msg_fwd_mfma_k<RBF, true, 4>), grid64, dim3(256), 0, st, phi, v, geom_d, rowptr_d, src_d,
                           Wd, bd, ds, dv, n_feat, n_dst, npx, tiles64, s_res, v_res);
      else
        hipLaunchKernelGGL((cgv::equi_msg_fwd_mfma_k<RBF, false, 4>), grid64, dim3(256), 0, st, phi, v, geom_d, rowptr_d,
                           src_d, Wd, bd, ds, dv, n_feat, n_dst, npx, tiles64, s_res, v_res);
    });
    return cgv::check_launch("cgv_equi_msg_fwd");
  }
  CGV_DISPATCH_RBF(n_rbf, {
    if (with_dv) { CGV_FWD_PICK(true) } else { CGV_FWD_PICK(false) }
  });
#undef CGV_FWD_PICK
#undef CGV_FWD_LAUNCH
  return cgv::check_launch("cgv_equi_msg_fwd");
}

size_t cgv_equi_msg_bwd_workspace_bytes(int n_src, int n_feat, int n_rbf) {
  (void)n_src;
  return sizeof(float) * (size_t)(cgv::BWD_MAX_CHUNKS + 8) * 3 * (n_rbf + 1) * n_feat + 256;
}

static int equi_msg_bwd_impl(const float* phi, const float* v, const float* geom_s, const int32_t* rowptr_s,
                             const int32_t* dst_s, const float* Wd, const float* bd, const float* gs, const float* gv,
                             float* g_phi, float* g_v, float* gWd, float* gbd, int n_src, int n_feat, int n_rbf,
                             int64_t n_edges_hint, int64_t n_rows_hint, void* workspace, size_t workspace_bytes, void* stream,
                             int* n_chunks_out, int* k_live_out);

int cgv_equi_msg_bwd(const float* phi, const float* v, const float* geom_s, const int32_t* rowptr_s,
                     const int32_t* dst_s, const float* Wd, const float* bd, const float* gs, const float* gv,
                     float* g_phi, float* g_v, float* gWd, float* gbd, int n_src, int n_feat, int n_rbf,
                     int64_t n_edges_hint, int64_t n_rows_hint, void* workspace, size_t workspace_bytes, void* stream) {
  CGV_REQUIRE(gWd && gbd, "null pointer");
  return equi_msg_bwd_impl(phi, v, geom_s, rowptr_s, dst_s, Wd, bd, gs, gv, g_phi, g_v, gWd, gbd, n_src, n_feat, n_rbf,
                           n_edges_hint, n_rows_hint, workspace, workspace_bytes, stream, nullptr, nullptr);
}

/* cgv_equi_msg_bwd without its second stage: g_phi / g_v are final, the filter gradients stay as per-chunk partial sums
 * in `workspace` (which must then live until cgv_filter_reduce_jobs has consumed it); *n_chunks / *k_live describe them. */
int cgv_equi_msg_bwd_deferred(const float* phi, const float* v, const float* geom_s, const int32_t* rowptr_s,
                              const int32_t* dst_s, const float* Wd, const float* bd, const float* gs, const float* gv,
                              float* g_phi, float* g_v, int n_src, int n_feat, int n_rbf, int64_t n_edges_hint,
                              int64_t n_rows_hint, void* workspace, size_t workspace_bytes, int* n_chunks, int* k_live,
                              void* stream) {
  CGV_REQUIRE(n_chunks && k_live, "null pointer");
  return equi_msg_bwd_impl(phi, v, geom_s, rowptr_s, dst_s, Wd, bd, gs, gv, g_phi, g_v, nullptr, nullptr, n_src, n_feat, n_rbf,
                           n_edges_hint, n_rows_hint, workspace, workspace_bytes, stream, n_chunks, k_live);
}

int cgv_filter_reduce_jobs_max(void) { return cgv::FR_MAX; }
int cgv_filter_reduce_job_bytes(void) { return (int)sizeof(cgv::FilterReduceJob); }

/* The deferred second stages of up to cgv_filter_reduce_jobs_max() message-block backward launches in one launch.
 * jobs_host: n records laid out as cgv::FilterReduceJob {part, gWd, gbd, n_chunks, K (live slices: 1 or 3; 9: the partial
 * sums of cgv_pseudo_msg_bwd_deferred), R, F}. */
int cgv_filter_reduce_jobs(const void* jobs_host, int n_jobs, void* stream) {
  CGV_REQUIRE(jobs_host && n_jobs >= 1 && n_jobs <= cgv::FR_MAX, "bad job table");
  cgv::FilterReduceJobs J;
  std::memset(static_cast<void*>(&J), 0, sizeof(J));
  std::memcpy(static_cast<void*>(J.job), jobs_host, sizeof(cgv::FilterReduceJob) * (size_t)n_jobs);
  J.n = n_jobs;
  int fmax = 0, rmax = 0, kz = 3;
  for (int j = 0; j < n_jobs; ++j) {
    const cgv::FilterReduceJob& q = J.job[j];
    CGV_REQUIRE(q.part && q.gWd && q.gbd && q.n_chunks >= 1 && (q.K == 1 || q.K == 3 || q.K == 9) && q.R >= 1 && q.F >= 1, "bad job");
    fmax = q.F > fmax ? q.F : fmax;
    rmax = q.R > rmax ? q.R : rmax;
    if (q.K == 9) kz = 9;
  }
  J.pad = kz;
  dim3 rgrid((fmax + 63) / 64, rmax + 1, kz * n_jobs);
  hipLaunchKernelGGL(cgv::equi_msg_bwd_reduce_jobs_k, rgrid, dim3(64, cgv::RED_SLICES), 0, (hipStream_t)stream, J);
  return cgv::check_launch("cgv_filter_reduce_jobs");
}

static int equi_msg_bwd_impl(const float* phi, const float* v, const float* geom_s, const int32_t* rowptr_s,
                             const int32_t* dst_s, const float* Wd, const float* bd, const float* gs, const float* gv,
                             float* g_phi, float* g_v, float* gWd, float* gbd, int n_src, int n_feat, int n_rbf,
                             int64_t n_edges_hint, int64_t n_rows_hint, void* workspace, size_t workspace_bytes, void* stream,
                             int* n_chunks_out, int* k_live_out) {
  CGV_REQUIRE(n_src >= 0 && n_feat > 0, "bad size");
  CGV_REQUIRE(phi && rowptr_s && Wd && bd && g_phi && workspace, "null pointer");
  CGV_REQUIRE(!gv || (v && g_v), "gv needs v and g_v");
  if (workspace_bytes < cgv_equi_msg_bwd_workspace_bytes(n_src, n_feat, n_rbf)) {
    cgv::set_error("cgv_equi_msg_bwd: workspace too small");
    return CGV_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const cgv::BwdShape sh = cgv::bwd_shape(n_src, n_edges_hint);
  const int tiles = (n_feat + 127) / 128;
  float* part = reinterpret_cast<float*>(workspace);
  const dim3 grid(8 * sh.cpx * tiles), block(64 * cgv::BWD_WAVES);
  const bool pair = (n_feat % 2 == 0) && (n_rbf % 2 == 0) && n_rows_hint > 0 &&
                    (uint64_t)n_rows_hint * 12u * (uint64_t)n_feat < 0x7fffffffull &&
                    ((((uintptr_t)phi | (uintptr_t)v | (uintptr_t)gs | (uintptr_t)gv | (uintptr_t)g_phi |
                       (uintptr_t)g_v | (uintptr_t)part) & 7) == 0) &&
                    ((((uintptr_t)Wd) & 15) == 0) && ((((uintptr_t)bd) & 7) == 0);
  // scalar-only upstream on a high-degree graph: the matrix-core kernel.  From ~200 edges per node on (the 2000-atom graph:
  // 425, 390 -> 265 us per launch); at chignolin's 123 -- 8 batches of 4 edges per wave and node -- the exchange of the
  // tiles at the end of every node outweighs it (27.0 against 24.0 us).  cgv_set_option(CGV_OPT_MSG_BWD_MFMA, 0 | 1): never /
  // wherever it applies (A/B runs, tests).
  const int mfma_opt = cgv::option(CGV_OPT_MSG_BWD_MFMA);
  // (and on graphs of at most two blocks per CU -- the 64 beads of the 2000-atom config, 61 edges each: 7.5 against 10.4 us,
  // the launch is a chain of round trips there and the matrix-core kernel has fewer)
  const bool mfma_pays = mfma_opt == 1 || (mfma_opt < 0 && (n_edges_hint >= 200LL * (n_src > 0 ? n_src : 1) || 8 * sh.cpx * tiles <= 512));
  if (!gv && gs && sh.split && n_rbf + 1 <= 16 && (n_feat % 4) == 0 && mfma_pays &&
      ((((uintptr_t)gs) & 15) == 0) && ((((uintptr_t)phi | (uintptr_t)g_phi | (uintptr_t)part) & 7) == 0)) {
    CGV_DISPATCH_RBF(n_rbf, {
      hipLaunchKernelGGL((cgv::equi_msg_bwd_mfma_k<RBF>), grid, block, 0, st, phi, geom_s, rowptr_s, dst_s, Wd, bd, gs, g_phi, part,
                         n_feat, n_src, sh.npc, sh.cpx, tiles);
    });
  } else {
#define CGV_BWD_LAUNCH(GV, SP, PR)                                                                                 \
  hipLaunchKernelGGL((cgv::equi_msg_bwd_k<RBF, GV, SP, PR>), grid, block, 0, st, phi, v, geom_s, rowptr_s, dst_s, Wd, \
                     bd, gs, gv, g_phi, g_v, part, n_feat, n_src, sh.npc, sh.cpx, tiles)
#define CGV_BWD_PICK(GV)                                                                   \
  if (pair) { if (sh.split) CGV_BWD_LAUNCH(GV, true, true); else CGV_BWD_LAUNCH(GV, false, true); } \
  else      { if (sh.split) CGV_BWD_LAUNCH(GV, true, false); else CGV_BWD_LAUNCH(GV, false, false); }
  CGV_DISPATCH_RBF(n_rbf, {
    if (gv) { CGV_BWD_PICK(true) } else { CGV_BWD_PICK(false) }
  });
#undef CGV_BWD_PICK
#undef CGV_BWD_LAUNCH
  }
  // chunks beyond sh.chunks (padding to a multiple of 8) still write zero partials: sum them all
  if (n_chunks_out) {                    // deferred second stage (cgv_filter_reduce_jobs)
    *n_chunks_out = 8 * sh.cpx;
    *k_live_out = gv ? 3 : 1;
    return cgv::check_launch("cgv_equi_msg_bwd_deferred");
  }
  dim3 rgrid((n_feat + 63) / 64, n_rbf + 1, 3);
  hipLaunchKernelGGL(cgv::equi_msg_bwd_reduce, rgrid, dim3(64, cgv::RED_SLICES), 0, st, part, 8 * sh.cpx, gv ? 3 : 1, n_rbf,
                     n_feat, gWd, gbd);
  return cgv::check_launch("cgv_equi_msg_bwd");
}

}  // extern "C"
