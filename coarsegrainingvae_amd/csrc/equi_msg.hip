// K2 / K4: fused EquiMessageBlock and ContractiveMessageBlock (forward + backward).
//
// Reference math (CoarseGrainingVAE/conv.py:505-563, InvariantMessage 63-75, DistanceEmbed
// modules.py:192-197; contraction variant conv.py:703-733), per directed edge e = (i <- j):
//     w_k(e,f) = sum_n Wd[kF+f][n] * a_n(e) + bd[kF+f] * env(e)            k = 0,1,2
//     m_k(e,f) = phi[j, kF+f] * w_k(e,f)
//     ds[i,f]   = sum_e m_1                dv[i,f,:] = sum_e ( m_2 * unit_e + m_0 * v[j,f,:] )
// The reference materialises ~10 tensors of shape [E,3F] / [E,F,3] per layer and reduces them
// with an unsorted scatter_add.  Here one wave owns (destination node, 64 channels): it walks
// the node's CSR segment, gathers phi[j]/v[j] rows coalesced along channels (L2-resident),
// rebuilds the filter from the per-edge geometry record held in SGPRs (the record address is
// wave-uniform -> scalar loads), and accumulates in VGPRs.  Nothing of size E*F is written.
//
// Mapping: thread <-> channel f, so every global access is a contiguous 256 B (phi) or 768 B
// (v, dwordx3) wave transaction; the 3*(R+1) filter weights of the channel live in registers
// for the whole segment.
#include "cgv_common.h"

namespace cgv {

template <int R>
__device__ __forceinline__ float filter(const float (&W)[R + 1], const float* __restrict__ g) {
  float w = W[R] * g[R];
#pragma unroll
  for (int n = 0; n < R; ++n) w = fmaf(W[n], g[n], w);
  return w;
}

template <int R>
__device__ __forceinline__ void load_filter_row(float (&W)[R + 1], const float* __restrict__ Wd,
                                                const float* __restrict__ bd, int c) {
#pragma unroll
  for (int n = 0; n < R; ++n) W[n] = Wd[(size_t)c * R + n];
  W[R] = bd[c];
}

// ------------------------------------------------------------------ forward
// grid = (n_dst, ceil(F / BLOCK)), block = BLOCK threads (BLOCK/64 waves, each its own 64 channels)
template <int R, bool WITH_DV, int BLOCK>
__global__ __launch_bounds__(BLOCK) void equi_msg_fwd_k(const float* __restrict__ phi, const float* __restrict__ v,
                                                        const float* __restrict__ geom,
                                                        const int* __restrict__ rowptr, const int* __restrict__ src,
                                                        const float* __restrict__ Wd, const float* __restrict__ bd,
                                                        float* __restrict__ ds, float* __restrict__ dv, int F) {
  constexpr int GS = (R + 4 + 3) & ~3;
  const int node = blockIdx.x;
  const int f_raw = blockIdx.y * BLOCK + threadIdx.x;
  const bool live = f_raw < F;
  const int f = live ? f_raw : F - 1;     // clamp: idle lanes load valid addresses, never store

  float W0[R + 1], W1[R + 1], W2[R + 1];
  load_filter_row<R>(W1, Wd, bd, F + f);
  if constexpr (WITH_DV) {
    load_filter_row<R>(W0, Wd, bd, f);
    load_filter_row<R>(W2, Wd, bd, 2 * F + f);
  }

  float acc_s = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
  const int beg = rowptr[node], end = rowptr[node + 1];
#pragma unroll 2
  for (int e = beg; e < end; ++e) {
    const float* __restrict__ g = geom + (size_t)e * GS;   // wave-uniform -> s_load
    const int j = src[e];
    const float* __restrict__ prow = phi + (size_t)j * 3 * F;
    const float p1 = prow[F + f];
    acc_s = fmaf(p1, filter<R>(W1, g), acc_s);
    if constexpr (WITH_DV) {
      const float p0 = prow[f];
      const float p2 = prow[2 * F + f];
      const f3 vj = ld3(v + ((size_t)j * F + f) * 3);
      const float m0 = p0 * filter<R>(W0, g);
      const float m2 = p2 * filter<R>(W2, g);
      ax = fmaf(m2, g[R + 1], fmaf(m0, vj.x, ax));
      ay = fmaf(m2, g[R + 2], fmaf(m0, vj.y, ay));
      az = fmaf(m2, g[R + 3], fmaf(m0, vj.z, az));
    }
  }
  if (live) {
    ds[(size_t)node * F + f] = acc_s;
    if (WITH_DV) st3(dv + ((size_t)node * F + f) * 3, ax, ay, az);
  }
}

// ------------------------------------------------------------------ backward
// Upstream gs[i,f], gv[i,f,:] at the receivers.  With gq_1 = gs, gq_2 = gv.unit, gq_0 = gv.v_j:
//     g_phi[j,kF+f] = sum_{e: src(e)=j} gq_k * w_k          g_v[j,f,:] = sum_e m_0 * gv_i
//     gWd[kF+f][n]  = sum_e gq_k * phi[j,kF+f] * a_n(e)      gbd[kF+f]  = sum_e gq_k * phi * env
// One wave owns (chunk of source nodes, 64 channels) and walks the SRC-sorted view, so g_phi
// and g_v are plain stores; gWd/gbd accumulate in registers over the whole chunk and leave as
// one partial per block (LDS reduction over the block's waves), summed by a second kernel in a
// fixed order -> deterministic, no atomics.
// grid = (n_chunks, ceil(F/64)), block = 64*WAVES; wave w of chunk c takes nodes c*npc + w, +WAVES, ...
template <int R, bool HAS_GV, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void equi_msg_bwd_k(
    const float* __restrict__ phi, const float* __restrict__ v, const float* __restrict__ geom,
    const int* __restrict__ rowptr, const int* __restrict__ dst, const float* __restrict__ Wd,
    const float* __restrict__ bd, const float* __restrict__ gs, const float* __restrict__ gv,
    float* __restrict__ g_phi, float* __restrict__ g_v, float* __restrict__ part, int F, int n_src,
    int nodes_per_chunk) {
  constexpr int GS = (R + 4 + 3) & ~3;
  constexpr int K = HAS_GV ? 3 : 1;           // live filter slices (k = 1 only without gv)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f_raw = blockIdx.y * 64 + lane;
  const bool live = f_raw < F;
  const int f = live ? f_raw : F - 1;

  float W[K][R + 1], G[K][R + 1];
  if constexpr (HAS_GV) {
    load_filter_row<R>(W[0], Wd, bd, f);
    load_filter_row<R>(W[1], Wd, bd, F + f);
    load_filter_row<R>(W[2], Wd, bd, 2 * F + f);
  } else {
    load_filter_row<R>(W[0], Wd, bd, F + f);
  }
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int n = 0; n <= R; ++n) G[k][n] = 0.f;

  const int n_beg = blockIdx.x * nodes_per_chunk;
  const int n_end = min(n_beg + nodes_per_chunk, n_src);
  for (int j = n_beg + wave; j < n_end; j += WAVES) {
    const float* __restrict__ prow = phi + (size_t)j * 3 * F;
    const float p1 = prow[F + f];
    float p0 = 0.f, p2 = 0.f;
    f3 vj{0.f, 0.f, 0.f};
    if constexpr (HAS_GV) {
      p0 = prow[f];
      p2 = prow[2 * F + f];
      vj = ld3(v + ((size_t)j * F + f) * 3);
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
    const int beg = rowptr[j], end = rowptr[j + 1];
#pragma unroll 2
    for (int e = beg; e < end; ++e) {
      const float* __restrict__ g = geom + (size_t)e * GS;
      const int i = dst[e];
      const float gq1 = gs ? gs[(size_t)i * F + f] : 0.f;
      if constexpr (HAS_GV) {
        const f3 gvi = ld3(gv + ((size_t)i * F + f) * 3);
        const float w0 = filter<R>(W[0], g), w1 = filter<R>(W[1], g), w2 = filter<R>(W[2], g);
        const float gq0 = gvi.x * vj.x + gvi.y * vj.y + gvi.z * vj.z;
        const float gq2 = gvi.x * g[R + 1] + gvi.y * g[R + 2] + gvi.z * g[R + 3];
        a0 = fmaf(gq0, w0, a0);
        a1 = fmaf(gq1, w1, a1);
        a2 = fmaf(gq2, w2, a2);
        const float m0 = p0 * w0;
        bx = fmaf(m0, gvi.x, bx);
        by = fmaf(m0, gvi.y, by);
        bz = fmaf(m0, gvi.z, bz);
        const float t0 = gq0 * p0, t1 = gq1 * p1, t2 = gq2 * p2;
#pragma unroll
        for (int n = 0; n <= R; ++n) {
          G[0][n] = fmaf(t0, g[n], G[0][n]);
          G[1][n] = fmaf(t1, g[n], G[1][n]);
          G[2][n] = fmaf(t2, g[n], G[2][n]);
        }
      } else {
        a1 = fmaf(gq1, filter<R>(W[0], g), a1);
        const float t1 = gq1 * p1;
#pragma unroll
        for (int n = 0; n <= R; ++n) G[0][n] = fmaf(t1, g[n], G[0][n]);
      }
    }
    if (live) {
      float* __restrict__ grow = g_phi + (size_t)j * 3 * F;
      grow[f] = a0;
      grow[F + f] = a1;
      grow[2 * F + f] = a2;
      if constexpr (HAS_GV) st3(g_v + ((size_t)j * F + f) * 3, bx, by, bz);
    }
  }

  // block partial of the filter-weight gradient: part[chunk][k][n][F]  (channel fastest -> coalesced)
  __shared__ float red[WAVES > 1 ? (WAVES - 1) * K * (R + 1) * 64 : 1];
  if (WAVES > 1) {
    if (wave > 0) {
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int n = 0; n <= R; ++n) red[(((wave - 1) * K + k) * (R + 1) + n) * 64 + lane] = G[k][n];
    }
    __syncthreads();
    if (wave == 0) {
      for (int w = 0; w < WAVES - 1; ++w)
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int n = 0; n <= R; ++n) G[k][n] += red[((w * K + k) * (R + 1) + n) * 64 + lane];
    }
  }
  if (wave == 0 && live) {
    float* __restrict__ out = part + (size_t)blockIdx.x * K * (R + 1) * F;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int n = 0; n <= R; ++n) out[((size_t)k * (R + 1) + n) * F + f] = G[k][n];
  }
}

// second stage: gWd[c][n] = sum_chunk part[chunk][k][n][f]; c = kk*F + f over all 3 slices.
// K live slices: K == 3 -> kk = k; K == 1 -> only kk == 1 is live, the rest is written as 0.
__global__ __launch_bounds__(256) void equi_msg_bwd_reduce(const float* __restrict__ part, int n_chunks, int K, int R,
                                                           int F, float* __restrict__ gWd, float* __restrict__ gbd) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;        // 0..R  (R = bias)
  const int kk = blockIdx.z;       // 0..2
  if (f >= F) return;
  float acc = 0.f;
  const int k = (K == 3) ? kk : (kk == 1 ? 0 : -1);
  if (k >= 0) {
    const size_t stride = (size_t)K * (R + 1) * F;
    const float* p = part + ((size_t)k * (R + 1) + n) * F + f;
    for (int c = 0; c < n_chunks; ++c) acc += p[c * stride];
  }
  const int c_out = kk * F + f;
  if (n < R) gWd[(size_t)c_out * R + n] = acc; else gbd[c_out] = acc;
}

static inline int bwd_chunks(int n_src) {
  // enough chunks to fill the chip, few enough that the partial buffer stays small
  int c = n_src < 96 ? n_src : 96;
  return c > 0 ? c : 1;
}
constexpr int BWD_WAVES = 4;

}  // namespace cgv

extern "C" {

int cgv_equi_msg_fwd(const float* phi, const float* v, const float* geom_d, const int32_t* rowptr_d,
                     const int32_t* src_d, const float* Wd, const float* bd, float* ds, float* dv, int n_dst,
                     int n_feat, int n_rbf, int with_dv, void* stream) {
  CGV_REQUIRE(n_dst >= 0 && n_feat > 0, "bad size");
  if (n_dst == 0) return 0;
  CGV_REQUIRE(phi && rowptr_d && Wd && bd && ds, "null pointer");
  CGV_REQUIRE(!with_dv || (v && dv), "with_dv needs v and dv");
  constexpr int BLOCK = 64;
  dim3 grid(n_dst, (n_feat + BLOCK - 1) / BLOCK), block(BLOCK);
  hipStream_t st = (hipStream_t)stream;
  CGV_DISPATCH_RBF(n_rbf, {
    if (with_dv)
      hipLaunchKernelGGL((cgv::equi_msg_fwd_k<RBF, true, BLOCK>), grid, block, 0, st, phi, v, geom_d, rowptr_d, src_d,
                         Wd, bd, ds, dv, n_feat);
    else
      hipLaunchKernelGGL((cgv::equi_msg_fwd_k<RBF, false, BLOCK>), grid, block, 0, st, phi, v, geom_d, rowptr_d, src_d,
                         Wd, bd, ds, dv, n_feat);
  });
  return cgv::check_launch("cgv_equi_msg_fwd");
}

size_t cgv_equi_msg_bwd_workspace_bytes(int n_src, int n_feat, int n_rbf) {
  return sizeof(float) * (size_t)cgv::bwd_chunks(n_src) * 3 * (n_rbf + 1) * n_feat + 256;
}

int cgv_equi_msg_bwd(const float* phi, const float* v, const float* geom_s, const int32_t* rowptr_s,
                     const int32_t* dst_s, const float* Wd, const float* bd, const float* gs, const float* gv,
                     float* g_phi, float* g_v, float* gWd, float* gbd, int n_src, int n_feat, int n_rbf,
                     void* workspace, size_t workspace_bytes, void* stream) {
  CGV_REQUIRE(n_src >= 0 && n_feat > 0, "bad size");
  CGV_REQUIRE(phi && rowptr_s && Wd && bd && g_phi && gWd && gbd && workspace, "null pointer");
  CGV_REQUIRE(!gv || (v && g_v), "gv needs v and g_v");
  if (workspace_bytes < cgv_equi_msg_bwd_workspace_bytes(n_src, n_feat, n_rbf)) {
    cgv::set_error("cgv_equi_msg_bwd: workspace too small");
    return CGV_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int chunks = cgv::bwd_chunks(n_src);
  const int npc = n_src > 0 ? (n_src + chunks - 1) / chunks : 1;
  float* part = reinterpret_cast<float*>(workspace);
  constexpr int WV = cgv::BWD_WAVES;
  dim3 grid(chunks, (n_feat + 63) / 64), block(64 * WV);
  CGV_DISPATCH_RBF(n_rbf, {
    if (gv)
      hipLaunchKernelGGL((cgv::equi_msg_bwd_k<RBF, true, WV>), grid, block, 0, st, phi, v, geom_s, rowptr_s, dst_s, Wd,
                         bd, gs, gv, g_phi, g_v, part, n_feat, n_src, npc);
    else
      hipLaunchKernelGGL((cgv::equi_msg_bwd_k<RBF, false, WV>), grid, block, 0, st, phi, v, geom_s, rowptr_s, dst_s, Wd,
                         bd, gs, gv, g_phi, g_v, part, n_feat, n_src, npc);
  });
  dim3 rgrid((n_feat + 255) / 256, n_rbf + 1, 3);
  hipLaunchKernelGGL(cgv::equi_msg_bwd_reduce, rgrid, dim3(256), 0, st, part, chunks, gv ? 3 : 1, n_rbf, n_feat, gWd, gbd);
  return cgv::check_launch("cgv_equi_msg_bwd");
}

}  // extern "C"
