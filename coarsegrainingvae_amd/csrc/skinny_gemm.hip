// Skinny fp32 GEMMs for the node-level Dense layers on the bead graph (M = beads or 3*beads
// rows, 12..64; K, N = n_basis multiples, 600..5400).
//
// Reference: Dense / nn.Linear forward (CoarseGrainingVAE/modules.py:103-114, activation Swish
// modules.py:16-21) and its autograd backward, ~70 layer-uses per training step of the chignolin
// config.  With M <= 64 these products are weight-streaming (GEMV-like): the [N,K] weight
// (1.4 - 13 MB) is read once, the arithmetic is negligible, and what a step pays is mostly the
// NUMBER of launches (measured: ~4-5 us of GPU time per small kernel inside the captured step
// graph, profiles/r01d).  So the design goal here is launches, then bytes:
//   fwd        z = x W^T + b ; y = act(z)          ONE launch (bias + Swish in the epilogue)
//   bwd_input  gx = (gy * act'(z)) W               ONE launch (activation backward in the operand
//                                                   load, N-split over 16 waves reduced in LDS --
//                                                   no partial buffers, no second kernel)
//   wgrad      gW (+)= (gy * act'(z))^T x, gb      ONE launch PER STEP for all layers: a grouped
//                                                   kernel over a device table of problems, queued
//                                                   during backward and flushed before the optimiser
// Matrix products use v_mfma_f32_16x16x4_f32: exact fp32 FMA chains (bitwise an fmaf loop).
//
// MFMA 16x16x4 f32 operand map (cdna_hip_programming.md 3): lane l holds A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15]; D: col j = l&15, row i = 4*(l>>4) + reg.
#include <cstdlib>
#include <cstring>
#include "cgv_common.h"

namespace cgv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ldg4_or_zero(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float4 ldg4g_or_zero(const float* p, bool ok);   // the same through address space 1 (below)
// Pointers that come out of a record table are generic to the compiler: loads and stores through them are flat_*, which
// count on lgkmcnt as well as vmcnt -- an LDS wait then also waits for them.  These go through address space 1 (global_*).
typedef float gf32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4_global(const float* p) {
  const gf32x4 t = *reinterpret_cast<const __attribute__((address_space(1))) gf32x4*>((const __attribute__((address_space(1))) float*)p);
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ float ldg_global(const float* p) {
  return *((const __attribute__((address_space(1))) float*)p);
}
__device__ __forceinline__ float4 ldg4g_or_zero(const float* p, bool ok) {
  return ok ? ldg4_global(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void stg4_global(float* p, const float4& v) {
  *reinterpret_cast<__attribute__((address_space(1))) gf32x4*>((__attribute__((address_space(1))) float*)p) = gf32x4{v.x, v.y, v.z, v.w};
}
// ------------------------------------------------------------------ fwd
// Block = 16 output columns n0..n0+15, WAVES waves splitting K in whole 16-float steps.  Lane
// (i = l&15, q = l>>4) loads W[n0+i, 16 s + 4 q .. +3] (16 rows x 64 contiguous bytes per wave
// instruction) and the matching x[m, ...] float4; component c feeds MFMA c (k = 16 s + 4 q + c).
// D: lane holds y[m = 16 mb + (l&15)][n0 + 4 q + r], r = 0..3 -> one 16-byte store per m-block.
template <int MB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void skinny_fwd_k(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           float* __restrict__ zout, int M, int N, int K, int act) {
  __shared__ float red[(WAVES - 1) * MB * 4 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16;
  {                                     // blockIdx.y: this block's 16 MB rows (more, smaller blocks when there are few column blocks)
    const int m0 = blockIdx.y * 16 * MB;
    x += (size_t)m0 * K; y += (size_t)m0 * N;
    if (zout) zout += (size_t)m0 * N;
    M = min(M - m0, 16 * MB);
  }
  const int nrow = n0 + i;
  const bool nok = nrow < N;
  const float* wrow = W + (size_t)(nok ? nrow : 0) * K;
  const int steps = (K + 15) / 16;
  const int per = (steps + WAVES - 1) / WAVES;
  const int s_beg = wave * per, s_end = min(s_beg + per, steps);
  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Loads are unconditional (rows / columns clamped into range): rows beyond M or N give products that are never
  // stored, and the reduction tail is zeroed on W's side only.  A guarded load is a branch: the loop then neither
  // unrolls nor keeps more than one step's loads in flight (s_waitcnt vmcnt(0) per step), and a wave's 2 - 3 steps
  // each paid a full memory round trip.
  const float* xrow[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) xrow[mb] = x + (size_t)(mb * 16 + i < M ? mb * 16 + i : 0) * K;
  constexpr int SB = MB >= 3 ? 2 : 4;             // steps whose loads are issued together (the compiler does not
  for (int s0 = s_beg; s0 < s_end; s0 += SB) {    // unroll this loop by itself: runtime bounds)
    float4 a[SB], b[SB][MB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int k = (s0 + u) * 16 + 4 * q;
      const bool kok = s0 + u < s_end && k < K;   // K % 4 == 0: a float4 is entirely in or out
      const int kc = kok ? k : 0;
      a[u] = *reinterpret_cast<const float4*>(wrow + kc);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) b[u][mb] = *reinterpret_cast<const float4*>(xrow[mb] + kc);
    }
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      if (s0 + u < s_end) {                       // wave-uniform
        const bool kok = (s0 + u) * 16 + 4 * q < K;
        const float4 w = make_float4(kok ? a[u].x : 0.f, kok ? a[u].y : 0.f, kok ? a[u].z : 0.f, kok ? a[u].w : 0.f);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, b[u][mb].x, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, b[u][mb].y, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, b[u][mb].z, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, b[u][mb].w, acc[mb], 0, 0, 0);
        }
      }
    }
  }
  if (WAVES > 1) {
    if (wave > 0) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(((wave - 1) * MB + mb) * 4 + r) * 64 + lane] = acc[mb][r];
    }
    __syncthreads();
    if (wave != 0) return;
  }
  const int n = n0 + 4 * q;
  if (n >= N) return;                              // N % 4 == 0
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bv = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mb * 16 + i;
    f32x4 t = acc[mb];
    for (int w = 0; w < WAVES - 1; ++w)
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] += red[((w * MB + mb) * 4 + r) * 64 + lane];
    if (m < M) {
      const float4 zv = make_float4(t[0] + bv.x, t[1] + bv.y, t[2] + bv.z, t[3] + bv.w);
      if (act) {
        if (zout) *reinterpret_cast<float4*>(zout + (size_t)m * N + n) = zv;
        *reinterpret_cast<float4*>(y + (size_t)m * N + n) =
            make_float4(act_fwd(zv.x, act), act_fwd(zv.y, act), act_fwd(zv.z, act), act_fwd(zv.w, act));
      } else {
        *reinterpret_cast<float4*>(y + (size_t)m * N + n) = zv;
      }
    }
  }
}

// ------------------------------------------------------------------ bwd_input
// gx[m, k] = sum_n g[m, n] W[n, k],  g = gy * act'(z).  The weight is read in ROW-contiguous
// pieces: block (kt, ns) owns 64 columns k0..k0+63 (256 contiguous bytes per weight row) and the
// row slice [ns*rpb, ns*rpb + rpb); wave w of 4 takes rows 4w..4w+3 of every group of 16.
// Lane (j = l&15, q = l>>4) loads the float4 W[n + q, k0 + 4j .. +3]; component c is the B operand
// B[kk = q][col j] of MFMA c, A[i = j][kk = q] = g[m = 16 mb + j][n + q] comes from an LDS stage
// of g (coalesced row reads, activation derivative applied once).  D_c: lane holds
// gx[m = 16 mb + 4 q + r][k0 + 4 j + c] -> the 4 MFMAs give one 16-byte store per r.
// The first version gave each block a 16-column strip over ALL rows: 38 blocks, each touching
// every page of the weight 64 bytes at a time -- 11-22 us per call (profiles/r01l), 48 calls a
// step.  Here ~300 blocks each read a compact [rpb x 256 B] piece with every load in flight at
// once; the row slices meet in a second, tiny launch that sums the ns partials in index order
// (deterministic).  A single-launch "last block reduces" variant behind an agent-scope acq_rel counter
// was measured at 30-85 us: every __threadfence writes back / invalidates the XCD's whole L2.
constexpr int BI_COLS = 64;       // weight columns per block
constexpr int BI_ROUND = 64;      // weight rows per staging round (4 waves x 4 steps x 4 rows)
constexpr int BI_LD = 68;         // LDS row stride of the g stage

// LAZY: gy arrives as row-slice partial sums of the product that produced it (SliceSum: no reduction launch in
// between); the kt = 0 blocks also write the summed g[:, their rows] to g_dense (the weight-gradient launch reads it).
template <int MB, bool LAZY = false>
__device__ __forceinline__ void skinny_bwd_input_body(const int bx /* block index within the problem */,
                                                      const float* __restrict__ gy, const float* __restrict__ z,
                                                      const float* __restrict__ W, float* __restrict__ gx,
                                                      float* __restrict__ part, int M, int N, int K, int act, int KT,
                                                      int NS, int rpb, SliceSum gsum = SliceSum{nullptr, nullptr, 0, 0},
                                                      float* __restrict__ g_dense = nullptr) {
  // g stage [16 MB rows][BI_LD], then the wave reduction: up to 4 row blocks all three partner waves deposit at once,
  // beyond that (MB 5..8: 65-128 rows) one wave at a time through a third of the space
  constexpr int SM_RED = (MB <= 4 ? 4 : 1) * MB * 16 * 64, SM_G = MB * 16 * BI_LD;
  __shared__ __attribute__((aligned(16))) float sm[SM_RED > SM_G ? SM_RED : SM_G];
  const int kt = bx % KT, ns = bx / KT;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int kcol = kt * BI_COLS + 4 * j;
  const bool kok = kcol < K;                                               // K % 4 == 0
  const int n_beg = ns * rpb, n_end = min(n_beg + rpb, N);
  f32x4 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[mb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Up to four rounds (a row slice of <= 256 weight rows: every split product of the model), plain aligned operands, no
  // activation of its own (the usual case since Swish' travels downstream): EVERY round's weights and g tile are requested
  // up front -- a block is alone on its CU and each round used to wait out its own round trip (~2 us of W from HBM in
  // front of 0.45 us of MFMAs: 96 x 5400 -> 600 took 20.5 us) -- then the rounds only stage, meet and multiply.
  constexpr int RMAX = 4;
  const int rounds = (n_end - n_beg + BI_ROUND - 1) / BI_ROUND;
  bool upfront = false;
  if constexpr (!LAZY) upfront = act == 0 && rounds <= RMAX && (((uintptr_t)gy) & 15) == 0;
  if (upfront) {
    float4 wA[RMAX][4], gA[RMAX][MB];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const int nb = n_beg + min(r, rounds - 1) * BI_ROUND;              // (surplus rounds repeat the last one's addresses; never used)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int row = nb + 16 * s + 4 * wave + q;
        const bool ok = row < n_end && kok;
        wA[r][s] = *reinterpret_cast<const float4*>(W + (size_t)(ok ? row : 0) * K + (ok ? kcol : 0));
      }
#pragma unroll
      for (int u = 0; u < MB; ++u) {
        const int unit = u * 256 + (int)threadIdx.x, m = unit >> 4, n = nb + 4 * (unit & 15);
        const bool ok = m < M && n < n_end;
        gA[r][u] = *reinterpret_cast<const float4*>(gy + (ok ? (size_t)m * N + n : 0));
      }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      if (r >= rounds) break;                                              // block-uniform
      const int nb = n_beg + r * BI_ROUND;
      if (r) __syncthreads();                                              // readers of the previous round
#pragma unroll
      for (int u = 0; u < MB; ++u) {
        const int unit = u * 256 + (int)threadIdx.x, m = unit >> 4, n = nb + 4 * (unit & 15);
        *reinterpret_cast<float4*>(sm + m * BI_LD + 4 * (unit & 15)) = (m < M && n < n_end) ? gA[r][u] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool ok = nb + 16 * s + 4 * wave + q < n_end && kok;
        const float4 w4 = ok ? wA[r][s] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const float a = sm[(mb * 16 + j) * BI_LD + 16 * s + 4 * wave + q];
          acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w4.x, acc[mb][0], 0, 0, 0);
          acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w4.y, acc[mb][1], 0, 0, 0);
          acc[mb][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w4.z, acc[mb][2], 0, 0, 0);
          acc[mb][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w4.w, acc[mb][3], 0, 0, 0);
        }
      }
    }
  }
  for (int nb = upfront ? n_end : n_beg; nb < n_end; nb += BI_ROUND) {
    float4 wv[4];                                                          // weights first: the long latency
    bool wok[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {                                          // unconditional (clamped) requests, zeroed below:
      const int row = nb + 16 * s + 4 * wave + q;                          // a guarded load is a branch and ten instructions
      wok[s] = row < n_end && kok;
      wv[s] = *reinterpret_cast<const float4*>(W + (size_t)(wok[s] ? row : 0) * K + (wok[s] ? kcol : 0));
    }
    float g[MB * 4], zz[MB * 4];
    // Plain operands, 16-byte aligned: the [16 MB x 64] tile of g as MB float4 per thread (row = unit / 16, 4 columns),
    // requested unconditionally -- 6 requests for the 24 + 24 guarded dword loads of a 96-row round, whose address
    // arithmetic alone was ~500 instructions per wave and round (a round took 4 us for 1.5 us of MFMAs)
    const bool vec = !LAZY && (((uintptr_t)gy | (uintptr_t)(act ? z : gy)) & 15) == 0;
    float4 g4[MB], z4[MB];
    if (vec) {
#pragma unroll
      for (int u = 0; u < MB; ++u) {
        const int unit = u * 256 + (int)threadIdx.x, m = unit >> 4, n = nb + 4 * (unit & 15);
        const bool ok = m < M && n < n_end;                                // N % 4 == 0, rounds of 64: a float4 is all in or all out
        const size_t at = ok ? (size_t)m * N + n : 0;
        g4[u] = *reinterpret_cast<const float4*>(gy + at);
        if (act) z4[u] = *reinterpret_cast<const float4*>(z + at);         // wave-uniform
      }
    } else if constexpr (LAZY) {
      constexpr int SC = MB == 1 ? 8 : (MB == 2 ? 4 : 2);                  // slices in flight per element
#pragma unroll
      for (int t = 0; t < MB * 4; ++t) {
        const int m = 4 * t + wave, n = nb + lane;
        const bool ok = m < M && n < n_end;
        g[t] = (ok && gsum.base) ? gsum.base[(size_t)m * N + n] : 0.f;
        zz[t] = (ok && act) ? z[(size_t)m * N + n] : 0.f;
      }
      for (int s0 = 0; s0 < gsum.n; s0 += SC) {
        float v[MB * 4][SC];
#pragma unroll
        for (int t = 0; t < MB * 4; ++t) {
          const int m = min(4 * t + wave, M - 1), n = min(nb + lane, n_end - 1);      // clamped: always-valid addresses
#pragma unroll
          for (int u = 0; u < SC; ++u) v[t][u] = gsum.slices[(size_t)min(s0 + u, gsum.n - 1) * gsum.stride + (size_t)m * N + n];
        }
#pragma unroll
        for (int t = 0; t < MB * 4; ++t)
#pragma unroll
          for (int u = 0; u < SC; ++u) g[t] += (s0 + u < gsum.n) ? v[t][u] : 0.f;
      }
      if (g_dense && kt == 0) {
#pragma unroll
        for (int t = 0; t < MB * 4; ++t) {
          const int m = 4 * t + wave, n = nb + lane;
          if (m < M && n < n_end) g_dense[(size_t)m * N + n] = g[t];
        }
      }
#pragma unroll
      for (int t = 0; t < MB * 4; ++t) {
        const int m = 4 * t + wave, n = nb + lane;
        if (!(m < M && n < n_end)) g[t] = 0.f;
      }
    } else {
#pragma unroll
      for (int t = 0; t < MB * 4; ++t) {                                   // g[:, nb .. nb+63], row m = 4t + wave
        const int m = 4 * t + wave, n = nb + lane;
        const bool ok = m < M && n < n_end;
        g[t] = ok ? gy[(size_t)m * N + n] : 0.f;
        zz[t] = (ok && act) ? z[(size_t)m * N + n] : 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (!wok[s]) wv[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nb != n_beg) __syncthreads();                                      // readers of the previous round
    if (vec) {
#pragma unroll
      for (int u = 0; u < MB; ++u) {
        const int unit = u * 256 + (int)threadIdx.x, m = unit >> 4, n = nb + 4 * (unit & 15);
        float4 v = g4[u];
        if (act) { v.x *= act_bwd(z4[u].x, act); v.y *= act_bwd(z4[u].y, act); v.z *= act_bwd(z4[u].z, act); v.w *= act_bwd(z4[u].w, act); }
        if (!(m < M && n < n_end)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(sm + m * BI_LD + 4 * (unit & 15)) = v;
      }
    } else {
#pragma unroll
      for (int t = 0; t < MB * 4; ++t) sm[(4 * t + wave) * BI_LD + lane] = act ? g[t] * act_bwd(zz[t], act) : g[t];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float a = sm[(mb * 16 + j) * BI_LD + 16 * s + 4 * wave + q];
        acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wv[s].x, acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wv[s].y, acc[mb][1], 0, 0, 0);
        acc[mb][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wv[s].z, acc[mb][2], 0, 0, 0);
        acc[mb][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wv[s].w, acc[mb][3], 0, 0, 0);
      }
  }
  __syncthreads();
  float4 tot[MB][4];                                                       // [mb][r] = gx[16 mb + 4 q + r][kcol .. +3]
  // Up to 4 row blocks: every wave deposits its partial sums and wave w FINISHES row block w (sums the four partials in
  // wave order -- the same sum as ever -- and stores) instead of wave 0 finishing all of them alone: 192 LDS reads and
  // adds per lane at the end of every launch at 64 rows (the tile kernels' tail of rounds 1-5, see tile_bwd_input_k).
  constexpr bool SHARED_FINISH = MB <= 4;
  if constexpr (MB <= 4) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) sm[(((wave * MB + mb) * 4 + r) * 4 + c) * 64 + lane] = acc[mb][c][r];
    __syncthreads();
    if (wave >= MB) return;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      if (mb != wave) continue;                                            // wave-uniform
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          t[c] = sm[(((0 * MB + mb) * 4 + r) * 4 + c) * 64 + lane];
#pragma unroll
          for (int w = 1; w < 4; ++w) t[c] += sm[(((w * MB + mb) * 4 + r) * 4 + c) * 64 + lane];
        }
        tot[mb][r] = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
  } else {
    for (int w = 1; w < 4; ++w) {                                          // waves 1, 2, 3 in turn (same summation order)
      if (wave == w) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[(((mb * 4 + r) * 4 + c) * 64) + lane] = acc[mb][c][r];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mb][c][r] += sm[(((mb * 4 + r) * 4 + c) * 64) + lane];
      }
      __syncthreads();
    }
    if (wave != 0) return;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) tot[mb][r] = make_float4(acc[mb][0][r], acc[mb][1][r], acc[mb][2][r], acc[mb][3][r]);
  }
  if (NS > 1 || (part && !gx)) {    // row slices meet in the next launch on the stream (reduce kernel or a SliceSum consumer)
    float* mine = part + ((size_t)ns * M) * K;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      if (SHARED_FINISH && mb != wave) continue;                           // wave-uniform: this wave's row block
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mb * 16 + 4 * q + r;
        if (m < M && kok) *reinterpret_cast<float4*>(mine + (size_t)m * K + kcol) = tot[mb][r];
      }
    }
    return;
  }
  if (!kok) return;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    if (SHARED_FINISH && mb != wave) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mb * 16 + 4 * q + r;
      if (m < M) *reinterpret_cast<float4*>(gx + (size_t)m * K + kcol) = tot[mb][r];
    }
  }
}

template <int MB, bool LAZY = false>
__global__ __launch_bounds__(256) void skinny_bwd_input_k(const float* __restrict__ gy, const float* __restrict__ z,
                                                          const float* __restrict__ W, float* __restrict__ gx,
                                                          float* __restrict__ part, int M, int N, int K, int act, int KT,
                                                          int NS, int rpb, SliceSum gsum = SliceSum{nullptr, nullptr, 0, 0},
                                                          float* __restrict__ g_dense = nullptr) {
  skinny_bwd_input_body<MB, LAZY>(blockIdx.x, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, gsum, g_dense);
}

// Two products of one shape in one launch (blockIdx.y picks the problem): the two heads of an MLP pair (mu / sigma:
// cgvae.py:366-371) are independent of each other, as separate launches they are two more dependent links in the chain.
constexpr int BI_MULTI_MAX = 4;
struct BiPair {
  const float* gy[BI_MULTI_MAX]; const float* z[BI_MULTI_MAX]; const float* W[BI_MULTI_MAX];
  float* gx[BI_MULTI_MAX]; float* part[BI_MULTI_MAX];
  int act[BI_MULTI_MAX];
};
template <int MB>
__global__ __launch_bounds__(256) void skinny_bwd_input_pair_k(BiPair p, int M, int N, int K, int KT, int NS, int rpb) {
  const int y = blockIdx.y;
  skinny_bwd_input_body<MB, false>(blockIdx.x, p.gy[y], p.z[y], p.W[y], p.gx[y], p.part[y], M, N, K, p.act[y], KT, NS, rpb);
}

// gx[i] = sum_p part[p][i] (i over M*K/4 float4s), p ascending: deterministic.  4 lanes share one output
// float4 and take every 4th slice, then combine by two xor-shuffles.
// z_out: the stored sum is multiplied by act_out'(z_out) -- the gradient of the pre-activation of the layer that produced
// this product's input (see OutAct in tile_gemm.hip: that layer's own backward then runs without an activation).
__global__ __launch_bounds__(256) void skinny_bwd_input_reduce_k(const float* __restrict__ part, float* __restrict__ gx,
                                                                 int n4, int NS, const float* __restrict__ base = nullptr,
                                                                 long long slice_stride4 = 0,
                                                                 const float* __restrict__ z_out = nullptr, int act_out = 0) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t >> 2, sub = t & 3;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4 && base && sub == 0) acc = reinterpret_cast<const float4*>(base)[i];
  if (i < n4) {
    const float4* p4 = reinterpret_cast<const float4*>(part) + i;
    // a lane's 5 - 8 slices are loaded together (slice index clamped, the surplus zeroed afterwards): the runtime-bounded
    // loop left a remainder of single loads, each a memory round trip of this 4 us launch; summation order unchanged
    constexpr int RB = 8;
    for (int p0 = sub; p0 < NS; p0 += 4 * RB) {
      float4 v[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) v[u] = p4[(size_t)min(p0 + 4 * u, NS - 1) * (slice_stride4 ? (size_t)slice_stride4 : (size_t)n4)];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const bool ok = p0 + 4 * u < NS;
        acc.x += ok ? v[u].x : 0.f; acc.y += ok ? v[u].y : 0.f; acc.z += ok ? v[u].z : 0.f; acc.w += ok ? v[u].w : 0.f;
      }
    }
  }
#pragma unroll
  for (int d = 1; d <= 2; d <<= 1) {
    acc.x += __shfl_xor(acc.x, d); acc.y += __shfl_xor(acc.y, d);
    acc.z += __shfl_xor(acc.z, d); acc.w += __shfl_xor(acc.w, d);
  }
  if (i < n4 && sub == 0) {
    if (z_out) {
      const float4 z4 = reinterpret_cast<const float4*>(z_out)[i];
      acc.x *= act_bwd(z4.x, act_out); acc.y *= act_bwd(z4.y, act_out); acc.z *= act_bwd(z4.z, act_out); acc.w *= act_bwd(z4.w, act_out);
    }
    reinterpret_cast<float4*>(gx)[i] = acc;
  }
}

// row slicing of one bwd_input problem: ~320 blocks when a workspace is available
static inline void bwd_input_plan(int N, int K, bool split, int* KT, int* NS, int* rpb) {
  *KT = (K + BI_COLS - 1) / BI_COLS;
  if (!split) { *NS = 1; *rpb = N; return; }
#ifndef CGV_BI_WANT
#define CGV_BI_WANT 250   /* one round of blocks on 256 CUs: 96 x 5400 x 600 21.9 against 31.7 us at 320 (tools/bwd_input_bench.py), 64 rows 15.4 / 17.1 */
#endif
  const int want = (CGV_BI_WANT + *KT - 1) / *KT;
  int r = (N + want - 1) / want;
  r = (r + 15) / 16 * 16;
  if (r < 32) r = 32;
  *rpb = r;
  *NS = (N + r - 1) / r;
}

// ------------------------------------------------------------------ grouped weight gradient
// One launch for every queued layer: problem p (device table) is  gW_p[N,K] (+)= g_p^T x_p,
// gb_p[n] (+)= sum_m g_p[m,n]  with g_p = gy_p * act'(z_p).  Write-bound: a block produces ROWS
// rows x one k tile of one problem from an LDS-staged x tile; blocks of all problems are
// concatenated (block_begin prefix in the table, binary search per block).
constexpr int WG_ROWS = 16;        // rows of gW per pass = float4 accumulators per thread
constexpr int WG_PASSES = 4;       // passes per block: 64 rows of gW share one LDS-staged x tile
constexpr int WG_BLOCK_ROWS = WG_ROWS * WG_PASSES;

struct WgradProblem {       // mirrors the 88-byte host record built in python (primitives.WeightGradQueue)
  const float* gy;
  const float* x;
  const float* z;           // pre-activation or NULL
  float* gW;
  float* gb;                // or NULL
  int M, N, K;
  int accumulate, act;
  int block_begin;          // first global block index of this problem
  int tiles_k;              // k tiles per row block
  int tile_w;               // floats per k tile (multiple of 4)
  int seg_rows;             // gathered operands (gathered_wgrad_k): rows per rank segment (multiple of 4) ...
  int seg_stride;           // ... and floats between the segments of consecutive ranks; 0 / 0 = one plain [M, .] block
  int pad;
};
static_assert(sizeof(WgradProblem) == 88, "host/device record layout");
// float offset of operand row m (rows of `width` floats): one plain [M, width] block, or -- gathered operands -- rank
// segment m / seg_rows of the all-gathered buffer
__device__ __forceinline__ size_t wg_row(const WgradProblem& pr, int m, int width) {
  if (pr.seg_rows <= 0) return (size_t)m * width;
  const int seg = m / pr.seg_rows;
  return (size_t)seg * pr.seg_stride + (size_t)(m - seg * pr.seg_rows) * width;
}

// The record of this block: block_begin is ascending, so the index is the number of records that begin at or before
// blockIdx.x, minus one -- every lane reads one record's block_begin (64 records per round trip) and a ballot counts.
// (The binary search this replaces was log2(n) DEPENDENT global loads in front of every block's work: 6 at the 57
// problems of a chignolin step.)
template <typename Problem>
__device__ __forceinline__ int wg_find_problem(const Problem* __restrict__ table, int n_problems, int block) {
  const int lane = threadIdx.x & 63;
  int count = 0;
  for (int base = 0; base < n_problems; base += 64) {
    const int i = base + lane;
    const int bb = i < n_problems ? table[i].block_begin : 0x7fffffff;
    count += __popcll(__ballot(bb <= block));
  }
  return __builtin_amdgcn_readfirstlane(count > 0 ? count - 1 : 0);
}
template <typename Problem>
__device__ __forceinline__ int wg_find_problem(const Problem* __restrict__ table, int n_problems) {
  return wg_find_problem(table, n_problems, (int)blockIdx.x);
}

// Rank update (ADAM = true): the tile of gW is never stored -- it goes, clipped, straight into the Adam update of the
// weights it belongs to.  A bead-level layer sees M = 12 rows against 0.36 - 3.2 M weights: its gradient g^T x has rank
// <= 12 and costs 12 FMAs per weight to form, against 12 bytes per weight to write it, read it for the norm and read it
// again in the parameter pass.  The norm comes from the operands instead (wgrad_gram_k), so the step moves 24 bytes per
// weight of these layers (p, m, v read + written) instead of 36.  The arenas are addressed through gW's offset in the
// gradient arena: p = arena_p + (gW - arena_g), likewise m and v.
struct RankUpdateArgs {
  const float* arena_g;
  float* arena_p;
  float* arena_m;
  float* arena_v;
  const float* state;
  float lr, beta1, beta2, eps;
};

template <bool ADAM>
__device__ __forceinline__ void grouped_wgrad_body(const WgradProblem* __restrict__ table, int n_problems, const RankUpdateArgs& ra,
                                                   int block, float* smem) {
  if (ADAM && ra.state[ST_SKIP] != 0.f) return;              // skipped step (utils.py:145): parameters stay
  // locate the problem of this block (table is tiny; block_begin ascending)
  const int lo = wg_find_problem(table, n_problems, block);
  const WgradProblem pr = table[lo];
  const int local = block - pr.block_begin;
  const int rb = local / pr.tiles_k, kt = local - rb * pr.tiles_k;
  const int M = pr.M, N = pr.N, K = pr.K, tile_w = pr.tile_w;
  float* xs = smem;                          // [M][tile_w]
  float* gs = smem + (size_t)M * tile_w;     // [M][WG_BLOCK_ROWS]
  const int t = threadIdx.x;
  const int n0 = rb * WG_BLOCK_ROWS;
  const int t4 = tile_w >> 2;                                   // float4 columns of the k tile
  const int kbase = kt * tile_w;
  // x tile: all 256 threads, 4 float4 in flight each (rows x float4 columns, coalesced along k)
  for (int base = 0; base < M * t4; base += 1024) {
    float4 val[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * 256 + t;
      const int m = idx / t4, c = idx - m * t4;
      val[u] = ldg4g_or_zero(pr.x + wg_row(pr, idx < M * t4 ? m : 0, K) + kbase + 4 * c, idx < M * t4 && kbase + 4 * c < K);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * 256 + t;
      if (idx < M * t4) reinterpret_cast<float4*>(xs)[idx] = val[u];
    }
  }
  // g tile, coalesced along n: M / 4 rounds of 256 elements, 4 rounds' loads in flight (clamped addresses; the plain
  // loop -- guarded load, activation, store -- was a memory round trip per round before the block could start)
  for (int base = 0; base < M * WG_BLOCK_ROWS; base += 1024) {
    float gv[4], zv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = min(base + 256 * u + t, M * WG_BLOCK_ROWS - 1);
      const int m = idx / WG_BLOCK_ROWS, r = idx - m * WG_BLOCK_ROWS;
      const size_t at = wg_row(pr, m, N) + min(n0 + r, N - 1);
      gv[u] = ldg_global(pr.gy + at);
      zv[u] = pr.act ? ldg_global(pr.z + at) : 0.f;              // block-uniform
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + 256 * u + t;
      if (idx < M * WG_BLOCK_ROWS) {
        const int m = idx / WG_BLOCK_ROWS, r = idx - m * WG_BLOCK_ROWS;
        float g = pr.act ? gv[u] * act_bwd(zv[u], pr.act) : gv[u];
        gs[idx] = n0 + r < N ? g : 0.f;
      }
    }
  }
  __syncthreads();
  // narrow k tiles leave threads without a column: the 4 row passes are dealt to 2 or 4 thread groups instead
  const int lanes = t4 <= 64 ? 64 : t4 <= 128 ? 128 : 256;     // threads per group (whole waves)
  const int groups = 256 / lanes;
  const int tc = t & (lanes - 1), grp = t / lanes;
  const int k = kbase + 4 * tc;
  if (tc < t4 && k < K) {
    // rank update: 8 rows per pass, and the p / m / v of ALL of them are requested before the tile is formed (two register
    // sets of 4 rows): with 16 rows per pass only the first 4 rows' requests travelled under the FMAs and each later group
    // of 4 paid a whole memory round trip in front of its update
    constexpr int ROWS = ADAM ? 8 : WG_ROWS, PASSES = WG_BLOCK_ROWS / ROWS;
    for (int pass = grp; pass < PASSES; pass += groups) {
      const int nr = n0 + pass * ROWS;
      if (nr >= N) break;
      float4 acc[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      typedef float f4v __attribute__((ext_vector_type(4)));
      float4 pp[ADAM ? ROWS : 1], mm[ADAM ? ROWS : 1], vv[ADAM ? ROWS : 1];
      const size_t at = ADAM ? (size_t)(pr.gW - ra.arena_g) + (size_t)nr * K + k : 0;
      if (ADAM) {
        // rows beyond N are clamped onto the last one (their results are not stored): no branch around a request
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const size_t o = at + (size_t)min(r, N - 1 - nr) * K;
          pp[r] = ldg4_global(ra.arena_p + o);
          const f4v tm = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) f4v*>((const __attribute__((address_space(1))) float*)(ra.arena_m + o)));
          const f4v tv = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) f4v*>((const __attribute__((address_space(1))) float*)(ra.arena_v + o)));
          mm[r] = make_float4(tm.x, tm.y, tm.z, tm.w);
          vv[r] = make_float4(tv.x, tv.y, tv.z, tv.w);
        }
        asm volatile("" ::: "memory");                            // the requests stay in front of the FMAs
      }
      for (int m = 0; m < M; ++m) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + (size_t)m * tile_w + 4 * tc);
        const float4* g4 = reinterpret_cast<const float4*>(gs + m * WG_BLOCK_ROWS + pass * ROWS);   // LDS broadcast
#pragma unroll
        for (int i = 0; i < ROWS / 4; ++i) {
          const float4 gv = g4[i];
          const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float4& a = acc[4 * i + c];
            a.x = fmaf(gg[c], xv.x, a.x); a.y = fmaf(gg[c], xv.y, a.y);
            a.z = fmaf(gg[c], xv.z, a.z); a.w = fmaf(gg[c], xv.w, a.w);
          }
        }
      }
      if (ADAM) {
        const AdamStep a = adam_step_of(ra.state, ra.lr, ra.beta1, ra.beta2, ra.eps);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          if (nr + r < N) {
            const size_t o = at + (size_t)r * K;
            const float4 g = acc[r];
            adam_elem(a, pp[r].x, g.x, mm[r].x, vv[r].x); adam_elem(a, pp[r].y, g.y, mm[r].y, vv[r].y);
            adam_elem(a, pp[r].z, g.z, mm[r].z, vv[r].z); adam_elem(a, pp[r].w, g.w, mm[r].w, vv[r].w);
            *reinterpret_cast<float4*>(ra.arena_p + o) = pp[r];
            __builtin_nontemporal_store(f4v{mm[r].x, mm[r].y, mm[r].z, mm[r].w}, reinterpret_cast<f4v*>(ra.arena_m + o));
            __builtin_nontemporal_store(f4v{vv[r].x, vv[r].y, vv[r].z, vv[r].w}, reinterpret_cast<f4v*>(ra.arena_v + o));
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          if (nr + r < N) {
            float* dst = pr.gW + (size_t)(nr + r) * K + k;
            float4 o = acc[r];
            if (pr.accumulate) { const float4 old = ldg4_global(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
            stg4_global(dst, o);
          }
        }
      }
    }
  }
  if (ADAM) return;                                            // the bias gradient was written by wgrad_gram_k
  if (pr.gb && kt == 0 && t < WG_BLOCK_ROWS && n0 + t < N) {
    float sum = 0.f;
    for (int m = 0; m < M; ++m) sum += gs[m * WG_BLOCK_ROWS + t];
    pr.gb[n0 + t] = pr.accumulate ? pr.gb[n0 + t] + sum : sum;
  }
}

template <bool ADAM>
__global__ __launch_bounds__(256) void grouped_wgrad_t(const WgradProblem* __restrict__ table, int n_problems, RankUpdateArgs ra) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  grouped_wgrad_body<ADAM>(table, n_problems, ra, (int)blockIdx.x, smem);
}

// Rank update, FLAT layout (few operand rows: M <= RF_MAX_ROWS).  The update is elementwise over the weight's [N, K] array,
// which is contiguous: a block takes a contiguous range of q4 float4 of it -- whole 128-byte lines of p, m and v, each read
// and written exactly once -- instead of 64 rows x one k tile.  (K = 600 is cut into 3 tiles of 200 columns there: row
// segments of 800 bytes at a stride of 2400, 6.25 lines each, so the three blocks of a row block -- on three XCDs --
// fetch the lines at the tile borders twice: 860 MB read for 730 MB of p / m / v on the chignolin step, FETCH_SIZE.)
// Each thread's 8 float4 of a round lie 256 float4 apart, each with its own (row n, column k): the operand tile in LDS is
// x for ALL K columns plus g for the rows the range touches, and an output reads one float4 of x and one float of g per
// operand row.  The sum over the operand rows runs in the same order as in grouped_wgrad_t<true>: bit-identical results.
constexpr int RF_MAX_ROWS = 16;          // LDS reads per FMA grow with the rows: beyond this the tiled kernel's shared x / g reads win
#ifndef CGV_RF_UNR
#define CGV_RF_UNR 4
#endif
#ifndef CGV_RF_PIPE
#define CGV_RF_PIPE 0
#endif
constexpr int RF_UNR = CGV_RF_UNR;       // float4 per thread and round (3 x RF_UNR requests of 16 bytes in flight)
constexpr bool RF_PIPE = CGV_RF_PIPE;    // the next round's p / m / v requested before this round's tile is formed
constexpr int RF_ROUND_F4 = 256 * RF_UNR;
constexpr int RF_QUANTUM_F4 = 2048;      // q4 is a multiple of this (and of RF_ROUND_F4)
static_assert(RF_QUANTUM_F4 % RF_ROUND_F4 == 0, "rounds tile the quantum");

__device__ __forceinline__ void rank_update_flat_body(const WgradProblem* __restrict__ table, int n_problems, const RankUpdateArgs& ra,
                                                      int q4 /* float4 per block: a multiple of RF_QUANTUM_F4 */, int block, float* smem) {
  if (ra.state[ST_SKIP] != 0.f) return;                          // skipped step (utils.py:145): parameters stay
  const int lo = wg_find_problem(table, n_problems, block);
  const WgradProblem pr = table[lo];
  const int local = block - pr.block_begin;
  const int M = pr.M, N = pr.N, K = pr.K, K4 = K >> 2;
  const int total4 = N * K4;
  const int f_lo = local * q4, f_hi = min(total4, f_lo + q4);
  if (f_lo >= f_hi) return;
  const int r_lo = f_lo / K4;
  const int G = min(q4 / K4 + 2, N);                              // rows of g staged: the range touches at most that many
  float* xs = smem;                                              // [M][K]
  float* gs = smem + (size_t)M * K;                              // [M][G]
  const int t = threadIdx.x;
  typedef float f4v __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) f4v* gptr;
  const size_t arena0 = (size_t)(pr.gW - ra.arena_g);
  const int dq = 256 / K4, dr = 256 - dq * K4;                    // (n, k4) of an index 256 float4 further on
  const int n_last = (f_hi - 1) / K4, k_last = (f_hi - 1) - n_last * K4;

  // p / m / v of round c (thread t: float4 c + t + 256 i) requested, with the LDS offsets of each float4's x and g
  auto request = [&](int c, float4 (&pp)[RF_UNR], float4 (&mm)[RF_UNR], float4 (&vv)[RF_UNR], int (&xo)[RF_UNR], int (&go)[RF_UNR]) {
    const int i0 = min(c + t, f_hi - 1);
    int n = i0 / K4, k4 = i0 - n * K4;
#pragma unroll
    for (int i = 0; i < RF_UNR; ++i) {
      // indices beyond the range are clamped onto its last float4 (their results are not stored): no branch around a request
      const bool in = c + t + 256 * i < f_hi;
      const int nn = in ? n : n_last, kk = in ? k4 : k_last;
      xo[i] = 4 * kk;
      go[i] = nn - r_lo;
      const size_t o = arena0 + 4 * ((size_t)nn * K4 + kk);
      pp[i] = ldg4_global(ra.arena_p + o);
      const f4v tm = __builtin_nontemporal_load(reinterpret_cast<gptr>((const __attribute__((address_space(1))) float*)(ra.arena_m + o)));
      const f4v tv = __builtin_nontemporal_load(reinterpret_cast<gptr>((const __attribute__((address_space(1))) float*)(ra.arena_v + o)));
      mm[i] = make_float4(tm.x, tm.y, tm.z, tm.w);
      vv[i] = make_float4(tv.x, tv.y, tv.z, tv.w);
      k4 += dr; n += dq;
      if (k4 >= K4) { k4 -= K4; ++n; }
    }
  };
  float4 pa[RF_UNR], ma[RF_UNR], va[RF_UNR];
  int xa[RF_UNR], ga[RF_UNR];
  request(f_lo, pa, ma, va, xa, ga);                              // ... under the staging of the operand rows
  asm volatile("" ::: "memory");

  for (int base = 0; base < M * K4; base += 1024) {
    float4 val[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = min(base + u * 256 + t, M * K4 - 1);
      const int m = idx / K4, c = idx - m * K4;
      val[u] = ldg4_global(pr.x + wg_row(pr, m, K) + 4 * c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * 256 + t;
      if (idx < M * K4) reinterpret_cast<float4*>(xs)[idx] = val[u];
    }
  }
  const int rows = min(G, N - r_lo);
  for (int base = 0; base < M * rows; base += 1024) {
    float gv[4], zv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = min(base + 256 * u + t, M * rows - 1);
      const int m = idx / rows, r = idx - m * rows;
      const size_t at = wg_row(pr, m, N) + r_lo + r;
      gv[u] = ldg_global(pr.gy + at);
      zv[u] = pr.act ? ldg_global(pr.z + at) : 0.f;              // block-uniform
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + 256 * u + t;
      if (idx < M * rows) {
        const int m = idx / rows, r = idx - m * rows;
        gs[m * G + r] = pr.act ? gv[u] * act_bwd(zv[u], pr.act) : gv[u];
      }
    }
  }
  __syncthreads();
  const AdamStep a = adam_step_of(ra.state, ra.lr, ra.beta1, ra.beta2, ra.eps);
  // tile of round c formed (operand rows in ascending order, as in grouped_wgrad_t), through Adam, stored
  auto finish = [&](int c, float4 (&pp)[RF_UNR], float4 (&mm)[RF_UNR], float4 (&vv)[RF_UNR], const int (&xo)[RF_UNR], const int (&go)[RF_UNR]) {
    float4 acc[RF_UNR];
#pragma unroll
    for (int i = 0; i < RF_UNR; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = 0; m < M; ++m) {
      const float* xr = xs + (size_t)m * K;
      const float* gr = gs + m * G;
#pragma unroll
      for (int i = 0; i < RF_UNR; ++i) {
        const float4 xv = *reinterpret_cast<const float4*>(xr + xo[i]);
        const float g = gr[go[i]];
        acc[i].x = fmaf(g, xv.x, acc[i].x); acc[i].y = fmaf(g, xv.y, acc[i].y);
        acc[i].z = fmaf(g, xv.z, acc[i].z); acc[i].w = fmaf(g, xv.w, acc[i].w);
      }
    }
#pragma unroll
    for (int i = 0; i < RF_UNR; ++i) {
      if (c + t + 256 * i < f_hi) {
        const size_t o = arena0 + 4 * (size_t)(c + t + 256 * i);
        const float4 g = acc[i];
        adam_elem(a, pp[i].x, g.x, mm[i].x, vv[i].x); adam_elem(a, pp[i].y, g.y, mm[i].y, vv[i].y);
        adam_elem(a, pp[i].z, g.z, mm[i].z, vv[i].z); adam_elem(a, pp[i].w, g.w, mm[i].w, vv[i].w);
        *reinterpret_cast<float4*>(ra.arena_p + o) = pp[i];
        __builtin_nontemporal_store(f4v{mm[i].x, mm[i].y, mm[i].z, mm[i].w}, reinterpret_cast<f4v*>(ra.arena_m + o));
        __builtin_nontemporal_store(f4v{vv[i].x, vv[i].y, vv[i].z, vv[i].w}, reinterpret_cast<f4v*>(ra.arena_v + o));
      }
    }
  };
  if constexpr (!RF_PIPE) {
    for (int c = f_lo; c < f_hi; c += RF_ROUND_F4) {
      if (c != f_lo) { request(c, pa, ma, va, xa, ga); asm volatile("" ::: "memory"); }   // the requests stay in front of the FMAs
      finish(c, pa, ma, va, xa, ga);
    }
  } else {
    float4 pb[RF_UNR], mb[RF_UNR], vb[RF_UNR];
    int xb[RF_UNR], gb[RF_UNR];
    for (int c = f_lo; c < f_hi; c += 2 * RF_ROUND_F4) {
      const bool second = c + RF_ROUND_F4 < f_hi;                 // block-uniform
      if (second) request(c + RF_ROUND_F4, pb, mb, vb, xb, gb);
      asm volatile("" ::: "memory");
      finish(c, pa, ma, va, xa, ga);
      if (second) {
        if (c + 2 * RF_ROUND_F4 < f_hi) request(c + 2 * RF_ROUND_F4, pa, ma, va, xa, ga);
        asm volatile("" ::: "memory");
        finish(c + RF_ROUND_F4, pb, mb, vb, xb, gb);
      }
    }
  }
}

// One launch for a table whose first n_flat records take the flat layout (f_blocks blocks) and whose other records the
// tiled one (t_blocks blocks, their own block prefix).  The tiled records are the layers of more rows (36: the three
// stacked heads), whose blocks are bound by the FMAs of forming the tile, not by p / m / v: dealt evenly among the flat
// blocks -- every P-th block of the launch -- they run beside blocks that wait for memory instead of after them.
__global__ __launch_bounds__(256) void rank_update_mixed_k(const WgradProblem* __restrict__ table, int n_flat, int n_problems,
                                                           RankUpdateArgs ra, int q4, int t_blocks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  if (t_blocks > 0) {
    const int P = (int)gridDim.x / t_blocks;                      // >= 1
    const int q = b / P, r = b - q * P;
    if (r == 0 && q < t_blocks) {
      grouped_wgrad_body<true>(table + n_flat, n_problems - n_flat, ra, q, smem);
      return;
    }
    rank_update_flat_body(table, n_flat, ra, q4, b - min(q + 1, t_blocks), smem);
    return;
  }
  rank_update_flat_body(table, n_flat, ra, q4, b, smem);
}

// ------------------------------------------------------------------ norm of a weight gradient from its operands
// ||g^T x||_F^2 = sum_{a<=b} c_ab (g_a . g_b)(x_a . x_b),  c = 1 on the diagonal and 2 off it, over the M operand rows
// (g = gy * act'(z)): the squared norm of a rank-update layer's gradient without forming it.
//   wgrad_gram_k         grid (GRAM_SLICES, problems): block s takes the column slices s, s + GRAM_SLICES, ... of the
//                        problem's rows -- first g's N columns, then x's K -- C4 float4 per row at a time, staged in LDS
//                        (activation derivative applied once, at staging); 8 lanes share a pair's partial dot product
//                        over the slice (double), the block's slices are summed per pair in LDS and leave as its own
//                        workspace rows ws[problem][s][g | x][pair].  One global round trip per slice, blocks independent.
//   wgrad_gram_reduce_k  one block per problem: sums the slices per pair (fixed order) and the pairs' products.
// The blocks of wgrad_gram_k also write the bias gradient gb[n] (+)= sum_m g[m, n] (a column slice each), which the
// fused update does not produce.
constexpr int GRAM_SLICES = 8;
constexpr int GRAM_WAVES = 8;
constexpr int GRAM_THREADS = 64 * GRAM_WAVES;
constexpr int GRAM_F4_PER_THREAD = 6;                                // staged float4 per thread and slice (<= 3072)
constexpr int GRAM_TILE_F4 = 3200;                                   // rows are padded by one float4 (bank spread)
// Rows: a single GPU's bead-level layers have 12 (<= 40: the LDS request stays below 64 KB, two blocks per CU); the
// gathered operands of the data-parallel exchange have world x 12 -- up to 64 (cgv_rank_update_supported), beyond
// which walking M^2 / 2 row pairs and re-forming the tiles stops paying against materialising the gradient.
constexpr int GRAM_MAX_ROWS = 64;
constexpr int GRAM_MAX_PAIRS = GRAM_MAX_ROWS * (GRAM_MAX_ROWS + 1) / 2;                         // 2080
constexpr size_t GRAM_WS_DOUBLES = (size_t)GRAM_SLICES * 2 * GRAM_MAX_PAIRS;                    // per problem
static size_t gram_lds_bytes(int max_rows) { return sizeof(float4) * GRAM_TILE_F4 + sizeof(double) * (size_t)max_rows * (max_rows + 1); }

constexpr int GRAM_TICKETS = 512;                                    // >= primitives.WeightGradQueue.MAX_PROBLEMS
__device__ unsigned gram_tickets[GRAM_TICKETS];                      // zero at load, every launch leaves them zero

__global__ __launch_bounds__(GRAM_THREADS) void wgrad_gram_k(const WgradProblem* __restrict__ table, double* __restrict__ ws,
                                                             int pair_cap /* pairs the LDS sums hold per operand */,
                                                             double* __restrict__ out /* [problems] or NULL */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float4* tile = reinterpret_cast<float4*>(smem);                // [M][C4 + 1]
  double* sums = reinterpret_cast<double*>(tile + GRAM_TILE_F4); // [g | x][pair_cap]: this block's slices, summed
  const WgradProblem pr = table[blockIdx.y];
  const int sl = blockIdx.x;
  const int M = pr.M, N = pr.N, K = pr.K, act = pr.act;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  double* mine = ws + (size_t)blockIdx.y * GRAM_WS_DOUBLES + (size_t)sl * 2 * GRAM_MAX_PAIRS;
  for (int i = t; i < 2 * pair_cap; i += GRAM_THREADS) sums[i] = 0.0;
  const bool unsupported = M > GRAM_MAX_ROWS || M * (M + 1) / 2 > pair_cap;
  if (pr.gb && unsupported) {                                   // bias gradient (supported shapes: from the staged slices, below)
    for (int n = sl * GRAM_THREADS + t; n < N; n += GRAM_SLICES * GRAM_THREADS) {
      float sum = 0.f;
#pragma unroll 4
      for (int m = 0; m < M; ++m) {
        const size_t at = wg_row(pr, m, N) + n;
        float g = ldg_global(pr.gy + at);
        if (act) g *= act_bwd(ldg_global(pr.z + at), act);
        sum += g;
      }
      pr.gb[n] = pr.accumulate ? pr.gb[n] + sum : sum;
    }
  }
  if (unsupported) {                                            // (cgv_rank_update_supported): poison the norm
    if (t == 0) mine[0] = __builtin_nan("");
    return;
  }
  int C4 = (GRAM_TILE_F4 / M - 1) & ~63;
  if (C4 == 0) C4 = (GRAM_TILE_F4 / M - 1) & ~15;                // more than 49 rows: 48 / 32 float4 per row and slice
  C4 = C4 > 256 ? 256 : C4;
  const int RS = C4 + 1;                                         // row stride (float4)
  const int n4 = N >> 2, k4 = K >> 2;
  const int g_slices = (n4 + C4 - 1) / C4, slices = g_slices + (k4 + C4 - 1) / C4;
  const int pairs = M * (M + 1) / 2;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int slice, float4 (&buf)[GRAM_F4_PER_THREAD]) {
    const bool is_g = slice < g_slices;
    const int c0 = (is_g ? slice : slice - g_slices) * C4, cols4 = is_g ? n4 : k4;
#pragma unroll
    for (int u = 0; u < GRAM_F4_PER_THREAD; ++u) {
      const int idx = t + GRAM_THREADS * u;
      const int m = idx / C4, col4 = c0 + idx - m * C4;
      buf[u] = zero4;
      if (m < M && col4 < cols4) {
        if (is_g) {
          float4 g = ldg4_global(pr.gy + wg_row(pr, m, N) + 4 * col4);
          if (act) {
            const float4 z = ldg4_global(pr.z + wg_row(pr, m, N) + 4 * col4);
            g.x *= act_bwd(z.x, act); g.y *= act_bwd(z.y, act); g.z *= act_bwd(z.z, act); g.w *= act_bwd(z.w, act);
          }
          buf[u] = g;
        } else {
          buf[u] = ldg4_global(pr.x + wg_row(pr, m, K) + 4 * col4);
        }
      }
    }
  };
  // 8 lanes share a pair (an eighth of the slice's columns each), a wave pass covers 8 pairs
  const int sub = lane & 7, pl = lane >> 3;
  float4 buf[GRAM_F4_PER_THREAD];
  if (sl < slices) fetch(sl, buf);
  for (int slice = sl; slice < slices; slice += GRAM_SLICES) {
    __syncthreads();                                             // previous slice consumed (and `sums` zeroed)
#pragma unroll
    for (int u = 0; u < GRAM_F4_PER_THREAD; ++u) {
      const int idx = t + GRAM_THREADS * u;
      const int m = idx / C4;
      if (m < M) tile[m * RS + idx - m * C4] = buf[u];
    }
    if (slice + GRAM_SLICES < slices) fetch(slice + GRAM_SLICES, buf);       // in flight while this slice is used
    __syncthreads();
    if (pr.gb && slice < g_slices) {
      // bias gradient of this slice's columns: column sums of the staged g (rows ascending) -- as a loop over global
      // memory in front of the first fetch it was M dependent row loads per column, ~6 round trips before the block started
      const int c0 = slice * C4;
      for (int c = t; c < C4 && c0 + c < n4; c += GRAM_THREADS) {
        float4 sum = zero4;
        for (int m = 0; m < M; ++m) {
          const float4 g = tile[m * RS + c];
          sum.x += g.x; sum.y += g.y; sum.z += g.z; sum.w += g.w;
        }
        float* dst = pr.gb + 4 * (c0 + c);
        if ((reinterpret_cast<uintptr_t>(pr.gb) & 15) == 0) {             // (block-uniform; arena slots are 256-byte aligned)
          if (pr.accumulate) { const float4 old = ldg4_global(dst); sum.x += old.x; sum.y += old.y; sum.z += old.z; sum.w += old.w; }
          stg4_global(dst, sum);
        } else {
          const float v4[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e] = pr.accumulate ? dst[e] + v4[e] : v4[e];
        }
      }
    }
    double* row = sums + (slice < g_slices ? 0 : pair_cap);
    for (int base = 8 * w; base < pairs; base += 8 * GRAM_WAVES) {
      const int pidx = base + pl;
      const bool live = pidx < pairs;
      // pair index -> (a <= b), row-major upper triangle: rows before a hold S(a) = a M - a (a - 1) / 2 pairs
      const int q = live ? pidx : 0;
      const float disc = (float)((2 * M + 1) * (2 * M + 1) - 8 * q);
      int a = (int)(((float)(2 * M + 1) - sqrtf(disc)) * 0.5f);
      a = a < 0 ? 0 : (a > M - 1 ? M - 1 : a);
      while (a + 1 < M && (a + 1) * M - (a + 1) * a / 2 <= q) ++a;
      while (a > 0 && a * M - a * (a - 1) / 2 > q) --a;
      const int b = a + q - (a * M - a * (a - 1) / 2);
      double acc = 0.0;
      if (live) {
        for (int c = sub; c < C4; c += 8) {
          const float4 u4 = tile[a * RS + c], v4 = tile[b * RS + c];
          acc += (double)u4.x * v4.x + (double)u4.y * v4.y + (double)u4.z * v4.z + (double)u4.w * v4.w;
        }
      }
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      acc += __shfl_xor(acc, 4);
      if (live && sub == 0) row[pidx] += acc;                    // this lane owns the pair in every slice of the block
    }
  }
  __syncthreads();
  // agent-scope stores: the block that arrives LAST at this problem's ticket sums all slices (below)
  for (int i = t; i < pairs; i += GRAM_THREADS) {
    __hip_atomic_store(mine + i, sums[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(mine + GRAM_MAX_PAIRS + i, sums[pair_cap + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (out == nullptr) return;                                     // (two-launch form: wgrad_gram_reduce_k follows)
  // One launch instead of two: the eight slice blocks of a problem meet at a ticket (eight arrivals -- the pattern that is
  // too slow for the 1620 blocks of the flat norm pass pays here); atomicInc wraps to zero at the eighth, so the tickets
  // need no reset and a launch that never finished cannot wedge the next one.
  // (No __threadfence: an agent-scope release writes back the WHOLE L2 -- right behind the backward pass that is 60 us per
  // launch, measured.  The partials travel as agent-scope atomics, which are coherent across the XCDs by themselves; every
  // thread waits for its own stores to be acknowledged before the barrier that precedes the ticket.)
  __shared__ unsigned s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    s_last = atomicInc(gram_tickets + (blockIdx.y % GRAM_TICKETS), (unsigned)GRAM_SLICES - 1u) == (unsigned)GRAM_SLICES - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  const double* all = ws + (size_t)blockIdx.y * GRAM_WS_DOUBLES;
  double local = 0.0;
  for (int p = t; p < pairs; p += GRAM_THREADS) {
    double gg = 0.0, xx = 0.0;
    for (int s2 = 0; s2 < GRAM_SLICES; ++s2) {
      gg += __hip_atomic_load(all + (size_t)s2 * 2 * GRAM_MAX_PAIRS + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      xx += __hip_atomic_load(all + (size_t)s2 * 2 * GRAM_MAX_PAIRS + GRAM_MAX_PAIRS + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int a = 0, rem = p;                                           // diagonal pairs (a, a) count once (see wgrad_gram_reduce_k)
    while (rem >= M - a) { rem -= M - a; ++a; }
    local += (rem == 0 ? 1.0 : 2.0) * gg * xx;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) local += __shfl_xor(local, d);
  __shared__ double wsum[GRAM_WAVES];
  if (lane == 0) wsum[w] = local;
  __syncthreads();
  if (t == 0) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < GRAM_WAVES; ++k) tot += wsum[k];
    out[blockIdx.y] = tot;
  }
}

__global__ __launch_bounds__(256) void wgrad_gram_reduce_k(const WgradProblem* __restrict__ table, const double* __restrict__ ws,
                                                           double* __restrict__ out) {
  __shared__ double part[4];
  const int M = table[blockIdx.x].M;
  const double* mine = ws + (size_t)blockIdx.x * GRAM_WS_DOUBLES;
  const int pairs = M <= GRAM_MAX_ROWS ? M * (M + 1) / 2 : 1;
  double local = 0.0;
  for (int p = threadIdx.x; p < pairs; p += 256) {
    double gg = 0.0, xx = 0.0;
    for (int s = 0; s < GRAM_SLICES; ++s) {
      gg += mine[(size_t)s * 2 * GRAM_MAX_PAIRS + p];
      xx += mine[(size_t)s * 2 * GRAM_MAX_PAIRS + GRAM_MAX_PAIRS + p];
    }
    // diagonal pairs are (a, a): p = a M - a (a - 1) / 2; cheaper to recover a by walking than to store it
    int a = 0, rem = p;
    while (rem >= M - a) { rem -= M - a; ++a; }
    local += (rem == 0 ? 1.0 : 2.0) * gg * (M <= GRAM_MAX_ROWS ? xx : 1.0);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) local += __shfl_xor(local, d);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// The same norm for MANY operand rows (41 .. 128: gathered rows of 4 - 8 ranks, bead rows of a large batch), where walking
// M^2 / 2 row pairs on the vector ALU costs more than forming the gradient tiles (48 rows: 140 us against 82 us for the
// tile pass with a squaring epilogue).  The two Gram matrices G = g g^T and X = x x^T are built from 16 x 16 tiles of
// the upper triangle with v_mfma_f64_16x16x4_f64 -- operands widened on the way out of LDS, so products and sums are the
// doubles of wgrad_gram_k -- and ||g^T x||^2 = sum_ab G_ab X_ab (off-diagonal tiles count twice).
//   wgrad_gram_mfma_k     grid (GRAMM_BLOCKS, problems), the column slices and LDS layout of wgrad_gram_k (rows M .. 16 NT - 1
//                         stay zero); block s takes the slices s, s + GRAMM_BLOCKS, ... -- one or two for the model's
//                         layers: the instruction runs at a quarter of the fp32 rate, the work has to be spread by
//                         columns (with 8 blocks per problem the 5400-column layers' blocks set the launch's length) --
//                         and wave w keeps the tiles w, w + 8, ... in registers over them.  They leave as
//                         ws[problem][block][G | X][tile][lane][4] (G when the block's slices turn from g to x, X at the end).
//                         Bias gradients as in wgrad_gram_k.
//   wgrad_gram_mfma_dot_k grid (tiles, problems): blocks' parts summed per element (fixed order), tile's sum of G_ab X_ab
//   wgrad_gram_mfma_sum_k one thread per problem: the tiles' sums in fixed order.
// A tile's 256 elements sit in the same lanes / registers for G and for X (same instruction), and both operands of a
// step read column c + (lane >> 4) of row (lane & 15), so the result does not depend on the instruction's register map.
constexpr int GRAMM_MAX_ROWS = 128;
constexpr int GRAMM_MAX_TILES = 36;                                  // upper triangle of 8 x 8 row groups
constexpr int GRAMM_BLOCKS = 32;
constexpr int GRAMM_LDS_BYTES = 52 * 1024;
typedef double f64x4 __attribute__((ext_vector_type(4)));
// workspace doubles per problem for a launch whose largest problem has `tiles` tiles: the blocks' parts, then the tiles' sums
__device__ __host__ inline size_t gramm_ws_doubles(int tiles) { return (size_t)GRAMM_BLOCKS * 2 * tiles * 256 + GRAMM_MAX_TILES; }
// columns per slice: 16 NT rows x C / 4 float4 <= 3072 (six per thread), rows padded by one float4, at most 52 KB
__device__ __host__ inline int gramm_cols(int MP) { const int c = (12288 / MP) & ~31; return c > 256 ? 256 : c; }
__device__ __forceinline__ void gramm_tile_of(int tt, int NT, int& a, int& b) {
  a = 0;
  while (tt >= NT - a) { tt -= NT - a; ++a; }
  b = a + tt;
}

template <int TPW>   // tiles per wave: 3 up to 96 rows (21 tiles), 5 up to 128 (36)
__global__ __launch_bounds__(GRAM_THREADS) void wgrad_gram_mfma_k(const WgradProblem* __restrict__ table, double* __restrict__ ws,
                                                                  int launch_tiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float4* tile = reinterpret_cast<float4*>(smem);                // [16 NT][C4 + 1]
  const WgradProblem pr = table[blockIdx.y];
  const int sl = blockIdx.x;
  const int M = pr.M, N = pr.N, K = pr.K, act = pr.act;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int NT = (M + 15) >> 4, MP = 16 * NT, T = NT * (NT + 1) / 2;
  if (M > GRAMM_MAX_ROWS || T > launch_tiles || T > TPW * GRAM_WAVES) return;          // (wgrad_gram_mfma_dot_k poisons the norm)
  const int C4 = gramm_cols(MP) >> 2, RS = C4 + 1;
  const int n4 = N >> 2, k4 = K >> 2;
  const int g_slices = (n4 + C4 - 1) / C4, slices = g_slices + (k4 + C4 - 1) / C4;
  if (sl >= slices) return;
  double* mine = ws + (size_t)blockIdx.y * gramm_ws_doubles(launch_tiles) + (size_t)sl * 2 * launch_tiles * 256;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int idx = t; idx < (MP - M) * RS; idx += GRAM_THREADS) tile[M * RS + idx] = zero4;     // never written again
  auto fetch = [&](int slice, float4 (&buf)[GRAM_F4_PER_THREAD]) {
    const bool is_g = slice < g_slices;
    const int c0 = (is_g ? slice : slice - g_slices) * C4, cols4 = is_g ? n4 : k4;
#pragma unroll
    for (int u = 0; u < GRAM_F4_PER_THREAD; ++u) {
      const int idx = t + GRAM_THREADS * u;
      const int m = idx / C4, col4 = c0 + idx - m * C4;
      buf[u] = zero4;
      if (m < M && col4 < cols4) {
        if (is_g) {
          float4 g = ldg4_global(pr.gy + wg_row(pr, m, N) + 4 * col4);
          if (act) {
            const float4 z = ldg4_global(pr.z + wg_row(pr, m, N) + 4 * col4);
            g.x *= act_bwd(z.x, act); g.y *= act_bwd(z.y, act); g.z *= act_bwd(z.z, act); g.w *= act_bwd(z.w, act);
          }
          buf[u] = g;
        } else {
          buf[u] = ldg4_global(pr.x + wg_row(pr, m, K) + 4 * col4);
        }
      }
    }
  };
  int ta[TPW], tb[TPW];
  f64x4 acc[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const int tt = w + GRAM_WAVES * u;
    gramm_tile_of(tt < T ? tt : 0, NT, ta[u], tb[u]);
    acc[u] = f64x4{0.0, 0.0, 0.0, 0.0};
  }
  const int i = lane & 15, q = lane >> 4;
  const float* tf = reinterpret_cast<const float*>(tile);
  const int RSf = 4 * RS, C = 4 * C4;
  float4 buf[GRAM_F4_PER_THREAD];
  fetch(sl, buf);
  for (int slice = sl; slice < slices; slice += GRAMM_BLOCKS) {
    __syncthreads();                                             // previous slice consumed (and the zero rows written)
#pragma unroll
    for (int u = 0; u < GRAM_F4_PER_THREAD; ++u) {
      const int idx = t + GRAM_THREADS * u;
      const int m = idx / C4;
      if (m < M) tile[m * RS + idx - m * C4] = buf[u];
    }
    if (slice + GRAMM_BLOCKS < slices) fetch(slice + GRAMM_BLOCKS, buf);     // in flight while this slice is used
    __syncthreads();
    if (pr.gb && slice < g_slices) {                             // bias gradient: column sums of the staged g (rows ascending)
      const int c0 = slice * C4;
      for (int c = t; c < C4 && c0 + c < n4; c += GRAM_THREADS) {
        float4 sum = zero4;
        for (int m = 0; m < M; ++m) {
          const float4 g = tile[m * RS + c];
          sum.x += g.x; sum.y += g.y; sum.z += g.z; sum.w += g.w;
        }
        float* dst = pr.gb + 4 * (c0 + c);
        if ((reinterpret_cast<uintptr_t>(pr.gb) & 15) == 0) {
          if (pr.accumulate) { const float4 old = ldg4_global(dst); sum.x += old.x; sum.y += old.y; sum.z += old.z; sum.w += old.w; }
          stg4_global(dst, sum);
        } else {
          const float v4[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e] = pr.accumulate ? dst[e] + v4[e] : v4[e];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      if (w + GRAM_WAVES * u < T) {                                // (wave-uniform)
        const float* pa = tf + (16 * ta[u] + i) * RSf + q;
        const float* pb = tf + (16 * tb[u] + i) * RSf + q;
        f64x4 c = acc[u];
#pragma unroll 4
        for (int col = 0; col < C; col += 4)
          c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)pa[col], (double)pb[col], c, 0, 0, 0);
        acc[u] = c;
      }
    }
    // the block's g slices are done: their tiles leave as its G part (the x slices start from zero)
    const bool last_g = slice < g_slices && slice + GRAMM_BLOCKS >= g_slices;
    const bool last = slice + GRAMM_BLOCKS >= slices;
    if (last_g || last) {
      double* dst = mine + (slice < g_slices ? 0 : (size_t)launch_tiles * 256);
#pragma unroll
      for (int u = 0; u < TPW; ++u) {
        const int tt = w + GRAM_WAVES * u;
        if (tt < T) *reinterpret_cast<f64x4*>(dst + (size_t)tt * 256 + 4 * lane) = acc[u];
        acc[u] = f64x4{0.0, 0.0, 0.0, 0.0};
      }
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_gram_mfma_dot_k(const WgradProblem* __restrict__ table, double* __restrict__ ws,
                                                             int launch_tiles) {
  __shared__ double part[4];
  const WgradProblem pr = table[blockIdx.y];
  const int M = pr.M;
  double* mine = ws + (size_t)blockIdx.y * gramm_ws_doubles(launch_tiles);
  double* sums = mine + (size_t)GRAMM_BLOCKS * 2 * launch_tiles * 256;
  const int tt = blockIdx.x;
  const int NT = (M + 15) >> 4, T = NT * (NT + 1) / 2;
  if (M > GRAMM_MAX_ROWS || T > launch_tiles) { if (threadIdx.x == 0) sums[tt] = __builtin_nan(""); return; }
  if (tt >= T) { if (threadIdx.x == 0) sums[tt] = 0.0; return; }
  const int C4 = gramm_cols(16 * NT) >> 2;
  const int g_slices = ((pr.N >> 2) + C4 - 1) / C4, slices = g_slices + ((pr.K >> 2) + C4 - 1) / C4;
  int a, b;
  gramm_tile_of(tt, NT, a, b);
  double gg = 0.0, xx = 0.0;
  for (int s = 0; s < GRAMM_BLOCKS && s < slices; ++s) {
    const double* blk = mine + (size_t)s * 2 * launch_tiles * 256 + (size_t)tt * 256 + threadIdx.x;
    const int last_slice = s + (slices - 1 - s) / GRAMM_BLOCKS * GRAMM_BLOCKS;      // of block s
    if (s < g_slices) gg += blk[0];
    if (last_slice >= g_slices) xx += blk[(size_t)launch_tiles * 256];
  }
  double local = (a == b ? 1.0 : 2.0) * gg * xx;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) local += __shfl_xor(local, d);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) sums[tt] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(64) void wgrad_gram_mfma_sum_k(const double* __restrict__ ws, double* __restrict__ out, int launch_tiles) {
  if (threadIdx.x != 0) return;
  const double* sums = ws + (size_t)blockIdx.x * gramm_ws_doubles(launch_tiles) + (size_t)GRAMM_BLOCKS * 2 * launch_tiles * 256;
  double sum = 0.0;
  for (int tt = 0; tt < launch_tiles; ++tt) sum += sums[tt];
  out[blockIdx.x] = sum;
}

// ------------------------------------------------------------------ grouped weight gradient over GATHERED operands
// Data-parallel exchange of the bead-level layers (trainer.OperandExchange): a weight gradient g^T x has rank <= rows,
// and the bead-level layers see 12 rows per GPU against 0.36 - 3.2 M weights, so the ranks all-gather their operand
// rows (g = gy * act'(z) and x, packed by pack_operands_k) instead of all-reducing gW, and every rank forms the
// global gradient itself:  gW[N,K] (+)= sum over ALL ranks' rows of g[m,:]^T x[m,:]  -- what a single process would
// compute on the concatenated batch.  Row m of the problem lives in rank segment m / seg_rows of the gathered buffer:
//   g_row(m) = gy + (m / seg_rows) * seg_stride + (m % seg_rows) * N        x_row(m) likewise with K
// (seg_rows % 4 == 0 is required by the host protocol; the kernel itself takes any).
// A block owns a 64 x 64 tile of one gW; wave w its rows 16 w .. 16 w + 15.  The operand rows of the tile (64 columns
// of g, 64 of x) are staged through LDS in chunks of GW_CHUNK rows with coalesced 16-byte loads that are all in
// flight at once (the direct-from-L2 version spent 264 us at 8 x 12 rows on dependent load rounds; this one is
// bound by the gW stores).  MFMA 16x16x4 f32 steps over 4 rows: lane (i = l&15, q = l>>4) supplies
// A = g[m0+q][16w+i] and B_s = x[m0+q][4i+s], so that D_s holds gW[n0+16w+4q+r][k0+4i+s] and leaves as 16-byte
// stores.  LDS strides 80 / 64 floats keep the b32 / b128 reads conflict free.  Exact fp32 FMA chains, fixed order.
// Straight-line staging: every request goes to a valid (clamped) address and is zeroed by a select afterwards -- with
// predicated loads the compiler builds a branch and a vmcnt(0) per request.  The clobber keeps the requests above the
// MFMA loop they are meant to travel under (LLVM otherwise sinks them to their first use behind it).
__device__ __forceinline__ void strip_pin() { asm volatile("" ::: "memory"); }
// Operand pointers come out of the record (generic address space): as they are, the requests become flat_load, which
// counts on lgkmcnt as well -- the first LDS wait of the MFMA loop would then wait for the whole next x tile.
typedef const float __attribute__((address_space(1)))* strip_gptr;
__device__ __forceinline__ float4 strip_ldg4(const float* p) {
  const f32x4 t = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>((strip_gptr)p);
  return make_float4(t.x, t.y, t.z, t.w);
}
#ifndef CGV_GW_CHUNK
#define CGV_GW_CHUNK 48
#endif
// rows per staged chunk (multiple of 16); -DCGV_GW_CHUNK=<n> for A/B builds: 16 / 32 / 48 are within 3 % of each other
// (chignolin 72 / 76 / 74 us, dipeptide 365 / 360 / 371 us), 96 is 20 % slower (two blocks per CU)
constexpr int GW_CHUNK = CGV_GW_CHUNK;
constexpr int GW_GS = 80, GW_XS = 64;
// gathered_wgrad_k's tile is 64 rows x GW_TW columns of gW: the g columns of a staged chunk (with z: two thirds of the
// staged bytes of an activated layer) serve twice as many FMAs as in a 64 x 64 tile -- 8 instead of 5.3 FMAs per staged
// byte; the kernel is bound by the L2 -> LDS traffic of its four blocks per CU, not by the MFMA pipe.
constexpr int GW_TW = 128, GW_XW = 128;          // tile width in k; LDS row stride of the x chunk

// MODE: GW_STORE writes the tile (and the bias gradient); the other two are the halves of a RANK UPDATE over gathered
// rows too many for the FMA-per-row kernel (grouped_wgrad_t<true>: VALU bound from ~48 rows): GW_SUMSQ forms the tile,
// leaves its sum of squares as this block's entry of `partial` (double; summed per problem in block order by
// gathered_sumsq_reduce_k) and writes the bias gradient; GW_ADAM forms the tile again and runs it, clipped, through
// the Adam update of its weights (p / m / v addressed through gW's offset in the gradient arena) -- the gradient itself
// is never stored.
enum { GW_STORE = 0, GW_SUMSQ = 1, GW_ADAM = 2 };
template <int MODE>
__global__ __launch_bounds__(256) void gathered_wgrad_k(const WgradProblem* __restrict__ table, int n_problems,
                                                        double* __restrict__ partial, RankUpdateArgs ra) {
  __shared__ __attribute__((aligned(16))) float gs[GW_CHUNK * GW_GS];
  __shared__ __attribute__((aligned(16))) float xs[GW_CHUNK * GW_XW];
  if (MODE == GW_ADAM && ra.state[ST_SKIP] != 0.f) return;    // skipped step (utils.py:145): parameters stay
  const int lo = wg_find_problem(table, n_problems);
  const WgradProblem pr = table[lo];
  const int local = blockIdx.x - pr.block_begin;
  const int nb = local / pr.tiles_k, kt = local - nb * pr.tiles_k;
  const int M = pr.M, N = pr.N, K = pr.K;
  const int sr = pr.seg_rows > 0 ? pr.seg_rows : M;
  const int n0 = nb * 64, k0 = kt * GW_TW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc[8];                                                      // [half h of the tile's columns][component]
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const int c4 = threadIdx.x & 15, rr = threadIdx.x >> 4;          // staging: 16 float4 columns x 16 rows per pass
  const bool gcol = n0 + 4 * c4 < N, xcol0 = k0 + 4 * c4 < K, xcol1 = k0 + 64 + 4 * c4 < K;
  constexpr int NP = GW_CHUNK / 16;                                  // staging passes per chunk
  float4 gq[NP], zq[NP], xq[NP][2];
  // operand rows of one chunk into registers: the loads only -- g is multiplied by act'(z) when the chunk is stored to
  // LDS (chunk_store), so that the next chunk's loads really travel under this chunk's MFMAs.  Straight line: every
  // request goes to a valid (clamped) address through a global-address-space pointer and is zeroed by a select when it
  // is stored (a predicated load is a branch with a wait for everything outstanding; a generic-pointer load is a
  // flat_load, which also counts on lgkmcnt and made the MFMA loop's first LDS wait a wait for the whole next chunk).
  const int gcol_at = gcol ? n0 + 4 * c4 : 0, xcol0_at = xcol0 ? k0 + 4 * c4 : 0, xcol1_at = xcol1 ? k0 + 64 + 4 * c4 : 0;
  const float* zsrc = pr.act ? pr.z : pr.gy;                          // (no activation: a second look at g instead of a branch)
  auto chunk_load = [&](int m0) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int m = min(m0 + rr + 16 * p, M - 1);
      const int seg = m / sr, row = m - seg * sr;
      const size_t base = (size_t)seg * pr.seg_stride;
      gq[p] = strip_ldg4(pr.gy + base + (size_t)row * N + gcol_at);
      zq[p] = strip_ldg4(zsrc + base + (size_t)row * N + gcol_at);
      xq[p][0] = strip_ldg4(pr.x + base + (size_t)row * K + xcol0_at);
      xq[p][1] = strip_ldg4(pr.x + base + (size_t)row * K + xcol1_at);
    }
    strip_pin();
  };
  auto chunk_store = [&](int m0) {
    if (pr.act == 1) {                                                // Swish: the model's activation, kept free of the switch
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        gq[p].x *= act_bwd(zq[p].x, 1); gq[p].y *= act_bwd(zq[p].y, 1);
        gq[p].z *= act_bwd(zq[p].z, 1); gq[p].w *= act_bwd(zq[p].w, 1);
      }
    } else if (pr.act) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        gq[p].x *= act_bwd(zq[p].x, pr.act); gq[p].y *= act_bwd(zq[p].y, pr.act);
        gq[p].z *= act_bwd(zq[p].z, pr.act); gq[p].w *= act_bwd(zq[p].w, pr.act);
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {                                   // rows beyond M and columns beyond N / K: zeros
      const bool live = m0 + rr + 16 * p < M;
      const bool gk = live && gcol, xk0 = live && xcol0, xk1 = live && xcol1;
      *reinterpret_cast<float4*>(gs + (rr + 16 * p) * GW_GS + 4 * c4) =
          make_float4(gk ? gq[p].x : 0.f, gk ? gq[p].y : 0.f, gk ? gq[p].z : 0.f, gk ? gq[p].w : 0.f);
      *reinterpret_cast<float4*>(xs + (rr + 16 * p) * GW_XW + 4 * c4) =
          make_float4(xk0 ? xq[p][0].x : 0.f, xk0 ? xq[p][0].y : 0.f, xk0 ? xq[p][0].z : 0.f, xk0 ? xq[p][0].w : 0.f);
      *reinterpret_cast<float4*>(xs + (rr + 16 * p) * GW_XW + 64 + 4 * c4) =
          make_float4(xk1 ? xq[p][1].x : 0.f, xk1 ? xq[p][1].y : 0.f, xk1 ? xq[p][1].z : 0.f, xk1 ? xq[p][1].w : 0.f);
    }
  };
  chunk_load(0);
  for (int m0 = 0; m0 < M; m0 += GW_CHUNK) {
    chunk_store(m0);
    __syncthreads();
    chunk_load(min(m0 + GW_CHUNK, M - 1));                           // the next chunk travels under this chunk's MFMAs
                                                                     // (the last trip asks for the last row again: no branch)
    const float* ga = gs + q * GW_GS + 16 * wave + i;
    const float* xb = xs + q * GW_XW + 4 * i;
    // whole trip count (the rows beyond the chunk are zeros): unrolled, LDS reads issued two steps ahead of their MFMAs
#pragma unroll
    for (int st = 0; st < GW_CHUNK / 4; ++st) {
      const float a = ga[(4 * st) * GW_GS];
      const float4 b0 = *reinterpret_cast<const float4*>(xb + (4 * st) * GW_XW);
      const float4 b1 = *reinterpret_cast<const float4*>(xb + (4 * st) * GW_XW + 64);
      bsum += a;
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0.w, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1.x, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1.y, acc[5], 0, 0, 0);
      acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1.z, acc[6], 0, 0, 0);
      acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1.w, acc[7], 0, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int st = 0; st < GW_CHUNK / 4 - 2; ++st) {
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    __syncthreads();
  }
  const int n = n0 + 16 * wave + i;
  if (MODE == GW_STORE) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kcol = k0 + 64 * h + 4 * i;
      if (kcol >= K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + 16 * wave + 4 * q + r;
        if (row >= N) continue;
        float* dst = pr.gW + (size_t)row * K + kcol;
        float4 o = make_float4(acc[4 * h][r], acc[4 * h + 1][r], acc[4 * h + 2][r], acc[4 * h + 3][r]);
        if (pr.accumulate) { const float4 old = ldg4_global(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        stg4_global(dst, o);
      }
    }
  } else if (MODE == GW_SUMSQ) {
    double sq = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (k0 + 64 * h + 4 * i >= K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n0 + 16 * wave + 4 * q + r >= N) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) sq += (double)acc[4 * h + c][r] * (double)acc[4 * h + c][r];
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sq += __shfl_xor(sq, d);
    __shared__ double wave_sq[4];
    if (lane == 0) wave_sq[wave] = sq;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wave_sq[0] + wave_sq[1]) + (wave_sq[2] + wave_sq[3]);
  } else {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const AdamStep a = adam_step_of(ra.state, ra.lr, ra.beta1, ra.beta2, ra.eps);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kcol = k0 + 64 * h + 4 * i;
      if (kcol >= K) continue;
      const size_t at0 = (size_t)(pr.gW - ra.arena_g) + kcol;
      float4 pp[4], mm[4], vv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + 16 * wave + 4 * q + r;
        const size_t o = at0 + (size_t)(row < N ? row : 0) * K;
        pp[r] = *reinterpret_cast<const float4*>(ra.arena_p + o);
        const f4v tm = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(ra.arena_m + o));
        const f4v tv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(ra.arena_v + o));
        mm[r] = make_float4(tm.x, tm.y, tm.z, tm.w);
        vv[r] = make_float4(tv.x, tv.y, tv.z, tv.w);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + 16 * wave + 4 * q + r;
        if (row >= N) continue;
        const size_t o = at0 + (size_t)row * K;
        adam_elem(a, pp[r].x, acc[4 * h][r], mm[r].x, vv[r].x); adam_elem(a, pp[r].y, acc[4 * h + 1][r], mm[r].y, vv[r].y);
        adam_elem(a, pp[r].z, acc[4 * h + 2][r], mm[r].z, vv[r].z); adam_elem(a, pp[r].w, acc[4 * h + 3][r], mm[r].w, vv[r].w);
        *reinterpret_cast<float4*>(ra.arena_p + o) = pp[r];
        __builtin_nontemporal_store(f4v{mm[r].x, mm[r].y, mm[r].z, mm[r].w}, reinterpret_cast<f4v*>(ra.arena_m + o));
        __builtin_nontemporal_store(f4v{vv[r].x, vv[r].y, vv[r].z, vv[r].w}, reinterpret_cast<f4v*>(ra.arena_v + o));
      }
    }
    return;                                                         // the bias gradient was written by the GW_SUMSQ pass
  }
  if (pr.gb && kt == 0) {                                           // bias: the 4 row groups q of a step meet by shuffle
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (q == 0 && n < N) pr.gb[n] = pr.accumulate ? pr.gb[n] + bsum : bsum;
  }
}

// block partials of gathered_wgrad_k<GW_SUMSQ> -> one double per problem, summed in block order
__global__ __launch_bounds__(256) void gathered_sumsq_reduce_k(const WgradProblem* __restrict__ table, int n_problems, int total_blocks,
                                                               const double* __restrict__ partial, double* __restrict__ out) {
  __shared__ double part[256];
  const int pr = blockIdx.x;
  const int beg = table[pr].block_begin, end = pr + 1 < n_problems ? table[pr + 1].block_begin : total_blocks;
  double local = 0.0;
  for (int b = beg + (int)threadIdx.x; b < end; b += 256) local += partial[b];
  part[threadIdx.x] = local;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[pr] = part[0];
}

// STRIP variant for few operand rows (M <= 128: bead-level layers of a large bead batch, gathered rows of 4 - 8 ranks).
// gathered_wgrad_k gives every 64 x 64 tile its own block, which stages BOTH operand tiles and spends most of its short
// life on the problem lookup and the first loads (PMC at 96 rows: MFMA pipe 55 % busy in the norm pass, 39 % in the
// store pass with act').  Here a block owns a 64-row STRIP of gW: the g columns of those rows are staged (and multiplied
// by act'(z)) ONCE, then the block walks the strip's K / 64 column tiles with the x tile of the next one loading while
// the MFMAs of the current one run (two LDS buffers, one barrier per tile).  Same lane maps, same per-element FMA order
// as gathered_wgrad_k (rows ascending), so results are bit-identical to it.
constexpr int GS_MAX_ROWS = 128;
#ifndef CGV_GS_SINGLE_FROM
#define CGV_GS_SINGLE_FROM 7
#endif
// row classes (NP) from which the x tile has ONE LDS buffer (a second barrier per tile, more blocks per CU)
constexpr int GS_SINGLE_FROM = CGV_GS_SINGLE_FROM;
template <int MODE, int NP>   // NP: staging passes of 16 rows (M <= 16 NP)
__global__ __launch_bounds__(256) void gathered_wgrad_strip_k(const WgradProblem* __restrict__ table, int n_problems,
                                                              double* __restrict__ partial, RankUpdateArgs ra) {
  extern __shared__ __attribute__((aligned(16))) float strip_smem[];
  constexpr int MP = 16 * NP;
  float* gs = strip_smem;                              // [MP][GW_GS]
  constexpr bool DB = NP < GS_SINGLE_FROM;
  float* xs0 = gs + MP * GW_GS;                        // [MP][GW_XS] x 2
  float* xs1 = DB ? xs0 + MP * GW_XS : xs0;
  if (MODE == GW_ADAM && ra.state[ST_SKIP] != 0.f) return;
  const int lo = wg_find_problem(table, n_problems);
  const WgradProblem pr = table[lo];
  const int nb = blockIdx.x - pr.block_begin;
  const int M = pr.M, N = pr.N, K = pr.K;
  const int n0 = nb * 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int c4 = threadIdx.x & 15, rr = threadIdx.x >> 4;          // staging: 16 float4 columns x 16 rows per pass
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f4v __attribute__((ext_vector_type(4)));
  size_t xrow[NP], grow[NP];                                        // operand row offsets (rank segments resolved once, branch-free)
  bool live[NP];
  {
    const int seg_rows = pr.seg_rows > 0 ? pr.seg_rows : 0x7fffffff;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int m = rr + 16 * p;
      live[p] = m < M;
      const int mm = live[p] ? m : 0;
      const int seg = mm / seg_rows, in_seg = mm - seg * seg_rows;
      xrow[p] = (size_t)seg * pr.seg_stride + (size_t)in_seg * K;
      grow[p] = (size_t)seg * pr.seg_stride + (size_t)in_seg * N;
    }
  }
  const int tiles_k = (K + 63) / 64;
  float4 xq[NP];
  auto x_load = [&](int kt) {
    const int kc = kt * 64 + 4 * c4;
    const int col = kc < K ? kc : 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) xq[p] = strip_ldg4(pr.x + xrow[p] + col);
  };
  auto x_store = [&](float* xs, int kt) {
    const bool xcol = kt * 64 + 4 * c4 < K;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const bool ok = live[p] && xcol;
      *reinterpret_cast<float4*>(xs + (rr + 16 * p) * GW_XS + 4 * c4) =
          make_float4(ok ? xq[p].x : 0.f, ok ? xq[p].y : 0.f, ok ? xq[p].z : 0.f, ok ? xq[p].w : 0.f);
    }
  };
  // ---- the strip's g columns, once
  {
    const bool gcol = n0 + 4 * c4 < N;
    const int col = gcol ? n0 + 4 * c4 : 0;
    const float* zsrc = pr.act ? pr.z : pr.gy;                      // (no activation: a second look at g instead of a branch)
    float4 gq[NP], zq[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      gq[p] = strip_ldg4(pr.gy + grow[p] + col);
      zq[p] = strip_ldg4(zsrc + grow[p] + col);
    }
    x_load(0);                                                      // first x tile: arrives while act'(z) is applied
    strip_pin();
    if (pr.act == 1) {                                              // Swish: the model's activation, kept free of the switch
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        gq[p].x *= act_bwd(zq[p].x, 1); gq[p].y *= act_bwd(zq[p].y, 1);
        gq[p].z *= act_bwd(zq[p].z, 1); gq[p].w *= act_bwd(zq[p].w, 1);
      }
    } else if (pr.act) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        gq[p].x *= act_bwd(zq[p].x, pr.act); gq[p].y *= act_bwd(zq[p].y, pr.act);
        gq[p].z *= act_bwd(zq[p].z, pr.act); gq[p].w *= act_bwd(zq[p].w, pr.act);
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const bool ok = live[p] && gcol;
      *reinterpret_cast<float4*>(gs + (rr + 16 * p) * GW_GS + 4 * c4) =
          make_float4(ok ? gq[p].x : 0.f, ok ? gq[p].y : 0.f, ok ? gq[p].z : 0.f, ok ? gq[p].w : 0.f);
    }
  }
  x_store(xs0, 0);
  __syncthreads();
  double sq = 0.0;
  // Adam: p / m / v of a tile are requested ONE TILE AHEAD (two named register sets, the loop below alternates them): with
  // the requests in front of the tile's own MFMAs a block had 48 KB in flight for about a third of its time and the pass
  // ran at 3 TB/s whatever the row count (365 / 378 us at 48 / 96 rows for 46 M weights).  The last tile requests itself again.
  float4 pA[4], mA[4], vA[4], pB[4], mB[4], vB[4];
  size_t atA = 0, atB = 0;
  // (the step's constants once, in front of the loop: read per tile they put a vmcnt(0) -- every request in flight -- into each trip)
  const AdamStep a = MODE == GW_ADAM ? adam_step_of(ra.state, ra.lr, ra.beta1, ra.beta2, ra.eps) : AdamStep{};
  auto pmv_load = [&](int kt, float4 (&pp)[4], float4 (&mm)[4], float4 (&vv)[4], size_t& at0) {
    const int kc = (kt < tiles_k ? kt : tiles_k - 1) * 64 + 4 * i;
    at0 = (size_t)(pr.gW - ra.arena_g) + (kc < K ? kc : 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = n0 + 16 * wave + 4 * q + r;
      const size_t o = at0 + (size_t)(row < N ? row : 0) * K;
      pp[r] = strip_ldg4(ra.arena_p + o);
      const f4v tm = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) f4v*>((strip_gptr)(ra.arena_m + o)));
      const f4v tv = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) f4v*>((strip_gptr)(ra.arena_v + o)));
      mm[r] = make_float4(tm.x, tm.y, tm.z, tm.w);
      vv[r] = make_float4(tv.x, tv.y, tv.z, tv.w);
    }
  };
  auto tile_step = [&](int kt, float4 (&pp)[4], float4 (&mm)[4], float4 (&vv)[4], size_t at0,
                       float4 (&pn)[4], float4 (&mn)[4], float4 (&vn)[4], size_t& atn) {
    const float* xs = (kt & 1) ? xs1 : xs0;
    const int kcol = kt * 64 + 4 * i;
    const int kn = kt + 1 < tiles_k ? kt + 1 : kt;                  // (the last trip requests its own tile again: no branch)
    x_load(kn);                                                     // travels under this tile's MFMAs
    if (MODE == GW_ADAM) pmv_load(kt + 1, pn, mn, vn, atn);         // used one trip from now
    strip_pin();
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ga = gs + q * GW_GS + 16 * wave + i;
    const float* xb = xs + q * GW_XS + 4 * i;
    // whole trip count of the row class (rows M .. 16 NP - 1 are zeros in LDS): unrolled, so that the LDS reads are
    // scheduled ahead of the MFMAs that use them (a rolled loop waits for each pair of reads in front of its MFMAs)
#pragma unroll
    for (int st = 0; st < 4 * NP; ++st) {
      const float a = ga[(4 * st) * GW_GS];
      const float4 b = *reinterpret_cast<const float4*>(xb + (4 * st) * GW_XS);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.w, acc[3], 0, 0, 0);
    }
    // issue order: the reads of two steps, then { 4 MFMAs, the reads of the step after next } -- the scheduler on its own
    // reuses one register set and waits for every read right in front of its MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int st = 0; st < 4 * NP - 2; ++st) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    if (kcol < K) {
      if (MODE == GW_STORE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = n0 + 16 * wave + 4 * q + r;
          if (row >= N) continue;
          float* dst = pr.gW + (size_t)row * K + kcol;
          float4 o = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
          if (pr.accumulate) { const float4 old = ldg4_global(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
          stg4_global(dst, o);
        }
      } else if (MODE == GW_SUMSQ) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n0 + 16 * wave + 4 * q + r >= N) continue;
#pragma unroll
          for (int c = 0; c < 4; ++c) sq += (double)acc[c][r] * (double)acc[c][r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = n0 + 16 * wave + 4 * q + r;
          if (row >= N) continue;
          const size_t o = at0 + (size_t)row * K;
          adam_elem(a, pp[r].x, acc[0][r], mm[r].x, vv[r].x); adam_elem(a, pp[r].y, acc[1][r], mm[r].y, vv[r].y);
          adam_elem(a, pp[r].z, acc[2][r], mm[r].z, vv[r].z); adam_elem(a, pp[r].w, acc[3][r], mm[r].w, vv[r].w);
          *reinterpret_cast<float4*>(ra.arena_p + o) = pp[r];
          __builtin_nontemporal_store(f4v{mm[r].x, mm[r].y, mm[r].z, mm[r].w}, reinterpret_cast<f4v*>(ra.arena_m + o));
          __builtin_nontemporal_store(f4v{vv[r].x, vv[r].y, vv[r].z, vv[r].w}, reinterpret_cast<f4v*>(ra.arena_v + o));
        }
      }
    }
    if (!DB) __syncthreads();
    x_store((kt & 1) ? xs0 : xs1, kn);                             // that buffer was last read one trip ago, behind a barrier
    __syncthreads();
  };
  if (MODE == GW_ADAM) pmv_load(0, pA, mA, vA, atA);
  for (int kt = 0; kt < tiles_k; kt += 2) {
    tile_step(kt, pA, mA, vA, atA, pB, mB, vB, atB);
    if (kt + 1 < tiles_k) tile_step(kt + 1, pB, mB, vB, atB, pA, mA, vA, atA);
  }
  if (MODE == GW_SUMSQ) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sq += __shfl_xor(sq, d);
    __shared__ double wave_sq[4];
    if (lane == 0) wave_sq[wave] = sq;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wave_sq[0] + wave_sq[1]) + (wave_sq[2] + wave_sq[3]);
  }
  if (MODE != GW_ADAM && pr.gb && threadIdx.x < 64 && n0 + (int)threadIdx.x < N) {   // bias gradient: column sums of the staged strip
    float b4[4] = {0.f, 0.f, 0.f, 0.f};                            // rows 4 t + q per group q, groups paired as in the tile layout
    for (int t = 0; t < (M + 3) / 4; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) b4[g] += gs[(4 * t + g) * GW_GS + threadIdx.x];
    }
    const float bsum = (b4[0] + b4[1]) + (b4[2] + b4[3]);
    float* dst = pr.gb + n0 + threadIdx.x;
    *dst = pr.accumulate ? *dst + bsum : bsum;
  }
}

// The same with 128 x 128 output tiles (waves as a 2 x 2 grid of 64 x 64 quadrants: 4 row tiles x one 64-column group
// each): every staged operand row feeds twice the MFMAs, so the L2 -> LDS traffic per gW element halves.  Measured
// SLOWER than the 64 x 64 kernel on every workload (fewer, bigger blocks: 3 per CU; see primitives.wgrad_tile): opt-in.
constexpr int GW2_CHUNK = 32;
constexpr int GW2_GS = 144, GW2_XS = 128;

__global__ __launch_bounds__(256) void gathered_wgrad128_k(const WgradProblem* __restrict__ table, int n_problems) {
  __shared__ __attribute__((aligned(16))) float gs[GW2_CHUNK * GW2_GS];
  __shared__ __attribute__((aligned(16))) float xs[GW2_CHUNK * GW2_XS];
  const int lo = wg_find_problem(table, n_problems);
  const WgradProblem pr = table[lo];
  const int local = blockIdx.x - pr.block_begin;
  const int nb = local / pr.tiles_k, kt = local - nb * pr.tiles_k;
  const int M = pr.M, N = pr.N, K = pr.K;
  const int sr = pr.seg_rows > 0 ? pr.seg_rows : M;
  const int n0 = nb * 128, k0 = kt * 128;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave >> 1, wk = wave & 1;                          // quadrant: rows n0 + 64 wn .., columns k0 + 64 wk ..
  const int i = lane & 15, q = lane >> 4;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const int c4 = threadIdx.x & 31, rr = threadIdx.x >> 5;           // staging: 32 float4 columns x 8 rows per pass
  const bool gcol = n0 + 4 * c4 < N, xcol = k0 + 4 * c4 < K;
  constexpr int NP = GW2_CHUNK / 8;
  float4 gq[NP], xq[NP];
  auto chunk_load = [&](int m0) {
    const int rows = min(GW2_CHUNK, M - m0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int r = rr + 8 * p;
      const bool ok = r < rows;
      const int m = m0 + (ok ? r : 0);
      const int seg = m / sr, row = m - seg * sr;
      const size_t base = (size_t)seg * pr.seg_stride;
      gq[p] = ldg4_or_zero(pr.gy + base + (size_t)row * N + (gcol ? n0 + 4 * c4 : 0), ok && gcol);
      if (pr.act) {
        const float4 zz = ldg4_or_zero(pr.z + base + (size_t)row * N + (gcol ? n0 + 4 * c4 : 0), ok && gcol);
        gq[p].x *= act_bwd(zz.x, pr.act); gq[p].y *= act_bwd(zz.y, pr.act);
        gq[p].z *= act_bwd(zz.z, pr.act); gq[p].w *= act_bwd(zz.w, pr.act);
      }
      xq[p] = ldg4_or_zero(pr.x + base + (size_t)row * K + (xcol ? k0 + 4 * c4 : 0), ok && xcol);
    }
  };
  chunk_load(0);
  for (int m0 = 0; m0 < M; m0 += GW2_CHUNK) {
    const int rows = min(GW2_CHUNK, M - m0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      *reinterpret_cast<float4*>(gs + (rr + 8 * p) * GW2_GS + 4 * c4) = gq[p];
      *reinterpret_cast<float4*>(xs + (rr + 8 * p) * GW2_XS + 4 * c4) = xq[p];
    }
    __syncthreads();
    if (m0 + GW2_CHUNK < M) chunk_load(m0 + GW2_CHUNK);
    const int steps = (rows + 3) / 4;
#pragma unroll 2
    for (int st = 0; st < steps; ++st) {
      const float4 b = *reinterpret_cast<const float4*>(xs + (4 * st + q) * GW2_XS + 64 * wk + 4 * i);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float a = gs[(4 * st + q) * GW2_GS + 64 * wn + 16 * t + i];
        bsum[t] += a;
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.x, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.y, acc[t][1], 0, 0, 0);
        acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.z, acc[t][2], 0, 0, 0);
        acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.w, acc[t][3], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const int kcol = k0 + 64 * wk + 4 * i;
  if (kcol < K) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + 64 * wn + 16 * t + 4 * q + r;
        if (row >= N) continue;
        float4* dst = reinterpret_cast<float4*>(pr.gW + (size_t)row * K + kcol);
        float4 o = make_float4(acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]);
        if (pr.accumulate) { const float4 old = *dst; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *dst = o;
      }
  }
  if (pr.gb && kt == 0 && wk == 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float b = bsum[t];
      b += __shfl_xor(b, 16);
      b += __shfl_xor(b, 32);
      const int n = n0 + 64 * wn + 16 * t + i;
      if (q == 0 && n < N) pr.gb[n] = pr.accumulate ? pr.gb[n] + b : b;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same 128 x 128 tiles on the bf16 matrix path at fp32 accuracy ("split operands").  Every fp32 operand value is
// written as the EXACT sum of three bf16 numbers, x = x1 + x2 + x3 (round-to-nearest splits: |x2| <= 2^-8 |x|,
// |x3| <= 2^-16 |x|; each residual is exactly representable, so nothing is lost in the operands), and a product
// sum_m g[m] x[m] is taken as six bf16 MFMA products with fp32 accumulation,
//     g1 x1 + (g1 x2 + g2 x1) + (g1 x3 + g2 x2 + g3 x1)
// -- every bf16 x bf16 product is exact in fp32; the dropped terms (g2 x3, g3 x2, g3 x3) are below 2^-23 of the product,
// i.e. under the rounding of the fp32 accumulation itself.  v_mfma_f32_16x16x32_bf16 retires 16x the MACs per cycle of
// v_mfma_f32_16x16x4_f32, so six of them cost 3/8 of the one fp32 instruction they replace.  The fp32 kernels above
// spend 60 % of their time in the MFMA pipe on the atom-level layers (704 - 2000 operand rows); this one is bound by
// LDS traffic and the split arithmetic instead: three planes per operand mean 3 (T + C) fragment reads per 6 T C MFMAs of a
// wave tile of T x C 16-blocks (0.25 reads per MFMA at 64 x 64) against the 0.5 an LDS of 128 B/clk can deliver per MFMA
// slot, plus the staging writes -- measured 1.15 - 1.3x the fp32 kernel, not the 2.7x of the MFMA rates.
// Measured error against fp64: the same as the fp32 MFMA kernel's
// (tests/test_hip_parity.py::test_split_bf16_weight_gradients_have_fp32_accuracy).
//
// Operands are staged TRANSPOSED ([column][m], 32 rows of m per chunk = one MFMA k step) because a lane's fragment is 8
// consecutive m of one column: a thread loads one float4 of 4 consecutive rows and writes, per column and per split,
// one 8-byte group of 4 bf16.  Rows of x are stored permuted (column 4 i + c of a 64-column group at row 16 c + i) so
// that lane i of the B fragment for c reads row 16 c + i: accumulator c of a lane is then column 4 i + c -- float4 stores.
typedef __bf16 sp_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
constexpr int SP_CHUNK = 32;                 // rows of m per chunk
constexpr int SP_LD = 40;                    // bf16 per LDS row (80 bytes: 16 consecutive rows hit 16 distinct bank groups)

__device__ __forceinline__ void sp_split(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const sp_f32x2 v = {a, b};
  const sp_bf16x2 h = __builtin_convertvector(v, sp_bf16x2);
  const sp_f32x2 r = v - __builtin_convertvector(h, sp_f32x2);        // exact
  const sp_bf16x2 m = __builtin_convertvector(r, sp_bf16x2);
  const sp_f32x2 r2 = r - __builtin_convertvector(m, sp_f32x2);       // exact, at most 8 significant bits
  const sp_bf16x2 l = __builtin_convertvector(r2, sp_bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  mid = __builtin_bit_cast(unsigned, m);
  lo = __builtin_bit_cast(unsigned, l);
}
// four consecutive m of one column -> the three 8-byte groups at dst (split s at dst + s * plane)
__device__ __forceinline__ void sp_store4(unsigned short* dst, int plane, float v0, float v1, float v2, float v3) {
  unsigned h0, m0, l0, h1, m1, l1;
  sp_split(v0, v1, h0, m0, l0);
  sp_split(v2, v3, h1, m1, l1);
  *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(dst + plane) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(dst + 2 * plane) = make_uint2(l0, l1);
}

__global__ __launch_bounds__(256, 2) void wgrad_split128_k(const WgradProblem* __restrict__ table, int n_problems) {
  constexpr int PLANE = 128 * SP_LD;
  __shared__ __attribute__((aligned(16))) unsigned short gs[3 * PLANE];
  __shared__ __attribute__((aligned(16))) unsigned short xs[3 * PLANE];
  const int item = blockIdx.x;
  const int lo = wg_find_problem(table, n_problems, item);
  const WgradProblem pr = table[lo];
  const int local = item - pr.block_begin;
  const int nb = local / pr.tiles_k, kt = local - nb * pr.tiles_k;
  const int M = pr.M, N = pr.N, K = pr.K;
  const int sr = pr.seg_rows > 0 ? pr.seg_rows : M;
  const int n0 = nb * 128, k0 = kt * 128;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave >> 1, wk = wave & 1;                          // quadrant: rows n0 + 64 wn .., columns k0 + 64 wk ..
  const int i = lane & 15, q = lane >> 4;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging unit of a thread: float4 column c4 (of 32), rows 4 rq .. 4 rq + 3 of the chunk
  const int c4 = 8 * wave + (lane & 7), rq = lane >> 3;
  const bool gcol = n0 + 4 * c4 < N, xcol = k0 + 4 * c4 < K;
  const int gcol_at = gcol ? n0 + 4 * c4 : 0, xcol_at = xcol ? k0 + 4 * c4 : 0;
  const float* zsrc = pr.act ? pr.z : pr.gy;
  float4 gq[4], zq[4], xq[4];
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  auto chunk_load = [&](int m0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = min(m0 + 4 * rq + r, M - 1);
      const int seg = m / sr, row = m - seg * sr;
      const size_t base = (size_t)seg * pr.seg_stride;
      gq[r] = strip_ldg4(pr.gy + base + (size_t)row * N + gcol_at);
      zq[r] = strip_ldg4(zsrc + base + (size_t)row * N + gcol_at);
      xq[r] = strip_ldg4(pr.x + base + (size_t)row * K + xcol_at);
    }
    strip_pin();
  };
  unsigned short* gdst = gs + (4 * c4) * SP_LD + 4 * rq;
  unsigned short* xdst = xs + (64 * (c4 >> 4) + (c4 & 15)) * SP_LD + 4 * rq;          // + 16 j rows for component j
  auto chunk_store = [&](int m0) {
    if (pr.act == 1) {                                                // Swish: the model's activation, kept free of the switch
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gq[r].x *= act_bwd(zq[r].x, 1); gq[r].y *= act_bwd(zq[r].y, 1);
        gq[r].z *= act_bwd(zq[r].z, 1); gq[r].w *= act_bwd(zq[r].w, 1);
      }
    } else if (pr.act) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gq[r].x *= act_bwd(zq[r].x, pr.act); gq[r].y *= act_bwd(zq[r].y, pr.act);
        gq[r].z *= act_bwd(zq[r].z, pr.act); gq[r].w *= act_bwd(zq[r].w, pr.act);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                    // rows beyond M, columns beyond N / K: zeros
      const bool live = m0 + 4 * rq + r < M;
      if (!(live && gcol)) gq[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!(live && xcol)) xq[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bs[0] += (gq[0].x + gq[1].x) + (gq[2].x + gq[3].x);
    bs[1] += (gq[0].y + gq[1].y) + (gq[2].y + gq[3].y);
    bs[2] += (gq[0].z + gq[1].z) + (gq[2].z + gq[3].z);
    bs[3] += (gq[0].w + gq[1].w) + (gq[2].w + gq[3].w);
    sp_store4(gdst, PLANE, gq[0].x, gq[1].x, gq[2].x, gq[3].x);
    sp_store4(gdst + SP_LD, PLANE, gq[0].y, gq[1].y, gq[2].y, gq[3].y);
    sp_store4(gdst + 2 * SP_LD, PLANE, gq[0].z, gq[1].z, gq[2].z, gq[3].z);
    sp_store4(gdst + 3 * SP_LD, PLANE, gq[0].w, gq[1].w, gq[2].w, gq[3].w);
    sp_store4(xdst, PLANE, xq[0].x, xq[1].x, xq[2].x, xq[3].x);
    sp_store4(xdst + 16 * SP_LD, PLANE, xq[0].y, xq[1].y, xq[2].y, xq[3].y);
    sp_store4(xdst + 32 * SP_LD, PLANE, xq[0].z, xq[1].z, xq[2].z, xq[3].z);
    sp_store4(xdst + 48 * SP_LD, PLANE, xq[0].w, xq[1].w, xq[2].w, xq[3].w);
  };
  const unsigned short* ga = gs + (64 * wn + i) * SP_LD + 8 * q;      // + 16 t rows, + s planes
  const unsigned short* xb = xs + (64 * wk + i) * SP_LD + 8 * q;      // + 16 c rows, + s planes
  chunk_load(0);
  for (int m0 = 0; m0 < M; m0 += SP_CHUNK) {
    chunk_store(m0);
    __syncthreads();
    chunk_load(min(m0 + SP_CHUNK, M - 1));                           // the next chunk travels under this chunk's MFMAs
    sp_bf16x8 a[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) a[s][t] = *reinterpret_cast<const sp_bf16x8*>(ga + s * PLANE + 16 * t * SP_LD);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      sp_bf16x8 b[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) b[s] = *reinterpret_cast<const sp_bf16x8*>(xb + s * PLANE + 16 * c * SP_LD);
      // small terms first; four independent accumulators between two uses of one
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][t], b[2], acc[t][c], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][t], b[0], acc[t][c], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][t], b[1], acc[t][c], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][t], b[1], acc[t][c], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][t], b[0], acc[t][c], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][t], b[0], acc[t][c], 0, 0, 0);
    }
    __syncthreads();
  }
  const int kcol = k0 + 64 * wk + 4 * i;
  if (kcol < K) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + 64 * wn + 16 * t + 4 * q + r;
        if (row >= N) continue;
        float* dst = pr.gW + (size_t)row * K + kcol;
        float4 o = make_float4(acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]);
        if (pr.accumulate) { const float4 old = ldg4_global(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        stg4_global(dst, o);
      }
  }
  if (pr.gb && kt == 0) {                                           // bias: the 8 row groups of a column meet by shuffle
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float b = bs[j];
      b += __shfl_xor(b, 8);
      b += __shfl_xor(b, 16);
      b += __shfl_xor(b, 32);
      const int n = n0 + 4 * c4 + j;
      if (rq == 0 && n < N) pr.gb[n] = pr.accumulate ? pr.gb[n] + b : b;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// STRIP layout on the bf16 matrix path with split operands -- the weight gradients of the bead-level layers of a LARGE bead
// batch (33 .. 96 operand rows: dipeptide's 96 beads, the 64 beads of the 2000-atom graph, 8 ranks x 12 gathered rows).
// gathered_wgrad_strip_k above is bound by the fp32 MFMA pipe there (96 rows: 11.7 GF in 99 us = 118 TF/s for 243 MB of
// gradients that HBM takes in ~50 us); wgrad_split128_k re-derives the three bf16 terms of BOTH operand tiles in every
// 128 x 128 block and ends up no faster (154 us on the same problems).  Here the split is done where it is cheap:
//   * x (M x K, shared by all N / 64 strips of a problem) is split ONCE, by strip_xplanes_k, into its three bf16 planes,
//     laid out as the LDS images the strips stage: [k tile of 64][plane][64 rows][MP] with m contiguous (MP = M rounded up to
//     the MFMA's 32-deep step) and the tile's columns permuted (column 4 j + c at row 16 c + j: accumulator c of lane j is
//     then column 4 j + c -- float4 stores).  3 x K x MP x 2 bytes per problem (345 KB at 96 x 600), L2 resident.
//   * g = gy * act'(z) of a strip's 64 columns is staged once per block (fp32, as the fp32 strip kernel does), each lane
//     takes its A fragments -- 8 consecutive m of one column -- out of it, splits them in registers and KEEPS them for
//     the whole walk over the strip's K / 64 column tiles.
// Per tile a block then copies 3 x 64 x MP bf16 to LDS (no arithmetic), reads 36 fragments per wave and issues 72
// v_mfma_f32_16x16x32_bf16 (at 96 rows) for 64 x 64 outputs: bound by the gW stores.  Same six products per fp32 product as
// wgrad_split128_k (dropped terms below 2^-23 of a product), same accuracy class; NOT bit-identical to the fp32 kernels.
constexpr int SS_MAX_ROWS = 96;
__host__ __device__ constexpr int ss_mp(int M) { return (M + 31) / 32 * 32; }
__host__ __device__ constexpr size_t ss_plane_bytes(int M, int K) { return (size_t)((K + 63) / 64) * 3 * 64 * ss_mp(M) * 2; }

// x planes of every problem of the table: grid (max k tiles, problems); pr.pad = offset of the problem's planes in ws, in
// 256-byte units
__global__ __launch_bounds__(256) void strip_xplanes_k(const WgradProblem* __restrict__ table, unsigned short* __restrict__ ws) {
  const WgradProblem pr = table[blockIdx.y];
  const int kt = blockIdx.x;
  const int K = pr.K, M = pr.M;
  if (kt * 64 >= K) return;
  const int MP = ss_mp(M);
  unsigned short* tile = ws + (size_t)pr.pad * 128 + (size_t)kt * 3 * 64 * MP;
  const int kk = threadIdx.x & 63;                                   // column of the tile
  const int rr = 16 * (kk & 3) + (kk >> 2);                          // its row in the image
  const int col = kt * 64 + kk;
  const bool cok = col < K;
  for (int mg = threadIdx.x >> 6; mg < MP / 8; mg += 4) {            // groups of 8 consecutive m
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = 8 * mg + e;
      v[e] = (cok && m < M) ? pr.x[wg_row(pr, m, K) + col] : 0.f;
    }
    unsigned h[4], md[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) sp_split(v[2 * e], v[2 * e + 1], h[e], md[e], l[e]);
    unsigned short* dst = tile + (size_t)rr * MP + 8 * mg;
    *reinterpret_cast<uint4*>(dst) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(dst + 64 * MP) = make_uint4(md[0], md[1], md[2], md[3]);
    *reinterpret_cast<uint4*>(dst + 2 * 64 * MP) = make_uint4(l[0], l[1], l[2], l[3]);
  }
}

template <int KS>   // 32-deep steps of the reduction: M <= 32 KS
__global__ __launch_bounds__(256) void strip_split_k(const WgradProblem* __restrict__ table, int n_problems,
                                                     const unsigned short* __restrict__ ws) {
  constexpr int MP = 32 * KS;
  constexpr int XLD = MP + 8;                                        // bf16 per image row in LDS (16 bytes of padding)
  constexpr int XPLANE = 64 * XLD;
  constexpr int G_FLOATS = MP * GW_GS, X_SHORTS = 3 * XPLANE;
  constexpr int LDS_BYTES = (G_FLOATS * 4 > X_SHORTS * 2) ? G_FLOATS * 4 : X_SHORTS * 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  float* gs = reinterpret_cast<float*>(smem);                        // [MP][GW_GS] fp32 g' (first phase)
  unsigned short* xs = reinterpret_cast<unsigned short*>(smem);      // [3][64][XLD] bf16 x planes of a tile (afterwards)
  const int lo = wg_find_problem(table, n_problems);
  const WgradProblem pr = table[lo];
  const int nb = blockIdx.x - pr.block_begin;
  const int M = pr.M, N = pr.N, K = pr.K;
  const int n0 = nb * 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  // ---- the strip's g columns: staged as fp32 (rows beyond M, columns beyond N: zeros), the bias sum, the A fragments
  {
    const int c4 = threadIdx.x & 15, rr = threadIdx.x >> 4;          // 16 float4 columns x 16 rows per pass
    const bool gcol = n0 + 4 * c4 < N;
    const int col = gcol ? n0 + 4 * c4 : 0;
    const float* zsrc = pr.act ? pr.z : pr.gy;
#pragma unroll
    for (int p = 0; p < MP / 16; ++p) {
      const int m = rr + 16 * p;
      const bool ok = m < M && gcol;
      const size_t at = wg_row(pr, m < M ? m : 0, N) + col;
      float4 g4 = strip_ldg4(pr.gy + at);
      if (pr.act) {
        const float4 z4 = strip_ldg4(zsrc + at);
        g4.x *= act_bwd(z4.x, pr.act); g4.y *= act_bwd(z4.y, pr.act); g4.z *= act_bwd(z4.z, pr.act); g4.w *= act_bwd(z4.w, pr.act);
      }
      *reinterpret_cast<float4*>(gs + m * GW_GS + 4 * c4) = make_float4(ok ? g4.x : 0.f, ok ? g4.y : 0.f, ok ? g4.z : 0.f, ok ? g4.w : 0.f);
    }
  }
  __syncthreads();
  if (pr.gb) {                                                       // bias gradient: column sums (four row classes, added in order)
    const int cI = threadIdx.x & 63, cls = threadIdx.x >> 6;
    float b = 0.f;
    for (int m = cls; m < MP; m += 4) b += gs[m * GW_GS + cI];       // (rows beyond M are zeros)
    __shared__ float bias_part[4][64];
    bias_part[cls][cI] = b;
    __syncthreads();
    if (threadIdx.x < 64) {
      const int n = n0 + (int)threadIdx.x;
      const float t = (bias_part[0][cI] + bias_part[1][cI]) + (bias_part[2][cI] + bias_part[3][cI]);
      if (n < N) pr.gb[n] = pr.accumulate ? pr.gb[n] + t : t;
    }
  }
  sp_bf16x8 a[3][KS];                                                // lane (i, q): rows m = 32 ks + 8 q .. + 7 of column 16 wave + i
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    unsigned h[4], md[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* src = gs + (32 * ks + 8 * q + 2 * e) * GW_GS + 16 * wave + i;
      sp_split(src[0], src[GW_GS], h[e], md[e], l[e]);
    }
    a[0][ks] = __builtin_bit_cast(sp_bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
    a[1][ks] = __builtin_bit_cast(sp_bf16x8, make_uint4(md[0], md[1], md[2], md[3]));
    a[2][ks] = __builtin_bit_cast(sp_bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
  }
  // ---- walk over the strip's column tiles
  const int tiles_k = (K + 63) / 64;
  const unsigned short* planes = ws + (size_t)pr.pad * 128;
  const int MPp = ss_mp(M);                                          // this problem's plane rows hold MPp <= MP values of m
  constexpr int PIECES = 3 * 64 * MP / 8;                            // 16-byte pieces of a tile's LDS image
  constexpr int PER = (PIECES + 255) / 256;
  typedef unsigned su4 __attribute__((ext_vector_type(4)));
  su4 xr[PER];
  // piece pc of the image: row pc / (MP / 8) (= plane * 64 + image row), 16-byte piece pc % (MP / 8) of it; a problem with
  // fewer rows than the table's largest has shorter plane rows: the pieces beyond them are zeros (their a fragments are
  // zeros too, but what LDS holds there must not be a NaN pattern)
  auto x_load = [&](int kt) {
    const unsigned short* src = planes + (size_t)kt * 3 * 64 * MPp;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int pc = (int)threadIdx.x + 256 * u;
      const int row = pc / (MP / 8), piece = pc - row * (MP / 8);
      const bool ok = pc < PIECES && 8 * piece < MPp;
      const su4 v = *reinterpret_cast<const __attribute__((address_space(1))) su4*>(
          (strip_gptr)(const void*)(src + (ok ? row * MPp + 8 * piece : 0)));
      xr[u] = ok ? v : su4{0u, 0u, 0u, 0u};
    }
    strip_pin();
  };
  auto x_store = [&]() {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int pc = (int)threadIdx.x + 256 * u;
      if (pc < PIECES) {
        const int row = pc / (MP / 8), piece = pc - row * (MP / 8);
        *reinterpret_cast<su4*>(xs + row * XLD + 8 * piece) = xr[u];
      }
    }
  };
  x_load(0);
  const unsigned short* xb = xs + i * XLD + 8 * q;                    // + 16 c rows, + s planes, + 32 ks
  for (int kt = 0; kt < tiles_k; ++kt) {
    __syncthreads();                                                 // the fragments of g' / of the last tile are read
    x_store();
    __syncthreads();
    x_load(kt + 1 < tiles_k ? kt + 1 : kt);                          // travels under this tile's MFMAs
    f32x4v acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      sp_bf16x8 b[4][3];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) b[c][s3] = *reinterpret_cast<const sp_bf16x8*>(xb + s3 * XPLANE + 16 * c * XLD + 32 * ks);
      // small terms first (as wgrad_split128_k); the four column blocks' accumulators between two uses of one
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][ks], b[c][2], acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][ks], b[c][0], acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][ks], b[c][1], acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][ks], b[c][1], acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][ks], b[c][0], acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][ks], b[c][0], acc[c], 0, 0, 0);
    }
    const int kcol = kt * 64 + 4 * i;
    if (kcol < K) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = n0 + 16 * wave + 4 * q + r;
        if (row >= N) continue;
        float* dst = pr.gW + (size_t)row * K + kcol;
        float4 o = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        if (pr.accumulate) { const float4 old = ldg4_global(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        stg4_global(dst, o);
      }
    }
  }
}

// Packs the operands of queued weight-gradient problems into one contiguous send buffer:
//   dst_g[M,N] = gy * act'(z)      dst_x[M,K] = x        (float4 granularity; N % 4 == 0, K % 4 == 0)
struct PackProblem {        // mirrors the 64-byte host record built in python (trainer.OperandExchange)
  const float* gy;
  const float* z;           // pre-activation or NULL
  const float* x;
  float* dst_g;
  float* dst_x;
  int M, N, K, act;
  int block_begin;
  int pad[1];
};
static_assert(sizeof(PackProblem) == 64, "host/device record layout");
constexpr int PACK_F4_PER_BLOCK = 1024;

__global__ __launch_bounds__(256) void pack_operands_k(const PackProblem* __restrict__ table, int n_problems) {
  const int lo = wg_find_problem(table, n_problems);
  const PackProblem pr = table[lo];
  const int ng4 = pr.M * pr.N / 4, nx4 = pr.M * pr.K / 4;
  const int base = (blockIdx.x - pr.block_begin) * PACK_F4_PER_BLOCK;
#pragma unroll
  for (int t = 0; t < PACK_F4_PER_BLOCK / 256; ++t) {
    const int idx = base + t * 256 + threadIdx.x;
    if (idx < ng4) {
      float4 g = reinterpret_cast<const float4*>(pr.gy)[idx];
      if (pr.act) {
        const float4 zz = reinterpret_cast<const float4*>(pr.z)[idx];
        g.x *= act_bwd(zz.x, pr.act); g.y *= act_bwd(zz.y, pr.act); g.z *= act_bwd(zz.z, pr.act); g.w *= act_bwd(zz.w, pr.act);
      }
      reinterpret_cast<float4*>(pr.dst_g)[idx] = g;
    } else if (idx < ng4 + nx4) {
      reinterpret_cast<float4*>(pr.dst_x)[idx - ng4] = reinterpret_cast<const float4*>(pr.x)[idx - ng4];
    }
  }
}

// k tiling of one problem: tiles of at most 256 float4 within a 60 KiB LDS budget for the x + g tiles; among the
// admissible tile counts the one that wastes the fewest lanes -- a tile of t4 float4 columns occupies thread groups of
// 64 / 128 / 256 lanes (grouped_wgrad_t deals its 4 row passes to 256 / lanes groups), so K = 600 is cut into
// 3 x 50 columns (78 % of the lanes busy) rather than 1 x 150 (59 %).  cgv_set_option(CGV_OPT_WGRAD_TILING, 1): the widest tile.
static inline void wgrad_tiling(int M, int K, int* tiles_k, int* tile_w) {
  int max_t4 = (15360 / M - WG_BLOCK_ROWS) / 4;
  if (max_t4 > 256) max_t4 = 256;
  if (max_t4 < 1) max_t4 = 1;
  const int k4 = K / 4;
  int nt = (k4 + max_t4 - 1) / max_t4;
  const bool wide = cgv::option(CGV_OPT_WGRAD_TILING) == 1;
  if (!wide) {
    long best_cost = -1;
    int best = nt;
    for (int cand = nt; cand <= nt + 8 && cand <= k4; ++cand) {
      const int per = (k4 + cand - 1) / cand;
      const int lanes = per <= 64 ? 64 : per <= 128 ? 128 : 256;
      const long cost = (long)cand * lanes;
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cand; }
    }
    nt = best;
  }
  const int per = (k4 + nt - 1) / nt;
  *tiles_k = nt;
  *tile_w = per * 4;
}

// ------------------------------------------------------------------ Dense backward prologue (any M)
// g = gy * act'(z) (stored when g_out != NULL) and gb[n] (+)= sum_m g[m, n] in one pass: replaces the
// sigmoid / mul / add tensor ops and the separate column-sum reduction of the library-GEMM path
// (encoder Dense layers, M = atoms).  Block = 64 columns x 16 row groups, fixed summation order.
__global__ __launch_bounds__(1024) void dense_grad_prepare_k(const float* __restrict__ gy, const float* __restrict__ z,
                                                             float* __restrict__ g_out, float* __restrict__ gb, int M,
                                                             int N, int act, int accumulate) {
  __shared__ float4 red[64][16];
  const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;          // 16 float4 column groups x 64 row groups
  const int n = blockIdx.x * 64 + 4 * c4;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < N) {                                                     // N % 4 == 0
#pragma unroll 8
    for (int m = rg; m < M; m += 64) {
      const size_t at = (size_t)m * N + n;
      float4 g = *reinterpret_cast<const float4*>(gy + at);
      if (act) {
        const float4 zz = *reinterpret_cast<const float4*>(z + at);
        g.x *= act_bwd(zz.x, act); g.y *= act_bwd(zz.y, act); g.z *= act_bwd(zz.z, act); g.w *= act_bwd(zz.w, act);
        if (g_out) *reinterpret_cast<float4*>(g_out + at) = g;
      }
      sum.x += g.x; sum.y += g.y; sum.z += g.z; sum.w += g.w;
    }
  }
  red[rg][c4] = sum;
  __syncthreads();
  if (rg < 4 && n < N && gb) {                                     // thread (rg, c4) finishes column n + rg
    float t = 0.f;
#pragma unroll 16
    for (int k = 0; k < 64; ++k) t += reinterpret_cast<const float*>(&red[k][c4])[rg];
    gb[n + rg] = accumulate ? gb[n + rg] + t : t;
  }
}

// the reduction launch of up to four outputs: blockIdx.y picks (partials, output); NS slices each
struct ReduceMulti { const float* part[BI_MULTI_MAX]; float* gx[BI_MULTI_MAX]; };
__global__ __launch_bounds__(256) void skinny_bwd_input_reduce_multi_k(ReduceMulti rm, int n4, int NS) {
  const float* __restrict__ part = rm.part[blockIdx.y];
  float* __restrict__ gx = rm.gx[blockIdx.y];
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t >> 2, sub = t & 3;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* p4 = reinterpret_cast<const float4*>(part) + i;
    constexpr int RB = 8;
    for (int p0 = sub; p0 < NS; p0 += 4 * RB) {
      float4 v[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) v[u] = p4[(size_t)min(p0 + 4 * u, NS - 1) * (size_t)n4];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const bool ok = p0 + 4 * u < NS;
        acc.x += ok ? v[u].x : 0.f; acc.y += ok ? v[u].y : 0.f; acc.z += ok ? v[u].z : 0.f; acc.w += ok ? v[u].w : 0.f;
      }
    }
  }
  acc.x += __shfl_xor(acc.x, 1); acc.y += __shfl_xor(acc.y, 1); acc.z += __shfl_xor(acc.z, 1); acc.w += __shfl_xor(acc.w, 1);
  acc.x += __shfl_xor(acc.x, 2); acc.y += __shfl_xor(acc.y, 2); acc.z += __shfl_xor(acc.z, 2); acc.w += __shfl_xor(acc.w, 2);
  if (i < n4 && sub == 0) reinterpret_cast<float4*>(gx)[i] = acc;
}

// the reduction launch of a pair with two outputs: blockIdx.y picks (partials, output)
__global__ __launch_bounds__(256) void skinny_bwd_input_reduce_pair_k(const float* __restrict__ part0, const float* __restrict__ part1,
                                                                      float* __restrict__ gx0, float* __restrict__ gx1, int n4, int NS) {
  const float* part = blockIdx.y ? part1 : part0;
  float* gx = blockIdx.y ? gx1 : gx0;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t >> 2, sub = t & 3;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* p4 = reinterpret_cast<const float4*>(part) + i;
    constexpr int RB = 8;
    for (int p0 = sub; p0 < NS; p0 += 4 * RB) {
      float4 v[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) v[u] = p4[(size_t)min(p0 + 4 * u, NS - 1) * (size_t)n4];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const bool ok = p0 + 4 * u < NS;
        acc.x += ok ? v[u].x : 0.f; acc.y += ok ? v[u].y : 0.f; acc.z += ok ? v[u].z : 0.f; acc.w += ok ? v[u].w : 0.f;
      }
    }
  }
  acc.x += __shfl_xor(acc.x, 1); acc.y += __shfl_xor(acc.y, 1); acc.z += __shfl_xor(acc.z, 1); acc.w += __shfl_xor(acc.w, 1);
  acc.x += __shfl_xor(acc.x, 2); acc.y += __shfl_xor(acc.y, 2); acc.z += __shfl_xor(acc.z, 2); acc.w += __shfl_xor(acc.w, 2);
  if (i < n4 && sub == 0) reinterpret_cast<float4*>(gx)[i] = acc;
}

template <int MB>
static void launch_fwd(dim3 grid, int waves, hipStream_t st, const float* x, const float* W, const float* bias, float* y,
                       float* z, int M, int N, int K, int act) {
  if (waves >= 16) hipLaunchKernelGGL((skinny_fwd_k<MB, 16>), grid, dim3(1024), 0, st, x, W, bias, y, z, M, N, K, act);
  else if (waves >= 8) hipLaunchKernelGGL((skinny_fwd_k<MB, 8>), grid, dim3(512), 0, st, x, W, bias, y, z, M, N, K, act);
  else hipLaunchKernelGGL((skinny_fwd_k<MB, 4>), grid, dim3(256), 0, st, x, W, bias, y, z, M, N, K, act);
}
template <int MB>
static void launch_bwd_input(hipStream_t st, const float* gy, const float* z, const float* W, float* gx, float* part,
                             int M, int N, int K, int act, int KT, int NS, int rpb, const float* add = nullptr,
                             const float* z_out = nullptr, int act_out = 0) {
  hipLaunchKernelGGL((skinny_bwd_input_k<MB>), dim3(KT * NS), dim3(256), 0, st, gy, z, W, gx, part, M, N, K, act, KT, NS,
                     rpb);
  if (NS > 1) {
    const int n4 = M * K / 4;
    hipLaunchKernelGGL(skinny_bwd_input_reduce_k, dim3((4 * n4 + 255) / 256), dim3(256), 0, st, part, gx, n4, NS, add, 0ll,
                       z_out, act_out);
  }
}

template <int MODE, int NP>
static int strip_launch_np(const void* table_dev, int n_problems, int total_blocks, double* partial, RankUpdateArgs ra,
                           hipStream_t st, const char* what) {
  const size_t lds = sizeof(float) * (size_t)(16 * NP) * (GW_GS + (NP < GS_SINGLE_FROM ? 2 : 1) * GW_XS);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gathered_wgrad_strip_k<MODE, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("%s: %zu bytes of LDS: %s", what, lds, hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL((gathered_wgrad_strip_k<MODE, NP>), dim3(total_blocks), dim3(256), lds, st,
                     reinterpret_cast<const WgradProblem*>(table_dev), n_problems, partial, ra);
  return check_launch(what);
}
template <int MODE>
static int strip_launch(const void* table_dev, int n_problems, int total_blocks, int max_rows, double* partial, RankUpdateArgs ra,
                        hipStream_t st, const char* what) {
  if (max_rows <= 32) return strip_launch_np<MODE, 2>(table_dev, n_problems, total_blocks, partial, ra, st, what);
  if (max_rows <= 48) return strip_launch_np<MODE, 3>(table_dev, n_problems, total_blocks, partial, ra, st, what);
  if (max_rows <= 64) return strip_launch_np<MODE, 4>(table_dev, n_problems, total_blocks, partial, ra, st, what);
  if (max_rows <= 80) return strip_launch_np<MODE, 5>(table_dev, n_problems, total_blocks, partial, ra, st, what);
  if (max_rows <= 96) return strip_launch_np<MODE, 6>(table_dev, n_problems, total_blocks, partial, ra, st, what);
  return strip_launch_np<MODE, 8>(table_dev, n_problems, total_blocks, partial, ra, st, what);
}
}  // namespace cgv

extern "C" {

int cgv_skinny_max_rows(void) { return 64; }

int cgv_skinny_supported(int M, int N, int K) {
  return M >= 1 && M <= 64 && N >= 4 && K >= 4 && (N % 4) == 0 && (K % 4) == 0;
}

/* the forward alone takes any row count: a thread block owns 16 - 64 rows (blockIdx.y), cgv_skinny_linear_fwd picks how many */
int cgv_skinny_fwd_supported(int M, int N, int K) {
  return M >= 1 && (M + 15) / 16 <= 65535 && N >= 4 && K >= 4 && (N % 4) == 0 && (K % 4) == 0;
}

int cgv_skinny_linear_fwd(const float* x, const float* W, const float* bias, float* y, float* z, int M, int N, int K,
                          int act, void* stream) {
  CGV_REQUIRE(x && W && y, "null pointer");
  CGV_REQUIRE(act >= 0 && act <= cgv::CGV_ACT_MAX, "act must be 0 (identity), 1 (swish), 2 (tanh), 3 (relu), 4 / 5 (c + exp(z/2))");
  CGV_REQUIRE(cgv_skinny_fwd_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0, at most 65535 row blocks)");
  CGV_REQUIRE(((((uintptr_t)x | (uintptr_t)W | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)z)) & 15) == 0,
              "operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((N + 15) / 16);
  // row blocks (of 16 rows) per thread block: all of them, or -- when that leaves the chip mostly idle -- fewer, in more blocks
  const int row_blocks = (M + 15) / 16;
  // (64 bead rows, 2000-atom config: 600 x 600 8.98 -> 4.65 us, 1800 x 600 9.43 -> 6.02, 600 x 1200 13.3 -> 6.1 with one
  //  row block per thread block; 5400 x 600, 338 column blocks: 14.2 -> 16.0, left alone)
  int mb = cgv::option(CGV_OPT_SKINNY_ROWS);
  if (mb <= 0 || mb > 4) mb = (long)grid.x * row_blocks <= 512 ? 1 : 4;
  if (mb > row_blocks) mb = row_blocks;
  grid.y = (row_blocks + mb - 1) / mb;
  // enough waves to fill the chip: few blocks -> split K over more waves per block
  const long blocks = (long)grid.x * grid.y;
  const int waves = blocks >= 256 ? 4 : (blocks >= 96 ? 8 : 16);
  switch (mb) {
    case 1: cgv::launch_fwd<1>(grid, waves, st, x, W, bias, y, z, M, N, K, act); break;
    case 2: cgv::launch_fwd<2>(grid, waves, st, x, W, bias, y, z, M, N, K, act); break;
    case 3: cgv::launch_fwd<3>(grid, waves, st, x, W, bias, y, z, M, N, K, act); break;
    default: cgv::launch_fwd<4>(grid, waves, st, x, W, bias, y, z, M, N, K, act); break;
  }
  return cgv::check_launch("cgv_skinny_linear_fwd");
}

/* bwd_input alone also takes 65..128 rows (row blocks 5..8): the weight rows are split over ~300 blocks whatever M is,
 * where the tile kernel has one block per 16 x 64 outputs -- 60 blocks for 96 bead rows (dipeptide batch). */
int cgv_skinny_bwd_input_supported(int M, int N, int K) {
  return M >= 1 && M <= 128 && N >= 4 && K >= 4 && (N % 4) == 0 && (K % 4) == 0;
}

size_t cgv_skinny_bwd_input_workspace_bytes(int M, int N, int K) {
  if (!cgv_skinny_bwd_input_supported(M, N, K)) return 0;
  int KT, NS, rpb;
  cgv::bwd_input_plan(N, K, true, &KT, &NS, &rpb);
  return NS > 1 ? sizeof(float) * (size_t)NS * M * K : 0;
}

int cgv_skinny_linear_bwd_input(const float* gy, const float* z, const float* W, float* gx, int M, int N, int K, int act,
                                void* ws, size_t ws_bytes, void* stream) {
  CGV_REQUIRE(gy && W && gx, "null pointer");
  CGV_REQUIRE(act == 0 || (act >= 1 && act <= cgv::CGV_ACT_MAX && z), "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(cgv_skinny_bwd_input_supported(M, N, K), "unsupported shape (need M <= 128, N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)gx) | ((uintptr_t)W) | ((uintptr_t)ws)) & 15) == 0, "gx, W, ws must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int KT, NS, rpb;
  cgv::bwd_input_plan(N, K, ws != nullptr, &KT, &NS, &rpb);
  if (NS > 1) CGV_REQUIRE(ws_bytes >= cgv_skinny_bwd_input_workspace_bytes(M, N, K), "workspace too small");
  float* part = reinterpret_cast<float*>(ws);
  switch ((M + 15) / 16) {
    case 1: cgv::launch_bwd_input<1>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb); break;
    case 2: cgv::launch_bwd_input<2>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb); break;
    case 3: cgv::launch_bwd_input<3>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb); break;
    case 4: cgv::launch_bwd_input<4>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb); break;
    case 5:
    case 6: cgv::launch_bwd_input<6>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb); break;
    default: cgv::launch_bwd_input<8>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb); break;
  }
  return cgv::check_launch("cgv_skinny_linear_bwd_input");
}

/* gx = add + (gy * act'(z)) W: the accumulation of a second gradient of the layer's input (the input also feeds the
 * message kernel / the residual: blocks.py) rides in the reduction launch of the row-split product.  Only where that
 * launch exists (more than one row slice): CGV_E_UNSUPPORTED otherwise -- the caller then adds separately. */
int cgv_skinny_linear_bwd_input_add(const float* gy, const float* z, const float* W, const float* add, float* gx, int M, int N,
                                    int K, int act, void* ws, size_t ws_bytes, void* stream) {
  CGV_REQUIRE(gy && W && gx && add && ws, "null pointer");
  CGV_REQUIRE(act == 0 || (act >= 1 && act <= cgv::CGV_ACT_MAX && z), "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(cgv_skinny_bwd_input_supported(M, N, K), "unsupported shape (need M <= 128, N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)gx) | ((uintptr_t)W) | ((uintptr_t)ws) | ((uintptr_t)add)) & 15) == 0, "gx, W, ws, add must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int KT, NS, rpb;
  cgv::bwd_input_plan(N, K, true, &KT, &NS, &rpb);
  if (NS <= 1) { cgv::set_error("cgv_skinny_linear_bwd_input_add: one row slice, no reduction launch to carry the add"); return CGV_E_UNSUPPORTED; }
  CGV_REQUIRE(ws_bytes >= cgv_skinny_bwd_input_workspace_bytes(M, N, K), "workspace too small");
  float* part = reinterpret_cast<float*>(ws);
  switch ((M + 15) / 16) {
    case 1: cgv::launch_bwd_input<1>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, add); break;
    case 2: cgv::launch_bwd_input<2>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, add); break;
    case 3: cgv::launch_bwd_input<3>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, add); break;
    case 4: cgv::launch_bwd_input<4>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, add); break;
    case 5:
    case 6: cgv::launch_bwd_input<6>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, add); break;
    default: cgv::launch_bwd_input<8>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, add); break;
  }
  return cgv::check_launch("cgv_skinny_linear_bwd_input_add");
}

/* gx = (add + (gy * act'(z)) W) * act_out'(z_out) (add may be NULL): the reduction launch of the row-split product also
 * multiplies by the activation derivative of the layer that produced this layer's input (cgv_tile_linear_bwd_input_out for
 * few rows x very long reductions).  CGV_E_UNSUPPORTED when the product has one row slice only. */
int cgv_skinny_linear_bwd_input_out(const float* gy, const float* z, const float* W, const float* add, float* gx, int M, int N,
                                    int K, int act, const float* z_out, int act_out, void* ws, size_t ws_bytes, void* stream) {
  CGV_REQUIRE(gy && W && gx && z_out && ws, "null pointer");
  CGV_REQUIRE(act == 0 || (act >= 1 && act <= cgv::CGV_ACT_MAX && z), "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(act_out >= 1 && act_out <= cgv::CGV_ACT_MAX, "act_out must name an activation");
  CGV_REQUIRE(cgv_skinny_bwd_input_supported(M, N, K), "unsupported shape (need M <= 128, N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)gx) | ((uintptr_t)W) | ((uintptr_t)ws) | ((uintptr_t)add) | ((uintptr_t)z_out)) & 15) == 0,
              "gx, W, ws, add, z_out must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int KT, NS, rpb;
  cgv::bwd_input_plan(N, K, true, &KT, &NS, &rpb);
  if (NS <= 1) { cgv::set_error("cgv_skinny_linear_bwd_input_out: one row slice, no reduction launch to carry the epilogue"); return CGV_E_UNSUPPORTED; }
  CGV_REQUIRE(ws_bytes >= cgv_skinny_bwd_input_workspace_bytes(M, N, K), "workspace too small");
  float* part = reinterpret_cast<float*>(ws);
  switch ((M + 15) / 16) {
    case 1: cgv::launch_bwd_input<1>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, add, z_out, act_out); break;
    case 2: cgv::launch_bwd_input<2>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, add, z_out, act_out); break;
    case 3: cgv::launch_bwd_input<3>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, add, z_out, act_out); break;
    case 4: cgv::launch_bwd_input<4>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, add, z_out, act_out); break;
    case 5:
    case 6: cgv::launch_bwd_input<6>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, add, z_out, act_out); break;
    default: cgv::launch_bwd_input<8>(st, gy, z, W, gx, part, M, N, K, act, KT, NS, rpb, add, z_out, act_out); break;
  }
  return cgv::check_launch("cgv_skinny_linear_bwd_input_out");
}

/* Two backward-input products of ONE shape in one launch pair (+ one reduction launch): gx_j = (gy_j * act_j'(z_j)) W_j for
 * j = 0, 1 -- or, with gx1 == NULL, their SUM in gx0 (two layers that read the same input: the gradient accumulation is
 * part of the reduction).  M <= 64 rows.  ws: 2 x cgv_skinny_bwd_input_workspace_bytes(M, N, K). */
int cgv_pair_linear_bwd_input(const float* gy0, const float* gy1, const float* z0, const float* z1, const float* W0,
                              const float* W1, int act0, int act1, float* gx0, float* gx1, int M, int N, int K, void* ws,
                              size_t ws_bytes, void* stream) {
  CGV_REQUIRE(gy0 && gy1 && W0 && W1 && gx0 && ws, "null pointer");
  CGV_REQUIRE((act0 == 0 || (act0 >= 1 && act0 <= cgv::CGV_ACT_MAX && z0)) && (act1 == 0 || (act1 >= 1 && act1 <= cgv::CGV_ACT_MAX && z1)),
              "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(cgv_skinny_bwd_input_supported(M, N, K) && M <= 64, "unsupported shape (need M <= 64, N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)gx0) | ((uintptr_t)gx1) | ((uintptr_t)W0) | ((uintptr_t)W1) | ((uintptr_t)ws)) & 15) == 0, "16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  int KT, NS, rpb;
  cgv::bwd_input_plan(N, K, true, &KT, &NS, &rpb);
  const size_t one = sizeof(float) * (size_t)NS * M * K;
  CGV_REQUIRE(ws_bytes >= 2 * one, "workspace too small");
  float* part0 = reinterpret_cast<float*>(ws);
  float* part1 = part0 + (size_t)NS * M * K;
  // partial products always go through the workspace (also at NS == 1): the reduction launch writes / sums the outputs
  cgv::BiPair p{{gy0, gy1}, {z0, z1}, {W0, W1}, {nullptr, nullptr}, {part0, part1}, {act0, act1}};
  const dim3 grid(KT * NS, 2);
  switch ((M + 15) / 16) {
    case 1: hipLaunchKernelGGL((cgv::skinny_bwd_input_pair_k<1>), grid, dim3(256), 0, st, p, M, N, K, KT, NS, rpb); break;
    case 2: hipLaunchKernelGGL((cgv::skinny_bwd_input_pair_k<2>), grid, dim3(256), 0, st, p, M, N, K, KT, NS, rpb); break;
    case 3: hipLaunchKernelGGL((cgv::skinny_bwd_input_pair_k<3>), grid, dim3(256), 0, st, p, M, N, K, KT, NS, rpb); break;
    default: hipLaunchKernelGGL((cgv::skinny_bwd_input_pair_k<4>), grid, dim3(256), 0, st, p, M, N, K, KT, NS, rpb); break;
  }
  const int n4 = M * K / 4;
  if (gx1)
    hipLaunchKernelGGL(cgv::skinny_bwd_input_reduce_pair_k, dim3((4 * n4 + 255) / 256, 2), dim3(256), 0, st, part0, part1, gx0,
                       gx1, n4, NS);
  else        // one output: the 2 NS slices of both problems are adjacent in the workspace
    hipLaunchKernelGGL(cgv::skinny_bwd_input_reduce_k, dim3((4 * n4 + 255) / 256), dim3(256), 0, st, part0, gx0, n4, 2 * NS);
  return cgv::check_launch("cgv_pair_linear_bwd_input");
}

/* Up to four backward-input products of ONE shape in one launch + one reduction launch: gx_j = (gy_j * act_j'(z_j)) W_j.
 * ``group`` = 1: one output per problem (gx[0 .. n-1]); ``group`` = 2: problems 2o and 2o + 1 read the same input and
 * their products are SUMMED into gx[o] (n / 2 outputs) -- the accumulation is part of the reduction.  M <= 64 rows.
 * ws: n x cgv_skinny_bwd_input_workspace_bytes(M, N, K) (at least n x 4 M K bytes). */
int cgv_multi_linear_bwd_input(int n, int group, const float* const* gy, const float* const* z, const float* const* W,
                               const int* act, float* const* gx, int M, int N, int K, void* ws, size_t ws_bytes, void* stream) {
  CGV_REQUIRE(n >= 1 && n <= cgv::BI_MULTI_MAX && (group == 1 || group == 2) && n % group == 0, "1 .. 4 problems, group 1 or 2");
  CGV_REQUIRE(gy && W && act && gx && ws, "null pointer");
  CGV_REQUIRE(cgv_skinny_bwd_input_supported(M, N, K) && M <= 64, "unsupported shape (need M <= 64, N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE((((uintptr_t)ws) & 15) == 0, "16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  int KT, NS, rpb;
  cgv::bwd_input_plan(N, K, true, &KT, &NS, &rpb);
  const size_t one = (size_t)NS * M * K;
  CGV_REQUIRE(ws_bytes >= sizeof(float) * one * n, "workspace too small");
  cgv::BiPair p{};
  for (int j = 0; j < n; ++j) {
    CGV_REQUIRE(gy[j] && W[j] && ((((uintptr_t)W[j]) & 15) == 0), "null / unaligned operand");
    CGV_REQUIRE(act[j] == 0 || (act[j] >= 1 && act[j] <= cgv::CGV_ACT_MAX && z && z[j]), "act != 0 needs the saved pre-activation z");
    p.gy[j] = gy[j]; p.z[j] = z ? z[j] : nullptr; p.W[j] = W[j]; p.gx[j] = nullptr;
    p.part[j] = reinterpret_cast<float*>(ws) + one * j;      // partial products always go through the workspace
    p.act[j] = act[j];
  }
  const dim3 grid(KT * NS, n);
  switch ((M + 15) / 16) {
    case 1: hipLaunchKernelGGL((cgv::skinny_bwd_input_pair_k<1>), grid, dim3(256), 0, st, p, M, N, K, KT, NS, rpb); break;
    case 2: hipLaunchKernelGGL((cgv::skinny_bwd_input_pair_k<2>), grid, dim3(256), 0, st, p, M, N, K, KT, NS, rpb); break;
    case 3: hipLaunchKernelGGL((cgv::skinny_bwd_input_pair_k<3>), grid, dim3(256), 0, st, p, M, N, K, KT, NS, rpb); break;
    default: hipLaunchKernelGGL((cgv::skinny_bwd_input_pair_k<4>), grid, dim3(256), 0, st, p, M, N, K, KT, NS, rpb); break;
  }
  const int n4 = M * K / 4, outs = n / group;
  cgv::ReduceMulti rm{};
  for (int o = 0; o < outs; ++o) {
    CGV_REQUIRE(gx[o] && ((((uintptr_t)gx[o]) & 15) == 0), "null / unaligned output");
    rm.part[o] = reinterpret_cast<float*>(ws) + one * group * o;       // the slices of a group's problems are adjacent
    rm.gx[o] = gx[o];
  }
  hipLaunchKernelGGL(cgv::skinny_bwd_input_reduce_multi_k, dim3((4 * n4 + 255) / 256, outs), dim3(256), 0, st, rm, n4, group * NS);
  return cgv::check_launch("cgv_multi_linear_bwd_input");
}

/* Row slicing of the split product for this shape: n_slices partial matrices of slice_floats = M * K floats each. */
int cgv_skinny_bwd_input_plan(int M, int N, int K, int* n_slices, int64_t* slice_floats) {
  CGV_REQUIRE(n_slices && slice_floats, "null pointer");
  CGV_REQUIRE(cgv_skinny_bwd_input_supported(M, N, K) && M <= 64, "unsupported shape (need M <= 64, N % 4 == 0, K % 4 == 0)");
  int KT, NS, rpb;
  cgv::bwd_input_plan(N, K, true, &KT, &NS, &rpb);
  *n_slices = NS;
  *slice_floats = (int64_t)M * K;
  return 0;
}

int cgv_skinny_linear_bwd_input_slices(const float* gy_base, const float* gy_slices, int gy_n_slices, int64_t gy_slice_stride,
                                       float* g_dense, const float* z, const float* W, float* part, size_t part_bytes, int M,
                                       int N, int K, int act, void* stream) {
  CGV_REQUIRE((gy_base || gy_slices) && W && part, "null pointer");
  CGV_REQUIRE(gy_n_slices >= 0 && (gy_n_slices == 0 || (gy_slices && gy_slice_stride >= (int64_t)M * N)), "bad slices");
  CGV_REQUIRE(act == 0 || (act >= 1 && act <= cgv::CGV_ACT_MAX && z), "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(cgv_skinny_bwd_input_supported(M, N, K) && M <= 64, "unsupported shape (need M <= 64, N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)part) | ((uintptr_t)W)) & 15) == 0, "part, W must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int KT, NS, rpb;
  cgv::bwd_input_plan(N, K, true, &KT, &NS, &rpb);
  CGV_REQUIRE(part_bytes >= sizeof(float) * (size_t)NS * M * K, "partial buffer too small");
  const cgv::SliceSum gs{gy_base, gy_slices, gy_slices ? gy_n_slices : 0, gy_slice_stride};
  const bool lazy = gy_n_slices > 0 || g_dense != nullptr;
#define CGV_BI_SLICES(MBV)                                                                                          \
  if (lazy)                                                                                                         \
    hipLaunchKernelGGL((cgv::skinny_bwd_input_k<MBV, true>), dim3(KT * NS), dim3(256), 0, st, gy_base, z, W, (float*)nullptr, \
                       part, M, N, K, act, KT, NS, rpb, gs, g_dense);                                               \
  else                                                                                                              \
    hipLaunchKernelGGL((cgv::skinny_bwd_input_k<MBV, false>), dim3(KT * NS), dim3(256), 0, st, gy_base, z, W,       \
                       (float*)nullptr, part, M, N, K, act, KT, NS, rpb, gs, (float*)nullptr)
  switch ((M + 15) / 16) {
    case 1: CGV_BI_SLICES(1); break;
    case 2: CGV_BI_SLICES(2); break;
    case 3: CGV_BI_SLICES(3); break;
    default: CGV_BI_SLICES(4); break;
  }
#undef CGV_BI_SLICES
  return cgv::check_launch("cgv_skinny_linear_bwd_input_slices");
}

/* out[i] = base[i] + sum_s slices[s * stride + i] over n_floats (multiple of 4) floats: the reduction launch for a
 * slice sum whose consumer is not one of the kernels that add the slices themselves. */
int cgv_slice_sum(const float* base, const float* slices, int n_slices, int64_t slice_stride, float* out, int64_t n_floats,
                  void* stream) {
  CGV_REQUIRE(slices && out && n_slices >= 1 && n_floats >= 0, "bad argument");
  CGV_REQUIRE((n_floats % 4) == 0 && (slice_stride % 4) == 0 && slice_stride >= n_floats, "need multiples of 4 floats");
  CGV_REQUIRE(((((uintptr_t)base) | ((uintptr_t)slices) | ((uintptr_t)out)) & 15) == 0, "16-byte alignment");
  if (n_floats == 0) return 0;
  const int n4 = (int)(n_floats / 4);
  hipLaunchKernelGGL(cgv::skinny_bwd_input_reduce_k, dim3((4 * n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, slices, out,
                     n4, n_slices, base, (long long)(slice_stride / 4));
  return cgv::check_launch("cgv_slice_sum");
}

int cgv_dense_grad_prepare(const float* gy, const float* z, float* g_out, float* gb, int M, int N, int act, int accumulate,
                           void* stream) {
  CGV_REQUIRE(gy && M >= 0 && N > 0, "bad argument");
  CGV_REQUIRE(act == 0 || (act >= 1 && act <= cgv::CGV_ACT_MAX && z), "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(gb || (act && g_out), "nothing to compute");
  CGV_REQUIRE((N % 4) == 0 && ((((uintptr_t)gy | (uintptr_t)z | (uintptr_t)g_out)) & 15) == 0, "need N % 4 == 0, 16-byte aligned");
  if (M == 0 && !gb) return 0;
  hipLaunchKernelGGL(cgv::dense_grad_prepare_k, dim3((N + 63) / 64), dim3(1024), 0, (hipStream_t)stream, gy, z, g_out, gb, M,
                     N, act, accumulate);
  return cgv::check_launch("cgv_dense_grad_prepare");
}

int cgv_wgrad_record_bytes(void) { return (int)sizeof(cgv::WgradProblem); }

/* Fills tiles_k / tile_w / block count of one problem (host helper for building the table). */
int cgv_wgrad_plan(int M, int N, int K, int* tiles_k, int* tile_w, int* n_blocks) {
  CGV_REQUIRE(tiles_k && tile_w && n_blocks, "null pointer");
  CGV_REQUIRE(cgv_skinny_supported(M, N, K), "unsupported shape (need M <= 64, N % 4 == 0, K % 4 == 0)");
  cgv::wgrad_tiling(M, K, tiles_k, tile_w);
  *n_blocks = ((N + cgv::WG_BLOCK_ROWS - 1) / cgv::WG_BLOCK_ROWS) * *tiles_k;
  return 0;
}

int cgv_wgrad_lds_floats(int M, int tile_w) { return M * (tile_w + cgv::WG_BLOCK_ROWS); }

int cgv_grouped_wgrad(const void* table_dev, int n_problems, int total_blocks, int max_lds_floats, void* stream) {
  CGV_REQUIRE(n_problems >= 0 && total_blocks >= 0, "bad size");
  if (n_problems == 0 || total_blocks == 0) return 0;
  CGV_REQUIRE(table_dev, "null table");
  CGV_REQUIRE(max_lds_floats > 0 && max_lds_floats <= 16000, "LDS request out of range");
  hipLaunchKernelGGL(cgv::grouped_wgrad_t<false>, dim3(total_blocks), dim3(256), sizeof(float) * (size_t)max_lds_floats,
                     (hipStream_t)stream, reinterpret_cast<const cgv::WgradProblem*>(table_dev), n_problems,
                     cgv::RankUpdateArgs{});
  return cgv::check_launch("cgv_grouped_wgrad");
}

/* Rank-update layers, first half: sumsq[i] = ||gW_i||_F^2 from the operands of record i; bias gradients written.
 * Records must satisfy cgv_rank_update_supported (M <= max_rows <= 64; records may address gathered operands through
 * seg_rows / seg_stride); workspace: cgv_wgrad_gram_workspace_bytes(n_problems). */
int cgv_wgrad_gram(const void* table_dev, int n_problems, int max_rows, double* sumsq, void* workspace, size_t workspace_bytes,
                   void* stream) {
  CGV_REQUIRE(n_problems >= 0, "bad size");
  if (n_problems == 0) return 0;
  CGV_REQUIRE(max_rows >= 1 && max_rows <= cgv::GRAM_MAX_ROWS, "max_rows out of range (1..64)");
  CGV_REQUIRE(table_dev && sumsq && workspace, "null pointer");
  CGV_REQUIRE(workspace_bytes >= cgv_wgrad_gram_workspace_bytes(n_problems), "workspace too small");
  CGV_REQUIRE((((uintptr_t)workspace) & 7) == 0, "workspace must be 8-byte aligned");
  const cgv::WgradProblem* table = reinterpret_cast<const cgv::WgradProblem*>(table_dev);
  const size_t lds = cgv::gram_lds_bytes(max_rows);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cgv::wgrad_gram_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { cgv::set_error("cgv_wgrad_gram: %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return (int)e; }
  }
  // one launch (the last slice block of a problem sums its slices) up to GRAM_TICKETS problems, else the reduce launch follows
  const bool one = n_problems <= cgv::GRAM_TICKETS && cgv::option(CGV_OPT_OPTIM_ONE_LAUNCH) != 2;
  hipLaunchKernelGGL(cgv::wgrad_gram_k, dim3(cgv::GRAM_SLICES, n_problems), dim3(cgv::GRAM_THREADS), lds,
                     (hipStream_t)stream, table, reinterpret_cast<double*>(workspace), max_rows * (max_rows + 1) / 2,
                     one ? sumsq : (double*)nullptr);
  if (!one)
    hipLaunchKernelGGL(cgv::wgrad_gram_reduce_k, dim3(n_problems), dim3(256), 0, (hipStream_t)stream, table,
                       reinterpret_cast<const double*>(workspace), sumsq);
  return cgv::check_launch("cgv_wgrad_gram");
}

size_t cgv_wgrad_gram_workspace_bytes(int n_problems) {
  return n_problems > 0 ? (size_t)n_problems * cgv::GRAM_WS_DOUBLES * sizeof(double) : 0;
}

/* The same for records of up to 128 rows (cgv_wgrad_gram_mfma_max_rows): Gram matrices by fp64 MFMA tiles (csrc:
 * wgrad_gram_mfma_k).  N % 4 == 0 and K % 4 == 0; workspace: cgv_wgrad_gram_mfma_workspace_bytes(n_problems, max_rows). */
int cgv_wgrad_gram_mfma_max_rows(void) { return cgv::GRAMM_MAX_ROWS; }
size_t cgv_wgrad_gram_mfma_workspace_bytes(int n_problems, int max_rows) {
  if (n_problems <= 0 || max_rows < 1 || max_rows > cgv::GRAMM_MAX_ROWS) return 0;
  const int nt = (max_rows + 15) / 16;
  return (size_t)n_problems * cgv::gramm_ws_doubles(nt * (nt + 1) / 2) * sizeof(double);
}
int cgv_wgrad_gram_mfma(const void* table_dev, int n_problems, int max_rows, double* sumsq, void* workspace, size_t workspace_bytes,
                        void* stream) {
  CGV_REQUIRE(n_problems >= 0, "bad size");
  if (n_problems == 0) return 0;
  CGV_REQUIRE(max_rows >= 1 && max_rows <= cgv::GRAMM_MAX_ROWS, "max_rows out of range (1..128)");
  CGV_REQUIRE(table_dev && sumsq && workspace, "null pointer");
  CGV_REQUIRE(workspace_bytes >= cgv_wgrad_gram_mfma_workspace_bytes(n_problems, max_rows), "workspace too small");
  CGV_REQUIRE((((uintptr_t)workspace) & 31) == 0, "workspace must be 32-byte aligned");
  const cgv::WgradProblem* table = reinterpret_cast<const cgv::WgradProblem*>(table_dev);
  const int nt = (max_rows + 15) / 16, tiles = nt * (nt + 1) / 2;
  const bool few = tiles <= 3 * cgv::GRAM_WAVES;
  const void* fn = few ? reinterpret_cast<const void*>(cgv::wgrad_gram_mfma_k<3>) : reinterpret_cast<const void*>(cgv::wgrad_gram_mfma_k<5>);
  static bool lds_set[2] = {false, false};
  if (!lds_set[few]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cgv::GRAMM_LDS_BYTES);
    if (e != hipSuccess) { cgv::set_error("cgv_wgrad_gram_mfma: LDS request: %s", hipGetErrorString(e)); return (int)e; }
    lds_set[few] = true;
  }
  hipStream_t st = (hipStream_t)stream;
  double* ws = reinterpret_cast<double*>(workspace);
  const dim3 grid(cgv::GRAMM_BLOCKS, n_problems);
  if (few) hipLaunchKernelGGL(cgv::wgrad_gram_mfma_k<3>, grid, dim3(cgv::GRAM_THREADS), cgv::GRAMM_LDS_BYTES, st, table, ws, tiles);
  else hipLaunchKernelGGL(cgv::wgrad_gram_mfma_k<5>, grid, dim3(cgv::GRAM_THREADS), cgv::GRAMM_LDS_BYTES, st, table, ws, tiles);
  hipLaunchKernelGGL(cgv::wgrad_gram_mfma_dot_k, dim3(tiles, n_problems), dim3(256), 0, st, table, ws, tiles);
  hipLaunchKernelGGL(cgv::wgrad_gram_mfma_sum_k, dim3(n_problems), dim3(64), 0, st, ws, sumsq, tiles);
  return cgv::check_launch("cgv_wgrad_gram_mfma");
}

/* Shapes the rank update takes: the weight-streaming tiling (cgv_skinny_supported, at most 64 operand rows). */
int cgv_rank_update_supported(int M, int N, int K) { return cgv_skinny_supported(M, N, K) && M <= cgv::GRAM_MAX_ROWS; }

/* Rank-update layers, second half: the table of cgv_grouped_wgrad, but every gW tile goes through the clipped Adam
 * update of its weights (state from cgv_optim_prepare_extra) instead of being stored.  Every gW must lie inside the
 * gradient arena [arena_g, arena_g + arena_floats); p / m / v are the arenas of the same layout; accumulate must be 0. */
int cgv_grouped_wgrad_adam(const void* table_dev, int n_problems, int total_blocks, int max_lds_floats,
                           const float* arena_g, float* arena_p, float* arena_m, float* arena_v, float lr, float beta1,
                           float beta2, float eps, const float* state, void* stream) {
  CGV_REQUIRE(n_problems >= 0 && total_blocks >= 0, "bad size");
  if (n_problems == 0 || total_blocks == 0) return 0;
  CGV_REQUIRE(table_dev && arena_g && arena_p && arena_m && arena_v && state, "null pointer");
  CGV_REQUIRE(max_lds_floats > 0 && max_lds_floats <= 16000, "LDS request out of range");
  CGV_REQUIRE(((((uintptr_t)arena_g | (uintptr_t)arena_p | (uintptr_t)arena_m | (uintptr_t)arena_v)) & 15) == 0,
              "arenas must be 16-byte aligned");
  hipLaunchKernelGGL(cgv::grouped_wgrad_t<true>, dim3(total_blocks), dim3(256), sizeof(float) * (size_t)max_lds_floats,
                     (hipStream_t)stream, reinterpret_cast<const cgv::WgradProblem*>(table_dev), n_problems,
                     cgv::RankUpdateArgs{arena_g, arena_p, arena_m, arena_v, state, lr, beta1, beta2, eps});
  return cgv::check_launch("cgv_grouped_wgrad_adam");
}

/* The same update with the FLAT block layout (rank_update_flat_body: every block a contiguous range of a weight's p / m / v,
 * whole lines, read and written once).  cgv_rank_flat_plan: blocks + LDS floats of one record, non-zero when the shape
 * does not take it (more than 16 operand rows, or x [M, K] + the g rows beyond the LDS budget) -- then the whole launch
 * stays with cgv_grouped_wgrad_adam.  The records' block_begin must be the prefix of THESE block counts; tiles_k / tile_w
 * are unused.  q4: float4 per block, a multiple of 2048 (0: the default). */
int cgv_rank_flat_quantum(void) { return 2 * cgv::RF_QUANTUM_F4; }

int cgv_rank_flat_plan(int M, int N, int K, int q4, int* n_blocks, int* lds_floats) {
  CGV_REQUIRE(n_blocks && lds_floats, "null pointer");
  if (q4 <= 0) q4 = cgv_rank_flat_quantum();
  CGV_REQUIRE(q4 % cgv::RF_QUANTUM_F4 == 0, "q4 must be a multiple of 2048 float4");
  if (!cgv_rank_update_supported(M, N, K) || M > cgv::RF_MAX_ROWS) return CGV_E_UNSUPPORTED;
  const int K4 = K / 4;
  const int g_rows = q4 / K4 + 2 < N ? q4 / K4 + 2 : N;
  const long lds = (long)M * (K + g_rows);
  if (lds > 16000 || (long)N * K4 > 0x7fffffffL - q4) return CGV_E_UNSUPPORTED;
  *n_blocks = (int)(((long)N * K4 + q4 - 1) / q4);
  *lds_floats = (int)lds;
  return 0;
}

int cgv_grouped_wgrad_adam_flat(const void* table_dev, int n_problems, int total_blocks, int max_lds_floats, int q4,
                                const float* arena_g, float* arena_p, float* arena_m, float* arena_v, float lr, float beta1,
                                float beta2, float eps, const float* state, void* stream) {
  CGV_REQUIRE(n_problems >= 0 && total_blocks >= 0, "bad size");
  if (n_problems == 0 || total_blocks == 0) return 0;
  if (q4 <= 0) q4 = cgv_rank_flat_quantum();
  CGV_REQUIRE(q4 % cgv::RF_QUANTUM_F4 == 0, "q4 must be a multiple of 2048 float4");
  CGV_REQUIRE(table_dev && arena_g && arena_p && arena_m && arena_v && state, "null pointer");
  CGV_REQUIRE(max_lds_floats > 0 && max_lds_floats <= 16000, "LDS request out of range");
  CGV_REQUIRE(((((uintptr_t)arena_g | (uintptr_t)arena_p | (uintptr_t)arena_m | (uintptr_t)arena_v)) & 15) == 0,
              "arenas must be 16-byte aligned");
  hipLaunchKernelGGL(cgv::rank_update_mixed_k, dim3(total_blocks), dim3(256), sizeof(float) * (size_t)max_lds_floats,
                     (hipStream_t)stream, reinterpret_cast<const cgv::WgradProblem*>(table_dev), n_problems, n_problems,
                     cgv::RankUpdateArgs{arena_g, arena_p, arena_m, arena_v, state, lr, beta1, beta2, eps}, q4, 0);
  return cgv::check_launch("cgv_grouped_wgrad_adam_flat");
}

/* One launch for BOTH layouts: records [0, n_flat) of the table flat (flat_blocks blocks, prefix from 0), records
 * [n_flat, n_problems) tiled as for cgv_grouped_wgrad_adam (tiled_blocks blocks, their own prefix from 0).  The tiled
 * blocks -- layers of more operand rows, bound by forming the tile rather than by p / m / v -- are dealt evenly among the
 * flat ones.  max_lds_floats: the larger of the two layouts' requests. */
int cgv_grouped_wgrad_adam_mixed(const void* table_dev, int n_flat, int n_problems, int flat_blocks, int tiled_blocks,
                                 int max_lds_floats, int q4, const float* arena_g, float* arena_p, float* arena_m,
                                 float* arena_v, float lr, float beta1, float beta2, float eps, const float* state,
                                 void* stream) {
  CGV_REQUIRE(n_flat >= 0 && n_problems >= n_flat && flat_blocks >= 0 && tiled_blocks >= 0, "bad size");
  CGV_REQUIRE((n_flat > 0) == (flat_blocks > 0) && (n_problems > n_flat) == (tiled_blocks > 0), "records and blocks disagree");
  if (n_problems == 0) return 0;
  if (q4 <= 0) q4 = cgv_rank_flat_quantum();
  CGV_REQUIRE(q4 % cgv::RF_QUANTUM_F4 == 0, "q4 must be a multiple of 2048 float4");
  CGV_REQUIRE(table_dev && arena_g && arena_p && arena_m && arena_v && state, "null pointer");
  CGV_REQUIRE(max_lds_floats > 0 && max_lds_floats <= 16000, "LDS request out of range");
  CGV_REQUIRE(((((uintptr_t)arena_g | (uintptr_t)arena_p | (uintptr_t)arena_m | (uintptr_t)arena_v)) & 15) == 0,
              "arenas must be 16-byte aligned");
  const cgv::RankUpdateArgs ra{arena_g, arena_p, arena_m, arena_v, state, lr, beta1, beta2, eps};
  const cgv::WgradProblem* table = reinterpret_cast<const cgv::WgradProblem*>(table_dev);
  if (n_flat == 0)
    hipLaunchKernelGGL(cgv::grouped_wgrad_t<true>, dim3(tiled_blocks), dim3(256), sizeof(float) * (size_t)max_lds_floats,
                       (hipStream_t)stream, table, n_problems, ra);
  else
    hipLaunchKernelGGL(cgv::rank_update_mixed_k, dim3(flat_blocks + tiled_blocks), dim3(256), sizeof(float) * (size_t)max_lds_floats,
                       (hipStream_t)stream, table, n_flat, n_problems, ra, q4, tiled_blocks);
  return cgv::check_launch("cgv_grouped_wgrad_adam_mixed");
}

/* Weight gradients over gathered operand rows (include/cgvae_hip.h: data-parallel operand exchange). */
int cgv_wgrad_gathered_plan(int M, int N, int K, int seg_rows, int* tiles_k, int* n_blocks) {
  return cgv_wgrad_gathered_plan_tile(M, N, K, seg_rows, 64, tiles_k, n_blocks);
}

/* tile = 64 or 128: output tile edge of the launch (one value per launch, see cgv_grouped_wgrad_gathered_tile) */
int cgv_wgrad_gathered_plan_tile(int M, int N, int K, int seg_rows, int tile, int* tiles_k, int* n_blocks) {
  CGV_REQUIRE(tiles_k && n_blocks, "null pointer");
  CGV_REQUIRE(M >= 1 && N >= 4 && K >= 4 && (N % 4) == 0 && (K % 4) == 0, "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(seg_rows == 0 || (seg_rows > 0 && seg_rows % 4 == 0), "rank segments must hold a multiple of 4 rows");
  CGV_REQUIRE(tile == 64 || tile == 128, "tile must be 64 or 128");
  const int tile_k = tile == 64 ? cgv::GW_TW : tile;        // (the 64-row tiles are GW_TW = 128 columns wide)
  *tiles_k = (K + tile_k - 1) / tile_k;
  *n_blocks = ((N + tile - 1) / tile) * *tiles_k;
  return 0;
}

/* The grouped weight gradients of cgv_grouped_wgrad_gathered_tile(tile = 128: same table, same plan) on the bf16 matrix
 * path with split operands (three bf16 terms per fp32 value, six products, fp32 accumulation: fp32-class accuracy at
 * 3/8 of the fp32 MFMA time; wgrad_split128_k). */
int cgv_grouped_wgrad_split(const void* table_dev, int n_problems, int total_blocks, void* stream) {
  CGV_REQUIRE(n_problems >= 0 && total_blocks >= 0, "bad size");
  if (n_problems == 0 || total_blocks == 0) return 0;
  CGV_REQUIRE(table_dev, "null table");
  hipLaunchKernelGGL(cgv::wgrad_split128_k, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const cgv::WgradProblem*>(table_dev), n_problems);
  return cgv::check_launch("cgv_grouped_wgrad_split");
}

/* Strip layout on the bf16 matrix path with split operands (strip_split_k): records as cgv_grouped_wgrad_strip's with at
 * most cgv_wgrad_strip_split_max_rows() rows, `pad` = offset of the problem's x planes in ws in 256-byte units (each problem
 * needs cgv_wgrad_strip_split_plane_bytes(M, K), rounded up to 256).  Two launches: the x planes of every problem, the strips. */
int cgv_wgrad_strip_split_max_rows(void) { return cgv::SS_MAX_ROWS; }
size_t cgv_wgrad_strip_split_plane_bytes(int M, int K) {
  if (M < 1 || M > cgv::SS_MAX_ROWS || K < 4) return 0;
  return (cgv::ss_plane_bytes(M, K) + 255) & ~(size_t)255;
}
int cgv_grouped_wgrad_strip_split(const void* table_dev, int n_problems, int total_blocks, int max_rows, int max_k, void* ws,
                                  size_t ws_bytes, void* stream) {
  CGV_REQUIRE(n_problems >= 0 && total_blocks >= 0, "bad size");
  if (n_problems == 0 || total_blocks == 0) return 0;
  CGV_REQUIRE(table_dev && ws && max_rows >= 1 && max_rows <= cgv::SS_MAX_ROWS && max_k >= 4, "bad argument");
  CGV_REQUIRE((((uintptr_t)ws) & 255) == 0 && ws_bytes > 0, "workspace must be 256-byte aligned");
  CGV_REQUIRE(n_problems <= 65535, "too many problems");
  hipStream_t st = (hipStream_t)stream;
  const cgv::WgradProblem* table = reinterpret_cast<const cgv::WgradProblem*>(table_dev);
  hipLaunchKernelGGL(cgv::strip_xplanes_k, dim3((max_k + 63) / 64, n_problems), dim3(256), 0, st, table,
                     reinterpret_cast<unsigned short*>(ws));
  if (int rc = cgv::check_launch("cgv_grouped_wgrad_strip_split (planes)")) return rc;
  const unsigned short* planes = reinterpret_cast<const unsigned short*>(ws);
  if (max_rows <= 32) hipLaunchKernelGGL(cgv::strip_split_k<1>, dim3(total_blocks), dim3(256), 0, st, table, n_problems, planes);
  else if (max_rows <= 64) hipLaunchKernelGGL(cgv::strip_split_k<2>, dim3(total_blocks), dim3(256), 0, st, table, n_problems, planes);
  else hipLaunchKernelGGL(cgv::strip_split_k<3>, dim3(total_blocks), dim3(256), 0, st, table, n_problems, planes);
  return cgv::check_launch("cgv_grouped_wgrad_strip_split");
}

int cgv_grouped_wgrad_gathered(const void* table_dev, int n_problems, int total_blocks, void* stream) {
  return cgv_grouped_wgrad_gathered_tile(table_dev, n_problems, total_blocks, 64, stream);
}

int cgv_grouped_wgrad_gathered_tile(const void* table_dev, int n_problems, int total_blocks, int tile, void* stream) {
  CGV_REQUIRE(n_problems >= 0 && total_blocks >= 0, "bad size");
  CGV_REQUIRE(tile == 64 || tile == 128, "tile must be 64 or 128");
  if (n_problems == 0 || total_blocks == 0) return 0;
  CGV_REQUIRE(table_dev, "null table");
  if (tile == 128)
    hipLaunchKernelGGL(cgv::gathered_wgrad128_k, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const cgv::WgradProblem*>(table_dev), n_problems);
  else
    hipLaunchKernelGGL(cgv::gathered_wgrad_k<cgv::GW_STORE>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const cgv::WgradProblem*>(table_dev), n_problems, (double*)nullptr, cgv::RankUpdateArgs{});
  return cgv::check_launch("cgv_grouped_wgrad_gathered");
}

/* Rank update over (gathered) operand rows with MFMA tiles, for row counts beyond the FMA-per-row kernel's range
 * (cgv_grouped_wgrad_adam): the table and plan of cgv_grouped_wgrad_gathered (tile 64), accumulate = 0.
 *   _sumsq: sumsq[i] = ||gW_i||_F^2 for record i (tiles formed, squared, never stored; block partials in `partial`,
 *           total_blocks doubles) and the bias gradients written;
 *   _adam:  the tiles formed again and run through the clipped Adam update of their weights (state from
 *           cgv_optim_prepare_extra with those norms); every gW must lie inside the gradient arena. */
int cgv_grouped_wgrad_gathered_sumsq(const void* table_dev, int n_problems, int total_blocks, double* partial, double* sumsq,
                                     void* stream) {
  CGV_REQUIRE(n_problems >= 0 && total_blocks >= 0, "bad size");
  if (n_problems == 0 || total_blocks == 0) return 0;
  CGV_REQUIRE(table_dev && partial && sumsq, "null pointer");
  const cgv::WgradProblem* table = reinterpret_cast<const cgv::WgradProblem*>(table_dev);
  hipLaunchKernelGGL(cgv::gathered_wgrad_k<cgv::GW_SUMSQ>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table,
                     n_problems, partial, cgv::RankUpdateArgs{});
  hipLaunchKernelGGL(cgv::gathered_sumsq_reduce_k, dim3(n_problems), dim3(256), 0, (hipStream_t)stream, table, n_problems,
                     total_blocks, partial, sumsq);
  return cgv::check_launch("cgv_grouped_wgrad_gathered_sumsq");
}

int cgv_grouped_wgrad_gathered_adam(const void* table_dev, int n_problems, int total_blocks, const float* arena_g,
                                    float* arena_p, float* arena_m, float* arena_v, float lr, float beta1, float beta2,
                                    float eps, const float* state, void* stream) {
  CGV_REQUIRE(n_problems >= 0 && total_blocks >= 0, "bad size");
  if (n_problems == 0 || total_blocks == 0) return 0;
  CGV_REQUIRE(table_dev && arena_g && arena_p && arena_m && arena_v && state, "null pointer");
  CGV_REQUIRE(((((uintptr_t)arena_g | (uintptr_t)arena_p | (uintptr_t)arena_m | (uintptr_t)arena_v)) & 15) == 0,
              "arenas must be 16-byte aligned");
  hipLaunchKernelGGL(cgv::gathered_wgrad_k<cgv::GW_ADAM>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const cgv::WgradProblem*>(table_dev), n_problems, (double*)nullptr,
                     cgv::RankUpdateArgs{arena_g, arena_p, arena_m, arena_v, state, lr, beta1, beta2, eps});
  return cgv::check_launch("cgv_grouped_wgrad_gathered_adam");
}

/* The strip layout of the gathered launches for problems of at most 128 operand rows (csrc: gathered_wgrad_strip_k): a
 * block per 64 ROWS of gW (it walks that strip's column tiles itself).  Plan: n_blocks = ceil(N / 64); records as for
 * cgv_grouped_wgrad_gathered with block_begin counted in these blocks.  Results are bit-identical to the tile layout. */
int cgv_wgrad_strip_max_rows(void) { return cgv::GS_MAX_ROWS; }
int cgv_wgrad_strip_plan(int M, int N, int K, int seg_rows, int* n_blocks) {
  CGV_REQUIRE(n_blocks, "null pointer");
  CGV_REQUIRE(M >= 1 && M <= cgv::GS_MAX_ROWS && N >= 4 && K >= 4 && (N % 4) == 0 && (K % 4) == 0, "unsupported shape (need M <= 128, N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(seg_rows == 0 || (seg_rows > 0 && seg_rows % 4 == 0), "rank segments must hold a multiple of 4 rows");
  *n_blocks = (N + 63) / 64;
  return 0;
}

int cgv_grouped_wgrad_strip(const void* table_dev, int n_problems, int total_blocks, int max_rows, void* stream) {
  CGV_REQUIRE(n_problems >= 0 && total_blocks >= 0, "bad size");
  if (n_problems == 0 || total_blocks == 0) return 0;
  CGV_REQUIRE(table_dev && max_rows >= 1 && max_rows <= cgv::GS_MAX_ROWS, "bad argument");
  return cgv::strip_launch<cgv::GW_STORE>(table_dev, n_problems, total_blocks, max_rows, nullptr, cgv::RankUpdateArgs{},
                                          (hipStream_t)stream, "cgv_grouped_wgrad_strip");
}

int cgv_grouped_wgrad_strip_sumsq(const void* table_dev, int n_problems, int total_blocks, int max_rows, double* partial,
                                  double* sumsq, void* stream) {
  CGV_REQUIRE(n_problems >= 0 && total_blocks >= 0, "bad size");
  if (n_problems == 0 || total_blocks == 0) return 0;
  CGV_REQUIRE(table_dev && partial && sumsq && max_rows >= 1 && max_rows <= cgv::GS_MAX_ROWS, "bad argument");
  if (int rc = cgv::strip_launch<cgv::GW_SUMSQ>(table_dev, n_problems, total_blocks, max_rows, partial, cgv::RankUpdateArgs{},
                                                (hipStream_t)stream, "cgv_grouped_wgrad_strip_sumsq")) return rc;
  hipLaunchKernelGGL(cgv::gathered_sumsq_reduce_k, dim3(n_problems), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const cgv::WgradProblem*>(table_dev), n_problems, total_blocks, partial, sumsq);
  return cgv::check_launch("cgv_grouped_wgrad_strip_sumsq");
}

int cgv_grouped_wgrad_strip_adam(const void* table_dev, int n_problems, int total_blocks, int max_rows, const float* arena_g,
                                 float* arena_p, float* arena_m, float* arena_v, float lr, float beta1, float beta2, float eps,
                                 const float* state, void* stream) {
  CGV_REQUIRE(n_problems >= 0 && total_blocks >= 0, "bad size");
  if (n_problems == 0 || total_blocks == 0) return 0;
  CGV_REQUIRE(table_dev && arena_g && arena_p && arena_m && arena_v && state && max_rows >= 1 && max_rows <= cgv::GS_MAX_ROWS, "bad argument");
  CGV_REQUIRE(((((uintptr_t)arena_g | (uintptr_t)arena_p | (uintptr_t)arena_m | (uintptr_t)arena_v)) & 15) == 0,
              "arenas must be 16-byte aligned");
  return cgv::strip_launch<cgv::GW_ADAM>(table_dev, n_problems, total_blocks, max_rows, nullptr,
                                         cgv::RankUpdateArgs{arena_g, arena_p, arena_m, arena_v, state, lr, beta1, beta2, eps},
                                         (hipStream_t)stream, "cgv_grouped_wgrad_strip_adam");
}

int cgv_pack_record_bytes(void) { return (int)sizeof(cgv::PackProblem); }

int cgv_pack_plan(int M, int N, int K, int* n_blocks) {
  CGV_REQUIRE(n_blocks, "null pointer");
  CGV_REQUIRE(M >= 1 && N >= 4 && K >= 4 && (N % 4) == 0 && (K % 4) == 0, "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  const long long f4 = (long long)M * (N + K) / 4;
  *n_blocks = (int)((f4 + cgv::PACK_F4_PER_BLOCK - 1) / cgv::PACK_F4_PER_BLOCK);
  return 0;
}

int cgv_pack_operands(const void* table_dev, int n_problems, int total_blocks, void* stream) {
  CGV_REQUIRE(n_problems >= 0 && total_blocks >= 0, "bad size");
  if (n_problems == 0 || total_blocks == 0) return 0;
  CGV_REQUIRE(table_dev, "null table");
  hipLaunchKernelGGL(cgv::pack_operands_k, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const cgv::PackProblem*>(table_dev), n_problems);
  return cgv::check_launch("cgv_pack_operands");
}

}  // extern "C"
