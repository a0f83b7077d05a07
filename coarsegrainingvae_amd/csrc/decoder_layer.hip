// Decoder layer on the bead graph as CHANNEL-GROUP kernels (reference: cgvae.py:100-123 = per layer
// EquiMessagePsuedo conv.py:180-242 + UpdateBlock conv.py:588-616 + residual adds; Dense modules.py:103-114).
//
// Everything between two Dense products of this loop is LOCAL IN THE CHANNEL f: the pseudo-vector message (inputs
// phi[:, kF+f], state channel f), the norm / stack, the gate.  So a block that owns a group of CB = 4 channels and
// computes exactly the weight rows {g F + f} that feed those channels (g = 0..8 for the message's inv_dense.1, 0..1 for
// [u_mat; v_mat], 0..2 for s_dense.1) can run the local operation in the epilogue of the product -- and, backward, in the
// prologue of the backward-input product, whose row-split partial sums it leaves as ONE slice per block:
//
//   forward   F2  phi = a1 W2^T + b2 (9 x CB rows)  -> EquiMessagePsuedo on the block's channels -> S', Sbar', V', Vbar'
//             F3  [U | Vv] = V'rows [Wu; Wv]^T (2 x CB rows) -> stack = [S' | ||Vv||]
//             F5  a = a0 W1'^T + b1' (3 x CB rows)  -> gate -> S'' = S' + ds, V'' = V' + dv
//             (the two full-width products a1 = swish(S W1^T + b1), a0 = swish(stack W0^T + b0) stay skinny_fwd launches)
//   backward  B1  gate backward -> ga, gU, gVv ; slice = ga W1'[rows]                      (prologue: gS as slice sum)
//             B2  g_a0 = sum of B1's slices ; slice = (g_a0 swish'(z0)) W0[rows]
//             B3  norm backward (g_stack = sum of B2's slices) -> g_S', gVv ; slice = [gU | gVv] [Wu; Wv][rows]
//             B4  gV' = sum of B3's slices (+ residual) -> EquiMessagePsuedo backward (all of it: receiver side,
//                 source side, g_phi, filter gradients -- no cross-block partials: a block sees every edge of its
//                 channels) ; slice = g_phi W2[rows]
//             B5  g_a1 = sum of B4's slices ; slice = (g_a1 swish'(z1)) W1[rows]   -> gS of the layer below = g_s + slices
//
// 5 + 5 launches per layer instead of 8 + 21, and the weight rows of a backward phase are requested BEFORE its prologue
// runs (they do not depend on it), so the HBM round trip hides behind the local math.
//
// Slices are stored QUAD-MAJOR: slice[kq][m][4] (kq = column / 4, m < MP rows): the consumer that owns column quad kq
// reads MP x 16 contiguous bytes per slice.  Slice sums add in a fixed order (per lane ascending slice index, then the
// 36 lane classes in order): deterministic.
//
// Geometry: 576 threads = 9 waves; in the message kernels wave k owns filter k and lane = node * 4 + channel (so at most
// 16 nodes: this path serves small bead graphs -- chignolin: 12 beads; larger ones keep the per-block kernels).
// Products use v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains) with the operand maps of skinny_gemm.hip.
#include "cgv_common.h"

namespace cgv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DL_CB = 4;                 // channels per block
constexpr int DL_WAVES = 9;
constexpr int DL_THREADS = 64 * DL_WAVES;
constexpr int DL_MAX_NODES = 16;

struct dv3 { float x, y, z; };
__device__ __forceinline__ dv3 dldv(const float* p) { f3 t = ld3(p); return dv3{t.x, t.y, t.z}; }
__device__ __forceinline__ dv3 dcross(const dv3& a, const dv3& b) {
  return dv3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float ddot(const dv3& a, const dv3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ void daxpy(dv3& acc, float a, const dv3& x) {
  acc.x = fmaf(a, x.x, acc.x); acc.y = fmaf(a, x.y, acc.y); acc.z = fmaf(a, x.z, acc.z);
}
template <int R>
__device__ __forceinline__ float dfilt(const float (&W)[R + 1], const float* __restrict__ g) {
  float w = W[R] * g[R];
#pragma unroll
  for (int n = 0; n < R; ++n) w = fmaf(W[n], g[n], w);
  return w;
}

// ---------------------------------------------------------------------------------------------- slice sums (quad-major)
// tile[m][c] (m < 16 MB, c < 4) = sum_s slices[s][kq][m][c]; all threads take part; the result is valid after the
// trailing __syncthreads().  scratch: 36 * 16 * MB float4.
template <int MB>
__device__ __forceinline__ void quad_sum(float4* __restrict__ tile, float4* __restrict__ scratch,
                                         const float* __restrict__ slices, int n_slices, long long stride, int kq) {
  constexpr int MP = 16 * MB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, cls = wave * 4 + (lane >> 4);                 // 36 slice classes
  constexpr int SB = 5;                                                   // loads in flight per row tile (150 slices: one batch)
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = cls; s0 < n_slices; s0 += 36 * SB) {
      float4 v[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int s = min(s0 + 36 * u, n_slices - 1);
        v[u] = *reinterpret_cast<const float4*>(slices + (size_t)s * stride + ((size_t)kq * MP + mb * 16 + m) * 4);
      }
#pragma unroll
      for (int u = 0; u < SB; ++u)
        if (s0 + 36 * u < n_slices) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    scratch[(cls * MB + mb) * 16 + m] = acc;
  }
  __syncthreads();
  if (threadIdx.x < MP) {
    const int mm = threadIdx.x;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < 36; ++c) {
      const float4 v = scratch[(c * MB + mm / 16) * 16 + (mm & 15)];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    tile[mm] = t;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------- backward-input core
// The block's G row groups (4 consecutive weight rows each, first row row0[g]) times g_tile[m][g*4 + q] -> this block's
// slice, quad-major.  Wave w takes column tiles w, w + 9, ... of 64 columns.  The weight registers are loaded by
// bi_prefetch at the top of the kernel (independent of the prologue).
template <int G, int NT>
struct BiRegs { float4 w[NT][G]; };

template <int G, int NT>
__device__ __forceinline__ void bi_prefetch(BiRegs<G, NT>& r, const float* __restrict__ W, int K, const int (&row0)[G]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = (wave + DL_WAVES * t) * 64 + 4 * j;
    const int cc = col < K ? col : 0;                                     // clamped: always-valid address, result unused
#pragma unroll
    for (int g = 0; g < G; ++g) r.w[t][g] = *reinterpret_cast<const float4*>(W + (size_t)(row0[g] + q) * K + cc);
  }
}

template <int MB, int G, int NT>
__device__ __forceinline__ void bi_core(const BiRegs<G, NT>& r, const float* __restrict__ g_tile /*[16 MB][G*4] LDS*/,
                                        float* __restrict__ slice /*[K/4][16 MB][4]*/, int K) {
  constexpr int MP = 16 * MB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = (wave + DL_WAVES * t) * 64 + 4 * j;
    if ((wave + DL_WAVES * t) * 64 >= K) break;                           // wave-uniform
    f32x4 acc[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[mb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4 w = r.w[t][g];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float a = g_tile[(mb * 16 + j) * (G * 4) + g * 4 + q];
        acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.x, acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.y, acc[mb][1], 0, 0, 0);
        acc[mb][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.z, acc[mb][2], 0, 0, 0);
        acc[mb][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.w, acc[mb][3], 0, 0, 0);
      }
    }
    if (col < K) {
      const int kq = col >> 2;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int m = mb * 16 + 4 * q + rr;
          *reinterpret_cast<float4*>(slice + ((size_t)kq * MP + m) * 4) =
              make_float4(acc[mb][0][rr], acc[mb][1][rr], acc[mb][2][rr], acc[mb][3][rr]);
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------- forward core
// out[m][g][c] = sum_k x[m][k] W[row0[g] + c][k] for the block's G row groups, m < M (<= 16 MB).  16-row tiles hold four
// groups; the 9 waves split K in whole 16-float steps and meet in LDS (wave order).  red: 9 * T * MB * 256 floats,
// out: 16 MB * G * 4 floats.  Valid after the trailing __syncthreads().
template <int MB, int G>
__device__ __forceinline__ void fwd_core(float* __restrict__ out, float* __restrict__ red, const float* __restrict__ x,
                                         int M, int K, const float* __restrict__ W, const int (&row0)[G]) {
  constexpr int T = (G + 3) / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int steps = (K + 15) / 16, per = (steps + DL_WAVES - 1) / DL_WAVES;
  const int s_beg = wave * per, s_end = min(s_beg + per, steps);
  const float* wrow[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int g = min(t * 4 + (i >> 2), G - 1);                           // surplus rows of the last tile: clamped, unused
    wrow[t] = W + (size_t)(row0[g] + (i & 3)) * K;
  }
  const float* xrow[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) xrow[mb] = x + (size_t)min(mb * 16 + i, M - 1) * K;
  f32x4 acc[T][MB];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[t][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int SB = (T * MB >= 3) ? 3 : 5;                               // steps whose loads are issued together
  for (int s0 = s_beg; s0 < s_end; s0 += SB) {
    float4 a[SB][T], b[SB][MB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int k = (s0 + u) * 16 + 4 * q;
      const int kc = (s0 + u < s_end && k < K) ? k : 0;
#pragma unroll
      for (int t = 0; t < T; ++t) a[u][t] = *reinterpret_cast<const float4*>(wrow[t] + kc);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) b[u][mb] = *reinterpret_cast<const float4*>(xrow[mb] + kc);
    }
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      if (s0 + u < s_end) {
        const bool kok = (s0 + u) * 16 + 4 * q < K;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float4 w = make_float4(kok ? a[u][t].x : 0.f, kok ? a[u][t].y : 0.f, kok ? a[u][t].z : 0.f, kok ? a[u][t].w : 0.f);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            acc[t][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, b[u][mb].x, acc[t][mb], 0, 0, 0);
            acc[t][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, b[u][mb].y, acc[t][mb], 0, 0, 0);
            acc[t][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, b[u][mb].z, acc[t][mb], 0, 0, 0);
            acc[t][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, b[u][mb].w, acc[t][mb], 0, 0, 0);
          }
        }
      }
    }
  }
  // D: lane holds [row = 4 q + r of the tile (group t*4 + q, channel r)][m = mb*16 + i]
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(((wave * T + t) * MB + mb) * 4 + r) * 64 + lane] = acc[t][mb][r];
  __syncthreads();
  for (int o = threadIdx.x; o < T * MB * 256; o += DL_THREADS) {
    const int l = o & 63, r = (o >> 6) & 3, rest = o >> 8;              // rest = t * MB + mb
    const int mb = rest % MB, t = rest / MB;
    const int g = t * 4 + (l >> 4), m = mb * 16 + (l & 15);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < DL_WAVES; ++w) v += red[(((w * T + t) * MB + mb) * 4 + r) * 64 + l];
    if (g < G) out[(m * G + g) * 4 + r] = v;
  }
  __syncthreads();
}

// ============================================================================================== F2: phi + message forward
template <int R>
__global__ __launch_bounds__(DL_THREADS) void dec_msg_fwd_k(
    const float* __restrict__ a1, const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ s,
    const float* __restrict__ sbar, const float* __restrict__ v, const float* __restrict__ vbar,
    const float* __restrict__ geom, const int* __restrict__ rowptr, const int* __restrict__ src,
    const float* __restrict__ Wd, const float* __restrict__ bd, float* __restrict__ phi_out, float* __restrict__ stack,
    float* __restrict__ sbar_out, float* __restrict__ v_out, float* __restrict__ vbar_out, float* __restrict__ rows_out,
    int n, int F) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  __shared__ __attribute__((aligned(16))) float red[DL_WAVES * 3 * 256];
  __shared__ __attribute__((aligned(16))) float phi_l[16 * 9 * 4];
  __shared__ float red2[8][3][64];
  const int f0 = blockIdx.x * DL_CB;
  const int lane = threadIdx.x & 63;
  const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane >> 2, c = lane & 3;
  const bool live = i < n;
  const int ic = live ? i : 0;
  const int f = f0 + c;
  // receiver state and this wave's filter row: requested before the product so that they arrive behind it
  float W[R + 1];
#pragma unroll
  for (int nn = 0; nn < R; ++nn) W[nn] = Wd[((size_t)k * F + f) * R + nn];
  W[R] = bd[(size_t)k * F + f];
  const size_t nf = (size_t)ic * F + f;
  const float s_i = s[nf], sb_i = sbar[nf];
  const dv3 v_i = dldv(v + nf * 3), vb_i = dldv(vbar + nf * 3);
  int row0[9];
#pragma unroll
  for (int g = 0; g < 9; ++g) row0[g] = g * F + f0;
  fwd_core<1, 9>(phi_l, red, a1, n, F, W2, row0);
  // bias, dense copy for the backward pass (phi[m][g F + f0 .. +3])
  for (int o = threadIdx.x; o < 16 * 9; o += DL_THREADS) {
    const int m = o / 9, g = o - m * 9;
    float4 p = *reinterpret_cast<float4*>(phi_l + (m * 9 + g) * 4);
    const float4 b = *reinterpret_cast<const float4*>(b2 + (size_t)g * F + f0);
    p.x += b.x; p.y += b.y; p.z += b.z; p.w += b.w;
    *reinterpret_cast<float4*>(phi_l + (m * 9 + g) * 4) = p;
    if (m < n) *reinterpret_cast<float4*>(phi_out + (size_t)m * 9 * F + (size_t)g * F + f0) = p;
  }
  __syncthreads();
  // EquiMessagePsuedo, wave k = filter k (pseudo_msg.hip: pseudo_fwd_k), lane = (receiver i, channel c)
  float ah = 0.f, ahb = 0.f;
  dv3 acc{0.f, 0.f, 0.f};
  const float* __restrict__ vsrc = (k == 2 || k == 6 || k == 7) ? v : vbar;
  const int e_beg = live ? rowptr[i] : 0, e_end = live ? rowptr[i + 1] : 0;
  constexpr int EB = 8;
  for (int eb = e_beg; eb < e_end; eb += EB) {
    int jj[EB];
    dv3 vj[EB];
#pragma unroll
    for (int u = 0; u < EB; ++u) jj[u] = src[min(eb + u, e_end - 1)];
#pragma unroll
    for (int u = 0; u < EB; ++u) vj[u] = dldv(vsrc + ((size_t)jj[u] * F + f) * 3);
#pragma unroll
    for (int u = 0; u < EB; ++u) {
      if (eb + u < e_end) {
        const float* __restrict__ g = geom + (size_t)(eb + u) * GS;
        const float q = phi_l[(jj[u] * 9 + k) * 4 + c] * dfilt<R>(W, g);
        switch (k) {                                   // wave-uniform
          case 0: ah = fmaf(q, s_i, ah); ahb += ddot(v_i, vj[u]); break;
          case 1: daxpy(acc, q, dv3{g[U], g[U + 1], g[U + 2]}); break;
          case 2: daxpy(acc, q, vj[u]); break;
          case 3: daxpy(acc, q, dcross(v_i, vj[u])); break;
          case 4: daxpy(acc, q * sb_i, vj[u]); break;
          case 5: daxpy(acc, q, vj[u]); break;
          case 6: daxpy(acc, q * sb_i, vj[u]); break;
          case 7: daxpy(acc, q, dcross(v_i, vj[u])); break;
          default: daxpy(acc, q, dcross(vb_i, vj[u])); break;
        }
      }
    }
  }
  if (k > 0) { red2[k - 1][0][lane] = acc.x; red2[k - 1][1][lane] = acc.y; red2[k - 1][2][lane] = acc.z; }
  __syncthreads();
  if (k != 0 || !live) return;
  dv3 av{0.f, 0.f, 0.f}, avb{0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 4; ++w) { av.x += red2[w][0][lane]; av.y += red2[w][1][lane]; av.z += red2[w][2][lane]; }
#pragma unroll
  for (int w = 4; w < 8; ++w) { avb.x += red2[w][0][lane]; avb.y += red2[w][1][lane]; avb.z += red2[w][2][lane]; }
  // updated states (cgvae.py:108-111)
  ah += s_i; ahb += sb_i;
  av.x += v_i.x; av.y += v_i.y; av.z += v_i.z;
  avb.x += vb_i.x; avb.y += vb_i.y; avb.z += vb_i.z;
  stack[(size_t)i * 2 * F + f] = ah;                    // S' is the first half of the update block's stack (conv.py:601)
  sbar_out[nf] = ahb;
  st3(v_out + nf * 3, av.x, av.y, av.z);
  st3(vbar_out + nf * 3, avb.x, avb.y, avb.z);
  rows_out[((size_t)3 * i + 0) * F + f] = av.x;
  rows_out[((size_t)3 * i + 1) * F + f] = av.y;
  rows_out[((size_t)3 * i + 2) * F + f] = av.z;
}

// ============================================================================================== F3: [U | Vv] + norm
__global__ __launch_bounds__(DL_THREADS) void dec_uv_fwd_k(const float* __restrict__ rows, const float* __restrict__ Wuv,
                                                           float* __restrict__ UV, float* __restrict__ stack, int n, int F) {
  __shared__ __attribute__((aligned(16))) float red[DL_WAVES * 3 * 256];
  __shared__ __attribute__((aligned(16))) float uv_l[48 * 2 * 4];
  const int f0 = blockIdx.x * DL_CB;
  const int row0[2] = {f0, F + f0};
  fwd_core<3, 2>(uv_l, red, rows, 3 * n, F, Wuv, row0);
  for (int o = threadIdx.x; o < 3 * n * 2; o += DL_THREADS) {
    const int m = o >> 1, g = o & 1;
    *reinterpret_cast<float4*>(UV + (size_t)m * 2 * F + (size_t)g * F + f0) = *reinterpret_cast<const float4*>(uv_l + (m * 2 + g) * 4);
  }
  if (threadIdx.x < n * 4) {
    const int i = threadIdx.x >> 2, c = threadIdx.x & 3;
    const float x = uv_l[((3 * i + 0) * 2 + 1) * 4 + c], y = uv_l[((3 * i + 1) * 2 + 1) * 4 + c], z = uv_l[((3 * i + 2) * 2 + 1) * 4 + c];
    stack[(size_t)i * 2 * F + F + f0 + c] = sqrtf(((x * x + 1e-10f) + (y * y + 1e-10f)) + (z * z + 1e-10f));      // conv.py:600
  }
}

// ============================================================================================== F5: a + gate
__global__ __launch_bounds__(DL_THREADS) void dec_gate_fwd_k(const float* __restrict__ a0, const float* __restrict__ W1p,
                                                             const float* __restrict__ b1p, const float* __restrict__ UV,
                                                             const float* __restrict__ stack, const float* __restrict__ v2,
                                                             float* __restrict__ a_out, float* __restrict__ s3,
                                                             float* __restrict__ v3, int n, int F) {
  __shared__ __attribute__((aligned(16))) float red[DL_WAVES * 256];
  __shared__ __attribute__((aligned(16))) float a_l[16 * 3 * 4];
  const int f0 = blockIdx.x * DL_CB;
  const int row0[3] = {f0, F + f0, 2 * F + f0};
  fwd_core<1, 3>(a_l, red, a0, n, F, W1p, row0);
  if (threadIdx.x < n * 4) {
    const int i = threadIdx.x >> 2, c = threadIdx.x & 3, f = f0 + c;
    const float a_vv = a_l[(i * 3 + 0) * 4 + c] + b1p[f], a_sv = a_l[(i * 3 + 1) * 4 + c] + b1p[F + f],
                a_ss = a_l[(i * 3 + 2) * 4 + c] + b1p[2 * F + f];
    float* ao = a_out + (size_t)i * 3 * F + f;
    ao[0] = a_vv; ao[F] = a_sv; ao[2 * F] = a_ss;
    const size_t b = (size_t)i * 3 * 2 * F + f;
    const float ux = UV[b], uy = UV[b + 2 * F], uz = UV[b + 4 * F];
    const float vx = UV[b + F], vy = UV[b + 2 * F + F], vz = UV[b + 4 * F + F];
    const size_t nf = (size_t)i * F + f;
    const f3 r = ld3(v2 + nf * 3);
    st3(v3 + nf * 3, ux * a_vv + r.x, uy * a_vv + r.y, uz * a_vv + r.z);                  // conv.py:607, cgvae.py:123
    s3[nf] = ((ux * vx + uy * vy + uz * vz) * a_sv + a_ss) + stack[(size_t)i * 2 * F + f];    // conv.py:612-614, cgvae.py:122
  }
}

// ============================================================================================== B1: gate backward + W1' rows
__global__ __launch_bounds__(DL_THREADS) void dec_gate_bwd_k(
    const float* __restrict__ UV, const float* __restrict__ a, const float* __restrict__ gs_base,
    const float* __restrict__ gs_slices, int gs_n, long long gs_stride, const float* __restrict__ gv,
    const float* __restrict__ W1p, float* __restrict__ ga, float* __restrict__ gUV, float* __restrict__ gs_sum,
    float* __restrict__ slices_out, long long out_stride, int n, int F) {
  __shared__ __attribute__((aligned(16))) float4 scratch[36 * 16];
  __shared__ __attribute__((aligned(16))) float4 gs_l[16];
  __shared__ __attribute__((aligned(16))) float g_l[16 * 3 * 4];
  const int f0 = blockIdx.x * DL_CB;
  const int row0[3] = {f0, F + f0, 2 * F + f0};
  BiRegs<3, 2> wr;
  bi_prefetch<3, 2>(wr, W1p, F, row0);
  if (gs_slices && gs_n > 0) quad_sum<1>(gs_l, scratch, gs_slices, gs_n, gs_stride, blockIdx.x);
  else { if (threadIdx.x < 16) gs_l[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f); __syncthreads(); }
  for (int o = threadIdx.x; o < 16 * 12; o += DL_THREADS) g_l[o] = 0.f;
  __syncthreads();
  if (threadIdx.x < n * 4) {
    const int i = threadIdx.x >> 2, c = threadIdx.x & 3, f = f0 + c;
    const size_t nf = (size_t)i * F + f;
    float gs = reinterpret_cast<const float*>(gs_l)[i * 4 + c];
    if (gs_base) gs += gs_base[nf];
    gs_sum[nf] = gs;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (gv) { const f3 t = ld3(gv + nf * 3); gx = t.x; gy = t.y; gz = t.z; }
    const size_t b = (size_t)i * 3 * 2 * F + f, cc = (size_t)i * 3 * F + f;
    const float ux = UV[b], uy = UV[b + 2 * F], uz = UV[b + 4 * F];
    const float vx = UV[b + F], vy = UV[b + 2 * F + F], vz = UV[b + 4 * F + F];
    const float a_vv = a[cc], a_sv = a[cc + F];
    const float inner = ux * vx + uy * vy + uz * vz;
    const float cs = gs * a_sv;
    const float g0 = gx * ux + gy * uy + gz * uz, g1 = gs * inner, g2 = gs;
    ga[cc] = g0; ga[cc + F] = g1; ga[cc + 2 * F] = g2;
    g_l[(i * 3 + 0) * 4 + c] = g0; g_l[(i * 3 + 1) * 4 + c] = g1; g_l[(i * 3 + 2) * 4 + c] = g2;
    gUV[b] = fmaf(gx, a_vv, cs * vx); gUV[b + 2 * F] = fmaf(gy, a_vv, cs * vy); gUV[b + 4 * F] = fmaf(gz, a_vv, cs * vz);
    gUV[b + F] = cs * ux; gUV[b + 2 * F + F] = cs * uy; gUV[b + 4 * F + F] = cs * uz;
  }
  __syncthreads();
  bi_core<1, 3, 2>(wr, g_l, slices_out + (size_t)blockIdx.x * out_stride, F);
}

// ============================================================================================== B2 / B5: slice sum, act', one row group
template <int NT>
__global__ __launch_bounds__(DL_THREADS) void dec_dense_bwd_k(const float* __restrict__ g_slices, int g_n, long long g_stride,
                                                              const float* __restrict__ z, int act, const float* __restrict__ W,
                                                              float* __restrict__ g_dense, float* __restrict__ slices_out,
                                                              long long out_stride, int n, int N, int K) {
  __shared__ __attribute__((aligned(16))) float4 scratch[36 * 16];
  __shared__ __attribute__((aligned(16))) float4 sum_l[16];
  __shared__ __attribute__((aligned(16))) float g_l[16 * 4];
  const int n0 = blockIdx.x * DL_CB;
  const int row0[1] = {n0};
  BiRegs<1, NT> wr;
  bi_prefetch<1, NT>(wr, W, K, row0);
  quad_sum<1>(sum_l, scratch, g_slices, g_n, g_stride, blockIdx.x);
  if (threadIdx.x < 64) {
    const int i = threadIdx.x >> 2, c = threadIdx.x & 3;
    float g = 0.f;
    if (i < n) {
      g = reinterpret_cast<const float*>(sum_l)[i * 4 + c];
      const size_t at = (size_t)i * N + n0 + c;
      g_dense[at] = g;                                                    // the weight-gradient launch applies act'(z) itself
      if (act) g *= act_bwd(z[at], act);
    }
    g_l[i * 4 + c] = g;
  }
  __syncthreads();
  bi_core<1, 1, NT>(wr, g_l, slices_out + (size_t)blockIdx.x * out_stride, K);
}

// ============================================================================================== B3: norm backward + [Wu; Wv] rows
__global__ __launch_bounds__(DL_THREADS) void dec_uv_bwd_k(const float* __restrict__ gstack_slices, int gs_n, long long gs_stride,
                                                           const float* __restrict__ UV, const float* __restrict__ stack,
                                                           const float* __restrict__ gs_res, const float* __restrict__ Wuv,
                                                           float* __restrict__ gUV, float* __restrict__ g_s2,
                                                           float* __restrict__ slices_out, long long out_stride, int n, int F) {
  __shared__ __attribute__((aligned(16))) float4 scratch[36 * 16];
  __shared__ __attribute__((aligned(16))) float4 gss_l[16], gsn_l[16];
  __shared__ __attribute__((aligned(16))) float g_l[48 * 2 * 4];
  const int f0 = blockIdx.x * DL_CB;
  const int row0[2] = {f0, F + f0};
  BiRegs<2, 2> wr;
  bi_prefetch<2, 2>(wr, Wuv, F, row0);
  quad_sum<1>(gss_l, scratch, gstack_slices, gs_n, gs_stride, blockIdx.x);             // columns f0 .. f0+3 of g_stack
  quad_sum<1>(gsn_l, scratch, gstack_slices, gs_n, gs_stride, F / 4 + blockIdx.x);     // columns F + f0 .. (the norm half)
  for (int o = threadIdx.x; o < 48 * 8; o += DL_THREADS) g_l[o] = 0.f;
  __syncthreads();
  if (threadIdx.x < n * 4) {
    const int i = threadIdx.x >> 2, c = threadIdx.x & 3, f = f0 + c;
    const size_t nf = (size_t)i * F + f;
    g_s2[nf] = reinterpret_cast<const float*>(gss_l)[i * 4 + c] + gs_res[nf];         // S' also reaches S'' directly
    const float t = reinterpret_cast<const float*>(gsn_l)[i * 4 + c] / stack[(size_t)i * 2 * F + F + f];
    const size_t b = (size_t)i * 3 * 2 * F + f;
#pragma unroll
    for (int xyz = 0; xyz < 3; ++xyz) {
      const size_t at = b + (size_t)xyz * 2 * F;
      const float gvv = gUV[at + F] + t * UV[at + F];
      gUV[at + F] = gvv;                                                  // the weight-gradient launch reads the total
      g_l[((3 * i + xyz) * 2 + 0) * 4 + c] = gUV[at];
      g_l[((3 * i + xyz) * 2 + 1) * 4 + c] = gvv;
    }
  }
  __syncthreads();
  bi_core<3, 2, 2>(wr, g_l, slices_out + (size_t)blockIdx.x * out_stride, F);
}

// ============================================================================================== B4: message backward + W2 rows
template <int R>
__global__ __launch_bounds__(DL_THREADS) void dec_msg_bwd_k(
    const float* __restrict__ phi, const float* __restrict__ s, const float* __restrict__ sbar, const float* __restrict__ v,
    const float* __restrict__ vbar, const float* __restrict__ geom_d, const int* __restrict__ rowptr_d,
    const int* __restrict__ src_d, const float* __restrict__ geom_s, const int* __restrict__ rowptr_s,
    const int* __restrict__ dst_s, const float* __restrict__ Wd, const float* __restrict__ bd,
    const float* __restrict__ gh, const float* __restrict__ ghb, const float* __restrict__ gvrows_slices, int gvr_n,
    long long gvr_stride, const float* __restrict__ gv_res, const float* __restrict__ gvb, const float* __restrict__ W2,
    float* __restrict__ g_phi, float* __restrict__ g_s, float* __restrict__ g_sbar, float* __restrict__ g_v,
    float* __restrict__ g_vbar, float* __restrict__ gWd, float* __restrict__ gbd, float* __restrict__ slices_out,
    long long out_stride, int n, int F) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  __shared__ __attribute__((aligned(16))) float4 scratch[36 * 16 * 3];
  __shared__ __attribute__((aligned(16))) float4 gvr_l[48];             // gV' rows [3 i + xyz][c]
  __shared__ __attribute__((aligned(16))) float gphi_l[16 * 9 * 4];
  __shared__ float gv_l[16][4][3];                                       // upstream gv at every node (receivers of other lanes' edges)
  __shared__ float red_src[8][6][64];                                    // waves 1..8: source-side (av, avb) of the lane's node
  __shared__ float red_rcv[9][8][64];                                    // receiver-side partials (as, asb, av, avb)
  const int f0 = blockIdx.x * DL_CB;
  int row0[9];
#pragma unroll
  for (int g = 0; g < 9; ++g) row0[g] = g * F + f0;
  BiRegs<9, 2> wr;
  bi_prefetch<9, 2>(wr, W2, F, row0);
  const int lane = threadIdx.x & 63;
  const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int node = lane >> 2, c = lane & 3, f = f0 + c;
  const bool live = node < n;
  const int nc = live ? node : 0;
  const size_t jf = (size_t)nc * F + f;
  float W[R + 1], G[R + 1];
#pragma unroll
  for (int nn = 0; nn < R; ++nn) W[nn] = Wd[((size_t)k * F + f) * R + nn];
  W[R] = bd[(size_t)k * F + f];
#pragma unroll
  for (int nn = 0; nn <= R; ++nn) G[nn] = 0.f;
  // this lane's node as SOURCE j and as RECEIVER i (state of channel f)
  const float p_j = phi[(size_t)nc * 9 * F + (size_t)k * F + f];
  const float s_n = s[jf], sb_n = sbar[jf];
  const dv3 v_n = dldv(v + jf * 3), vb_n = dldv(vbar + jf * 3);
  const float gh_n = gh ? gh[jf] : 0.f, ghb_n = ghb ? ghb[jf] : 0.f;
  const dv3 gvb_n = gvb ? dldv(gvb + jf * 3) : dv3{0.f, 0.f, 0.f};
  // gV' = sum of the slices of B3 (rows 3 i + xyz) + the residual path V' -> V''
  quad_sum<3>(gvr_l, scratch, gvrows_slices, gvr_n, gvr_stride, blockIdx.x);
  if (threadIdx.x < 64) {
    dv3 t{0.f, 0.f, 0.f};
    if (live) {
      const float* gr = reinterpret_cast<const float*>(gvr_l);
      t = dv3{gr[(3 * node + 0) * 4 + c], gr[(3 * node + 1) * 4 + c], gr[(3 * node + 2) * 4 + c]};
      if (gv_res) { const f3 r = ld3(gv_res + jf * 3); t.x += r.x; t.y += r.y; t.z += r.z; }
    }
    gv_l[node][c][0] = t.x; gv_l[node][c][1] = t.y; gv_l[node][c][2] = t.z;
  }
  __syncthreads();
  const dv3 gv_n{gv_l[node][c][0], gv_l[node][c][1], gv_l[node][c][2]};
  // ---- pass B: the lane's node as source j, edges of the src-sorted view (pseudo_msg.hip: pseudo_bwd_src_k)
  float a = 0.f;
  dv3 av{0.f, 0.f, 0.f}, avb{0.f, 0.f, 0.f};
  {
    const int e_beg = live ? rowptr_s[node] : 0, e_end = live ? rowptr_s[node + 1] : 0;
    for (int e = e_beg; e < e_end; ++e) {
      const float* __restrict__ g = geom_s + (size_t)e * GS;
      const int i = dst_s[e];
      const size_t nf = (size_t)i * F + f;
      const dv3 zero{0.f, 0.f, 0.f};
      const dv3 gv_i{gv_l[i][c][0], gv_l[i][c][1], gv_l[i][c][2]};
      float gq = 0.f;
      dv3 cav = zero, cavb = zero;
      switch (k) {                                   // wave-uniform
        case 0: gq = (gh ? gh[nf] : 0.f) * s[nf]; break;
        case 1: gq = ddot(gv_i, dv3{g[U], g[U + 1], g[U + 2]}); break;
        case 2: gq = ddot(gv_i, v_n); cav = gv_i; break;
        case 3: { const dv3 v_i = dldv(v + nf * 3); gq = ddot(gv_i, dcross(v_i, vb_n)); cavb = dcross(gv_i, v_i); break; }
        case 4: { const float sb_i = sbar[nf]; gq = sb_i * ddot(gv_i, vb_n); cavb = dv3{sb_i * gv_i.x, sb_i * gv_i.y, sb_i * gv_i.z}; break; }
        case 5: { const dv3 gvb_i = gvb ? dldv(gvb + nf * 3) : zero; gq = ddot(gvb_i, vb_n); cavb = gvb_i; break; }
        case 6: { const dv3 gvb_i = gvb ? dldv(gvb + nf * 3) : zero; const float sb_i = sbar[nf];
                  gq = sb_i * ddot(gvb_i, v_n); cav = dv3{sb_i * gvb_i.x, sb_i * gvb_i.y, sb_i * gvb_i.z}; break; }
        case 7: { const dv3 gvb_i = gvb ? dldv(gvb + nf * 3) : zero; const dv3 v_i = dldv(v + nf * 3);
                  gq = ddot(gvb_i, dcross(v_i, v_n)); cav = dcross(gvb_i, v_i); break; }
        default: { const dv3 gvb_i = gvb ? dldv(gvb + nf * 3) : zero; const dv3 vb_i = dldv(vbar + nf * 3);
                   gq = ddot(gvb_i, dcross(vb_i, vb_n)); cavb = dcross(gvb_i, vb_i); break; }
      }
      const float w = dfilt<R>(W, g);
      a = fmaf(gq, w, a);
      const float t = gq * p_j;
#pragma unroll
      for (int nn = 0; nn <= R; ++nn) G[nn] = fmaf(t, g[nn], G[nn]);
      const float q = p_j * w;
      daxpy(av, q, cav);
      daxpy(avb, q, cavb);
      if (k == 0 && ghb) daxpy(avb, ghb[nf], dldv(v + nf * 3));            // the filter-free term ghb_i v_i
    }
  }
  gphi_l[(node * 9 + k) * 4 + c] = live ? a : 0.f;
  if (k > 0) {
    float* r = &red_src[k - 1][0][lane];
    r[0] = av.x; r[64] = av.y; r[128] = av.z; r[192] = avb.x; r[256] = avb.y; r[320] = avb.z;
  }
  // filter gradients: sum over the source nodes (lanes node*4 + c, fixed butterfly order), written once per block
#pragma unroll
  for (int nn = 0; nn <= R; ++nn) {
    float x = G[nn];
    x += __shfl_xor(x, 4); x += __shfl_xor(x, 8); x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
    G[nn] = x;
  }
  if (lane < 4) {
#pragma unroll
    for (int nn = 0; nn < R; ++nn) gWd[((size_t)k * F + f) * R + nn] = G[nn];
    gbd[(size_t)k * F + f] = G[R];
  }
  // ---- pass A: the lane's node as receiver i, edges of the dst-sorted view (pseudo_bwd_recv_k); wave k carries the
  //      terms with q_k (k = 0, 3, 4, 6, 7, 8) and wave 1 the filter-free one
  float as = 0.f, asb = 0.f;
  dv3 rv{0.f, 0.f, 0.f}, rvb{0.f, 0.f, 0.f};
  if (k == 0 || k == 1 || k == 3 || k == 4 || k == 6 || k == 7 || k == 8) {
    const int e_beg = live ? rowptr_d[node] : 0, e_end = live ? rowptr_d[node + 1] : 0;
    for (int e = e_beg; e < e_end; ++e) {
      const int j = src_d[e];
      if (k == 1) { daxpy(rv, ghb_n, dldv(vbar + ((size_t)j * F + f) * 3)); continue; }
      const float* __restrict__ g = geom_d + (size_t)e * GS;
      const float q = phi[(size_t)j * 9 * F + (size_t)k * F + f] * dfilt<R>(W, g);
      switch (k) {
        case 0: as = fmaf(gh_n, q, as); break;
        case 3: daxpy(rv, q, dcross(dldv(vbar + ((size_t)j * F + f) * 3), gv_n)); break;
        case 4: asb = fmaf(q, ddot(gv_n, dldv(vbar + ((size_t)j * F + f) * 3)), asb); break;
        case 6: asb = fmaf(q, ddot(gvb_n, dldv(v + ((size_t)j * F + f) * 3)), asb); break;
        case 7: daxpy(rv, q, dcross(dldv(v + ((size_t)j * F + f) * 3), gvb_n)); break;
        default: daxpy(rvb, q, dcross(dldv(vbar + ((size_t)j * F + f) * 3), gvb_n)); break;
      }
    }
  }
  {
    float* r = &red_rcv[k][0][lane];
    r[0] = as; r[64] = asb; r[128] = rv.x; r[192] = rv.y; r[256] = rv.z; r[320] = rvb.x; r[384] = rvb.y; r[448] = rvb.z;
  }
  __syncthreads();
  // g_phi: dense copy for the weight-gradient launch
  for (int o = threadIdx.x; o < n * 9; o += DL_THREADS) {
    const int m = o / 9, g = o - m * 9;
    *reinterpret_cast<float4*>(g_phi + (size_t)m * 9 * F + (size_t)g * F + f0) = *reinterpret_cast<const float4*>(gphi_l + (m * 9 + g) * 4);
  }
  if (k == 0 && live) {
    // receiver-side sums in wave order, source-side sums in wave order, residual pass-through (outputs were state + delta)
    float ts = gh_n, tsb = ghb_n;
    dv3 tv = gv_n, tvb = gvb_n;
#pragma unroll
    for (int w = 0; w < 9; ++w) {
      const float* r = &red_rcv[w][0][lane];
      ts += r[0]; tsb += r[64];
      tv.x += r[128]; tv.y += r[192]; tv.z += r[256];
      tvb.x += r[320]; tvb.y += r[384]; tvb.z += r[448];
    }
    tv.x += av.x; tv.y += av.y; tv.z += av.z;
    tvb.x += avb.x; tvb.y += avb.y; tvb.z += avb.z;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float* r = &red_src[w][0][lane];
      tv.x += r[0]; tv.y += r[64]; tv.z += r[128]; tvb.x += r[192]; tvb.y += r[256]; tvb.z += r[320];
    }
    g_s[jf] = ts;
    g_sbar[jf] = tsb;
    st3(g_v + jf * 3, tv.x, tv.y, tv.z);
    st3(g_vbar + jf * 3, tvb.x, tvb.y, tvb.z);
  }
  bi_core<1, 9, 2>(wr, gphi_l, slices_out + (size_t)blockIdx.x * out_stride, F);
}

// out[m][f] = base[m][f] + sum_s slices[s][f/4][m][f%4]: the decoder input's gradient leaves the slice format here
__global__ __launch_bounds__(256) void dec_quad_to_dense_k(const float* __restrict__ base, const float* __restrict__ slices,
                                                           int n_slices, long long stride, float* __restrict__ out, int n,
                                                           int F) {
  const int idx = blockIdx.x * 256 + threadIdx.x;          // (kq, m): one float4 of the output
  const int Fq = F / 4;
  if (idx >= Fq * n) return;
  const int kq = idx / n, m = idx - kq * n;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (base) acc = *reinterpret_cast<const float4*>(base + (size_t)m * F + 4 * kq);
  constexpr int SB = 8;
  for (int s0 = 0; s0 < n_slices; s0 += SB) {
    float4 v[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) v[u] = *reinterpret_cast<const float4*>(slices + (size_t)min(s0 + u, n_slices - 1) * stride + ((size_t)kq * 16 + m) * 4);
#pragma unroll
    for (int u = 0; u < SB; ++u)
      if (s0 + u < n_slices) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  *reinterpret_cast<float4*>(out + (size_t)m * F + 4 * kq) = acc;
}

}  // namespace cgv

extern "C" {

int cgv_decoder_layer_supported(int n_nodes, int n_feat, int n_rbf) {
  return n_nodes >= 1 && n_nodes <= cgv::DL_MAX_NODES && n_feat >= 16 && (n_feat % 4) == 0 && n_feat <= 864 &&
         cgv_rbf_supported(n_rbf);
}

/* floats of one slice of a phase's output: (K / 4) column quads x 16 (or 48) rows x 4 */
int64_t cgv_decoder_slice_floats(int K, int rows48) { return (int64_t)(K / 4) * (rows48 ? 48 : 16) * 4; }

#define CGV_DL_CHECK(name)                                                                                    \
  CGV_REQUIRE(cgv_decoder_layer_supported(n_nodes, n_feat, n_rbf), "unsupported shape (n <= 16 nodes, F % 4 == 0, 16 <= F <= 864)"); \
  hipStream_t st = (hipStream_t)stream;                                                                       \
  const int blocks = n_feat / cgv::DL_CB

int cgv_decoder_msg_fwd(const float* a1, const float* W2, const float* b2, const float* s, const float* sbar, const float* v,
                        const float* vbar, const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* Wd,
                        const float* bd, float* phi, float* stack, float* sbar_out, float* v_out, float* vbar_out,
                        float* rows_out, int n_nodes, int n_feat, int n_rbf, void* stream) {
  CGV_REQUIRE(a1 && W2 && b2 && s && sbar && v && vbar && geom_d && rowptr_d && src_d && Wd && bd, "null input");
  CGV_REQUIRE(phi && stack && sbar_out && v_out && vbar_out && rows_out, "null output");
  CGV_DL_CHECK();
  CGV_DISPATCH_RBF(n_rbf, hipLaunchKernelGGL((cgv::dec_msg_fwd_k<RBF>), dim3(blocks), dim3(cgv::DL_THREADS), 0, st, a1, W2, b2,
                                             s, sbar, v, vbar, geom_d, rowptr_d, src_d, Wd, bd, phi, stack, sbar_out, v_out,
                                             vbar_out, rows_out, n_nodes, n_feat));
  return cgv::check_launch("cgv_decoder_msg_fwd");
}

int cgv_decoder_uv_fwd(const float* rows, const float* Wuv, float* UV, float* stack, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(rows && Wuv && UV && stack, "null pointer");
  const int n_rbf = 8;
  CGV_DL_CHECK();
  hipLaunchKernelGGL(cgv::dec_uv_fwd_k, dim3(blocks), dim3(cgv::DL_THREADS), 0, st, rows, Wuv, UV, stack, n_nodes, n_feat);
  return cgv::check_launch("cgv_decoder_uv_fwd");
}

int cgv_decoder_gate_fwd(const float* a0, const float* W1p, const float* b1p, const float* UV, const float* stack,
                         const float* v2, float* a, float* s3, float* v3, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(a0 && W1p && b1p && UV && stack && v2 && a && s3 && v3, "null pointer");
  const int n_rbf = 8;
  CGV_DL_CHECK();
  hipLaunchKernelGGL(cgv::dec_gate_fwd_k, dim3(blocks), dim3(cgv::DL_THREADS), 0, st, a0, W1p, b1p, UV, stack, v2, a, s3, v3,
                     n_nodes, n_feat);
  return cgv::check_launch("cgv_decoder_gate_fwd");
}

int cgv_decoder_gate_bwd(const float* UV, const float* a, const float* gs_base, const float* gs_slices, int gs_n_slices,
                         int64_t gs_slice_stride, const float* gv, const float* W1p, float* ga, float* gUV, float* gs_sum,
                         float* slices_out, int64_t out_slice_stride, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(UV && a && W1p && ga && gUV && gs_sum && slices_out, "null pointer");
  CGV_REQUIRE(gs_n_slices >= 0 && out_slice_stride >= cgv_decoder_slice_floats(n_feat, 0), "bad slices");
  const int n_rbf = 8;
  CGV_DL_CHECK();
  hipLaunchKernelGGL(cgv::dec_gate_bwd_k, dim3(blocks), dim3(cgv::DL_THREADS), 0, st, UV, a, gs_base, gs_slices, gs_n_slices,
                     (long long)gs_slice_stride, gv, W1p, ga, gUV, gs_sum, slices_out, (long long)out_slice_stride, n_nodes,
                     n_feat);
  return cgv::check_launch("cgv_decoder_gate_bwd");
}

int cgv_decoder_dense_bwd(const float* g_slices, int g_n_slices, int64_t g_slice_stride, const float* z, int act, const float* W,
                          float* g_dense, float* slices_out, int64_t out_slice_stride, int n_nodes, int N, int K, void* stream) {
  CGV_REQUIRE(g_slices && W && g_dense && slices_out && g_n_slices >= 1, "null pointer");
  CGV_REQUIRE(act == 0 || (act >= 1 && act <= cgv::CGV_ACT_MAX && z), "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(n_nodes >= 1 && n_nodes <= cgv::DL_MAX_NODES && (N % 4) == 0 && (K % 4) == 0 && K <= 64 * 27 && N >= 4, "unsupported shape");
  CGV_REQUIRE(out_slice_stride >= cgv_decoder_slice_floats(K, 0), "bad slices");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = N / cgv::DL_CB;
  const int tiles = (K + 63) / 64;
#define CGV_DL_DENSE(NTV)                                                                                                \
  hipLaunchKernelGGL((cgv::dec_dense_bwd_k<NTV>), dim3(blocks), dim3(cgv::DL_THREADS), 0, st, g_slices, g_n_slices,       \
                     (long long)g_slice_stride, z, act, W, g_dense, slices_out, (long long)out_slice_stride, n_nodes, N, K)
  if (tiles <= 9) CGV_DL_DENSE(1); else if (tiles <= 18) CGV_DL_DENSE(2); else CGV_DL_DENSE(3);
#undef CGV_DL_DENSE
  return cgv::check_launch("cgv_decoder_dense_bwd");
}

int cgv_decoder_uv_bwd(const float* gstack_slices, int n_slices, int64_t slice_stride, const float* UV, const float* stack,
                       const float* gs_res, const float* Wuv, float* gUV, float* g_s2, float* slices_out,
                       int64_t out_slice_stride, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(gstack_slices && UV && stack && gs_res && Wuv && gUV && g_s2 && slices_out && n_slices >= 1, "null pointer");
  CGV_REQUIRE(out_slice_stride >= cgv_decoder_slice_floats(n_feat, 1), "bad slices");
  const int n_rbf = 8;
  CGV_DL_CHECK();
  hipLaunchKernelGGL(cgv::dec_uv_bwd_k, dim3(blocks), dim3(cgv::DL_THREADS), 0, st, gstack_slices, n_slices,
                     (long long)slice_stride, UV, stack, gs_res, Wuv, gUV, g_s2, slices_out, (long long)out_slice_stride,
                     n_nodes, n_feat);
  return cgv::check_launch("cgv_decoder_uv_bwd");
}

int cgv_decoder_msg_bwd(const float* phi, const float* s, const float* sbar, const float* v, const float* vbar,
                        const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* geom_s,
                        const int32_t* rowptr_s, const int32_t* dst_s, const float* Wd, const float* bd, const float* gh,
                        const float* ghb, const float* gvrows_slices, int n_slices, int64_t slice_stride, const float* gv_res,
                        const float* gvb, const float* W2, float* g_phi, float* g_s, float* g_sbar, float* g_v, float* g_vbar,
                        float* gWd, float* gbd, float* slices_out, int64_t out_slice_stride, int n_nodes, int n_feat, int n_rbf,
                        void* stream) {
  CGV_REQUIRE(phi && s && sbar && v && vbar && geom_d && rowptr_d && src_d && geom_s && rowptr_s && dst_s && Wd && bd && W2,
              "null input");
  CGV_REQUIRE(gvrows_slices && n_slices >= 1 && g_phi && g_s && g_sbar && g_v && g_vbar && gWd && gbd && slices_out, "null pointer");
  CGV_REQUIRE(out_slice_stride >= cgv_decoder_slice_floats(n_feat, 0), "bad slices");
  CGV_DL_CHECK();
  CGV_DISPATCH_RBF(n_rbf, hipLaunchKernelGGL((cgv::dec_msg_bwd_k<RBF>), dim3(blocks), dim3(cgv::DL_THREADS), 0, st, phi, s, sbar,
                                             v, vbar, geom_d, rowptr_d, src_d, geom_s, rowptr_s, dst_s, Wd, bd, gh, ghb,
                                             gvrows_slices, n_slices, (long long)slice_stride, gv_res, gvb, W2, g_phi, g_s,
                                             g_sbar, g_v, g_vbar, gWd, gbd, slices_out, (long long)out_slice_stride, n_nodes,
                                             n_feat));
  return cgv::check_launch("cgv_decoder_msg_bwd");
}

int cgv_decoder_slices_to_dense(const float* base, const float* slices, int n_slices, int64_t slice_stride, float* out,
                                int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(slices && out && n_slices >= 1 && n_nodes >= 1 && n_nodes <= 16 && (n_feat % 4) == 0, "bad argument");
  const int total = (n_feat / 4) * n_nodes;
  hipLaunchKernelGGL(cgv::dec_quad_to_dense_k, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, base, slices,
                     n_slices, (long long)slice_stride, out, n_nodes, n_feat);
  return cgv::check_launch("cgv_decoder_slices_to_dense");
}

}  // extern "C"
