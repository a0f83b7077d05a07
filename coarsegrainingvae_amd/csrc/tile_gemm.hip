// Tiled fp32 GEMMs for the Dense / nn.Linear layers whose row count is beyond the skinny kernels
// (M > 64: the atom-level layers of the encoder, M = atoms of the batch = 332 on the chignolin
// config; every bead-level layer of the dipeptide config, M = 96).
//
// Reference: Dense / nn.Linear forward (CoarseGrainingVAE/modules.py:103-114, Swish modules.py:16-21)
// and its autograd backward.  These products are small (0.24 - 0.72 GFLOP): a library GEMM spends
// 13-18 us on each (profiles/r01l: hipBLASLt MT32x32 tiles walk K in ~19 dependent LDS rounds),
// plus separate bias / activation / column-sum launches.  Here each block splits the REDUCTION
// over its 4 waves (a quarter of the dependent chain each), operands go from L2 straight into
// MFMA layout with 16-byte loads (the whole working set, <= 4 MB, is L2 resident; no LDS staging),
// and the wave partials meet in LDS.  Bias + Swish are fused into the forward epilogue.
//
//   fwd        z = x W^T + b ; y = act(z)     tile 32 x 32, K split over the waves
//   bwd_input  gx = g W                       tile (16 | 32) x 64, N split over the waves
//   wgrad      gW (+)= g^T x                  tile 16 x 64, M split over the waves
// (g = gy * act'(z) and the bias gradient come from cgv_dense_grad_prepare.)
//
// v_mfma_f32_16x16x4_f32 operand map (cdna_hip_programming.md 3): lane l holds A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15]; D: col j = l&15, row i = 4*(l>>4) + reg.  A float4 loaded along the
// reduction axis feeds 4 consecutive MFMAs: component c of lane group q stands for reduction index
// 4q + c of the 16-wide step -- any bijection works as long as A and B use the same one.
#include <stdlib.h>
#include <initializer_list>
#include <type_traits>
#include "cgv_common.h"
#include "streamk_gemm.h"

namespace cgv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4z(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

#define CGV_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N).  Register rings indexed by a LOOP variable end up
// in scratch memory even when the loop is later unrolled (the array is demoted before the unroller runs); indices that
// are constants from the start keep every slot in VGPRs.
template <int I0, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < N) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, N>(f);
  }
}

// ------------------------------------------------------------------ fwd
// Block tile = (32 QM) x (32 QN); wave = (quadrant, k-slice): QM*QN quadrants of 32 x 32 outputs, the K loop of
// each split over KW waves (wave kq walks k-steps kq, kq+KW, ..., 16 floats each).  Lane (i = l&15, q = l>>4):
// A_mb = x[m0 + 16 mb + i][k + 4q ..+3], B_nb = W[n0 + 16 nb + i][k + 4q ..+3].  <1,1,8>: 32 x 32 tiles for
// problems with few tiles (every SIMD needs loads in flight); <2,2,4>: 64 x 64 tiles when there are enough of
// them -- the four quadrants share operand rows through L1, halving the L2 traffic per flop.
// Epilogue: all waves drop their accumulators in LDS, then wave kq of a quadrant finishes sub-tile kq (KW >= 4).
// The SECOND problem of a pair launch: two layers of one shape (M, N, K, act) in one grid -- the extra grid dimension
// selects the operands.  Two atom-level layers that are ready at the same time (the first Dense of contractive block i and
// of message block i + 1 read the same state, cgvae.py:286-305; their second layers follow together) then cost one launch
// boundary, and the launch has twice the tiles to fill the chip with.  All pointers NULL: an ordinary single launch.
struct TileSecond { const float* x; const float* W; const float* bias; float* y; float* z; int act; };

template <int QM, int QN, int KW>
__global__ __launch_bounds__(64 * QM * QN * KW) void tile_fwd_k(const float* __restrict__ x, const float* __restrict__ W,
                                                                 const float* __restrict__ bias, float* __restrict__ y,
                                                                 float* __restrict__ zout, int M, int N, int K, int act,
                                                                 TileSecond s2 = TileSecond{nullptr, nullptr, nullptr, nullptr, nullptr, 0}) {
  static_assert(KW >= 4, "the epilogue spreads the 4 sub-tiles of a quadrant over the k-slice waves");
  if (blockIdx.z) { x = s2.x; W = s2.W; bias = s2.bias; y = s2.y; zout = s2.z; act = s2.act; }
  __shared__ float red[QM * QN][KW][4][4][64];       // [quadrant][k-slice][sub-tile][reg][lane]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int quad = wave / KW, kq = wave - quad * KW;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * (32 * QN) + 32 * (quad % QN), m0 = blockIdx.y * (32 * QM) + 32 * (quad / QN);
  const float* xr[2];
  const float* wr[2];
  bool xok[2], wok[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int m = m0 + 16 * t + i, n = n0 + 16 * t + i;
    xok[t] = m < M; wok[t] = n < N;
    xr[t] = x + (size_t)(xok[t] ? m : 0) * K + 4 * q;
    wr[t] = W + (size_t)(wok[t] ? n : 0) * K + 4 * q;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // contiguous k range per wave: consecutive steps read the two 64-byte halves of the same 128-byte lines
  // (strided assignment fetched every line twice: the block's 8 waves thrash the L1 between the two uses)
  const int steps = (K + 15) / 16, per = (steps + KW - 1) / KW;
  const int s_end = min((kq + 1) * per, steps);
  // The loads of SB steps are issued together, unconditionally (rows clamped into range: products of rows beyond M / N
  // are never stored; the reduction tail is zeroed on W's side).  Guarded loads are branches: the compiler neither
  // unrolled the step loop (`#pragma unroll 4` was refused) nor kept more than one step in flight -- every one of a
  // wave's ~5 steps waited out its own memory round trip (s_waitcnt vmcnt(0)).
  constexpr int SB = 3;
  for (int s0 = kq * per; s0 < s_end; s0 += SB) {
    float4 a0[SB], a1[SB], b0[SB], b1[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int k = 16 * (s0 + u);
      // K % 4 == 0: a float4 is entirely in or out; out of range -> the row's first float4 (xr / wr carry + 4 q)
      const int kc = (s0 + u < s_end && k + 4 * q < K) ? k : -4 * q;
      a0[u] = *reinterpret_cast<const float4*>(xr[0] + kc); a1[u] = *reinterpret_cast<const float4*>(xr[1] + kc);
      b0[u] = *reinterpret_cast<const float4*>(wr[0] + kc); b1[u] = *reinterpret_cast<const float4*>(wr[1] + kc);
    }
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      if (s0 + u >= s_end) break;                  // wave-uniform
      const bool kok = 16 * (s0 + u) + 4 * q < K;
      const float4 w0 = make_float4(kok ? b0[u].x : 0.f, kok ? b0[u].y : 0.f, kok ? b0[u].z : 0.f, kok ? b0[u].w : 0.f);
      const float4 w1 = make_float4(kok ? b1[u].x : 0.f, kok ? b1[u].y : 0.f, kok ? b1[u].z : 0.f, kok ? b1[u].w : 0.f);
      const float4 x0 = a0[u], x1 = a1[u];
      acc[0][0] = CGV_MFMA(x0.x, w0.x, acc[0][0]); acc[0][1] = CGV_MFMA(x0.x, w1.x, acc[0][1]);
      acc[1][0] = CGV_MFMA(x1.x, w0.x, acc[1][0]); acc[1][1] = CGV_MFMA(x1.x, w1.x, acc[1][1]);
      acc[0][0] = CGV_MFMA(x0.y, w0.y, acc[0][0]); acc[0][1] = CGV_MFMA(x0.y, w1.y, acc[0][1]);
      acc[1][0] = CGV_MFMA(x1.y, w0.y, acc[1][0]); acc[1][1] = CGV_MFMA(x1.y, w1.y, acc[1][1]);
      acc[0][0] = CGV_MFMA(x0.z, w0.z, acc[0][0]); acc[0][1] = CGV_MFMA(x0.z, w1.z, acc[0][1]);
      acc[1][0] = CGV_MFMA(x1.z, w0.z, acc[1][0]); acc[1][1] = CGV_MFMA(x1.z, w1.z, acc[1][1]);
      acc[0][0] = CGV_MFMA(x0.w, w0.w, acc[0][0]); acc[0][1] = CGV_MFMA(x0.w, w1.w, acc[0][1]);
      acc[1][0] = CGV_MFMA(x1.w, w0.w, acc[1][0]); acc[1][1] = CGV_MFMA(x1.w, w1.w, acc[1][1]);
    }
  }
  // sub-tile t = kq = (mb, nb) is finished by wave kq: lane holds y[m0 + 16 mb + 4 q + r][n0 + 16 nb + i].  Its bias is
  // requested ahead of the barrier (behind it the load is one more exposed round trip at the end of every launch)
  const int t = kq & 3, mb = t >> 1, nb = t & 1;
  const int n = n0 + 16 * nb + i;
  const float bv = (bias && kq < 4 && n < N) ? bias[n] : 0.f;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[quad][kq][tt][r][lane] = acc[tt >> 1][tt & 1][r];
  __syncthreads();
  if (kq >= 4) return;
  if (n >= N) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + 16 * mb + 4 * q + r;
    if (m >= M) continue;
    float zv = 0.f;
#pragma unroll
    for (int w = 0; w < KW; ++w) zv += red[quad][w][t][r][lane];
    zv += bv;
    if (act) {
      if (zout) zout[(size_t)m * N + n] = zv;
      zv = act_fwd(zv, act);
    }
    y[(size_t)m * N + n] = zv;
  }
}

// ------------------------------------------------------------------ fwd, ONE register tile per CU (round 4)
// What bounds these products is how fast a CU pulls its operand rows (~25 GB/s per CU in the 64-byte fragments of the MFMA
// layout -- the same figure explains every measured time of the kernels above and below: 32 x 32 tiles of 332 x 1800 x 600
// = 3 tiles x 154 KB per CU = 18 us, the 64 x 64 ring on 2000 x 5400 = 11 tiles x 307 KB = 135 us), so the time is the
// BYTES PER CU: (tile rows + tile columns) x K x 4 x tiles per CU.  This kernel computes a (16 MT) x (16 NT) tile per block
// with the whole tile in every wave's accumulators -- each loaded fragment feeds MT or NT MFMAs from registers -- the
// reduction split over 8 waves as above, and the launcher picks MT x NT so that the shape gives AT MOST one tile per CU
// (332 x 1800: 32 x 80 -> 253 tiles x 269 KB; 704 x 1800: 64 x 80 -> 253 x 346 KB; 704 x 600: 64 x 32 -> 209 x 230 KB).
// Tile order is XCD-aware: workgroup L runs on XCD L % 8, and XCD c is given run c of the tiles in (column tile, row tile)
// order, so the weight rows an XCD's L2 holds (~1/8 of W) are shared by all its row tiles and x (<= 1.7 MB) stays resident.
// What it bought (same tool): 7 - 17 % on the wide layers, nothing at 600 outputs -- NOT the 2 x the bytes-per-CU
// picture promises: with all of a CU's 280 KB requested up front the chip still delivers ~6.4 TB/s in aggregate
// (~30 GB/s per CU), i.e. the limit is the rate at which a CU's 64-byte row fragments are served (16 separate half lines
// per load instruction in the MFMA operand layout), not their number in flight.
template <int MT, int NT>
__global__ __launch_bounds__(512) void tile_fwd_bal_k(const float* __restrict__ x, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      float* __restrict__ zout, int M, int N, int K, int act, int MB, int NB,
                                                      TileSecond s2 = TileSecond{nullptr, nullptr, nullptr, nullptr, nullptr, 0}) {
  constexpr int KW = 8, TT = MT * NT;
  if (blockIdx.y) { x = s2.x; W = s2.W; bias = s2.bias; y = s2.y; zout = s2.z; act = s2.act; }
  // LDS for the cross-wave sum: two rounds (waves 4-7 hand over, then waves 0-3) keep it at 4 x TT KB
  extern __shared__ __attribute__((aligned(16))) float bal_red[];      // [4][TT][4][64]
  const int lane = threadIdx.x & 63;
  const int kq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  // tiles in (column tile, row tile) order, cut into 8 equal runs: XCD c (= workgroups c, c + 8, ..) works through run c
  const int L = blockIdx.x, xcd = L & 7, chunk = (MB * NB + 7) >> 3;
  const int pos = xcd * chunk + (L >> 3);
  if (pos >= MB * NB) return;                                          // block-uniform: at most 7 surplus blocks
  const int nt = pos / MB, mt = pos - nt * MB;
  const int m0 = mt * 16 * MT, n0 = nt * 16 * NT;
  const float* xr[MT];
  const float* wr[NT];
#pragma unroll
  for (int a = 0; a < MT; ++a) { const int m = m0 + 16 * a + i; xr[a] = x + (size_t)(m < M ? m : 0) * K + 4 * q; }
#pragma unroll
  for (int b = 0; b < NT; ++b) { const int n = n0 + 16 * b + i; wr[b] = W + (size_t)(n < N ? n : 0) * K + 4 * q; }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int steps = (K + 15) / 16, per = (steps + KW - 1) / KW;
  const int s_end = min((kq + 1) * per, steps);
  // steps whose loads are in flight together: ALL of a wave's five steps at K = 600 where the registers allow it (a lone
  // block per CU has nobody to hide a second round trip behind)
  constexpr int SB = ((MT + NT) * 20 + MT * NT * 4 <= 228) ? 5 : 4;
  for (int s0 = kq * per; s0 < s_end; s0 += SB) {
    float4 av[SB][MT], bv[SB][NT];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int k = 16 * (s0 + u);
      const int kc = (s0 + u < s_end && k + 4 * q < K) ? k : -4 * q;   // out of range: the row's first float4 (never used)
#pragma unroll
      for (int a = 0; a < MT; ++a) av[u][a] = *reinterpret_cast<const float4*>(xr[a] + kc);
#pragma unroll
      for (int b = 0; b < NT; ++b) bv[u][b] = *reinterpret_cast<const float4*>(wr[b] + kc);
    }
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      if (s0 + u >= s_end) break;                  // wave-uniform
      const bool kok = 16 * (s0 + u) + 4 * q < K;
      // the reduction tail is zeroed on W's side; consecutive MFMAs go to DIFFERENT accumulators (no dependent chains)
#pragma unroll
      for (int b = 0; b < NT; ++b)
        if (!kok) bv[u][b] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int a = 0; a < MT; ++a) acc[a][b] = CGV_MFMA(av[u][a].x, bv[u][b].x, acc[a][b]);
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int a = 0; a < MT; ++a) acc[a][b] = CGV_MFMA(av[u][a].y, bv[u][b].y, acc[a][b]);
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int a = 0; a < MT; ++a) acc[a][b] = CGV_MFMA(av[u][a].z, bv[u][b].z, acc[a][b]);
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int a = 0; a < MT; ++a) acc[a][b] = CGV_MFMA(av[u][a].w, bv[u][b].w, acc[a][b]);
    }
  }
  // round 1: waves 4..7 park their tiles, waves 0..3 add them (wave w takes wave w + 4's)
  if (kq >= 4) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) bal_red[(((kq - 4) * TT + a * NT + b) * 4 + r) * 64 + lane] = acc[a][b][r];
  }
  __syncthreads();
  if (kq < 4) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] += bal_red[((kq * TT + a * NT + b) * 4 + r) * 64 + lane];
  }
  __syncthreads();
  // round 2: waves 0..3 park the pair sums; all 512 threads finish the outputs (4 partials each, fixed order)
  if (kq < 4) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) bal_red[((kq * TT + a * NT + b) * 4 + r) * 64 + lane] = acc[a][b][r];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < TT * 256; o += 512) {
    const int l = o & 63, r = (o >> 6) & 3, t = o >> 8;
    const int n = n0 + 16 * (t % NT) + (l & 15), m = m0 + 16 * (t / NT) + 4 * (l >> 4) + r;
    if (n >= N || m >= M) continue;
    float zv = bal_red[((0 * TT + t) * 4 + r) * 64 + l] + bal_red[((1 * TT + t) * 4 + r) * 64 + l] +
               bal_red[((2 * TT + t) * 4 + r) * 64 + l] + bal_red[((3 * TT + t) * 4 + r) * 64 + l];
    zv += bias ? bias[n] : 0.f;
    if (act) {
      if (zout) zout[(size_t)m * N + n] = zv;
      zv = act_fwd(zv, act);
    }
    y[(size_t)m * N + n] = zv;
  }
}

// (MT, NT) of the balanced kernel for a shape: at most one tile per CU (256), the fewest operand bytes per tile; 0 when no
// compiled tile gives <= 256 tiles (large shapes: the LDS-staged kernels) or the shape already has one 32 x 32 tile per CU
static inline int bal_pick(int M, int N, int* mt_out, int* nt_out) {
  static const int cand[][2] = {{2, 2}, {2, 3}, {3, 2}, {3, 3}, {2, 4}, {4, 2}, {2, 5}, {3, 4}, {4, 3}, {4, 4}, {4, 5}, {3, 5}, {2, 6}, {4, 6}};
  long long best = -1;
  for (const auto& c : cand) {
    const int mb = (M + 16 * c[0] - 1) / (16 * c[0]), nb = (N + 16 * c[1] - 1) / (16 * c[1]);
    if ((long long)mb * nb > 256) continue;
    const long long cost = 16 * (c[0] + c[1]);                          // operand rows per tile (x K x 4 bytes)
    if (best < 0 || cost < best) { best = cost; *mt_out = c[0]; *nt_out = c[1]; }
  }
  return best >= 0;
}

// ------------------------------------------------------------------ fwd, LDS-staged (shapes with many output tiles)
// The L2-fed tiles above level off near 50 TF/s: every wave pulls its own operand fragments through the L1 (256 bytes
// per MFMA).  With several 64 x 64 output tiles per CU there is no need to split the reduction for parallelism, so the
// block stages 64 x 32 slabs of x and W in LDS once (coalesced 128-byte row segments, the next slab's loads in flight
// during the current slab's MFMAs, two buffers, one barrier per slab) and its 4 waves (2 x 2 quadrants of 32 x 32) read
// MFMA fragments from there: rows padded to 36 floats make the 16-row x 4-k-group ds_read_b128 conflict-free.
// Measured (tools/fwd_lds_ab.sh) against the L2-fed tiles with batched step loads: 2000 x 5400 x 600: 168 vs 201 us
// (77 TF/s), 2000 x 1800: 57 vs 70, 2000 x 1200: 43 vs 50; 704 x 5400: 63 vs 59, 332 x 5400: 36 vs 35; a lone block
// per CU walks its 19 slabs in ~21 us (1.1 us per slab, the same with a three-slab look-ahead, so not the global
// loads), which loses to the split-reduction tiles: used from 1024 rows and 448 output tiles.
// The product is formed transposed -- W fragments as the A operand, x fragments as B -- so that a lane ends up with
// 4 consecutive n of one row m: bias, activation and the stores are 16-byte wide.
constexpr int LT = 64, LBK = 32, LLD = LBK + 4;

__global__ __launch_bounds__(256) void tile_fwd_lds_k(const float* __restrict__ x, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      float* __restrict__ zout, int M, int N, int K, int act) {
  __shared__ __attribute__((aligned(16))) float As[2][LT * LLD];
  __shared__ __attribute__((aligned(16))) float Bs[2][LT * LLD];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * LT, n0 = blockIdx.x * LT;
  const int sr = t >> 3, sc = 4 * (t & 7);           // staging: rows sr, sr + 32; float4 column sc of the slab
  const float* xrow[2];
  const float* wrow[2];
  bool xok[2], wok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int m = m0 + sr + 32 * u, n = n0 + sr + 32 * u;
    xok[u] = m < M; wok[u] = n < N;
    xrow[u] = x + (size_t)(xok[u] ? m : 0) * K + sc;
    wrow[u] = W + (size_t)(wok[u] ? n : 0) * K + sc;
  }
  float4 ra[2], rb[2];
  auto fetch = [&](int k0) {
    const bool kok = k0 + sc < K;                    // K % 4 == 0
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      ra[u] = ld4z(xrow[u] + (kok ? k0 : 0), kok && xok[u]);
      rb[u] = ld4z(wrow[u] + (kok ? k0 : 0), kok && wok[u]);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *reinterpret_cast<float4*>(&As[buf][(sr + 32 * u) * LLD + sc]) = ra[u];
      *reinterpret_cast<float4*>(&Bs[buf][(sr + 32 * u) * LLD + sc]) = rb[u];
    }
  };
  f32x4 acc[2][2];                                   // [n sub-block a][m sub-block b]: D[n][m]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int slabs = (K + LBK - 1) / LBK;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int kt = 0; kt < slabs; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < slabs) fetch((kt + 1) * LBK);
    const float* __restrict__ a_s = As[buf] + (32 * wm + i) * LLD + 4 * q;
    const float* __restrict__ b_s = Bs[buf] + (32 * wn + i) * LLD + 4 * q;
#pragma unroll
    for (int ks = 0; ks < LBK / 16; ++ks) {
      const float4 x0 = *reinterpret_cast<const float4*>(a_s + 16 * ks), x1 = *reinterpret_cast<const float4*>(a_s + 16 * LLD + 16 * ks);
      const float4 w0 = *reinterpret_cast<const float4*>(b_s + 16 * ks), w1 = *reinterpret_cast<const float4*>(b_s + 16 * LLD + 16 * ks);
      acc[0][0] = CGV_MFMA(w0.x, x0.x, acc[0][0]); acc[0][1] = CGV_MFMA(w0.x, x1.x, acc[0][1]);
      acc[1][0] = CGV_MFMA(w1.x, x0.x, acc[1][0]); acc[1][1] = CGV_MFMA(w1.x, x1.x, acc[1][1]);
      acc[0][0] = CGV_MFMA(w0.y, x0.y, acc[0][0]); acc[0][1] = CGV_MFMA(w0.y, x1.y, acc[0][1]);
      acc[1][0] = CGV_MFMA(w1.y, x0.y, acc[1][0]); acc[1][1] = CGV_MFMA(w1.y, x1.y, acc[1][1]);
      acc[0][0] = CGV_MFMA(w0.z, x0.z, acc[0][0]); acc[0][1] = CGV_MFMA(w0.z, x1.z, acc[0][1]);
      acc[1][0] = CGV_MFMA(w1.z, x0.z, acc[1][0]); acc[1][1] = CGV_MFMA(w1.z, x1.z, acc[1][1]);
      acc[0][0] = CGV_MFMA(w0.w, x0.w, acc[0][0]); acc[0][1] = CGV_MFMA(w0.w, x1.w, acc[0][1]);
      acc[1][0] = CGV_MFMA(w1.w, x0.w, acc[1][0]); acc[1][1] = CGV_MFMA(w1.w, x1.w, acc[1][1]);
    }
    if (kt + 1 < slabs) stash(buf ^ 1);              // last read one barrier ago
    __syncthreads();
  }
  // lane: m = m0 + 32 wm + 16 b + i, n = n0 + 32 wn + 16 a + 4 q .. + 3
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int n = n0 + 32 * wn + 16 * a + 4 * q;
    if (n >= N) continue;                            // N % 4 == 0: the float4 is entirely in or out
    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m = m0 + 32 * wm + 16 * b + i;
      if (m >= M) continue;
      float4 zv = make_float4(acc[a][b][0] + bv.x, acc[a][b][1] + bv.y, acc[a][b][2] + bv.z, acc[a][b][3] + bv.w);
      if (act) {
        if (zout) *reinterpret_cast<float4*>(zout + (size_t)m * N + n) = zv;
        zv.x = act_fwd(zv.x, act); zv.y = act_fwd(zv.y, act); zv.z = act_fwd(zv.z, act); zv.w = act_fwd(zv.w, act);
      }
      *reinterpret_cast<float4*>(y + (size_t)m * N + n) = zv;
    }
  }
}

// ------------------------------------------------------------------ fwd, LDS-staged, three slabs of look-ahead
// tile_fwd_lds_k above keeps ONE slab of global loads in flight: a block that is alone on its CU (mid-size problems: one
// 64 x 64 tile per CU or fewer) then pays a memory round trip per slab -- 1.1 us against 0.43 us of MFMA work, 21 us for
// K = 600.  Here the staging registers form a ring of three slabs: the loads of slab kt + 3 are issued before the MFMAs of
// slab kt, so two slabs (32 KB per block) are always in flight; same LDS image, same fragment reads, same transposed
// product and 16-byte epilogue as above.
// How it has to be written for the compiler: (i) the slab loop is FULLY unrolled (SLABS = ceil(K / 32) is a template
// argument; the layer widths of this model give K = 600 and 1200): in straight-line code hipcc counts its loads and waits
// with exact vmcnt(N), around a loop's back edge it drains to vmcnt(0) once per trip; (ii) the ring slots are NAMED
// variables picked by a switch on kt % 3 that folds after unrolling -- a slot ARRAY indexed by the loop variable (or
// reached through a lambda capture) is demoted to scratch memory before the unroller runs: 171 scratch instructions and
// 30 us per call, against 21 us for the one-slab look-ahead; so is anything whose ADDRESS is selected (see stash).
struct RingSlot { f32x4 a0, a1, b0, b1; };          // ext vectors: plain SSA values (HIP's float4 is a struct of unions)

// (__launch_bounds__(256, 2): two waves per SIMD are asked for, not the four the LDS footprint would admit -- at four the
// compiler caps the kernel at 128 registers and spills the ring it was given to hide latency with)
template <int SLABS>
__global__ __launch_bounds__(256, 2) void tile_fwd_ring_k(const float* __restrict__ x, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       float* __restrict__ zout, int M, int N, int K, int act) {
  __shared__ __attribute__((aligned(16))) float As[2][LT * LLD];
  __shared__ __attribute__((aligned(16))) float Bs[2][LT * LLD];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * LT, n0 = blockIdx.x * LT;
  const int sr = t >> 3, sc = 4 * (t & 7);           // staging: rows sr, sr + 32; float4 column sc of the slab
  // rows beyond M / N are clamped into range: their products are never stored
  const float* const xr0 = x + (size_t)min(m0 + sr, M - 1) * K + sc;
  const float* const xr1 = x + (size_t)min(m0 + sr + 32, M - 1) * K + sc;
  const float* const wr0 = W + (size_t)min(n0 + sr, N - 1) * K + sc;
  const float* const wr1 = W + (size_t)min(n0 + sr + 32, N - 1) * K + sc;
  // unconditional loads (a guarded load is a branch and breaks the batch): a float4 beyond K reads the row's first one
  // and is zeroed on W's side when stashed (K % 4 == 0: a float4 is entirely in or out)
  auto fetch = [&](RingSlot& r, int kt) __attribute__((always_inline)) {
    const int k0 = kt * LBK;
    const int off = (k0 + sc < K) ? k0 : -sc;
    r.a0 = *reinterpret_cast<const f32x4*>(xr0 + off); r.a1 = *reinterpret_cast<const f32x4*>(xr1 + off);
    r.b0 = *reinterpret_cast<const f32x4*>(wr0 + off); r.b1 = *reinterpret_cast<const f32x4*>(wr1 + off);
  };
  auto stash = [&](const RingSlot& r, int kt, int buf) __attribute__((always_inline)) {
    const bool kok = kt * LBK + sc < K;
    // (selects of VALUES: `kok ? r.b0 : zero` on two lvalues selects an ADDRESS and sends the slots to scratch)
    *reinterpret_cast<f32x4*>(&As[buf][sr * LLD + sc]) = r.a0;
    *reinterpret_cast<f32x4*>(&As[buf][(sr + 32) * LLD + sc]) = r.a1;
    *reinterpret_cast<f32x4*>(&Bs[buf][sr * LLD + sc]) = kok ? r.b0 : f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(&Bs[buf][(sr + 32) * LLD + sc]) = kok ? r.b1 : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  RingSlot s0, s1, s2;
  fetch(s0, 0);
  fetch(s1, SLABS > 1 ? 1 : 0);
  fetch(s2, SLABS > 2 ? 2 : 0);
  f32x4 acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc10 = acc00, acc11 = acc00;      // [n sub-block][m sub-block]: D[n][m]
  stash(s0, 0, 0);
  __syncthreads();
#pragma unroll
  for (int kt = 0; kt < SLABS; ++kt) {
    const int buf = kt & 1;
    // slot kt % 3 held slab kt (already in LDS): refill it with slab kt + 3
    if (kt + 3 < SLABS) {
      switch (kt % 3) { case 0: fetch(s0, kt + 3); break; case 1: fetch(s1, kt + 3); break; default: fetch(s2, kt + 3); break; }
    }
    const float* __restrict__ a_s = As[buf] + (32 * wm + i) * LLD + 4 * q;
    const float* __restrict__ b_s = Bs[buf] + (32 * wn + i) * LLD + 4 * q;
#pragma unroll
    for (int ks = 0; ks < LBK / 16; ++ks) {
      const float4 x0 = *reinterpret_cast<const float4*>(a_s + 16 * ks), x1 = *reinterpret_cast<const float4*>(a_s + 16 * LLD + 16 * ks);
      const float4 w0 = *reinterpret_cast<const float4*>(b_s + 16 * ks), w1 = *reinterpret_cast<const float4*>(b_s + 16 * LLD + 16 * ks);
      acc00 = CGV_MFMA(w0.x, x0.x, acc00); acc01 = CGV_MFMA(w0.x, x1.x, acc01);
      acc10 = CGV_MFMA(w1.x, x0.x, acc10); acc11 = CGV_MFMA(w1.x, x1.x, acc11);
      acc00 = CGV_MFMA(w0.y, x0.y, acc00); acc01 = CGV_MFMA(w0.y, x1.y, acc01);
      acc10 = CGV_MFMA(w1.y, x0.y, acc10); acc11 = CGV_MFMA(w1.y, x1.y, acc11);
      acc00 = CGV_MFMA(w0.z, x0.z, acc00); acc01 = CGV_MFMA(w0.z, x1.z, acc01);
      acc10 = CGV_MFMA(w1.z, x0.z, acc10); acc11 = CGV_MFMA(w1.z, x1.z, acc11);
      acc00 = CGV_MFMA(w0.w, x0.w, acc00); acc01 = CGV_MFMA(w0.w, x1.w, acc01);
      acc10 = CGV_MFMA(w1.w, x0.w, acc10); acc11 = CGV_MFMA(w1.w, x1.w, acc11);
    }
    // slab kt + 1 sits in slot (kt + 1) % 3 (requested two slabs ago): into the other buffer, last read one barrier ago
    if (kt + 1 < SLABS) {
      switch ((kt + 1) % 3) { case 0: stash(s0, kt + 1, buf ^ 1); break; case 1: stash(s1, kt + 1, buf ^ 1); break; default: stash(s2, kt + 1, buf ^ 1); break; }
      __syncthreads();
    }
  }
  // lane: m = m0 + 32 wm + 16 b + i, n = n0 + 32 wn + 16 a + 4 q .. + 3
  const f32x4 acc[2][2] = {{acc00, acc01}, {acc10, acc11}};
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int n = n0 + 32 * wn + 16 * a + 4 * q;
    if (n >= N) continue;                            // N % 4 == 0: the float4 is entirely in or out
    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m = m0 + 32 * wm + 16 * b + i;
      if (m >= M) continue;
      float4 zv = make_float4(acc[a][b][0] + bv.x, acc[a][b][1] + bv.y, acc[a][b][2] + bv.z, acc[a][b][3] + bv.w);
      if (act) {
        if (zout) *reinterpret_cast<float4*>(zout + (size_t)m * N + n) = zv;
        zv.x = act_fwd(zv.x, act); zv.y = act_fwd(zv.y, act); zv.z = act_fwd(zv.z, act); zv.w = act_fwd(zv.w, act);
      }
      *reinterpret_cast<float4*>(y + (size_t)m * N + n) = zv;
    }
  }
}

// ------------------------------------------------------------------ bwd_input
// gx[m, k] = sum_n g[m, n] W[n, k].  grid (ceil(K/64), ceil(M/(16 MB))), 256 threads; wave w walks n-steps
// w, w+4, ... (16 rows of W each).  Lane (j = l&15, q = l>>4): A_mb = g[m0 + 16 mb + j][n + 4q ..+3];
// for c = 0..3 one float4 B_c = W[n + 4q + c][k0 + 4j ..+3] (4 rows x 256 contiguous bytes per
// instruction); MFMA (c, s) uses A.comp(c) and B_c.comp(s) and accumulates D_s = gx[..][k0 + 4j + s].
// A THIRD gradient of the same input, held as one row per SEGMENT of the rows (the backward of
// scatter_mean / scatter_add of this very input, cgvae.py:297: g[m, :] += src[seg(m), :] (/ len(seg(m)) for the mean)):
// added in the store epilogue instead of by a broadcast launch + an accumulation add.
// A second SOURCE of one output: gx = (g * act'(z)) W + (g2 * act2'(z2)) W2 (+ add ...), both reductions in one launch --
// two layers that read the same input (the first Dense of contractive block i and of message block i + 1) return their
// input gradients as one product instead of a chain of two launches.  Same N, K.
struct BwdSource { const float* g; const float* W; const float* z; int act; };
struct BwdSecond { const float* g; const float* W; float* gx; const float* z; const float* add; int act; };   // see TileSecond
// The output is the gradient of y = act(zo) of the layer BEFORE this one (this layer's input): with zo given the stored
// value is the gradient of zo, gx * act'(zo) -- applied once per element here instead of by that layer's backward-input /
// weight-gradient launches in their operand loads, where every one of the K / 64 column-tile blocks that share a row tile
// evaluates it again (704 x 600 x 600: 17.7 against 11.2 us per product, tools/bwd_act_bench.py).
struct OutAct { const float* z; int act; const float* z2; int act2; };      // z2 / act2: the second problem of a pair launch
// The reduction of ONE output tile over several blocks (grid y = row tiles x n): few output tiles and a long reduction (96 bead
// rows x 1800 columns = 60 tiles: 60 of 256 CUs busy, each streaming 460 KB of weight rows through three dependent batches
// of requests per wave).  Block zs takes the zs-th share of the reduction steps, leaves its partial tile in `part`
// (agent-scope write-through stores, acknowledged before its ticket is drawn: the hand-over of loss_tail.hip / equi_msg_grp.hip)
// and the block that draws the tile's LAST ticket adds the n partial tiles in share order and runs the store epilogue: the
// result does not depend on who arrives last.  Tickets reset themselves; part / ticket: cgv_tile_bwd_input_split.
struct SplitN { float* part; unsigned* ticket; int n; };
// The output is g_stack of UpdateBlock's s_dense.0 (conv.py:600-603: stack = [s | ||Vv||]) and goes straight through the backward
// of the norm / stack step instead of to memory (update.hip: update_norm_stack_bwd, one launch less per layer):
//   columns k < F:   g_s[m, k]  = o + g_res[m, k]                        (g_res: the residual's gradient or NULL)
//   columns F + f:   gVv[3 m + xyz, f] (+)= o / stack[m, F + f] * Vv[3 m + xyz, f]      (rows of `ld` floats)
struct NormStackOut { const float* stack; const float* Vv; const float* g_res; float* g_s; float* gVv; int F; int ld; int accumulate; };
struct BcastAdd {
  const float* src;          // [n_seg, K] or NULL
  const int64_t* row2seg;    // [M] segment of every row (the CG mapping)
  const int* rowptr;         // [n_seg + 1] CSR of the segments (their lengths)
  int mean;
};

#ifndef CGV_BWD_INPUT_PIPELINE
#define CGV_BWD_INPUT_PIPELINE 0   /* measured SLOWER (tools/gemm_shapes.py, same box): 332 x 600 x 600 8.7 against 7.3 us, 704 x 600 x 1200 23.2 / 17.2 -- two batches in registers cost the second resident block per CU, which hides the round trips better */
#endif
template <int MB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void tile_bwd_input_k(const float* __restrict__ g, const float* __restrict__ W,
                                                        float* __restrict__ gx, int M, int N, int K,
                                                        const float* __restrict__ z, int act,
                                                        const float* __restrict__ add = nullptr,
                                                        BcastAdd bc = BcastAdd{nullptr, nullptr, nullptr, 0},
                                                        BwdSecond s2 = BwdSecond{nullptr, nullptr, nullptr, nullptr, nullptr, 0},
                                                        BwdSource more = BwdSource{nullptr, nullptr, nullptr, 0},
                                                        OutAct oa = OutAct{nullptr, 0, nullptr, 0},
                                                        SplitN sp = SplitN{nullptr, nullptr, 1},
                                                        NormStackOut ns = NormStackOut{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0}) {
  if (blockIdx.z) { g = s2.g; W = s2.W; gx = s2.gx; z = s2.z; add = s2.add; act = s2.act; bc.src = nullptr; oa.z = oa.z2; oa.act = oa.act2; }
  __shared__ float red[WAVES][MB * 4][4][64];        // [wave][mb*4 + s][reg][lane]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int ry = (int)blockIdx.y / sp.n, zs = (int)blockIdx.y - ry * sp.n;          // row tile, share of the reduction
  const int k0 = blockIdx.x * 64, m0 = ry * (16 * MB);
  const int kcol = k0 + 4 * j;
  const bool kok = kcol < K;
  const float* gr[MB];
  bool gok[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = m0 + 16 * mb + j;
    gok[mb] = m < M;
    gr[mb] = g + (size_t)(gok[mb] ? m : 0) * N + 4 * q;
  }
  const ptrdiff_t zoff = act ? z - g : 0;            // z has g's layout: the upstream gradient is corrected on the fly
  f32x4 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[mb][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int steps_all = (N + 15) / 16, zper = (steps_all + sp.n - 1) / sp.n;
  const int z_beg = min(zs * zper, steps_all), steps = min(z_beg + zper, steps_all);   // this block's steps [z_beg, steps)
  const int per = (steps - z_beg + WAVES - 1) / WAVES;                                // contiguous n range per wave (see fwd)
  const int st_beg = min(z_beg + wave * per, steps), st_end = min(st_beg + per, steps);
  // SB steps' loads are issued together and unconditionally (see tile_fwd_k: a guarded load is a branch, and a wave's
  // 14 - 21 steps each waited out their own memory round trip).  Rows beyond M and columns beyond K are clamped into
  // range -- their products are never stored -- and the reduction tail (n >= N) is zeroed on g's side.
  // 3 at 16 waves (1024-thread blocks have 128 VGPRs per lane); 2 otherwise: 4 cost the 8-wave variant its second
  // resident block per CU (132 VGPRs; 704 rows: 12.7 -> 13.2 us per call, while 332 rows gained)
  constexpr int SB = WAVES >= 16 ? 3 : 2;
  const float* wcol = W + (kok ? kcol : 0);
  ptrdiff_t zo = zoff;
  const int n_src = more.g ? 2 : 1;
  for (int src = 0; src < n_src; ++src) {
  if (src == 1) {                                    // the second source: same rows / columns, other operands
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) gr[mb] = more.g + (gr[mb] - g);
    act = more.act;
    zo = act ? more.z - more.g : 0;
    wcol = more.W + (kok ? kcol : 0);
  }
  // Software pipeline over batches of SB steps: the requests of batch i + 1 are issued BEFORE batch i is multiplied, so a
  // wave's memory round trip hides behind its own MFMAs (each batch used to wait out its round trip with nothing of this
  // wave in flight: 7 round trips per wave at N = 1800, 16 - 20 us against 7 us of MFMAs for 332 x 600 x 1800).  Two named
  // buffers, loop unrolled by two; every batch is requested unconditionally (steps beyond the wave's range are clamped to
  // valid addresses and never multiplied), so the counters the compiler waits on are the same on every path.
  struct Batch { float4 a[SB][MB], zz[SB][MB], b[SB][4]; };
  auto request = [&](Batch& B, int st0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int n = 16 * (st0 + u);
      // N % 4 == 0: rows n + 4q .. + 3 are all in or all out; out of range -> g's first float4 of the row (gr carries
      // + 4 q) and W's rows 0 .. 3, both always there
      const bool in = st0 + u < st_end && n + 4 * q < N;
      const int ng = in ? n : -4 * q, nw = in ? n + 4 * q : 0;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        B.a[u][mb] = *reinterpret_cast<const float4*>(gr[mb] + ng);
        if (act) B.zz[u][mb] = *reinterpret_cast<const float4*>(gr[mb] + zo + ng);       // wave-uniform
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) B.b[u][c] = *reinterpret_cast<const float4*>(wcol + (size_t)(nw + c) * K);
    }
  };
  auto multiply = [&](const Batch& B, int st0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      if (st0 + u >= st_end) break;                  // wave-uniform
      const bool nok = 16 * (st0 + u) + 4 * q < N;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        float4 av = B.a[u][mb];
        if (act) {                                   // g = gy * act'(z) (cgv_dense_grad_prepare's job, without its launch)
          av.x *= act_bwd(B.zz[u][mb].x, act); av.y *= act_bwd(B.zz[u][mb].y, act);
          av.z *= act_bwd(B.zz[u][mb].z, act); av.w *= act_bwd(B.zz[u][mb].w, act);
        }
        const float ac[4] = {nok ? av.x : 0.f, nok ? av.y : 0.f, nok ? av.z : 0.f, nok ? av.w : 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[mb][0] = CGV_MFMA(ac[c], B.b[u][c].x, acc[mb][0]);
          acc[mb][1] = CGV_MFMA(ac[c], B.b[u][c].y, acc[mb][1]);
          acc[mb][2] = CGV_MFMA(ac[c], B.b[u][c].z, acc[mb][2]);
          acc[mb][3] = CGV_MFMA(ac[c], B.b[u][c].w, acc[mb][3]);
        }
      }
    }
  };
  if constexpr (CGV_BWD_INPUT_PIPELINE && WAVES < 16) {          // (16 waves = 1024 threads: 128 registers, no room for two batches)
    Batch b0, b1;
    int st0 = st_beg;
    request(b0, st0);
    for (; st0 < st_end; st0 += 2 * SB) {
      request(b1, st0 + SB);
      multiply(b0, st0);
      request(b0, st0 + 2 * SB);
      multiply(b1, st0 + SB);
    }
  } else {
    for (int st0 = st_beg; st0 < st_end; st0 += SB) {
      Batch b0;
      request(b0, st0);
      multiply(b0, st0);
    }
  }
  }
  // The tile is FINISHED by as many waves as it has (m-block, accumulator row) pairs -- wave c takes pair c, c + WAVES, ..:
  // it adds the WAVES partial sums of its 4 output columns (wave order: the same sum as ever) and runs the store epilogue.
  // With wave 0 finishing the whole tile alone (rounds 1-5) every launch ended in a serial tail of (WAVES - 1) x 16 MB LDS
  // reads and adds per lane plus, since the downstream activation, 16 MB exp / rcp evaluations: 41 -> 49 us on the 440-block
  // pair launch of dipeptide.  What the epilogue adds / multiplies is requested by the finishing wave ahead of the barrier
  // (one float4 each now): behind it each would be one more exposed round trip at the end of the launch.
  constexpr int PAIRS = MB * 4, PER = (PAIRS + WAVES - 1) / WAVES;
  float4 pre_add[PER], pre_z[PER];
  if (kok && (add || oa.z)) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int c = wave + u * WAVES;
      if (c < PAIRS) {
        const size_t at = (size_t)min(m0 + 16 * (c >> 2) + 4 * q + (c & 3), M - 1) * K + kcol;
        if (add) pre_add[u] = *reinterpret_cast<const float4*>(add + at);
        if (oa.z) pre_z[u] = *reinterpret_cast<const float4*>(oa.z + at);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < MB * 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][t][r][lane] = acc[t >> 2][t & 3][r];
  __syncthreads();
  float* tile_part = nullptr;
  if (sp.n > 1) {
    // this block's partial tile -> part[tile][zs][16 MB rows][64], then the ticket; only the last arriver goes on
    unsigned* s_last = reinterpret_cast<unsigned*>(&red[0][0][0][0]);      // (red is spent behind the barrier below; LDS is full)
    const size_t tid = ((size_t)blockIdx.z * (gridDim.y / sp.n) + ry) * gridDim.x + blockIdx.x;
    tile_part = sp.part + tid * sp.n * (size_t)(MB * 16 * 64);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int c = wave + u * WAVES;
      if (c >= PAIRS) break;
      const int mb = c >> 2, r = c & 3;
      f32x4 o;
#pragma unroll
      for (int sI = 0; sI < 4; ++sI) {
        const int t = mb * 4 + sI;
        float a = red[0][t][r][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) a += red[w][t][r][lane];
        o[sI] = a;
      }
      float* dst = tile_part + (size_t)zs * (MB * 16 * 64) + (16 * mb + 4 * q + r) * 64 + 4 * j;
      asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(o) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *s_last = atomicAdd(sp.ticket + tid, 1u) == (unsigned)(sp.n - 1) ? 1u : 0u;
    __syncthreads();
    if (!*s_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(sp.ticket + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!kok) return;
  // D_s of m-block mb: lane holds gx[m0 + 16 mb + 4 q + r][k0 + 4 j + s] -> one float4 (s = 0..3) per r
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int c = wave + u * WAVES;
    if (c >= PAIRS) break;                           // wave-uniform
    const int mb = c >> 2, r = c & 3;
    const int m = m0 + 16 * mb + 4 * q + r;
    if (m >= M) continue;
    float o[4];
    if (sp.n > 1) {                                  // the shares' partial tiles, in share order (agent-scope loads)
      const float* src = tile_part + (16 * mb + 4 * q + r) * 64 + 4 * j;
#pragma unroll
      for (int sI = 0; sI < 4; ++sI) o[sI] = __hip_atomic_load(src + sI, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int zz = 1; zz < sp.n; ++zz)
#pragma unroll
        for (int sI = 0; sI < 4; ++sI)
          o[sI] += __hip_atomic_load(src + (size_t)zz * (MB * 16 * 64) + sI, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
    for (int sI = 0; sI < 4; ++sI) {
      const int t = mb * 4 + sI;
      o[sI] = red[0][t][r][lane];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) o[sI] += red[w][t][r][lane];
    }
    }
    if (add) {                                       // a second gradient of the same input (blocks.py: fork of the first Dense)
      const float4 a4 = pre_add[u];
      o[0] += a4.x; o[1] += a4.y; o[2] += a4.z; o[3] += a4.w;
    }
    if (bc.src) {                                    // ... and a third, one row per segment of the rows
      const int sg = (int)bc.row2seg[m];
      const float4 b4 = *reinterpret_cast<const float4*>(bc.src + (size_t)sg * K + kcol);
      const int len = bc.rowptr[sg + 1] - bc.rowptr[sg];
      const float sc = bc.mean ? 1.0f / (float)(len > 1 ? len : 1) : 1.0f;
      o[0] = fmaf(b4.x, sc, o[0]); o[1] = fmaf(b4.y, sc, o[1]); o[2] = fmaf(b4.z, sc, o[2]); o[3] = fmaf(b4.w, sc, o[3]);
    }
    if (oa.z) {                                      // gradient of the previous layer's pre-activation (see OutAct)
      const float4 z4 = pre_z[u];
      o[0] *= act_bwd(z4.x, oa.act); o[1] *= act_bwd(z4.y, oa.act); o[2] *= act_bwd(z4.z, oa.act); o[3] *= act_bwd(z4.w, oa.act);
    }
    if (ns.stack) {                                  // UpdateBlock: through the norm / stack backward instead of to memory
      if (kcol < ns.F) {
        if (ns.g_res) {
          const float4 r4 = *reinterpret_cast<const float4*>(ns.g_res + (size_t)m * ns.F + kcol);
          o[0] += r4.x; o[1] += r4.y; o[2] += r4.z; o[3] += r4.w;
        }
        *reinterpret_cast<float4*>(ns.g_s + (size_t)m * ns.F + kcol) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        const int f = kcol - ns.F;
        const float4 nr = *reinterpret_cast<const float4*>(ns.stack + (size_t)m * 2 * ns.F + kcol);
        const float t0 = o[0] / nr.x, t1 = o[1] / nr.y, t2 = o[2] / nr.z, t3 = o[3] / nr.w;
#pragma unroll
        for (int xyz = 0; xyz < 3; ++xyz) {
          const size_t at = (size_t)(3 * m + xyz) * ns.ld + f;
          const float4 v4 = *reinterpret_cast<const float4*>(ns.Vv + at);
          float4 w = make_float4(t0 * v4.x, t1 * v4.y, t2 * v4.z, t3 * v4.w);
          if (ns.accumulate) { const float4 old = *reinterpret_cast<const float4*>(ns.gVv + at); w.x += old.x; w.y += old.y; w.z += old.z; w.w += old.w; }
          *reinterpret_cast<float4*>(ns.gVv + at) = w;
        }
      }
      continue;
    }
    *reinterpret_cast<float4*>(gx + (size_t)m * K + kcol) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ------------------------------------------------------------------ wgrad
// gW[n, k] (+)= sum_m g[m, n] x[m, k].  grid (ceil(K/64), ceil(N/16)), 256 threads; wave w walks m-steps
// w, w+4, ... (4 rows each).  Lane (i = l&15, q = l>>4): A = g[m + q][n0 + i] (4 rows x 64 contiguous bytes),
// B = x[m + q][k0 + 4 i ..+3] (4 rows x 256 bytes); MFMA s uses B.comp(s): D_s = gW[n0 + ..][k0 + 4 j + s].
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void tile_wgrad_k(const float* __restrict__ g, const float* __restrict__ x,
                                                    float* __restrict__ gW, int M, int N, int K, int accumulate) {
  __shared__ float red[WAVES - 1][4][4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 16;
  const int ncol = n0 + i, kcol = k0 + 4 * i;
  const bool nok = ncol < N, kok = kcol < K;
  f32x4 acc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int steps = (M + 3) / 4, per = (steps + WAVES - 1) / WAVES;
  const int st_end = min((wave + 1) * per, steps);
#pragma unroll 8
  for (int st = wave * per; st < st_end; ++st) {
    const int m = 4 * st + q;
    const bool mok = m < M;
    const float a = (mok && nok) ? g[(size_t)m * N + ncol] : 0.f;
    const float4 b = ld4z(x + (size_t)(mok && kok ? m : 0) * K + (kok ? kcol : 0), mok && kok);
    acc[0] = CGV_MFMA(a, b.x, acc[0]);
    acc[1] = CGV_MFMA(a, b.y, acc[1]);
    acc[2] = CGV_MFMA(a, b.z, acc[2]);
    acc[3] = CGV_MFMA(a, b.w, acc[3]);
  }
  if (wave > 0) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 1][s][r][lane] = acc[s][r];
  }
  __syncthreads();
  if (wave != 0 || !kok) return;
  // D_s: lane holds gW[n0 + 4 q + r][k0 + 4 i + s]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + 4 * q + r;
    if (n >= N) continue;
    float o[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      o[s] = acc[s][r];
#pragma unroll
      for (int w = 0; w < WAVES - 1; ++w) o[s] += red[w][s][r][lane];
    }
    float4* dst = reinterpret_cast<float4*>(gW + (size_t)n * K + kcol);
    float4 v = make_float4(o[0], o[1], o[2], o[3]);
    if (accumulate) { const float4 old = *dst; v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
    *dst = v;
  }
}

}  // namespace cgv

// workspace of the split reduction / the stream-K kernel (cgv_tile_bwd_input_split): per host thread, for launches on ONE
// stream, which are ordered -- they share its tickets (self-resetting) and its partial tiles
namespace cgv {
struct SplitWs { void* ws; size_t bytes; void* stream; };
static thread_local SplitWs g_split_ws = {nullptr, 0, nullptr};
constexpr size_t SPLIT_TICKET_BYTES = 64 * 1024;       // head of the workspace: one ticket per output tile

/* Blocks per CU of the stream-K launch that should compute np problems of `rows` x `cols` outputs over a `red`-deep
 * reduction, or 0: the register-tile kernels.  CGV_OPT_STREAMK: 0 the measured rule, 1 never, 2 / 3 always (1 / 2 per CU). */
static int sk_wanted(int rows, int cols, int red, int np, void* stream) {
  const int opt = option(CGV_OPT_STREAMK);
  if (opt == 1 || !g_split_ws.ws || g_split_ws.stream != stream) return 0;
  if (opt == 2) return 1;
  if (opt == 3) return 2;
  if (opt >= 16) return -opt;                            /* experiments: a grid of exactly `opt` blocks */
  // Measured against the register-tile kernels (tools/probes/sk_ab.py, rotating operands, profiles/r06_gemm_shapes.txt): the
  // stream-K kernel wins where the output is small and the reduction long at many rows -- 2000 x 1800 -> 600: 52.4 against
  // 56.2 us, 2000 x 5400 -> 600: 136 / 158, 2000 x 1200 -> 600 forward: 41.9 / 46.5 -- and loses everywhere else (its parts
  // cost ~10 us per launch: store, ticket, wait, read are four dependent trips to memory; at 704 rows that is half a launch).
  // Pair launches stay on the register tiles (two problems double the operands in flight: 2000 x 5400 pair 382 against 276 us).
  if (np != 1 || rows < 1536) return 0;
  return (cols <= 640 && red >= 1200) ? 1 : 0;
}
static bool sk_aligned(std::initializer_list<const void*> ptrs) {
  uintptr_t u = 0;
  for (const void* p : ptrs) u |= (uintptr_t)p;
  return (u & 15) == 0;
}
}  // namespace cgv

extern "C" {

int cgv_tile_supported(int M, int N, int K) {
  return M >= 1 && N >= 4 && K >= 4 && (N % 4) == 0 && (K % 4) == 0 && (int64_t)M * N < (1ll << 31) &&
         (int64_t)M * K < (1ll << 31) && (int64_t)N * K < (1ll << 31);
}

/* ``second`` != NULL: a pair launch (TileSecond).  Only the register-tile kernels take one; *pair_ok = 0 (nothing launched)
 * tells the caller that this shape runs on the LDS-staged kernels. */
static int tile_fwd_launch(const float* x, const float* W, const float* bias, float* y, float* z, int M, int N, int K, int act,
                           void* stream, const cgv::TileSecond* second, int* pair_ok) {
  hipStream_t st = (hipStream_t)stream;
  const cgv::TileSecond s2 = second ? *second : cgv::TileSecond{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  const unsigned np = second ? 2u : 1u;
  if (pair_ok) *pair_ok = 1;
  if (const int bpc = cgv::sk_wanted(M, N, K, (int)np, stream);
      bpc != 0 && cgv::sk_aligned({x, W, bias, y, z, s2.x, s2.W, s2.bias, s2.y, s2.z})) {
    cgv::SkArgs a{};
    a.np = (int)np; a.M = M; a.N = N;
    a.p[0].A[0] = x; a.p[0].B[0] = W; a.p[0].R[0] = K; a.p[0].out = y; a.p[0].bias = bias; a.p[0].zout = z; a.p[0].act = act;
    if (second) { a.p[1].A[0] = s2.x; a.p[1].B[0] = s2.W; a.p[1].R[0] = K; a.p[1].out = s2.y; a.p[1].bias = s2.bias; a.p[1].zout = s2.z; a.p[1].act = s2.act; }
    if (cgv::sk_launch(a, false, cgv::g_split_ws.ws, cgv::g_split_ws.bytes, cgv::SPLIT_TICKET_BYTES, st, bpc) == 0)
      return cgv::check_launch("cgv_tile_linear_fwd");
  }
  const int tiles32 = ((N + 31) / 32) * ((M + 31) / 32);
  const int tiles64 = ((N + 63) / 64) * ((M + 63) / 64);
  const int lds_min = cgv::option(CGV_OPT_TILE_FWD_LDS_MIN);
  const bool aligned16 = ((((uintptr_t)y | (uintptr_t)z | (uintptr_t)bias)) & 15) == 0;
  // several 64 x 64 tiles per CU: the LDS-staged kernels -- with the three-slab ring for the reductions it is compiled for
  // (K = 600 / 1200: 19 / 38 slabs), else with one slab of look-ahead.  Measured (tools/gemm_shapes.py, rotating operands),
  // ring vs split-reduction tiles: 2000 x 5400 x 600 137 vs 155 us (95 TF/s), 2000 x 1800 52 vs 56, 704 x 5400 53 vs 77,
  // 332 x 5400 32 vs 39, 2000 x 1200 40 vs 43; below ~450 tiles the split-reduction tiles win (704 x 1800: 27 vs 31,
  // 2000 x 600: 24 vs 30 -- a lone block per CU still spends ~1 us per slab on its read -> MFMA -> write -> barrier chain).
  // cgv_set_option(CGV_OPT_TILE_FWD_LDS_MIN, 1): the one-slab kernel for every shape, 3: the ring kernel (tests / A-B)
  if (const int bal = cgv::option(CGV_OPT_TILE_FWD_BAL); bal == 2 || (bal == 1 && tiles32 > 256 && N >= 1200)) {
    // wide layers with more than one 32 x 32 tile per CU: one larger register tile per CU instead.  Measured
    // (tools/gemm_ab.py, rotating operands): 332 x 1800 x 600 17.7 -> 15.7 us, 704 x 1800 26.9 -> 25.0, 288 x 1200 11.8 -> 10.8,
    // 64 x 5400 12.8 -> 10.6, 2000 x 1800 51.7 -> 51.0; NOT at 600 outputs (704 x 600: 12.6 -> 13.3, 2000 x 600: 24.8 -> 26.3).
    // (bal == 2: every shape a compiled tile fits, for tests / A-B)
    int mt = 0, nt = 0;
    if (cgv::bal_pick(M, N, &mt, &nt)) {
      const int MB = (M + 16 * mt - 1) / (16 * mt), NB = (N + 16 * nt - 1) / (16 * nt);
      const dim3 grid(8 * ((MB * NB + 7) / 8), np);
      const size_t lds = (size_t)4 * mt * nt * 1024;
#define CGV_BAL(A, B)                                                                                                      \
  if (mt == A && nt == B) {                                                                                                \
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(cgv::tile_fwd_bal_k<A, B>),                     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
    hipLaunchKernelGGL((cgv::tile_fwd_bal_k<A, B>), grid, dim3(512), lds, st, x, W, bias, y, z, M, N, K, act, MB, NB, s2); \
    return cgv::check_launch("cgv_tile_linear_fwd");                                                                       \
  }
      CGV_BAL(2, 2) CGV_BAL(2, 3) CGV_BAL(3, 2) CGV_BAL(3, 3) CGV_BAL(2, 4) CGV_BAL(4, 2) CGV_BAL(2, 5) CGV_BAL(3, 4)
      CGV_BAL(4, 3) CGV_BAL(4, 4) CGV_BAL(4, 5) CGV_BAL(3, 5) CGV_BAL(2, 6) CGV_BAL(4, 6)
#undef CGV_BAL
    }
  }
  const int slabs32 = (K + 31) / 32;
  const bool ring_ok = aligned16 && (slabs32 == 19 || slabs32 == 38) && lds_min != 1 && !(lds_min >= 5 && lds_min <= 7);
  const bool staged = (ring_ok && (tiles64 >= lds_min || lds_min == 3)) ||
                      (tiles64 >= lds_min && (M >= 1024 || lds_min <= 1) && aligned16 && !(lds_min >= 5 && lds_min <= 7)) ||
                      (lds_min >= 5 && lds_min <= 7);
  if (second && staged) {                       // the LDS-staged kernels take one problem: the pair as two launches of them
    const int rc = tile_fwd_launch(x, W, bias, y, z, M, N, K, act, stream, nullptr, nullptr);
    return rc ? rc : tile_fwd_launch(s2.x, s2.W, s2.bias, s2.y, s2.z, M, N, K, s2.act, stream, nullptr, nullptr);
  }
  if (ring_ok && (tiles64 >= lds_min || lds_min == 3)) {
    const dim3 grid((N + 63) / 64, (M + 63) / 64);
    if (slabs32 == 19) hipLaunchKernelGGL((cgv::tile_fwd_ring_k<19>), grid, dim3(256), 0, st, x, W, bias, y, z, M, N, K, act);
    else hipLaunchKernelGGL((cgv::tile_fwd_ring_k<38>), grid, dim3(256), 0, st, x, W, bias, y, z, M, N, K, act);
  } else if (tiles64 >= lds_min && (M >= 1024 || lds_min <= 1) && aligned16 && !(lds_min >= 5 && lds_min <= 7))
    hipLaunchKernelGGL(cgv::tile_fwd_lds_k, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, x, W, bias, y, z, M, N, K,
                       act);
  else if (lds_min == 5)                       /* A/B: 32 x 64 tiles, 16 waves */
    hipLaunchKernelGGL((cgv::tile_fwd_k<1, 2, 8>), dim3((N + 63) / 64, (M + 31) / 32), dim3(1024), 0, st, x, W, bias, y, z, M, N, K, act);
  else if (lds_min == 6)                       /* A/B: 64 x 32 tiles, 16 waves */
    hipLaunchKernelGGL((cgv::tile_fwd_k<2, 1, 8>), dim3((N + 31) / 32, (M + 63) / 64), dim3(1024), 0, st, x, W, bias, y, z, M, N, K, act);
  else if (lds_min == 7)                       /* A/B: 64 x 64 tiles, 16 waves */
    hipLaunchKernelGGL((cgv::tile_fwd_k<2, 2, 4>), dim3((N + 63) / 64, (M + 63) / 64), dim3(1024), 0, st, x, W, bias, y, z, M, N, K, act);
  else if (tiles32 >= 2048)                    // enough work for several 64 x 64 tiles on every CU (measured: no gain below)
    hipLaunchKernelGGL((cgv::tile_fwd_k<2, 2, 4>), dim3((N + 63) / 64, (M + 63) / 64, np), dim3(1024), 0, st, x, W, bias, y, z, M,
                       N, K, act, s2);
  else                                    // few tiles: 32 x 32, reduction split 8 ways so every SIMD has loads in flight
    hipLaunchKernelGGL((cgv::tile_fwd_k<1, 1, 8>), dim3((N + 31) / 32, (M + 31) / 32, np), dim3(512), 0, st, x, W, bias, y, z, M,
                       N, K, act, s2);
  return cgv::check_launch("cgv_tile_linear_fwd");
}

int cgv_tile_linear_fwd(const float* x, const float* W, const float* bias, float* y, float* z, int M, int N, int K, int act,
                        void* stream) {
  CGV_REQUIRE(x && W && y, "null pointer");
  CGV_REQUIRE(act >= 0 && act <= cgv::CGV_ACT_MAX, "act must be 0 (identity), 1 (swish), 2 (tanh), 3 (relu), 4 / 5 (c + exp(z/2))");
  CGV_REQUIRE(cgv_tile_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)x | (uintptr_t)W)) & 15) == 0, "x and W must be 16-byte aligned");
  return tile_fwd_launch(x, W, bias, y, z, M, N, K, act, stream, nullptr, nullptr);
}

/* Two layers of ONE shape in one launch: y_a = act(x_a W_a^T + b_a), y_b = act(x_b W_b^T + b_b) (x_a may equal x_b).
 * Returns CGV_E_UNSUPPORTED-style non-zero with nothing launched when the shape runs on the LDS-staged kernels
 * (cgv_tile_pair_supported tells beforehand). */
int cgv_tile_pair_supported(int M, int N, int K) {
  /* every tile-supported shape: the stream-K kernel and the register-tile kernels take the second problem in the same
   * launch, the LDS-staged single-problem kernels run the pair as two launches inside the one call */
  return cgv_tile_supported(M, N, K);
}

int cgv_tile_pair_linear_fwd(const float* x_a, const float* W_a, const float* bias_a, float* y_a, float* z_a, const float* x_b,
                             const float* W_b, const float* bias_b, float* y_b, float* z_b, int M, int N, int K, int act_a,
                             int act_b, void* stream) {
  const int act = act_a;
  CGV_REQUIRE(x_a && W_a && y_a && x_b && W_b && y_b, "null pointer");
  CGV_REQUIRE(act_a >= 0 && act_a <= cgv::CGV_ACT_MAX && act_b >= 0 && act_b <= cgv::CGV_ACT_MAX,
              "act must be 0 (identity), 1 (swish), 2 (tanh), 3 (relu), 4 / 5 (c + exp(z/2))");
  CGV_REQUIRE((act_a == 0 || z_a) && (act_b == 0 || z_b), "act != 0 needs the pre-activation output");
  CGV_REQUIRE(cgv_tile_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)x_a | (uintptr_t)W_a | (uintptr_t)x_b | (uintptr_t)W_b | (uintptr_t)y_a | (uintptr_t)y_b |
                 (uintptr_t)z_a | (uintptr_t)z_b | (uintptr_t)bias_a | (uintptr_t)bias_b)) & 15) == 0, "operands must be 16-byte aligned");
  const cgv::TileSecond s2{x_b, W_b, bias_b, y_b, z_b, act_b};
  int ok = 1;
  const int rc = tile_fwd_launch(x_a, W_a, bias_a, y_a, z_a, M, N, K, act, stream, &s2, &ok);
  if (!ok) { cgv::set_error("cgv_tile_pair_linear_fwd: this shape runs on the LDS-staged kernels (no pair launch)"); return CGV_E_UNSUPPORTED; }
  return rc;
}


/* blocks per output tile of the split reduction for `tiles` 16-row tiles over an N-deep reduction (1: unsplit) -- ONE rule
 * for the launcher and for cgv_tile_bwd_input_plan */
static int split_shares(int tiles, int N) {
  const int opt = cgv::option(CGV_OPT_BWD_INPUT_SPLIT);
  if (opt == 1 || tiles > 128 || tiles < 1) return 1;
  int n = opt >= 2 ? opt : 256 / tiles;
  n = n > 4 ? 4 : n;
  while (n > 1 && (N + 15) / 16 / n < 16) --n;                            // at least one step per wave and share
  return n < 1 ? 1 : n;
}

static int tile_bwd_input_launch(const float* g, const float* z, int act, const float* W, float* gx, int M, int N, int K,
                                 void* stream, const char* what, const float* add = nullptr,
                                 cgv::BcastAdd bc = cgv::BcastAdd{nullptr, nullptr, nullptr, 0},
                                 const cgv::BwdSecond* second = nullptr,
                                 cgv::BwdSource more = cgv::BwdSource{nullptr, nullptr, nullptr, 0},
                                 cgv::OutAct oa = cgv::OutAct{nullptr, 0, nullptr, 0},
                                 cgv::NormStackOut ns = cgv::NormStackOut{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0}) {
  hipStream_t st = (hipStream_t)stream;
  const cgv::BwdSecond s2 = second ? *second : cgv::BwdSecond{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  const unsigned np = second ? 2u : 1u;
  // no activation in the operand loads (act_downstream chains carry none), operands 16-byte aligned: the stream-K kernel
  if (const int bpc = cgv::sk_wanted(M, K, N, (int)np, stream);
      bpc != 0 && !ns.stack && act == 0 && (!more.g || more.act == 0) && (!second || s2.act == 0) &&
      cgv::sk_aligned({g, W, gx, add, bc.src, more.g, more.W, oa.z, oa.z2, s2.g, s2.W, s2.gx, s2.add})) {
    cgv::SkArgs a{};
    a.np = (int)np; a.M = M; a.N = K;
    cgv::SkProblem& p0 = a.p[0];
    p0.A[0] = g; p0.B[0] = W; p0.R[0] = N; p0.out = gx; p0.add = add;
    if (more.g) { p0.A[1] = more.g; p0.B[1] = more.W; p0.R[1] = N; }
    p0.bc_src = bc.src; p0.bc_row2seg = bc.row2seg; p0.bc_rowptr = bc.rowptr; p0.bc_mean = bc.mean;
    p0.oz = oa.z; p0.oact = oa.act;
    if (second) {
      cgv::SkProblem& p1 = a.p[1];
      p1.A[0] = s2.g; p1.B[0] = s2.W; p1.R[0] = N; p1.out = s2.gx; p1.add = s2.add; p1.oz = oa.z2; p1.oact = oa.act2;
    }
    if (cgv::sk_launch(a, true, cgv::g_split_ws.ws, cgv::g_split_ws.bytes, cgv::SPLIT_TICKET_BYTES, st, bpc) == 0)
      return cgv::check_launch(what);
  }
  const int kt = (K + 63) / 64;
  const int blocks32 = kt * ((M + 31) / 32);
  const int blocks16 = kt * ((M + 15) / 16);
  int waves = 8;
  if (const int o = cgv::option(CGV_OPT_BWD_INPUT_WAVES); o > 0 && o != 32) waves = o;          // experiments only
  else if (blocks16 < 128 && N >= 1024) waves = 16;
  if (cgv::option(CGV_OPT_BWD_INPUT_WAVES) == 32)          /* A/B: 32-row tiles, 8 waves */
    hipLaunchKernelGGL((cgv::tile_bwd_input_k<2, 8>), dim3(kt, (M + 31) / 32, np), dim3(512), 0, st, g, W, gx, M, N, K, z, act, add, bc, s2, more, oa, cgv::SplitN{nullptr, nullptr, 1}, ns);
  else if (blocks32 >= 200 && blocks32 < 512 && waves == 8)
    // 200 .. 511 32-row tiles (704 rows x 600 / 1200 columns): still 32-row tiles, with the reduction split over 8 waves --
    // half the weight re-reads of the 16-row tiles (704 x 1800 x 600: 22.9 against 25.0 us, 704 x 5400: 54.6 / 64.1;
    // at 332 rows the 16-row tiles win, 14.5 against 22.1 us: tools/gemm_shapes.py)
    hipLaunchKernelGGL((cgv::tile_bwd_input_k<2, 8>), dim3(kt, (M + 31) / 32, np), dim3(512), 0, st, g, W, gx, M, N, K, z, act, add, bc, s2, more, oa, cgv::SplitN{nullptr, nullptr, 1}, ns);
  else if (blocks32 >= 512)               // enough 32-row tiles to fill the chip: halve the weight re-reads
    hipLaunchKernelGGL((cgv::tile_bwd_input_k<2, 4>), dim3(kt, (M + 31) / 32, np), dim3(256), 0, st, g, W, gx, M, N, K, z, act, add, bc, s2, more, oa, cgv::SplitN{nullptr, nullptr, 1}, ns);
  else if (waves == 16) {                 // few output tiles and a long reduction (96 bead rows x 5400 columns: 60 blocks):
    // 16 waves per block split it -- 60 blocks of 8 waves left three quarters of the chip idle (17.8 us per call) -- and,
    // with a registered workspace, 2 - 4 blocks per tile split it further (SplitN)
    cgv::SplitN sp{nullptr, nullptr, 1};
    const int tiles = blocks16 * (int)np;
    const cgv::SplitWs& w = cgv::g_split_ws;
    if (w.ws && w.stream == stream) {
      const int n = split_shares(tiles, N);
      const size_t need = cgv::SPLIT_TICKET_BYTES + (size_t)tiles * n * (16 * 64) * sizeof(float);
      if (n > 1 && need <= w.bytes && (size_t)tiles * sizeof(unsigned) <= cgv::SPLIT_TICKET_BYTES) {
        sp.ticket = reinterpret_cast<unsigned*>(w.ws);
        sp.part = reinterpret_cast<float*>(reinterpret_cast<char*>(w.ws) + cgv::SPLIT_TICKET_BYTES);
        sp.n = n;
      }
    }
    hipLaunchKernelGGL((cgv::tile_bwd_input_k<1, 16>), dim3(kt, ((M + 15) / 16) * sp.n, np), dim3(1024), 0, st, g, W, gx, M, N, K, z, act,
                       add, bc, s2, more, oa, sp, ns);
  }
  else
    hipLaunchKernelGGL((cgv::tile_bwd_input_k<1, 8>), dim3(kt, (M + 15) / 16, np), dim3(512), 0, st, g, W, gx, M, N, K, z, act, add, bc, s2, more, oa, cgv::SplitN{nullptr, nullptr, 1}, ns);
  return cgv::check_launch(what);
}

/* What a backward-input launch of this shape does with a registered workspace: *shares = blocks per output tile of the
 * split reduction (1: unsplit), *streamk = 1 when the stream-K kernel takes it (np = 1 single launch, 2 pair launch).
 * The dispatch rules of tile_bwd_input_launch, for callers that choose between this entry point and the row-split kernel
 * (primitives._LinearFn, ops._dense_bwd_input): the workspace is per (host thread, stream) state the call itself cannot see. */
int cgv_tile_bwd_input_plan(int M, int N, int K, int np, int* shares, int* streamk) {
  CGV_REQUIRE(shares && streamk && np >= 1 && np <= 2, "bad argument");
  CGV_REQUIRE(cgv_tile_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  *shares = 1;
  const int opt_sk = cgv::option(CGV_OPT_STREAMK);
  *streamk = (opt_sk == 2 || opt_sk == 3 || opt_sk >= 16) ? 1 : (opt_sk == 1 ? 0 : (np == 1 && M >= 1536 && K <= 640 && N >= 1200));
  if (*streamk) return 0;
  const int kt = (K + 63) / 64, blocks16 = kt * ((M + 15) / 16);
  const int ow = cgv::option(CGV_OPT_BWD_INPUT_WAVES);
  const bool waves16 = (ow > 0 && ow != 32) ? ow == 16 : (blocks16 < 128 && N >= 1024);
  const int blocks32 = kt * ((M + 31) / 32);
  if (ow == 32 || (blocks32 >= 200 && !waves16) || !waves16) return 0;      // (the 8-wave / 32-row kernels never split)
  *shares = split_shares(blocks16 * np, N);
  return 0;
}

/* Registers the workspace of the split reduction for the CALLING host thread's launches on `stream` (ws = NULL: none).
 * ws: zero-filled once by the caller (its first 64 KB are self-resetting tickets), 16-byte aligned, >= 64 KB + 2 MB. */
int cgv_tile_bwd_input_split(void* ws, size_t bytes, void* stream) {
  CGV_REQUIRE(!ws || ((((uintptr_t)ws) & 15) == 0 && bytes >= cgv::SPLIT_TICKET_BYTES + (2u << 20)), "workspace too small / misaligned");
  cgv::g_split_ws = cgv::SplitWs{ws, ws ? bytes : 0, stream};
  return 0;
}

int cgv_tile_linear_bwd_input(const float* g, const float* W, float* gx, int M, int N, int K, void* stream) {
  CGV_REQUIRE(g && W && gx, "null pointer");
  CGV_REQUIRE(cgv_tile_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)g | (uintptr_t)W | (uintptr_t)gx)) & 15) == 0, "g, W and gx must be 16-byte aligned");
  return tile_bwd_input_launch(g, nullptr, 0, W, gx, M, N, K, stream, "cgv_tile_linear_bwd_input");
}

/* gx = (gy * act'(z)) W with the activation derivative applied in the operand loads (no cgv_dense_grad_prepare pass). */
int cgv_tile_linear_bwd_input_act(const float* gy, const float* z, const float* W, float* gx, int M, int N, int K, int act,
                                  void* stream) {
  CGV_REQUIRE(gy && W && gx, "null pointer");
  CGV_REQUIRE(act == 0 || (act >= 1 && act <= cgv::CGV_ACT_MAX && z), "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(cgv_tile_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)gy | (uintptr_t)z | (uintptr_t)W | (uintptr_t)gx)) & 15) == 0, "operands must be 16-byte aligned");
  return tile_bwd_input_launch(gy, act ? z : nullptr, act, W, gx, M, N, K, stream, "cgv_tile_linear_bwd_input_act");
}

/* gx = add + (gy * act'(z)) W: as above with a second gradient of the same input added in the store epilogue. */
int cgv_tile_linear_bwd_input_act_add(const float* gy, const float* z, const float* W, const float* add, float* gx, int M, int N,
                                      int K, int act, void* stream) {
  CGV_REQUIRE(gy && W && gx && add, "null pointer");
  CGV_REQUIRE(act == 0 || (act >= 1 && act <= cgv::CGV_ACT_MAX && z), "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(cgv_tile_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)gy | (uintptr_t)z | (uintptr_t)W | (uintptr_t)gx | (uintptr_t)add)) & 15) == 0, "operands must be 16-byte aligned");
  return tile_bwd_input_launch(gy, act ? z : nullptr, act, W, gx, M, N, K, stream, "cgv_tile_linear_bwd_input_act_add", add);
}

/* ... plus a gradient that lives as ONE ROW PER SEGMENT of the rows (the backward of a scatter_mean / scatter_add of the
 * same input): gx[m, :] += seg_grad[row2seg[m], :] (/ max(len(segment), 1) when mean).  add may be NULL here. */
int cgv_tile_linear_bwd_input_act_add_bcast(const float* gy, const float* z, const float* W, const float* add, const float* seg_grad,
                                            const int64_t* row2seg, const int32_t* seg_rowptr, int mean, float* gx, int M, int N,
                                            int K, int act, void* stream) {
  CGV_REQUIRE(gy && W && gx && seg_grad && row2seg && seg_rowptr, "null pointer");
  CGV_REQUIRE(act == 0 || (act >= 1 && act <= cgv::CGV_ACT_MAX && z), "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(cgv_tile_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)gy | (uintptr_t)z | (uintptr_t)W | (uintptr_t)gx | (uintptr_t)add | (uintptr_t)seg_grad)) & 15) == 0,
              "operands must be 16-byte aligned");
  return tile_bwd_input_launch(gy, act ? z : nullptr, act, W, gx, M, N, K, stream, "cgv_tile_linear_bwd_input_act_add_bcast", add,
                               cgv::BcastAdd{seg_grad, row2seg, seg_rowptr, mean});
}

/* gx = add + (gy_a * act_a'(z_a)) W_a + (gy_b * act_b'(z_b)) W_b [+ seg_grad spread over the rows]: the input gradient of TWO
 * layers of one shape that read the same input, as one product (add, seg_grad may be NULL; seg_* as in
 * cgv_tile_linear_bwd_input_act_add_bcast). */
int cgv_tile_linear_bwd_input_sum2(const float* gy_a, const float* z_a, const float* W_a, const float* gy_b, const float* z_b,
                                   const float* W_b, const float* add, const float* seg_grad, const int64_t* row2seg,
                                   const int32_t* seg_rowptr, int mean, float* gx, int M, int N, int K, int act_a, int act_b,
                                   void* stream) {
  CGV_REQUIRE(gy_a && W_a && gy_b && W_b && gx, "null pointer");
  CGV_REQUIRE((act_a == 0 || (act_a >= 1 && act_a <= cgv::CGV_ACT_MAX && z_a)) && (act_b == 0 || (act_b >= 1 && act_b <= cgv::CGV_ACT_MAX && z_b)),
              "act != 0 needs the saved pre-activation");
  CGV_REQUIRE(!seg_grad || (row2seg && seg_rowptr), "seg_grad needs its row -> segment map and the segments' row pointers");
  CGV_REQUIRE(cgv_tile_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)gy_a | (uintptr_t)z_a | (uintptr_t)W_a | (uintptr_t)gy_b | (uintptr_t)z_b | (uintptr_t)W_b | (uintptr_t)gx |
                 (uintptr_t)add | (uintptr_t)seg_grad)) & 15) == 0, "operands must be 16-byte aligned");
  return tile_bwd_input_launch(gy_a, act_a ? z_a : nullptr, act_a, W_a, gx, M, N, K, stream, "cgv_tile_linear_bwd_input_sum2", add,
                               cgv::BcastAdd{seg_grad, row2seg, seg_rowptr, mean}, nullptr,
                               cgv::BwdSource{gy_b, W_b, act_b ? z_b : nullptr, act_b});
}

/* The backward-input products of two layers of ONE shape in one launch: gx_a = add_a + (gy_a * act'(z_a)) W_a, likewise b
 * (add_* may be NULL; outputs must not alias). */
int cgv_tile_pair_linear_bwd_input(const float* gy_a, const float* z_a, const float* W_a, const float* add_a, float* gx_a,
                                   const float* gy_b, const float* z_b, const float* W_b, const float* add_b, float* gx_b, int M,
                                   int N, int K, int act_a, int act_b, void* stream) {
  const int act = act_a;
  CGV_REQUIRE(gy_a && W_a && gx_a && gy_b && W_b && gx_b && gx_a != gx_b, "null pointer / aliased outputs");
  CGV_REQUIRE((act_a == 0 || (act_a >= 1 && act_a <= cgv::CGV_ACT_MAX && z_a)) && (act_b == 0 || (act_b >= 1 && act_b <= cgv::CGV_ACT_MAX && z_b)),
              "act != 0 needs the saved pre-activation");
  CGV_REQUIRE(cgv_tile_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)gy_a | (uintptr_t)z_a | (uintptr_t)W_a | (uintptr_t)gx_a | (uintptr_t)add_a | (uintptr_t)gy_b |
                 (uintptr_t)z_b | (uintptr_t)W_b | (uintptr_t)gx_b | (uintptr_t)add_b)) & 15) == 0, "operands must be 16-byte aligned");
  const cgv::BwdSecond s2{gy_b, W_b, gx_b, act_b ? z_b : nullptr, add_b, act_b};
  return tile_bwd_input_launch(gy_a, act ? z_a : nullptr, act, W_a, gx_a, M, N, K, stream, "cgv_tile_pair_linear_bwd_input", add_a,
                               cgv::BcastAdd{nullptr, nullptr, nullptr, 0}, &s2);
}

/* cgv_tile_linear_bwd_input_act[_add] (add may be NULL) whose OUTPUT is multiplied by act_out'(z_out), z_out [M, K] the
 * pre-activation of the layer that produced this layer's input: the stored gx is the gradient of that pre-activation, and
 * the producing layer's backward then runs with act = 0 (no activation derivative in its operand loads). */
int cgv_tile_linear_bwd_input_out(const float* gy, const float* z, const float* W, const float* add, float* gx, int M, int N,
                                  int K, int act, const float* z_out, int act_out, void* stream) {
  CGV_REQUIRE(gy && W && gx && z_out, "null pointer");
  CGV_REQUIRE(act == 0 || (act >= 1 && act <= cgv::CGV_ACT_MAX && z), "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(act_out >= 1 && act_out <= cgv::CGV_ACT_MAX, "act_out must name an activation");
  CGV_REQUIRE(cgv_tile_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)gy | (uintptr_t)z | (uintptr_t)W | (uintptr_t)gx | (uintptr_t)add | (uintptr_t)z_out)) & 15) == 0,
              "operands must be 16-byte aligned");
  return tile_bwd_input_launch(gy, act ? z : nullptr, act, W, gx, M, N, K, stream, "cgv_tile_linear_bwd_input_out", add,
                               cgv::BcastAdd{nullptr, nullptr, nullptr, 0}, nullptr, cgv::BwdSource{nullptr, nullptr, nullptr, 0},
                               cgv::OutAct{z_out, act_out, nullptr, 0});
}

/* UpdateBlock's backward through s_dense.0 AND the norm / stack step in one launch (conv.py:600-603): the product
 * g_stack = (gy * act'(z)) W (M beads, K = 2 F columns) is not stored; its columns k < F become g_s = g_stack + g_res (g_res
 * may be NULL), its columns F + f go through d ||Vv|| / d Vv into gVv[3 m + xyz, f] (rows of ld floats; accumulate != 0 adds
 * to what update_gate_bwd left there).  stack [M, 2 F] as cgv_update_norm_stack_fwd wrote it, Vv rows of ld floats. */
int cgv_tile_linear_bwd_input_norm_stack(const float* gy, const float* z, const float* W, int M, int N, int K, int act,
                                         const float* stack, const float* Vv, const float* g_res, float* g_s, float* gVv, int ld,
                                         int accumulate, void* stream) {
  CGV_REQUIRE(gy && W && stack && Vv && g_s && gVv, "null pointer");
  CGV_REQUIRE(act == 0 || (act >= 1 && act <= cgv::CGV_ACT_MAX && z), "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(cgv_tile_supported(M, N, K) && K % 8 == 0 && ld >= K / 2 && ld % 4 == 0, "unsupported shape (need K = 2 F, F % 4 == 0, ld % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)gy | (uintptr_t)z | (uintptr_t)W | (uintptr_t)stack | (uintptr_t)Vv | (uintptr_t)g_res | (uintptr_t)g_s |
                 (uintptr_t)gVv)) & 15) == 0, "operands must be 16-byte aligned");
  return tile_bwd_input_launch(gy, act ? z : nullptr, act, W, g_s /* never written as gx */, M, N, K, stream,
                               "cgv_tile_linear_bwd_input_norm_stack", nullptr, cgv::BcastAdd{nullptr, nullptr, nullptr, 0}, nullptr,
                               cgv::BwdSource{nullptr, nullptr, nullptr, 0}, cgv::OutAct{nullptr, 0, nullptr, 0},
                               cgv::NormStackOut{stack, Vv, g_res, g_s, gVv, K / 2, ld, accumulate});
}

/* cgv_tile_pair_linear_bwd_input with the outputs multiplied by act_out_*'(z_out_*) (either may be NULL / 0: that output
 * is stored as it is). */
int cgv_tile_pair_linear_bwd_input_out(const float* gy_a, const float* z_a, const float* W_a, const float* add_a, float* gx_a,
                                       const float* gy_b, const float* z_b, const float* W_b, const float* add_b, float* gx_b,
                                       int M, int N, int K, int act_a, int act_b, const float* z_out_a, int act_out_a,
                                       const float* z_out_b, int act_out_b, void* stream) {
  const int act = act_a;
  CGV_REQUIRE(gy_a && W_a && gx_a && gy_b && W_b && gx_b && gx_a != gx_b, "null pointer / aliased outputs");
  CGV_REQUIRE((act_a == 0 || (act_a >= 1 && act_a <= cgv::CGV_ACT_MAX && z_a)) && (act_b == 0 || (act_b >= 1 && act_b <= cgv::CGV_ACT_MAX && z_b)),
              "act != 0 needs the saved pre-activation");
  CGV_REQUIRE((!z_out_a || (act_out_a >= 1 && act_out_a <= cgv::CGV_ACT_MAX)) && (!z_out_b || (act_out_b >= 1 && act_out_b <= cgv::CGV_ACT_MAX)),
              "z_out needs its activation code");
  CGV_REQUIRE(cgv_tile_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)gy_a | (uintptr_t)z_a | (uintptr_t)W_a | (uintptr_t)gx_a | (uintptr_t)add_a | (uintptr_t)gy_b |
                 (uintptr_t)z_b | (uintptr_t)W_b | (uintptr_t)gx_b | (uintptr_t)add_b | (uintptr_t)z_out_a | (uintptr_t)z_out_b)) & 15) == 0,
              "operands must be 16-byte aligned");
  const cgv::BwdSecond s2{gy_b, W_b, gx_b, act_b ? z_b : nullptr, add_b, act_b};
  return tile_bwd_input_launch(gy_a, act ? z_a : nullptr, act, W_a, gx_a, M, N, K, stream, "cgv_tile_pair_linear_bwd_input_out", add_a,
                               cgv::BcastAdd{nullptr, nullptr, nullptr, 0}, &s2, cgv::BwdSource{nullptr, nullptr, nullptr, 0},
                               cgv::OutAct{z_out_a, z_out_a ? act_out_a : 0, z_out_b, z_out_b ? act_out_b : 0});
}

int cgv_tile_linear_wgrad(const float* g, const float* x, float* gW, int M, int N, int K, int accumulate, void* stream) {
  CGV_REQUIRE(g && x && gW, "null pointer");
  CGV_REQUIRE(cgv_tile_supported(M, N, K), "unsupported shape (need N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)x | (uintptr_t)gW)) & 15) == 0, "x and gW must be 16-byte aligned");
  const dim3 grid((K + 63) / 64, (N + 15) / 16);
  if (grid.x * grid.y >= 1024 || M <= 128)
    hipLaunchKernelGGL((cgv::tile_wgrad_k<4>), grid, dim3(256), 0, (hipStream_t)stream, g, x, gW, M, N, K, accumulate);
  else
    hipLaunchKernelGGL((cgv::tile_wgrad_k<8>), grid, dim3(512), 0, (hipStream_t)stream, g, x, gW, M, N, K, accumulate);
  return cgv::check_launch("cgv_tile_linear_wgrad");
}

}  // extern "C"
