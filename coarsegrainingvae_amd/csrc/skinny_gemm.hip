// Skinny fp32 GEMMs for the node-level Dense layers on the bead graph (M = beads or 3*beads
// rows, 12..64; K, N = n_basis multiples, 600..5400).
//
// Reference: Dense / nn.Linear forward (CoarseGrainingVAE/modules.py:103-114) and its autograd
// backward, called ~220 times per training step of the chignolin config.  With M <= 64 these
// products are weight-streaming (GEMV-like): the [N,K] weight (1.4 - 13 MB) is read once, the
// arithmetic is negligible.  hipBLASLt's tiled kernels take 10-24 us for them (profiles/r01c);
// the kernels below stream the weight once at HBM/L2 speed:
//   fwd        y[M,N]  = x[M,K] W[N,K]^T + b      v_mfma_f32_16x16x4_f32, W rows as A, x^T as B
//   bwd_input  gx[M,K] = gy[M,N] W[N,K]           same instruction, W^T as A, split over N
//   bwd_weight gW[N,K] (+)= gy[M,N]^T x[M,K]      write-bound outer products, packed VALU
// f32-input MFMA is an exact fp32 FMA chain (bitwise an fmaf loop), so parity is unaffected.
//
// MFMA 16x16x4 f32 operand map (cdna_hip_programming.md 3): lane l holds A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15]; D: col j = l&15, row i = 4*(l>>4) + reg.  Each lane loads a float4 that
// is contiguous in memory and feeds its four components to four MFMAs whose k (fwd) or i
// (bwd_input) index is the component -- any consistent bijection is legal -- so every weight
// load instruction is 16 rows x 64 B (fwd) or 4 rows x 256 B (bwd_input) of contiguous bytes.
#include "cgv_common.h"

namespace cgv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ldg4_or_zero(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------ fwd
// grid = ceil(N/16) blocks, block = 256 = 4 waves splitting K; MB = ceil(M/16) accumulators.
template <int MB>
__global__ __launch_bounds__(256) void skinny_fwd_k(const float* __restrict__ x, const float* __restrict__ W,
                                                    const float* __restrict__ bias, float* __restrict__ y, int M, int N,
                                                    int K) {
  __shared__ float red[3][MB][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int nrow = n0 + i;
  const bool nok = nrow < N;
  const float* wrow = W + (size_t)(nok ? nrow : 0) * K;
  // K split into 4 contiguous ranges of whole 16-float steps
  const int steps = (K + 15) / 16;
  const int per = (steps + 3) / 4;
  const int s_beg = wave * per, s_end = min(s_beg + per, steps);
  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int s = s_beg; s < s_end; ++s) {
    const int k = s * 16 + 4 * q;
    const bool kok = k < K;                       // K % 4 == 0: a float4 is entirely in or out
    const float4 a = ldg4_or_zero(wrow + k, nok && kok);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int m = mb * 16 + i;
      const float4 b = ldg4_or_zero(x + (size_t)(m < M ? m : 0) * K + k, m < M && kok);
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[mb], 0, 0, 0);
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[mb], 0, 0, 0);
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[mb], 0, 0, 0);
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[mb], 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 1][mb][r][lane] = acc[mb][r];
  }
  __syncthreads();
  if (wave != 0) return;
  // D[i = n_local = 4q + r][j = m_local = lane & 15]: 4 consecutive n per lane -> one 16-byte store
  const int n = n0 + 4 * q;
  if (n >= N) return;                              // N % 4 == 0
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bv = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mb * 16 + i;
    f32x4 t = acc[mb];
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] += red[w][mb][r][lane];
    if (m < M) *reinterpret_cast<float4*>(y + (size_t)m * N + n) = make_float4(t[0] + bv.x, t[1] + bv.y, t[2] + bv.z, t[3] + bv.w);
  }
}

// ------------------------------------------------------------------ bwd_input
// gx[m, k] = sum_n gy[m, n] W[n, k].  A wave owns 64 consecutive k (four MFMAs per 4-row step of n:
// component c of the lane's float4 is output row k = k0 + 4 (l&15) + c); the 4 waves of a block
// and `nsplit` blocks along grid.y split N; partials part[split][M][K] are summed by a second
// launch in a fixed order.  grid = (ceil(K/64), nsplit), block = 256.
template <int MB>
__global__ __launch_bounds__(256) void skinny_bwd_input_k(const float* __restrict__ gy, const float* __restrict__ W,
                                                          float* __restrict__ part, int M, int N, int K, int n_per_block) {
  __shared__ float red[3][MB][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int k0 = blockIdx.x * 64;
  const int k = k0 + 4 * i;
  const bool kok = k < K;
  const int nb_beg = blockIdx.y * n_per_block;
  const int nb_end = min(nb_beg + n_per_block, N);
  // this wave's quarter of the block's n range, in whole steps of 4 rows
  const int steps = (nb_end - nb_beg + 3) / 4;
  const int per = (steps + 3) / 4;
  const int s_beg = wave * per, s_end = min(s_beg + per, steps);
  f32x4 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[mb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int s = s_beg; s < s_end; ++s) {
    const int n = nb_beg + 4 * s + q;
    const bool nok = n < nb_end;
    const float4 a = ldg4_or_zero(W + (size_t)(nok ? n : 0) * K + (kok ? k : 0), nok && kok);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int m = mb * 16 + i;
      const float b = (nok && m < M) ? gy[(size_t)m * N + n] : 0.f;       // B[kk = q][j = m]
      acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b, acc[mb][0], 0, 0, 0);
      acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b, acc[mb][1], 0, 0, 0);
      acc[mb][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b, acc[mb][2], 0, 0, 0);
      acc[mb][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b, acc[mb][3], 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 1][mb][c * 4 + r][lane] = acc[mb][c][r];
  }
  __syncthreads();
  if (wave != 0) return;
  // MFMA c, register r: output row i' = 4q + r  ->  k = k0 + 4 i' + c = k0 + 16 q + 4 r + c ; column j = m
  float* out = part + (size_t)blockIdx.y * M * K;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mb * 16 + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        t[c] = acc[mb][c][r];
#pragma unroll
        for (int w = 0; w < 3; ++w) t[c] += red[w][mb][c * 4 + r][lane];
      }
      const int kk = k0 + 16 * q + 4 * r;
      if (m < M && kk < K) *reinterpret_cast<float4*>(out + (size_t)m * K + kk) = make_float4(t[0], t[1], t[2], t[3]);
    }
  }
}

__global__ __launch_bounds__(256) void skinny_sum_partials(const float* __restrict__ part, int nsplit, size_t count4,
                                                           float* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= count4) return;
  const float4* p = reinterpret_cast<const float4*>(part) + idx;
  float4 acc = p[0];
  for (int s = 1; s < nsplit; ++s) {
    const float4 t = p[(size_t)s * count4];
    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
  }
  reinterpret_cast<float4*>(out)[idx] = acc;
}

// ------------------------------------------------------------------ bwd_weight
// gW[n, k] (+)= sum_m gy[m, n] x[m, k];  gb[n] (+)= sum_m gy[m, n].  Write-bound: every output
// float4 is produced by one thread from an LDS-staged x tile (M x tile_w floats) and the block's
// gy columns; ROWS output rows per block along grid.x, k tiles along grid.y, one float4 of k per
// thread.
template <int ROWS>
__global__ __launch_bounds__(256) void skinny_bwd_weight_k(const float* __restrict__ gy, const float* __restrict__ x,
                                                           float* __restrict__ gW, float* __restrict__ gb, int M, int N,
                                                           int K, int tile_w, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                          // [M][tile_w]
  float* gs = smem + (size_t)M * tile_w;     // [M][ROWS]
  const int t = threadIdx.x;
  const int n0 = blockIdx.x * ROWS;
  const int k = blockIdx.y * tile_w + 4 * t;
  const bool kok = 4 * t < tile_w && k < K;
  if (4 * t < tile_w)
    for (int m = 0; m < M; ++m)
      *reinterpret_cast<float4*>(xs + (size_t)m * tile_w + 4 * t) = ldg4_or_zero(x + (size_t)m * K + (kok ? k : 0), kok);
  for (int idx = t; idx < M * ROWS; idx += blockDim.x) {
    const int m = idx / ROWS, r = idx - m * ROWS;
    gs[idx] = (n0 + r < N) ? gy[(size_t)m * N + n0 + r] : 0.f;
  }
  __syncthreads();
  if (kok) {
    float4 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = 0; m < M; ++m) {
      const float4 xv = *reinterpret_cast<const float4*>(xs + (size_t)m * tile_w + 4 * t);
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const float g = gs[m * ROWS + r];              // LDS broadcast
        acc[r].x = fmaf(g, xv.x, acc[r].x); acc[r].y = fmaf(g, xv.y, acc[r].y);
        acc[r].z = fmaf(g, xv.z, acc[r].z); acc[r].w = fmaf(g, xv.w, acc[r].w);
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (n0 + r < N) {
        float4* dst = reinterpret_cast<float4*>(gW + (size_t)(n0 + r) * K + k);
        float4 o = acc[r];
        if (accumulate) { const float4 old = *dst; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *dst = o;
      }
    }
  }
  if (gb && blockIdx.y == 0 && t < ROWS && n0 + t < N) {
    float sum = 0.f;
    for (int m = 0; m < M; ++m) sum += gs[m * ROWS + t];
    gb[n0 + t] = accumulate ? gb[n0 + t] + sum : sum;
  }
}

static inline int bwd_input_splits(int N, int K) {
  const int kblocks = (K + 63) / 64;
  int s = (N + 255) / 256;                      // about 256 rows of W per block ...
  const int want = (160 + kblocks - 1) / kblocks;   // ... but at least ~160 blocks in flight
  if (s < want) s = want;
  const int cap = (N + 15) / 16;                // never fewer than 16 rows per block
  if (s > cap) s = cap;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}
static inline int bwd_input_splits(int N) { return bwd_input_splits(N, 64 * 160); }



}  // namespace cgv

extern "C" {

int cgv_skinny_max_rows(void) { return 64; }

int cgv_skinny_supported(int M, int N, int K) {
  return M >= 1 && M <= 64 && N >= 4 && K >= 4 && (N % 4) == 0 && (K % 4) == 0;
}

int cgv_skinny_linear_fwd(const float* x, const float* W, const float* bias, float* y, int M, int N, int K, void* stream) {
  CGV_REQUIRE(x && W && y, "null pointer");
  CGV_REQUIRE(cgv_skinny_supported(M, N, K), "unsupported shape (need M <= 64, N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)x | (uintptr_t)W | (uintptr_t)y | (uintptr_t)bias)) & 15) == 0, "operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((N + 15) / 16), block(256);
  switch ((M + 15) / 16) {
    case 1: hipLaunchKernelGGL(cgv::skinny_fwd_k<1>, grid, block, 0, st, x, W, bias, y, M, N, K); break;
    case 2: hipLaunchKernelGGL(cgv::skinny_fwd_k<2>, grid, block, 0, st, x, W, bias, y, M, N, K); break;
    case 3: hipLaunchKernelGGL(cgv::skinny_fwd_k<3>, grid, block, 0, st, x, W, bias, y, M, N, K); break;
    default: hipLaunchKernelGGL(cgv::skinny_fwd_k<4>, grid, block, 0, st, x, W, bias, y, M, N, K); break;
  }
  return cgv::check_launch("cgv_skinny_linear_fwd");
}

size_t cgv_skinny_bwd_input_workspace_bytes(int M, int N, int K) {
  return sizeof(float) * (size_t)32 * M * K + 256;   /* upper bound on the N-splits */
}

int cgv_skinny_linear_bwd_input(const float* gy, const float* W, float* gx, int M, int N, int K, void* workspace,
                                size_t workspace_bytes, void* stream) {
  CGV_REQUIRE(gy && W && gx && workspace, "null pointer");
  CGV_REQUIRE(cgv_skinny_supported(M, N, K), "unsupported shape (need M <= 64, N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)W | (uintptr_t)gx | (uintptr_t)workspace)) & 15) == 0, "operands must be 16-byte aligned");
  if (workspace_bytes < cgv_skinny_bwd_input_workspace_bytes(M, N, K)) {
    cgv::set_error("cgv_skinny_linear_bwd_input: workspace too small");
    return CGV_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nsplit = cgv::bwd_input_splits(N, K);
  int npb = (N + nsplit - 1) / nsplit;
  npb = (npb + 3) & ~3;
  float* part = reinterpret_cast<float*>(workspace);
  float* target = nsplit == 1 ? gx : part;
  const dim3 grid((K + 63) / 64, nsplit), block(256);
  switch ((M + 15) / 16) {
    case 1: hipLaunchKernelGGL(cgv::skinny_bwd_input_k<1>, grid, block, 0, st, gy, W, target, M, N, K, npb); break;
    case 2: hipLaunchKernelGGL(cgv::skinny_bwd_input_k<2>, grid, block, 0, st, gy, W, target, M, N, K, npb); break;
    case 3: hipLaunchKernelGGL(cgv::skinny_bwd_input_k<3>, grid, block, 0, st, gy, W, target, M, N, K, npb); break;
    default: hipLaunchKernelGGL(cgv::skinny_bwd_input_k<4>, grid, block, 0, st, gy, W, target, M, N, K, npb); break;
  }
  if (nsplit > 1) {
    const size_t count4 = (size_t)M * K / 4;
    hipLaunchKernelGGL(cgv::skinny_sum_partials, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, st, part, nsplit,
                       count4, gx);
  }
  return cgv::check_launch("cgv_skinny_linear_bwd_input");
}

int cgv_skinny_linear_bwd_weight(const float* gy, const float* x, float* gW, float* gb, int M, int N, int K,
                                 int accumulate, void* stream) {
  CGV_REQUIRE(gy && x && gW, "null pointer");
  CGV_REQUIRE(cgv_skinny_supported(M, N, K), "unsupported shape (need M <= 64, N % 4 == 0, K % 4 == 0)");
  CGV_REQUIRE(((((uintptr_t)x | (uintptr_t)gW)) & 15) == 0, "operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  constexpr int ROWS = 16;
  // k tile: as wide as a 60 KiB LDS budget for the x tile allows, at most 256 float4 (one per thread)
  int max_t4 = (15000 / M) / 4;
  if (max_t4 > 256) max_t4 = 256;
  const int k4 = K / 4;
  const int ntiles = (k4 + max_t4 - 1) / max_t4;
  const int per_t4 = (k4 + ntiles - 1) / ntiles;
  const int threads = ((per_t4 + 63) / 64) * 64;
  const int tile_w = per_t4 * 4;
  const dim3 grid((N + ROWS - 1) / ROWS, ntiles), block(threads);
  const size_t lds = sizeof(float) * ((size_t)M * tile_w + (size_t)M * ROWS);
  hipLaunchKernelGGL(cgv::skinny_bwd_weight_k<ROWS>, grid, block, lds, st, gy, x, gW, gb, M, N, K, tile_w, accumulate);
  return cgv::check_launch("cgv_skinny_linear_bwd_weight");
}

}  // extern "C"
