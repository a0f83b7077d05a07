// LDS-staged fp32 GEMMs with a stream-K work split, for the Dense / nn.Linear layers with MANY rows (atom level of the
// dipeptide batch: 704 rows; the 2000-atom graph) -- modules.py:103-114 forward and the input gradient of its autograd:
//
//   NT  (forward)         out[M, N] = act(A[M, R] B[N, R]^T + bias)          A = x, B = W, R = in_features
//   NN  (backward-input)  out[M, N] = (A[M, R] B[R, N] (+ A1 B1) + add + bcast) * act'(oz)   A = g, B = W, R = out_features
//
// Why another kernel.  The register-tile kernels of tile_gemm.hip (32 x 32 / 16 x 64 tiles fed from L2, the reduction split
// over the waves of a block) are built for FEW rows; from ~700 rows on they level off at 60-95 TF/s because every wave
// pulls its own operand fragments through the L1 (256 bytes per MFMA).  The products of these layers have small OUTPUTS
// and long reductions (2000 x 600 out of an 1800- or 5400-deep sum): with 128 x 128 output tiles there are 80 tiles for
// 256 CUs, with 64 x 64 tiles the operands are re-read twice as often and a CU's last tile still decides the launch.
// Here the unit of work is (output tile, 32-deep slab of the reduction); the units of a launch -- of BOTH problems of a
// pair launch, of BOTH sources of a two-source product -- form one sequence that is cut into EQUAL contiguous ranges, one
// per block, one or two blocks per CU (stream-K).  A block walks its range tile by tile; a tile it holds only a part of is
// left as a partial tile in the workspace (write-through stores, then a ticket: the hand-over of tile_bwd_input_k's
// SplitN), and the block that draws the tile's last ticket adds the parts IN RANGE ORDER (not arrival order: repeated
// launches are bit-identical) and runs the store epilogue once.
//
// Inside a block: 256 threads = 2 x 2 waves, each wave a 64 x 64 quarter of the tile as 4 x 4 v_mfma_f32_16x16x4_f32
// accumulators (exact fp32 products, fp32 accumulation: the same arithmetic as the register-tile kernels).  A slab of A
// ([128][32], rows padded to 36 floats) and of B (NT: [128][32] likewise; NN: [32][128], rows padded to 132) is staged
// through registers: two register sets, so two slabs of global loads are in flight while a third is multiplied from LDS;
// two LDS buffers, one barrier per slab (128 MFMAs per wave between barriers).  Fragments: 8 ds_read_b128 per 64 MFMAs.
//   NT: the product is formed transposed (B fragments as the MFMA's A operand) so that a lane ends with 4 consecutive n of
//       one row m: bias, activation and stores are 16 bytes wide.
//   NN: B fragments are float4s ALONG the output (W[r][n .. n + 3], r = the MFMA's reduction index); component s feeds
//       accumulator s, which then holds columns {4 j + s}: a lane's four accumulators give 4 consecutive columns of a row.
#include <type_traits>
#include "cgv_common.h"
#include "streamk_gemm.h"

namespace cgv {

typedef float sk_f4 __attribute__((ext_vector_type(4)));
#define SK_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

typedef unsigned sk_u4 __attribute__((ext_vector_type(4)));
constexpr int SK_SC1 = 16;                          // buffer cache policy: agent scope (sc1) -- past the XCD's own L2
constexpr int SK_SPIN_LIMIT = 1 << 22;              // ~2 s of polling: a lost contributor ends the wait instead of hanging the GPU

// The store epilogue of one float4 of the output at (m, n .. n + 3).
template <bool NN>
__device__ __forceinline__ void sk_store(const SkProblem& P, sk_f4 v, int m, int n, int M, int N) {
  if (m >= M || n >= N) return;                      // N % 4 == 0: the float4 is entirely in or out
  const size_t at = (size_t)m * N + n;
  if constexpr (!NN) {
    if (P.bias) v += *reinterpret_cast<const sk_f4*>(P.bias + n);
    if (P.act) {
      if (P.zout) *reinterpret_cast<sk_f4*>(P.zout + at) = v;
      v = sk_f4{act_fwd(v[0], P.act), act_fwd(v[1], P.act), act_fwd(v[2], P.act), act_fwd(v[3], P.act)};
    }
  } else {
    if (P.add) v += *reinterpret_cast<const sk_f4*>(P.add + at);
    if (P.bc_src) {
      const int sg = (int)P.bc_row2seg[m];
      const sk_f4 b4 = *reinterpret_cast<const sk_f4*>(P.bc_src + (size_t)sg * N + n);
      const int len = P.bc_rowptr[sg + 1] - P.bc_rowptr[sg];
      const float sc = P.bc_mean ? 1.0f / (float)(len > 1 ? len : 1) : 1.0f;
      v = sk_f4{fmaf(b4[0], sc, v[0]), fmaf(b4[1], sc, v[1]), fmaf(b4[2], sc, v[2]), fmaf(b4[3], sc, v[3])};
    }
    if (P.oz) {
      const sk_f4 z4 = *reinterpret_cast<const sk_f4*>(P.oz + at);
      v = sk_f4{v[0] * act_bwd(z4[0], P.oact), v[1] * act_bwd(z4[1], P.oact), v[2] * act_bwd(z4[2], P.oact),
                v[3] * act_bwd(z4[3], P.oact)};
    }
  }
  *reinterpret_cast<sk_f4*>(P.out + at) = v;
}

// Row / column of piece p (0..7) of a thread: see the kernel's accumulator layouts (wave tile 32 x 64).
template <bool NN>
__device__ __forceinline__ void sk_piece_at(int p, int m0, int n0, int wm, int wn, int i, int q, int& m, int& n) {
  if constexpr (!NN) { m = m0 + 32 * wm + 16 * (p & 1) + i; n = n0 + 64 * wn + 16 * (p >> 1) + 4 * q; }
  else { m = m0 + 32 * wm + 16 * (p >> 2) + 4 * q + (p & 3); n = n0 + 64 * wn + 4 * i; }
}

template <bool NN>
__global__ __launch_bounds__(SK_THREADS) void sk_gemm_k(const SkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sk_lds[];
  float* const As0 = sk_lds;
  float* const As1 = sk_lds + SK_A_FLOATS;
  float* const Bs0 = sk_lds + 2 * SK_A_FLOATS;
  float* const Bs1 = Bs0 + SK_B_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;             // 4 x 2 waves, each 32 rows x 64 columns of the tile
  // units x grid < 2^31 (sk_launch): 32-bit arithmetic throughout
  // Range id of this block: blocks are dealt to the 8 XCDs round robin (block b runs on XCD b % 8), ranges are numbered so
  // that each XCD holds a CONTIGUOUS eighth of the unit sequence -- neighbouring tiles (same rows of A: the column tiles of
  // one row tile follow each other) and the parts of one tile meet in one L2 instead of being fetched into all eight.
  const unsigned G = gridDim.x, units = (unsigned)a.units;
#if defined(SK_DBG_NOXCD)
  const unsigned me = blockIdx.x;
#else
  const unsigned per = G >> 3, big = G & 7u, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned me = xcd * per + (xcd < big ? xcd : big) + idx;       // (XCDs 0 .. big - 1 hold one range more)
#endif
  const int slabs = a.slabs, slabs0 = a.slabs0, tiles = a.tiles_m * a.tiles_n;
  const unsigned u0 = me * units / G, u_end = (me + 1) * units / G;
  if (u0 >= u_end) return;
  const int first_tile = (int)(u0 / (unsigned)slabs);
  const int M = a.M, N = a.N;
  // staging coordinates
  const int ar = tid >> 3, ac = 4 * (tid & 7);         // A (and NT B): rows ar + 64 u, float4 column ac of the slab
  const int br = tid >> 5, bcn = 4 * (tid & 31);       // NN B: reduction rows br + 16 u, float4 column bcn of the tile
  const __amdgpu_buffer_rsrc_t r_part = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, 0x7fffffff, 0x00020000);
  int pend0 = -1, pend1 = -1;                          // tiles this block holds a PART of (its first and / or its last)

  // ---- the FETCH side: tile of the slab being requested.  Operand rows as 32-bit byte offsets into buffer descriptors of
  // the matrices (exact sizes: a row of B beyond the reduction's end, NN, reads as zero), the slab's position as the
  // scalar offset of the load: no vector address arithmetic per slab.
  int f_tile = first_tile, f_s = (int)(u0 - (unsigned)first_tile * (unsigned)slabs);
  const float* fA0; const float* fB0; const float* fA1; const float* fB1;
  int fR = 0;
  unsigned voffA[2], voffB[2];
  auto fetch_tile = [&](int tile_g) __attribute__((always_inline)) {
    const int pi = tile_g / tiles;
    const int t = tile_g - pi * tiles;
    const int tm = t / a.tiles_n, tn = t - tm * a.tiles_n;
    const int m0 = tm * SK_BM, n0 = tn * SK_BN;
    const SkProblem& P = a.p[pi];
    fA0 = P.A[0]; fB0 = P.B[0]; fA1 = P.A[1]; fB1 = P.B[1];
    fR = P.R[0];                                       // (both sources have the same depth: sk_launch)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      voffA[u] = ((unsigned)min(m0 + ar + 64 * u, M - 1) * (unsigned)fR + (unsigned)ac) * 4u;
      if constexpr (!NN) voffB[u] = ((unsigned)min(n0 + ar + 64 * u, N - 1) * (unsigned)fR + (unsigned)ac) * 4u;
      else voffB[u] = ((unsigned)(br + 16 * u) * (unsigned)N + (unsigned)min(n0 + bcn, N - 4)) * 4u;
    }
  };
  struct Slot { sk_f4 a[2], b[2]; };
  auto fetch = [&](Slot& r, int s) __attribute__((always_inline)) {
    const bool second = s >= slabs0;
    const int k0 = (second ? s - slabs0 : s) * SK_BK;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(second ? fA1 : fA0), 0, (unsigned)M * (unsigned)fR * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(second ? fB1 : fB0), 0, (unsigned)N * (unsigned)fR * 4u, 0x00020000);
#pragma unroll
    for (int u = 0; u < 2; ++u) r.a[u] = __builtin_bit_cast(sk_f4, __builtin_amdgcn_raw_buffer_load_b128(rA, voffA[u], (unsigned)k0 * 4u, 0));
    const unsigned soffB = NN ? (unsigned)k0 * (unsigned)N * 4u : (unsigned)k0 * 4u;
#pragma unroll
    for (int u = 0; u < 2; ++u) r.b[u] = __builtin_bit_cast(sk_f4, __builtin_amdgcn_raw_buffer_load_b128(rB, voffB[u], soffB, 0));
  };
  auto stash = [&](const Slot& r, int s, float* As, float* Bs) __attribute__((always_inline)) {
    const bool second = s >= slabs0;
    const int k0 = (second ? s - slabs0 : s) * SK_BK;
    const sk_f4 zero = {0.f, 0.f, 0.f, 0.f};
    // the last slab of a source may reach beyond the reduction's end: the float4s of A (and of B, NT) that do are zeroed
    // (fR % 4 == 0: a float4 is entirely in or out); B's rows beyond it (NN) lie behind the matrix and were read as zero
    const bool kok = k0 + SK_BK <= fR || k0 + ac < fR;
#pragma unroll
    for (int u = 0; u < 2; ++u) *reinterpret_cast<sk_f4*>(&As[(ar + 64 * u) * SK_LDA + ac]) = kok ? r.a[u] : zero;
    if constexpr (!NN) {
#pragma unroll
      for (int u = 0; u < 2; ++u) *reinterpret_cast<sk_f4*>(&Bs[(ar + 64 * u) * SK_LDA + ac]) = kok ? r.b[u] : zero;
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) *reinterpret_cast<sk_f4*>(&Bs[(br + 16 * u) * SK_LDB + bcn]) = r.b[u];
    }
  };
  sk_f4 acc[8];                                        // NT: [nb 4][mb 2] = D[n][m]; NN: [mb 2][s 4] = D[m][4 j + s]
#pragma unroll
  for (int x = 0; x < 8; ++x) acc[x] = sk_f4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const float* As, const float* Bs) __attribute__((always_inline)) {
    const float* __restrict__ a_s = As + (32 * wm + i) * SK_LDA + 4 * q;
    sk_f4 av[2][2], bv[2][4];
    if constexpr (!NN) {
      const float* __restrict__ b_s = Bs + (64 * wn + i) * SK_LDA + 4 * q;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) av[ks][mb] = *reinterpret_cast<const sk_f4*>(a_s + 16 * mb * SK_LDA + 16 * ks);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) bv[ks][nb] = *reinterpret_cast<const sk_f4*>(b_s + 16 * nb * SK_LDA + 16 * ks);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc[2 * nb + mb] = SK_MFMA(bv[ks][nb][c], av[ks][mb][c], acc[2 * nb + mb]);
    } else {
      const float* __restrict__ b_s = Bs + (4 * q) * SK_LDB + 64 * wn + 4 * i;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) av[ks][mb] = *reinterpret_cast<const sk_f4*>(a_s + 16 * mb * SK_LDA + 16 * ks);
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[ks][c] = *reinterpret_cast<const sk_f4*>(b_s + (16 * ks + c) * SK_LDB);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[4 * mb + s] = SK_MFMA(av[ks][mb][c], bv[ks][c][s], acc[4 * mb + s]);
    }
  };

  // ---- ONE pipeline over the block's whole range: while slab u is multiplied out of LDS the loads of slab u + 1 -- of the
  // NEXT tile when u ends one -- are in flight, then stored into the other LDS buffer; one barrier per slab.  Two waves share
  // a SIMD: one's loads, LDS traffic and waits run under the other's MFMAs (a wave does not issue anything beside its own
  // MFMA: 4 waves of 64 x 64 quarters measured 2.4 us per slab against 1.7 us of MFMA time, two such blocks per CU 2.0).
  // A tile (or the block's part of one) that ends is stored / published between the products and the stash, with the next
  // slab's loads in flight: the pipeline does not drain at tile boundaries.
  int c_tile = f_tile, c_s = f_s, seg_s0 = f_s;        // the COMPUTE side: tile and slab of the slab in LDS
  int pub = -1;                                        // a published part whose ticket is still to be drawn
  fetch_tile(f_tile);
  Slot r;
  fetch(r, f_s);
  stash(r, f_s, As0, Bs0);
  __syncthreads();
  bool odd = false;
#pragma unroll 1
  for (unsigned u = u0; u < u_end; ++u) {
    const bool has_next = u + 1 < u_end;
    if (has_next) {
      if (++f_s == slabs) { f_s = 0; fetch_tile(++f_tile); }
    }
#if !defined(SK_DBG_NOFETCH)
    fetch(r, f_s);                                     // (the last trip re-reads its own slab into the idle buffer)
#endif
    __builtin_amdgcn_sched_barrier(0);                 // requests first, then the products
#if !defined(SK_DBG_NOCOMPUTE)
    compute(odd ? As1 : As0, odd ? Bs1 : Bs0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (c_s == slabs - 1 || !has_next) {
      // ---- this block's slabs [seg_s0, c_s] of tile c_tile are summed
      const int pi = c_tile / tiles;
      const int t = c_tile - pi * tiles;
      const int tm = t / a.tiles_n, tn = t - tm * a.tiles_n;
      const int m0 = tm * SK_BM, n0 = tn * SK_BN;
      const SkProblem& P = a.p[pi];
      sk_f4 piece[8];                                  // 8 float4 per thread, piece p at (row, col .. col + 3): sk_piece_at
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        if constexpr (!NN) piece[p] = acc[p];                                                  // p = 2 nb + mb
        else piece[p] = sk_f4{acc[4 * (p >> 2)][p & 3], acc[4 * (p >> 2) + 1][p & 3], acc[4 * (p >> 2) + 2][p & 3], acc[4 * (p >> 2) + 3][p & 3]};
      }
#pragma unroll
      for (int x = 0; x < 8; ++x) acc[x] = sk_f4{0.f, 0.f, 0.f, 0.f};
      if (seg_s0 == 0 && c_s == slabs - 1) {           // the whole tile: straight to the store epilogue
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          int m, n;
          sk_piece_at<NN>(p, m0, n0, wm, wn, i, q, m, n);
          sk_store<NN>(P, piece[p], m, n, M, N);
        }
      } else {
        // a part of the tile: into this block's slot (agent-scope write-through stores; the tile's ticket is drawn once they
        // are acknowledged -- below, behind the next barrier: the hand-over of tile_bwd_input_k / loss_tail.hip); finished
        // when this block's range is done
        const unsigned slot = 2u * me + (c_tile != first_tile ? 1u : 0u);
#pragma unroll
        for (int p = 0; p < 8; ++p)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sk_u4, piece[p]), r_part,
                                                 slot * (unsigned)(SK_PART_FLOATS * 4) + (unsigned)(p * SK_THREADS + tid) * 16u, 0, SK_SC1);
        pub = c_tile;
        if (c_tile == first_tile) pend0 = c_tile; else pend1 = c_tile;
      }
      seg_s0 = 0;
    }
#if !defined(SK_DBG_NOSTASH)
    stash(r, f_s, odd ? As0 : As1, odd ? Bs0 : Bs1);
#endif
    if (pub >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (wave-uniform, rare: a part was just published)
    __syncthreads();
    if (pub >= 0) {
      if (tid == 0) __hip_atomic_fetch_add(a.ticket + pub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pub = -1;
    }
    odd = !odd;
    if (has_next) { c_tile = f_tile; c_s = f_s; }
  }

  // ---- the tiles this block holds a part of.  EVERY contributor of such a tile finishes a share of it: once all k parts
  // are published (low half of the ticket), contributor r (in range order) takes the pieces r, r + k, ... and adds their k
  // parts IN RANGE ORDER -- the sum of an element does not depend on who arrived when, repeated launches are bit-identical --
  // and runs the store epilogue on them.  (One block finishing the whole tile would read k x 64 KB through one CU at the very
  // end of the launch.)  The last contributor to have read its share resets the ticket (its high half counts them).  The
  // wait is short: every part is published before any block waits, the blocks of a launch are all resident (grid <= CUs),
  // and a tile's parts are computed at the start (later ranges) or the end (earlier ranges) of equally long ranges.
#pragma unroll 1
  for (int e = 0; e < 2; ++e) {
    const int tile_g = e == 0 ? pend0 : pend1;
    if (tile_g < 0) continue;
    const unsigned ua = (unsigned)tile_g * (unsigned)slabs, ub = ua + (unsigned)slabs - 1u;
    const int b_first = (int)(((ua + 1u) * G - 1u) / units), b_last = (int)(((ub + 1u) * G - 1u) / units);
    const int k = b_last - b_first + 1, rank = (int)me - b_first;
    const int pi = tile_g / tiles;
    const int t = tile_g - pi * tiles;
    const int tm = t / a.tiles_n, tn = t - tm * a.tiles_n;
    const int m0 = tm * SK_BM, n0 = tn * SK_BN;
    const SkProblem& P = a.p[pi];
    if (tid == 0) {
      int spins = 0;
      while ((int)(__hip_atomic_load(a.ticket + tile_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu) < k &&
             ++spins < SK_SPIN_LIMIT)
        __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    if (rank < 8) {
      sk_f4 sum[4];
#pragma unroll 1
      for (int c0 = b_first; c0 <= b_last; c0 += 4) {
        sk_f4 term[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned c = (unsigned)min(c0 + u, b_last);
          const int c_first = (int)((c * units / G) / (unsigned)slabs);
          const unsigned base = (2u * c + (tile_g != c_first ? 1u : 0u)) * (unsigned)(SK_PART_FLOATS * 4) + (unsigned)tid * 16u;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int p = min(rank + j * k, 7);
            term[u][j] = __builtin_bit_cast(sk_f4, __builtin_amdgcn_raw_buffer_load_b128(r_part, base + (unsigned)p * (unsigned)(SK_THREADS * 16), 0, SK_SC1));
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (c0 + u > b_last) break;
#pragma unroll
          for (int j = 0; j < 4; ++j) sum[j] = (c0 + u == b_first) ? term[u][j] : sum[j] + term[u][j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int p = rank + j * k;
        if (p >= 8) break;
        int m, n;
        sk_piece_at<NN>(p, m0, n0, wm, wn, i, q, m, n);
        sk_store<NN>(P, sum[j], m, n, M, N);
      }
    }
    __syncthreads();                                   // every thread of the block has its share in registers
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(a.ticket + tile_g, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)(old >> 16) == k - 1) __hip_atomic_store(a.ticket + tile_g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- host
static int sk_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

size_t sk_workspace_part_bytes(int grid) { return (size_t)2 * grid * SK_PART_FLOATS * sizeof(float); }

/* Fills the geometry of ``a`` (p[], np, M, N set by the caller) and launches; ws = [tickets: ticket_bytes][parts].
 * Returns 0, or -1 with nothing launched when the workspace is too small / the shape is not for this kernel. */
int sk_launch(SkArgs& a, bool nn, void* ws, size_t ws_bytes, size_t ticket_bytes, hipStream_t st, int blocks_per_cu) {
  a.tiles_m = (a.M + SK_BM - 1) / SK_BM;
  a.tiles_n = (a.N + SK_BN - 1) / SK_BN;
  a.slabs0 = (a.p[0].R[0] + SK_BK - 1) / SK_BK;
  a.slabs = a.slabs0 + (a.p[0].A[1] ? (a.p[0].R[1] + SK_BK - 1) / SK_BK : 0);
  const long long tiles = (long long)a.np * a.tiles_m * a.tiles_n;
  a.units = tiles * a.slabs;
  int grid = blocks_per_cu < 0 ? -blocks_per_cu : sk_cu_count() * (blocks_per_cu > 0 ? blocks_per_cu : 1);
  if (grid > SK_MAX_GRID) grid = SK_MAX_GRID;
  if (blocks_per_cu >= 0 && tiles <= grid) {
    // fewer tiles than blocks: every tile is shared by the SAME number of blocks (ranges then begin and end at tile boundaries:
    // one part per block instead of two, a few CUs idle -- 80 tiles x 3 on 256 CUs: 55.5 against 59.6 us)
    const int share = grid / (int)tiles;
    grid = (int)tiles * (share < a.slabs ? share : a.slabs);
  }
  if ((long long)grid > a.units) grid = (int)a.units;
  if (grid < 1 || a.units * grid >= (1ll << 31) || tiles >= (1 << 15)) return -1;     // (32-bit unit arithmetic, 16-bit ticket halves)
  // 32-bit byte offsets into the operands; both sources of a two-source product have one depth
  const long long R = a.p[0].R[0];
  if ((long long)a.M * R >= (1ll << 30) || (long long)a.N * R >= (1ll << 30)) return -1;
  for (int pi = 0; pi < a.np; ++pi) {
    if (a.p[pi].R[0] != R || (a.p[pi].A[1] && a.p[pi].R[1] != R) || (!!a.p[pi].A[1] != !!a.p[0].A[1])) return -1;
  }
  if (!ws || tiles * sizeof(unsigned) > ticket_bytes || ticket_bytes + sk_workspace_part_bytes(grid) > ws_bytes) return -1;
  a.ticket = reinterpret_cast<unsigned*>(ws);
  a.part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ticket_bytes);
  static bool attr_done[2] = {false, false};
  if (!attr_done[nn ? 1 : 0]) {
    if (nn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sk_gemm_k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES);
    else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sk_gemm_k<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES);
    attr_done[nn ? 1 : 0] = true;
  }
  if (nn) hipLaunchKernelGGL(sk_gemm_k<true>, dim3(grid), dim3(SK_THREADS), SK_LDS_BYTES, st, a);
  else hipLaunchKernelGGL(sk_gemm_k<false>, dim3(grid), dim3(SK_THREADS), SK_LDS_BYTES, st, a);
  return 0;
}

}  // namespace cgv
