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

template <bool NN>
__global__ __launch_bounds__(256, 2) void sk_gemm_k(const SkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sk_lds[];
  float* const As0 = sk_lds;
  float* const As1 = sk_lds + SK_A_FLOATS;
  float* const Bs0 = sk_lds + 2 * SK_A_FLOATS;
  float* const Bs1 = Bs0 + SK_B_FLOATS;
  unsigned* const s_last = reinterpret_cast<unsigned*>(Bs1 + SK_B_FLOATS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const long long G = gridDim.x, me = blockIdx.x;
  const int slabs = a.slabs, tiles = a.tiles_m * a.tiles_n;
  long long u0 = me * a.units / G;
  const long long u_end = (me + 1) * a.units / G;
  const long long first_tile = u0 / slabs;
  // staging coordinates
  const int ar = tid >> 3, ac = 4 * (tid & 7);         // A (and NT B): rows ar + 32 u, float4 column ac of the slab
  const int br = tid >> 5, bcn = 4 * (tid & 31);       // NN B: reduction rows br + 8 u, float4 column bcn of the tile

  while (u0 < u_end) {
    const long long tile_g = u0 / slabs;
    const int s0 = (int)(u0 - tile_g * slabs);
    const int s1 = (int)((u_end - u0 < (long long)(slabs - s0)) ? s0 + (u_end - u0) : slabs);
    const int pi = (int)(tile_g / tiles);
    const int t = (int)(tile_g - (long long)pi * tiles);
    const int tm = t / a.tiles_n, tn = t - tm * a.tiles_n;
    const int m0 = tm * SK_BM, n0 = tn * SK_BN;
    const SkProblem& P = a.p[pi];
    const int M = a.M, N = a.N;

    // per-thread row bases (clamped into range: products of rows / columns beyond the matrix are never stored)
    size_t arow[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) arow[u] = (size_t)min(m0 + ar + 32 * u, M - 1);
    size_t brow[4];                                      // NT: W row; NN: unused
    if constexpr (!NN) {
#pragma unroll
      for (int u = 0; u < 4; ++u) brow[u] = (size_t)min(n0 + ar + 32 * u, N - 1);
    }
    const int bcol = min(n0 + bcn, N - 4);               // NN

    // operands of both sources as scalars (selected per slab: no kernel-argument loads inside the slab loop)
    const float* const A_0 = P.A[0];
    const float* const B_0 = P.B[0];
    const float* const A_1 = P.A[1];
    const float* const B_1 = P.B[1];
    const int R_0 = P.R[0], R_1 = P.R[1], slabs0 = a.slabs0;
    struct Slot { sk_f4 a[4], b[4]; };
    auto fetch = [&](Slot& r, int s) __attribute__((always_inline)) {
      const bool second = s >= slabs0;
      const int k0 = (second ? s - slabs0 : s) * SK_BK;
      const float* __restrict__ Ap = second ? A_1 : A_0;
      const float* __restrict__ Bp = second ? B_1 : B_0;
      const int R = second ? R_1 : R_0;
      const int ka = (k0 + ac < R) ? k0 + ac : 0;         // R % 4 == 0: a float4 is entirely in or out (zeroed when stashed)
#pragma unroll
      for (int u = 0; u < 4; ++u) r.a[u] = *reinterpret_cast<const sk_f4*>(Ap + arow[u] * (size_t)R + ka);
      if constexpr (!NN) {
#pragma unroll
        for (int u = 0; u < 4; ++u) r.b[u] = *reinterpret_cast<const sk_f4*>(Bp + brow[u] * (size_t)R + ka);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rr = min(k0 + br + 8 * u, R - 1);
          r.b[u] = *reinterpret_cast<const sk_f4*>(Bp + (size_t)rr * N + bcol);
        }
      }
    };
    auto stash = [&](const Slot& r, int s, float* As, float* Bs) __attribute__((always_inline)) {
      const bool second = s >= slabs0;
      const int k0 = (second ? s - slabs0 : s) * SK_BK;
      const int R = second ? R_1 : R_0;
      const bool kok = k0 + ac < R;
      const sk_f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<sk_f4*>(&As[(ar + 32 * u) * SK_LDA + ac]) = kok ? r.a[u] : zero;
      if constexpr (!NN) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<sk_f4*>(&Bs[(ar + 32 * u) * SK_LDA + ac]) = kok ? r.b[u] : zero;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool rok = k0 + br + 8 * u < R;
          *reinterpret_cast<sk_f4*>(&Bs[(br + 8 * u) * SK_LDB + bcn]) = rok ? r.b[u] : zero;
        }
      }
    };
    sk_f4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) acc[x][y] = sk_f4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const float* As, const float* Bs) __attribute__((always_inline)) {
      const float* __restrict__ a_s = As + (64 * wm + i) * SK_LDA + 4 * q;
#pragma unroll
      for (int ks = 0; ks < SK_BK / 16; ++ks) {
        sk_f4 av[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) av[mb] = *reinterpret_cast<const sk_f4*>(a_s + 16 * mb * SK_LDA + 16 * ks);
        if constexpr (!NN) {
          const float* __restrict__ b_s = Bs + (64 * wn + i) * SK_LDA + 4 * q;
          sk_f4 bv[4];
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) bv[nb] = *reinterpret_cast<const sk_f4*>(b_s + 16 * nb * SK_LDA + 16 * ks);
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
              for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = SK_MFMA(bv[nb][c], av[mb][c], acc[nb][mb]);       // D[n][m]
        } else {
          const float* __restrict__ b_s = Bs + (16 * ks + 4 * q) * SK_LDB + 64 * wn + 4 * i;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const sk_f4 bc = *reinterpret_cast<const sk_f4*>(b_s + c * SK_LDB);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
              for (int s = 0; s < 4; ++s) acc[mb][s] = SK_MFMA(av[mb][c], bc[s], acc[mb][s]);               // D[m][4 j + s]
          }
        }
      }
    };

    // ---- the segment: slabs [s0, s1) of this tile.  One register set: the loads of slab s + 1 are in flight while slab s
    // is multiplied out of LDS (128 MFMAs per wave, ~2 us: longer than a round trip to L2 / the memory-side cache), then
    // stored into the other LDS buffer; one barrier per slab.  (Two register sets -- two slabs in flight, the loop unrolled
    // by two -- made the compiler rotate the 64 accumulators through v_accvgpr moves at every trip.)
    {
      Slot r;
      fetch(r, s0);
      stash(r, s0, As0, Bs0);
      __syncthreads();
      for (int s = s0; s < s1; ++s) {
        const bool odd = (s - s0) & 1;
        float* const Ac = odd ? As1 : As0;
        float* const Bc = odd ? Bs1 : Bs0;
        // (unconditional: a stash under `if (s + 1 < s1)` lets the compiler SINK the loads into that branch, behind the
        // MFMAs, where their whole round trip is exposed; the last trip re-reads slab s1 - 1 into the idle buffer)
        const int sn = min(s + 1, s1 - 1);
        fetch(r, sn);
        __builtin_amdgcn_sched_barrier(0);               // requests first, then the products
        compute(Ac, Bc);
        __builtin_amdgcn_sched_barrier(0);
        stash(r, sn, odd ? As0 : As1, odd ? Bs0 : Bs1);
        __syncthreads();
      }
    }

    // ---- pieces: 16 float4 per thread, piece p at (row, col .. col + 3)
    sk_f4 piece[16];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        if constexpr (!NN) piece[4 * x + y] = acc[x][y];                                     // x = nb, y = mb
        else piece[4 * x + y] = sk_f4{acc[x][0][y], acc[x][1][y], acc[x][2][y], acc[x][3][y]};      // x = mb, y = r
      }
    const bool whole = s0 == 0 && s1 == slabs;
    bool finish = whole;
    if (!whole) {
      // contributors of this tile: the blocks whose ranges meet [tile_g * slabs, (tile_g + 1) * slabs)
      const long long ua = tile_g * slabs, ub = ua + slabs - 1;
      const long long b_first = ((ua + 1) * G - 1) / a.units, b_last = ((ub + 1) * G - 1) / a.units;
      float* mine = a.part + (size_t)(2 * me + (tile_g != first_tile ? 1 : 0)) * SK_PART_FLOATS;
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        float* dst = mine + (size_t)(p * 256 + tid) * 4;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(piece[p]) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) *s_last = atomicAdd(a.ticket + tile_g, 1u) == (unsigned)(b_last - b_first) ? 1u : 0u;
      __syncthreads();
      finish = *s_last != 0u;
      if (finish) {
        if (tid == 0) __hip_atomic_store(a.ticket + tile_g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the parts in RANGE order, this block's own included (re-read from its slot: keeping it in registers beside the
        // running sum costs the kernel its second resident block): the sum does not depend on who arrived last
        for (long long c = b_first; c <= b_last; ++c) {
          const long long c_first = (c * a.units / G) / slabs;
          const float* src = a.part + (size_t)(2 * c + (tile_g != c_first ? 1 : 0)) * SK_PART_FLOATS;
#pragma unroll
          for (int p = 0; p < 16; ++p) {
            const unsigned long long* w = reinterpret_cast<const unsigned long long*>(src + (size_t)(p * 256 + tid) * 4);
            const unsigned long long lo = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long hi = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const sk_f4 term = sk_f4{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                                     __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32))};
            piece[p] = (c == b_first) ? term : piece[p] + term;
          }
        }
      }
      __syncthreads();                                   // (s_last is rewritten by the next partial tile)
    }
    if (finish) {
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        int m, n;
        if constexpr (!NN) { m = m0 + 64 * wm + 16 * (p & 3) + i; n = n0 + 64 * wn + 16 * (p >> 2) + 4 * q; }
        else { m = m0 + 64 * wm + 16 * (p >> 2) + 4 * q + (p & 3); n = n0 + 64 * wn + 4 * i; }
        if (m >= M || n >= N) continue;                  // N % 4 == 0: the float4 is entirely in or out
        sk_f4 v = piece[p];
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
    }
    u0 += s1 - s0;
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
  int grid = sk_cu_count() * (blocks_per_cu > 0 ? blocks_per_cu : 1);
  if (grid > SK_MAX_GRID) grid = SK_MAX_GRID;
  if ((long long)grid > a.units) grid = (int)a.units;
  if (grid < 1) return -1;
  if (!ws || tiles * sizeof(unsigned) > ticket_bytes || ticket_bytes + sk_workspace_part_bytes(grid) > ws_bytes) return -1;
  a.ticket = reinterpret_cast<unsigned*>(ws);
  a.part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ticket_bytes);
  static bool attr_done[2] = {false, false};
  if (!attr_done[nn ? 1 : 0]) {
    if (nn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sk_gemm_k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES);
    else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sk_gemm_k<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES);
    attr_done[nn ? 1 : 0] = true;
  }
  if (nn) hipLaunchKernelGGL(sk_gemm_k<true>, dim3(grid), dim3(256), SK_LDS_BYTES, st, a);
  else hipLaunchKernelGGL(sk_gemm_k<false>, dim3(grid), dim3(256), SK_LDS_BYTES, st, a);
  return 0;
}

}  // namespace cgv
