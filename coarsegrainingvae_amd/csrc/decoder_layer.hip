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

// phase clock of block 0 (measurement only: cgv_decoder_debug_clock); NULL in normal operation
// (__constant__: read by ONE scalar load per kernel; as a __device__ variable every tick was a vector load followed by
// s_waitcnt vmcnt(0) -- which also drained every prefetch in flight at that point)
__constant__ unsigned long long* g_dl_clock = nullptr;
// Compiled OUT by default: the ticks' scalar branches alone cost the decoder backward 12 us per chignolin step (same-box
// A/B of the two builds, tools/ab_lib.sh: 395.7 / 395.4 against 382.5 / 384.0 us).  The phase clock is measured on the
// variant build `tools/build_variant.sh decoder_layer -DCGV_DL_CLOCK=1` (tools/phase_clock.sh).
#ifndef CGV_DL_CLOCK
#define CGV_DL_CLOCK 0
#endif
// launch spans for a layer's timeline: slots 32 + 4 id + {0, 1}: first block's begin / end, {2, 3}: last block's
// (ids: 0 F1 dense, 1 F2 message, 2 F3 uv, 3 F4 dense, 4 F5 gate, 5 B1 gate, 6 B2 dense, 7 B3 uv, 8 B4 message, 9 B5 dense)
#if CGV_DL_CLOCK
#define DL_SPAN(id, end)                                                                                    \
  do {                                                                                                      \
    if (g_dl_clock && threadIdx.x == 0) {                                                                   \
      if (blockIdx.x == 0) g_dl_clock[32 + 4 * (id) + (end)] = wall_clock64();                              \
      if (blockIdx.x == gridDim.x - 1) g_dl_clock[32 + 4 * (id) + 2 + (end)] = wall_clock64();              \
    }                                                                                                       \
  } while (0)

// per-phase ticks of block 0 of EVERY decoder launch: slots 80 + 10 id + i (ids as above, i < 10); buf: 192 uint64
#define DL_PH(id, i) do { if (g_dl_clock && blockIdx.x == 0 && threadIdx.x == 0) g_dl_clock[80 + 10 * (id) + (i)] = wall_clock64(); } while (0)
#else
#define DL_SPAN(id, end) do { } while (0)
#define DL_PH(id, i) do { } while (0)
#endif

// per-WAVE ticks of block 0 of the message backward (B4): slots 192 + 8 wave + i; buf: 272 uint64
#if CGV_DL_CLOCK
#define DL_WV(i) do { if (g_dl_clock && blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_dl_clock[192 + 8 * (threadIdx.x >> 6) + (i)] = wall_clock64(); } while (0)
#else
#define DL_WV(i) do { } while (0)
#endif
#ifndef CGV_DL_FILTER_PERM
#define CGV_DL_FILTER_PERM 1
#endif
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
// Cross-lane sums without the LDS crossbar (__shfl_xor compiles to ds_bpermute_b32: an LDS round trip per step of a
// butterfly).  x + x[lane ^ 32], x + x[lane ^ 16]: gfx950's v_permlane32_swap / v_permlane16_swap on two copies of x --
// both halves of a pair compute own + partner or partner + own, the same bits as the shuffle form.  Sum over the four
// lanes {l, l + 4, l + 8, l + 12} of a 16-lane row: two DPP row rotations, (x_l + x_{l+8}) + (x_{l+4} + x_{l+12}).
__device__ __forceinline__ float add_xor32(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float add_xor16(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float add_row_stride4(float x) {
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x128 /*row_ror:8*/, 0xf, 0xf, false));
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x124 /*row_ror:4*/, 0xf, 0xf, false));
  return x;
}
template <int R>
__device__ __forceinline__ float dfilt(const float (&W)[R + 1], const float* __restrict__ g) {
  float w = W[R] * g[R];
#pragma unroll
  for (int n = 0; n < R; ++n) w = fmaf(W[n], g[n], w);
  return w;
}

// ---------------------------------------------------------------------------------------------- LDS carving
// All kernels take their LDS from one dynamic allocation (one block per CU: up to 160 KB are free to use).
extern __shared__ __attribute__((aligned(16))) unsigned char dl_smem[];
struct Carve {
  size_t off = 0;
  __device__ __forceinline__ float* take(size_t floats) {
    float* p = reinterpret_cast<float*>(dl_smem + off);
    off += ((floats * 4 + 15) / 16) * 16;
    return p;
  }
};

// ---------------------------------------------------------------------------------------------- pinned requests
// A prefetch must stay where it is written.  The operands are __restrict__ const data, so the compiler is free to sink
// an ordinary load down to its first use -- into the branch that stores it to LDS (one vmcnt(0) round trip per staged
// array) or in front of the product (a full HBM round trip there); it does both.  (volatile loads are no way out: the
// backend waits vmcnt(0) after each one.)  Instead: the kernel reads such arrays through LAUNDERED pointers (no longer
// provably distinct from what an opaque asm may write) and closes each batch of requests with pin_loads(), an empty asm
// that clobbers memory: a load cannot move below something that may overwrite its source.
// (The laundered pointer keeps its address space in its type: a generic pointer of unknown origin would compile to
// flat_load.)
#define CGV_GLOBAL_AS __attribute__((address_space(1)))
typedef const CGV_GLOBAL_AS float* gcf;
typedef const CGV_GLOBAL_AS int* gci;
__device__ __forceinline__ gcf launder(const float* p) {                                    // kernel arguments: uniform
  unsigned long long a = reinterpret_cast<unsigned long long>(p);
  asm volatile("" : "+s"(a));
  return reinterpret_cast<gcf>(a);
}
__device__ __forceinline__ gci launder(const int* p) {
  unsigned long long a = reinterpret_cast<unsigned long long>(p);
  asm volatile("" : "+s"(a));
  return reinterpret_cast<gci>(a);
}
__device__ __forceinline__ const float* generic(gcf p) { return (const float*)p; }
__device__ __forceinline__ void pin_loads() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ float4 ldg4_pinned(gcf p) {
  const f32x4 t = *reinterpret_cast<const CGV_GLOBAL_AS f32x4*>(p);
  return make_float4(t.x, t.y, t.z, t.w);
}
// streamed weights in whole 256-byte row pieces (the backward products).  Nontemporal loads pull 13 MB in 3.57 against
// 4.15 us in an isolated dependent chain (tools/probes/stream_probe.hip) but the decoder backward got no faster with them
// (468.9 against 460.6 us) -- off.  (Never for the forward products' 64-byte pieces: 7.98 against 6.22 us.)
#ifndef CGV_DL_NT_WEIGHTS
#define CGV_DL_NT_WEIGHTS 0
#endif
__device__ __forceinline__ float4 ldg4_stream(gcf p) {
#if CGV_DL_NT_WEIGHTS
  const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const CGV_GLOBAL_AS f32x4*>(p));
#else
  const f32x4 t = *reinterpret_cast<const CGV_GLOBAL_AS f32x4*>(p);
#endif
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ float ldg_pinned(gcf p) { return *p; }
__device__ __forceinline__ int ldgi_pinned(gci p) { return *p; }
__device__ __forceinline__ f3 ld3_pinned(gcf p) { return f3{p[0], p[1], p[2]}; }

// ---------------------------------------------------------------------------------------------- slice sums (quad-major)
// Slice layout: [K/4 column quads][rows][4] floats, rows = n (16-row phases) or 3 n (the [u_mat; v_mat] phase).
// tile[q][m][c] = sum_s slices[s][kq[q]][m][c] for NQ quads at once (one batch of loads); all threads take part; valid
// after the trailing __syncthreads().  scratch: NQ * 36 * 16 * MB float4 (the first NQ * 9 * 16 MB are used).  Order:
// per lane class ascending s; the wave's four classes as (c0 + c1) + (c2 + c3); then the 9 waves in order.
// QS = slices per lane class: 36 QS >= the number of slices (6: 216 >= F / 4 for F <= 864; 3: the 75 slices of the
// 8-channel phases at F = 600 -- half the registers and no clamped duplicate loads).
constexpr int DL_QS = 6;
template <int MB, int NQ, int QS = DL_QS>
struct QuadRegs { float4 v[QS][NQ][MB]; };

// Issue every load of the slice sum (straight-line, one batch: the compiler can count them, so that the weight prefetch
// issued afterwards stays in flight while these are consumed -- vmcnt returns in order).
template <int MB, int NQ, int QS>
__device__ __forceinline__ void quad_issue(QuadRegs<MB, NQ, QS>& r, gcf slices, int n_slices, long long stride,
                                           int rows, const int (&kq)[NQ]) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, cls = wave * 4 + (lane >> 4);                 // 36 slice classes
#pragma unroll
  for (int u = 0; u < QS; ++u) {
    const int s = min(cls + 36 * u, n_slices - 1);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        r.v[u][q][mb] = ldg4_pinned(slices + (size_t)s * stride + ((size_t)kq[q] * rows + min(mb * 16 + m, rows - 1)) * 4);
  }
}

// Sum: per class ascending slice index, then the 36 classes in order.  tile[q][m] valid after the trailing barrier.
template <int MB, int NQ, int QS>
__device__ __forceinline__ void quad_finish(const QuadRegs<MB, NQ, QS>& r, float4* __restrict__ tile /*[NQ][16 MB]*/,
                                            float4* __restrict__ scratch /*NQ * 36 * 16 MB*/, int n_slices, int rows, int ph = -1) {
  constexpr int MP = 16 * MB;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, cls = wave * 4 + (lane >> 4);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < QS; ++u)
        if (cls + 36 * u < n_slices) { acc.x += r.v[u][q][mb].x; acc.y += r.v[u][q][mb].y; acc.z += r.v[u][q][mb].z; acc.w += r.v[u][q][mb].w; }
      acc.x = add_xor16(acc.x); acc.y = add_xor16(acc.y); acc.z = add_xor16(acc.z); acc.w = add_xor16(acc.w);
      acc.x = add_xor32(acc.x); acc.y = add_xor32(acc.y); acc.z = add_xor32(acc.z); acc.w = add_xor32(acc.w);
      if (lane < 16) scratch[((q * DL_WAVES + wave) * MB + mb) * 16 + m] = acc;
    }
  if (ph >= 0) DL_PH(ph, 2);                                              // this wave's slices landed and summed
  if (ph == 8) DL_WV(3);
  __syncthreads();
  if (ph >= 0) DL_PH(ph, 3);                                              // every wave's
  if (threadIdx.x < NQ * MP) {
    const int q = threadIdx.x / MP, mm = threadIdx.x - q * MP;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < DL_WAVES; ++c) {
      const float4 v = scratch[((q * DL_WAVES + c) * MB + mm / 16) * 16 + (mm & 15)];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    if (mm >= rows) t = make_float4(0.f, 0.f, 0.f, 0.f);
    tile[q * MP + mm] = t;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------- backward-input core
// The block's G row groups (4 consecutive weight rows each, first row row0[g]) times g_tile[m][g*4 + q] -> this block's
// slice, quad-major.  Wave w takes column tiles w, w + 9, ... of 64 columns.  The weight registers are loaded by
// bi_prefetch at the top of the kernel (independent of the prologue).  The tile leaves through an LDS stage so that a
// wave's stores are contiguous (16 quads x rows x 16 bytes per tile) instead of 16-byte pieces 64 bytes apart.
template <int G, int NT>
struct BiRegs { float4 w[NT][G]; };

// Column tiles of a backward-input product.  These products are bound by the fp32 MFMA pipe of the block's CU (16x16x4:
// 32 cycles each; a 64-column tile of G row groups over MB row blocks is 4 G MB of them), so (i) the grid's y dimension
// splits the tiles of one channel group over gridDim.y blocks = CUs (each part repeats the prologue and writes its own
// columns of the group's slice; dense outputs are written by part 0) and (ii) a block's tiles go to its 9 waves in an
// order that keeps SIMD 0 -- which hosts waves 0, 4 and 8 -- from taking the surplus tiles of a round as well.
struct TileRange { int beg, end; };
__device__ __forceinline__ TileRange dl_tile_range(int K) {
  const int tiles = (K + 63) / 64, per = (tiles + (int)gridDim.y - 1) / (int)gridDim.y;
  const int beg = (int)blockIdx.y * per;
  return TileRange{beg, min(beg + per, tiles)};
}
// tile of wave `wave` in round t: round 0 in wave order, later rounds in the order 1, 2, 3, 5, 6, 7, 0, 4, 8
__device__ __forceinline__ int dl_tile_of(const TileRange& tr, int wave, int t) {
  const int pos = t == 0 ? wave : (int)((0x854372106ull >> (4 * wave)) & 15);
  return tr.beg + DL_WAVES * t + pos;
}

template <int G, int NT>
__device__ __forceinline__ void bi_prefetch(BiRegs<G, NT>& r, gcf W, int K, const int (&row0)[G], const TileRange& tr) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tile = dl_tile_of(tr, wave, t);
    if (tile < tr.end) {                                                  // wave-uniform: a wave without a tile in this round requests nothing
      const int col = tile * 64 + 4 * j;
      const int cc = col < K ? col : 0;                                   // clamped: always-valid address, result unused
#pragma unroll
      for (int g = 0; g < G; ++g) r.w[t][g] = ldg4_stream(W + (size_t)(row0[g] + q) * K + cc);
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) r.w[t][g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// one column tile (wave-uniform, in range) into slot SLOT of the registers
template <int SLOT, int G, int NT>
__device__ __forceinline__ void bi_prefetch_slot(BiRegs<G, NT>& r, gcf W, int K, const int (&row0)[G], int tile) {
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, q = lane >> 4;
  const int col = tile * 64 + 4 * j;
  const int cc = col < K ? col : 0;
#pragma unroll
  for (int g = 0; g < G; ++g) r.w[SLOT][g] = ldg4_stream(W + (size_t)(row0[g] + q) * K + cc);
}

// A slice is read by other CUs in the next launch and never again by this one.  CGV_DL_SLICE_STORE 1: write-through
// (sc1) stores -- the kernel then ends without a slab of dirty L2 lines to write back before the next launch may start.
#ifndef CGV_DL_SLICE_STORE
#define CGV_DL_SLICE_STORE 1
#endif
__device__ __forceinline__ void st4_slice(float4* p, const float4& v) {
#if CGV_DL_SLICE_STORE == 1
  const f32x4 t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(t) : "memory");
#else
  *p = v;
#endif
}

template <int MB>
__host__ __device__ constexpr size_t bi_stage_floats() { return (size_t)DL_WAVES * 16 * 16 * MB * 4; }

// NPRE of the NT column tiles per wave come from bi_prefetch's registers; the others (register budget: the message
// kernel keeps 9 row groups per tile) are loaded where they are used -- only wave 0 has a second tile at K = 600.
template <int MB, int G, int NT, int NPRE = NT>
__device__ __forceinline__ void bi_core(const BiRegs<G, NPRE>& r, const float* __restrict__ g_tile /*[16 MB][G*4] LDS*/,
                                        float* __restrict__ stage /*bi_stage_floats<MB>() LDS*/,
                                        float* __restrict__ slice /*[K/4][rows][4]*/, int K, int rows, const TileRange& tr,
                                        const float* __restrict__ W = nullptr, const int* row0 = nullptr, int ph = -1) {
  constexpr int MP = 16 * MB;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  float4* st = reinterpret_cast<float4*>(stage) + (size_t)wave * 16 * MP;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tile = dl_tile_of(tr, wave, t);
    if (tile >= tr.end) continue;                                         // wave-uniform
    f32x4 acc[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[mb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 late[G];
    if (t >= NPRE) {
      const int col = tile * 64 + 4 * j;
      const int cc = col < K ? col : 0;
#pragma unroll
      for (int g = 0; g < G; ++g) late[g] = *reinterpret_cast<const float4*>(W + (size_t)(row0[g] + q) * K + cc);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4 w = t < NPRE ? r.w[t < NPRE ? t : 0][g] : late[g];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float a = g_tile[(mb * 16 + j) * (G * 4) + g * 4 + q];
        acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.x, acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.y, acc[mb][1], 0, 0, 0);
        acc[mb][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.z, acc[mb][2], 0, 0, 0);
        acc[mb][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.w, acc[mb][3], 0, 0, 0);
      }
    }
    // lane holds rows m = 16 mb + 4 q + rr of column quad j of this tile
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        st[j * MP + mb * 16 + 4 * q + rr] = make_float4(acc[mb][0][rr], acc[mb][1][rr], acc[mb][2][rr], acc[mb][3][rr]);
    if (ph >= 0 && t == 0) DL_PH(ph, 7);                                 // first tile: weights landed, MFMAs done, staged
    // (same wave wrote, same wave reads: wave-synchronous through LDS)
    const int quads = min(16, (K - tile * 64) / 4);                       // K % 4 == 0
    float4* out = reinterpret_cast<float4*>(slice) + (size_t)tile * 16 * rows;
    for (int idx = lane; idx < quads * rows; idx += 64) {
      const int jj = idx / rows, m = idx - jj * rows;
      st4_slice(out + idx, st[jj * MP + m]);
    }
  }
}

// ---------------------------------------------------------------------------------------------- forward core
// out[m][g][c] = sum_k x[m][k] W[row0[g] + c][k] for the block's G row groups, m < M (<= 16 MB).  16-row tiles hold four
// groups; the 9 waves split K in whole 16-float steps and meet in LDS (wave order).  red: 9 * T * MB * 256 floats,
// out: 16 MB * G * 4 floats.  Valid after the trailing __syncthreads().
template <int MB, int G>
__host__ __device__ constexpr size_t fwd_red_floats() { return (size_t)DL_WAVES * ((G + 3) / 4) * MB * 256; }

struct NoHook { __device__ __forceinline__ void operator()() const {} };
// SB = 16-float steps whose loads are issued together (a wave has ceil(ceil(K / 16) / 9) steps: 5 at K = 600, so SB >= 5
// makes the product ONE round trip); after_issue() runs behind the first batch's loads, before they are consumed.
// ph: phase-clock id of the calling kernel (ticks 1: requests issued, 2: MFMAs done + partials in LDS, 3: barrier,
// 4: cross-wave sum + barrier), -1: none
template <int MB, int G, int SB = (((G + 3) / 4) * MB >= 3 ? 5 : 9), typename Hook = NoHook>
__device__ __forceinline__ void fwd_core(float* __restrict__ out, float* __restrict__ red, const float* __restrict__ x,
                                         int M, int K, const float* __restrict__ W, const int (&row0)[G],
                                         Hook after_issue = Hook(), int ph = -1) {
  constexpr int T = (G + 3) / 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
  // (tried: 32- and 64-float steps so that a row's four q-lanes read whole 128 / 256-byte lines instead of 64-byte pieces
  // -- not faster: 13.2 / 14.6 us against 13.1 for the message kernel; these products are latency-, not sector-bound)
  bool first = true;
  for (int s0 = s_beg; first || s0 < s_end; s0 += SB) {
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
    if (first) { after_issue(); first = false; if (ph >= 0) DL_PH(ph, 1); }
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
  if (ph >= 0) DL_PH(ph, 2);
  __syncthreads();
  if (ph >= 0) DL_PH(ph, 3);
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
  if (ph >= 0) DL_PH(ph, 4);
}

// ---------------------------------------------------------------------------------------------- forward core, weights by LDS-DMA
// The same product with the block's weight rows brought to LDS by global_load_lds_dwordx4: no registers, every request
// a whole contiguous kilobyte of a 4-row group (wave w moves groups w, w + 9, ..), all of them in flight from the first
// instruction of the kernel.  (The register path reads 16 rows x 64 bytes per instruction, the operand layout of the
// MFMA: 6.2 us for the message product's 13 MB against 4.15 us in contiguous pieces, tools/probes/stream_probe.hip.)
// w_l: [G][4][K] floats.
template <int G>
__device__ __forceinline__ void wlds_issue(float* __restrict__ w_l, const float* __restrict__ W, int K, const int (&row0)[G]) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if ((g % DL_WAVES) != wave) continue;                                 // wave-uniform
    const float* src = W + (size_t)row0[g] * K;
    for (int u = 0; u * 64 < K; ++u) {                                    // K float4 = 4 rows of K floats
      const int idx = u * 64 + lane;
      if (idx < K)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const CGV_GLOBAL_AS void*>(reinterpret_cast<unsigned long long>(src + 4 * (size_t)idx)),
                                         (__attribute__((address_space(3))) void*)(w_l + (size_t)g * 4 * K + u * 256), 16, 0, 0);
    }
  }
}

// MB = 1; K <= 16 * 6 * 9 (one batch of steps per wave).  after_issue() runs behind the x requests, before the barrier
// that publishes the DMA'd weights.
// Two halves, so that the caller's own code (the commit of its staged arrays) runs between the x requests and the wait --
// as a callable handed through here it was kept in scratch memory once it captured more than a handful of registers.
constexpr int WLDS_SB = 6;
__device__ __forceinline__ void wlds_x_issue(float4 (&b)[WLDS_SB], const float* __restrict__ x, int M, int K) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int steps = (K + 15) / 16, per = (steps + DL_WAVES - 1) / DL_WAVES;
  const int s_beg = wave * per, s_end = min(s_beg + per, steps);
  const float* xrow = x + (size_t)min(i, M - 1) * K;
#pragma unroll
  for (int u = 0; u < WLDS_SB; ++u) {
    const int k = (s_beg + u) * 16 + 4 * q;
    b[u] = *reinterpret_cast<const float4*>(xrow + ((s_beg + u < s_end && k < K) ? k : 0));
  }
}
template <int G>
__device__ __forceinline__ void fwd_core_wlds_finish(float* __restrict__ out, float* __restrict__ red, const float4 (&b)[WLDS_SB],
                                                     int K, const float* __restrict__ w_l, int ph = -1) {
  constexpr int T = (G + 3) / 4, SB = WLDS_SB;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int steps = (K + 15) / 16, per = (steps + DL_WAVES - 1) / DL_WAVES;
  const int s_beg = wave * per, s_end = min(s_beg + per, steps);
  if (ph >= 0) DL_PH(ph, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ph >= 0) DL_PH(ph, 2);
  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int wrow[T];
#pragma unroll
  for (int t = 0; t < T; ++t) wrow[t] = (min(t * 4 + (i >> 2), G - 1) * 4 + (i & 3)) * K;      // surplus rows: clamped, unused
#pragma unroll
  for (int u = 0; u < SB; ++u) {
    const int k = (s_beg + u) * 16 + 4 * q;
    const bool kok = s_beg + u < s_end && k < K;
    if (s_beg + u < s_end) {                                              // wave-uniform
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float4 w = *reinterpret_cast<const float4*>(w_l + wrow[t] + (kok ? k : 0));
        if (!kok) w = make_float4(0.f, 0.f, 0.f, 0.f);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, b[u].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, b[u].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, b[u].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, b[u].w, acc[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave * T + t) * 4 + r) * 64 + lane] = acc[t][r];
  if (ph >= 0) DL_PH(ph, 3);
  __syncthreads();
  for (int o = threadIdx.x; o < T * 256; o += DL_THREADS) {
    const int l = o & 63, r = (o >> 6) & 3, t = o >> 8;
    const int g = t * 4 + (l >> 4), m = l & 15;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < DL_WAVES; ++w) v += red[((w * T + t) * 4 + r) * 64 + l];
    if (g < G) out[(m * G + g) * 4 + r] = v;
  }
  __syncthreads();
  if (ph >= 0) DL_PH(ph, 4);
}

// ---------------------------------------------------------------------------------------------- staging of the bead graph
// Edge records, CSR arrays and the node state of the block's 4 channels go to LDS in one batch of loads at the top of a
// message kernel: the edge loops then touch no global memory (they were chains of dependent gathers: index -> row).
constexpr int DL_MAX_EDGES = 240;          // 16 nodes, no self loops / duplicates
// The *_issue / *_commit pairs keep a kernel's staging loads in ONE batch: every load is unconditional (index clamped to
// a valid element; counts >= 1), straight-line, and the LDS stores come later -- a loop or a branch per array would put a
// vmcnt(0) between the arrays (each store waits for its own load, and loads return in order).
template <int SLOTS> struct Slots4 { float4 v[SLOTS]; };
template <int SLOTS>
__device__ __forceinline__ void copy4_issue(Slots4<SLOTS>& r, gcf src, int n4) {
#pragma unroll
  for (int u = 0; u < SLOTS; ++u) r.v[u] = ldg4_pinned(src + 4 * (size_t)min((int)threadIdx.x + u * DL_THREADS, n4 - 1));
}
template <int SLOTS>
__device__ __forceinline__ void copy4_commit(const Slots4<SLOTS>& r, float* __restrict__ dst, int n4) {
#pragma unroll
  for (int u = 0; u < SLOTS; ++u)
    if ((int)threadIdx.x + u * DL_THREADS < n4) reinterpret_cast<float4*>(dst)[threadIdx.x + u * DL_THREADS] = r.v[u];
}
__device__ __forceinline__ int int_issue(gci src, int n) { return ldgi_pinned(src + min((int)threadIdx.x, n - 1)); }
__device__ __forceinline__ void int_commit(int v, int* __restrict__ dst, int n) { if ((int)threadIdx.x < n) dst[threadIdx.x] = v; }
// [n][F] array at channels f0..f0+3 -> node-major [n][4]; src == NULL reads `valid` (same shape) and commits zeros
__device__ __forceinline__ float4 scalar_issue(gcf src, gcf valid, int n, int F, int f0) {
  return ldg4_pinned((src ? src : valid) + (size_t)min((int)threadIdx.x, n - 1) * F + f0);
}
__device__ __forceinline__ void scalar_commit(float4 v, bool have, float* __restrict__ dst, int n) {
  if ((int)threadIdx.x < n) reinterpret_cast<float4*>(dst)[threadIdx.x] = have ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}
// [n][F][3] array -> [n][4][3] (12 contiguous floats per node = 3 float4)
__device__ __forceinline__ float4 vector_issue(gcf src, gcf valid, int n, int F, int f0) {
  const int t = min((int)threadIdx.x, 3 * n - 1), m = t / 3, part = t - 3 * m;
  return ldg4_pinned((src ? src : valid) + ((size_t)m * F + f0) * 3 + 4 * part);
}
__device__ __forceinline__ void vector_commit(float4 v, bool have, float* __restrict__ dst, int n) {
  if ((int)threadIdx.x < 3 * n) reinterpret_cast<float4*>(dst)[threadIdx.x] = have ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}
constexpr int geom_slots(int R) { return (DL_MAX_EDGES * geom_stride(R) / 4 + DL_THREADS - 1) / DL_THREADS; }
__device__ __forceinline__ void stage_copy4(float* __restrict__ dst, const float* __restrict__ src, int n4) {
  for (int i = threadIdx.x; i < n4; i += DL_THREADS) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
}
__device__ __forceinline__ void stage_ints(int* __restrict__ dst, const int* __restrict__ src, int n) {
  for (int i = threadIdx.x; i < n; i += DL_THREADS) dst[i] = src[i];
}
// node-major [n][4] float4 of a [n][F] array at channels f0..f0+3
__device__ __forceinline__ void stage_scalar(float* __restrict__ dst, const float* __restrict__ src, int n, int F, int f0) {
  if (threadIdx.x < n)
    reinterpret_cast<float4*>(dst)[threadIdx.x] = src ? *reinterpret_cast<const float4*>(src + (size_t)threadIdx.x * F + f0)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
}
// [n][4][3] of a [n][F][3] array (12 contiguous floats per node)
__device__ __forceinline__ void stage_vector(float* __restrict__ dst, const float* __restrict__ src, int n, int F, int f0) {
  if (threadIdx.x < 3 * n) {
    const int m = threadIdx.x / 3, part = threadIdx.x - 3 * m;
    reinterpret_cast<float4*>(dst)[threadIdx.x] =
        src ? *reinterpret_cast<const float4*>(src + ((size_t)m * F + f0) * 3 + 4 * part) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ dv3 lds_v3(const float* p) { return dv3{p[0], p[1], p[2]}; }

// ============================================================================================== F2: phi + message forward
template <int R, bool WLDS>
__global__ __launch_bounds__(DL_THREADS) void dec_msg_fwd_k(
    const float* __restrict__ a1, const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ s_,
    const float* __restrict__ sbar_, const float* __restrict__ v_, const float* __restrict__ vbar_,
    const float* __restrict__ geom_, const int* __restrict__ rowptr_, const int* __restrict__ src_,
    const float* __restrict__ Wd_, const float* __restrict__ bd_, float* __restrict__ phi_out, float* __restrict__ stack,
    float* __restrict__ sbar_out, float* __restrict__ v_out, float* __restrict__ vbar_out, float* __restrict__ rows_out,
    int n, int F, int E) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  Carve cv;
  float* red = cv.take(fwd_red_floats<1, 9>());
  float* phi_l = cv.take(16 * 9 * 4);
  float* red2 = cv.take(8 * 3 * 64);
  float* geom_l = cv.take((size_t)DL_MAX_EDGES * GS);
  int* rp_l = reinterpret_cast<int*>(cv.take(20));
  int* src_l = reinterpret_cast<int*>(cv.take(DL_MAX_EDGES));
  float* s_l = cv.take(64); float* sb_l = cv.take(64);
  float* v_l = cv.take(192); float* vb_l = cv.take(192);
  float* w_l = cv.take(WLDS ? (size_t)36 * F : 0);
  const int f0 = blockIdx.x * DL_CB;
  const int lane = threadIdx.x & 63;
  const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane >> 2, c = lane & 3;
  const bool live = i < n;
  const int f = f0 + c;
  int row0[9];
#pragma unroll
  for (int g = 0; g < 9; ++g) row0[g] = g * F + f0;
  if (WLDS) wlds_issue<9>(w_l, W2, F, row0);
  DL_PH(1, 0);
  DL_SPAN(1, 0);
  const gcf geom = launder(geom_); const gci rowptr = launder(rowptr_); const gci src = launder(src_);
  const gcf s = launder(s_); const gcf sbar = launder(sbar_); const gcf v = launder(v_);
  const gcf vbar = launder(vbar_); const gcf Wd = launder(Wd_); const gcf bd = launder(bd_);
  // bead graph + node state of these channels: requested in one batch with the product's operands, stored to LDS behind
  // the product's own requests (consumed after the product)
  // (one wave per small array, branch-free roles: see dec_msg_bwd_k.  float4: wave 0 [s | sbar] (32 lanes each), 1 v,
  //  2 vbar (3 n lanes); ids: waves 0..3 src (64 each), 4 rowptr)
  const int n4 = E * GS / 4;
  // (the record copy stays unconditional here: its registers are captured by the commit hook below, and a conditionally
  //  written capture is kept in scratch memory)
  Slots4<geom_slots(R)> r_geom;
  copy4_issue(r_geom, geom, n4);
  const int half = k == 0 ? lane >> 5 : 0;
  const gcf cand = k == 0 ? (half == 0 ? s : sbar) : k == 1 ? v : vbar;
  const int sm_D = k == 0 ? 1 : 3, sm_cnt = k == 0 ? n : 3 * n, sm_A = k == 0 ? F : 3 * F, sm_B = k == 0 ? 0 : 4;
  const int sm_C = k == 0 ? f0 : 3 * f0, sm_item = k == 0 ? (lane & 31) : lane;
  float4 r_small;
  {
    const int it = min(sm_item, sm_cnt - 1);
    const int m = (it * (sm_D == 1 ? 65536 : 21846)) >> 16, part = it - m * sm_D;                       // it < 48
    r_small = ldg4_pinned(cand + ((size_t)m * sm_A + part * sm_B + sm_C));
  }
  const int id_item = k < 4 ? k * 64 + lane : lane, id_cnt = k < 4 ? E : n + 1;
  const int r_ids = ldgi_pinned((k < 4 ? src : rowptr) + min(id_item, id_cnt - 1));
  float W[R + 1];
#pragma unroll
  for (int nn = 0; nn < R; ++nn) W[nn] = ldg_pinned(Wd + ((size_t)k * F + f) * R + nn);
  W[R] = ldg_pinned(bd + (size_t)k * F + f);
  pin_loads();
  // (the commit is written out at its place: as a callable it was kept in scratch memory)
  float4 xb[WLDS_SB];
  if (WLDS) wlds_x_issue(xb, a1, n, F);
  else fwd_core<1, 9, 5>(phi_l, red, a1, n, F, W2, row0, NoHook(), 1);  // (register path: whole product first; the bias pass's barrier publishes the commit)
  {
    copy4_commit(r_geom, geom_l, n4);
    float* dst = k == 0 ? (half == 0 ? s_l : sb_l) : k == 1 ? v_l : vb_l;
    if (k <= 2 && sm_item < sm_cnt) reinterpret_cast<float4*>(dst)[sm_item] = r_small;
    if (k <= 4 && id_item < id_cnt) (k < 4 ? src_l : rp_l)[id_item] = r_ids;
  }
  if (WLDS) fwd_core_wlds_finish<9>(phi_l, red, xb, F, w_l, 1);
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
  DL_PH(1, 5);
  // EquiMessagePsuedo, wave k = filter k (pseudo_msg.hip: pseudo_fwd_k), lane = (receiver i, channel c)
  const int ic = live ? i : 0;
  const float s_i = s_l[ic * 4 + c], sb_i = sb_l[ic * 4 + c];
  const dv3 v_i = lds_v3(v_l + (ic * 4 + c) * 3), vb_i = lds_v3(vb_l + (ic * 4 + c) * 3);
  float ah = 0.f, ahb = 0.f;
  dv3 acc{0.f, 0.f, 0.f};
  const float* __restrict__ vsrc = (k == 2 || k == 6 || k == 7) ? v_l : vb_l;
  const int e_beg = live ? rp_l[i] : 0, e_end = live ? min(rp_l[i + 1], E) : 0;      // E = records staged (>= the graph's edges)
  int j_nx = e_beg < e_end ? src_l[e_beg] : 0;                            // (the next edge's source id travels with this edge's operands)
  for (int e = e_beg; e < e_end; ++e) {
    const int j = j_nx;
    j_nx = src_l[min(e + 1, e_end - 1)];
    const float* __restrict__ g = geom_l + (size_t)e * GS;
    const dv3 vj = lds_v3(vsrc + (j * 4 + c) * 3);
    const float q = phi_l[(j * 9 + k) * 4 + c] * dfilt<R>(W, g);
    switch (k) {                                   // wave-uniform
      case 0: ah = fmaf(q, s_i, ah); ahb += ddot(v_i, vj); break;
      case 1: daxpy(acc, q, dv3{g[U], g[U + 1], g[U + 2]}); break;
      case 2: daxpy(acc, q, vj); break;
      case 3: daxpy(acc, q, dcross(v_i, vj)); break;
      case 4: daxpy(acc, q * sb_i, vj); break;
      case 5: daxpy(acc, q, vj); break;
      case 6: daxpy(acc, q * sb_i, vj); break;
      case 7: daxpy(acc, q, dcross(v_i, vj)); break;
      default: daxpy(acc, q, dcross(vb_i, vj)); break;
    }
  }
  DL_PH(1, 6);
  if (k > 0) { red2[((k - 1) * 3 + 0) * 64 + lane] = acc.x; red2[((k - 1) * 3 + 1) * 64 + lane] = acc.y; red2[((k - 1) * 3 + 2) * 64 + lane] = acc.z; }
  __syncthreads();
  if (k != 0 || !live) { DL_SPAN(1, 1); return; }
  dv3 av{0.f, 0.f, 0.f}, avb{0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 4; ++w) { av.x += red2[(w * 3 + 0) * 64 + lane]; av.y += red2[(w * 3 + 1) * 64 + lane]; av.z += red2[(w * 3 + 2) * 64 + lane]; }
#pragma unroll
  for (int w = 4; w < 8; ++w) { avb.x += red2[(w * 3 + 0) * 64 + lane]; avb.y += red2[(w * 3 + 1) * 64 + lane]; avb.z += red2[(w * 3 + 2) * 64 + lane]; }
  // updated states (cgvae.py:108-111)
  ah += s_i; ahb += sb_i;
  av.x += v_i.x; av.y += v_i.y; av.z += v_i.z;
  avb.x += vb_i.x; avb.y += vb_i.y; avb.z += vb_i.z;
  const size_t nf = (size_t)i * F + f;
  stack[(size_t)i * 2 * F + f] = ah;                    // S' is the first half of the update block's stack (conv.py:601)
  sbar_out[nf] = ahb;
  st3(v_out + nf * 3, av.x, av.y, av.z);
  st3(vbar_out + nf * 3, avb.x, avb.y, avb.z);
  rows_out[((size_t)3 * i + 0) * F + f] = av.x;
  rows_out[((size_t)3 * i + 1) * F + f] = av.y;
  rows_out[((size_t)3 * i + 2) * F + f] = av.z;
  DL_PH(1, 7);
  DL_SPAN(1, 1);
}

// ============================================================================================== F1 / F4: full-width Dense
// y = act(x W^T + b), z = pre-activation, for M <= 16 rows: N / 4 blocks of 4 output columns each (150 CUs pull the
// weight instead of the 38 of skinny_fwd_k's 16-column blocks).
__global__ __launch_bounds__(DL_THREADS) void dec_dense_fwd_k(const float* __restrict__ x, const float* __restrict__ W,
                                                              const float* __restrict__ bias_, float* __restrict__ y,
                                                              float* __restrict__ zout, int n, int N, int K, int act) {
  Carve cv;
  float* red = cv.take(fwd_red_floats<1, 1>());
  float* o_l = cv.take(16 * 4);
  const int n0 = blockIdx.x * DL_CB;
  const int row0[1] = {n0};
  const bool mine = threadIdx.x < n * 4;
  const int i = threadIdx.x >> 2, c = threadIdx.x & 3;
  const gcf bias = launder(bias_);
  const float b = (mine && bias) ? ldg_pinned(bias + n0 + c) : 0.f;
  pin_loads();
  const int ph = K > N ? 3 : 0;
  DL_SPAN(ph, 0);
  DL_PH(ph, 0);
  // one batch of requests per wave either way; the narrower batch issues no clamped surplus loads at K <= 720
  if (K <= 16 * DL_WAVES * 5) fwd_core<1, 1, 5>(o_l, red, x, n, K, W, row0, NoHook(), ph);
  else fwd_core<1, 1, 9>(o_l, red, x, n, K, W, row0, NoHook(), ph);
  if (mine) {
    const float zv = o_l[i * 4 + c] + b;
    const size_t at = (size_t)i * N + n0 + c;
    if (act) { if (zout) zout[at] = zv; y[at] = act_fwd(zv, act); } else y[at] = zv;
  }
  DL_PH(ph, 5);
  DL_SPAN(ph, 1);
}

// Up to four Dense layers of one shape in one launch (blockIdx.y picks the layer): the mu / sigma heads are independent of
// each other (cgvae.py:366-371, 502-503) and the prior's pair (cgvae.py:398-401) of the encoder's; x may be shared.
constexpr int DL_MULTI_MAX = 4;            // (the four heads of a step: prior mu / sigma, encoder mu / sigma)
struct DensePair {
  const float* x[DL_MULTI_MAX]; const float* W[DL_MULTI_MAX]; const float* bias[DL_MULTI_MAX];
  float* y[DL_MULTI_MAX]; float* z[DL_MULTI_MAX];
  int act[DL_MULTI_MAX];
};
// CQ = column quads per block: 1 (N / 4 blocks per problem) while the launch fits the chip in one round, 2 when three or
// four problems would need more than 256 blocks -- the product of a block is one 16-row MFMA tile either way, so the
// 8-column block costs what the 4-column block costs and the launch takes one round instead of 2.3.
template <int CQ>
__global__ __launch_bounds__(DL_THREADS) void dec_dense_fwd_pair_k(DensePair p, int n, int N, int K) {
  Carve cv;
  float* red = cv.take(fwd_red_floats<1, CQ>());
  float* o_l = cv.take(16 * CQ * 4);
  const int j = blockIdx.y;
  const float* __restrict__ x = p.x[j];
  const float* __restrict__ W = p.W[j];
  float* __restrict__ y = p.y[j];
  float* __restrict__ zout = p.z[j];
  const int act = p.act[j];
  const int n0 = blockIdx.x * DL_CB * CQ;
  int row0[CQ];
#pragma unroll
  for (int g = 0; g < CQ; ++g) row0[g] = n0 + 4 * g;
  const bool mine = threadIdx.x < n * 4 * CQ;
  const int i = threadIdx.x / (4 * CQ), c = threadIdx.x - i * 4 * CQ;      // column n0 + c of row i
  const gcf bias = launder(p.bias[j]);
  const float b = (mine && bias) ? ldg_pinned(bias + n0 + c) : 0.f;
  pin_loads();
  if (K <= 16 * DL_WAVES * 5) fwd_core<1, CQ, 5>(o_l, red, x, n, K, W, row0);
  else fwd_core<1, CQ, 9>(o_l, red, x, n, K, W, row0);
  if (mine) {
    const float zv = o_l[(i * CQ + (c >> 2)) * 4 + (c & 3)] + b;
    const size_t at = (size_t)i * N + n0 + c;
    if (act) { if (zout) zout[at] = zv; y[at] = act_fwd(zv, act); } else y[at] = zv;
  }
}

// ============================================================================================== F3: [U | Vv] + norm
__global__ __launch_bounds__(DL_THREADS) void dec_uv_fwd_k(const float* __restrict__ rows, const float* __restrict__ Wuv,
                                                           float* __restrict__ UV, float* __restrict__ stack, int n, int F) {
  Carve cv;
  float* red = cv.take(fwd_red_floats<3, 2>());
  float* uv_l = cv.take(48 * 2 * 4);
  const int f0 = blockIdx.x * DL_CB;
  const int row0[2] = {f0, F + f0};
  DL_SPAN(2, 0);
  DL_PH(2, 0);
  fwd_core<3, 2>(uv_l, red, rows, 3 * n, F, Wuv, row0, NoHook(), 2);
  for (int o = threadIdx.x; o < 3 * n * 2; o += DL_THREADS) {
    const int m = o >> 1, g = o & 1;
    *reinterpret_cast<float4*>(UV + (size_t)m * 2 * F + (size_t)g * F + f0) = *reinterpret_cast<const float4*>(uv_l + (m * 2 + g) * 4);
  }
  if (threadIdx.x < n * 4) {
    const int i = threadIdx.x >> 2, c = threadIdx.x & 3;
    const float x = uv_l[((3 * i + 0) * 2 + 1) * 4 + c], y = uv_l[((3 * i + 1) * 2 + 1) * 4 + c], z = uv_l[((3 * i + 2) * 2 + 1) * 4 + c];
    stack[(size_t)i * 2 * F + F + f0 + c] = sqrtf(((x * x + 1e-10f) + (y * y + 1e-10f)) + (z * z + 1e-10f));      // conv.py:600
  }
  DL_PH(2, 5);
  DL_SPAN(2, 1);
}

// The same by NODE GROUPS (grid y) of 8-channel blocks: the 48-row form above is bound by the fp32 MFMA pipe of its CU (three
// 16-row blocks x 8 of 16 weight rows per tile: 37 % of the instruction's outputs used, 1.8 us of MFMAs on the SIMD that
// hosts the most waves).  A part takes npp nodes (3 npp <= 16 rows: one row block) and 16 weight rows (a full tile): a
// third of the MFMAs per block, a third of the x rows; the norm is local to a node, so parts never meet.
__global__ __launch_bounds__(DL_THREADS) void dec_uv_fwd_nodes_k(const float* __restrict__ rows, const float* __restrict__ Wuv,
                                                                 float* __restrict__ UV, float* __restrict__ stack, int n, int F,
                                                                 int npp) {
  Carve cv;
  float* red = cv.take(fwd_red_floats<1, 4>());
  float* uv_l = cv.take(16 * 4 * 4);
  const int f0 = blockIdx.x * 8;
  const int row0[4] = {f0, f0 + 4, F + f0, F + f0 + 4};
  const int i0 = blockIdx.y * npp, ni = min(npp, n - i0);                // this part's nodes i0 .. i0 + ni - 1
  DL_SPAN(2, 0);
  DL_PH(2, 0);
  const float* x = rows + (size_t)3 * i0 * F;
  if (F <= 16 * DL_WAVES * 5) fwd_core<1, 4, 5>(uv_l, red, x, 3 * ni, F, Wuv, row0, NoHook(), 2);
  else fwd_core<1, 4, 9>(uv_l, red, x, 3 * ni, F, Wuv, row0, NoHook(), 2);
  for (int o = threadIdx.x; o < 3 * ni * 4; o += DL_THREADS) {
    const int m = o >> 2, g = o & 3;
    *reinterpret_cast<float4*>(UV + (size_t)(3 * i0 + m) * 2 * F + (size_t)(g >> 1) * F + f0 + 4 * (g & 1)) =
        *reinterpret_cast<const float4*>(uv_l + (m * 4 + g) * 4);
  }
  if (threadIdx.x < ni * 8) {
    const int il = threadIdx.x >> 3, c = threadIdx.x & 7, g = 2 + (c >> 2), c4 = c & 3;
    const float x_ = uv_l[((3 * il + 0) * 4 + g) * 4 + c4], y_ = uv_l[((3 * il + 1) * 4 + g) * 4 + c4], z_ = uv_l[((3 * il + 2) * 4 + g) * 4 + c4];
    stack[(size_t)(i0 + il) * 2 * F + F + f0 + c] = sqrtf(((x_ * x_ + 1e-10f) + (y_ * y_ + 1e-10f)) + (z_ * z_ + 1e-10f));      // conv.py:600
  }
  DL_PH(2, 5);
  DL_SPAN(2, 1);
}

// ============================================================================================== F5: a + gate
__global__ __launch_bounds__(DL_THREADS) void dec_gate_fwd_k(const float* __restrict__ a0, const float* __restrict__ W1p,
                                                             const float* __restrict__ b1p_, const float* __restrict__ UV_,
                                                             const float* __restrict__ stack_, const float* __restrict__ v2_,
                                                             float* __restrict__ a_out, float* __restrict__ s3,
                                                             float* __restrict__ v3, int n, int F) {
  Carve cv;
  float* red = cv.take(fwd_red_floats<1, 3>());
  float* a_l = cv.take(16 * 3 * 4);
  const int f0 = blockIdx.x * DL_CB;
  const int row0[3] = {f0, F + f0, 2 * F + f0};
  // the gate's other operands: requested before the product
  float ux = 0.f, uy = 0.f, uz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f, s2 = 0.f, bvv = 0.f, bsv = 0.f, bss = 0.f;
  f3 r{0.f, 0.f, 0.f};
  const bool mine = threadIdx.x < n * 4;
  const int i = threadIdx.x >> 2, c = threadIdx.x & 3, f = f0 + c;
  DL_SPAN(4, 0);
  DL_PH(4, 0);
  const gcf UV = launder(UV_); const gcf v2 = launder(v2_); const gcf stack = launder(stack_);
  const gcf b1p = launder(b1p_);
  if (mine) {
    const size_t b = (size_t)i * 3 * 2 * F + f;
    ux = ldg_pinned(UV + b); uy = ldg_pinned(UV + b + 2 * F); uz = ldg_pinned(UV + b + 4 * F);
    vx = ldg_pinned(UV + b + F); vy = ldg_pinned(UV + b + 2 * F + F); vz = ldg_pinned(UV + b + 4 * F + F);
    r = ld3_pinned(v2 + ((size_t)i * F + f) * 3);
    s2 = ldg_pinned(stack + (size_t)i * 2 * F + f);
    bvv = ldg_pinned(b1p + f); bsv = ldg_pinned(b1p + F + f); bss = ldg_pinned(b1p + 2 * F + f);
  }
  pin_loads();
  if (F <= 16 * DL_WAVES * 5) fwd_core<1, 3, 5>(a_l, red, a0, n, F, W1p, row0, NoHook(), 4);
  else fwd_core<1, 3, 9>(a_l, red, a0, n, F, W1p, row0, NoHook(), 4);
  if (mine) {
    const float a_vv = a_l[(i * 3 + 0) * 4 + c] + bvv, a_sv = a_l[(i * 3 + 1) * 4 + c] + bsv, a_ss = a_l[(i * 3 + 2) * 4 + c] + bss;
    float* ao = a_out + (size_t)i * 3 * F + f;
    ao[0] = a_vv; ao[F] = a_sv; ao[2 * F] = a_ss;
    const size_t nf = (size_t)i * F + f;
    st3(v3 + nf * 3, ux * a_vv + r.x, uy * a_vv + r.y, uz * a_vv + r.z);                  // conv.py:607, cgvae.py:123
    s3[nf] = ((ux * vx + uy * vy + uz * vz) * a_sv + a_ss) + s2;                           // conv.py:612-614, cgvae.py:122
  }
  DL_PH(4, 5);
  DL_SPAN(4, 1);
}

// ============================================================================================== UpdateBlock forward for 17 .. 96 bead rows
// The two epilogues above for bead graphs BEYOND the channel-group decoder's 16 nodes (dipeptide 96 beads, the 2000-atom
// graph 64): the per-block path ran the product (tile / skinny kernel), then an element-wise launch (update_gate_fwd,
// update_norm_stack_fwd: 4.6 - 4.9 us each, a boundary and two round trips for a pass over n x 600 values).  The gate and
// the norm are local in the channel (gate: and in the row; norm: in the node), so the same channel-group blocks compute
// them in the product's epilogue at any row count: MB row blocks of 16 (reference: conv.py:593-616).
// (Forward only: the backward of the per-block path exchanges dense tensors, not slices.)
// MEASURED SLOWER than what it replaces and therefore off by default (options.HOST["update_fused_fwd"]): dipeptide 2.645 /
// 2.675 against 2.636 / 2.637 ms per step, the 2000-atom graph 5.996 against 5.968 -- a 96-row channel-group block is
// ~900 fp32 MFMAs on ONE CU (3.7 us on its busiest SIMD), the tile kernels spread the same product over the whole chip and
// the element-wise launch behind them costs less than that difference.  Kept for the A/B and its parity tests.
template <int MB>
__global__ __launch_bounds__(DL_THREADS) void upd_gate_fwd_rows_k(const float* __restrict__ a0, const float* __restrict__ W1p,
                                                                  const float* __restrict__ b1p, const float* __restrict__ UV,
                                                                  const float* __restrict__ s_res, const float* __restrict__ v_res,
                                                                  float* __restrict__ a_out, float* __restrict__ s_out,
                                                                  float* __restrict__ v_out, int n, int F) {
  Carve cv;
  float* red = cv.take(fwd_red_floats<MB, 3>());
  float* a_l = cv.take(16 * MB * 3 * 4);
  const int f0 = blockIdx.x * DL_CB;
  const int row0[3] = {f0, F + f0, 2 * F + f0};
  fwd_core<MB, 3, (MB == 1 ? 5 : 2)>(a_l, red, a0, n, F, W1p, row0);
  for (int o = threadIdx.x; o < n * 4; o += DL_THREADS) {
    const int i = o >> 2, c = o & 3, f = f0 + c;
    const float a_vv = a_l[(i * 3 + 0) * 4 + c] + b1p[f], a_sv = a_l[(i * 3 + 1) * 4 + c] + b1p[F + f],
                a_ss = a_l[(i * 3 + 2) * 4 + c] + b1p[2 * F + f];
    float* ao = a_out + (size_t)i * 3 * F + f;
    ao[0] = a_vv; ao[F] = a_sv; ao[2 * F] = a_ss;
    const size_t b = (size_t)i * 3 * 2 * F + f, nf = (size_t)i * F + f;
    const float ux = UV[b], uy = UV[b + 2 * F], uz = UV[b + 4 * F];
    const float vx = UV[b + F], vy = UV[b + 3 * F], vz = UV[b + 5 * F];
    f3 r{0.f, 0.f, 0.f};
    if (v_res) r = ld3(v_res + nf * 3);
    st3(v_out + nf * 3, ux * a_vv + r.x, uy * a_vv + r.y, uz * a_vv + r.z);                // conv.py:607 (+ cgvae.py:123)
    s_out[nf] = ((ux * vx + uy * vy + uz * vz) * a_sv + a_ss) + (s_res ? s_res[nf] : 0.f);  // conv.py:612-614 (+ cgvae.py:122)
  }
}

// [U | Vv] = rows [u_mat; v_mat]^T for the block's 8 channels and its node group (npp nodes: 3 npp <= 16 MB rows), the norm
// of Vv and -- the update block's stack is [s | norm] (conv.py:601) -- the copy of s into the stack's first half.
template <int MB>
__global__ __launch_bounds__(DL_THREADS) void upd_uv_norm_fwd_rows_k(const float* __restrict__ rows, const float* __restrict__ Wuv,
                                                                     const float* __restrict__ s_in, float* __restrict__ UV,
                                                                     float* __restrict__ stack, int n, int F, int npp) {
  Carve cv;
  float* red = cv.take(fwd_red_floats<MB, 4>());
  float* uv_l = cv.take(16 * MB * 4 * 4);
  const int f0 = blockIdx.x * 8;
  const int row0[4] = {f0, f0 + 4, F + f0, F + f0 + 4};
  const int i0 = blockIdx.y * npp, ni = min(npp, n - i0);
  const float* x = rows + (size_t)3 * i0 * F;
  fwd_core<MB, 4, (MB == 1 ? 5 : 2)>(uv_l, red, x, 3 * ni, F, Wuv, row0);
  for (int o = threadIdx.x; o < 3 * ni * 4; o += DL_THREADS) {
    const int m = o >> 2, g = o & 3;
    *reinterpret_cast<float4*>(UV + (size_t)(3 * i0 + m) * 2 * F + (size_t)(g >> 1) * F + f0 + 4 * (g & 1)) =
        *reinterpret_cast<const float4*>(uv_l + (m * 4 + g) * 4);
  }
  for (int o = threadIdx.x; o < ni * 8; o += DL_THREADS) {
    const int il = o >> 3, c = o & 7, g = 2 + (c >> 2), c4 = c & 3;
    const float x_ = uv_l[((3 * il + 0) * 4 + g) * 4 + c4], y_ = uv_l[((3 * il + 1) * 4 + g) * 4 + c4], z_ = uv_l[((3 * il + 2) * 4 + g) * 4 + c4];
    float* st = stack + (size_t)(i0 + il) * 2 * F;
    st[F + f0 + c] = sqrtf(((x_ * x_ + 1e-10f) + (y_ * y_ + 1e-10f)) + (z_ * z_ + 1e-10f));       // conv.py:600
    if (s_in) st[f0 + c] = s_in[(size_t)(i0 + il) * F + f0 + c];
  }
}

// ============================================================================================== B1: gate backward + W1' rows
// CBQ = channel quads per block (1: 4 channels, F / 4 blocks; 2: 8 channels, F / 8 blocks -- half as many, twice as fat
// slices: a phase's slice volume is blocks x rows x K, so fatter blocks halve what the next phase reads back).
template <int CBQ, int QS, int NT>
__global__ __launch_bounds__(DL_THREADS) void dec_gate_bwd_k(
    const float* __restrict__ UV_, const float* __restrict__ a_, const float* __restrict__ gs_base_,
    const float* __restrict__ gs_slices_, int gs_n, long long gs_stride, const float* __restrict__ gv_,
    const float* __restrict__ W1p_, float* __restrict__ ga, float* __restrict__ gUV, float* __restrict__ gs_sum,
    float* __restrict__ slices_out, long long out_stride, int n, int F) {
  constexpr int C = 4 * CBQ, G = 3 * CBQ;
  Carve cv;
  float* stage = cv.take(bi_stage_floats<1>());
  float4* scratch = reinterpret_cast<float4*>(cv.take(CBQ * 36 * 16 * 4));
  float4* gs_l = reinterpret_cast<float4*>(cv.take(CBQ * 16 * 4));
  float* g_l = cv.take(16 * G * 4);
  const int f0 = blockIdx.x * C;
  int row0[G];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int q = 0; q < CBQ; ++q) row0[g * CBQ + q] = g * F + f0 + 4 * q;
  DL_SPAN(5, 0);
  DL_PH(5, 0);
  const gcf UV = launder(UV_); const gcf a = launder(a_); const gcf gs_base = launder(gs_base_);
  const gcf gs_slices = launder(gs_slices_); const gcf gv = launder(gv_); const gcf W1p = launder(W1p_);
  // order of requests: the slices and the prologue's own operands first, the weights behind them
  const bool have_slices = gs_slices && gs_n > 0;
  int kq[CBQ];
#pragma unroll
  for (int q = 0; q < CBQ; ++q) kq[q] = have_slices ? (int)blockIdx.x * CBQ + q : 0;
  QuadRegs<1, CBQ, QS> qr;
  quad_issue<1, CBQ>(qr, have_slices ? gs_slices : UV, have_slices ? gs_n : 1, have_slices ? gs_stride : 0, have_slices ? n : 1, kq);
  const bool mine = threadIdx.x < n * C;
  const int i = threadIdx.x / C, c = threadIdx.x - i * C, f = f0 + c;
  const size_t nf = (size_t)i * F + f;
  const size_t b = (size_t)i * 3 * 2 * F + f, cc = (size_t)i * 3 * F + f;
  float ux = 0.f, uy = 0.f, uz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f, a_vv = 0.f, a_sv = 0.f, gsb = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
  if (mine) {
    ux = ldg_pinned(UV + b); uy = ldg_pinned(UV + b + 2 * F); uz = ldg_pinned(UV + b + 4 * F);
    vx = ldg_pinned(UV + b + F); vy = ldg_pinned(UV + b + 2 * F + F); vz = ldg_pinned(UV + b + 4 * F + F);
    a_vv = ldg_pinned(a + cc); a_sv = ldg_pinned(a + cc + F);
    if (gs_base) gsb = ldg_pinned(gs_base + nf);
    if (gv) { const f3 t = ld3_pinned(gv + nf * 3); gx = t.x; gy = t.y; gz = t.z; }
  }
  const TileRange tr = dl_tile_range(F);
  const bool part0 = blockIdx.y == 0;                                   // dense outputs: written once per channel group
  BiRegs<G, NT> wr;
  bi_prefetch<G, NT>(wr, W1p, F, row0, tr);
  pin_loads();
  DL_PH(5, 1);
  for (int o = threadIdx.x; o < 16 * G * 4; o += DL_THREADS) g_l[o] = 0.f;
  quad_finish<1, CBQ>(qr, gs_l, scratch, have_slices ? gs_n : 0, n, 5);      // no slices: every class empty -> zeros
  DL_PH(5, 4);
  if (mine) {
    const int q = c >> 2, c4 = c & 3;
    const float gs = reinterpret_cast<const float*>(gs_l + q * 16)[i * 4 + c4] + gsb;
    const float inner = ux * vx + uy * vy + uz * vz;
    const float cs = gs * a_sv;
    const float g0 = gx * ux + gy * uy + gz * uz, g1 = gs * inner, g2 = gs;
    g_l[(i * G + 0 * CBQ + q) * 4 + c4] = g0; g_l[(i * G + 1 * CBQ + q) * 4 + c4] = g1; g_l[(i * G + 2 * CBQ + q) * 4 + c4] = g2;
    if (part0) {
      gs_sum[nf] = gs;
      ga[cc] = g0; ga[cc + F] = g1; ga[cc + 2 * F] = g2;
      gUV[b] = fmaf(gx, a_vv, cs * vx); gUV[b + 2 * F] = fmaf(gy, a_vv, cs * vy); gUV[b + 4 * F] = fmaf(gz, a_vv, cs * vz);
      gUV[b + F] = cs * ux; gUV[b + 2 * F + F] = cs * uy; gUV[b + 4 * F + F] = cs * uz;
    }
  }
  __syncthreads();
  DL_PH(5, 6);
  bi_core<1, G, NT>(wr, g_l, stage, slices_out + (size_t)blockIdx.x * out_stride, F, n, tr, nullptr, nullptr, 5);
  DL_PH(5, 8);
  DL_SPAN(5, 1);
}

// ============================================================================================== B2 / B5: slice sum, act', CBQ row groups
template <int NT, int CBQ, int QS>
__global__ __launch_bounds__(DL_THREADS) void dec_dense_bwd_k(const float* __restrict__ g_slices_, int g_n, long long g_stride,
                                                              const float* __restrict__ z_, int act, const float* __restrict__ W_,
                                                              float* __restrict__ g_dense, float* __restrict__ slices_out,
                                                              long long out_stride, int n, int N, int K) {
  constexpr int C = 4 * CBQ;
  Carve cv;
  float* stage = cv.take(bi_stage_floats<1>());
  float4* scratch = reinterpret_cast<float4*>(cv.take(CBQ * 36 * 16 * 4));
  float4* sum_l = reinterpret_cast<float4*>(cv.take(CBQ * 16 * 4));
  float* g_l = cv.take(16 * CBQ * 4);
  const int n0 = blockIdx.x * C;
  const int ph = K > N ? 6 : 9;
  DL_SPAN(ph, 0);
  DL_PH(ph, 0);
  const gcf g_slices = launder(g_slices_); const gcf z = launder(z_); const gcf W = launder(W_);
  int row0[CBQ], kq[CBQ];
#pragma unroll
  for (int q = 0; q < CBQ; ++q) { row0[q] = n0 + 4 * q; kq[q] = (int)blockIdx.x * CBQ + q; }
  QuadRegs<1, CBQ, QS> qr;
  quad_issue<1, CBQ>(qr, g_slices, g_n, g_stride, n, kq);
  const int i = threadIdx.x / C, c = threadIdx.x - i * C;
  const size_t at = (size_t)i * N + n0 + c;
  float zz = 0.f;
  if (threadIdx.x < 16 * C && i < n && act) zz = ldg_pinned(z + at);
  const TileRange tr = dl_tile_range(K);
  BiRegs<CBQ, NT> wr;
  bi_prefetch<CBQ, NT>(wr, W, K, row0, tr);
  pin_loads();
  DL_PH(ph, 1);
  quad_finish<1, CBQ>(qr, sum_l, scratch, g_n, n, ph);
  DL_PH(ph, 4);
  if (threadIdx.x < 16 * C) {
    const int q = c >> 2, c4 = c & 3;
    float g = 0.f;
    if (i < n) {
      g = reinterpret_cast<const float*>(sum_l + q * 16)[i * 4 + c4];
      if (blockIdx.y == 0) g_dense[at] = g;                               // the weight-gradient launch applies act'(z) itself
      if (act) g *= act_bwd(zz, act);
    }
    g_l[(i * CBQ + q) * 4 + c4] = g;
  }
  __syncthreads();
  DL_PH(ph, 6);
  bi_core<1, CBQ, NT>(wr, g_l, stage, slices_out + (size_t)blockIdx.x * out_stride, K, n, tr, nullptr, nullptr, ph);
  DL_PH(ph, 8);
  DL_SPAN(ph, 1);
}

// ============================================================================================== B3: norm backward + [Wu; Wv] rows
template <int CBQ, int QS, int NT>
__global__ __launch_bounds__(DL_THREADS) void dec_uv_bwd_k(const float* __restrict__ gstack_slices_, int gs_n, long long gs_stride,
                                                           const float* __restrict__ UV_, const float* __restrict__ stack_,
                                                           const float* __restrict__ gs_res_, const float* __restrict__ Wuv_,
                                                           const float* gUV_in_, float* gUV_out, float* __restrict__ g_s2,
                                                           float* __restrict__ slices_out, long long out_stride, int n, int F) {
  constexpr int C = 4 * CBQ, G = 2 * CBQ;
  Carve cv;
  float* stage = cv.take(bi_stage_floats<3>());
  float4* scratch = reinterpret_cast<float4*>(cv.take(G * 36 * 16 * 4));
  float4* gsum_l = reinterpret_cast<float4*>(cv.take(G * 16 * 4));       // [q]: columns f0 + 4 q .. of g_stack, [CBQ + q]: the norm half
  float* g_l = cv.take(48 * G * 4);
  const int f0 = blockIdx.x * C;
  DL_SPAN(7, 0);
  DL_PH(7, 0);
  const gcf gstack_slices = launder(gstack_slices_); const gcf UV = launder(UV_); const gcf stack = launder(stack_);
  const gcf gs_res = launder(gs_res_); const gcf Wuv = launder(Wuv_);
  const gcf gUV_in = launder(gUV_in_);
  int row0[G], kq[G];
#pragma unroll
  for (int q = 0; q < CBQ; ++q) {
    row0[q] = f0 + 4 * q; row0[CBQ + q] = F + f0 + 4 * q;
    kq[q] = (int)blockIdx.x * CBQ + q; kq[CBQ + q] = F / 4 + (int)blockIdx.x * CBQ + q;
  }
  QuadRegs<1, G, QS> qr;
  quad_issue<1, G>(qr, gstack_slices, gs_n, gs_stride, n, kq);
  const bool mine = threadIdx.x < n * C;
  const int i = threadIdx.x / C, c = threadIdx.x - i * C, f = f0 + c;
  const size_t nf = (size_t)i * F + f;
  const size_t b = (size_t)i * 3 * 2 * F + f;
  float gu[3] = {0.f, 0.f, 0.f}, gvv[3] = {0.f, 0.f, 0.f}, vv[3] = {0.f, 0.f, 0.f}, res = 0.f, nrm = 1.f;
  if (mine) {
#pragma unroll
    for (int xyz = 0; xyz < 3; ++xyz) {
      const size_t at = b + (size_t)xyz * 2 * F;
      gu[xyz] = ldg_pinned(gUV_in + at); gvv[xyz] = ldg_pinned(gUV_in + at + F); vv[xyz] = ldg_pinned(UV + at + F);
    }
    res = ldg_pinned(gs_res + nf);
    nrm = ldg_pinned(stack + (size_t)i * 2 * F + F + f);
  }
  const TileRange tr = dl_tile_range(F);
  BiRegs<G, NT> wr;
  bi_prefetch<G, NT>(wr, Wuv, F, row0, tr);
  pin_loads();
  DL_PH(7, 1);
  for (int o = threadIdx.x; o < 48 * G * 4; o += DL_THREADS) g_l[o] = 0.f;
  quad_finish<1, G>(qr, gsum_l, scratch, gs_n, n, 7);
  DL_PH(7, 4);
  if (mine) {
    const int q = c >> 2, c4 = c & 3;
    if (blockIdx.y == 0) g_s2[nf] = reinterpret_cast<const float*>(gsum_l + q * 16)[i * 4 + c4] + res;   // S' also reaches S'' directly
    const float t = reinterpret_cast<const float*>(gsum_l + (CBQ + q) * 16)[i * 4 + c4] / nrm;
#pragma unroll
    for (int xyz = 0; xyz < 3; ++xyz) {
      const float tot = gvv[xyz] + t * vv[xyz];
      if (blockIdx.y == 0) {                                              // [gU | gVv total]: the weight-gradient launch's operand
        if (gUV_out != gUV_in_) gUV_out[b + (size_t)xyz * 2 * F] = gu[xyz];
        gUV_out[b + (size_t)xyz * 2 * F + F] = tot;
      }
      g_l[((3 * i + xyz) * G + q) * 4 + c4] = gu[xyz];
      g_l[((3 * i + xyz) * G + CBQ + q) * 4 + c4] = tot;
    }
  }
  __syncthreads();
  DL_PH(7, 6);
  bi_core<3, G, NT>(wr, g_l, stage, slices_out + (size_t)blockIdx.x * out_stride, F, 3 * n, tr, nullptr, nullptr, 7);
  DL_PH(7, 8);
  DL_SPAN(7, 1);
}

// ============================================================================================== B4: message backward + W2 rows
template <int R, int QS>
__global__ __launch_bounds__(DL_THREADS) void dec_msg_bwd_k(
    const float* __restrict__ phi_, const float* __restrict__ s_, const float* __restrict__ sbar_, const float* __restrict__ v_,
    const float* __restrict__ vbar_, const float* __restrict__ geom_d_, const int* __restrict__ rowptr_d_,
    const int* __restrict__ src_d_, const float* __restrict__ geom_s_, const int* __restrict__ rowptr_s_,
    const int* __restrict__ dst_s_, const float* __restrict__ Wd_, const float* __restrict__ bd_,
    const float* __restrict__ gh_, const float* __restrict__ ghb_, const float* __restrict__ gvrows_slices_, int gvr_n,
    long long gvr_stride, const float* __restrict__ gv_res_, const float* __restrict__ gvb_, const float* __restrict__ W2_,
    float* __restrict__ g_phi, float* __restrict__ g_s, float* __restrict__ g_sbar, float* __restrict__ g_v,
    float* __restrict__ g_vbar, float* __restrict__ gWd, float* __restrict__ gbd, float* __restrict__ slices_out,
    long long out_stride, int n, int F, int E) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  Carve cv;
  float* stage = cv.take(bi_stage_floats<1>());
  float4* scratch = reinterpret_cast<float4*>(stage);                    // the slice sum is over before the product stages its tiles
  static_assert(bi_stage_floats<1>() >= 36 * 16 * 3 * 4, "scratch aliases the stage");
  float4* gvr_l = reinterpret_cast<float4*>(cv.take(48 * 4));           // gV' rows [3 i + xyz][c]
  float* gphi_l = cv.take(16 * 9 * 4);
  float* phi_l = cv.take(16 * 9 * 4);
  float* red_src = cv.take(9 * 6 * 64);                                  // source-side (av, avb) of the lane's node, per wave
  float* red_rcv = cv.take(9 * 8 * 64);                                  // receiver-side partials (as, asb, av, avb)
  float* geomd_l = cv.take((size_t)DL_MAX_EDGES * GS);
  float* geoms_l = cv.take((size_t)DL_MAX_EDGES * GS);
  int* rpd_l = reinterpret_cast<int*>(cv.take(20)); int* rps_l = reinterpret_cast<int*>(cv.take(20));
  int* srcd_l = reinterpret_cast<int*>(cv.take(DL_MAX_EDGES)); int* dsts_l = reinterpret_cast<int*>(cv.take(DL_MAX_EDGES));
  float* s_l = cv.take(64); float* sb_l = cv.take(64); float* gh_l = cv.take(64); float* ghb_l = cv.take(64);
  float* v_l = cv.take(192); float* vb_l = cv.take(192); float* gv_l = cv.take(192); float* gvb_l = cv.take(192);
  float* gvres_l = cv.take(192);
  const int f0 = blockIdx.x * DL_CB;
  int row0[9];
#pragma unroll
  for (int g = 0; g < 9; ++g) row0[g] = g * F + f0;
  DL_PH(8, 0);
  DL_WV(0);
  DL_SPAN(8, 0);
  const gcf geom_d = launder(geom_d_); const gcf geom_s = launder(geom_s_); const gci rowptr_d = launder(rowptr_d_);
  const gci rowptr_s = launder(rowptr_s_); const gci src_d = launder(src_d_); const gci dst_s = launder(dst_s_);
  const gcf s = launder(s_); const gcf sbar = launder(sbar_); const gcf v = launder(v_);
  const gcf vbar = launder(vbar_); const gcf gh = launder(gh_); const gcf ghb = launder(ghb_);
  const gcf gvb = launder(gvb_); const gcf gv_res = launder(gv_res_); const gcf phi = launder(phi_);
  const gcf Wd = launder(Wd_); const gcf bd = launder(bd_); const gcf gvrows_slices = launder(gvrows_slices_);
  const gcf W2 = launder(W2_);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // filter of this wave.  The receiver-side pass has work for filters 0, 3, 4, 6, 7, 8 (and one line for 1); SIMD 0 hosts
  // three of the nine waves (0, 4, 8), so it gets the filters with nothing to do there (2, 5, 1) and every other SIMD one
  // cross-product filter and one light one: wave -> 2, 3, 7, 8, 5, 0, 4, 6, 1
#if CGV_DL_FILTER_PERM
  const int k = (int)((0x164058732ull >> (4 * wave)) & 15);
#else
  const int k = wave;
#endif
  const int node = lane >> 2, c = lane & 3, f = f0 + c;
  const bool live = node < n;
  const int nc = live ? node : 0;
  // order of requests, all in one batch: the bead graph / node state / upstream gradients of these channels (small,
  // stored to LDS as soon as they land), the slices, and -- once the staging registers are free -- the weights, which
  // have the two edge passes to arrive in.
  // Every request of a wave costs its SIMD the address arithmetic and the CU's one address unit 4 - 16 cycles whether the
  // lanes' data are wanted or not (nine waves x 21 staging requests each, most of them clamped duplicates, kept this
  // kernel's first five microseconds busy ISSUING): the small arrays are requested by ONE wave each -- a wave-uniform
  // role -- and a slot of the record copy only by the waves whose threads it reaches.
  const int n4 = E * GS / 4;
  Slots4<geom_slots(R)> r_gd, r_gs;
#pragma unroll
  for (int u = 0; u < geom_slots(R); ++u) {
    if (wave * 64 + u * DL_THREADS < n4) {                                // wave-uniform; the commit below is guarded alike
      const int at = min((int)threadIdx.x + u * DL_THREADS, n4 - 1);
      r_gd.v[u] = ldg4_pinned(geom_d + 4 * (size_t)at);
      r_gs.v[u] = ldg4_pinned(geom_s + 4 * (size_t)at);
    }
  }
  // Roles, branch-free (a branch per role makes the compiler wait for loads of OTHER roles' registers at the joins): every
  // wave issues ONE float4 request and ONE index request; what differs per wave are uniform parameters.
  //   float4:  wave 0 [s | sbar | gh | ghb] (16 lanes each), 1..4 v, vbar, gvb, gv_res (3 n lanes each), 5..7 phi (9 n
  //            float4 in chunks of 64); element = array[m * A + part * B + C] with (m, part) = divmod(item, D)
  //   index:   waves 0..3 src_d, 4..7 dst_s (64 ids each), 8 rowptr_d (lanes 0..31) | rowptr_s (lanes 32..63)
  const int quarter = wave == 0 ? lane >> 4 : 0;
  const gcf cand = wave == 0 ? (quarter == 0 ? s : quarter == 1 ? sbar : quarter == 2 ? gh : ghb)
                 : wave == 1 ? v : wave == 2 ? vbar : wave == 3 ? gvb : wave == 4 ? gv_res : phi;
  const bool small_have = cand != nullptr;                                // an absent array (NULL) is committed as zeros
  const int sm_D = wave == 0 ? 1 : wave <= 4 ? 3 : 9;
  const int sm_cnt = wave == 0 ? n : wave <= 4 ? 3 * n : n * 9;
  const int sm_A = wave == 0 ? F : wave <= 4 ? 3 * F : 9 * F, sm_B = wave == 0 ? 0 : wave <= 4 ? 4 : F;
  const int sm_C = wave <= 4 && wave >= 1 ? 3 * f0 : f0;
  const int sm_item = wave == 0 ? (lane & 15) : wave <= 4 ? lane : (wave - 5) * 64 + lane;
  float4 r_small;
  {
    const int it = min(sm_item, sm_cnt - 1);
    const int m = (it * (sm_D == 1 ? 65536 : sm_D == 3 ? 21846 : 7282)) >> 16, part = it - m * sm_D;     // it < 144
    r_small = ldg4_pinned((small_have ? cand : (wave >= 1 && wave <= 4 ? v : s)) + ((size_t)m * sm_A + part * sm_B + sm_C));   // absent: any valid address of that shape
  }
  const int id_item = wave < 8 ? (wave & 3) * 64 + lane : (lane & 31), id_cnt = wave < 8 ? E : n + 1;
  const gci id_src = wave < 4 ? src_d : wave < 8 ? dst_s : (lane < 32 ? rowptr_d : rowptr_s);
  const int r_ids = ldgi_pinned(id_src + min(id_item, id_cnt - 1));
  float W[R + 1], G[R + 1];
#pragma unroll
  for (int nn = 0; nn < R; ++nn) W[nn] = ldg_pinned(Wd + ((size_t)k * F + f) * R + nn);
  W[R] = ldg_pinned(bd + (size_t)k * F + f);
#pragma unroll
  for (int nn = 0; nn <= R; ++nn) G[nn] = 0.f;
  QuadRegs<3, 1, QS> qr;
  { const int kq[1] = {(int)blockIdx.x}; quad_issue<3, 1>(qr, gvrows_slices, gvr_n, gvr_stride, 3 * n, kq); }
  pin_loads();
  DL_WV(1);
#pragma unroll
  for (int u = 0; u < geom_slots(R); ++u)
    if ((int)threadIdx.x + u * DL_THREADS < n4) {
      reinterpret_cast<float4*>(geomd_l)[threadIdx.x + u * DL_THREADS] = r_gd.v[u];
      reinterpret_cast<float4*>(geoms_l)[threadIdx.x + u * DL_THREADS] = r_gs.v[u];
    }
  {
    float* dst = wave == 0 ? (quarter == 0 ? s_l : quarter == 1 ? sb_l : quarter == 2 ? gh_l : ghb_l)
               : wave == 1 ? v_l : wave == 2 ? vb_l : wave == 3 ? gvb_l : wave == 4 ? gvres_l : phi_l;
    if (sm_item < sm_cnt && wave <= 7) reinterpret_cast<float4*>(dst)[sm_item] = small_have ? r_small : make_float4(0.f, 0.f, 0.f, 0.f);
    int* idst = wave < 4 ? srcd_l : wave < 8 ? dsts_l : (lane < 32 ? rpd_l : rps_l);
    if (id_item < id_cnt) idst[id_item] = r_ids;
  }
  DL_PH(8, 1);
  DL_WV(2);
  BiRegs<9, 2> wr;                                                       // slot 1: the late tile, requested below
  const TileRange tr{0, (F + 63) / 64};
  bi_prefetch_slot<0>(wr, W2, F, row0, dl_tile_of(tr, wave, 0));
  pin_loads();
  // gV' = sum of the slices of B3 (rows 3 i + xyz) + the residual path V' -> V''
  quad_finish<3, 1>(qr, gvr_l, scratch, gvr_n, 3 * n, 8);
  if (threadIdx.x < 64) {
    const float* gr = reinterpret_cast<const float*>(gvr_l);
#pragma unroll
    for (int xyz = 0; xyz < 3; ++xyz)
      gv_l[(node * 4 + c) * 3 + xyz] = live ? gr[(3 * node + xyz) * 4 + c] + gvres_l[(node * 4 + c) * 3 + xyz] : 0.f;
  }
  __syncthreads();
  DL_PH(8, 4);
  // the 10th.. column tiles (K = 600: 24 columns, wave 0 only): requested now, used after the passes
  if (dl_tile_of(tr, wave, 1) < tr.end) {                                 // wave-uniform
    bi_prefetch_slot<1>(wr, W2, F, row0, dl_tile_of(tr, wave, 1));
  } else {
#pragma unroll
    for (int g = 0; g < 9; ++g) wr.w[1][g] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  pin_loads();
  const float p_j = phi_l[(nc * 9 + k) * 4 + c];
  const float s_n = s_l[nc * 4 + c], sb_n = sb_l[nc * 4 + c];
  const dv3 v_n = lds_v3(v_l + (nc * 4 + c) * 3), vb_n = lds_v3(vb_l + (nc * 4 + c) * 3);
  const float gh_n = gh_l[nc * 4 + c], ghb_n = ghb_l[nc * 4 + c];
  const dv3 gvb_n = lds_v3(gvb_l + (nc * 4 + c) * 3), gv_n = lds_v3(gv_l + (nc * 4 + c) * 3);
  (void)s_n;
  // ---- pass B: the lane's node as source j, edges of the src-sorted view (pseudo_msg.hip: pseudo_bwd_src_k)
  float a = 0.f;
  dv3 av{0.f, 0.f, 0.f}, avb{0.f, 0.f, 0.f};
  {
    // One LDS round trip per edge: the receiver id of the NEXT edge is requested with this edge's operands, and every
    // filter reads the same set of arrays through wave-uniform base pointers (a load inside the filter switch would be a
    // second dependent trip: index -> operand -> switch -> operand was four of them, ~0.3 us per edge with nine waves).
    DL_WV(4);
    const float* __restrict__ pA = k >= 5 ? gvb_l : gv_l;                 // the gradient the filter's term carries
    const float* __restrict__ pB = k == 8 ? vb_l : v_l;                   // the receiver-side state in its cross product (k = 0: v_i of the filter-free term)
    const float* __restrict__ pS = k == 0 ? gh_l : sb_l;
    const int e_beg = live ? rps_l[node] : 0, e_end = live ? min(rps_l[node + 1], E) : 0;
    int i_nx = e_beg < e_end ? dsts_l[e_beg] : 0;
    for (int e = e_beg; e < e_end; ++e) {
      const float* __restrict__ g = geoms_l + (size_t)e * GS;
      const int ic = i_nx * 4 + c;
      i_nx = dsts_l[min(e + 1, e_end - 1)];
      const dv3 gA = lds_v3(pA + ic * 3), sB = lds_v3(pB + ic * 3);
      const float sc = pS[ic], s_i = s_l[ic], ghb_i = ghb_l[ic];
      float gr[R + 1];
#pragma unroll
      for (int nn = 0; nn <= R; ++nn) gr[nn] = g[nn];
      const dv3 unit{g[U], g[U + 1], g[U + 2]};
      const dv3 zero{0.f, 0.f, 0.f};
      float gq = 0.f;
      dv3 cav = zero, cavb = zero;
      switch (k) {                                   // wave-uniform; registers only
        case 0: gq = sc * s_i; break;
        case 1: gq = ddot(gA, unit); break;
        case 2: gq = ddot(gA, v_n); cav = gA; break;
        case 3: gq = ddot(gA, dcross(sB, vb_n)); cavb = dcross(gA, sB); break;
        case 4: gq = sc * ddot(gA, vb_n); cavb = dv3{sc * gA.x, sc * gA.y, sc * gA.z}; break;
        case 5: gq = ddot(gA, vb_n); cavb = gA; break;
        case 6: gq = sc * ddot(gA, v_n); cav = dv3{sc * gA.x, sc * gA.y, sc * gA.z}; break;
        case 7: gq = ddot(gA, dcross(sB, v_n)); cav = dcross(gA, sB); break;
        default: gq = ddot(gA, dcross(sB, vb_n)); cavb = dcross(gA, sB); break;
      }
      const float w = dfilt<R>(W, gr);
      a = fmaf(gq, w, a);
      const float t = gq * p_j;
#pragma unroll
      for (int nn = 0; nn <= R; ++nn) G[nn] = fmaf(t, gr[nn], G[nn]);
      const float q = p_j * w;
      daxpy(av, q, cav);
      daxpy(avb, q, cavb);
      if (k == 0) daxpy(avb, ghb_i, sB);                                  // the filter-free term ghb_i v_i
    }
  }
  DL_PH(8, 5);
  DL_WV(5);
  gphi_l[(node * 9 + k) * 4 + c] = live ? a : 0.f;
  {
    float* r = red_src + (size_t)k * 6 * 64 + lane;
    r[0] = av.x; r[64] = av.y; r[128] = av.z; r[192] = avb.x; r[256] = avb.y; r[320] = avb.z;
  }
  // filter gradients: sum over the source nodes (lanes node*4 + c, fixed butterfly order), written once per block
#pragma unroll
  for (int nn = 0; nn <= R; ++nn) {
    float x = G[nn];
    x = add_xor32(add_xor16(add_row_stride4(x)));
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
    const float* __restrict__ pV = (k == 6 || k == 7) ? v_l : vb_l;       // (one LDS round trip per edge, as in pass B)
    const int e_beg = live ? rpd_l[node] : 0, e_end = live ? min(rpd_l[node + 1], E) : 0;
    int j_nx = e_beg < e_end ? srcd_l[e_beg] : 0;
    for (int e = e_beg; e < e_end; ++e) {
      const int j = j_nx;
      j_nx = srcd_l[min(e + 1, e_end - 1)];
      const dv3 vj = lds_v3(pV + (j * 4 + c) * 3);
      if (k == 1) { daxpy(rv, ghb_n, vj); continue; }
      const float* __restrict__ g = geomd_l + (size_t)e * GS;
      const float q = phi_l[(j * 9 + k) * 4 + c] * dfilt<R>(W, g);
      switch (k) {                                   // registers only
        case 0: as = fmaf(gh_n, q, as); break;
        case 3: daxpy(rv, q, dcross(vj, gv_n)); break;
        case 4: asb = fmaf(q, ddot(gv_n, vj), asb); break;
        case 6: asb = fmaf(q, ddot(gvb_n, vj), asb); break;
        case 7: daxpy(rv, q, dcross(vj, gvb_n)); break;
        default: daxpy(rvb, q, dcross(vj, gvb_n)); break;
      }
    }
  }
  {
    float* r = red_rcv + (size_t)k * 8 * 64 + lane;
    r[0] = as; r[64] = asb; r[128] = rv.x; r[192] = rv.y; r[256] = rv.z; r[320] = rvb.x; r[384] = rvb.y; r[448] = rvb.z;
  }
  DL_PH(8, 6);
  DL_WV(6);
  __syncthreads();
  DL_PH(8, 9);
  // g_phi: dense copy for the weight-gradient launch
  for (int o = threadIdx.x; o < n * 9; o += DL_THREADS) {
    const int m = o / 9, g = o - m * 9;
    *reinterpret_cast<float4*>(g_phi + (size_t)m * 9 * F + (size_t)g * F + f0) = *reinterpret_cast<const float4*>(gphi_l + (m * 9 + g) * 4);
  }
  if (k < 8 && live) {
    // wave k sums output component k of (g_s, g_sbar, g_v.xyz, g_vbar.xyz): the residual pass-through (outputs were
    // state + delta), the receiver-side partials in wave order, then the source-side partials in wave order
    float t = k == 0 ? gh_n : k == 1 ? ghb_n : k == 2 ? gv_n.x : k == 3 ? gv_n.y : k == 4 ? gv_n.z
            : k == 5 ? gvb_n.x : k == 6 ? gvb_n.y : gvb_n.z;
#pragma unroll
    for (int w = 0; w < 9; ++w) t += red_rcv[((size_t)w * 8 + k) * 64 + lane];
    if (k >= 2) {
#pragma unroll
      for (int w = 0; w < 9; ++w) t += red_src[((size_t)w * 6 + (k - 2)) * 64 + lane];
    }
    const size_t jf = (size_t)node * F + f;
    if (k == 0) g_s[jf] = t;
    else if (k == 1) g_sbar[jf] = t;
    else if (k < 5) g_v[jf * 3 + (k - 2)] = t;
    else g_vbar[jf * 3 + (k - 5)] = t;
  }
  bi_core<1, 9, 2>(wr, gphi_l, stage, slices_out + (size_t)blockIdx.x * out_stride, F, n, tr, nullptr, nullptr, 8);
  DL_WV(7);
  DL_PH(8, 8);
  DL_SPAN(8, 1);
}

// out[m][4 kq ..] = base + sum_s slices[s][kq][m][:]: the decoder input's gradient leaves the slice format here
template <int QS>
__global__ __launch_bounds__(DL_THREADS) void dec_quad_to_dense_k(const float* __restrict__ base, const float* __restrict__ slices,
                                                                  int n_slices, long long stride, float* __restrict__ out, int n,
                                                                  int F) {
  Carve cv;
  float4* scratch = reinterpret_cast<float4*>(cv.take(36 * 16 * 4));
  float4* sum_l = reinterpret_cast<float4*>(cv.take(16 * 4));
  const int kq[1] = {(int)blockIdx.x};
  QuadRegs<1, 1, QS> qr;
  quad_issue<1, 1>(qr, launder(slices), n_slices, stride, n, kq);
  quad_finish<1, 1>(qr, sum_l, scratch, n_slices, n);
  if (threadIdx.x < n) {
    float4 t = sum_l[threadIdx.x];
    if (base) {
      const float4 b = *reinterpret_cast<const float4*>(base + (size_t)threadIdx.x * F + 4 * blockIdx.x);
      t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)threadIdx.x * F + 4 * blockIdx.x) = t;
  }
}

// ============================================================================================== prior / small-graph EquiMessageBlock
// The same channel-group scheme for an EquiMessageBlock layer on a small bead graph (CGprior.forward, cgvae.py:391-392:
// h += ds, v += dv with (ds, dv) = EquiMessageBlock(h, v), conv.py:505-563) -- 12 nodes / 60 edges on chignolin, where the
// per-block path spends 3 + 5 launches per layer on six dependent round trips each:
//   forward   P1  a1 = swish(h W1^T + b1)                                    (dec_dense_fwd_k)
//             P2  phi = a1 W2^T + b2 (3 x CB rows) -> message on the block's channels -> h' = h + ds, v' = v + dv
//   backward  Q1  g_h' = sum of the slices from the layer above (+ base) ; the SCALAR path of the message backward (the
//                 vector channel of the prior / encoder never reaches an output, cgvae.py:393-396: no gradient arrives for
//                 v', so g_q0 = g_q2 = 0 and g_v = 0) ; slice = g_phi1 W2[F + rows]
//             Q2  g_a1 = sum of Q1's slices ; slice = (g_a1 swish'(z1)) W1[rows]          (dec_dense_bwd_k)
// Filter slices: q0 = phi[:, f] (v_j term), q1 = phi[:, F + f] (ds), q2 = phi[:, 2F + f] (unit term).
template <int R>
__global__ __launch_bounds__(DL_THREADS) void prior_msg_fwd_k(
    const float* __restrict__ a1, const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ s_,
    const float* __restrict__ v_, const float* __restrict__ geom_, const int* __restrict__ rowptr_, const int* __restrict__ src_,
    const float* __restrict__ Wd_, const float* __restrict__ bd_, float* __restrict__ phi_out, float* __restrict__ s_out,
    float* __restrict__ v_out, int n, int F, int E, int with_dv) {
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  Carve cv;
  float* red = cv.take(fwd_red_floats<1, 3>());
  float* phi_l = cv.take(16 * 3 * 4);
  float* red2 = cv.take(2 * 3 * 64);
  float* geom_l = cv.take((size_t)DL_MAX_EDGES * GS);
  int* rp_l = reinterpret_cast<int*>(cv.take(20));
  int* src_l = reinterpret_cast<int*>(cv.take(DL_MAX_EDGES));
  float* s_l = cv.take(64);
  float* v_l = cv.take(192);
  const int f0 = blockIdx.x * DL_CB;
  const int lane = threadIdx.x & 63;
  const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kf = k < 3 ? k : 0;                                          // waves 3..8 only take part in the product
  const int i = lane >> 2, c = lane & 3;
  const bool live = i < n;
  const int f = f0 + c;
  const int row0[3] = {f0, F + f0, 2 * F + f0};
  const gcf geom = launder(geom_); const gci rowptr = launder(rowptr_); const gci src = launder(src_);
  const gcf s = launder(s_); const gcf v = launder(v_); const gcf Wd = launder(Wd_); const gcf bd = launder(bd_);
  Slots4<geom_slots(R)> r_geom;
  copy4_issue(r_geom, geom, E * GS / 4);
  const int r_rp = int_issue(rowptr, n + 1), r_src = int_issue(src, E);
  const float4 r_s = scalar_issue(s, s, n, F, f0);
  const float4 r_v = vector_issue(v, v, n, F, f0);
  float W[R + 1];
#pragma unroll
  for (int nn = 0; nn < R; ++nn) W[nn] = ldg_pinned(Wd + ((size_t)kf * F + f) * R + nn);
  W[R] = ldg_pinned(bd + (size_t)kf * F + f);
  pin_loads();
  auto commit = [&]() {
    copy4_commit(r_geom, geom_l, E * GS / 4);
    int_commit(r_rp, rp_l, n + 1); int_commit(r_src, src_l, E);
    scalar_commit(r_s, true, s_l, n);
    vector_commit(r_v, true, v_l, n);
  };
  fwd_core<1, 3, 5>(phi_l, red, a1, n, F, W2, row0, commit);
  for (int o = threadIdx.x; o < 16 * 3; o += DL_THREADS) {
    const int m = o / 3, g = o - m * 3;
    float4 p = *reinterpret_cast<float4*>(phi_l + (m * 3 + g) * 4);
    const float4 b = *reinterpret_cast<const float4*>(b2 + (size_t)g * F + f0);
    p.x += b.x; p.y += b.y; p.z += b.z; p.w += b.w;
    *reinterpret_cast<float4*>(phi_l + (m * 3 + g) * 4) = p;
    if (m < n) *reinterpret_cast<float4*>(phi_out + (size_t)m * 3 * F + (size_t)g * F + f0) = p;
  }
  __syncthreads();
  // EquiMessageBlock, wave k = filter k (k < 3), lane = (receiver i, channel c); edges in the plan's order (equi_msg.hip)
  float as = 0.f;
  dv3 acc{0.f, 0.f, 0.f};
  if (k < 3 && (k == 1 || with_dv)) {
    const int e_beg = live ? rp_l[i] : 0, e_end = live ? min(rp_l[i + 1], E) : 0;
    for (int e = e_beg; e < e_end; ++e) {
      const int j = src_l[e];
      const float* __restrict__ g = geom_l + (size_t)e * GS;
      const float q = phi_l[(j * 3 + k) * 4 + c] * dfilt<R>(W, g);
      if (k == 1) as += q;
      else if (k == 2) daxpy(acc, q, dv3{g[U], g[U + 1], g[U + 2]});
      else daxpy(acc, q, lds_v3(v_l + (j * 4 + c) * 3));
    }
  }
  if (k == 0 || k == 2) {
    float* r = red2 + (size_t)(k >> 1) * 3 * 64 + lane;
    r[0] = acc.x; r[64] = acc.y; r[128] = acc.z;
  }
  __syncthreads();
  if (k != 1 || !live) return;
  const size_t nf = (size_t)i * F + f;
  s_out[nf] = s_l[i * 4 + c] + as;
  const dv3 v_i = lds_v3(v_l + (i * 4 + c) * 3);
  // dv = (unit term: wave 2) + (v_j term: wave 0), the order of equi_msg.hip's per-edge sum is NOT reproduced (two partial
  // sums over the same edges instead of one interleaved sum: rounding-level difference)
  st3(v_out + nf * 3, v_i.x + (red2[3 * 64 + lane] + red2[lane]), v_i.y + (red2[4 * 64 + lane] + red2[64 + lane]),
      v_i.z + (red2[5 * 64 + lane] + red2[128 + lane]));
}

template <int R, int QS>
__global__ __launch_bounds__(DL_THREADS) void prior_msg_bwd_k(
    const float* __restrict__ phi_, const float* __restrict__ geom_s_, const int* __restrict__ rowptr_s_,
    const int* __restrict__ dst_s_, const float* __restrict__ Wd_, const float* __restrict__ bd_,
    const float* __restrict__ gh_base_, const float* __restrict__ gh_slices_, int gh_n, long long gh_stride,
    const float* __restrict__ W2_, float* __restrict__ g_phi, float* __restrict__ g_h, float* __restrict__ gWd,
    float* __restrict__ gbd, float* __restrict__ slices_out, long long out_stride, int n, int F, int E) {
  constexpr int GS = geom_stride(R);
  Carve cv;
  float* stage = cv.take(bi_stage_floats<1>());
  float4* scratch = reinterpret_cast<float4*>(stage);                    // the slice sum is over before the product stages its tiles
  float4* sum_l = reinterpret_cast<float4*>(cv.take(16 * 4));
  float* gphi_l = cv.take(16 * 4);
  float* gh_l = cv.take(64);
  float* p1_l = cv.take(64);
  float* geoms_l = cv.take((size_t)DL_MAX_EDGES * GS);
  int* rps_l = reinterpret_cast<int*>(cv.take(20));
  int* dsts_l = reinterpret_cast<int*>(cv.take(DL_MAX_EDGES));
  const int f0 = blockIdx.x * DL_CB;
  const int row0[1] = {F + f0};
  const gcf geom_s = launder(geom_s_); const gci rowptr_s = launder(rowptr_s_); const gci dst_s = launder(dst_s_);
  const gcf phi = launder(phi_); const gcf Wd = launder(Wd_); const gcf bd = launder(bd_);
  const gcf gh_base = launder(gh_base_); const gcf gh_slices = launder(gh_slices_); const gcf W2 = launder(W2_);
  const int lane = threadIdx.x & 63;
  const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int node = lane >> 2, c = lane & 3, f = f0 + c;
  const bool live = node < n;
  Slots4<geom_slots(R)> r_gs;
  copy4_issue(r_gs, geom_s, E * GS / 4);
  const int r_rps = int_issue(rowptr_s, n + 1), r_dsts = int_issue(dst_s, E);
  const float4 r_p1 = ldg4_pinned(phi + (size_t)min((int)threadIdx.x, n - 1) * 3 * F + F + f0);
  const float4 r_base = ldg4_pinned((gh_base ? gh_base : phi) + (size_t)min((int)threadIdx.x, n - 1) * (gh_base ? F : 3 * F) + f0);
  float W[R + 1], G[R + 1];
#pragma unroll
  for (int nn = 0; nn < R; ++nn) W[nn] = ldg_pinned(Wd + ((size_t)F + f) * R + nn);
  W[R] = ldg_pinned(bd + (size_t)F + f);
#pragma unroll
  for (int nn = 0; nn <= R; ++nn) G[nn] = 0.f;
  QuadRegs<1, 1, QS> qr;
  { const int kq[1] = {(int)blockIdx.x}; quad_issue<1, 1>(qr, gh_slices ? gh_slices : phi, gh_slices ? gh_n : 0, gh_stride, n, kq); }
  BiRegs<1, 2> wr;
  const TileRange tr{0, (F + 63) / 64};
  bi_prefetch<1, 2>(wr, W2, F, row0, tr);
  pin_loads();
  copy4_commit(r_gs, geoms_l, E * GS / 4);
  int_commit(r_rps, rps_l, n + 1); int_commit(r_dsts, dsts_l, E);
  if ((int)threadIdx.x < n) reinterpret_cast<float4*>(p1_l)[threadIdx.x] = r_p1;
  quad_finish<1, 1>(qr, sum_l, scratch, gh_slices ? gh_n : 0, n);        // (no slices: the tile comes out zero)
  if ((int)threadIdx.x < n) {
    float4 t = sum_l[threadIdx.x];
    if (gh_base) { t.x += r_base.x; t.y += r_base.y; t.z += r_base.z; t.w += r_base.w; }
    reinterpret_cast<float4*>(gh_l)[threadIdx.x] = t;
    // h' = h + ds: the gradient of h' reaches h directly too (the residual); dense, for the layer below
    *reinterpret_cast<float4*>(g_h + (size_t)threadIdx.x * F + f0) = t;
  }
  __syncthreads();
  // scalar path of the message backward, the lane's node as SOURCE j (edges of the source-sorted view): g_q1 = g_h'[i]
  if (k == 0) {
    float a = 0.f;
    const float p_j = live ? p1_l[node * 4 + c] : 0.f;
    const int e_beg = live ? rps_l[node] : 0, e_end = live ? min(rps_l[node + 1], E) : 0;
    for (int e = e_beg; e < e_end; ++e) {
      const float* __restrict__ g = geoms_l + (size_t)e * GS;
      const float gq = gh_l[dsts_l[e] * 4 + c];
      a = fmaf(gq, dfilt<R>(W, g), a);
      const float t = gq * p_j;
#pragma unroll
      for (int nn = 0; nn <= R; ++nn) G[nn] = fmaf(t, g[nn], G[nn]);
    }
    gphi_l[node * 4 + c] = live ? a : 0.f;
#pragma unroll
    for (int nn = 0; nn <= R; ++nn) {
      float x = G[nn];
      x = add_xor32(add_xor16(add_row_stride4(x)));
      G[nn] = x;
    }
    if (lane < 4) {
#pragma unroll
      for (int nn = 0; nn < R; ++nn) gWd[((size_t)F + f) * R + nn] = G[nn];
      gbd[(size_t)F + f] = G[R];
    }
  } else if (k == 1 || k == 2) {
    // the dead filter slices (q0, q2: no gradient reaches the vector channel) get explicit zeros: these parameters'
    // arena slices are written, not accumulated
    const int kk = k == 1 ? 0 : 2;
    if (lane < 4) {
#pragma unroll
      for (int nn = 0; nn < R; ++nn) gWd[((size_t)kk * F + f) * R + nn] = 0.f;
      gbd[(size_t)kk * F + f] = 0.f;
    }
  }
  __syncthreads();
  // g_phi: dense copy for the weight-gradient launch (zeros in the dead slices)
  for (int o = threadIdx.x; o < n * 3; o += DL_THREADS) {
    const int m = o / 3, g = o - m * 3;
    const float4 val = g == 1 ? *reinterpret_cast<const float4*>(gphi_l + m * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(g_phi + (size_t)m * 3 * F + (size_t)g * F + f0) = val;
  }
  bi_core<1, 1, 2>(wr, gphi_l, stage, slices_out + (size_t)blockIdx.x * out_stride, F, n, tr);
}

}  // namespace cgv

namespace cgv {
// dynamic LDS per kernel (floats, mirroring the Carve sequence of each kernel) + slack for the 16-byte rounding
static size_t lds_bytes(size_t floats) { return floats * 4 + 512; }
template <typename Kern>
static int allow_lds(Kern kern, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { cgv::set_error("hipFuncSetAttribute(%zu bytes of LDS): %s", bytes, hipGetErrorString(e)); return (int)e; }
  return 0;
}
}  // namespace cgv

extern "C" {

int cgv_decoder_layer_supported(int n_nodes, int n_feat, int n_rbf) {
  return n_nodes >= 1 && n_nodes <= cgv::DL_MAX_NODES && n_feat >= 16 && (n_feat % 4) == 0 && n_feat <= 864 &&
         cgv_rbf_supported(n_rbf);
}
int cgv_decoder_max_edges(void) { return cgv::DL_MAX_EDGES; }
/* channels (weight rows per row group set) a block of gate_bwd / dense_bwd / uv_bwd owns for a width: 8 when the width
 * is a multiple of 8 (half as many, twice as fat slices), else 4; their slice count is width / this. */
int cgv_decoder_block_channels(int width) { return (width % 8) == 0 && cgv::option(CGV_OPT_DECODER_FAT) != 0 ? 8 : 4; }
/* blocks (grid y) that share one channel group's backward-input product by column tiles: 2 when there are tiles to share */
static int cgv_decoder_column_parts(int K, bool heavy = false) {
  const int o = cgv::option(CGV_OPT_DECODER_COLSPLIT), tiles = (K + 63) / 64;
  const int want = o <= 0 ? 1 : o == 1 ? 2 : o == 2 ? (heavy ? 3 : 2) : 3;
  return want < tiles ? want : tiles;
}
/* measurement: every decoder kernel stores begin / end of its first and last block into buf[32 + 4 id ..] and block 0's
 * phase boundaries into buf[80 + 10 id + i] (ids and phases in decoder_layer.hip), per-wave ticks of msg_bwd into buf[192 + 8 wave + i].  buf: 272 uint64.  NULL: off */
int cgv_decoder_debug_clock(uint64_t* buf) {
#if !CGV_DL_CLOCK
  if (buf) { cgv::set_error("this build has no phase clock: rebuild csrc/decoder_layer.hip with -DCGV_DL_CLOCK=1"); return CGV_E_UNSUPPORTED; }
#endif
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(cgv::g_dl_clock), &p, sizeof(p));
  return e == hipSuccess ? 0 : (int)e;
}

/* floats of one slice of a phase's output: (K / 4) column quads x rows x 4 */
int64_t cgv_decoder_slice_floats(int K, int rows) { return (int64_t)K * rows; }


/* slices per lane class of a slice sum: 3 up to 108 slices, else 6 (up to 216) */
#define CGV_DL_QS(n_slices, ...)                                           \
  do {                                                                     \
    if ((n_slices) <= 108) { constexpr int QS = 3; __VA_ARGS__ }           \
    else { constexpr int QS = 6; __VA_ARGS__ }                             \
  } while (0)

#define CGV_DL_CHECK()                                                                                        \
  CGV_REQUIRE(cgv_decoder_layer_supported(n_nodes, n_feat, n_rbf), "unsupported shape (n <= 16 nodes, F % 4 == 0, 16 <= F <= 864)"); \
  hipStream_t st = (hipStream_t)stream;                                                                       \
  const int blocks = n_feat / cgv::DL_CB

int cgv_decoder_msg_fwd(const float* a1, const float* W2, const float* b2, const float* s, const float* sbar, const float* v,
                        const float* vbar, const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* Wd,
                        const float* bd, float* phi, float* stack, float* sbar_out, float* v_out, float* vbar_out,
                        float* rows_out, int n_nodes, int n_feat, int n_rbf, int n_edges, void* stream) {
  CGV_REQUIRE(a1 && W2 && b2 && s && sbar && v && vbar && geom_d && rowptr_d && src_d && Wd && bd, "null input");
  CGV_REQUIRE(phi && stack && sbar_out && v_out && vbar_out && rows_out, "null output");
  CGV_REQUIRE(n_edges >= 1 && n_edges <= cgv::DL_MAX_EDGES, "the staged bead graph holds 1..cgv_decoder_max_edges() edges");
  CGV_DL_CHECK();
  const size_t base_floats = cgv::fwd_red_floats<1, 9>() + 576 + 1536 + (size_t)cgv::DL_MAX_EDGES * cgv::geom_stride(n_rbf) + 20 +
                             cgv::DL_MAX_EDGES + 128 + 384;
  /* the block's 36 weight rows by LDS-DMA when they fit beside the rest (F <= 672 at n_rbf = 10) */
  const bool wlds = cgv::option(CGV_OPT_DECODER_WLDS) != 0 && cgv::lds_bytes(base_floats + (size_t)36 * n_feat) <= 160 * 1024;
  const size_t lds = cgv::lds_bytes(base_floats + (wlds ? (size_t)36 * n_feat : 0));
  CGV_DISPATCH_RBF(n_rbf, {
    if (wlds) {
      if (int rc = cgv::allow_lds(cgv::dec_msg_fwd_k<RBF, true>, lds)) return rc;
      hipLaunchKernelGGL((cgv::dec_msg_fwd_k<RBF, true>), dim3(blocks), dim3(cgv::DL_THREADS), lds, st, a1, W2, b2, s, sbar, v, vbar,
                         geom_d, rowptr_d, src_d, Wd, bd, phi, stack, sbar_out, v_out, vbar_out, rows_out, n_nodes, n_feat, n_edges);
    } else {
      if (int rc = cgv::allow_lds(cgv::dec_msg_fwd_k<RBF, false>, lds)) return rc;
      hipLaunchKernelGGL((cgv::dec_msg_fwd_k<RBF, false>), dim3(blocks), dim3(cgv::DL_THREADS), lds, st, a1, W2, b2, s, sbar, v, vbar,
                         geom_d, rowptr_d, src_d, Wd, bd, phi, stack, sbar_out, v_out, vbar_out, rows_out, n_nodes, n_feat, n_edges);
    }
  });
  return cgv::check_launch("cgv_decoder_msg_fwd");
}

int cgv_prior_msg_fwd(const float* a1, const float* W2, const float* b2, const float* s, const float* v, const float* geom_d,
                      const int32_t* rowptr_d, const int32_t* src_d, const float* Wd, const float* bd, float* phi, float* s_out,
                      float* v_out, int n_nodes, int n_feat, int n_rbf, int n_edges, int with_dv, void* stream) {
  CGV_REQUIRE(a1 && W2 && b2 && s && v && geom_d && rowptr_d && src_d && Wd && bd && phi && s_out && v_out, "null pointer");
  CGV_REQUIRE(n_edges >= 1 && n_edges <= cgv::DL_MAX_EDGES, "the staged bead graph holds 1..cgv_decoder_max_edges() edges");
  CGV_DL_CHECK();
  const size_t lds = cgv::lds_bytes(cgv::fwd_red_floats<1, 3>() + 192 + 384 + (size_t)cgv::DL_MAX_EDGES * cgv::geom_stride(n_rbf) + 20 +
                                    cgv::DL_MAX_EDGES + 64 + 192);
  CGV_DISPATCH_RBF(n_rbf, {
    if (int rc = cgv::allow_lds(cgv::prior_msg_fwd_k<RBF>, lds)) return rc;
    hipLaunchKernelGGL((cgv::prior_msg_fwd_k<RBF>), dim3(blocks), dim3(cgv::DL_THREADS), lds, st, a1, W2, b2, s, v, geom_d, rowptr_d,
                       src_d, Wd, bd, phi, s_out, v_out, n_nodes, n_feat, n_edges, with_dv);
  });
  return cgv::check_launch("cgv_prior_msg_fwd");
}

int cgv_prior_msg_bwd(const float* phi, const float* geom_s, const int32_t* rowptr_s, const int32_t* dst_s, const float* Wd,
                      const float* bd, const float* gh_base, const float* gh_slices, int gh_n_slices, int64_t gh_slice_floats,
                      const float* W2, float* g_phi, float* g_h, float* gWd, float* gbd, float* slices_out,
                      int64_t out_slice_floats, int n_nodes, int n_feat, int n_rbf, int n_edges, void* stream) {
  CGV_REQUIRE(phi && geom_s && rowptr_s && dst_s && Wd && bd && W2 && g_phi && g_h && gWd && gbd && slices_out, "null pointer");
  CGV_REQUIRE((gh_base || gh_slices) && (gh_slices ? gh_n_slices >= 1 && gh_n_slices <= 216 : true), "the upstream gradient: base and / or 1 .. 216 slices");
  CGV_REQUIRE(n_edges >= 1 && n_edges <= cgv::DL_MAX_EDGES, "the staged bead graph holds 1..cgv_decoder_max_edges() edges");
  CGV_DL_CHECK();
  const size_t lds = cgv::lds_bytes(cgv::bi_stage_floats<1>() + 64 + 64 + 64 + 64 + (size_t)cgv::DL_MAX_EDGES * cgv::geom_stride(n_rbf) +
                                    20 + cgv::DL_MAX_EDGES);
  CGV_DISPATCH_RBF(n_rbf, {
    CGV_DL_QS(gh_slices ? gh_n_slices : 1, {
      if (int rc = cgv::allow_lds(cgv::prior_msg_bwd_k<RBF, QS>, lds)) return rc;
      hipLaunchKernelGGL((cgv::prior_msg_bwd_k<RBF, QS>), dim3(blocks), dim3(cgv::DL_THREADS), lds, st, phi, geom_s, rowptr_s, dst_s,
                         Wd, bd, gh_base, gh_slices, gh_n_slices, (long long)gh_slice_floats, W2, g_phi, g_h, gWd, gbd, slices_out,
                         (long long)out_slice_floats, n_nodes, n_feat, n_edges);
    });
  });
  return cgv::check_launch("cgv_prior_msg_bwd");
}

int cgv_decoder_dense_fwd(const float* x, const float* W, const float* bias, float* y, float* z, int n_nodes, int N, int K,
                          int act, void* stream) {
  CGV_REQUIRE(x && W && y, "null pointer");
  CGV_REQUIRE(act >= 0 && act <= cgv::CGV_ACT_MAX, "unknown activation");
  CGV_REQUIRE(n_nodes >= 1 && n_nodes <= cgv::DL_MAX_NODES && (N % 4) == 0 && (K % 4) == 0 && N >= 4 && K >= 4, "unsupported shape");
  hipLaunchKernelGGL(cgv::dec_dense_fwd_k, dim3(N / cgv::DL_CB), dim3(cgv::DL_THREADS), cgv::lds_bytes(cgv::fwd_red_floats<1, 1>() + 64),
                     (hipStream_t)stream, x, W, bias, y, z, n_nodes, N, K, act);
  return cgv::check_launch("cgv_decoder_dense_fwd");
}

int cgv_pair_linear_fwd(const float* x0, const float* x1, const float* W0, const float* W1, const float* bias0,
                        const float* bias1, float* y0, float* y1, float* z0, float* z1, int act0, int act1, int n_rows, int N,
                        int K, void* stream) {
  CGV_REQUIRE(x0 && x1 && W0 && W1 && y0 && y1, "null pointer");
  CGV_REQUIRE(act0 >= 0 && act0 <= cgv::CGV_ACT_MAX && act1 >= 0 && act1 <= cgv::CGV_ACT_MAX, "unknown activation");
  CGV_REQUIRE(n_rows >= 1 && n_rows <= cgv::DL_MAX_NODES && (N % 4) == 0 && (K % 4) == 0 && N >= 4 && K >= 4, "unsupported shape");
  cgv::DensePair p{{x0, x1}, {W0, W1}, {bias0, bias1}, {y0, y1}, {z0, z1}, {act0, act1}};
  hipLaunchKernelGGL(cgv::dec_dense_fwd_pair_k<1>, dim3(N / cgv::DL_CB, 2), dim3(cgv::DL_THREADS),
                     cgv::lds_bytes(cgv::fwd_red_floats<1, 1>() + 64), (hipStream_t)stream, p, n_rows, N, K);
  return cgv::check_launch("cgv_pair_linear_fwd");
}

int cgv_multi_linear_max(void) { return cgv::DL_MULTI_MAX; }

int cgv_multi_linear_fwd(int n, const float* const* x, const float* const* W, const float* const* bias, float* const* y,
                         float* const* z, const int* act, int n_rows, int N, int K, void* stream) {
  CGV_REQUIRE(n >= 1 && n <= cgv::DL_MULTI_MAX && x && W && y && act, "1 .. 4 problems, non-null tables");
  CGV_REQUIRE(n_rows >= 1 && n_rows <= cgv::DL_MAX_NODES && (N % 4) == 0 && (K % 4) == 0 && N >= 4 && K >= 4, "unsupported shape");
  cgv::DensePair p{};
  for (int j = 0; j < n; ++j) {
    CGV_REQUIRE(x[j] && W[j] && y[j], "null pointer");
    CGV_REQUIRE(act[j] >= 0 && act[j] <= cgv::CGV_ACT_MAX && (act[j] == 0 || (z && z[j])), "activation needs its pre-activation buffer");
    p.x[j] = x[j]; p.W[j] = W[j]; p.bias[j] = bias ? bias[j] : nullptr; p.y[j] = y[j]; p.z[j] = z ? z[j] : nullptr; p.act[j] = act[j];
  }
  if ((N % 8) == 0 && (N / cgv::DL_CB) * n > 256)       /* more blocks than CUs: 8-column blocks, one round */
    hipLaunchKernelGGL(cgv::dec_dense_fwd_pair_k<2>, dim3(N / (2 * cgv::DL_CB), n), dim3(cgv::DL_THREADS),
                       cgv::lds_bytes(cgv::fwd_red_floats<1, 2>() + 128), (hipStream_t)stream, p, n_rows, N, K);
  else
    hipLaunchKernelGGL(cgv::dec_dense_fwd_pair_k<1>, dim3(N / cgv::DL_CB, n), dim3(cgv::DL_THREADS),
                       cgv::lds_bytes(cgv::fwd_red_floats<1, 1>() + 64), (hipStream_t)stream, p, n_rows, N, K);
  return cgv::check_launch("cgv_multi_linear_fwd");
}

int cgv_decoder_uv_fwd(const float* rows, const float* Wuv, float* UV, float* stack, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(rows && Wuv && UV && stack, "null pointer");
  const int n_rbf = 8;
  CGV_DL_CHECK();
  if ((n_feat % 8) == 0 && cgv::option(CGV_OPT_DECODER_NODESPLIT) != 0) {
    /* node groups of at most 5 nodes (15 rows), as equal as possible: 12 nodes -> 3 x 4, 16 -> 4 x 4 */
    const int parts = (n_nodes + 4) / 5, npp = (n_nodes + parts - 1) / parts;
    hipLaunchKernelGGL(cgv::dec_uv_fwd_nodes_k, dim3(n_feat / 8, (n_nodes + npp - 1) / npp), dim3(cgv::DL_THREADS),
                       cgv::lds_bytes(cgv::fwd_red_floats<1, 4>() + 256), st, rows, Wuv, UV, stack, n_nodes, n_feat, npp);
    return cgv::check_launch("cgv_decoder_uv_fwd");
  }
  hipLaunchKernelGGL(cgv::dec_uv_fwd_k, dim3(blocks), dim3(cgv::DL_THREADS), cgv::lds_bytes(cgv::fwd_red_floats<3, 2>() + 384), st, rows,
                     Wuv, UV, stack, n_nodes, n_feat);
  return cgv::check_launch("cgv_decoder_uv_fwd");
}

int cgv_decoder_gate_fwd(const float* a0, const float* W1p, const float* b1p, const float* UV, const float* stack,
                         const float* v2, float* a, float* s3, float* v3, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(a0 && W1p && b1p && UV && stack && v2 && a && s3 && v3, "null pointer");
  const int n_rbf = 8;
  CGV_DL_CHECK();
  hipLaunchKernelGGL(cgv::dec_gate_fwd_k, dim3(blocks), dim3(cgv::DL_THREADS), cgv::lds_bytes(cgv::fwd_red_floats<1, 3>() + 192), st, a0,
                     W1p, b1p, UV, stack, v2, a, s3, v3, n_nodes, n_feat);
  return cgv::check_launch("cgv_decoder_gate_fwd");
}

int cgv_update_rows_fused_supported(int n_rows, int n_feat) {
  return n_rows >= 1 && n_rows <= 96 && n_feat >= 16 && (n_feat % 8) == 0 && n_feat <= 864;
}

int cgv_update_gate_fwd_fused(const float* a0, const float* W1, const float* b1, const float* UV, const float* s_res,
                              const float* v_res, float* a, float* s_out, float* v_out, int n_rows, int n_feat, void* stream) {
  CGV_REQUIRE(a0 && W1 && b1 && UV && a && s_out && v_out, "null pointer");
  CGV_REQUIRE(cgv_update_rows_fused_supported(n_rows, n_feat), "unsupported shape (1..96 rows, F % 8 == 0, 16 <= F <= 864)");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(n_feat / cgv::DL_CB);
#define CGV_UG(MBV)                                                                                                           \
  {                                                                                                                          \
    const size_t lds = cgv::lds_bytes(cgv::fwd_red_floats<MBV, 3>() + 16 * MBV * 12);                                         \
    if (int rc = cgv::allow_lds(cgv::upd_gate_fwd_rows_k<MBV>, lds)) return rc;                                               \
    hipLaunchKernelGGL((cgv::upd_gate_fwd_rows_k<MBV>), grid, dim3(cgv::DL_THREADS), lds, st, a0, W1, b1, UV, s_res, v_res, a, \
                       s_out, v_out, n_rows, n_feat);                                                                        \
  }
  switch ((n_rows + 15) / 16) {
    case 1: CGV_UG(1) break; case 2: CGV_UG(2) break; case 3: CGV_UG(3) break;
    case 4: CGV_UG(4) break; case 5: CGV_UG(5) break; default: CGV_UG(6) break;
  }
#undef CGV_UG
  return cgv::check_launch("cgv_update_gate_fwd_fused");
}

int cgv_update_uv_norm_fwd_fused(const float* rows, const float* Wuv, const float* s, float* UV, float* stack, int n_nodes,
                                 int n_feat, void* stream) {
  CGV_REQUIRE(rows && Wuv && UV && stack, "null pointer");
  CGV_REQUIRE(cgv_update_rows_fused_supported(n_nodes, n_feat), "unsupported shape (1..96 nodes, F % 8 == 0, 16 <= F <= 864)");
  hipStream_t st = (hipStream_t)stream;
  /* node groups of at most 16 nodes (48 rows = three row blocks), as equal as possible */
  const int parts = (n_nodes + 15) / 16, npp = (n_nodes + parts - 1) / parts, mb = (3 * npp + 15) / 16;
  const dim3 grid(n_feat / 8, (n_nodes + npp - 1) / npp);
#define CGV_UU(MBV)                                                                                                           \
  {                                                                                                                          \
    const size_t lds = cgv::lds_bytes(cgv::fwd_red_floats<MBV, 4>() + 16 * MBV * 16);                                         \
    if (int rc = cgv::allow_lds(cgv::upd_uv_norm_fwd_rows_k<MBV>, lds)) return rc;                                            \
    hipLaunchKernelGGL((cgv::upd_uv_norm_fwd_rows_k<MBV>), grid, dim3(cgv::DL_THREADS), lds, st, rows, Wuv, s, UV, stack,     \
                       n_nodes, n_feat, npp);                                                                                \
  }
  switch (mb) { case 1: CGV_UU(1) break; case 2: CGV_UU(2) break; default: CGV_UU(3) break; }
#undef CGV_UU
  return cgv::check_launch("cgv_update_uv_norm_fwd_fused");
}

int cgv_decoder_gate_bwd(const float* UV, const float* a, const float* gs_base, const float* gs_slices, int gs_n_slices,
                         int64_t gs_slice_stride, const float* gv, const float* W1p, float* ga, float* gUV, float* gs_sum,
                         float* slices_out, int64_t out_slice_stride, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(UV && a && W1p && ga && gUV && gs_sum && slices_out, "null pointer");
  CGV_REQUIRE(gs_n_slices >= 0 && gs_n_slices <= 216 && out_slice_stride >= cgv_decoder_slice_floats(n_feat, n_nodes), "bad slices");
  const int n_rbf = 8;
  CGV_DL_CHECK();
  (void)blocks;
  const int parts = cgv_decoder_column_parts(n_feat);
  const bool one_round = (((n_feat + 63) / 64 + parts - 1) / parts) <= cgv::DL_WAVES;     /* tiles per block <= waves */
#define CGV_DL_GATE(NTV)                                                                                                        \
  CGV_DL_QS(gs_n_slices, {                                                                                                      \
    if (cgv_decoder_block_channels(n_feat) == 8)                                                                                \
      hipLaunchKernelGGL((cgv::dec_gate_bwd_k<2, QS, NTV>), dim3(n_feat / 8, parts), dim3(cgv::DL_THREADS),                     \
                         cgv::lds_bytes(cgv::bi_stage_floats<1>() + 2 * 2304 + 2 * 64 + 384), st, UV, a, gs_base, gs_slices,   \
                         gs_n_slices, (long long)gs_slice_stride, gv, W1p, ga, gUV, gs_sum, slices_out,                        \
                         (long long)out_slice_stride, n_nodes, n_feat);                                                        \
    else                                                                                                                        \
      hipLaunchKernelGGL((cgv::dec_gate_bwd_k<1, QS, NTV>), dim3(n_feat / 4, parts), dim3(cgv::DL_THREADS),                     \
                         cgv::lds_bytes(cgv::bi_stage_floats<1>() + 2304 + 64 + 192), st, UV, a, gs_base, gs_slices, gs_n_slices, \
                         (long long)gs_slice_stride, gv, W1p, ga, gUV, gs_sum, slices_out, (long long)out_slice_stride, n_nodes, \
                         n_feat);                                                                                               \
  })
  if (one_round) { CGV_DL_GATE(1); } else { CGV_DL_GATE(2); }
#undef CGV_DL_GATE
  return cgv::check_launch("cgv_decoder_gate_bwd");
}
int cgv_decoder_dense_bwd(const float* g_slices, int g_n_slices, int64_t g_slice_stride, const float* z, int act, const float* W,
                          float* g_dense, float* slices_out, int64_t out_slice_stride, int n_nodes, int N, int K, void* stream) {
  CGV_REQUIRE(g_slices && W && g_dense && slices_out && g_n_slices >= 1 && g_n_slices <= 216, "null pointer / slice count");
  CGV_REQUIRE(act == 0 || (act >= 1 && act <= cgv::CGV_ACT_MAX && z), "act != 0 needs the saved pre-activation z");
  CGV_REQUIRE(n_nodes >= 1 && n_nodes <= cgv::DL_MAX_NODES && (N % 4) == 0 && (K % 4) == 0 && K <= 64 * 27 && N >= 4, "unsupported shape");
  CGV_REQUIRE(out_slice_stride >= cgv_decoder_slice_floats(K, n_nodes), "bad slices");
  hipStream_t st = (hipStream_t)stream;
  const int parts = cgv_decoder_column_parts(K);
  const int tiles = ((K + 63) / 64 + parts - 1) / parts;                  /* per block */
  const bool fat = cgv_decoder_block_channels(N) == 8;
  const dim3 blocks(fat ? N / 8 : N / 4, parts);
  const size_t lds = cgv::lds_bytes(cgv::bi_stage_floats<1>() + 2 * 2304 + 2 * 64 + 2 * 64);
#define CGV_DL_DENSE(NTV)                                                                                                  \
  CGV_DL_QS(g_n_slices, {                                                                                                  \
    if (fat)                                                                                                               \
      hipLaunchKernelGGL((cgv::dec_dense_bwd_k<NTV, 2, QS>), blocks, dim3(cgv::DL_THREADS), lds, st, g_slices,             \
                         g_n_slices, (long long)g_slice_stride, z, act, W, g_dense, slices_out, (long long)out_slice_stride, \
                         n_nodes, N, K);                                                                                   \
    else                                                                                                                   \
      hipLaunchKernelGGL((cgv::dec_dense_bwd_k<NTV, 1, QS>), blocks, dim3(cgv::DL_THREADS), lds, st, g_slices,             \
                         g_n_slices, (long long)g_slice_stride, z, act, W, g_dense, slices_out, (long long)out_slice_stride, \
                         n_nodes, N, K);                                                                                   \
  })
  if (tiles <= 9) { CGV_DL_DENSE(1); } else if (tiles <= 18) { CGV_DL_DENSE(2); } else { CGV_DL_DENSE(3); }
#undef CGV_DL_DENSE
  return cgv::check_launch("cgv_decoder_dense_bwd");
}

int cgv_decoder_uv_bwd(const float* gstack_slices, int n_slices, int64_t slice_stride, const float* UV, const float* stack,
                       const float* gs_res, const float* Wuv, const float* gUV, float* gUV_out, float* g_s2, float* slices_out,
                       int64_t out_slice_stride, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(gstack_slices && UV && stack && gs_res && Wuv && gUV && gUV_out && g_s2 && slices_out && n_slices >= 1 && n_slices <= 216,
              "null pointer / slice count");
  const int parts = cgv_decoder_column_parts(n_feat, true);
  CGV_REQUIRE(parts == 1 || gUV_out != gUV, "gUV_out may alias gUV only with one block per channel group (CGV_OPT_DECODER_COLSPLIT = 0)");
  CGV_REQUIRE(out_slice_stride >= cgv_decoder_slice_floats(n_feat, 3 * n_nodes), "bad slices");
  const int n_rbf = 8;
  CGV_DL_CHECK();
  (void)blocks;
  const bool one_round = (((n_feat + 63) / 64 + parts - 1) / parts) <= cgv::DL_WAVES;     /* tiles per block <= waves */
#define CGV_DL_UV(NTV)                                                                                                          \
  CGV_DL_QS(n_slices, {                                                                                                         \
    if (cgv_decoder_block_channels(n_feat) == 8) {                                                                              \
      const size_t lds = cgv::lds_bytes(cgv::bi_stage_floats<3>() + 2 * 4608 + 2 * 128 + 2 * 384);                              \
      if (int rc = cgv::allow_lds(cgv::dec_uv_bwd_k<2, QS, NTV>, lds)) return rc;                                               \
      hipLaunchKernelGGL((cgv::dec_uv_bwd_k<2, QS, NTV>), dim3(n_feat / 8, parts), dim3(cgv::DL_THREADS), lds, st, gstack_slices, \
                         n_slices, (long long)slice_stride, UV, stack, gs_res, Wuv, gUV, gUV_out, g_s2, slices_out,             \
                         (long long)out_slice_stride, n_nodes, n_feat);                                                        \
    } else {                                                                                                                    \
      const size_t lds = cgv::lds_bytes(cgv::bi_stage_floats<3>() + 4608 + 128 + 384);                                          \
      if (int rc = cgv::allow_lds(cgv::dec_uv_bwd_k<1, QS, NTV>, lds)) return rc;                                               \
      hipLaunchKernelGGL((cgv::dec_uv_bwd_k<1, QS, NTV>), dim3(n_feat / 4, parts), dim3(cgv::DL_THREADS), lds, st, gstack_slices, \
                         n_slices, (long long)slice_stride, UV, stack, gs_res, Wuv, gUV, gUV_out, g_s2, slices_out,             \
                         (long long)out_slice_stride, n_nodes, n_feat);                                                        \
    }                                                                                                                           \
  })
  if (one_round) { CGV_DL_UV(1); } else { CGV_DL_UV(2); }
#undef CGV_DL_UV
  return cgv::check_launch("cgv_decoder_uv_bwd");
}

int cgv_decoder_msg_bwd(const float* phi, const float* s, const float* sbar, const float* v, const float* vbar,
                        const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* geom_s,
                        const int32_t* rowptr_s, const int32_t* dst_s, const float* Wd, const float* bd, const float* gh,
                        const float* ghb, const float* gvrows_slices, int n_slices, int64_t slice_stride, const float* gv_res,
                        const float* gvb, const float* W2, float* g_phi, float* g_s, float* g_sbar, float* g_v, float* g_vbar,
                        float* gWd, float* gbd, float* slices_out, int64_t out_slice_stride, int n_nodes, int n_feat, int n_rbf,
                        int n_edges, void* stream) {
  CGV_REQUIRE(phi && s && sbar && v && vbar && geom_d && rowptr_d && src_d && geom_s && rowptr_s && dst_s && Wd && bd && W2,
              "null input");
  CGV_REQUIRE(gvrows_slices && n_slices >= 1 && n_slices <= 216 && g_phi && g_s && g_sbar && g_v && g_vbar && gWd && gbd && slices_out,
              "null pointer / slice count");
  CGV_REQUIRE(out_slice_stride >= cgv_decoder_slice_floats(n_feat, n_nodes), "bad slices");
  CGV_REQUIRE(n_edges >= 1 && n_edges <= cgv::DL_MAX_EDGES, "the staged bead graph holds 1..cgv_decoder_max_edges() edges");
  CGV_DL_CHECK();
  const size_t lds = cgv::lds_bytes(cgv::bi_stage_floats<1>() + 192 + 576 + 576 + 3456 + 4608 +
                               2 * (size_t)cgv::DL_MAX_EDGES * cgv::geom_stride(n_rbf) + 40 + 2 * cgv::DL_MAX_EDGES + 256 + 960);
  CGV_DISPATCH_RBF(n_rbf, {
    CGV_DL_QS(n_slices, {
      if (int rc = cgv::allow_lds(cgv::dec_msg_bwd_k<RBF, QS>, lds)) return rc;
      hipLaunchKernelGGL((cgv::dec_msg_bwd_k<RBF, QS>), dim3(blocks), dim3(cgv::DL_THREADS), lds, st, phi, s, sbar, v, vbar, geom_d,
                         rowptr_d, src_d, geom_s, rowptr_s, dst_s, Wd, bd, gh, ghb, gvrows_slices, n_slices,
                         (long long)slice_stride, gv_res, gvb, W2, g_phi, g_s, g_sbar, g_v, g_vbar, gWd, gbd, slices_out,
                         (long long)out_slice_stride, n_nodes, n_feat, n_edges);
    });
  });
  return cgv::check_launch("cgv_decoder_msg_bwd");
}

int cgv_decoder_slices_to_dense(const float* base, const float* slices, int n_slices, int64_t slice_stride, float* out,
                                int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(slices && out && n_slices >= 1 && n_slices <= 216 && n_nodes >= 1 && n_nodes <= 16 && (n_feat % 4) == 0, "bad argument");
  CGV_DL_QS(n_slices, {
    hipLaunchKernelGGL((cgv::dec_quad_to_dense_k<QS>), dim3(n_feat / 4), dim3(cgv::DL_THREADS), cgv::lds_bytes(2304 + 64),
                       (hipStream_t)stream, base, slices, n_slices, (long long)slice_stride, out, n_nodes, n_feat);
  });
  return cgv::check_launch("cgv_decoder_slices_to_dense");
}

}  // extern "C"
