// Device helpers shared by the fused message kernels (equi_msg.hip, equi_msg_grp.hip): packed-fp32
// two-channel lanes, LDS-staged filter rows, buffer-descriptor row gathers.
#pragma once
#include "cgv_common.h"

namespace cgv {

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 ld2(const float* p) { return *reinterpret_cast<const f2*>(p); }
__device__ __forceinline__ void st2(float* p, f2 x) { *reinterpret_cast<f2*>(p) = x; }
__device__ __forceinline__ f2 lo2(f2 a) { return __builtin_shufflevector(a, a, 0, 0); }
__device__ __forceinline__ f2 hi2(f2 a) { return __builtin_shufflevector(a, a, 1, 1); }

// filter of two adjacent channels: W[n] = (Wd[c][n], Wd[c+1][n]), W[R] = (bd[c], bd[c+1])
template <int R>
__device__ __forceinline__ f2 filter2(const f2 (&W)[R + 1], const float* __restrict__ g) {
  f2 w = W[R] * splat(g[R]);
#pragma unroll
  for (int n = 0; n < R; ++n) w = fma2(W[n], splat(g[n]), w);
  return w;
}

template <int R>
__device__ __forceinline__ void load_filter_rows2(f2 (&W)[R + 1], const float* __restrict__ Wd,
                                                  const float* __restrict__ bd, int c, int c1) {
#pragma unroll
  for (int n = 0; n < R; ++n) W[n] = f2{Wd[(size_t)c * R + n], Wd[(size_t)c1 * R + n]};
  W[R] = f2{bd[c], bd[c1]};
}

// Filter rows of a block's 128-channel tile, staged through LDS.  Reading them straight from global
// memory in register layout (lane c takes Wd[c][n]: lanes 2R floats apart) costs ~40 cache lines per
// load instruction and 66 instructions per wave -- 2640 line requests per wave, which at 6720 waves was
// HALF of the kernel's time (the texture path was saturated by the prologue, not by the edge loop).
// Staged: the tile's rows are contiguous (128 R floats per slice) -> coalesced float4 loads once per
// block, then conflict-tolerant LDS reads.  Needs F and R even (16-byte aligned slices).
template <int R, int NSL>
__device__ __forceinline__ void stage_filter_tile(float* __restrict__ wt /*[NSL][128*R]*/, const float* __restrict__ Wd,
                                                  const int (&slice)[NSL], int F, int c0) {
  const int cw = min(128, F - c0);                 // valid channels of this tile
  const int n4 = cw * R / 4;                       // float4 per slice
#pragma unroll
  for (int k = 0; k < NSL; ++k) {
    const float4* g = reinterpret_cast<const float4*>(Wd + (size_t)(slice[k] * F + c0) * R);
    float4* d = reinterpret_cast<float4*>(wt + k * 128 * R);
    for (int t = threadIdx.x; t < n4; t += blockDim.x) d[t] = g[t];
  }
  __syncthreads();
}
template <int R>
__device__ __forceinline__ void read_filter_rows2(f2 (&W)[R + 1], const float* __restrict__ wt_slice,
                                                  const float* __restrict__ bd, int cl /*even, tile-local*/,
                                                  int c_global) {
#pragma unroll
  for (int n = 0; n < R; ++n) W[n] = f2{wt_slice[cl * R + n], wt_slice[(cl + 1) * R + n]};
  W[R] = ld2(bd + c_global);
}

// Lane -> channel pair.  F even: lanes own (c, c+1) with 8-byte vector accesses.  F odd: the last
// lane's second channel is a duplicate of its first (PAIR = false -> scalar memory accesses).
struct ChanPair {
  int c, c1;       // the two channels (c1 == c + 1, or == c when duplicated)
  bool live, live1;
};
__device__ __forceinline__ ChanPair chan_pair(int tile, int lane, int F) {
  ChanPair p;
  const int raw = tile * 128 + 2 * lane;
  p.live = raw < F;
  p.live1 = raw + 1 < F;
  const int last = F >= 2 ? ((F - 2) & ~1) : 0;              // clamp: idle lanes read valid rows, never store
  p.c = p.live ? raw : last;
  p.c1 = p.live ? (p.live1 ? raw + 1 : raw) : (F >= 2 ? last + 1 : 0);
  return p;
}

// "scalar base + 32-bit lane byte offset" loads: with a wave-uniform row pointer and a loop-invariant
// per-lane byte offset the compiler emits global_load ... v_off, s[base] (saddr form) -- no per-edge
// 64-bit vector address arithmetic (PMC before this: 29 SALU + ~10 address VALU instructions per edge)
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
// Row gathers go through a buffer descriptor (SRSRC): address = base + lane byte offset (VGPR, loop
// invariant) + row byte offset (SGPR, one s_mul per edge).  The generic pointer form costs 29 scalar
// + ~10 vector address instructions per edge (PMC: SQ_INSTS_SALU) -- the address math, not the FMAs,
// was what the SIMDs were issuing.  Rows are < 2 GiB apart by the host-side check in the launcher.
__device__ __forceinline__ rsrc_t make_rsrc(const float* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f2 ld2_buf(rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  return __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, voff_bytes, soff_bytes, 0));
}

// xyz triples of two neighbouring channels = 24 contiguous bytes per lane.  As three b64 loads every
// instruction walks the wave's whole 1536-byte span (12 lines) for a third of its bytes; two b96 loads
// walk it twice instead of three times (tools/probes/k2_probe.cpp: 59.5 -> 47.4 us for the whole kernel).
typedef float f3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void ldvec_buf(rsrc_t r, unsigned voff_bytes, unsigned soff_bytes, f2& A, f2& B, f2& C) {
  const f3v x = __builtin_bit_cast(f3v, __builtin_amdgcn_raw_buffer_load_b96(r, voff_bytes, soff_bytes, 0));
  const f3v y = __builtin_bit_cast(f3v, __builtin_amdgcn_raw_buffer_load_b96(r, voff_bytes + 12u, soff_bytes, 0));
  A = f2{x.x, x.y}; B = f2{x.z, y.x}; C = f2{y.y, y.z};
}

template <bool PAIR>
__device__ __forceinline__ f2 ldpair(const float* base, const ChanPair& cp) {
  if constexpr (PAIR) return ld2(base + cp.c);
  else return f2{base[cp.c], base[cp.c1]};
}
// xyz triples of the two channels as (x0,y0)(z0,x1)(y1,z1)
template <bool PAIR>
__device__ __forceinline__ void ldvec(const float* row /* [F,3] of one node */, const ChanPair& cp, f2& A, f2& B, f2& C) {
  if constexpr (PAIR) {
    const float* p = row + (size_t)cp.c * 3;
    A = ld2(p); B = ld2(p + 2); C = ld2(p + 4);
  } else {
    const float* p = row + (size_t)cp.c * 3;
    const float* q = row + (size_t)cp.c1 * 3;
    A = f2{p[0], p[1]}; B = f2{p[2], q[0]}; C = f2{q[1], q[2]};
  }
}
template <bool PAIR>
__device__ __forceinline__ void stvec(float* row, const ChanPair& cp, f2 A, f2 B, f2 C) {
  if constexpr (PAIR) {
    float* p = row + (size_t)cp.c * 3;
    st2(p, A); st2(p + 2, B); st2(p + 4, C);
  } else {
    float* p = row + (size_t)cp.c * 3;
    p[0] = A.x; p[1] = A.y; p[2] = B.x;
    if (cp.live1) { float* q = row + (size_t)cp.c1 * 3; q[0] = B.y; q[1] = C.x; q[2] = C.y; }
  }
}
template <bool PAIR>
__device__ __forceinline__ void stpair(float* base, const ChanPair& cp, f2 x) {
  if constexpr (PAIR) st2(base + cp.c, x);
  else { base[cp.c] = x.x; if (cp.live1) base[cp.c1] = x.y; }
}

// ---- the shared-source walk's per-wave state and edge math (equi_msg_grp.hip, equi_msg_bal.hip)
struct RowBuf { f2 p0, p1, p2, A, B, C; };
struct Acc { f2 s, A, B, C; };

__device__ __forceinline__ void gather_row(RowBuf& b, rsrc_t r_phi, rsrc_t r_v, unsigned oc, unsigned oF, unsigned ov,
                                           unsigned so) {
  b.p1 = ld2_buf(r_phi, oc + oF, so);
  b.p0 = ld2_buf(r_phi, oc, so);
  b.p2 = ld2_buf(r_phi, oc + 2u * oF, so);
  ldvec_buf(r_v, ov, so, b.A, b.B, b.C);
}

template <int R>
__device__ __forceinline__ void edge_math(const f2 (&W0)[R + 1], const f2 (&W1)[R + 1], const f2 (&W2)[R + 1],
                                          const float* __restrict__ gc /* group record: a_n, env, -, ux, uy, uz */,
                                          const RowBuf& b, Acc& a) {
  constexpr int U = geom_group_unit_offset(R);
  a.s = fma2(b.p1, filter2<R>(W1, gc), a.s);
  const f2 m0 = b.p0 * filter2<R>(W0, gc);
  const f2 m2 = b.p2 * filter2<R>(W2, gc);
  const f2 u01 = f2{gc[U], gc[U + 1]}, u20 = f2{gc[U + 2], gc[U]}, u12 = f2{gc[U + 1], gc[U + 2]};
  a.A = fma2(lo2(m2), u01, fma2(lo2(m0), b.A, a.A));
  a.B = fma2(m2, u20, fma2(m0, b.B, a.B));
  a.C = fma2(hi2(m2), u12, fma2(hi2(m0), b.C, a.C));
}

}  // namespace cgv
