// K2m: fused EquiMessageBlock forward with the FILTER on the matrix cores (reference conv.py:505-563, InvariantMessage
// 63-75, DistanceEmbed modules.py:192-197; same math as equi_msg.hip / equi_msg_grp.hip).
//
// The filter rebuild  w_k(e, c) = sum_n a_n(e) Wd[kF+c][n] + env(e) bd[kF+c]  is a dense [edges x (R+1)] x [(R+1) x 3F]
// product -- 33 of the 42 packed FMAs the VALU kernels spend per (edge, channel pair).  f32 MFMA runs at exactly the
// packed-VALU rate, so the f32-MFMA variant of round 2 (equi_msg.hip: equi_msg_fwd_mfma_k) could not win.  Here the
// product runs on the F16 matrix path (16 x the f32 rate) at f32-class accuracy: every operand is split into two f16
// values, x * S = hi + lo (11 + 11 mantissa bits, S a power of two that keeps lo out of the subnormals), and
//     a . w = a_hi w_hi + a_hi w_lo + a_lo w_hi + a_lo w_lo
// is TWO v_mfma_f32_16x16x32_f16 with the reduction axis holding [hi half | lo half] of w: A = [a_hi | a_hi], then
// A = [a_lo | a_lo], both against B = [w_hi | w_lo]; products are exact in the f32 accumulator.  Measured error of w
// against fp64: 3e-7 of max |w| (f32 FMA chain: 1e-7).  K = 16 >= R + 1 covers the whole contraction in one step.
//
// Layout: A rows = 16 consecutive edges of ONE receiver (destination-sorted view), B columns = channels.  The MFMA leaves
// lane (j = l & 15, q = l >> 4) with w of one channel for the edges 4q .. 4q + 3; the four MFMA tiles of a wave take the
// channel sets {4j + n} (n = 0..3), so that lane j owns the FOUR CONSECUTIVE channels 4j .. 4j + 3 of its 64-channel
// group: the gathers of phi / v for an edge are 16-byte loads, 256 contiguous bytes per 16 lanes (the round-2 variant
// read one dword per lane and tile: 4 x the load instructions, which is what it was bound by).  The per-edge math
// (3 products, 7 FMAs per channel) follows in the same lane without a shuffle; the four edge groups q meet through two
// xor-shuffles when the receiver is finished.  A wave keeps its B operands for `rpw` receivers in a row.
#include <stdlib.h>
#include "cgv_common.h"
#include "equi_msg_dev.h"

namespace cgv {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4m __attribute__((ext_vector_type(4)));

constexpr float M16_SA = 256.0f;            // scale of the edge-side operand (|a_n| <= ~4: far from f16's 65504)
constexpr float M16_SW = 64.0f;             // scale of the filter weights (|w| < 1000)
constexpr float M16_INV = 1.0f / (M16_SA * M16_SW);

__device__ __forceinline__ void split16(const float (&x)[8], float s, h8& hi, h8& lo) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float xs = x[t] * s;
    const _Float16 h = (_Float16)xs;
    hi[t] = h;
    lo[t] = (_Float16)(xs - (float)h);
  }
}

// grid = ceil(items / 4) blocks of 4 waves; item = (receiver chunk of rpw, channel group of 64), group fastest
template <int R, bool WITH_DV>
__global__ __launch_bounds__(256) void equi_msg_fwd_mfma16_k(
    const float* __restrict__ phi, const float* __restrict__ v, const float* __restrict__ geom, const int* __restrict__ rowptr,
    const int* __restrict__ src, const float* __restrict__ Wd, const float* __restrict__ bd, float* __restrict__ ds,
    float* __restrict__ dv, int F, int n_dst, int groups, int rpw, const float* __restrict__ s_res,
    const float* __restrict__ v_res) {
  static_assert(R + 1 <= 16, "one 16-wide reduction step holds the filter terms");
  constexpr int GS = geom_stride(R), U = geom_unit_offset(R);
  constexpr int NK = WITH_DV ? 3 : 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * 4 + wave;
  const int chunk = item / groups, g = item - chunk * groups;
  const int node0 = chunk * rpw;
  if (node0 >= n_dst) return;
  const int j = lane & 15, q = lane >> 4;
  const int c0 = 64 * g + 4 * j;                       // the lane's four channels c0 .. c0 + 3
  const bool live = c0 < F;                            // F % 4 == 0
  const int cc = live ? c0 : 0;                        // clamped for addresses

  // ---- B operands: B[kk][n] = [w_hi | w_lo] of filter slice k, channel c0 + n; lane (j, q): reduction entries
  //      8 (q & 1) .. + 7 of the hi (q < 2) or lo (q >= 2) half
  h8 B[NK][4];
  {
    const int kb = 8 * (q & 1);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const int k = WITH_DV ? kk : 1;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const size_t row = (size_t)k * F + cc + n;
        float x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int kx = kb + t;
          x[t] = (kx < R) ? Wd[row * R + (kx < R ? kx : 0)] : (kx == R ? bd[row] : 0.f);
          if (!live) x[t] = 0.f;
        }
        h8 hi, lo;
        split16(x, M16_SW, hi, lo);
        B[kk][n] = q < 2 ? hi : lo;
      }
    }
  }
  const unsigned row_bytes = 12u * (unsigned)F;
  const rsrc_t r_phi = make_rsrc(phi), r_v = make_rsrc(WITH_DV ? v : phi);
  const unsigned oc = 4u * (unsigned)cc, ov = 12u * (unsigned)cc, oF = 4u * (unsigned)F;
  auto ld4_buf = [](rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(f32x4m, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0u, 0));
  };

  const int node1 = min(node0 + rpw, n_dst);
  for (int node = node0; node < node1; ++node) {
    float as[4] = {0.f, 0.f, 0.f, 0.f};
    float ax[4] = {0.f, 0.f, 0.f, 0.f}, ay[4] = {0.f, 0.f, 0.f, 0.f}, az[4] = {0.f, 0.f, 0.f, 0.f};
    const int beg = rowptr[node], end = rowptr[node + 1];
    for (int t0 = beg; t0 < end; t0 += 16) {
      // A operand: lane (i = j, q) supplies record entries 8 (q & 1) .. + 7 of edge t0 + j (zero past the end)
      const int eA = t0 + j;
      float xa[8];
      {
        const float4* rec = reinterpret_cast<const float4*>(geom + (size_t)min(eA, end - 1) * GS + 8 * (q & 1));
        const float4 u0 = rec[0], u1 = rec[1];
        const bool ok = eA < end;
        xa[0] = ok ? u0.x : 0.f; xa[1] = ok ? u0.y : 0.f; xa[2] = ok ? u0.z : 0.f; xa[3] = ok ? u0.w : 0.f;
        xa[4] = ok ? u1.x : 0.f; xa[5] = ok ? u1.y : 0.f; xa[6] = ok ? u1.z : 0.f; xa[7] = ok ? u1.w : 0.f;
      }
      h8 a_hi, a_lo;
      split16(xa, M16_SA, a_hi, a_lo);
      // this lane's 4 edges t0 + 4 q + r: source rows and unit vectors (addresses clamped; w == 0 past the end)
      unsigned so[4];
      f3 un[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = min(t0 + 4 * q + r, end - 1);
        so[r] = (unsigned)src[e] * row_bytes;
        if constexpr (WITH_DV) un[r] = ld3(geom + (size_t)e * GS + U);
      }
      f32x4m D[NK][4];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const f32x4m zero = {0.f, 0.f, 0.f, 0.f};
          D[kk][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, B[kk][n], __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, B[kk][n], zero, 0, 0, 0), 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4m P1 = ld4_buf(r_phi, so[r] + oc + oF);
        if constexpr (WITH_DV) {
          const f32x4m P0 = ld4_buf(r_phi, so[r] + oc), P2 = ld4_buf(r_phi, so[r] + oc + 2u * oF);
          const f32x4m V0 = ld4_buf(r_v, so[r] + ov), V1 = ld4_buf(r_v, so[r] + ov + 16u), V2 = ld4_buf(r_v, so[r] + ov + 32u);
          const float vv[12] = {V0[0], V0[1], V0[2], V0[3], V1[0], V1[1], V1[2], V1[3], V2[0], V2[1], V2[2], V2[3]};
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            as[n] = fmaf(P1[n], D[1][n][r], as[n]);
            const float m0 = P0[n] * D[0][n][r], m2 = P2[n] * D[2][n][r];
            ax[n] = fmaf(m2, un[r].x, fmaf(m0, vv[3 * n], ax[n]));
            ay[n] = fmaf(m2, un[r].y, fmaf(m0, vv[3 * n + 1], ay[n]));
            az[n] = fmaf(m2, un[r].z, fmaf(m0, vv[3 * n + 2], az[n]));
          }
        } else {
#pragma unroll
          for (int n = 0; n < 4; ++n) as[n] = fmaf(P1[n], D[0][n][r], as[n]);
        }
      }
    }
    // the 4 edge groups (lanes l, l ^ 16, l ^ 32, l ^ 48) hold partial sums of the same channels: fixed-order butterfly
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      as[n] += __shfl_xor(as[n], 16); as[n] += __shfl_xor(as[n], 32);
      if constexpr (WITH_DV) {
        ax[n] += __shfl_xor(ax[n], 16); ax[n] += __shfl_xor(ax[n], 32);
        ay[n] += __shfl_xor(ay[n], 16); ay[n] += __shfl_xor(ay[n], 32);
        az[n] += __shfl_xor(az[n], 16); az[n] += __shfl_xor(az[n], 32);
      }
    }
    if (q == 0 && live) {
      const size_t o = (size_t)node * F + c0;
      float4 s4 = make_float4(as[0] * M16_INV, as[1] * M16_INV, as[2] * M16_INV, as[3] * M16_INV);
      if (s_res) { const float4 t = *reinterpret_cast<const float4*>(s_res + o); s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w; }
      *reinterpret_cast<float4*>(ds + o) = s4;
      if constexpr (WITH_DV) {
        float out[12];
#pragma unroll
        for (int n = 0; n < 4; ++n) { out[3 * n] = ax[n] * M16_INV; out[3 * n + 1] = ay[n] * M16_INV; out[3 * n + 2] = az[n] * M16_INV; }
        if (v_res) {
          const float4* t4 = reinterpret_cast<const float4*>(v_res + o * 3);
#pragma unroll
          for (int u = 0; u < 3; ++u) { const float4 t = t4[u]; out[4 * u] += t.x; out[4 * u + 1] += t.y; out[4 * u + 2] += t.z; out[4 * u + 3] += t.w; }
        }
        float4* d4 = reinterpret_cast<float4*>(dv + o * 3);
        d4[0] = make_float4(out[0], out[1], out[2], out[3]);
        d4[1] = make_float4(out[4], out[5], out[6], out[7]);
        d4[2] = make_float4(out[8], out[9], out[10], out[11]);
      }
    }
  }
}

}  // namespace cgv

extern "C" {

int cgv_equi_msg_mfma16_supported(int n_feat, int n_rbf) {
  return (n_feat % 4) == 0 && n_feat >= 4 && n_rbf + 1 <= 16 && cgv_rbf_supported(n_rbf) && (n_rbf % 2) == 0;
}

/* cgv_equi_msg_fwd with the distance filter evaluated on the f16 matrix path through hi / lo operand splits (f32-class
 * accuracy: csrc/equi_msg_mfma16.hip).  Needs n_feat % 4 == 0, n_rbf <= 15, 16-byte aligned operands, all n_rows rows of
 * phi / v within 2 GiB.  rpw: receivers a wave processes with one set of filter operands (>= 1). */
int cgv_equi_msg_fwd_mfma16(const float* phi, const float* v, const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d,
                            const float* Wd, const float* bd, float* ds, float* dv, int n_dst, int n_feat, int n_rbf,
                            int with_dv, int rpw, int64_t n_rows, const float* s_res, const float* v_res, void* stream) {
  CGV_REQUIRE(n_dst >= 0 && n_feat > 0, "bad size");
  if (n_dst == 0) return 0;
  CGV_REQUIRE(phi && geom_d && rowptr_d && src_d && Wd && bd && ds, "null pointer");
  CGV_REQUIRE(!with_dv || (v && dv), "with_dv needs v and dv");
  CGV_REQUIRE(cgv_equi_msg_mfma16_supported(n_feat, n_rbf), "unsupported shape (need n_feat % 4 == 0, even n_rbf <= 14)");
  CGV_REQUIRE(n_rows > 0 && (uint64_t)n_rows * 12u * (uint64_t)n_feat < 0x7fffffffull, "rows must lie within 2 GiB");
  CGV_REQUIRE((((uintptr_t)phi | (uintptr_t)v | (uintptr_t)ds | (uintptr_t)dv | (uintptr_t)s_res | (uintptr_t)v_res |
                (uintptr_t)geom_d) & 15) == 0, "operands must be 16-byte aligned");
  if (rpw < 1) rpw = 1;
  const int groups = (n_feat + 63) / 64;
  const long long items = (long long)((n_dst + rpw - 1) / rpw) * groups;
  const dim3 grid((unsigned)((items + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  CGV_DISPATCH_RBF(n_rbf, {
    if constexpr (RBF + 1 <= 16) {
      if (with_dv)
        hipLaunchKernelGGL((cgv::equi_msg_fwd_mfma16_k<RBF, true>), grid, dim3(256), 0, st, phi, v, geom_d, rowptr_d, src_d, Wd, bd,
                           ds, dv, n_feat, n_dst, groups, rpw, s_res, v_res);
      else
        hipLaunchKernelGGL((cgv::equi_msg_fwd_mfma16_k<RBF, false>), grid, dim3(256), 0, st, phi, v, geom_d, rowptr_d, src_d, Wd,
                           bd, ds, dv, n_feat, n_dst, groups, rpw, s_res, v_res);
    } else {
      cgv::set_error("cgv_equi_msg_fwd_mfma16: n_rbf=%d exceeds one reduction step", n_rbf);
      return CGV_E_UNSUPPORTED;
    }
  });
  return cgv::check_launch("cgv_equi_msg_fwd_mfma16");
}

}  // extern "C"
