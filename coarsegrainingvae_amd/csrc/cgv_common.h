// Shared helpers for libcgvae_hip.so (gfx950 only: wave64, no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "cgvae_hip.h"

namespace cgv {

void set_error(const char* fmt, ...);   // thread-local message (api.cpp)
int option(int id);                     // current value of a CGV_OPT_* switch (api.cpp: cgv_set_option)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

#define CGV_REQUIRE(cond, msg)                         \
  do {                                                 \
    if (!(cond)) {                                     \
      cgv::set_error("%s: %s", __func__, msg);         \
      return CGV_E_BADARG;                             \
    }                                                  \
  } while (0)

constexpr int WAVE = 64;

// Edge-geometry record layout (floats):  [0,R) a_n ; [R] env ; pad to even ; [U,U+6) ux,uy,uz,ux,uy,uz
// The unit vector is stored twice so that every adjacent pair (ux,uy) (uz,ux) (uy,uz) is an
// aligned 64-bit scalar-register pair: the packed-fp32 kernels (v_pk_fma_f32, two channels per
// lane) use them directly as broadcast operands for their AoS (x0,y0)(z0,x1)(y1,z1) accumulators.
__host__ __device__ constexpr int geom_unit_offset(int R) { return (R + 2) & ~1; }
__host__ __device__ constexpr int geom_stride(int R) { return (geom_unit_offset(R) + 6 + 3) & ~3; }
// Record of the receiver-group order (shared-source forward, equi_msg_grp.hip), R even:
//   [0,R) a_n ; [R] env ; [R+1] meta.x (int bits) ; [R+2,R+5) ux,uy,uz ; [R+5] meta.y (int bits) ; pad to a multiple of 4
// meta = cgv_group_plan_build's per-edge words.  At n_rbf = 10 this is exactly one aligned 64-byte scalar load per edge
// (the doubled-unit layout above needs 80 bytes that straddle cache lines, plus a second array for the meta words).
__host__ __device__ constexpr int geom_group_unit_offset(int R) { return R + 2; }
__host__ __device__ constexpr int geom_group_stride(int R) { return (R + 6 + 3) & ~3; }

// Nontemporal 16-byte load.  (Tried on the weight rows of the bead-level weight-streaming products, so that they would not
// displace activations / slices / code from the L2: the decoder got SLOWER -- 352 -> 408 us forward, 527 -> 556 us backward
// on chignolin -- the 243 MB of bead-level weights just fit the 256 MB memory-side cache, which nt loads bypass.)
__device__ __forceinline__ float4 ldg4_nt(const float* p) {
  typedef float nt_f4 __attribute__((ext_vector_type(4)));
  const nt_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}

// 12-byte vector with 4-byte alignment: one global_load_dwordx3 / global_store_dwordx3.
struct __attribute__((packed, aligned(4))) f3 {
  float x, y, z;
};

__device__ inline f3 ld3(const float* p) { return *reinterpret_cast<const f3*>(p); }
__device__ inline void st3(float* p, float x, float y, float z) {
  f3 t{x, y, z};
  *reinterpret_cast<f3*>(p) = t;
}

// Dispatch a runtime n_rbf onto the compiled template instances.
#define CGV_RBF_LIST(X) X(4) X(6) X(8) X(10) X(12) X(16) X(20)

#define CGV_DISPATCH_RBF(R_runtime, CALL)                                   \
  switch (R_runtime) {                                                      \
    case 4: { constexpr int RBF = 4; CALL; } break;                         \
    case 6: { constexpr int RBF = 6; CALL; } break;                         \
    case 8: { constexpr int RBF = 8; CALL; } break;                         \
    case 10: { constexpr int RBF = 10; CALL; } break;                       \
    case 12: { constexpr int RBF = 12; CALL; } break;                       \
    case 16: { constexpr int RBF = 16; CALL; } break;                       \
    case 20: { constexpr int RBF = 20; CALL; } break;                       \
    default:                                                                \
      cgv::set_error("%s: n_rbf=%d has no compiled kernel", __func__, R_runtime); \
      return CGV_E_UNSUPPORTED;                                             \
  }

// Activations of the Dense / nn.Linear kernels (fused into epilogues and operand loads):
//   0 identity   1 Swish x*sigmoid(x) (modules.py:16-21)   2 tanh   3 ReLU   (nn.Tanh / nn.ReLU of the mu / sigma heads)
//   4  1e-12 + exp(z/2)   (sigma of the encoder head, cgvae.py:503)      5  1e-9 + exp(z/2)   (prior std, cgvae.py:401)
// act_bwd is the derivative as a function of the PRE-activation z.
constexpr int CGV_ACT_MAX = 5;
__device__ __forceinline__ float act_sigmoid(float z) { return 1.0f / (1.0f + expf(-z)); }
__device__ __forceinline__ float act_fwd(float z, int act) {
  switch (act) {
    case 1: return z * act_sigmoid(z);
    case 2: return tanhf(z);
    case 3: return z > 0.f ? z : 0.f;
    case 4: return 1e-12f + expf(z / 2.0f);
    case 5: return 1e-9f + expf(z / 2.0f);
    default: return z;
  }
}
__device__ __forceinline__ float act_bwd(float z, int act) {
  switch (act) {
    case 1: { const float s = act_sigmoid(z); return s * (1.0f + z * (1.0f - s)); }
    case 2: { const float t = tanhf(z); return 1.0f - t * t; }
    case 3: return z > 0.f ? 1.0f : 0.f;
    case 4:
    case 5: return 0.5f * expf(z / 2.0f);
    default: return 1.0f;
  }
}


// ------------------------------------------------------------------ slice sums
// A gradient operand handed over as "base + row-slice partial sums": value(i) = base[i] + sum_s slices[s * stride + i],
// s ascending (deterministic).  This is how the split backward-input products (skinny_gemm.hip: the row slices of one
// product each leave a partial [M, K] matrix) reach their consumers without a reduction launch in between: the consumer
// adds the slices while it loads its operand.  base or slices may be NULL (n = 0 when slices is NULL).
struct SliceSum {
  const float* base;
  const float* slices;
  int n;
  long long stride;       // floats between consecutive slices
};
__device__ __forceinline__ float slice_sum_at(const SliceSum& ss, size_t i) {
  float acc = ss.base ? ss.base[i] : 0.f;
  constexpr int SB = 8;                                     // slices whose loads are in flight together
  for (int s0 = 0; s0 < ss.n; s0 += SB) {
    float v[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) v[u] = ss.slices[(size_t)min(s0 + u, ss.n - 1) * ss.stride + i];
#pragma unroll
    for (int u = 0; u < SB; ++u) acc += (s0 + u < ss.n) ? v[u] : 0.f;
  }
  return acc;
}

// ------------------------------------------------------------------ fused optimiser (optim.hip, skinny_gemm.hip)
// state[] layout (device floats) written by optim_finalize, read by every parameter pass
enum { ST_STEP = 0, ST_NORM = 1, ST_CLIP = 2, ST_BC1 = 3, ST_BC2SQRT = 4, ST_SKIP = 5, ST_NSKIPPED = 6 };

struct AdamStep {            // per-launch constants of one parameter pass (torch.optim.Adam defaults otherwise)
  float clip, step_size, inv_bc2, beta1, beta2, eps;
};
__device__ __forceinline__ AdamStep adam_step_of(const float* __restrict__ state, float lr, float beta1, float beta2, float eps) {
  return AdamStep{state[ST_CLIP], lr / state[ST_BC1], 1.0f / state[ST_BC2SQRT], beta1, beta2, eps};
}
__device__ __forceinline__ void adam_elem(const AdamStep& a, float& pp, float gg, float& mm, float& vv) {
  gg *= a.clip;
  mm = fmaf(1.f - a.beta1, gg - mm, mm);                       // exp_avg.lerp_(grad, 1 - beta1)
  vv = fmaf(1.f - a.beta2, gg * gg, a.beta2 * vv);             // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
  const float denom = sqrtf(vv) * a.inv_bc2 + a.eps;
  pp -= a.step_size * (mm / denom);
}

}  // namespace cgv
