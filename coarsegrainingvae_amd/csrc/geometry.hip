// K6 edge geometry: distance, unit vector, cosine envelope and sinc radial basis per edge,
// computed once per (graph, cutoff, n_rbf) and reused by every layer.
// Restates preprocess_r (reference conv.py:25-29), PainnRadialBasis (modules.py:148-172) and
// CosineEnvelope (modules.py:52-58); the operation order of each fp32 expression follows the
// reference so the only differences are device sqrt/sin/cos roundings (<= 1-2 ulp).
#include "cgv_common.h"

namespace cgv {

__global__ __launch_bounds__(256) void edge_geometry(const float* __restrict__ r_edges, const int* __restrict__ eid,
                                                     const float* __restrict__ pos_dst, const float* __restrict__ pos_src,
                                                     const int* __restrict__ dst, const int* __restrict__ src, int E, int R,
                                                     int GS, float cutoff, float pi_f, const float* __restrict__ coef,
                                                     float* __restrict__ geom, const int2* __restrict__ meta) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= E) return;
  float rx, ry, rz;
  if (r_edges) {
    const float* r = r_edges + 3 * (size_t)eid[p];
    rx = r[0]; ry = r[1]; rz = r[2];
  } else {
    const float* a = pos_src + 3 * (size_t)src[p];
    const float* b = pos_dst + 3 * (size_t)dst[p];
    rx = __fsub_rn(a[0], b[0]); ry = __fsub_rn(a[1], b[1]); rz = __fsub_rn(a[2], b[2]);
  }
  // conv.py:26  dist = ((r**2 + 1e-8).sum(-1)) ** 0.5   (sum order ((x+y)+z), no contraction)
  float sx = __fadd_rn(__fmul_rn(rx, rx), 1e-8f);
  float sy = __fadd_rn(__fmul_rn(ry, ry), 1e-8f);
  float sz = __fadd_rn(__fmul_rn(rz, rz), 1e-8f);
  float d = __fsqrt_rn(__fadd_rn(__fadd_rn(sx, sy), sz));
  float* g = geom + (size_t)p * GS;
  // modules.py:54-56  0.5 * (cos(pi * d / cutoff) + 1), zero at and beyond the cutoff
  float env = 0.5f * (cosf(__fdiv_rn(__fmul_rn(pi_f, d), cutoff)) + 1.0f);
  const bool outside = d >= cutoff;
  if (outside) env = 0.0f;
  // modules.py:161-170  sin(coef*d)/d, coef at d == 0, 0 beyond the cutoff
  for (int n = 0; n < R; ++n) {
    float c = coef[n];
    float val = (d == 0.0f) ? c : __fdiv_rn(sinf(__fmul_rn(c, d)), d);
    if (outside) val = 0.0f;
    g[n] = val * env;
  }
  g[R] = env;
  const float ux = __fdiv_rn(rx, d), uy = __fdiv_rn(ry, d), uz = __fdiv_rn(rz, d);   // conv.py:27 unit = r / dist
  if (meta) {                                     // receiver-group record (cgv_common.h: geom_group_*)
    const int2 m = meta[p];
    g[R + 1] = __int_as_float(m.x);
    g[R + 2] = ux; g[R + 3] = uy; g[R + 4] = uz;
    g[R + 5] = __int_as_float(m.y);
    for (int k = R + 6; k < GS; ++k) g[k] = 0.0f;
    return;
  }
  const int U = geom_unit_offset(R);
  for (int k = R + 1; k < U; ++k) g[k] = 0.0f;
  g[U + 0] = ux; g[U + 1] = uy; g[U + 2] = uz;
  g[U + 3] = ux; g[U + 4] = uy; g[U + 5] = uz;
  for (int k = U + 6; k < GS; ++k) g[k] = 0.0f;
}

}  // namespace cgv

extern "C" int cgv_edge_geometry(const float* r_edges, const int32_t* eid, const float* pos_dst, const float* pos_src,
                                 const int32_t* dst, const int32_t* src, int n_edges, int n_rbf, float cutoff,
                                 const float* coef, float* geom, void* stream) {
  CGV_REQUIRE(n_edges >= 0 && n_rbf > 0 && coef, "bad argument");
  if (n_edges == 0) return 0;
  CGV_REQUIRE(geom, "null geom");
  CGV_REQUIRE((r_edges && eid) || (pos_dst && pos_src && dst && src), "need r_edges+eid or positions+dst+src");
  const float pi_f = 3.14159265358979323846f;
  hipLaunchKernelGGL(cgv::edge_geometry, dim3((n_edges + 255) / 256), dim3(256), 0, (hipStream_t)stream, r_edges, eid,
                     pos_dst, pos_src, dst, src, n_edges, n_rbf, cgv::geom_stride(n_rbf), cutoff, pi_f, coef, geom,
                     (const int2*)nullptr);
  return cgv::check_launch("cgv_edge_geometry");
}

/* Records in receiver-group order with the per-edge meta words folded in (cgv_common.h: geom_group_stride). */
extern "C" int cgv_geom_group_stride(int n_rbf) { return cgv::geom_group_stride(n_rbf); }
extern "C" int cgv_geom_group_unit_offset(int n_rbf) { return cgv::geom_group_unit_offset(n_rbf); }
extern "C" int cgv_edge_geometry_grouped(const float* pos_dst, const float* pos_src, const int32_t* dst_g,
                                         const int32_t* src_g, const int32_t* meta_g, int n_edges, int n_rbf, float cutoff,
                                         const float* coef, float* geom_g, void* stream) {
  CGV_REQUIRE(n_edges >= 0 && n_rbf > 0 && (n_rbf % 2) == 0 && coef, "bad argument (n_rbf must be even)");
  if (n_edges == 0) return 0;
  CGV_REQUIRE(geom_g && pos_dst && pos_src && dst_g && src_g && meta_g, "null pointer");
  CGV_REQUIRE((((uintptr_t)meta_g) & 7) == 0, "meta_g must be 8-byte aligned");
  const float pi_f = 3.14159265358979323846f;
  hipLaunchKernelGGL(cgv::edge_geometry, dim3((n_edges + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const float*)nullptr, (const int*)nullptr, pos_dst, pos_src, dst_g, src_g, n_edges, n_rbf,
                     cgv::geom_group_stride(n_rbf), cutoff, pi_f, coef, geom_g, reinterpret_cast<const int2*>(meta_g));
  return cgv::check_launch("cgv_edge_geometry_grouped");
}
