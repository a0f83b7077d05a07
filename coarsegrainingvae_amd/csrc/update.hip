// K5: element-wise core of UpdateBlock (reference CoarseGrainingVAE/conv.py:588-616), forward +
// backward.  The four K=F GEMMs (u_mat, v_mat, s_dense.0/.1) stay with the caller; these
// kernels replace the ~20 ATen element-wise / reduction launches between them.
//   U, Vv : u_mat / v_mat applied to v, laid out [N,3,F] (row = node*3 + xyz)   conv.py:593-598
//   stack = [ s | vnorm ],  vnorm[n,f] = sqrt(sum_k (Vv[n,k,f]^2 + 1e-10))       conv.py:600-601
//   a = s_dense(stack) viewed [N,3,F]  (a_vv, a_sv, a_ss)                        conv.py:603-612
//   dv[n,f,k] = U[n,k,f] a_vv[n,f] ;  ds[n,f] = (sum_k U Vv) a_sv + a_ss          conv.py:607-614
#include "cgv_common.h"

namespace cgv {

__global__ __launch_bounds__(256) void update_norm_stack_fwd(const float* __restrict__ s, const float* __restrict__ Vv,
                                                             float* __restrict__ stack, int N, int F) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * F) return;
  const int n = idx / F, f = idx - n * F;
  const float* vv = Vv + (size_t)n * 3 * F + f;
  const float x = vv[0], y = vv[F], z = vv[2 * F];
  const float nrm = sqrtf(((x * x + 1e-10f) + (y * y + 1e-10f)) + (z * z + 1e-10f));
  stack[(size_t)n * 2 * F + f] = s[idx];
  stack[(size_t)n * 2 * F + F + f] = nrm;
}

__global__ __launch_bounds__(256) void update_norm_stack_bwd(const float* __restrict__ gstack,
                                                             const float* __restrict__ Vv,
                                                             const float* __restrict__ stack, float* __restrict__ g_s,
                                                             float* __restrict__ gVv, int N, int F) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * F) return;
  const int n = idx / F, f = idx - n * F;
  g_s[idx] = gstack[(size_t)n * 2 * F + f];
  const float t = gstack[(size_t)n * 2 * F + F + f] / stack[(size_t)n * 2 * F + F + f];
  const float* vv = Vv + (size_t)n * 3 * F + f;
  float* o = gVv + (size_t)n * 3 * F + f;
  o[0] = t * vv[0];
  o[F] = t * vv[F];
  o[2 * F] = t * vv[2 * F];
}

__global__ __launch_bounds__(256) void update_gate_fwd(const float* __restrict__ U, const float* __restrict__ Vv,
                                                       const float* __restrict__ a, const float* __restrict__ s_res,
                                                       const float* __restrict__ v_res, float* __restrict__ ds,
                                                       float* __restrict__ dv, int N, int F) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * F) return;
  const int n = idx / F, f = idx - n * F;
  const size_t b = (size_t)n * 3 * F + f;
  const float ux = U[b], uy = U[b + F], uz = U[b + 2 * F];
  const float vx = Vv[b], vy = Vv[b + F], vz = Vv[b + 2 * F];
  const float a_vv = a[b], a_sv = a[b + F], a_ss = a[b + 2 * F];
  float ox = ux * a_vv, oy = uy * a_vv, oz = uz * a_vv;
  float os = (ux * vx + uy * vy + uz * vz) * a_sv + a_ss;
  if (s_res) {         // emit S + dS_update, V + dV_update (cgvae.py:122-123) from the same launch
    const f3 r = ld3(v_res + (size_t)idx * 3);
    ox += r.x; oy += r.y; oz += r.z;
    os += s_res[idx];
  }
  st3(dv + (size_t)idx * 3, ox, oy, oz);
  ds[idx] = os;
}

__global__ __launch_bounds__(256) void update_gate_bwd(const float* __restrict__ U, const float* __restrict__ Vv,
                                                       const float* __restrict__ a, const float* __restrict__ g_ds,
                                                       const float* __restrict__ g_dv, float* __restrict__ gU,
                                                       float* __restrict__ gVv, float* __restrict__ ga, int N, int F) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * F) return;
  const int n = idx / F, f = idx - n * F;
  const size_t b = (size_t)n * 3 * F + f;
  const float ux = U[b], uy = U[b + F], uz = U[b + 2 * F];
  const float vx = Vv[b], vy = Vv[b + F], vz = Vv[b + 2 * F];
  const float a_vv = a[b], a_sv = a[b + F];
  const float gs = g_ds ? g_ds[idx] : 0.f;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (g_dv) { const f3 t = ld3(g_dv + (size_t)idx * 3); gx = t.x; gy = t.y; gz = t.z; }
  const float inner = ux * vx + uy * vy + uz * vz;
  const float c = gs * a_sv;
  ga[b] = gx * ux + gy * uy + gz * uz;
  ga[b + F] = gs * inner;
  ga[b + 2 * F] = gs;
  gU[b] = fmaf(gx, a_vv, c * vx);
  gU[b + F] = fmaf(gy, a_vv, c * vy);
  gU[b + 2 * F] = fmaf(gz, a_vv, c * vz);
  gVv[b] = c * ux;
  gVv[b + F] = c * uy;
  gVv[b + 2 * F] = c * uz;
}

}  // namespace cgv

extern "C" {

#define CGV_EW_LAUNCH(kernel, ...)                                                                       \
  CGV_REQUIRE(n_nodes >= 0 && n_feat > 0, "bad size");                                                   \
  if (n_nodes == 0) return 0;                                                                            \
  hipLaunchKernelGGL(kernel, dim3(((size_t)n_nodes * n_feat + 255) / 256), dim3(256), 0, (hipStream_t)stream, \
                     __VA_ARGS__, n_nodes, n_feat);                                                      \
  return cgv::check_launch(#kernel)

int cgv_update_norm_stack_fwd(const float* s, const float* Vv, float* stack, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(s && Vv && stack, "null pointer");
  CGV_EW_LAUNCH(cgv::update_norm_stack_fwd, s, Vv, stack);
}

int cgv_update_norm_stack_bwd(const float* gstack, const float* Vv, const float* stack, float* g_s, float* gVv,
                              int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(gstack && Vv && stack && g_s && gVv, "null pointer");
  CGV_EW_LAUNCH(cgv::update_norm_stack_bwd, gstack, Vv, stack, g_s, gVv);
}

int cgv_update_gate_fwd(const float* U, const float* Vv, const float* a, const float* s_res, const float* v_res, float* ds,
                        float* dv, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(U && Vv && a && ds && dv, "null pointer");
  CGV_REQUIRE((s_res == nullptr) == (v_res == nullptr), "s_res and v_res go together");
  CGV_EW_LAUNCH(cgv::update_gate_fwd, U, Vv, a, s_res, v_res, ds, dv);
}

int cgv_update_gate_bwd(const float* U, const float* Vv, const float* a, const float* g_ds, const float* g_dv,
                        float* gU, float* gVv, float* ga, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(U && Vv && a && gU && gVv && ga, "null pointer");
  CGV_EW_LAUNCH(cgv::update_gate_bwd, U, Vv, a, g_ds, g_dv, gU, gVv, ga);
}

}  // extern "C"
