// K5: element-wise core of UpdateBlock (reference CoarseGrainingVAE/conv.py:588-616), forward +
// backward.  The K=F products (u_mat, v_mat, s_dense.0/.1) are separate launches (skinny_gemm.hip);
// these kernels replace the ~20 ATen element-wise / reduction launches between them.
//   U, Vv : u_mat / v_mat applied to v, rows r = node*3 + xyz, row stride `ld` floats            conv.py:593-598
//           (ld = F for separate buffers; ld = 2F when both live in one [3N, 2F] product of the
//            concatenated weights [u_mat; v_mat] -- one GEMM instead of two)
//   stack = [ s | vnorm ],  vnorm[n,f] = sqrt(sum_k (Vv[n,k,f]^2 + 1e-10))       conv.py:600-601
//   a = s_dense(stack) viewed [N,3,F]  (a_vv, a_sv, a_ss)                        conv.py:603-612
//   dv[n,f,k] = U[n,k,f] a_vv[n,f] ;  ds[n,f] = (sum_k U Vv) a_sv + a_ss          conv.py:607-614
#include "cgv_common.h"

namespace cgv {

// v [N,F,3] -> rows [N,3,F]  (conv.py:591  v.transpose(1,2).reshape(-1, F))
__global__ __launch_bounds__(256) void update_rows_from_vec(const float* __restrict__ v, float* __restrict__ rows, int N,
                                                            int F) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * F) return;
  const int n = idx / F, f = idx - n * F;
  const f3 t = ld3(v + (size_t)idx * 3);
  float* o = rows + (size_t)n * 3 * F + f;
  o[0] = t.x; o[F] = t.y; o[2 * F] = t.z;
}

// rows [N,3,F] (+ res [N,F,3]) -> vec [N,F,3]: the transposed copy of the backward, with the
// pass-through gradient of the fused residual added in the same launch
__global__ __launch_bounds__(256) void update_vec_from_rows(SliceSum rows, const float* __restrict__ res,
                                                            float* __restrict__ vec, int N, int F) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * F) return;
  const int n = idx / F, f = idx - n * F;
  const size_t r = (size_t)n * 3 * F + f;
  float x = slice_sum_at(rows, r), y = slice_sum_at(rows, r + F), z = slice_sum_at(rows, r + 2 * F);
  if (res) { const f3 t = ld3(res + (size_t)idx * 3); x += t.x; y += t.y; z += t.z; }
  st3(vec + (size_t)idx * 3, x, y, z);
}

__global__ __launch_bounds__(256) void update_norm_stack_fwd(const float* __restrict__ s, const float* __restrict__ Vv,
                                                             float* __restrict__ stack, int N, int F, int ld) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * F) return;
  const int n = idx / F, f = idx - n * F;
  const float* vv = Vv + (size_t)n * 3 * ld + f;
  const float x = vv[0], y = vv[ld], z = vv[2 * ld];
  const float nrm = sqrtf(((x * x + 1e-10f) + (y * y + 1e-10f)) + (z * z + 1e-10f));
  stack[(size_t)n * 2 * F + f] = s[idx];
  stack[(size_t)n * 2 * F + F + f] = nrm;
}

// g_s = gstack[:, :F] (+ g_res);  gVv (+)= gstack[:, F:] * Vv / vnorm
__global__ __launch_bounds__(256) void update_norm_stack_bwd(SliceSum gstack,
                                                             const float* __restrict__ Vv,
                                                             const float* __restrict__ stack,
                                                             SliceSum g_res, float* __restrict__ g_s,
                                                             float* __restrict__ gVv, int N, int F, int ld,
                                                             int accumulate) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * F) return;
  const int n = idx / F, f = idx - n * F;
  float gs = slice_sum_at(gstack, (size_t)n * 2 * F + f);
  gs += slice_sum_at(g_res, idx);                            // 0 when no residual gradient is given
  g_s[idx] = gs;
  const float t = slice_sum_at(gstack, (size_t)n * 2 * F + F + f) / stack[(size_t)n * 2 * F + F + f];
  const float* vv = Vv + (size_t)n * 3 * ld + f;
  float* o = gVv + (size_t)n * 3 * ld + f;
  if (accumulate) {
    o[0] += t * vv[0]; o[ld] += t * vv[ld]; o[2 * ld] += t * vv[2 * ld];
  } else {
    o[0] = t * vv[0]; o[ld] = t * vv[ld]; o[2 * ld] = t * vv[2 * ld];
  }
}

__global__ __launch_bounds__(256) void update_gate_fwd(const float* __restrict__ U, const float* __restrict__ Vv,
                                                       const float* __restrict__ a, const float* __restrict__ s_res,
                                                       const float* __restrict__ v_res, float* __restrict__ ds,
                                                       float* __restrict__ dv, int N, int F, int ld) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * F) return;
  const int n = idx / F, f = idx - n * F;
  const size_t b = (size_t)n * 3 * ld + f, c = (size_t)n * 3 * F + f;
  const float ux = U[b], uy = U[b + ld], uz = U[b + 2 * ld];
  const float vx = Vv[b], vy = Vv[b + ld], vz = Vv[b + 2 * ld];
  const float a_vv = a[c], a_sv = a[c + F], a_ss = a[c + 2 * F];
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
                                                       const float* __restrict__ a, SliceSum g_ds,
                                                       const float* __restrict__ g_dv, float* __restrict__ gU,
                                                       float* __restrict__ gVv, float* __restrict__ ga, int N, int F,
                                                       int ld) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * F) return;
  const int n = idx / F, f = idx - n * F;
  const size_t b = (size_t)n * 3 * ld + f, c = (size_t)n * 3 * F + f;
  const float ux = U[b], uy = U[b + ld], uz = U[b + 2 * ld];
  const float vx = Vv[b], vy = Vv[b + ld], vz = Vv[b + 2 * ld];
  const float a_vv = a[c], a_sv = a[c + F];
  const float gs = slice_sum_at(g_ds, idx);                  // 0 when neither base nor slices are given
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (g_dv) { const f3 t = ld3(g_dv + (size_t)idx * 3); gx = t.x; gy = t.y; gz = t.z; }
  const float inner = ux * vx + uy * vy + uz * vz;
  const float cc = gs * a_sv;
  ga[c] = gx * ux + gy * uy + gz * uz;
  ga[c + F] = gs * inner;
  ga[c + 2 * F] = gs;
  gU[b] = fmaf(gx, a_vv, cc * vx);
  gU[b + ld] = fmaf(gy, a_vv, cc * vy);
  gU[b + 2 * ld] = fmaf(gz, a_vv, cc * vz);
  gVv[b] = cc * ux;
  gVv[b + ld] = cc * uy;
  gVv[b + 2 * ld] = cc * uz;
}

}  // namespace cgv

extern "C" {

#define CGV_EW_LAUNCH(kernel, ...)                                                                       \
  CGV_REQUIRE(n_nodes >= 0 && n_feat > 0, "bad size");                                                   \
  if (n_nodes == 0) return 0;                                                                            \
  hipLaunchKernelGGL(kernel, dim3(((size_t)n_nodes * n_feat + 255) / 256), dim3(256), 0, (hipStream_t)stream, \
                     __VA_ARGS__);                                                                       \
  return cgv::check_launch(#kernel)

int cgv_update_rows_from_vec(const float* v, float* rows, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(v && rows, "null pointer");
  CGV_EW_LAUNCH(cgv::update_rows_from_vec, v, rows, n_nodes, n_feat);
}

int cgv_update_vec_from_rows(const float* rows, const float* res, float* vec, int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(rows && vec, "null pointer");
  CGV_EW_LAUNCH(cgv::update_vec_from_rows, (cgv::SliceSum{rows, nullptr, 0, 0}), res, vec, n_nodes, n_feat);
}

int cgv_update_vec_from_rows_slices(const float* rows_slices, int n_slices, int64_t slice_stride, const float* res, float* vec,
                                    int n_nodes, int n_feat, void* stream) {
  CGV_REQUIRE(rows_slices && vec && n_slices >= 1 && slice_stride >= (int64_t)3 * n_nodes * n_feat, "bad argument");
  CGV_EW_LAUNCH(cgv::update_vec_from_rows, (cgv::SliceSum{nullptr, rows_slices, n_slices, slice_stride}), res, vec, n_nodes, n_feat);
}

int cgv_update_norm_stack_fwd(const float* s, const float* Vv, float* stack, int n_nodes, int n_feat, int ld, void* stream) {
  CGV_REQUIRE(s && Vv && stack && ld >= n_feat, "bad argument");
  CGV_EW_LAUNCH(cgv::update_norm_stack_fwd, s, Vv, stack, n_nodes, n_feat, ld);
}

int cgv_update_norm_stack_bwd(const float* gstack, const float* Vv, const float* stack, const float* g_res, float* g_s,
                              float* gVv, int n_nodes, int n_feat, int ld, int accumulate, void* stream) {
  CGV_REQUIRE(gstack && Vv && stack && g_s && gVv && ld >= n_feat, "bad argument");
  CGV_EW_LAUNCH(cgv::update_norm_stack_bwd, (cgv::SliceSum{gstack, nullptr, 0, 0}), Vv, stack, (cgv::SliceSum{g_res, nullptr, 0, 0}),
                g_s, gVv, n_nodes, n_feat, ld, accumulate);
}

int cgv_update_norm_stack_bwd_slices(const float* gstack_slices, int n_slices, int64_t slice_stride, const float* Vv,
                                     const float* stack, const float* g_res_base, const float* g_res_slices, int n_res_slices,
                                     int64_t res_slice_stride, float* g_s, float* gVv, int n_nodes, int n_feat, int ld,
                                     int accumulate, void* stream) {
  CGV_REQUIRE(gstack_slices && Vv && stack && g_s && gVv && ld >= n_feat, "bad argument");
  CGV_REQUIRE(n_slices >= 1 && slice_stride >= (int64_t)2 * n_nodes * n_feat, "bad slices");
  CGV_REQUIRE(n_res_slices >= 0 && (n_res_slices == 0 || (g_res_slices && res_slice_stride >= (int64_t)n_nodes * n_feat)),
              "bad residual slices");
  CGV_EW_LAUNCH(cgv::update_norm_stack_bwd, (cgv::SliceSum{nullptr, gstack_slices, n_slices, slice_stride}), Vv, stack,
                (cgv::SliceSum{g_res_base, g_res_slices, g_res_slices ? n_res_slices : 0, res_slice_stride}), g_s, gVv, n_nodes,
                n_feat, ld, accumulate);
}

int cgv_update_gate_fwd(const float* U, const float* Vv, const float* a, const float* s_res, const float* v_res, float* ds,
                        float* dv, int n_nodes, int n_feat, int ld, void* stream) {
  CGV_REQUIRE(U && Vv && a && ds && dv && ld >= n_feat, "bad argument");
  CGV_REQUIRE((s_res == nullptr) == (v_res == nullptr), "s_res and v_res go together");
  CGV_EW_LAUNCH(cgv::update_gate_fwd, U, Vv, a, s_res, v_res, ds, dv, n_nodes, n_feat, ld);
}

int cgv_update_gate_bwd(const float* U, const float* Vv, const float* a, const float* g_ds, const float* g_dv,
                        float* gU, float* gVv, float* ga, int n_nodes, int n_feat, int ld, void* stream) {
  CGV_REQUIRE(U && Vv && a && gU && gVv && ga && ld >= n_feat, "bad argument");
  CGV_EW_LAUNCH(cgv::update_gate_bwd, U, Vv, a, (cgv::SliceSum{g_ds, nullptr, 0, 0}), g_dv, gU, gVv, ga, n_nodes, n_feat, ld);
}

int cgv_update_gate_bwd_slices(const float* U, const float* Vv, const float* a, const float* g_ds_base,
                               const float* g_ds_slices, int n_slices, int64_t slice_stride, const float* g_dv, float* gU,
                               float* gVv, float* ga, int n_nodes, int n_feat, int ld, void* stream) {
  CGV_REQUIRE(U && Vv && a && gU && gVv && ga && ld >= n_feat, "bad argument");
  CGV_REQUIRE(n_slices >= 0 && (n_slices == 0 || (g_ds_slices && slice_stride >= (int64_t)n_nodes * n_feat)), "bad slices");
  CGV_EW_LAUNCH(cgv::update_gate_bwd, U, Vv, a, (cgv::SliceSum{g_ds_base, g_ds_slices, n_slices, slice_stride}), g_dv, gU, gVv, ga,
                n_nodes, n_feat, ld);
}

}  // extern "C"
