// Decoder tail: atom coordinates from the bead vector channels (reference cgvae.py:462-481,
// CGequiVAE.decoder):  xyz_rel[a] = cg_v[mapping[a], chan[a], :]            (chan = rank of the atom in its bead)
//                      xyz_rel   -= scatter_mean(xyz_rel, mapping)[mapping]   (offset=True)
//                      xyz_recon  = xyz_rel + cg_xyz[mapping]
// As tensor ops this is 3 advanced-index gathers + a scatter_mean + 2 adds forward and, backward, two
// sort-based index_put accumulations (~30 small launches, ~100 us of the 2.8 ms chignolin step).  Here one
// wave per bead walks the bead's atoms (bead -> atoms CSR of the contraction plan) in both directions.
// Backward: g_v[b, chan[a], :] = g[a] - mean_b(g) (offset) -- (b, chan) slots are unique per atom, every other
// entry of g_v is zero and is written here too (no separate fill).
#include "cgv_common.h"

namespace cgv {

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
  return x;
}

__global__ __launch_bounds__(64) void reconstruct_fwd(const float* __restrict__ v, const float* __restrict__ cg_xyz,
                                                      const int* __restrict__ rowptr, const int* __restrict__ atom,
                                                      const int64_t* __restrict__ chan, int F, int offset,
                                                      float* __restrict__ xyz) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int beg = rowptr[b], end = rowptr[b + 1];
  const float* vb = v + (size_t)b * F * 3;
  float mx = 0.f, my = 0.f, mz = 0.f;
  if (offset) {
    for (int p = beg + lane; p < end; p += 64) {
      const float* r = vb + 3 * (size_t)chan[atom[p]];
      mx += r[0]; my += r[1]; mz += r[2];
    }
    const float inv = 1.0f / (float)max(end - beg, 1);
    mx = wave_sum(mx) * inv; my = wave_sum(my) * inv; mz = wave_sum(mz) * inv;
  }
  const float cx = cg_xyz[3 * b], cy = cg_xyz[3 * b + 1], cz = cg_xyz[3 * b + 2];
  for (int p = beg + lane; p < end; p += 64) {
    const int a = atom[p];
    const float* r = vb + 3 * (size_t)chan[a];
    xyz[3 * (size_t)a] = (r[0] - mx) + cx;
    xyz[3 * (size_t)a + 1] = (r[1] - my) + cy;
    xyz[3 * (size_t)a + 2] = (r[2] - mz) + cz;
  }
}

__global__ __launch_bounds__(64) void reconstruct_bwd(const float* __restrict__ g, const int* __restrict__ rowptr,
                                                      const int* __restrict__ atom, const int64_t* __restrict__ chan,
                                                      int F, int offset, float* __restrict__ g_v,
                                                      float* __restrict__ g_cg) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int beg = rowptr[b], end = rowptr[b + 1];
  float* gb = g_v + (size_t)b * F * 3;
  for (int t = lane; t < 3 * F; t += 64) gb[t] = 0.f;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int p = beg + lane; p < end; p += 64) {
    const float* r = g + 3 * (size_t)atom[p];
    sx += r[0]; sy += r[1]; sz += r[2];
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  if (g_cg && lane == 0) { g_cg[3 * b] = sx; g_cg[3 * b + 1] = sy; g_cg[3 * b + 2] = sz; }
  const float inv = offset ? 1.0f / (float)max(end - beg, 1) : 0.f;
  __syncthreads();                                   // zero fill of the row is complete (single wave: ordering only)
  for (int p = beg + lane; p < end; p += 64) {
    const int a = atom[p];
    const float* r = g + 3 * (size_t)a;
    float* o = gb + 3 * (size_t)chan[a];
    o[0] = r[0] - sx * inv; o[1] = r[1] - sy * inv; o[2] = r[2] - sz * inv;
  }
}

}  // namespace cgv

extern "C" {

int cgv_reconstruct_fwd(const float* v, const float* cg_xyz, const int32_t* rowptr, const int32_t* atom, const int64_t* chan,
                        int n_beads, int n_feat, int offset, float* xyz, void* stream) {
  CGV_REQUIRE(n_beads >= 0 && n_feat > 0, "bad size");
  if (n_beads == 0) return 0;
  CGV_REQUIRE(v && cg_xyz && rowptr && atom && chan && xyz, "null pointer");
  hipLaunchKernelGGL(cgv::reconstruct_fwd, dim3(n_beads), dim3(64), 0, (hipStream_t)stream, v, cg_xyz, rowptr, atom, chan,
                     n_feat, offset, xyz);
  return cgv::check_launch("cgv_reconstruct_fwd");
}

int cgv_reconstruct_bwd(const float* g_xyz, const int32_t* rowptr, const int32_t* atom, const int64_t* chan, int n_beads,
                        int n_feat, int offset, float* g_v, float* g_cg_xyz, void* stream) {
  CGV_REQUIRE(n_beads >= 0 && n_feat > 0, "bad size");
  if (n_beads == 0) return 0;
  CGV_REQUIRE(g_xyz && rowptr && atom && chan && g_v, "null pointer");
  hipLaunchKernelGGL(cgv::reconstruct_bwd, dim3(n_beads), dim3(64), 0, (hipStream_t)stream, g_xyz, rowptr, atom, chan,
                     n_feat, offset, g_v, g_cg_xyz);
  return cgv::check_launch("cgv_reconstruct_bwd");
}

}  // extern "C"
