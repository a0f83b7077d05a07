// Fused ELBO loss of the reference training loop (scripts/utils.py:81-86 KL, 117-141 loss), forward
// and gradient in one launch each instead of ~40 + ~80 tensor-op launches per step:
//   KL     = 0.5 * mean_I [ sum_f s1^2/s2^2 + (m1-m2)^2/s2 + log s2^2 - log s1^2 ] - 0.5 F      (sic: / s2, not s2^2)
//   recon  = mean( (xr - x)^2 )
//   graph  = mean_b ( (|xr_a - xr_b|_eps - |x_a - x_b|_eps)^2 ),  |d|_eps = sqrt(d.d + 1e-6)
//   loss   = recon + beta * KL + gamma * graph
// One 1024-thread block: the tensors are a few thousand elements; sums run in double in a fixed
// order (deterministic).  The same launch stores d loss / d{mu, sigma, prior_mu, prior_std, xyz_recon};
// backward scales them by the upstream scalar (elbo_scale).
#include "cgv_common.h"

namespace cgv {

__device__ inline double block_sum(double x, double* sh) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = x;
  __syncthreads();
  double t = 0.0;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sh[k];
  return t;
}

// KL terms of a large bead batch on many blocks (dipeptide, 32 frames: 57 600 elements took 2/3 of the single-block
// kernel's 109 us): element gradients as in elbo_fwd, one double partial sum per block in a fixed order.
__global__ __launch_bounds__(256) void elbo_kl_k(const float* __restrict__ mu, const float* __restrict__ sigma,
                                                 const float* __restrict__ pmu, const float* __restrict__ pstd, int nk,
                                                 float ck, float* __restrict__ g_mu, float* __restrict__ g_sigma,
                                                 float* __restrict__ g_pmu, float* __restrict__ g_pstd,
                                                 double* __restrict__ part) {
  __shared__ double sh[16];
  double kl = 0.0;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nk; idx += gridDim.x * blockDim.x) {
    const float m1 = mu[idx], s1 = sigma[idx], m2 = pmu[idx], s2 = pstd[idx];
    const float s1s = s1 * s1, s2s = s2 * s2, dm = m1 - m2;
    kl += (double)(s1s / s2s + dm * dm / s2 + logf(s2s) - logf(s1s));
    g_mu[idx] = ck * (2.f * dm / s2);
    g_pmu[idx] = -ck * (2.f * dm / s2);
    g_sigma[idx] = ck * (2.f * s1 / s2s - 2.f / s1);
    g_pstd[idx] = ck * (-2.f * s1s / (s2s * s2) - dm * dm / s2s + 2.f / s2);
  }
  kl = block_sum(kl, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = kl;
}

// Bond-graph term of a large molecule on many blocks (2000 atoms / 1999 bonds: the single block's atom x bond scan was
// 200 of that kernel's 220 us).  Block b owns atoms [64 b, 64 b + 64): every block stages all bonds chunk by chunk
// (value and d / d xr_a0 per bond: cheap, recomputed per block), wave g of 16 scans slice g of the chunk for the
// block's atoms (lane = atom), the waves' sums meet in LDS in wave order.  g_xr receives the COMPLETE gradient
// (reconstruction part + bond part); block 0 also leaves sum_b diff_b^2 (double) for elbo_fwd.
constexpr int BOND_ATOMS = 64, BOND_WAVES = 16, BOND_CH = 2048;
__global__ __launch_bounds__(1024) void elbo_bond_k(const float* __restrict__ xyz, const float* __restrict__ xr,
                                                    const int64_t* __restrict__ bonds, int n_atoms, int n_bonds, float gamma,
                                                    float* __restrict__ g_xr, double* __restrict__ gr_out) {
  __shared__ double sh[16];
  __shared__ __attribute__((aligned(16))) int sb_a0[BOND_CH], sb_a1[BOND_CH];
  __shared__ __attribute__((aligned(16))) float sb_cx[BOND_CH], sb_cy[BOND_CH], sb_cz[BOND_CH];
  __shared__ float part[BOND_WAVES][3][BOND_ATOMS];
  const int t = threadIdx.x, T = blockDim.x;
  const int lane = t & 63, wave = t >> 6;
  const int a = blockIdx.x * BOND_ATOMS + lane;
  const float cg = n_bonds > 0 ? gamma * 2.f / (float)n_bonds : 0.f;
  const int nr = n_atoms * 3;
  double gr = 0.0;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  for (int base = 0; base < n_bonds; base += BOND_CH) {
    const int cnt = min(BOND_CH, n_bonds - base);
    __syncthreads();
    for (int b = t; b < cnt; b += T) {
      const int a0 = (int)bonds[2 * (size_t)(base + b)], a1 = (int)bonds[2 * (size_t)(base + b) + 1];
      const float ex = xr[3 * a0] - xr[3 * a1], ey = xr[3 * a0 + 1] - xr[3 * a1 + 1], ez = xr[3 * a0 + 2] - xr[3 * a1 + 2];
      const float fx = xyz[3 * a0] - xyz[3 * a1], fy = xyz[3 * a0 + 1] - xyz[3 * a1 + 1], fz = xyz[3 * a0 + 2] - xyz[3 * a1 + 2];
      const float lg = sqrtf(1e-6f + ex * ex + ey * ey + ez * ez), ld = sqrtf(1e-6f + fx * fx + fy * fy + fz * fz);
      const float diff = lg - ld;
      gr += (double)(diff * diff);
      const float c = (a0 == a1) ? 0.f : cg * diff / lg;
      sb_a0[b] = a0; sb_a1[b] = a1;
      sb_cx[b] = c * ex; sb_cy[b] = c * ey; sb_cz[b] = c * ez;
    }
    for (int b = cnt + t; b < ((cnt + 15) & ~15); b += T) { sb_a0[b] = sb_a1[b] = -1; sb_cx[b] = sb_cy[b] = sb_cz[b] = 0.f; }
    __syncthreads();
    const int trips = (cnt + 15) / 16, tper = (trips + BOND_WAVES - 1) / BOND_WAVES;
    const int b_lo = 16 * tper * wave, b_hi = min(cnt, 16 * tper * (wave + 1));
    for (int b0 = b_lo; b0 < b_hi; b0 += 16) {           // branch-free, all reads of a trip issued together (see elbo_fwd)
      int4 i0[4], i1[4];
      float4 cx[4], cy[4], cz[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = b0 + 4 * u;
        i0[u] = *reinterpret_cast<const int4*>(sb_a0 + b); i1[u] = *reinterpret_cast<const int4*>(sb_a1 + b);
        cx[u] = *reinterpret_cast<const float4*>(sb_cx + b); cy[u] = *reinterpret_cast<const float4*>(sb_cy + b);
        cz[u] = *reinterpret_cast<const float4*>(sb_cz + b);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float w0 = i0[u].x == a ? 1.f : (i1[u].x == a ? -1.f : 0.f);
        const float w1 = i0[u].y == a ? 1.f : (i1[u].y == a ? -1.f : 0.f);
        const float w2 = i0[u].z == a ? 1.f : (i1[u].z == a ? -1.f : 0.f);
        const float w3 = i0[u].w == a ? 1.f : (i1[u].w == a ? -1.f : 0.f);
        gx = fmaf(w3, cx[u].w, fmaf(w2, cx[u].z, fmaf(w1, cx[u].y, fmaf(w0, cx[u].x, gx))));
        gy = fmaf(w3, cy[u].w, fmaf(w2, cy[u].z, fmaf(w1, cy[u].y, fmaf(w0, cy[u].x, gy))));
        gz = fmaf(w3, cz[u].w, fmaf(w2, cz[u].z, fmaf(w1, cz[u].y, fmaf(w0, cz[u].x, gz))));
      }
    }
  }
  part[wave][0][lane] = gx; part[wave][1][lane] = gy; part[wave][2][lane] = gz;
  __syncthreads();
  if (wave == 0 && a < n_atoms) {
    float tx = 0.f, ty = 0.f, tz = 0.f;
#pragma unroll
    for (int w = 0; w < BOND_WAVES; ++w) { tx += part[w][0][lane]; ty += part[w][1][lane]; tz += part[w][2][lane]; }
    const float sc = 2.f / (float)nr;                                    // reconstruction part: 2 (xr - x) / nr
    g_xr[3 * a] = sc * (xr[3 * a] - xyz[3 * a]) + tx;
    g_xr[3 * a + 1] = sc * (xr[3 * a + 1] - xyz[3 * a + 1]) + ty;
    g_xr[3 * a + 2] = sc * (xr[3 * a + 2] - xyz[3 * a + 2]) + tz;
  }
  if (blockIdx.x == 0) {
    gr = block_sum(gr, sh);
    if (t == 0) *gr_out = gr;
  }
}

__global__ __launch_bounds__(1024) void elbo_fwd(const float* __restrict__ mu, const float* __restrict__ sigma,
                                                 const float* __restrict__ pmu, const float* __restrict__ pstd,
                                                 const float* __restrict__ xyz, const float* __restrict__ xr,
                                                 const int64_t* __restrict__ bonds, int n_beads, int F, int n_atoms,
                                                 int n_bonds, float beta, float gamma, float* __restrict__ out /*[4]*/,
                                                 float* __restrict__ loss_out /*[1] or NULL: a second copy of out[0]*/,
                                                 float* __restrict__ g_mu, float* __restrict__ g_sigma,
                                                 float* __restrict__ g_pmu, float* __restrict__ g_pstd,
                                                 float* __restrict__ g_xr, const double* __restrict__ kl_part,
                                                 int n_kl_part, const double* __restrict__ bond_sum) {
  __shared__ double sh[16];
  const int t = threadIdx.x, T = blockDim.x;
  // ---- KL and its gradients (mean over beads of per-bead sums); for large bead batches elbo_kl_k has already
  // produced the gradients and per-block partial sums (added here in block order)
  const int nk = n_kl_part > 0 ? 0 : n_beads * F;
  const float ck = 0.5f * beta / (float)n_beads;
  double kl = 0.0;
  // one partial per thread (<= 128 of them), summed by block_sum below: a single thread adding them in a loop paid a
  // memory round trip per partial (57 on the dipeptide batch: most of this kernel's 44 us there)
  for (int b = t; b < n_kl_part; b += T) kl += kl_part[b];
  // small bead batches (chignolin: 7200 elements = 7 rounds): 4 rounds' loads in flight instead of one round trip each
  constexpr int KB = 4;
  for (int base = 0; base < nk; base += KB * T) {
    float m1[KB], s1[KB], m2[KB], s2[KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int idx = min(base + u * T + t, nk - 1);
      m1[u] = mu[idx]; s1[u] = sigma[idx]; m2[u] = pmu[idx]; s2[u] = pstd[idx];
    }
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int idx = base + u * T + t;
      if (idx < nk) {
        const float s1s = s1[u] * s1[u], s2s = s2[u] * s2[u], dm = m1[u] - m2[u];
        kl += (double)(s1s / s2s + dm * dm / s2[u] + logf(s2s) - logf(s1s));
        g_mu[idx] = ck * (2.f * dm / s2[u]);
        g_pmu[idx] = -ck * (2.f * dm / s2[u]);
        g_sigma[idx] = ck * (2.f * s1[u] / s2s - 2.f / s1[u]);
        g_pstd[idx] = ck * (-2.f * s1s / (s2s * s2[u]) - dm * dm / s2s + 2.f / s2[u]);
      }
    }
  }
  kl = block_sum(kl, sh);
  const double kl_val = 0.5 * (kl / (double)n_beads - (double)F);
  // ---- reconstruction MSE
  const int nr = n_atoms * 3;
  double rec = 0.0;
  for (int idx = t; idx < nr; idx += T) {
    const float d = xr[idx] - xyz[idx];
    rec += (double)(d * d);
    if (!bond_sum) g_xr[idx] = 2.f * d / (float)nr;          // (elbo_bond_k has written the complete gradient otherwise)
  }
  rec = block_sum(rec, sh);
  const double rec_val = rec / (double)(nr > 0 ? nr : 1);
  // ---- bond-graph term.  Bonds go through LDS in chunks: each bond's value and its d/d xr_a0 vector are
  // computed once; then every thread owns atoms and scans the chunk for its own id (LDS broadcast reads, bond
  // order ascending: no atomics, fixed summation order).  Scanning the int64 list in global memory instead
  // cost ~40 us of serialized L1 latency on the 340-bond chignolin batch.
  constexpr int CH = 2048;
  __shared__ __attribute__((aligned(16))) int sb_a0[CH], sb_a1[CH];
  __shared__ __attribute__((aligned(16))) float sb_cx[CH], sb_cy[CH], sb_cz[CH];     // d value / d xr_a0
  __shared__ float sb_part[1024 * 3];                      // slices of an atom's bond sum parked by thread groups 1 .. 3 ((G - 1) per <= 1024 threads)
  const bool want_grad = gamma != 0.f && n_bonds > 0;
  const float cg = n_bonds > 0 ? gamma * 2.f / (float)n_bonds : 0.f;
  double gr = 0.0;
  for (int base = 0; base < (bond_sum ? 0 : n_bonds); base += CH) {
    const int cnt = min(CH, n_bonds - base);
    __syncthreads();                       // previous chunk fully scanned; first round: g_xr holds the recon part
    for (int b = t; b < cnt; b += T) {
      const int a0 = (int)bonds[2 * (size_t)(base + b)], a1 = (int)bonds[2 * (size_t)(base + b) + 1];
      const float ex = xr[3 * a0] - xr[3 * a1], ey = xr[3 * a0 + 1] - xr[3 * a1 + 1], ez = xr[3 * a0 + 2] - xr[3 * a1 + 2];
      const float fx = xyz[3 * a0] - xyz[3 * a1], fy = xyz[3 * a0 + 1] - xyz[3 * a1 + 1], fz = xyz[3 * a0 + 2] - xyz[3 * a1 + 2];
      const float lg = sqrtf(1e-6f + ex * ex + ey * ey + ez * ez), ld = sqrtf(1e-6f + fx * fx + fy * fy + fz * fz);
      const float diff = lg - ld;
      gr += (double)(diff * diff);
      const float c = (a0 == a1) ? 0.f : cg * diff / lg;       // self bonds contribute no gradient
      sb_a0[b] = a0; sb_a1[b] = a1;
      sb_cx[b] = c * ex; sb_cy[b] = c * ey; sb_cz[b] = c * ez;
    }
    for (int b = cnt + t; b < ((cnt + 15) & ~15); b += T) { sb_a0[b] = sb_a1[b] = -1; sb_cx[b] = sb_cy[b] = sb_cz[b] = 0.f; }
    __syncthreads();
    if (want_grad) {
      // branch-free scan of the staged bonds for one atom: every read is unconditional so the LDS latency pipelines (a
      // scan that branches on each id waits a full LDS round trip per bond, and the compiler re-introduces those
      // branches if the ids are read one by one); 16 bonds per trip, the 20 reads issued before the first is used
      auto scan = [&](int a, int b_lo, int b_hi, float& gx, float& gy, float& gz) {
        for (int b0 = b_lo; b0 < b_hi; b0 += 16) {
          int4 i0[4], i1[4];
          float4 cx[4], cy[4], cz[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int b = b0 + 4 * u;
            i0[u] = *reinterpret_cast<const int4*>(sb_a0 + b); i1[u] = *reinterpret_cast<const int4*>(sb_a1 + b);
            cx[u] = *reinterpret_cast<const float4*>(sb_cx + b); cy[u] = *reinterpret_cast<const float4*>(sb_cy + b);
            cz[u] = *reinterpret_cast<const float4*>(sb_cz + b);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float w0 = i0[u].x == a ? 1.f : (i1[u].x == a ? -1.f : 0.f);       // d/d xr_a1 = -d/d xr_a0
            const float w1 = i0[u].y == a ? 1.f : (i1[u].y == a ? -1.f : 0.f);
            const float w2 = i0[u].z == a ? 1.f : (i1[u].z == a ? -1.f : 0.f);
            const float w3 = i0[u].w == a ? 1.f : (i1[u].w == a ? -1.f : 0.f);
            gx = fmaf(w3, cx[u].w, fmaf(w2, cx[u].z, fmaf(w1, cx[u].y, fmaf(w0, cx[u].x, gx))));
            gy = fmaf(w3, cy[u].w, fmaf(w2, cy[u].z, fmaf(w1, cy[u].y, fmaf(w0, cy[u].x, gy))));
            gz = fmaf(w3, cz[u].w, fmaf(w2, cz[u].z, fmaf(w1, cz[u].y, fmaf(w0, cz[u].x, gz))));
          }
        }
      };
      // few atoms (chignolin: 332 of 1024 threads): G thread groups take a slice of the bond range each for the same
      // atoms and group 0 adds the slices in group order -- the scan is bound by the LDS reads every wave repeats
      const int G = n_atoms * 2 <= T ? min(4, T / n_atoms) : 1;
      if (G == 1) {
        for (int a = t; a < n_atoms; a += T) {
          float gx = 0.f, gy = 0.f, gz = 0.f;
          scan(a, 0, cnt, gx, gy, gz);
          g_xr[3 * a] += gx; g_xr[3 * a + 1] += gy; g_xr[3 * a + 2] += gz;
        }
      } else {
        const int per = T / G, gi = t / per, a = t - gi * per;
        const int trips = (cnt + 15) / 16, tper = (trips + G - 1) / G;
        const bool mine = gi < G && a < n_atoms;
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (mine) scan(a, 16 * tper * gi, min(cnt, 16 * tper * (gi + 1)), gx, gy, gz);
        if (mine && gi > 0) { float* q = sb_part + ((gi - 1) * per + a) * 3; q[0] = gx; q[1] = gy; q[2] = gz; }
        __syncthreads();
        if (mine && gi == 0) {
          for (int g2 = 1; g2 < G; ++g2) { const float* q = sb_part + ((g2 - 1) * per + a) * 3; gx += q[0]; gy += q[1]; gz += q[2]; }
          g_xr[3 * a] += gx; g_xr[3 * a + 1] += gy; g_xr[3 * a + 2] += gz;
        }
      }
    }
  }
  gr = block_sum(gr, sh);
  if (bond_sum) gr = *bond_sum;
  const double gr_val = n_bonds > 0 ? gr / (double)n_bonds : 0.0;
  if (t == 0) {
    out[0] = (float)(rec_val + (double)beta * kl_val + (double)gamma * gr_val);
    if (loss_out) loss_out[0] = out[0];
    out[1] = (float)kl_val;
    out[2] = (float)rec_val;
    out[3] = gamma != 0.f ? (float)gr_val : 0.f;          // utils.py:134-135: zero when gamma == 0
  }
}

// grads *= upstream scalar (device); all five tensors in one launch
__global__ __launch_bounds__(256) void elbo_scale(const float* __restrict__ g_loss, float* g0, float* g1, float* g2,
                                                  float* g3, int nk, float* g4, int nr) {
  const float s = *g_loss;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < nk) { g0[idx] *= s; g1[idx] *= s; g2[idx] *= s; g3[idx] *= s; }
  if (idx < nr) g4[idx] *= s;
}

}  // namespace cgv

extern "C" {

// KL elements beyond which the KL terms run on their own multi-block launch; partial sums in the workspace
static inline int elbo_kl_blocks(int n_beads, int n_feat) {
  const long long nk = (long long)n_beads * n_feat;
  if (nk < 16384) return 0;
  const long long b = (nk + 1023) / 1024;
  return (int)(b > 128 ? 128 : b);
}

// one double per KL block + one for the bond sum of the multi-block bond kernel
size_t cgv_elbo_workspace_bytes(int n_beads, int n_feat) { return sizeof(double) * ((size_t)elbo_kl_blocks(n_beads, n_feat) + 1); }

int cgv_elbo_fwd(const float* mu, const float* sigma, const float* prior_mu, const float* prior_std, const float* xyz,
                 const float* xyz_recon, const int64_t* bonds, int n_beads, int n_feat, int n_atoms, int n_bonds,
                 float beta, float gamma, float* out4, float* loss_out, float* g_mu, float* g_sigma, float* g_prior_mu,
                 float* g_prior_std, float* g_xyz_recon, void* workspace, size_t workspace_bytes, void* stream) {
  CGV_REQUIRE(mu && sigma && prior_mu && prior_std && xyz && xyz_recon && out4, "null input");
  CGV_REQUIRE(g_mu && g_sigma && g_prior_mu && g_prior_std && g_xyz_recon, "null gradient buffer");
  CGV_REQUIRE(n_beads > 0 && n_feat > 0 && n_atoms > 0 && n_bonds >= 0 && (n_bonds == 0 || bonds), "bad size");
  hipStream_t st = (hipStream_t)stream;
  int nb = elbo_kl_blocks(n_beads, n_feat);
  const bool ws_ok = workspace && workspace_bytes >= sizeof(double) * ((size_t)nb + 1) && (((uintptr_t)workspace) & 7) == 0;
  if (!ws_ok) nb = 0;
  double* part = reinterpret_cast<double*>(workspace);
  // atoms x bonds beyond what one block scans in a few microseconds: the bond term on its own multi-block launch
  double* bond_sum = nullptr;
  if (ws_ok && gamma != 0.f && n_bonds > 0 && (long long)n_atoms * n_bonds >= 512LL * 1024) {
    bond_sum = part + nb;
    hipLaunchKernelGGL(cgv::elbo_bond_k, dim3((n_atoms + cgv::BOND_ATOMS - 1) / cgv::BOND_ATOMS), dim3(1024), 0, st, xyz,
                       xyz_recon, bonds, n_atoms, n_bonds, gamma, g_xyz_recon, bond_sum);
  }
  if (nb > 0)
    hipLaunchKernelGGL(cgv::elbo_kl_k, dim3(nb), dim3(256), 0, st, mu, sigma, prior_mu, prior_std, n_beads * n_feat,
                       0.5f * beta / (float)n_beads, g_mu, g_sigma, g_prior_mu, g_prior_std, part);
  hipLaunchKernelGGL(cgv::elbo_fwd, dim3(1), dim3(1024), 0, st, mu, sigma, prior_mu, prior_std, xyz, xyz_recon, bonds,
                     n_beads, n_feat, n_atoms, n_bonds, beta, gamma, out4, loss_out, g_mu, g_sigma, g_prior_mu, g_prior_std,
                     g_xyz_recon, (const double*)part, nb, (const double*)bond_sum);
  return cgv::check_launch("cgv_elbo_fwd");
}

int cgv_elbo_scale(const float* g_loss, float* g_mu, float* g_sigma, float* g_prior_mu, float* g_prior_std, int n_bead_elems,
                   float* g_xyz_recon, int n_atom_elems, void* stream) {
  CGV_REQUIRE(g_loss && g_mu && g_sigma && g_prior_mu && g_prior_std && g_xyz_recon, "null pointer");
  const int n = n_bead_elems > n_atom_elems ? n_bead_elems : n_atom_elems;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cgv::elbo_scale, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, g_loss, g_mu, g_sigma,
                     g_prior_mu, g_prior_std, n_bead_elems, g_xyz_recon, n_atom_elems);
  return cgv::check_launch("cgv_elbo_scale");
}

}  // extern "C"
