// Decoder tail + ELBO + their gradients in ONE launch (reference: cgvae.py:462-481 CGequiVAE.decoder's tail,
// scripts/utils.py:81-86 KL, 117-141 loss):
//   xyz_rel[a] = V[bead(a), chan[a], :] ; xyz_rel -= mean over the bead (offset) ; xyz_recon = xyz_rel + cg_xyz[bead(a)]
//   KL, recon, graph, loss as in elbo.hip
//   d loss / d{mu, sigma, prior_mu, prior_std}, d loss / d xyz_recon, and d loss / d V (the backward of the tail:
//   g_V[b, chan[a], :] = g_xr[a] - mean_b(g_xr), zero elsewhere)
// As three launches (reconstruct_fwd, elbo_fwd -- one 1024-thread block --, reconstruct_bwd) this was 31.7 us of the
// chignolin step and 52 us of the dipeptide step, almost all of it dependent round trips of a single block.
//
// One block per BEAD.  Every block recomputes the reconstructed coordinates of ALL atoms into LDS (a bond's partner atom
// may sit in any bead of the frame; 12 bytes per atom, a few thousand atoms), scans the bond list for the atoms of its
// own bead, and owns the KL terms of its bead's F channels.  The three partial sums of a block leave as doubles; the
// block that arrives last (device-scope ticket) adds them in block order -- deterministic -- and writes the scalars.
#include "cgv_common.h"
// The last-block hand-over below (relaxed agent-scope stores, an explicit `s_waitcnt vmcnt(0)`, then a relaxed ticket
// atomic) relies on stores being counted by vmcnt -- true on the gfx9 family this library is written for, not part of the
// HIP memory model.  Refuse to build for anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "ticket hand-over ordered by s_waitcnt vmcnt(0): gfx942 / gfx950 only"
#endif

namespace cgv {

constexpr int LT_THREADS = 1024;
constexpr int LT_NP = 2;                    // register slots per thread of the one-batch fast path: sizes up to 2048
constexpr int LT_CH = 2048;                 // bonds staged per chunk
constexpr int LT_MAX_ATOMS = 4096;          // 2 x 48 KB of LDS for the coordinates (reconstructed + data)
constexpr int LT_MAX_BEADS = 2048;
constexpr int LT_SLOTS = 32;                // atoms of the block's bead scanned per pass (32 thread groups split the bonds)
constexpr int LT_GROUPS = LT_THREADS / LT_SLOTS;

__device__ __forceinline__ void lt_block_sum3(double& a, double& b, double& c, double* sh /*[3][16]*/) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); c += __shfl_xor(c, d); }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sh[w] = a; sh[16 + w] = b; sh[32 + w] = c; }
  __syncthreads();
  double ta = 0.0, tb = 0.0, tc = 0.0;
  for (int k = 0; k < LT_THREADS / 64; ++k) { ta += sh[k]; tb += sh[16 + k]; tc += sh[32 + k]; }
  a = ta; b = tb; c = tc;
}

// agent-scope (write-through / L2-bypassing) accesses for the hand-over of the partial sums to the last block
__device__ __forceinline__ void lt_store_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double lt_load_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

extern __shared__ __attribute__((aligned(16))) unsigned char lt_smem[];

// The kernel is a chain of dependent memory round trips on a handful of blocks, so what matters is how FEW there are:
// every load that depends on nothing is requested in the first batch (KL operands, the bead-sorted atom / bead ids, the
// bond list, the bead centres), the second batch takes what depends on an id (channel index, data coordinates), the third
// the gather of V; everything after that runs on LDS.  Sizes beyond LT_NP x 1024 take the same statements as loops.
__global__ __launch_bounds__(LT_THREADS) void loss_tail_k(
    const float* __restrict__ V, const float* __restrict__ cg_xyz, const int* __restrict__ rowptr,
    const int* __restrict__ atom_of, const int* __restrict__ bead_of, const int64_t* __restrict__ chan,
    const float* __restrict__ mu, const float* __restrict__ sigma, const float* __restrict__ pmu, const float* __restrict__ pstd,
    const float* __restrict__ xyz, const int64_t* __restrict__ bonds, int n_beads, int F, int n_atoms, int n_bonds, int offset,
    float beta, float gamma, float* __restrict__ xr_out, float* __restrict__ out /*[4]*/, float* __restrict__ loss_out,
    float* __restrict__ g_mu, float* __restrict__ g_sigma, float* __restrict__ g_pmu, float* __restrict__ g_pstd,
    float* __restrict__ g_xr, float* __restrict__ g_V, double* __restrict__ part /*[3][n_beads]*/, unsigned int* __restrict__ ticket) {
  __shared__ double sh[48];
  __shared__ int sb_a0[LT_CH], sb_a1[LT_CH];
  __shared__ float sb_cx[LT_CH], sb_cy[LT_CH], sb_cz[LT_CH];
  __shared__ float sp[LT_GROUPS][LT_SLOTS][3];
  __shared__ float gsum[3];
  __shared__ unsigned int s_last;
  float* xr = reinterpret_cast<float*>(lt_smem);                   // [n_atoms][3] reconstructed coordinates
  float* xd = xr + 3 * (size_t)n_atoms;                              // [n_atoms][3] data coordinates
  float* mean = xd + 3 * (size_t)n_atoms;                            // [n_beads][3]
  float* cgl = mean + 3 * (size_t)n_beads;                           // [n_beads][3] bead centres
  int* atom_l = reinterpret_cast<int*>(cgl + 3 * (size_t)n_beads);   // [n_atoms] atom id of bead-sorted position p
  int* chan_l = atom_l + n_atoms;                                    // [n_atoms] channel of atom a
  const int b = blockIdx.x, t = threadIdx.x, T = LT_THREADS;

  // ---- batch 1: everything that depends on nothing
  float k_m1[LT_NP], k_s1[LT_NP], k_m2[LT_NP], k_s2[LT_NP];
  int p_atom[LT_NP], p_bead[LT_NP];
  int bd_a0[LT_NP], bd_a1[LT_NP];
#pragma unroll
  for (int u = 0; u < LT_NP; ++u) {
    const int f = min(t + u * T, F - 1);
    const size_t idx = (size_t)b * F + f;
    k_m1[u] = mu[idx]; k_s1[u] = sigma[idx]; k_m2[u] = pmu[idx]; k_s2[u] = pstd[idx];
    const int p = min(t + u * T, n_atoms - 1);
    p_atom[u] = atom_of[p]; p_bead[u] = bead_of[p];
    const int k = min(t + u * T, max(n_bonds, 1) - 1);
    bd_a0[u] = n_bonds > 0 ? (int)bonds[2 * (size_t)k] : 0;
    bd_a1[u] = n_bonds > 0 ? (int)bonds[2 * (size_t)k + 1] : 0;
  }
  for (int k = t; k < 3 * n_beads; k += T) cgl[k] = cg_xyz[k];
  const int beg = rowptr[b], end = rowptr[b + 1], n_own = end - beg;
  // ---- batch 2: what depends on an atom id
  int p_chan[LT_NP];
  f3 p_x[LT_NP];
#pragma unroll
  for (int u = 0; u < LT_NP; ++u) { p_chan[u] = (int)chan[p_atom[u]]; p_x[u] = ld3(xyz + 3 * (size_t)p_atom[u]); }
  // ---- batch 3: the gather of V (cgvae.py:470-475)
  f3 p_v[LT_NP];
#pragma unroll
  for (int u = 0; u < LT_NP; ++u) p_v[u] = ld3(V + ((size_t)p_bead[u] * F + (size_t)p_chan[u]) * 3);

  // ---- KL terms of this bead's F channels (scripts/utils.py:81-86, the (mu1 - mu2)^2 / std2 of the source included)
  const float ck = 0.5f * beta / (float)n_beads;
  double kl = 0.0;
  auto kl_elem = [&](int f, float m1, float s1, float m2, float s2) {
    const size_t idx = (size_t)b * F + f;
    const float s1s = s1 * s1, s2s = s2 * s2, dm = m1 - m2;
    kl += (double)(s1s / s2s + dm * dm / s2 + logf(s2s) - logf(s1s));
    g_mu[idx] = ck * (2.f * dm / s2);
    g_pmu[idx] = -ck * (2.f * dm / s2);
    g_sigma[idx] = ck * (2.f * s1 / s2s - 2.f / s1);
    g_pstd[idx] = ck * (-2.f * s1s / (s2s * s2) - dm * dm / s2s + 2.f / s2);
  };
#pragma unroll
  for (int u = 0; u < LT_NP; ++u)
    if (t + u * T < F) kl_elem(t + u * T, k_m1[u], k_s1[u], k_m2[u], k_s2[u]);
  for (int f = t + LT_NP * T; f < F; f += T) { const size_t idx = (size_t)b * F + f; kl_elem(f, mu[idx], sigma[idx], pmu[idx], pstd[idx]); }

  // ---- reconstructed coordinates of ALL atoms, bead-sorted position p -> atom atom_of[p]
#pragma unroll
  for (int u = 0; u < LT_NP; ++u)
    if (t + u * T < n_atoms) {
      const int a = p_atom[u];
      xr[3 * a] = p_v[u].x; xr[3 * a + 1] = p_v[u].y; xr[3 * a + 2] = p_v[u].z;
      xd[3 * a] = p_x[u].x; xd[3 * a + 1] = p_x[u].y; xd[3 * a + 2] = p_x[u].z;
      atom_l[t + u * T] = a; chan_l[a] = p_chan[u];
    }
  for (int p = t + LT_NP * T; p < n_atoms; p += T) {
    const int a = atom_of[p], bd = bead_of[p], ch = (int)chan[a];
    const f3 r = ld3(V + ((size_t)bd * F + (size_t)ch) * 3), x0 = ld3(xyz + 3 * (size_t)a);
    xr[3 * a] = r.x; xr[3 * a + 1] = r.y; xr[3 * a + 2] = r.z;
    xd[3 * a] = x0.x; xd[3 * a + 1] = x0.y; xd[3 * a + 2] = x0.z;
    atom_l[p] = a; chan_l[a] = ch;
  }
  __syncthreads();
  // bead means: a wave per bead, lanes stride over its atoms (LDS only), fixed butterfly order
  {
    const int lane = t & 63, w = t >> 6;
    for (int m = w; m < n_beads; m += T / 64) {
      const int mb = rowptr[m], me = rowptr[m + 1];
      float sx = 0.f, sy = 0.f, sz = 0.f;
      if (offset)
        for (int p = mb + lane; p < me; p += 64) { const int a = atom_l[p]; sx += xr[3 * a]; sy += xr[3 * a + 1]; sz += xr[3 * a + 2]; }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) { sx += __shfl_xor(sx, d); sy += __shfl_xor(sy, d); sz += __shfl_xor(sz, d); }
      if (lane == 0) {
        const float inv = 1.0f / (float)max(me - mb, 1);
        mean[3 * m] = sx * inv; mean[3 * m + 1] = sy * inv; mean[3 * m + 2] = sz * inv;
      }
    }
  }
  __syncthreads();
  auto shift = [&](int a, int bd) {
    xr[3 * a] = (xr[3 * a] - mean[3 * bd]) + cgl[3 * bd];
    xr[3 * a + 1] = (xr[3 * a + 1] - mean[3 * bd + 1]) + cgl[3 * bd + 1];
    xr[3 * a + 2] = (xr[3 * a + 2] - mean[3 * bd + 2]) + cgl[3 * bd + 2];
  };
#pragma unroll
  for (int u = 0; u < LT_NP; ++u)
    if (t + u * T < n_atoms) shift(p_atom[u], p_bead[u]);
  for (int p = t + LT_NP * T; p < n_atoms; p += T) shift(atom_l[p], bead_of[p]);
  __syncthreads();

  // ---- this bead's atoms: output coordinates, reconstruction term
  const int nr = 3 * n_atoms;
  double rec = 0.0;
  for (int p = beg + t; p < end; p += T) {
    const int a = atom_l[p];
    const float dx = xr[3 * a] - xd[3 * a], dy = xr[3 * a + 1] - xd[3 * a + 1], dz = xr[3 * a + 2] - xd[3 * a + 2];
    st3(xr_out + 3 * (size_t)a, xr[3 * a], xr[3 * a + 1], xr[3 * a + 2]);
    rec += (double)(dx * dx) + (double)(dy * dy) + (double)(dz * dz);
  }

  // ---- bond-graph term (scripts/utils.py:127-133): every block stages every bond (value + d / d xr_a0), counts the
  // bonds k = block (mod gridDim) towards the sum, and scans the chunk for the atoms of its own bead
  const bool want_grad = gamma != 0.f && n_bonds > 0;
  const float cg = n_bonds > 0 ? gamma * 2.f / (float)n_bonds : 0.f;
  const int slot = t & (LT_SLOTS - 1), grp = t / LT_SLOTS;
  double gr = 0.0;
  const float sc = 2.f / (float)nr;
  for (int s0 = 0; s0 < max(n_own, 1); s0 += LT_SLOTS) {             // passes of 32 own atoms (one pass unless the bead is large)
    const int a = (s0 + slot < n_own) ? atom_l[beg + s0 + slot] : -2;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int base = 0; base < n_bonds; base += LT_CH) {
      const int cnt = min(LT_CH, n_bonds - base);
      __syncthreads();
      for (int k = t, u = 0; k < cnt; k += T, ++u) {
        int a0, a1;
        if (base == 0 && u < LT_NP) { a0 = u == 0 ? bd_a0[0] : bd_a0[LT_NP - 1]; a1 = u == 0 ? bd_a1[0] : bd_a1[LT_NP - 1]; }   // (LT_NP == 2)
        else { a0 = (int)bonds[2 * (size_t)(base + k)]; a1 = (int)bonds[2 * (size_t)(base + k) + 1]; }
        const float ex = xr[3 * a0] - xr[3 * a1], ey = xr[3 * a0 + 1] - xr[3 * a1 + 1], ez = xr[3 * a0 + 2] - xr[3 * a1 + 2];
        const float fx = xd[3 * a0] - xd[3 * a1], fy = xd[3 * a0 + 1] - xd[3 * a1 + 1], fz = xd[3 * a0 + 2] - xd[3 * a1 + 2];
        const float lg = sqrtf(1e-6f + ex * ex + ey * ey + ez * ez), ld = sqrtf(1e-6f + fx * fx + fy * fy + fz * fz);
        const float diff = lg - ld;
        if (s0 == 0 && (base + k) % (int)gridDim.x == b) gr += (double)(diff * diff);
        const float c = (a0 == a1) ? 0.f : cg * diff / lg;       // self bonds contribute no gradient
        sb_a0[k] = a0; sb_a1[k] = a1;
        sb_cx[k] = c * ex; sb_cy[k] = c * ey; sb_cz[k] = c * ez;
      }
      __syncthreads();
      if (want_grad && a >= 0) {
        const int per = (cnt + LT_GROUPS - 1) / LT_GROUPS, k_lo = grp * per, k_hi = min(cnt, k_lo + per);
        for (int k = k_lo; k < k_hi; ++k) {
          const float w = sb_a0[k] == a ? 1.f : (sb_a1[k] == a ? -1.f : 0.f);     // d / d xr_a1 = - d / d xr_a0
          gx = fmaf(w, sb_cx[k], gx); gy = fmaf(w, sb_cy[k], gy); gz = fmaf(w, sb_cz[k], gz);
        }
      }
    }
    sp[grp][slot][0] = gx; sp[grp][slot][1] = gy; sp[grp][slot][2] = gz;
    __syncthreads();
    if (grp == 0 && a >= 0) {
      float tx = 0.f, ty = 0.f, tz = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < LT_GROUPS; ++g2) { tx += sp[g2][slot][0]; ty += sp[g2][slot][1]; tz += sp[g2][slot][2]; }
      // complete gradient of the loss w.r.t. this atom's reconstructed coordinates
      tx += sc * (xr[3 * a] - xd[3 * a]); ty += sc * (xr[3 * a + 1] - xd[3 * a + 1]); tz += sc * (xr[3 * a + 2] - xd[3 * a + 2]);
      st3(g_xr + 3 * (size_t)a, tx, ty, tz);
      // parked over the DATA coordinates of the atom: nothing reads xd of an own atom after its pass except the bond
      // staging of a LATER pass -- which is why the park happens below, once no later pass remains (single-pass beads),
      // and through global memory otherwise
      sp[0][slot][0] = tx; sp[0][slot][1] = ty; sp[0][slot][2] = tz;
    }
    __syncthreads();
    if (s0 == 0 && t < 3) gsum[t] = 0.f;
    __syncthreads();
    if (t == 0) {                                                    // fixed order: deterministic bead sums
      float x = gsum[0], y = gsum[1], z = gsum[2];
      for (int k = 0; k < min(LT_SLOTS, n_own - s0); ++k) { x += sp[0][k][0]; y += sp[0][k][1]; z += sp[0][k][2]; }
      gsum[0] = x; gsum[1] = y; gsum[2] = z;
    }
    __syncthreads();
  }

  // ---- backward of the tail: g_V[b, chan[a], :] = g_xr[a] - mean_b(g_xr) (offset), zero elsewhere
  float* gb = g_V + (size_t)b * F * 3;
  for (int k = t; k < 3 * F; k += T) gb[k] = 0.f;
  __syncthreads();
  {
    const float inv = offset ? 1.0f / (float)max(n_own, 1) : 0.f;
    const float mx = gsum[0] * inv, my = gsum[1] * inv, mz = gsum[2] * inv;
    if (n_own <= LT_SLOTS) {                                         // single pass: the gradients are still in LDS
      if (t < n_own) {
        const int a = atom_l[beg + t];
        st3(gb + 3 * (size_t)chan_l[a], sp[0][t][0] - mx, sp[0][t][1] - my, sp[0][t][2] - mz);
      }
    } else {
      for (int p = beg + t; p < end; p += T) {
        const int a = atom_l[p];
        const f3 g = ld3(g_xr + 3 * (size_t)a);                      // written by this block above (after barriers)
        st3(gb + 3 * (size_t)chan_l[a], g.x - mx, g.y - my, g.z - mz);
      }
    }
  }

  // ---- partial sums -> last block
  lt_block_sum3(kl, rec, gr, sh);
  if (t == 0) {
    lt_store_agent(part + b, kl);
    lt_store_agent(part + n_beads + b, rec);
    lt_store_agent(part + 2 * n_beads + b, gr);
    // (no __threadfence: an agent-scope release writes back the XCD's whole L2; the partials are agent-scope atomics --
    // coherent across XCDs by themselves -- and acknowledged before the ticket)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  double k2 = 0.0, r2 = 0.0, g2 = 0.0;
  for (int m = t; m < n_beads; m += T) { k2 += lt_load_agent(part + m); r2 += lt_load_agent(part + n_beads + m); g2 += lt_load_agent(part + 2 * n_beads + m); }
  lt_block_sum3(k2, r2, g2, sh);
  if (t == 0) {
    const double kl_val = 0.5 * (k2 / (double)n_beads - (double)F);
    const double rec_val = r2 / (double)(nr > 0 ? nr : 1);
    const double gr_val = n_bonds > 0 ? g2 / (double)n_bonds : 0.0;
    out[0] = (float)(rec_val + (double)beta * kl_val + (double)gamma * gr_val);
    if (loss_out) loss_out[0] = out[0];
    out[1] = (float)kl_val;
    out[2] = (float)rec_val;
    out[3] = gamma != 0.f ? (float)gr_val : 0.f;                    // utils.py:134-135: zero when gamma == 0
    *ticket = 0u;                                                    // the next launch (replay) starts from zero again
  }
}

}  // namespace cgv

extern "C" {

int cgv_loss_tail_supported(int n_beads, int n_feat, int n_atoms, int n_bonds) {
  /* LDS: ~53 KB static (bond staging) + 32 bytes per atom (reconstructed + data coordinates, id, channel) + 24 per bead */
  return n_beads >= 1 && n_beads <= cgv::LT_MAX_BEADS && n_atoms >= 1 && n_atoms <= cgv::LT_MAX_ATOMS &&
         (long long)n_atoms + n_beads <= 3072 && n_feat >= 1 && n_bonds >= 0;
}

/* doubles for the per-bead partial sums + the ticket word (must be ZERO before the first launch; every launch leaves it zero) */
size_t cgv_loss_tail_workspace_bytes(int n_beads) { return sizeof(double) * 3 * (size_t)n_beads + 16; }

int cgv_loss_tail(const float* V, const float* cg_xyz, const int32_t* rowptr, const int32_t* atom_of, const int32_t* bead_of,
                  const int64_t* chan, const float* mu, const float* sigma, const float* prior_mu, const float* prior_std,
                  const float* xyz, const int64_t* bonds, int n_beads, int n_feat, int n_atoms, int n_bonds, int offset,
                  float beta, float gamma, float* xyz_recon, float* out4, float* loss_out, float* g_mu, float* g_sigma,
                  float* g_prior_mu, float* g_prior_std, float* g_xyz_recon, float* g_V, void* workspace, size_t workspace_bytes,
                  void* stream) {
  CGV_REQUIRE(V && cg_xyz && rowptr && atom_of && bead_of && chan && mu && sigma && prior_mu && prior_std && xyz, "null input");
  CGV_REQUIRE(xyz_recon && out4 && g_mu && g_sigma && g_prior_mu && g_prior_std && g_xyz_recon && g_V && workspace, "null output");
  CGV_REQUIRE(cgv_loss_tail_supported(n_beads, n_feat, n_atoms, n_bonds) && (n_bonds == 0 || bonds), "unsupported size");
  CGV_REQUIRE(workspace_bytes >= cgv_loss_tail_workspace_bytes(n_beads) && (((uintptr_t)workspace) & 15) == 0, "workspace");
  double* part = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(workspace) + 16);
  unsigned int* ticket = reinterpret_cast<unsigned int*>(workspace);
  const size_t lds = sizeof(float) * (3 * (2 * (size_t)n_atoms + 2 * (size_t)n_beads) + 2 * (size_t)n_atoms);
  if (lds > 8 * 1024) {                  // beyond the default dynamic limit next to the kernel's ~53 KB of static LDS
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cgv::loss_tail_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { cgv::set_error("hipFuncSetAttribute(%zu bytes of LDS): %s", lds, hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(cgv::loss_tail_k, dim3(n_beads), dim3(cgv::LT_THREADS), lds, (hipStream_t)stream, V, cg_xyz, rowptr, atom_of,
                     bead_of, chan, mu, sigma, prior_mu, prior_std, xyz, bonds, n_beads, n_feat, n_atoms, n_bonds, offset, beta,
                     gamma, xyz_recon, out4, loss_out, g_mu, g_sigma, g_prior_mu, g_prior_std, g_xyz_recon, g_V, part, ticket);
  return cgv::check_launch("cgv_loss_tail");
}

}  // extern "C"
