// Reparametrisation with the noise drawn in the same launch: z = mu + sigma * eps, eps ~ N(0, 1)
// (reference: CGequiVAE.reparametrize cgvae.py:445-449 -- torch.randn_like + mul + add).
//
// As tensor ops inside a captured step this is five launches: the generator's two graph-safe offset fills, normal_,
// and the product / sum (addcmul).  Here: one.  The generator is Philox4x32-10 (the counter-based generator torch's
// device RNG is built on; Salmon et al., SC'11) keyed by a 64-bit seed, counter = (draw number, element quad);
// Box-Muller turns its four 32-bit words into four normals.  The draw number lives in device memory and is advanced by
// the launch itself -- every block reads it before it takes a ticket, the block that takes the last ticket advances it
// -- so a replayed hipGraph draws fresh noise every step with no host involvement.  The noise is stored: the backward
// pass needs it (d z / d sigma = eps).  Streams differ from torch.randn's (same distribution, different numbers);
// parity runs supply their eps from the host and do not come through here.
#include "cgv_common.h"

namespace cgv {

__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
  const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
  c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// two normals from two 32-bit words (Box-Muller; u1 in (0, 1], 24 bits each)
__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float& n0, float& n1) {
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  n0 = r * cs; n1 = r * sn;
}

// rng: [0] seed, [1] draw number, [2] ticket (low word)
__global__ __launch_bounds__(256) void reparam_sample_k(const float* __restrict__ mu, const float* __restrict__ sigma,
                                                        float* __restrict__ eps, float* __restrict__ z, int n,
                                                        unsigned long long* __restrict__ rng) {
  const unsigned long long seed = rng[0], draw = rng[1];
  const int quad = blockIdx.x * 256 + threadIdx.x;
  unsigned c[4] = {(unsigned)quad, 0u, (unsigned)draw, (unsigned)(draw >> 32)};
  philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
  float e[4];
  box_muller(c[0], c[1], e[0], e[1]);
  box_muller(c[2], c[3], e[2], e[3]);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = 4 * quad + t;
    if (i < n) {
      eps[i] = e[t];
      z[i] = fmaf(e[t], sigma[i], mu[i]);
    }
  }
  __syncthreads();                                               // every thread of the block has read the draw number
  if (threadIdx.x == 0) {
    __threadfence();
    unsigned* ticket = reinterpret_cast<unsigned*>(rng + 2);
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {                // every block has read it: advance, re-arm
      *ticket = 0u;
      rng[1] = draw + 1ull;
    }
  }
}

// backward of the above with the KL term's gradients folded in: g_mu = g + k_mu, g_sigma = g * eps + k_sigma
// (k_* = d(beta KL)/d{mu, sigma} from the ELBO launch; as separate autograd contributions they cost a mul and two adds)
__global__ __launch_bounds__(256) void reparam_bwd_k(const float4* __restrict__ g, const float4* __restrict__ eps,
                                                     const float4* __restrict__ k_mu, const float4* __restrict__ k_sigma,
                                                     float4* __restrict__ g_mu, float4* __restrict__ g_sigma, int n4) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 a = g[i], e = eps[i], km = k_mu[i], ks = k_sigma[i];
  g_mu[i] = make_float4(a.x + km.x, a.y + km.y, a.z + km.z, a.w + km.w);
  g_sigma[i] = make_float4(__fadd_rn(__fmul_rn(a.x, e.x), ks.x), __fadd_rn(__fmul_rn(a.y, e.y), ks.y),
                           __fadd_rn(__fmul_rn(a.z, e.z), ks.z), __fadd_rn(__fmul_rn(a.w, e.w), ks.w));
}

}  // namespace cgv

extern "C" {

/* g_mu = g + k_mu, g_sigma = g * eps + k_sigma over n floats (n % 4 == 0, 16-byte aligned): the backward of
 * cgv_reparam_sample with the KL gradients of the ELBO launch (cgv_elbo_fwd's g_mu / g_sigma) added in the same launch. */
int cgv_reparam_bwd(const float* g, const float* eps, const float* k_mu, const float* k_sigma, float* g_mu, float* g_sigma,
                    int64_t n, void* stream) {
  CGV_REQUIRE(n >= 0 && n < (1ll << 31) && (n % 4) == 0, "bad size (need n % 4 == 0)");
  if (n == 0) return 0;
  CGV_REQUIRE(g && eps && k_mu && k_sigma && g_mu && g_sigma, "null pointer");
  CGV_REQUIRE(((((uintptr_t)g | (uintptr_t)eps | (uintptr_t)k_mu | (uintptr_t)k_sigma | (uintptr_t)g_mu | (uintptr_t)g_sigma)) & 15) == 0,
              "16-byte alignment");
  const int n4 = (int)(n / 4);
  hipLaunchKernelGGL(cgv::reparam_bwd_k, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(g), reinterpret_cast<const float4*>(eps), reinterpret_cast<const float4*>(k_mu),
                     reinterpret_cast<const float4*>(k_sigma), reinterpret_cast<float4*>(g_mu), reinterpret_cast<float4*>(g_sigma), n4);
  return cgv::check_launch("cgv_reparam_bwd");
}

/* z = mu + sigma * eps with eps ~ N(0, 1) drawn in the launch and stored (n floats each).  rng: 3 x uint64 in device
 * memory {seed, draw number, 0}; the launch advances the draw number by one.  Launches that share an rng block must be
 * stream ordered. */
int cgv_reparam_sample(const float* mu, const float* sigma, float* eps, float* z, int64_t n, uint64_t* rng, void* stream) {
  CGV_REQUIRE(n >= 0 && n < (1ll << 31), "bad size");
  if (n == 0) return 0;
  CGV_REQUIRE(mu && sigma && eps && z && rng, "null pointer");
  const int quads = (int)((n + 3) / 4);
  hipLaunchKernelGGL(cgv::reparam_sample_k, dim3((quads + 255) / 256), dim3(256), 0, (hipStream_t)stream, mu, sigma, eps, z,
                     (int)n, reinterpret_cast<unsigned long long*>(rng));
  return cgv::check_launch("cgv_reparam_sample");
}

}  // extern "C"
