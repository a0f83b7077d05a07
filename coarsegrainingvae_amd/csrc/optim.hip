// Fused optimiser step over a flat parameter arena: global-norm clip + Adam in three launches,
// with the clip coefficient, the bias corrections, the step counter and the skip decision kept
// on the device (no host round trip, graph-capturable).
//
// Replaces, for the parameters that receive gradients (reference scripts/utils.py:145-157):
//     if loss >= 200*gamma or isnan(loss): skip
//     clip_grad_norm_(params, 0.01)  ->  g *= min(1, max_norm / (||g||_2 + 1e-6))
//     Adam.step()                    ->  torch.optim.Adam defaults (no amsgrad / weight decay)
// Traffic per step: g twice (norm + update), p/m/v read + written once: 9 floats per parameter
// instead of ~17 for separate clip (read, read+write) and multi-pass foreach Adam.
#include <stdlib.h>
#include "cgv_common.h"
// The last-block hand-over below (relaxed agent-scope stores, an explicit `s_waitcnt vmcnt(0)`, then a relaxed ticket
// atomic) relies on stores being counted by vmcnt -- true on the gfx9 family this library is written for, not part of the
// HIP memory model.  Refuse to build for anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "ticket hand-over ordered by s_waitcnt vmcnt(0): gfx942 / gfx950 only"
#endif

namespace cgv {

__device__ inline float wave_sum(float x) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
  return x;
}

__global__ __launch_bounds__(256) void sumsq_partial(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
  __shared__ float ws[4];
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 x = g4[i];
    acc = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, acc))));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    float x = g[(n4 << 2) + threadIdx.x];
    acc = fmaf(x, x, acc);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// The decision pass proper (one block's worth of work), shared by the stand-alone launch and by the LAST block of
// sumsq_decide_k.  ``partial`` may have been written by other XCDs in this same launch: read with agent scope there.
template <bool AGENT>
__device__ __forceinline__ void optim_decide(const float* __restrict__ partial, int nb, const double* __restrict__ extra, int n_extra,
                                             float grad_scale, float max_norm, float beta1, float beta2,
                                             const float* __restrict__ loss, float skip_threshold, float* __restrict__ state,
                                             double* ws /*[4] LDS*/) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x)
    acc += (double)(AGENT ? __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : partial[i]);
  for (int i = threadIdx.x; i < n_extra; i += blockDim.x) acc += extra[i];     // ||gW||^2 of never-materialised gradients
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double total = (ws[0] + ws[1]) + (ws[2] + ws[3]);
  const float norm = (float)sqrt(total) * fabsf(grad_scale);
  bool skip = false;
  if (loss) {
    const float l = *loss;
    skip = (l >= skip_threshold) || (l != l);                 // utils.py:145
  }
  state[ST_NORM] = norm;
  state[ST_SKIP] = skip ? 1.f : 0.f;
  if (skip) {
    state[ST_NSKIPPED] += 1.f;
    return;
  }
  float coef = max_norm / (norm + 1e-6f);                      // torch.nn.utils.clip_grad_norm_
  if (coef > 1.f) coef = 1.f;
  state[ST_CLIP] = coef * grad_scale;
  const double step = (double)state[ST_STEP] + 1.0;
  state[ST_STEP] = (float)step;
  state[ST_BC1] = (float)(1.0 - pow((double)beta1, step));
  state[ST_BC2SQRT] = (float)sqrt(1.0 - pow((double)beta2, step));
}

// Norm pass AND decision in one launch: every block leaves its partial sum (write-through), the block that arrives last
// at the ticket (device scope) runs the decision pass over all partials in block order -- the same sums in the same
// order as sumsq_partial + optim_finalize, one launch boundary less.  ticket: one word, zero before the first launch; the
// last block leaves it zero.  NOT the default: with the 1620 blocks the norm pass wants, the arrivals at the one ticket
// word (~12 ns each, MI355X_MICROARCH.md "fanin") cost more than the boundary: 1.774 against 1.763 ms per chignolin step.
// (The same pattern pays where few blocks arrive: csrc/loss_tail.hip, 12 - 96 blocks.)
__global__ __launch_bounds__(256) void sumsq_decide_k(const float* __restrict__ g, int64_t n, float* __restrict__ partial,
                                                      unsigned int* __restrict__ ticket, const double* __restrict__ extra, int n_extra,
                                                      float grad_scale, float max_norm, float beta1, float beta2,
                                                      const float* __restrict__ loss, float skip_threshold, float* __restrict__ state) {
  __shared__ float ws[4];
  __shared__ double wd[4];
  __shared__ unsigned int s_last;
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 x = g4[i];
    acc = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, acc))));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    float x = g[(n4 << 2) + threadIdx.x];
    acc = fmaf(x, x, acc);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    // (no __threadfence: an agent-scope release writes back the XCD's whole L2 -- 30-85 us behind the backward pass; the
    // partial travels as an agent-scope atomic, coherent across the XCDs by itself, and is acknowledged before the ticket)
    __hip_atomic_store(partial + blockIdx.x, (ws[0] + ws[1]) + (ws[2] + ws[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  optim_decide<true>(partial, (int)gridDim.x, extra, n_extra, grad_scale, max_norm, beta1, beta2, loss, skip_threshold, state, wd);
  if (threadIdx.x == 0) *ticket = 0u;
}

__global__ __launch_bounds__(256) void optim_finalize(const float* __restrict__ partial, int nb,
                                                      const double* __restrict__ extra, int n_extra, float grad_scale,
                                                      float max_norm, float beta1, float beta2,
                                                      const float* __restrict__ loss, float skip_threshold,
                                                      float* __restrict__ state) {
  __shared__ double ws[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += (double)partial[i];
  for (int i = threadIdx.x; i < n_extra; i += blockDim.x) acc += extra[i];     // ||gW||^2 of never-materialised gradients
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double total = (ws[0] + ws[1]) + (ws[2] + ws[3]);
  const float norm = (float)sqrt(total) * fabsf(grad_scale);
  bool skip = false;
  if (loss) {
    const float l = *loss;
    skip = (l >= skip_threshold) || (l != l);                 // utils.py:145
  }
  state[ST_NORM] = norm;
  state[ST_SKIP] = skip ? 1.f : 0.f;
  if (skip) {
    state[ST_NSKIPPED] += 1.f;
    return;
  }
  float coef = max_norm / (norm + 1e-6f);                      // torch.nn.utils.clip_grad_norm_
  if (coef > 1.f) coef = 1.f;
  state[ST_CLIP] = coef * grad_scale;
  const double step = (double)state[ST_STEP] + 1.0;
  state[ST_STEP] = (float)step;
  state[ST_BC1] = (float)(1.0 - pow((double)beta1, step));
  state[ST_BC2SQRT] = (float)sqrt(1.0 - pow((double)beta2, step));
}

__global__ __launch_bounds__(256) void adam_update(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float beta1, float beta2, float eps,
                                                   const float* __restrict__ state) {
  if (state[ST_SKIP] != 0.f) return;
  const AdamStep a = adam_step_of(state, lr, beta1, beta2, eps);
  const int64_t n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  auto upd = [&](float& pp, float gg, float& mm, float& vv) { adam_elem(a, pp, gg, mm, vv); };
  // streaming pass: nothing here is read again before the next step's kernels have flushed the caches, so
  // g / m / v go around them (nontemporal) and two float4 per array are in flight per thread
  typedef float f4v __attribute__((ext_vector_type(4)));
  auto ldnt = [](const float4* q) { const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(q)); return make_float4(t.x, t.y, t.z, t.w); };
  auto stnt = [](float4* q, float4 x) { __builtin_nontemporal_store(f4v{x.x, x.y, x.z, x.w}, reinterpret_cast<f4v*>(q)); };
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const int64_t j = i + stride;
    float4 pa = p4[i], ga = ldnt(g4 + i), ma = ldnt(m4 + i), va = ldnt(v4 + i);
    float4 pb = p4[j], gb = ldnt(g4 + j), mb = ldnt(m4 + j), vb = ldnt(v4 + j);
    upd(pa.x, ga.x, ma.x, va.x); upd(pa.y, ga.y, ma.y, va.y); upd(pa.z, ga.z, ma.z, va.z); upd(pa.w, ga.w, ma.w, va.w);
    upd(pb.x, gb.x, mb.x, vb.x); upd(pb.y, gb.y, mb.y, vb.y); upd(pb.z, gb.z, mb.z, vb.z); upd(pb.w, gb.w, mb.w, vb.w);
    p4[i] = pa; stnt(m4 + i, ma); stnt(v4 + i, va);
    p4[j] = pb; stnt(m4 + j, mb); stnt(v4 + j, vb);
  }
  if (i < n4) {
    float4 pp = p4[i], gg = ldnt(g4 + i), mm = ldnt(m4 + i), vv = ldnt(v4 + i);
    upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
    p4[i] = pp; stnt(m4 + i, mm); stnt(v4 + i, vv);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    upd(p[i], g[i], m[i], v[i]);
  }
}

__global__ __launch_bounds__(256) void sgd_update(float* __restrict__ p, const float* __restrict__ g, int64_t n, float lr,
                                                  const float* __restrict__ state) {
  if (state[ST_SKIP] != 0.f) return;
  const float step = lr * state[ST_CLIP];                      // clip coefficient x gradient scale (1 / world)
  const int64_t n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = p4[i];
    const float4 gg = g4[i];
    pp.x = fmaf(-step, gg.x, pp.x); pp.y = fmaf(-step, gg.y, pp.y); pp.z = fmaf(-step, gg.z, pp.z); pp.w = fmaf(-step, gg.w, pp.w);
    p4[i] = pp;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t j = (n4 << 2) + threadIdx.x;
    p[j] = fmaf(-step, g[j], p[j]);
  }
}

}  // namespace cgv

extern "C" {

int cgv_optim_state_floats(void) { return 8; }
int cgv_optim_partial_floats(void) { return 2048 + 16; }      /* 2048 partial sums + the ticket word of the one-launch decision pass (ZERO it once) */

static void launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                        float eps, const float* state, hipStream_t st) {
  // one streaming pass wants many more blocks than CUs: 2048 -> 16384 blocks: 297 -> 268 us on 67.5 M parameters
  // (7.05 TB/s); each thread still moves two float4 per array per trip
  const int64_t want = ((n >> 2) + 511) / 512;
  const int nba = (int)(want < 1 ? 1 : (want > 16384 ? 16384 : want));
  hipLaunchKernelGGL(cgv::adam_update, dim3(nba), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps, state);
}

int cgv_adam_clip_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                       float eps, float max_norm, float grad_scale, const float* loss, float skip_threshold,
                       float* state, float* partial, void* stream) {
  int rc = cgv_optim_prepare(g, n, beta1, beta2, max_norm, grad_scale, loss, skip_threshold, state, partial, stream);
  if (rc) return rc;
  return cgv_adam_apply(p, g, m, v, n, lr, beta1, beta2, eps, state, stream);
}

/* The two halves of cgv_adam_clip_step, so that the parameter pass can be issued per arena range and later than the
 * norm (trainer: the previous step's update of the decoder's range runs beside the next step's encoder). */
int cgv_optim_prepare(const float* g, int64_t n, float beta1, float beta2, float max_norm, float grad_scale,
                      const float* loss, float skip_threshold, float* state, float* partial, void* stream) {
  return cgv_optim_prepare_extra(g, n, nullptr, 0, beta1, beta2, max_norm, grad_scale, loss, skip_threshold, state, partial,
                                 stream);
}

/* As cgv_optim_prepare over g[0, n), plus n_extra squared norms (doubles) of gradients that live nowhere in g: the
 * rank-update layers (cgv_wgrad_gram / cgv_grouped_wgrad_adam). */
int cgv_optim_prepare_extra(const float* g, int64_t n, const double* extra, int n_extra, float beta1, float beta2,
                            float max_norm, float grad_scale, const float* loss, float skip_threshold, float* state,
                            float* partial, void* stream) {
  CGV_REQUIRE(g && state && partial && n >= 0, "bad argument");
  CGV_REQUIRE(n_extra >= 0 && (n_extra == 0 || extra), "bad extra partials");
  CGV_REQUIRE((((uintptr_t)g) & 15) == 0, "arena must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  // 2048 blocks (= cgv_optim_partial_floats()) for a whole arena -- more measured slower; fewer for a short range
  const int64_t want = ((n >> 2) + 1023) / 1024;
  const int nb = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  if (cgv::option(CGV_OPT_OPTIM_ONE_LAUNCH) != 0) {
    // the ticket word sits behind the 2048 partial sums (cgv_optim_partial_floats() = 2048 + 16; the caller zeroes the
    // buffer once, every launch leaves the word zero)
    unsigned int* ticket = reinterpret_cast<unsigned int*>(partial + 2048);
    hipLaunchKernelGGL(cgv::sumsq_decide_k, dim3(nb), dim3(256), 0, st, g, n, partial, ticket, extra, n_extra, grad_scale, max_norm,
                       beta1, beta2, loss, skip_threshold, state);
    return cgv::check_launch("cgv_optim_prepare");
  }
  hipLaunchKernelGGL(cgv::sumsq_partial, dim3(nb), dim3(256), 0, st, g, n, partial);
  hipLaunchKernelGGL(cgv::optim_finalize, dim3(1), dim3(256), 0, st, partial, nb, extra, n_extra, grad_scale, max_norm,
                     beta1, beta2, loss, skip_threshold, state);
  return cgv::check_launch("cgv_optim_prepare");
}

int cgv_adam_apply(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                   const float* state, void* stream) {
  CGV_REQUIRE(p && g && m && v && state && n >= 0, "bad argument");
  CGV_REQUIRE(((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v)) & 15) == 0, "range must be 16-byte aligned");
  if (n == 0) return 0;
  launch_adam(p, g, m, v, n, lr, beta1, beta2, eps, state, (hipStream_t)stream);
  return cgv::check_launch("cgv_adam_apply");
}

/* torch.optim.SGD defaults (no momentum / weight decay; run_ala.py:43 `-optimizer sgd`): p -= lr * clip * g over one range,
 * with the clip coefficient and the skip decision of cgv_optim_prepare's state (scripts/utils.py:145-157). */
int cgv_sgd_apply(float* p, const float* g, int64_t n, float lr, const float* state, void* stream) {
  CGV_REQUIRE(p && g && state && n >= 0, "bad argument");
  CGV_REQUIRE(((((uintptr_t)p | (uintptr_t)g)) & 15) == 0, "range must be 16-byte aligned");
  if (n == 0) return 0;
  const int nb = (int)((((n + 3) >> 2) + 255) / 256 < 4096 ? (((n + 3) >> 2) + 255) / 256 : 4096);
  hipLaunchKernelGGL(cgv::sgd_update, dim3(nb < 1 ? 1 : nb), dim3(256), 0, (hipStream_t)stream, p, g, n, lr, state);
  return cgv::check_launch("cgv_sgd_apply");
}

}  // extern "C"
