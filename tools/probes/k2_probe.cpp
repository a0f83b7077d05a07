// Ablation probe for the fused message forward: same loop skeleton, parts switched off at compile time.
// build+run: tools/probes/run_k2_probe.sh
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ rsrc_t make_rsrc(const float* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ f2 ld2_buf(rsrc_t r, unsigned v, unsigned s) { return __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, v, s, 0)); }

typedef float f3v __attribute__((ext_vector_type(3)));
typedef unsigned u3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ f3v ld3_buf(rsrc_t r, unsigned v, unsigned s) { return __builtin_bit_cast(f3v, __builtin_amdgcn_raw_buffer_load_b96(r, v, s, 0)); }
__device__ __forceinline__ float ld1_buf(rsrc_t r, unsigned v, unsigned s) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, v, s, 0)); }
constexpr int R = 10, GS = 20, U = 12;

template <bool GATHER, bool SMEM, bool FMA, int SPLIT, int UNROLL, int LAYOUT = 0>
__global__ __launch_bounds__(64 * SPLIT) void k2(const float* __restrict__ phi, const float* __restrict__ v,
                                                 const float* __restrict__ geom, const int* __restrict__ rowptr,
                                                 const int* __restrict__ src, const float* __restrict__ Wd,
                                                 float* __restrict__ ds, float* __restrict__ dv, int F, int n_dst, int tiles) {
  const int node = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = min(tile * 128 + 2 * lane, F - 2);
  f2 W0[R + 1], W1[R + 1], W2[R + 1];
#pragma unroll
  for (int n = 0; n <= R; ++n) { W0[n] = splat(Wd[n] + lane); W1[n] = splat(Wd[n + 16] - lane); W2[n] = splat(Wd[n + 32] * 0.5f); }
  f2 acc_s = splat(0.f), aA = splat(0.f), aB = splat(0.f), aC = splat(0.f);
  int beg = rowptr[node], end = rowptr[node + 1];
  if (SPLIT > 1) { const int len = (end - beg + SPLIT - 1) / SPLIT; beg = min(beg + wave * len, end); end = min(beg + len, end); }
  const unsigned rb = 12u * F, oc = 4u * c, oF = 4u * F, ov = 12u * c;
  const rsrc_t rp = make_rsrc(phi), rv = make_rsrc(v);
  float gfix[GS];
#pragma unroll
  for (int t = 0; t < GS; ++t) gfix[t] = geom[t];
#pragma unroll UNROLL
  for (int e = beg; e < end; ++e) {
    float g[GS];
#pragma unroll
    for (int t = 0; t < GS; ++t) g[t] = SMEM ? geom[(size_t)e * GS + t] : gfix[t] + (float)e;
    const unsigned so = SMEM ? (unsigned)src[e] * rb : (unsigned)(e & 255) * rb;
    f2 p0, p1, p2, A, B, C;
    if (GATHER) {
      if (LAYOUT == 0) {
        p1 = ld2_buf(rp, oc + oF, so); p0 = ld2_buf(rp, oc, so); p2 = ld2_buf(rp, oc + 2 * oF, so);
        A = ld2_buf(rv, ov, so); B = ld2_buf(rv, ov + 8, so); C = ld2_buf(rv, ov + 16, so);
      } else if (LAYOUT == 1) {   // same channel map, v as 2 x b96
        p1 = ld2_buf(rp, oc + oF, so); p0 = ld2_buf(rp, oc, so); p2 = ld2_buf(rp, oc + 2 * oF, so);
        const f3v x = ld3_buf(rv, ov, so), y = ld3_buf(rv, ov + 12, so);
        A = f2{x.x, y.x}; B = f2{x.y, y.y}; C = f2{x.z, y.z};
      } else if (LAYOUT == 2) {   // lane -> channels (l, l+64): phi 6 x b32, v 2 x b96 dense
        const unsigned o1 = 4u * (tile * 128 + lane), o3 = 12u * (tile * 128 + lane);
        p1 = f2{ld1_buf(rp, o1 + oF, so), ld1_buf(rp, o1 + oF + 256, so)};
        p0 = f2{ld1_buf(rp, o1, so), ld1_buf(rp, o1 + 256, so)};
        p2 = f2{ld1_buf(rp, o1 + 2 * oF, so), ld1_buf(rp, o1 + 2 * oF + 256, so)};
        const f3v x = ld3_buf(rv, o3, so), y = ld3_buf(rv, o3 + 768, so);
        A = f2{x.x, y.x}; B = f2{x.y, y.y}; C = f2{x.z, y.z};
      } else {                    // component-major v [N,3,F]: 6 coalesced b64
        p1 = ld2_buf(rp, oc + oF, so); p0 = ld2_buf(rp, oc, so); p2 = ld2_buf(rp, oc + 2 * oF, so);
        A = ld2_buf(rv, oc, so); B = ld2_buf(rv, oc + oF, so); C = ld2_buf(rv, oc + 2 * oF, so);
      }
    } else {
      p0 = splat(g[0]); p1 = splat(g[1]); p2 = splat(g[2]); A = splat(g[3]); B = splat(g[4]); C = splat(g[5]);
    }
    if (FMA) {
      f2 w0 = W0[R] * splat(g[R]), w1 = W1[R] * splat(g[R]), w2 = W2[R] * splat(g[R]);
#pragma unroll
      for (int n = 0; n < R; ++n) { w0 = fma2(W0[n], splat(g[n]), w0); w1 = fma2(W1[n], splat(g[n]), w1); w2 = fma2(W2[n], splat(g[n]), w2); }
      acc_s = fma2(p1, w1, acc_s);
      const f2 m0 = p0 * w0, m2 = p2 * w2;
      aA = fma2(m2, f2{g[U], g[U + 1]}, fma2(m0, A, aA));
      aB = fma2(m2, f2{g[U + 2], g[U + 3]}, fma2(m0, B, aB));
      aC = fma2(m2, f2{g[U + 4], g[U + 5]}, fma2(m0, C, aC));
    } else {
      acc_s += p1 + p0 + p2; aA += A; aB += B; aC += C + splat(g[0]);
    }
  }
  if (wave == 0 || SPLIT == 1) {
    *(f2*)(ds + (size_t)node * F + c) = acc_s + aA;
    *(f2*)(dv + ((size_t)node * F + c) * 3) = aB + aC;
  }
}

template <class K> float run(K kernel, dim3 grid, dim3 block, const float* phi, const float* v, const float* geom, const int* rp,
                             const int* src, const float* Wd, float* ds, float* dv, int F, int N, int tiles) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kernel, grid, block, 0, 0, phi, v, geom, rp, src, Wd, ds, dv, F, N, tiles);
  CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kernel, grid, block, 0, 0, phi, v, geom, rp, src, Wd, ds, dv, F, N, tiles);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return 1e3f * ms / 50;
}

int main() {
  const int N = 332, deg = 125, F = 600, E = N * deg, tiles = 5;
  std::vector<int> rp(N + 1), src(E);
  for (int i = 0; i <= N; ++i) rp[i] = i * deg;
  for (int i = 0; i < N; ++i) for (int k = 0; k < deg; ++k) src[i * deg + k] = (i / 166) * 166 + (k * 7 + i) % 166;
  float *phi, *v, *geom, *Wd, *ds, *dv; int *d_rp, *d_src;
  CK(hipMalloc(&phi, 4ull * N * 3 * F)); CK(hipMalloc(&v, 4ull * N * 3 * F)); CK(hipMalloc(&geom, 4ull * E * GS)); CK(hipMalloc(&Wd, 4096));
  CK(hipMalloc(&ds, 4ull * N * F)); CK(hipMalloc(&dv, 4ull * N * F * 3)); CK(hipMalloc(&d_rp, 4 * (N + 1))); CK(hipMalloc(&d_src, 4 * E));
  CK(hipMemset(phi, 0, 4ull * N * 3 * F)); CK(hipMemset(v, 0, 4ull * N * 3 * F)); CK(hipMemset(geom, 0, 4ull * E * GS)); CK(hipMemset(Wd, 0, 4096));
  CK(hipMemcpy(d_rp, rp.data(), 4 * (N + 1), hipMemcpyHostToDevice)); CK(hipMemcpy(d_src, src.data(), 4 * E, hipMemcpyHostToDevice));
  dim3 grid(N * tiles);
#define RUN(G, S, M, SP, UN, ...) printf("gather=%d smem=%d fma=%d split=%d unroll=%d layout=%s: %7.2f us\n", G, S, M, SP, UN, #__VA_ARGS__, \
    run(k2<G, S, M, SP, UN, ##__VA_ARGS__>, grid, dim3(64 * SP), phi, v, geom, d_rp, d_src, Wd, ds, dv, F, N, tiles))
  RUN(true, true, true, 4, 2);
  RUN(false, true, true, 4, 2);
  RUN(true, false, true, 4, 2);
  RUN(false, false, true, 4, 2);
  RUN(true, true, false, 4, 2);
  RUN(true, true, true, 1, 2);
  RUN(true, true, true, 2, 2);
  RUN(true, true, true, 8, 2);
  RUN(true, true, true, 4, 1);
  RUN(true, true, true, 4, 4);
  RUN(false, false, true, 1, 2);
  for (int rep = 0; rep < 2; ++rep) {
  RUN(true, true, true, 4, 2, 0);
  RUN(true, true, true, 4, 2, 1);
  RUN(true, true, true, 4, 2, 2);
  RUN(true, true, true, 4, 2, 3);
  RUN(true, true, false, 4, 2, 1);
  RUN(true, true, false, 4, 2, 2);
  RUN(true, true, false, 4, 2, 3);
  RUN(true, true, true, 8, 2, 2);
  RUN(true, true, true, 8, 2, 3);
  }
  return 0;
}
