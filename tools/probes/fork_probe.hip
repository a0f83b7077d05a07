// Fork/join inside a captured hipGraph: a chain of N small dependent launches (150 blocks x 576 threads, like the decoder
// kernels) beside ONE persistent HBM-bound pass of P blocks x 256 threads (Adam-like: 3 arrays read + written).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/fork_probe.hip -o tools/probes/fork_probe_bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(576) void small_k(float* a, int n) {
  __shared__ float s[576];
  const int i = blockIdx.x * 576 + threadIdx.x;
  s[threadIdx.x] = i < n ? a[i] : 0.f;
  __syncthreads();
  if (i < n) a[i] = s[(threadIdx.x + 1) % 576] * 0.5f + 1.f;
}

__global__ __launch_bounds__(256) void pass_k(float4* p, float4* m, float4* v, long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    float4 a[4], b[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long j = i + u * stride;
      if (j < n4) { a[u] = p[j]; b[u] = m[j]; c[u] = v[j]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long j = i + u * stride;
      if (j < n4) {
        b[u].x = 0.9f * b[u].x + 0.1f * a[u].x; c[u].x = 0.999f * c[u].x + 0.001f * a[u].x * a[u].x; a[u].x -= 1e-4f * b[u].x / (sqrtf(c[u].x) + 1e-8f);
        b[u].y = 0.9f * b[u].y + 0.1f * a[u].y; c[u].y = 0.999f * c[u].y + 0.001f * a[u].y * a[u].y; a[u].y -= 1e-4f * b[u].y / (sqrtf(c[u].y) + 1e-8f);
        b[u].z = 0.9f * b[u].z + 0.1f * a[u].z; c[u].z = 0.999f * c[u].z + 0.001f * a[u].z * a[u].z; a[u].z -= 1e-4f * b[u].z / (sqrtf(c[u].z) + 1e-8f);
        b[u].w = 0.9f * b[u].w + 0.1f * a[u].w; c[u].w = 0.999f * c[u].w + 0.001f * a[u].w * a[u].w; a[u].w -= 1e-4f * b[u].w / (sqrtf(c[u].w) + 1e-8f);
        p[j] = a[u]; m[j] = b[u]; v[j] = c[u];
      }
    }
  }
}

static float replay_us(hipGraphExec_t g, hipStream_t s, int reps = 30) {
  std::vector<float> t;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int r = 0; r < reps; ++r) {
    hipStreamSynchronize(s);
    hipEventRecord(a, s); hipGraphLaunch(g, s); hipEventRecord(b, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  const long n = 55L << 20;                 // 55 M weights x 3 arrays x 2 directions x 4 B = 1.38 GB per pass
  const int N = 60, small_n = 150 * 576;
  float *p, *m, *v, *a, *a2;
  CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&a, small_n * 4)); CK(hipMalloc(&a2, small_n * 4)); CK(hipMemset(a2, 0, small_n * 4));
  CK(hipMemset(p, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4)); CK(hipMemset(a, 0, small_n * 4));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  auto capture = [&](int P, int chain, int join_at, bool forked, hipGraphExec_t* out) -> int {
    hipGraph_t g;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    if (P > 0) {
      if (forked) { CK(hipEventRecord(fork, s1)); CK(hipStreamWaitEvent(s2, fork, 0)); }
      hipLaunchKernelGGL(pass_k, dim3(P), dim3(256), 0, forked ? s2 : s1, (float4*)p, (float4*)m, (float4*)v, n / 4);
      if (forked) CK(hipEventRecord(join, s2));
    }
    for (int i = 0; i < chain; ++i) {
      if (forked && P > 0 && i == join_at) CK(hipStreamWaitEvent(s1, join, 0));
      hipLaunchKernelGGL(small_k, dim3(150), dim3(576), 0, s1, a, small_n);
    }
    if (forked && P > 0 && join_at >= chain) CK(hipStreamWaitEvent(s1, join, 0));
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
    return 0;
  };
  hipGraphExec_t ge;
  if (capture(0, N, N, false, &ge)) return 1;
  printf("chain of %d alone              %8.1f us\n", N, replay_us(ge, s1));
  for (int P : {64, 256}) {
    if (capture(P, 0, 0, false, &ge)) return 1;
    const float alone = replay_us(ge, s1);
    if (capture(P, N, N, false, &ge)) return 1;
    const float serial = replay_us(ge, s1);
    if (capture(P, N, N, true, &ge)) return 1;
    const float fk = replay_us(ge, s1);
    if (capture(P, 2 * N, 2 * N, true, &ge)) return 1;
    const float fk2 = replay_us(ge, s1);
    if (capture(P, 4 * N, 4 * N, true, &ge)) return 1;
    const float fk4 = replay_us(ge, s1);
    printf("P=%5d pass alone %7.1f (%.2f TB/s)  serial+chain %7.1f  forked beside %d: %7.1f  beside %d: %7.1f  beside %d: %7.1f\n", P, alone,
           n * 24.0 / alone / 1e6, serial, N, fk, 2 * N, fk2, 4 * N, fk4);
  }
  // two latency-bound chains side by side (prior net beside the encoder): N launches on s1, N2 on s2, fork at the start, join at the end
  auto capture2 = [&](int n1, int n2, int blocks2, hipGraphExec_t* out) -> int {
    hipGraph_t g;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    if (n2 > 0) { CK(hipEventRecord(fork, s1)); CK(hipStreamWaitEvent(s2, fork, 0)); }
    for (int i = 0; i < std::max(n1, n2); ++i) {
      if (i < n1) hipLaunchKernelGGL(small_k, dim3(150), dim3(576), 0, s1, a, small_n);
      if (i < n2) hipLaunchKernelGGL(small_k, dim3(blocks2), dim3(576), 0, s2, a2, small_n);
    }
    if (n2 > 0) { CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s1, join, 0)); }
    hipLaunchKernelGGL(small_k, dim3(150), dim3(576), 0, s1, a, small_n);
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
    return 0;
  };
  for (int n2 : {0, 10, 30, 60}) {
    for (int b2 : {12, 150}) {
      if (capture2(60, n2, b2, &ge)) return 1;
      printf("chain 60 + forked chain of %2d (x%3d blocks): %7.1f us\n", n2, b2, replay_us(ge, s1));
    }
  }
  // two SEPARATE linear graphs replayed on two streams at the same time (event fork / join outside the graphs)
  auto linear = [&](int n1, int blocks, float* buf, hipStream_t st, hipGraphExec_t* out) -> int {
    hipGraph_t g;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n1; ++i) hipLaunchKernelGGL(small_k, dim3(blocks), dim3(576), 0, st, buf, small_n);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
    return 0;
  };
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int n2 : {10, 30, 60}) {
    hipGraphExec_t ga, gb;
    if (linear(60, 150, a, s1, &ga) || linear(n2, 12, a2, s2, &gb)) return 1;
    std::vector<float> t;
    for (int r = 0; r < 30; ++r) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, s1));
      CK(hipEventRecord(fork, s1)); CK(hipStreamWaitEvent(s2, fork, 0));
      CK(hipGraphLaunch(gb, s2));
      CK(hipGraphLaunch(ga, s1));
      CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s1, join, 0));
      CK(hipEventRecord(t1, s1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, t0, t1)); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    printf("two linear graphs on two streams: 60 + %2d launches: %7.1f us\n", n2, t[t.size() / 2]);
  }
  // the segmented step: main = G1(10) G2(20) [wait side] G3(1) G4(25) [wait side] G5(4); side = [after G1] Gp1(9), [after G3] Gp2(13)
  {
    hipGraphExec_t g1, g2, g3, g4, g5, p1, p2, all;
    if (linear(50, 150, a, s1, &g1) || linear(100, 150, a, s1, &g2) || linear(5, 150, a, s1, &g3) || linear(125, 150, a, s1, &g4) ||
        linear(20, 150, a, s1, &g5) || linear(45, 12, a2, s2, &p1) || linear(65, 12, a2, s2, &p2) || linear(410, 150, a, s1, &all)) return 1;
    hipEvent_t e1, e2, e3, e4, tt[6];
    for (int i = 0; i < 6; ++i) CK(hipEventCreate(&tt[i]));
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e3, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e4, hipEventDisableTiming));
    for (int mode = 0; mode < 3; ++mode) {
      std::vector<float> t;
      for (int r = 0; r < 30; ++r) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(t0, s1));
        if (mode == 0) {
          CK(hipGraphLaunch(all, s1));
        } else if (mode == 1) {            // the same 82 launches as 7 graphs on ONE stream, in the step's order
          CK(hipGraphLaunch(g1, s1)); CK(hipGraphLaunch(p1, s1)); CK(hipGraphLaunch(g2, s1)); CK(hipGraphLaunch(g3, s1));
          CK(hipGraphLaunch(p2, s1)); CK(hipGraphLaunch(g4, s1)); CK(hipGraphLaunch(g5, s1));
        } else {
          CK(hipGraphLaunch(g1, s1)); CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s2, e1, 0));
          CK(hipEventRecord(tt[0], s2)); CK(hipGraphLaunch(p1, s2)); CK(hipEventRecord(e2, s2));
          CK(hipEventRecord(tt[1], s1)); CK(hipGraphLaunch(g2, s1)); CK(hipEventRecord(tt[2], s1)); CK(hipStreamWaitEvent(s1, e2, 0));
          CK(hipGraphLaunch(g3, s1)); CK(hipEventRecord(e3, s1)); CK(hipStreamWaitEvent(s2, e3, 0));
          CK(hipEventRecord(tt[3], s2)); CK(hipGraphLaunch(p2, s2)); CK(hipEventRecord(e4, s2));
          CK(hipEventRecord(tt[4], s1)); CK(hipGraphLaunch(g4, s1)); CK(hipEventRecord(tt[5], s1)); CK(hipStreamWaitEvent(s1, e4, 0));
          CK(hipGraphLaunch(g5, s1));
        }
        CK(hipEventRecord(t1, s1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, t0, t1)); t.push_back(ms * 1e3f);
        if (mode == 2 && r == 29) {
          CK(hipEventSynchronize(e2)); CK(hipEventSynchronize(e4));
          const char* nm[6] = {"p1 begins", "g2 begins", "g2 ends", "p2 begins", "g4 begins", "g4 ends"};
          for (int i = 0; i < 6; ++i) { float x; CK(hipEventElapsedTime(&x, t0, tt[i])); printf("   %s at %7.1f us\n", nm[i], x * 1e3f); }
        }
      }
      std::sort(t.begin(), t.end());
      printf("%s: %7.1f us\n", mode == 0 ? "410 launches, one graph" : mode == 1 ? "410 launches, 7 graphs on one stream" : "300 main (5 graphs) + 110 side (2 graphs), events", t[t.size() / 2]);
    }
  }
  return 0;
}
