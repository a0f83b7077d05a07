// Measurement only: does it pay to have a kernel of the decoder chain TOUCH (one dword per 128-byte line) the weight
// rows the NEXT kernel's block with the same index will stream?  Pairs (touch, pull) in a captured chain; the touch
// goes to the pull's own buffer (same block index => same XCD: L2-warm), to the pull's buffer shifted by one block
// (another XCD: Infinity-Cache-warm only) or to an unrelated buffer (cold).  Every pair uses a fresh region of a 1.5 GB pool.
// Build: hipcc --offload-arch=gfx950 -O3 -o warm_probe warm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NLOAD>
__global__ __launch_bounds__(576) void pull_k(const float* __restrict__ W, float* __restrict__ out, size_t block_floats) {
  const float* base = W + (size_t)blockIdx.x * block_floats;
  float4 r[NLOAD];
#pragma unroll
  for (int u = 0; u < NLOAD; ++u) {
    size_t off = ((size_t)u * 576 + threadIdx.x) * 4;
    if (off >= block_floats) off = 0;
    r[u] = *reinterpret_cast<const float4*>(base + off);
  }
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < NLOAD; ++u) acc += r[u].x + r[u].y + r[u].z + r[u].w;
  if (acc == 123.456f) out[blockIdx.x * 576 + threadIdx.x] = acc;
}

// one dword per 128-byte line of block (blockIdx.x + shift) % gridDim.x's region; the result is never used
__global__ __launch_bounds__(576) void touch_k(const float* __restrict__ W, float* __restrict__ out, size_t block_floats, int shift, int wait) {
  const int b = (blockIdx.x + shift) % gridDim.x;
  const float* base = W + (size_t)b * block_floats;
  float acc = 0.f;
  for (size_t off = (size_t)threadIdx.x * 32; off < block_floats; off += 576 * 32) {
    float t;
    asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(base + off) : "memory");
    if (wait) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += t; }
  }
  if (wait && acc == 123.456f) out[0] = acc;
}

template <typename F>
static float chain_us(F launch, int chain, int reps, hipStream_t st) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int c = 0; c < chain; ++c) launch(c);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1000.f / chain;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t pool_floats = (size_t)384 << 20;      // 1.5 GB
  float* pool; CK(hipMalloc(&pool, pool_floats * 4));
  CK(hipMemset(pool, 0, pool_floats * 4));
  float* out; CK(hipMalloc(&out, 1 << 22));
  const int chain = 60;
  struct { const char* name; size_t rows; } shapes[] = {{"message W2 (36 rows x 600)", 36}, {"gate W1' (12 rows x 600)", 12}, {"uv (8 rows x 600)", 8}, {"dense (4 rows x 600)", 4}};
  for (auto& sh : shapes) {
    const size_t bf = sh.rows * 600, fl = 150 * bf;
    size_t cursor = 0;
    auto region = [&](int c) { return pool + ((size_t)c * 2 * fl) % (pool_floats - 2 * fl); };
    (void)cursor;
    float pull_only = chain_us([&](int c) { hipLaunchKernelGGL((pull_k<10>), dim3(150), dim3(576), 0, st, region(c), out, bf); }, chain, 5, st);
    float touch_only = chain_us([&](int c) { hipLaunchKernelGGL(touch_k, dim3(150), dim3(576), 0, st, region(c), out, bf, 0, 0); }, chain, 5, st);
    float touch_wait = chain_us([&](int c) { hipLaunchKernelGGL(touch_k, dim3(150), dim3(576), 0, st, region(c), out, bf, 0, 1); }, chain, 5, st);
    printf("%s: %.2f MB\n  pull alone (cold)                 %6.2f us\n  touch alone (no wait / wait)      %6.2f / %6.2f us\n", sh.name, fl * 4 / 1e6, pull_only, touch_only, touch_wait);
    for (int mode = 0; mode < 3; ++mode) {
      float pair = chain_us([&](int c) {
        const float* target = mode == 2 ? region(c) + fl : region(c);
        hipLaunchKernelGGL(touch_k, dim3(150), dim3(576), 0, st, target, out, bf, mode == 1 ? 1 : 0, 0);
        hipLaunchKernelGGL((pull_k<10>), dim3(150), dim3(576), 0, st, region(c), out, bf);
      }, chain, 5, st);
      const char* what[] = {"touch same block (L2-warm) + pull", "touch shifted block (MALL-warm) + pull", "touch unrelated (cold) + pull"};
      printf("  %-40s %6.2f us per pair  => pull %6.2f us\n", what[mode], pair, pair - touch_only);
    }
  }
  return 0;
}
