// How fast can the CUs pull operand rows that are resident in THEIR XCD's L2?  Block b runs on XCD b % 8.
//   mode 0: every XCD reads its own private region (region bytes per XCD, L2 resident when <= ~3 MB)
//   mode 1: all XCDs read one shared region of 8 x region bytes (every L2 has to hold all of it)
// Each block reads `per_block` bytes as float4 (unrolled x8 per thread, all in flight), reps times.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/l2_local_probe.hip -o tools/probes/l2_local_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void pull(const float4* __restrict__ src, float* __restrict__ out, size_t region_f4, int mode,
                                            int iters, int blocks_per_xcd) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const size_t total = mode == 0 ? region_f4 : region_f4 * 8;
  const float4* base = src + (mode == 0 ? (size_t)xcd * region_f4 : 0);
  // block `slot` of an XCD starts at its own offset and walks the whole region (so the XCD's blocks share lines)
  size_t at = ((size_t)slot * 8191 * 256 + threadIdx.x) % total;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = base[at]; at += 256; if (at >= total) at -= total; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  const size_t max_bytes = (size_t)8 * 32 << 20;
  float4* src; float* out;
  CK(hipMalloc(&src, max_bytes)); CK(hipMemset(src, 0, max_bytes)); CK(hipMalloc(&out, 4096 * 256 * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 2; ++mode)
    for (int waves : {1, 2, 4})                                     // blocks per CU
      for (size_t region_kb : {512, 1024, 2048, 3072, 8192, 32768}) {
        const size_t region_f4 = region_kb * 1024 / 16;
        const int blocks = 256 * waves, iters = 64;
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(pull, dim3(blocks), dim3(256), 0, 0, src, out, region_f4, mode, iters, blocks / 8);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, 0));
        const int reps = 10;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(pull, dim3(blocks), dim3(256), 0, 0, src, out, region_f4, mode, iters, blocks / 8);
        CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double bytes = (double)blocks * 256 * 8 * 16 * iters;
        printf("mode %d  %d blocks/CU  region %6zu KB per XCD: %7.1f us  %6.2f TB/s aggregate  %6.1f GB/s per CU\n", mode, waves, region_kb,
               1e3 * ms / reps, bytes * reps / (ms * 1e-3) / 1e12, bytes * reps / (ms * 1e-3) / 256 / 1e9);
      }
  return 0;
}
