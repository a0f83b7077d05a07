// How fast does a CU pull the operand rows of a mid-size GEMM tile, by access pattern?  One block of 512 threads per
// tile; block b reads the (TM + TN) rows of its tile of x [M, K] and W [N, K] (2400-byte rows), wave w the 256-byte slice
// [256 w, 256 w + 256) of every row, every load of a wave in flight at once, with one of these lane -> address maps per
// load instruction (64 lanes x 16 bytes):
//   0: 16 rows x  64 B   (the MFMA operand layout of the register-fed kernels)
//   1:  8 rows x 128 B
//   2:  4 rows x 256 B
// Operands rotate over 8 buffer sets between launches.  Prints us per launch and GB/s per CU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/frag_probe.hip -o tools/probes/frag_probe_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int PAT, int RB /*16-row blocks per tile*/>
__global__ __launch_bounds__(512) void pull_k(const float* __restrict__ x, const float* __restrict__ W, float* __restrict__ out,
                                              int M, int N, int K, int TM, int MB) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nt = blockIdx.x / MB, mt = blockIdx.x - nt * MB;
  const int m0 = mt * TM, n0 = nt * (16 * RB - TM);
  float4 v[RB][4];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int row, f4;
      if (PAT == 0) { row = lane & 15; f4 = 4 * u + (lane >> 4); }
      else if (PAT == 1) { row = 8 * (u & 1) + (lane >> 3); f4 = 8 * (u >> 1) + (lane & 7); }
      else { row = 4 * u + (lane >> 4); f4 = lane & 15; }
      const int grow = 16 * rb + row;
      const float* base = grow < TM ? x + (size_t)min(m0 + grow, M - 1) * K : W + (size_t)min(n0 + grow - TM, N - 1) * K;
      v[rb][u] = *reinterpret_cast<const float4*>(base + 64 * wave + 4 * f4);
    }
  float acc = 0.f;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[rb][u].x + v[rb][u].y + v[rb][u].z + v[rb][u].w;
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int RB>
static int run(const float* x, const float* W, float* out, int M, int N, int TM, int TN) {
  const int K = 600, NBUF = 8;
  const int MB = (M + TM - 1) / TM, NB = (N + TN - 1) / TN, blocks = MB * NB;
  const double kb = (double)(TM + TN) * 8 * 256 / 1024.0;
  printf("M=%d N=%d tile %dx%d: %d blocks, %.0f KB per block\n", M, N, TM, TN, blocks, kb);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int pat = 0; pat < 3; ++pat) {
    auto launch = [&](int i) {
      const float* xx = x + (size_t)(i % NBUF) * 2048 * K; const float* ww = W + (size_t)(i % NBUF) * 2048 * K;
      if (pat == 0) hipLaunchKernelGGL((pull_k<0, RB>), dim3(blocks), dim3(512), 0, 0, xx, ww, out, M, N, K, TM, MB);
      else if (pat == 1) hipLaunchKernelGGL((pull_k<1, RB>), dim3(blocks), dim3(512), 0, 0, xx, ww, out, M, N, K, TM, MB);
      else hipLaunchKernelGGL((pull_k<2, RB>), dim3(blocks), dim3(512), 0, 0, xx, ww, out, M, N, K, TM, MB);
    };
    for (int i = 0; i < 8; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    const int reps = 64;
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    printf("  %3d B per row and instruction: %6.2f us per launch, %.1f GB/s per CU, %.2f TB/s in total\n", 64 << pat, us,
           kb * 1024 / us / 1e3, kb * 1024 * blocks / us / 1e6);
  }
  return 0;
}

int main() {
  const int K = 600, NBUF = 8;
  float *x, *W, *out;
  CK(hipMalloc(&x, (size_t)NBUF * 2048 * K * 4)); CK(hipMalloc(&W, (size_t)NBUF * 2048 * K * 4)); CK(hipMalloc(&out, 2048 * 512 * 4));
  CK(hipMemset(x, 0, (size_t)NBUF * 2048 * K * 4)); CK(hipMemset(W, 0, (size_t)NBUF * 2048 * K * 4));
  if (run<7>(x, W, out, 332, 1800, 32, 80)) return 1;       // 253 tiles of 32 x 80
  if (run<9>(x, W, out, 704, 1800, 64, 80)) return 1;       // 253 tiles of 64 x 80
  if (run<6>(x, W, out, 704, 600, 48, 48)) return 1;        // 195 tiles of 48 x 48
  if (run<4>(x, W, out, 332, 1800, 32, 32)) return 1;       // 627 tiles of 32 x 32 (today's kernel: 3 per CU)
  if (run<4>(x, W, out, 704, 1800, 32, 32)) return 1;       // 1254 tiles of 32 x 32
  return 0;
}
