// How much of a weight-streaming launch is cache / TLB state?  Same skinny forward, rotating over C copies of W.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "cgvae_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <class F> float time_us(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 10; ++i) f();
  CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return 1e3f * ms / reps;
}
__global__ void empty_k() {}
int main() {
  const int M = 12;
  float t_e = time_us([&] { hipLaunchKernelGGL(empty_k, dim3(38), dim3(1024), 0, 0); }, 400);
  printf("empty kernel 38x1024: %.2f us per launch\n", t_e);
  const int shapes[][2] = {{600, 600}, {1800, 600}, {5400, 600}};
  const int copies[] = {1, 4, 16, 64, 160};
  for (auto& sh : shapes) for (int C : copies) {
    const int N = sh[0], K = sh[1];
    if ((size_t)N * K * C * 4 > (3ull << 30)) continue;
    float *x, *W, *b, *y, *z;
    CK(hipMalloc(&x, 4 * M * K)); CK(hipMalloc(&W, 4ull * N * K * C)); CK(hipMalloc(&b, 4 * N)); CK(hipMalloc(&y, 4 * M * N)); CK(hipMalloc(&z, 4 * M * N));
    CK(hipMemset(x, 0, 4 * M * K)); CK(hipMemset(W, 0, 4ull * N * K * C)); CK(hipMemset(b, 0, 4 * N));
    int it = 0;
    float t = time_us([&] { cgv_skinny_linear_fwd(x, W + (size_t)(it++ % C) * N * K, b, y, z, M, N, K, 1, 0); }, 640);
    printf("N=%4d K=%4d copies=%3d (%7.1f MB footprint): %6.2f us\n", N, K, C, 4.0 * N * K * C / 1e6, t);
    hipFree(x); hipFree(W); hipFree(b); hipFree(y); hipFree(z);
  }
  return 0;
}
