// Standalone timing harness for the skinny GEMM entry points (hipEvents over back-to-back launches).
// build+run on the GPU box: tools/probes/run_gemm_probe.sh
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "cgvae_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <class F> float time_us(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return 1e3f * ms / reps;
}

int main() {
  const int shapes[][3] = {{12, 600, 600}, {12, 1800, 600}, {12, 600, 1200}, {36, 1200, 600}, {12, 5400, 600}, {36, 600, 600}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    // 16 rotating copies of W so that the weights come from HBM/MALL like in a real step, not from L2
    const int COPIES = 16;
    float *x, *W, *b, *y, *z, *gy, *gx; void* ws;
    CK(hipMalloc(&x, sizeof(float) * M * K)); CK(hipMalloc(&W, sizeof(float) * (size_t)N * K * COPIES));
    CK(hipMalloc(&b, sizeof(float) * N)); CK(hipMalloc(&y, sizeof(float) * M * N)); CK(hipMalloc(&z, sizeof(float) * M * N));
    CK(hipMalloc(&gy, sizeof(float) * M * N)); CK(hipMalloc(&gx, sizeof(float) * M * K));
    const size_t wsb = cgv_skinny_bwd_input_workspace_bytes(M, N, K); CK(hipMalloc(&ws, wsb + 16));
    CK(hipMemset(x, 0, sizeof(float) * M * K)); CK(hipMemset(W, 0, sizeof(float) * (size_t)N * K * COPIES));
    CK(hipMemset(gy, 0, sizeof(float) * M * N)); CK(hipMemset(b, 0, sizeof(float) * N)); CK(hipMemset(z, 0, sizeof(float) * M * N));
    int it = 0;
    float t_f = time_us([&] { cgv_skinny_linear_fwd(x, W + (size_t)(it++ % COPIES) * N * K, b, y, z, M, N, K, 1, 0); }, 200);
    float t_i = time_us([&] { cgv_skinny_linear_bwd_input(gy, z, W + (size_t)(it++ % COPIES) * N * K, gx, M, N, K, 1, ws, wsb, 0); }, 200);
    float t_1 = time_us([&] { cgv_skinny_linear_bwd_input(gy, z, W + (size_t)(it++ % COPIES) * N * K, gx, M, N, K, 1, nullptr, 0, 0); }, 200);
    const double mb = 4.0 * N * K / 1e6;
    printf("M=%2d N=%4d K=%4d  W=%.2f MB | fwd %6.2f us (%5.0f GB/s) | bwd_input split %6.2f us (%5.0f GB/s) | single %6.2f us\n",
           M, N, K, mb, t_f, mb / t_f * 1e3, t_i, mb / t_i * 1e3, t_1);
    hipFree(x); hipFree(W); hipFree(b); hipFree(y); hipFree(z); hipFree(gy); hipFree(gx); hipFree(ws);
  }
  const int big[][3] = {{332, 600, 600}, {332, 1800, 600}, {332, 600, 1200}, {96, 600, 600}, {96, 5400, 600}, {704, 600, 600}};
  for (auto& sh : big) {
    const int M = sh[0], N = sh[1], K = sh[2];
    float *x, *W, *b, *y, *z, *g, *gx, *gW;
    CK(hipMalloc(&x, sizeof(float) * M * K)); CK(hipMalloc(&W, sizeof(float) * (size_t)N * K)); CK(hipMalloc(&gW, sizeof(float) * (size_t)N * K));
    CK(hipMalloc(&b, sizeof(float) * N)); CK(hipMalloc(&y, sizeof(float) * M * N)); CK(hipMalloc(&z, sizeof(float) * M * N));
    CK(hipMalloc(&g, sizeof(float) * M * N)); CK(hipMalloc(&gx, sizeof(float) * M * K));
    CK(hipMemset(x, 0, sizeof(float) * M * K)); CK(hipMemset(W, 0, sizeof(float) * (size_t)N * K));
    CK(hipMemset(g, 0, sizeof(float) * M * N)); CK(hipMemset(b, 0, sizeof(float) * N)); CK(hipMemset(z, 0, sizeof(float) * M * N));
    float t_f = time_us([&] { cgv_tile_linear_fwd(x, W, b, y, z, M, N, K, 1, 0); }, 200);
    float t_p = time_us([&] { cgv_dense_grad_prepare(y, z, g, b, M, N, 1, 0, 0); }, 200);
    float t_i = time_us([&] { cgv_tile_linear_bwd_input(g, W, gx, M, N, K, 0); }, 200);
    float t_w = time_us([&] { cgv_tile_linear_wgrad(g, x, gW, M, N, K, 0, 0); }, 200);
    const double gf = 2.0 * M * N * K / 1e9;
    printf("tile M=%3d N=%4d K=%4d %.2f GFLOP | fwd %6.2f us (%5.1f TF) | prepare %6.2f us | bwd_input %6.2f us (%5.1f TF) | wgrad %6.2f us (%5.1f TF)\n",
           M, N, K, gf, t_f, gf / t_f * 1e3, t_p, t_i, gf / t_i * 1e3, t_w, gf / t_w * 1e3);
    hipFree(x); hipFree(W); hipFree(b); hipFree(y); hipFree(z); hipFree(g); hipFree(gx); hipFree(gW);
  }
  return 0;
}
