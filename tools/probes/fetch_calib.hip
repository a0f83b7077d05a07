// Measurement only: what does rocprofv3's FETCH_SIZE report for a KNOWN number of bytes, by access width?
// Three kernels stream the same 1 GiB buffer once with 4-, 8- and 16-byte loads per lane (coalesced), and one gathers
// 8-byte fragments from 2400-byte rows the way the fused message kernels do (buffer of rows, random row per wave,
// lane l reads bytes [8 l, 8 l + 8) of three 512-byte pieces).  Run under `rocprofv3 --pmc FETCH_SIZE --` and divide.
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void stream_k(const T* __restrict__ p, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = p[i];
    acc += reinterpret_cast<const float*>(&v)[0];
  }
  if (acc == 123.456f) out[0] = acc;
}
typedef float f1;
struct alignas(8) f2s { float x, y; };
struct alignas(16) f4s { float x, y, z, w; };

// rows of 600 float2-pairs... : row = 2400 bytes; a wave reads 3 x 512 contiguous bytes of a pseudo-random row
__global__ __launch_bounds__(256) void gather8_k(const f2s* __restrict__ p, int n_rows, int reads_per_wave, float* out) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  unsigned s = 1234567u + 7919u * wave;
  float acc = 0.f;
  for (int r = 0; r < reads_per_wave; ++r) {
    s = s * 1664525u + 1013904223u;
    const size_t row = (s >> 8) % (unsigned)n_rows;
    const f2s* base = p + row * 300;                      // 300 float2 = 2400 bytes
    acc += base[lane].x + base[64 + lane].x + base[128 + lane].x;
  }
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  void* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  float* out; CK(hipMalloc(&out, 64));
  hipLaunchKernelGGL((stream_k<f1>), dim3(4096), dim3(256), 0, 0, (const f1*)buf, bytes / 4, out);
  hipLaunchKernelGGL((stream_k<f2s>), dim3(4096), dim3(256), 0, 0, (const f2s*)buf, bytes / 8, out);
  hipLaunchKernelGGL((stream_k<f4s>), dim3(4096), dim3(256), 0, 0, (const f4s*)buf, bytes / 16, out);
  // 4096 blocks x 4 waves x 64 reads x 1536 bytes = 1.5 GiB requested (rows re-read at random: hits in L2 / MALL do not reach HBM)
  hipLaunchKernelGGL(gather8_k, dim3(4096), dim3(256), 0, 0, (const f2s*)buf, (int)(bytes / 2400), 64, out);
  CK(hipDeviceSynchronize());
  printf("known bytes: stream 4 B/lane %zu, 8 B/lane %zu, 16 B/lane %zu; gather8 requests %zu bytes over %zu distinct-row draws\n", bytes, bytes, bytes,
         (size_t)4096 * 4 * 64 * 1536, (size_t)4096 * 4 * 64);
  return 0;
}
