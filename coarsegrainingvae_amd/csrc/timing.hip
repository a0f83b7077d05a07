// In-stream timestamps: one-thread kernels that store the GPU's constant-rate wall clock (s_memrealtime, 100 MHz on
// gfx950) into a caller-owned slot.  They can be captured into a hipGraph like any other launch, so the time BETWEEN two
// marks is what a section of the replayed step really takes -- rocprofv3's per-kernel durations include its own
// per-dispatch serialisation (a 1-block fill kernel reads 4.4 us there) and HIP events cannot sit inside a captured graph.
// No reference counterpart (measurement only).
#include "cgv_common.h"

namespace cgv {
__global__ void timestamp_k(unsigned long long* slot) { *slot = wall_clock64(); }

// The shader clock the chip SUSTAINS under packed fp32 FMAs on every SIMD -- the load of the fused message forward, whose
// roofline peak (157.3 TF/s) is the 2.4 GHz figure: every wave runs `iters` x 32 independent v_pk_fma_f32 and one wave
// of the grid reports shader cycles (s_memtime) and wall-clock ticks (s_memrealtime) over the same span.
typedef float pf2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void sustained_clock_k(unsigned long long* out, float* sink, int iters) {
  pf2 a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = pf2{(float)threadIdx.x * 1e-3f + (float)i, 1.0f};
  const pf2 m = pf2{0.999f, 1.001f}, c = pf2{1e-3f, -1e-3f};
  const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  pf2 t = a[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) t += a[i];
  if (t.x + t.y == 12345.678f) sink[0] = t.x;                     // (keeps the FMAs alive; never true)
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
}  // namespace cgv

extern "C" {
int cgv_timestamp(uint64_t* slot, void* stream) {
  CGV_REQUIRE(slot, "null pointer");
  hipLaunchKernelGGL(cgv::timestamp_k, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(slot));
  return cgv::check_launch("cgv_timestamp");
}
/* Measurement: out[0] = shader cycles, out[1] = wall-clock ticks (cgv_timestamp_hz) of one wave's span while `blocks` x 4 waves
 * issue packed fp32 FMAs back to back (~iters x 130 cycles) -- cycles / seconds = the clock sustained under that load. */
int cgv_sustained_clock_probe(uint64_t* out, float* sink, int blocks, int iters, void* stream) {
  CGV_REQUIRE(out && sink && blocks > 0 && iters > 0, "bad argument");
  hipLaunchKernelGGL(cgv::sustained_clock_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<unsigned long long*>(out), sink, iters);
  return cgv::check_launch("cgv_sustained_clock_probe");
}
int cgv_timestamp_hz(void) {
  int rate = 0;                                   // kHz
  if (hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0) != hipSuccess || rate <= 0) return 100000000;
  return rate * 1000;
}
}
