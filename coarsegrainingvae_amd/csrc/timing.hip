// In-stream timestamps: one-thread kernels that store the GPU's constant-rate wall clock (s_memrealtime, 100 MHz on
// gfx950) into a caller-owned slot.  They can be captured into a hipGraph like any other launch, so the time BETWEEN two
// marks is what a section of the replayed step really takes -- rocprofv3's per-kernel durations include its own
// per-dispatch serialisation (a 1-block fill kernel reads 4.4 us there) and HIP events cannot sit inside a captured graph.
// No reference counterpart (measurement only).
#include "cgv_common.h"

namespace cgv {
__global__ void timestamp_k(unsigned long long* slot) { *slot = wall_clock64(); }
}  // namespace cgv

extern "C" {
int cgv_timestamp(uint64_t* slot, void* stream) {
  CGV_REQUIRE(slot, "null pointer");
  hipLaunchKernelGGL(cgv::timestamp_k, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(slot));
  return cgv::check_launch("cgv_timestamp");
}
int cgv_timestamp_hz(void) {
  int rate = 0;                                   // kHz
  if (hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0) != hipSuccess || rate <= 0) return 100000000;
  return rate * 1000;
}
}
