// Measurement only: how fast does an Adam-like read-modify-write pass over p / m / v ([N][K] fp32, K = 600: row pitch 2400 B)
// stream as a function of the block's access footprint per step?  A block of 4 waves covers (16 WR) rows x (64 WC) columns
// per step (a wave: 16 rows x 64 columns, a lane: one float4 of 4 rows per request -- the register map of the MFMA tile
// of gathered_wgrad_strip_k) and walks along K.  WR x WC = 4 x 1 is that kernel's footprint (64 rows x 256 B per step),
// 1 x 4 is 16 rows x 1 KB.  "rows" is the footprint of grouped_wgrad_t<true>: 8 whole rows (19 KB contiguous) per pass.
// Build: hipcc --offload-arch=gfx950 -O3 -o adam_pattern_probe adam_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void adam(f4& p, f4& m, f4& v, float g) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    m[c] = 0.9f * m[c] + 0.1f * g;
    v[c] = 0.999f * v[c] + 0.001f * g * g;
    p[c] -= 1e-4f * m[c] / (sqrtf(v[c]) + 1e-8f);
  }
}

template <int WR, int WC, bool PREFETCH>
__global__ __launch_bounds__(256) void tile_walk_k(float* __restrict__ P, float* __restrict__ Mo, float* __restrict__ V, int N, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int i = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * 16 * WR + 16 * wr + 4 * q;
  const int steps = (K + 64 * WC - 1) / (64 * WC);
  f4 p[2][4], m[2][4], v[2][4];
  auto load = [&](int st, f4 (&pp)[4], f4 (&mm)[4], f4 (&vv)[4]) {
    int col = st * 64 * WC + 64 * wc + 4 * i;
    if (col >= K) col = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = (size_t)(row0 + r < N ? row0 + r : 0) * K + col;
      pp[r] = *reinterpret_cast<const f4*>(P + o);
      mm[r] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(Mo + o));
      vv[r] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(V + o));
    }
  };
  auto finish = [&](int st, f4 (&pp)[4], f4 (&mm)[4], f4 (&vv)[4]) {
    const int col = st * 64 * WC + 64 * wc + 4 * i;
    if (col >= K) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (row0 + r >= N) continue;
      const size_t o = (size_t)(row0 + r) * K + col;
      adam(pp[r], mm[r], vv[r], 1e-3f * (float)((row0 + r + col) & 7));
      *reinterpret_cast<f4*>(P + o) = pp[r];
      __builtin_nontemporal_store(mm[r], reinterpret_cast<f4*>(Mo + o));
      __builtin_nontemporal_store(vv[r], reinterpret_cast<f4*>(V + o));
    }
  };
  if (PREFETCH) {
    load(0, p[0], m[0], v[0]);
    for (int st = 0; st < steps; st += 2) {
      load(st + 1 < steps ? st + 1 : st, p[1], m[1], v[1]);
      finish(st, p[0], m[0], v[0]);
      if (st + 1 < steps) {
        load(st + 2 < steps ? st + 2 : st + 1, p[0], m[0], v[0]);
        finish(st + 1, p[1], m[1], v[1]);
      }
    }
  } else {
    for (int st = 0; st < steps; ++st) {
      load(st, p[0], m[0], v[0]);
      finish(st, p[0], m[0], v[0]);
    }
  }
}

// 8 whole rows per pass, a float4 column per thread (150 of 256 threads live at K = 600), 4 passes per block
__global__ __launch_bounds__(256) void rows_k(float* __restrict__ P, float* __restrict__ Mo, float* __restrict__ V, int N, int K) {
  const int col = 4 * threadIdx.x;
  if (col >= K) return;
  for (int pass = 0; pass < 4; ++pass) {
    const int row0 = blockIdx.x * 32 + 8 * pass;
    f4 p[8], m[8], v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const size_t o = (size_t)(row0 + r < N ? row0 + r : 0) * K + col;
      p[r] = *reinterpret_cast<const f4*>(P + o);
      m[r] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(Mo + o));
      v[r] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(V + o));
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (row0 + r >= N) continue;
      const size_t o = (size_t)(row0 + r) * K + col;
      adam(p[r], m[r], v[r], 1e-3f * (float)((row0 + r + col) & 7));
      *reinterpret_cast<f4*>(P + o) = p[r];
      __builtin_nontemporal_store(m[r], reinterpret_cast<f4*>(Mo + o));
      __builtin_nontemporal_store(v[r], reinterpret_cast<f4*>(V + o));
    }
  }
}

// the footprint of grouped_wgrad_t<true> at K = 600: a block owns 64 rows x TW columns (TW = 200: 50 of a wave's 64 lanes
// hold a float4 column), wave g takes the passes g, g + 4, ... of ROWS rows
template <int TW, int ROWS>
__global__ __launch_bounds__(256) void block_tile_k(float* __restrict__ P, float* __restrict__ Mo, float* __restrict__ V, int N, int K) {
  const int tiles_k = (K + TW - 1) / TW;
  const int rb = blockIdx.x / tiles_k, kt = blockIdx.x - rb * tiles_k;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int col = kt * TW + 4 * lane;
  if (4 * lane >= TW || col >= K) return;
  for (int pass = grp; pass < 64 / ROWS; pass += 4) {
    const int row0 = rb * 64 + pass * ROWS;
    f4 p[ROWS], m[ROWS], v[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const size_t o = (size_t)(row0 + r < N ? row0 + r : 0) * K + col;
      p[r] = *reinterpret_cast<const f4*>(P + o);
      m[r] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(Mo + o));
      v[r] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(V + o));
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (row0 + r >= N) continue;
      const size_t o = (size_t)(row0 + r) * K + col;
      adam(p[r], m[r], v[r], 1e-3f * (float)((row0 + r + col) & 7));
      *reinterpret_cast<f4*>(P + o) = p[r];
      __builtin_nontemporal_store(m[r], reinterpret_cast<f4*>(Mo + o));
      __builtin_nontemporal_store(v[r], reinterpret_cast<f4*>(V + o));
    }
  }
}

// a block owns RB whole rows (RB x K contiguous floats) and walks them FLAT: thread t takes the float4 elements t + 256 j of the
// region, U at a time in flight -- full waves, 1 KB per wave request, whatever K is (row / column of an element by division)
template <int RB, int U>
__global__ __launch_bounds__(256) void flat_rows_k(float* __restrict__ P, float* __restrict__ Mo, float* __restrict__ V, int N, int K) {
  const int row0 = blockIdx.x * RB;
  const int rows = min(RB, N - row0);
  const int n4 = rows * (K >> 2);
  const size_t base = (size_t)row0 * K;
  for (int j0 = threadIdx.x; j0 < n4; j0 += 256 * U) {
    f4 p[U], m[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = j0 + 256 * u < n4 ? j0 + 256 * u : j0;
      const size_t o = base + 4 * (size_t)idx;
      p[u] = *reinterpret_cast<const f4*>(P + o);
      m[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(Mo + o));
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(V + o));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = j0 + 256 * u;
      if (idx >= n4) continue;
      const size_t o = base + 4 * (size_t)idx;
      const int row = idx / (K >> 2), col = idx - row * (K >> 2);
      adam(p[u], m[u], v[u], 1e-3f * (float)((row0 + row + 4 * col) & 7));
      *reinterpret_cast<f4*>(P + o) = p[u];
      __builtin_nontemporal_store(m[u], reinterpret_cast<f4*>(Mo + o));
      __builtin_nontemporal_store(v[u], reinterpret_cast<f4*>(V + o));
    }
  }
}

// the same, PERSISTENT: gridDim.x blocks, block b takes the row chunks b, b + gridDim.x, ...
template <int RB, int U>
__global__ __launch_bounds__(256) void flat_rows_persistent_k(float* __restrict__ P, float* __restrict__ Mo, float* __restrict__ V, int N, int K) {
  const int chunks = (N + RB - 1) / RB;
  for (int ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
    const int row0 = ch * RB;
    const int rows = min(RB, N - row0);
    const int n4 = rows * (K >> 2);
    const size_t base = (size_t)row0 * K;
    for (int j0 = threadIdx.x; j0 < n4; j0 += 256 * U) {
      f4 p[U], m[U], v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = j0 + 256 * u < n4 ? j0 + 256 * u : j0;
        const size_t o = base + 4 * (size_t)idx;
        p[u] = *reinterpret_cast<const f4*>(P + o);
        m[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(Mo + o));
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(V + o));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = j0 + 256 * u;
        if (idx >= n4) continue;
        const size_t o = base + 4 * (size_t)idx;
        adam(p[u], m[u], v[u], 1e-3f * (float)((row0 + idx) & 7));
        *reinterpret_cast<f4*>(P + o) = p[u];
        __builtin_nontemporal_store(m[u], reinterpret_cast<f4*>(Mo + o));
        __builtin_nontemporal_store(v[u], reinterpret_cast<f4*>(V + o));
      }
    }
  }
}

// flat: the layout-blind parameter pass (adam_update): consecutive float4 per thread
__global__ __launch_bounds__(256) void flat_k(float* __restrict__ P, float* __restrict__ Mo, float* __restrict__ V, size_t n4) {
  for (size_t at = (size_t)blockIdx.x * 1024 + threadIdx.x; at < n4; at += (size_t)gridDim.x * 1024) {
    f4 p[4], m[4], v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t o = 4 * (at + 256 * u < n4 ? at + 256 * u : at);
      p[u] = *reinterpret_cast<const f4*>(P + o);
      m[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(Mo + o));
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(V + o));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (at + 256 * u >= n4) continue;
      const size_t o = 4 * (at + 256 * u);
      adam(p[u], m[u], v[u], 1e-3f);
      *reinterpret_cast<f4*>(P + o) = p[u];
      __builtin_nontemporal_store(m[u], reinterpret_cast<f4*>(Mo + o));
      __builtin_nontemporal_store(v[u], reinterpret_cast<f4*>(V + o));
    }
  }
}

// cold = a 1 GB scratch buffer is rewritten between the timed launches (L2 and the 256 MB Infinity Cache hold none of p / m / v:
// the condition of the pass inside a training step); warm = back to back
static float* g_scratch = nullptr;
static size_t g_scratch_n4 = 0;
__global__ __launch_bounds__(256) void flush_k(f4* __restrict__ s, size_t n4, float v) {
  for (size_t at = (size_t)blockIdx.x * 256 + threadIdx.x; at < n4; at += (size_t)gridDim.x * 256) s[at] = f4{v, v, v, v};
}
template <typename F>
static void timeit(const char* name, F launch, double bytes) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 2; ++w) launch();
  CK(hipDeviceSynchronize());
  float best[2] = {1e30f, 1e30f}, sum[2] = {0.f, 0.f};
  const int reps = 8;
  for (int cold = 0; cold < 2; ++cold) {
    for (int r = 0; r < reps; ++r) {
      if (cold) hipLaunchKernelGGL(flush_k, dim3(4096), dim3(256), 0, 0, reinterpret_cast<f4*>(g_scratch), g_scratch_n4, (float)r);
      CK(hipEventRecord(a, 0));
      launch();
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      best[cold] = ms < best[cold] ? ms : best[cold]; sum[cold] += ms;
    }
  }
  printf("%-36s warm best %6.1f mean %6.1f us (%4.2f TB/s)   cold best %6.1f mean %6.1f us (%4.2f TB/s)\n", name, best[0] * 1e3,
         sum[0] / reps * 1e3, bytes / (best[0] * 1e-3) / 1e12, best[1] * 1e3, sum[1] / reps * 1e3, bytes / (best[1] * 1e-3) / 1e12);
}

int main() {
  const int K = 600, N = 76800;                       // 46 M weights, the bead-level layers of a chignolin step
  const size_t n = (size_t)N * K;
  float *P, *Mo, *V;
  CK(hipMalloc(&P, 4 * n)); CK(hipMalloc(&Mo, 4 * n)); CK(hipMalloc(&V, 4 * n));
  CK(hipMemset(P, 0, 4 * n)); CK(hipMemset(Mo, 0, 4 * n)); CK(hipMemset(V, 0, 4 * n));
  g_scratch_n4 = (size_t)1 << 26;                      // 1 GiB
  CK(hipMalloc(&g_scratch, 16 * g_scratch_n4));
  const double bytes = 24.0 * n;
  printf("N = %d rows, K = %d (pitch %d B): %.0f M weights, %.2f GB moved per pass\n", N, K, 4 * K, n / 1e6, bytes / 1e9);
  timeit("flat (adam_update)", [&] { hipLaunchKernelGGL(flat_k, dim3(2048), dim3(256), 0, 0, P, Mo, V, n / 4); }, bytes);
  timeit("8 whole rows per pass", [&] { hipLaunchKernelGGL(rows_k, dim3((N + 31) / 32), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes);
  timeit("block 64 x 200, 8-row passes", [&] { hipLaunchKernelGGL((block_tile_k<200, 8>), dim3((N + 63) / 64 * 3), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes);
  timeit("block 64 x 200, 4-row passes", [&] { hipLaunchKernelGGL((block_tile_k<200, 4>), dim3((N + 63) / 64 * 3), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes);
  timeit("block 64 x 256 (+88), 8-row passes", [&] { hipLaunchKernelGGL((block_tile_k<256, 8>), dim3((N + 63) / 64 * 3), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes);
  timeit("flat inside 64 rows, 4 in flight", [&] { hipLaunchKernelGGL((flat_rows_k<64, 4>), dim3((N + 63) / 64), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes);
  timeit("flat inside 64 rows, 8 in flight", [&] { hipLaunchKernelGGL((flat_rows_k<64, 8>), dim3((N + 63) / 64), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes);
  timeit("flat inside 16 rows, 4 in flight", [&] { hipLaunchKernelGGL((flat_rows_k<16, 4>), dim3((N + 15) / 16), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes);
  timeit("flat inside 32 rows, 4 in flight", [&] { hipLaunchKernelGGL((flat_rows_k<32, 4>), dim3((N + 31) / 32), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes);
  timeit("persistent 2048 x flat 16 rows", [&] { hipLaunchKernelGGL((flat_rows_persistent_k<16, 4>), dim3(2048), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes);
  timeit("persistent 1024 x flat 16 rows", [&] { hipLaunchKernelGGL((flat_rows_persistent_k<16, 4>), dim3(1024), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes);
  timeit("persistent 2048 x flat 64 rows", [&] { hipLaunchKernelGGL((flat_rows_persistent_k<64, 4>), dim3(2048), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes);
  timeit("flat, 8192 blocks", [&] { hipLaunchKernelGGL(flat_k, dim3(8192), dim3(256), 0, 0, P, Mo, V, n / 4); }, bytes);
  timeit("flat, 45000 blocks (one trip)", [&] { hipLaunchKernelGGL(flat_k, dim3((unsigned)((n / 4 + 1023) / 1024)), dim3(256), 0, 0, P, Mo, V, n / 4); }, bytes);
#define TW(WR, WC) \
  timeit("tile walk " #WR " x " #WC, [&] { hipLaunchKernelGGL((tile_walk_k<WR, WC, false>), dim3((N + 16 * WR - 1) / (16 * WR)), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes); \
  timeit("tile walk " #WR " x " #WC " prefetch", [&] { hipLaunchKernelGGL((tile_walk_k<WR, WC, true>), dim3((N + 16 * WR - 1) / (16 * WR)), dim3(256), 0, 0, P, Mo, V, N, K); }, bytes);
  TW(4, 1) TW(2, 2) TW(1, 4)
  return 0;
}
