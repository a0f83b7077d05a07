// Measurement only: how fast can ONE short kernel of a dependent chain pull B bytes of weights that live in HBM
// (not in L2 / MALL)?  Variants of block count, access pattern and bytes per block; each launch reads a different
// buffer of a 1.5 GB pool so nothing is cached.  Build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// pattern 0: fwd_core-like: the block owns ROWS rows of K floats; wave w takes 16-float steps [w*per, (w+1)*per) of every
//            row; lane (i = row in a 16-row tile, q): float4 at row*K + step*16 + 4 q   (16 rows x 64 B per instruction)
// pattern 1: bi_core-like: wave w takes 64-column tiles; lane (j, q): float4 at (row0 + q) * K + tile*64 + 4 j  (4 rows x 256 B)
// pattern 2: flat: the block's bytes as one contiguous range, 1 KB per wave instruction
template <int NLOAD, int PATTERN, bool NT>
__global__ __launch_bounds__(576) void pull_k(const float* __restrict__ W, float* __restrict__ out, int rows_per_block, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* base = W + (size_t)blockIdx.x * rows_per_block * K;
  float4 r[NLOAD];
  typedef float f4v __attribute__((ext_vector_type(4)));
  auto ld = [](const float* p) {
    if (NT) { f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p)); return make_float4(t.x, t.y, t.z, t.w); }
    return *reinterpret_cast<const float4*>(p);
  };
  const size_t block_floats = (size_t)rows_per_block * K;
  if (PATTERN == 0) {
    const int i = lane & 15, q = lane >> 4;
    const int tiles = rows_per_block / 16 > 0 ? rows_per_block / 16 : 1;
    const int steps = K / 16, per = (steps + 8) / 9;
#pragma unroll
    for (int u = 0; u < NLOAD; ++u) {
      const int t = u % tiles, s = wave * per + u / tiles;
      size_t off = (size_t)((t * 16 + i) % rows_per_block) * K + (size_t)(s < steps ? s : 0) * 16 + 4 * q;
      r[u] = ld(base + off);
    }
  } else if (PATTERN == 1) {
    const int j = lane & 15, q = lane >> 4;
    const int groups = rows_per_block / 4;
#pragma unroll
    for (int u = 0; u < NLOAD; ++u) {
      const int g = u % groups, tile = wave + 9 * (u / groups);
      size_t off = (size_t)(g * 4 + q) * K + (size_t)((tile * 64 < K ? tile : 0) * 64) + 4 * j;
      r[u] = ld(base + off);
    }
  } else if (PATTERN == 3) {
    // 8 rows x 128 B per instruction: lane (i, q): row (i & 7) (+8 for odd instructions), k-quad q + 4 (i >> 3) of a 32-float step
    const int i = lane & 15, q = lane >> 4;
    const int tiles = rows_per_block / 16 > 0 ? rows_per_block / 16 : 1;
    const int steps32 = K / 32, per = (steps32 + 8) / 9;
#pragma unroll
    for (int u = 0; u < NLOAD; ++u) {
      const int half = u & 1, v = u >> 1;
      const int t = v % tiles, s = wave * per + v / tiles;
      size_t off = (size_t)((t * 16 + half * 8 + (i & 7)) % rows_per_block) * K + (size_t)(s < steps32 ? s : 0) * 32 + 4 * (q + 4 * (i >> 3));
      r[u] = ld(base + off);
    }
  } else if (PATTERN == 4) {
    // 4 rows x 256 B per instruction (64-float steps)
    const int i = lane & 15, q = lane >> 4;
    const int tiles = rows_per_block / 16 > 0 ? rows_per_block / 16 : 1;
    const int steps64 = K / 64, per = (steps64 + 8) / 9;
#pragma unroll
    for (int u = 0; u < NLOAD; ++u) {
      const int part = u & 3, v = u >> 2;
      const int t = v % tiles, s = wave * per + v / tiles;
      size_t off = (size_t)((t * 16 + part * 4 + (i & 3)) % rows_per_block) * K + (size_t)(s < steps64 ? s : 0) * 64 + 4 * (q + 4 * (i >> 2));
      r[u] = ld(base + off);
    }
  } else {
#pragma unroll
    for (int u = 0; u < NLOAD; ++u) {
      size_t off = ((size_t)u * 576 + threadIdx.x) * 4;
      if (off >= block_floats) off = 0;
      r[u] = ld(base + off);
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < NLOAD; ++u) acc += r[u].x + r[u].y + r[u].z + r[u].w;
  if (acc == 123.456f) out[blockIdx.x * 576 + threadIdx.x] = acc;
}

// the same flat pull as 32 distinct kernels (different code addresses, ~equal code): a chain that cycles through them
// runs every launch with a cold instruction cache, like the training step does
template <int ID>
__global__ __launch_bounds__(576) void pull_id_k(const float* __restrict__ W, float* __restrict__ out, int rows_per_block, int K) {
  const float* base = W + (size_t)blockIdx.x * rows_per_block * K;
  const size_t block_floats = (size_t)rows_per_block * K;
  float4 r[10];
#pragma unroll
  for (int u = 0; u < 10; ++u) {
    size_t off = ((size_t)u * 576 + threadIdx.x) * 4;
    if (off >= block_floats) off = 0;
    r[u] = *reinterpret_cast<const float4*>(base + off);
  }
  float acc = (float)ID;
#pragma unroll
  for (int u = 0; u < 10; ++u) acc += r[u].x * (ID + 1) + r[u].y + r[u].z + r[u].w;
  // ballast: ID-dependent arithmetic so that the variants are not merged and have a few KB of code
#pragma unroll
  for (int u = 0; u < 200; ++u) acc = acc * (1.0f + 1e-7f * (ID + u)) + (float)(u ^ ID);
  if (acc == 123.456f) out[blockIdx.x * 576 + threadIdx.x] = acc;
}
typedef void (*pull_fn)(const float*, float*, int, int);
template <int... IDS> struct IdList {};
template <int N, int... IDS> struct MakeIds : MakeIds<N - 1, N - 1, IDS...> {};
template <int... IDS> struct MakeIds<0, IDS...> { typedef IdList<IDS...> type; };
template <int... IDS> static std::vector<pull_fn> all_ids(IdList<IDS...>) { return {pull_id_k<IDS>...}; }

__global__ void tiny_k(float* out) { if (threadIdx.x == 1234567) out[0] = 1.f; }

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
  const int K = 600;
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t pool_floats = (size_t)384 << 20;      // 1.5 GB
  float* pool; CK(hipMalloc(&pool, pool_floats * 4));
  CK(hipMemset(pool, 0, pool_floats * 4));
  float* out; CK(hipMalloc(&out, 1 << 22));
  const int chain = 90;
  float t0 = chain_us([&](int) { hipLaunchKernelGGL(tiny_k, dim3(150), dim3(576), 0, st, out); }, chain, 5, st);
  printf("%-64s %7.2f us per launch\n", "empty kernel, 150 x 576 (boundary)", t0);
  {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(tiny_k), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    struct { int blocks, threads, lds; } cfg[] = {{1, 64, 0}, {75, 576, 0}, {150, 256, 0}, {256, 256, 0}, {150, 1024, 0}, {1024, 256, 0},
                                                  {150, 576, 64 * 1024}, {150, 576, 120 * 1024}, {150, 576, 150 * 1024}};
    for (auto& c : cfg) {
      float t = chain_us([&](int) { hipLaunchKernelGGL(tiny_k, dim3(c.blocks), dim3(c.threads), c.lds, st, out); }, chain, 5, st);
      printf("empty kernel, %4d x %4d, %3d KB dynamic LDS                       %7.2f us per launch\n", c.blocks, c.threads, c.lds / 1024, t);
    }
  }
  size_t cursor = 0;
  auto buf = [&](size_t floats) { if (cursor + floats > pool_floats) cursor = 0; float* p = pool + cursor; cursor += floats; return p; };
#define RUN(NLOAD, PATTERN, NT, BLOCKS, ROWS, LABEL)                                                                    \
  {                                                                                                                       \
    const size_t fl = (size_t)(BLOCKS) * (ROWS) * K;                                                                      \
    float t = chain_us([&](int) { hipLaunchKernelGGL((pull_k<NLOAD, PATTERN, NT>), dim3(BLOCKS), dim3(576), 0, st, buf(fl), out, ROWS, K); }, chain, 5, st); \
    printf("%-64s %7.2f us per launch  %6.2f MB  %5.2f TB/s incl. boundary\n", LABEL, t, fl * 4 / 1e6, fl * 4 / 1e6 / t); \
  }
  // 36 rows per block (the message product): 36 * 600 * 4 = 86 KB per block
  RUN(15, 0, false, 150, 36, "msg-like  150 blk x 36 rows, fwd_core pattern, 15 ld/lane");
  RUN(15, 0, true, 150, 36, "msg-like  150 blk x 36 rows, fwd_core pattern, nt");
  RUN(9, 1, false, 150, 36, "msg-like  150 blk x 36 rows, bi_core pattern (first tile only)");
  RUN(15, 3, false, 150, 36, "msg-like  150 blk x 36 rows, 8 rows x 128 B per instr");
  RUN(15, 4, false, 150, 36, "msg-like  150 blk x 36 rows, 4 rows x 256 B per instr");
  RUN(15, 0, false, 150, 36, "msg-like  150 blk x 36 rows, fwd_core pattern (again)");
  RUN(10, 2, false, 150, 36, "msg-like  150 blk x 86 KB flat");
  RUN(10, 2, true, 150, 36, "msg-like  150 blk x 86 KB flat, nt");
  {
    std::vector<pull_fn> fns = all_ids(MakeIds<48>::type());
    const size_t fl = (size_t)150 * 36 * K;
    for (int nv : {1, 2, 8, 48}) {
      float t = chain_us([&](int c) { hipLaunchKernelGGL(fns[c % nv], dim3(150), dim3(576), 0, st, buf(fl), out, 36, K); }, 96, 5, st);
      printf("flat pull + ballast, chain cycling through %2d distinct kernels       %7.2f us per launch\n", nv, t);
    }
    for (int nv : {1, 48}) {
      float t = chain_us([&](int c) { hipLaunchKernelGGL(fns[c % nv], dim3(150), dim3(576), 0, st, buf(4 * 150 * K), out, 4, K); }, 96, 5, st);
      printf("dense-size pull + ballast, chain cycling through %2d distinct kernels %7.2f us per launch\n", nv, t);
    }
  }
  RUN(6, 2, false, 256, 21, "same bytes over 256 blk flat (21 rows)");
  RUN(6, 2, true, 256, 21, "same bytes over 256 blk flat (21 rows), nt");
  RUN(3, 2, false, 512, 10, "same bytes over 512 blk flat (10 rows)");
  // 12 rows per block (gate product), 8 (uv), 4 (dense)
  RUN(5, 0, false, 150, 12, "gate-like 150 blk x 12 rows, fwd_core pattern");
  RUN(4, 2, false, 150, 12, "gate-like 150 blk flat");
  RUN(5, 0, false, 150, 4, "dense-like 150 blk x 4 rows, fwd_core pattern");
  RUN(1, 2, false, 150, 4, "dense-like 150 blk flat");
  RUN(30, 2, false, 150, 108, "3x msg bytes 150 blk flat (30 ld/lane)");
  RUN(18, 2, false, 256, 63, "3x msg bytes 256 blk flat (18 ld/lane)");
  return 0;
}
