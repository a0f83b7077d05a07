// Stream-K fp32 GEMM (streamk_gemm.hip): the argument block and the launcher, shared with the entry points of tile_gemm.hip.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace cgv {

constexpr int SK_BM = 128, SK_BN = 128, SK_BK = 32, SK_THREADS = 512;
constexpr int SK_LDA = SK_BK + 4;                  // [row][k] slabs: 36 floats per row (conflict-free ds_read_b128, tile_gemm.hip)
constexpr int SK_LDB = SK_BN + 4;                  // NN: [k][n] slab, 132 floats per row
constexpr int SK_A_FLOATS = SK_BM * SK_LDA;        // 4608
constexpr int SK_B_FLOATS = SK_BN * SK_LDA;        // NT 4608 (NN needs 32 * 132 = 4224)
constexpr int SK_LDS_BYTES = 2 * (SK_A_FLOATS + SK_B_FLOATS) * 4;
constexpr int SK_PART_FLOATS = SK_BM * SK_BN;      // one partial tile
constexpr int SK_MAX_GRID = 512;

struct SkProblem {
  const float* A[2];           // [M, R[s]] row-major; A[1] == nullptr: one source
  const float* B[2];           // NT: [N, R[s]]; NN: [R[s], N]
  int R[2];
  float* out;                  // [M, N]
  // NT epilogue: z = acc + bias ; zout = z (if act) ; out = act(z)
  const float* bias;
  float* zout;
  int act;
  // NN epilogue: out = (acc + add + bcast) * act'(oz)
  const float* add;
  const float* bc_src;         // one row per segment of the rows (BcastAdd of tile_gemm.hip), or nullptr
  const int64_t* bc_row2seg;
  const int* bc_rowptr;
  int bc_mean;
  const float* oz;
  int oact;
};

struct SkArgs {
  SkProblem p[2];
  int np, M, N;
  int tiles_m, tiles_n;
  int slabs0, slabs;           // slabs of source 0, of both sources
  long long units;             // np * tiles_m * tiles_n * slabs
  float* part;                 // [2 * grid][SK_PART_FLOATS]
  unsigned* ticket;            // [np * tiles], zero between launches (self-resetting): low half = parts published, high half = shares finished
};

size_t sk_workspace_part_bytes(int grid);
/* Fills the geometry of ``a`` (p[], np, M, N set by the caller) and launches; ws = [tickets: ticket_bytes][parts].
 * Returns 0, or -1 with nothing launched when the workspace is missing / too small. */
int sk_launch(SkArgs& a, bool nn, void* ws, size_t ws_bytes, size_t ticket_bytes, hipStream_t st, int blocks_per_cu);

}  // namespace cgv
