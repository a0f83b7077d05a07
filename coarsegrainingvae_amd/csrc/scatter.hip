// K1: segment reduction = torch_scatter.scatter_add / scatter_mean over a CSR view of the
// index (reference call sites: cgvae.py:297-298, 479; conv.py:553-561 when used unfused).
//
// HBM-streaming kernel: every src row is read exactly once as float4 (16 B/lane, 1 KiB per
// wave instruction), 8 rows in flight per thread; one workgroup owns (segment, 1024-channel
// tile) so the output is written once, no atomics, fixed summation order.
// Algorithmic bytes per launch: 4*E*C (src) + 4*E (perm/index) + 4*Nseg*C (out).
#include "cgv_common.h"

namespace cgv {

template <int VEC>
struct vec_t;
template <> struct vec_t<4> { using type = float4; };
template <> struct vec_t<2> { using type = float2; };
template <> struct vec_t<1> { using type = float; };

template <int VEC> __device__ inline void vzero(typename vec_t<VEC>::type& a);
template <> __device__ inline void vzero<4>(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ inline void vzero<2>(float2& a) { a = make_float2(0.f, 0.f); }
template <> __device__ inline void vzero<1>(float& a) { a = 0.f; }
__device__ inline void vadd(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ inline void vadd(float2& a, const float2& b) { a.x += b.x; a.y += b.y; }
__device__ inline void vadd(float& a, const float& b) { a += b; }
__device__ inline void vscale(float4& a, float s) { a.x *= s; a.y *= s; a.z *= s; a.w *= s; }
__device__ inline void vscale(float2& a, float s) { a.x *= s; a.y *= s; }
__device__ inline void vscale(float& a, float s) { a *= s; }

// grid = (n_seg, ceil(C / (BLOCK*VEC))); rows summed in CSR order.
template <int VEC, int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void segment_reduce_k(const float* __restrict__ src, const int* __restrict__ rowptr,
                                                          const int* __restrict__ perm, int C, int mean,
                                                          float* __restrict__ out) {
  using V = typename vec_t<VEC>::type;
  const int seg = blockIdx.x;
  const int c = (blockIdx.y * BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const int beg = rowptr[seg], end = rowptr[seg + 1];
  V acc;
  vzero<VEC>(acc);
  int p = beg;
  for (; p + UNROLL <= end; p += UNROLL) {
    V x[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int row = perm ? perm[p + u] : p + u;
      x[u] = *reinterpret_cast<const V*>(src + (size_t)row * C + c);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) vadd(acc, x[u]);
  }
  if (p < end) {                      // the last < UNROLL rows: one batch of clamped loads (a row-by-row tail is one
    V x[UNROLL];                      // memory round trip per row: 5 of a 125-row segment's ~20), same summation order
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int pp = min(p + u, end - 1);
      const int row = perm ? perm[pp] : pp;
      x[u] = *reinterpret_cast<const V*>(src + (size_t)row * C + c);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (p + u < end) vadd(acc, x[u]);
  }
  if (mean) {
    const int len = end - beg;
    vscale(acc, 1.0f / (float)(len > 1 ? len : 1));
  }
  *reinterpret_cast<V*>(out + (size_t)seg * C + c) = acc;
}

// Two reductions over ONE index in one launch (the encoder's H = scatter_mean(h), V = scatter_mean(v), cgvae.py:297-298):
// blockIdx.y < tiles_a serves (src_a, C_a), the rest (src_b, C_b).  Same per-segment order as segment_reduce_k.
template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void segment_reduce2_k(const float* __restrict__ src_a, int C_a, float* __restrict__ out_a,
                                                           const float* __restrict__ src_b, int C_b, float* __restrict__ out_b,
                                                           int tiles_a, const int* __restrict__ rowptr,
                                                           const int* __restrict__ perm, int mean) {
  const bool second = (int)blockIdx.y >= tiles_a;
  const float* __restrict__ src = second ? src_b : src_a;
  float* __restrict__ out = second ? out_b : out_a;
  const int C = second ? C_b : C_a;
  const int seg = blockIdx.x;
  const int c = (((int)blockIdx.y - (second ? tiles_a : 0)) * BLOCK + threadIdx.x) * 4;
  if (c >= C) return;
  const int beg = rowptr[seg], end = rowptr[seg + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = beg; p < end; p += UNROLL) {
    float4 x[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int pp = min(p + u, end - 1);
      const int row = perm ? perm[pp] : pp;
      x[u] = *reinterpret_cast<const float4*>(src + (size_t)row * C + c);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (p + u < end) vadd(acc, x[u]);
  }
  if (mean) {
    const int len = end - beg;
    vscale(acc, 1.0f / (float)(len > 1 ? len : 1));
  }
  *reinterpret_cast<float4*>(out + (size_t)seg * C + c) = acc;
}

template <int VEC, int BLOCK>
__global__ __launch_bounds__(BLOCK) void segment_broadcast_k(const float* __restrict__ gout,
                                                             const int* __restrict__ rowptr,
                                                             const int* __restrict__ perm, int C, int mean,
                                                             float* __restrict__ gsrc) {
  using V = typename vec_t<VEC>::type;
  const int seg = blockIdx.x;
  const int c = (blockIdx.y * BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const int beg = rowptr[seg], end = rowptr[seg + 1];
  V g = *reinterpret_cast<const V*>(gout + (size_t)seg * C + c);
  if (mean) {
    const int len = end - beg;
    vscale(g, 1.0f / (float)(len > 1 ? len : 1));
  }
  for (int p = beg; p < end; ++p) {
    const int row = perm ? perm[p] : p;
    *reinterpret_cast<V*>(gsrc + (size_t)row * C + c) = g;
  }
}

__global__ __launch_bounds__(256) void embedding_rows_k(const float* __restrict__ weight, const float* __restrict__ ids,
                                                        int id_stride, int n_types, int channels, float* __restrict__ out) {
  const int i = blockIdx.x;
  const int c4 = blockIdx.y * 256 + threadIdx.x;
  if (c4 * 4 >= channels) return;
  int t = (int)ids[(size_t)i * id_stride];
  t = t < 0 ? 0 : (t >= n_types ? n_types - 1 : t);                       // (nn.Embedding would raise: ids are validated on the host once)
  reinterpret_cast<float4*>(out + (size_t)i * channels)[c4] = reinterpret_cast<const float4*>(weight + (size_t)t * channels)[c4];
}

// Two embedding lookups in one launch (the encoder's atom types and the prior's bead types, cgvae.py:268 / 381): rows
// [0, a.n_rows) belong to job a, the rest to job b.  And their weight gradients: two segment sums over their own plans.
struct EmbJob { const float* weight; const float* ids; int id_stride, n_types, n_rows; float* out; };
__global__ __launch_bounds__(256) void embedding_rows2_k(EmbJob a, EmbJob b, int channels) {
  const bool second = (int)blockIdx.x >= a.n_rows;
  const EmbJob j = second ? b : a;
  const int i = (int)blockIdx.x - (second ? a.n_rows : 0);
  const int c4 = blockIdx.y * 256 + threadIdx.x;
  if (c4 * 4 >= channels) return;
  int t = (int)j.ids[(size_t)i * j.id_stride];
  t = t < 0 ? 0 : (t >= j.n_types ? j.n_types - 1 : t);
  reinterpret_cast<float4*>(j.out + (size_t)i * channels)[c4] = reinterpret_cast<const float4*>(j.weight + (size_t)t * channels)[c4];
}

struct SegJob { const float* src; const int* rowptr; const int* perm; int n_seg; float* out; };
template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void segment_reduce_jobs2_k(SegJob a, SegJob b, int C) {
  const bool second = (int)blockIdx.x >= a.n_seg;
  const SegJob j = second ? b : a;
  const int seg = (int)blockIdx.x - (second ? a.n_seg : 0);
  const int c = (blockIdx.y * BLOCK + threadIdx.x) * 4;
  if (c >= C) return;
  const int beg = j.rowptr[seg], end = j.rowptr[seg + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = beg; p < end; p += UNROLL) {                          // (same per-segment order as segment_reduce_k)
    float4 x[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int pp = min(p + u, end - 1);
      const int row = j.perm ? j.perm[pp] : pp;
      x[u] = *reinterpret_cast<const float4*>(j.src + (size_t)row * C + c);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (p + u < end) vadd(acc, x[u]);
  }
  *reinterpret_cast<float4*>(j.out + (size_t)seg * C + c) = acc;
}

}  // namespace cgv

extern "C" {

int cgv_embedding_rows2(const float* weight_a, const float* ids_a, int id_stride_a, int n_rows_a, int n_types_a, float* out_a,
                        const float* weight_b, const float* ids_b, int id_stride_b, int n_rows_b, int n_types_b, float* out_b,
                        int channels, void* stream) {
  CGV_REQUIRE(n_rows_a >= 1 && n_rows_b >= 1 && channels > 0 && n_types_a > 0 && n_types_b > 0 && id_stride_a >= 1 && id_stride_b >= 1, "bad size");
  CGV_REQUIRE(weight_a && ids_a && out_a && weight_b && ids_b && out_b, "null pointer");
  CGV_REQUIRE((channels % 4) == 0 && ((((uintptr_t)weight_a) | ((uintptr_t)out_a) | ((uintptr_t)weight_b) | ((uintptr_t)out_b)) & 15) == 0,
              "need channels % 4 == 0, 16-byte aligned");
  const cgv::EmbJob a{weight_a, ids_a, id_stride_a, n_types_a, n_rows_a, out_a}, b{weight_b, ids_b, id_stride_b, n_types_b, n_rows_b, out_b};
  hipLaunchKernelGGL(cgv::embedding_rows2_k, dim3(n_rows_a + n_rows_b, (channels / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b,
                     channels);
  return cgv::check_launch("cgv_embedding_rows2");
}

/* out_a[s] = sum of the rows of src_a in segment s of (rowptr_a, perm_a), likewise b: two segment sums over their own plans
 * in one launch (the weight gradients of the two embeddings of a step); channels % 4 == 0, 16-byte aligned. */
int cgv_segment_reduce_pair(const float* src_a, const int32_t* rowptr_a, const int32_t* perm_a, int n_seg_a, float* out_a,
                            const float* src_b, const int32_t* rowptr_b, const int32_t* perm_b, int n_seg_b, float* out_b,
                            int channels, void* stream) {
  CGV_REQUIRE(n_seg_a >= 1 && n_seg_b >= 1 && channels > 0, "bad size");
  CGV_REQUIRE(src_a && rowptr_a && out_a && src_b && rowptr_b && out_b, "null pointer");
  CGV_REQUIRE((channels % 4) == 0 && ((((uintptr_t)src_a) | ((uintptr_t)out_a) | ((uintptr_t)src_b) | ((uintptr_t)out_b)) & 15) == 0,
              "need channels % 4 == 0, 16-byte aligned");
  constexpr int BLOCK = 256, UNROLL = 8;
  const cgv::SegJob a{src_a, rowptr_a, perm_a, n_seg_a, out_a}, b{src_b, rowptr_b, perm_b, n_seg_b, out_b};
  hipLaunchKernelGGL((cgv::segment_reduce_jobs2_k<BLOCK, UNROLL>), dim3(n_seg_a + n_seg_b, (channels / 4 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0,
                     (hipStream_t)stream, a, b, channels);
  return cgv::check_launch("cgv_segment_reduce_pair");
}

int cgv_segment_reduce(const float* src, const int32_t* rowptr, const int32_t* perm, int n_seg, int channels, int mean,
                       float* out, void* stream) {
  CGV_REQUIRE(n_seg >= 0 && channels > 0, "bad size");
  if (n_seg == 0) return 0;
  CGV_REQUIRE(src && rowptr && out, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  constexpr int BLOCK = 256, UNROLL = 8;
  const bool a16 = (((uintptr_t)src | (uintptr_t)out) & 15) == 0;
  if (channels % 4 == 0 && a16) {
    dim3 grid(n_seg, (channels / 4 + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL((cgv::segment_reduce_k<4, BLOCK, UNROLL>), grid, dim3(BLOCK), 0, st, src, rowptr, perm, channels,
                       mean, out);
  } else {
    dim3 grid(n_seg, (channels + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL((cgv::segment_reduce_k<1, BLOCK, UNROLL>), grid, dim3(BLOCK), 0, st, src, rowptr, perm, channels,
                       mean, out);
  }
  return cgv::check_launch("cgv_segment_reduce");
}

int cgv_segment_reduce2(const float* src_a, int channels_a, float* out_a, const float* src_b, int channels_b, float* out_b,
                        const int32_t* rowptr, const int32_t* perm, int n_seg, int mean, void* stream) {
  CGV_REQUIRE(n_seg >= 0 && channels_a > 0 && channels_b > 0, "bad size");
  if (n_seg == 0) return 0;
  CGV_REQUIRE(src_a && src_b && out_a && out_b && rowptr, "null pointer");
  CGV_REQUIRE((channels_a % 4) == 0 && (channels_b % 4) == 0 &&
              ((((uintptr_t)src_a) | ((uintptr_t)src_b) | ((uintptr_t)out_a) | ((uintptr_t)out_b)) & 15) == 0,
              "need channel counts that are multiples of 4 and 16-byte aligned operands");
  constexpr int BLOCK = 256, UNROLL = 8;
  const int ta = (channels_a / 4 + BLOCK - 1) / BLOCK, tb = (channels_b / 4 + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL((cgv::segment_reduce2_k<BLOCK, UNROLL>), dim3(n_seg, ta + tb), dim3(BLOCK), 0, (hipStream_t)stream, src_a,
                     channels_a, out_a, src_b, channels_b, out_b, ta, rowptr, perm, mean);
  return cgv::check_launch("cgv_segment_reduce2");
}

/* nn.Embedding lookup with the type ids read as they sit in the batch (cgvae.py:268, 381: ids = nxyz[:, 0], a float
 * column): out[i, :] = weight[(int) ids[i * id_stride], :] in one launch (the tensor-op route: cast, fills, gather). */
int cgv_embedding_rows(const float* weight, const float* ids_f32, int id_stride, int n_rows, int n_types, int channels,
                       float* out, void* stream) {
  CGV_REQUIRE(n_rows >= 0 && channels > 0 && n_types > 0 && id_stride >= 1, "bad size");
  if (n_rows == 0) return 0;
  CGV_REQUIRE(weight && ids_f32 && out, "null pointer");
  CGV_REQUIRE((channels % 4) == 0 && ((((uintptr_t)weight) | ((uintptr_t)out)) & 15) == 0, "need channels % 4 == 0, 16-byte aligned");
  dim3 grid(n_rows, (channels / 4 + 255) / 256);
  hipLaunchKernelGGL(cgv::embedding_rows_k, grid, dim3(256), 0, (hipStream_t)stream, weight, ids_f32, id_stride, n_types,
                     channels, out);
  return cgv::check_launch("cgv_embedding_rows");
}

int cgv_segment_broadcast(const float* gout, const int32_t* rowptr, const int32_t* perm, int n_seg, int channels,
                          int mean, float* gsrc, void* stream) {
  CGV_REQUIRE(n_seg >= 0 && channels > 0, "bad size");
  if (n_seg == 0) return 0;
  CGV_REQUIRE(gout && rowptr && gsrc, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  constexpr int BLOCK = 256;
  const bool a16 = (((uintptr_t)gout | (uintptr_t)gsrc) & 15) == 0;
  if (channels % 4 == 0 && a16) {
    dim3 grid(n_seg, (channels / 4 + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL((cgv::segment_broadcast_k<4, BLOCK>), grid, dim3(BLOCK), 0, st, gout, rowptr, perm, channels,
                       mean, gsrc);
  } else {
    dim3 grid(n_seg, (channels + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL((cgv::segment_broadcast_k<1, BLOCK>), grid, dim3(BLOCK), 0, st, gout, rowptr, perm, channels,
                       mean, gsrc);
  }
  return cgv::check_launch("cgv_segment_broadcast");
}

}  // extern "C"
