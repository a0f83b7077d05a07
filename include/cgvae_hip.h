/*
 * cgvae_hip.h -- C ABI of libcgvae_hip.so: the MI355X (gfx950) kernels behind the
 * CoarseGrainingVAE message-passing hot path.
 *
 * The reference (wwang2/CoarseGrainingVAE) is pure Python: it has no FFI layer.  Its seam
 * for this path is (i) torch_scatter.scatter_add / scatter_mean and (ii) the forward()
 * signatures of the message blocks in CoarseGrainingVAE/conv.py.  Every entry point below
 * names the reference lines it replaces; INTEGRATION.md shows the ctypes stub a reference
 * maintainer would add to bind them.
 *
 * Conventions (all functions):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked [host];
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library never
 *     allocates or frees device memory and keeps no pointer after it returns;
 *   - tensors are contiguous, fp32 / int32 / int64 exactly as declared;
 *   - `stream` is a hipStream_t (NULL = default stream); launches are asynchronous and the
 *     library never synchronises the device;
 *   - return value: 0 = success; < 0 = CGV_E_* argument error; > 0 = hipError_t of a launch;
 *     cgv_last_error_string() gives a thread-local description;
 *   - re-entrant (autograd calls backward from worker threads).  The ONLY process-wide mutable state is the explicit
 *     option table below (cgv_set_option): nothing is read from the environment, and a call's result depends on its
 *     arguments and on that table alone.
 *
 * Notation: F = n_basis, R = n_rbf, E = directed edges, Nd / Ns = destination / source
 * node counts (equal for the atom and bead graphs; Nd = beads, Ns = atoms for the
 * atom->bead contraction).
 *
 * Edge geometry record ("geom"), one per edge, stride cgv_geom_stride(R) floats:
 *   [0..R)   a_n   = rbf_n(d) * env(d)          modules.py:148-172, 52-58
 *   [R]      env   = 0.5 (cos(pi d / cut) + 1)  (0 for d >= cut)
 *   [U..U+6) ux,uy,uz,ux,uy,uz with U = cgv_geom_unit_offset(R) (even); unit = r / d,
 *            d = sqrt(sum_k (r_k^2 + 1e-8))  conv.py:25-29.  Stored twice so that every adjacent
 *            pair is an aligned 64-bit scalar operand for the packed-fp32 kernels.
 * so that the distance filter of modules.py:192-197 is
 *   w[c] = sum_n Wd[c][n] * a_n + bd[c] * env.
 */
#ifndef CGVAE_HIP_H
#define CGVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGV_VERSION 100 /* 0.1.0 */

#define CGV_E_BADARG (-1)      /* null pointer / negative size */
#define CGV_E_UNSUPPORTED (-2) /* e.g. n_rbf outside the compiled set */
#define CGV_E_WORKSPACE (-3)   /* workspace too small */

int cgv_version(void);
const char* cgv_last_error_string(void);

/* ---------------------------------------------------------------------------------------
 * Option table: A/B switches of the launchers (no reference counterpart -- the reference has no kernels to choose
 * between).  Every option has a fixed default under which the library behaves as documented per entry point; the
 * alternatives exist for measurements and for the parity tests that cover the non-default kernels.
 * Thread semantics: the table is process wide; cgv_set_option is an atomic store, every launcher reads the options it
 * uses once (relaxed atomic load) at the top of the call.  Set options before issuing the launches they should
 * affect; concurrent launches from other threads see either the old or the new value.  Values do not change WHAT is
 * computed, only which kernel computes it (summation order may differ within the documented tolerance).
 * ------------------------------------------------------------------------------------- */
enum {
  CGV_OPT_MSG_FWD_SPLIT = 0,   /* cgv_equi_msg_fwd: 4 waves share a receiver's segment: -1 auto (>= 16 edges/receiver), 0, 1 */
  CGV_OPT_MSG_BWD_SPLIT = 1,   /* cgv_equi_msg_bwd: waves split a source's segment: -1 auto (>= 48 edges/source), 0, 1 */
  CGV_OPT_MSG_FWD_KERNEL = 2,  /* cgv_equi_msg_fwd: 0 packed-fp32 VALU kernel (default), 1 MFMA filter-evaluation variant */
  CGV_OPT_GRP_WAVES = 3,       /* cgv_equi_msg_fwd_grouped: waves per block, 4 (default) or 8 */
  CGV_OPT_GRP_RECORDS = 4,     /* cgv_equi_msg_fwd_grouped: record stream 0 scalar loads (default), 1 through an LDS ring */
  CGV_OPT_CSR_BUILD = 5,       /* cgv_csr_build: 0 by rows (default), 1 two-radix-pass construction */
  CGV_OPT_PSEUDO_CHUNKS = 6,   /* cgv_pseudo_msg_bwd: cap on node chunks, 0 = built-in rule */
  CGV_OPT_WGRAD_TILING = 7,    /* cgv_wgrad_plan: 0 balanced column tiles (default), 1 widest tile */
  CGV_OPT_TILE_FWD_LDS_MIN = 8,/* cgv_tile_linear_fwd: minimum 64x64 tile count for the LDS-staged kernels (default 448); 1 = the one-slab
                                  kernel always, 3 = the three-slab ring kernel wherever it is compiled (K = 577..608, 1185..1216) */
  CGV_OPT_BWD_INPUT_WAVES = 9, /* cgv_tile_linear_bwd_input*: waves per block, 0 = built-in rule */
  CGV_OPT_PSEUDO_FWD = 10,     /* cgv_pseudo_msg_fwd*: 0 built-in rule; 1..6 = (edges in flight, records staged in LDS) variants; cgv_pseudo_msg_bwd on dense bead graphs: 4 = 8 edges in flight, 5 = records not staged in LDS */
  CGV_OPT_DECODER_FAT = 11,    /* cgv_decoder_{gate,dense,uv}_bwd: 1 (default) 8-channel blocks where the width allows, 0 always 4 */
  CGV_OPT_DECODER_WLDS = 12,   /* cgv_decoder_msg_fwd: 1 (default) weight rows by LDS-DMA when they fit in LDS, 0 register path */
  CGV_OPT_SKINNY_ROWS = 13,    /* cgv_skinny_linear_fwd: row blocks (of 16) per thread block; 0 = built-in rule, 1..4 */
  CGV_OPT_TILE_FWD_BAL = 14,   /* cgv_tile_linear_fwd: 1 (default) layers of >= 1200 outputs with more than one 32 x 32 tile per CU run as ONE larger register tile per CU where a compiled tile fits (XCD-aware tile order), 0 off, 2 every shape (tests / A-B) */
  CGV_OPT_OPTIM_ONE_LAUNCH = 15, /* cgv_optim_prepare*: 0 (default) norm pass, then the decision launch; 1 both in ONE launch (the last block decides) -- measured SLOWER on the chignolin step (1.774 against 1.763 ms): 1620 blocks arriving at one device-scope ticket cost more (~12 ns each) than the launch boundary saved; 2: as 0, and cgv_wgrad_gram keeps its separate reduce launch too (A/B: by default the LAST of a problem's eight slice blocks sums them) */
  CGV_OPT_DECODER_COLSPLIT = 16, /* cgv_decoder_{gate,dense,uv}_bwd: the column tiles of a channel group's backward-input product are split over
                                  several blocks (= CUs; grid y): these products are bound by the fp32 MFMA pipe of ONE CU -- each part repeats the
                                  prologue and writes its own columns of the group's slice.  0 one block per channel group, 1 two, 2 (default) two,
                                  three for uv_bwd (48-row tiles), 3 three everywhere */
  CGV_OPT_DECODER_NODESPLIT = 17, /* cgv_decoder_uv_fwd (and uv_bwd, see there): 1 (default) 8-channel blocks by node groups of <= 5 nodes (grid y) --
                                   a third of the MFMAs and of the x rows per block; 0 one 4-channel block over all 3 n rows */
  CGV_OPT_MSG_FWD_BALANCED = 18, /* cgv_equi_msg_fwd_balanced: four-wave blocks per CU, 3 (default: what the kernel's registers admit, all resident) or 1..4 */
  CGV_OPT_BWD_INPUT_SPLIT = 19, /* cgv_tile_linear_bwd_input*: few output tiles and a long reduction: -1 (default) 2 - 4 blocks per tile
                                   when a workspace is registered (cgv_tile_bwd_input_split), 1 never, 2..4 that many */
  CGV_OPT_MSG_BWD_MFMA = 20,    /* cgv_equi_msg_bwd*, scalar-only upstream: the matrix-core kernel (4 edges per fp32 MFMA, records and
                                   rows through vector loads) -1 (default) from 200 edges per node on, 1 wherever it applies
                                   (>= 48 edges per node, n_rbf <= 15, n_feat % 4 == 0), 0 never (the packed-FMA walk) */
  CGV_OPT_STREAMK = 21,         /* cgv_tile_linear_fwd / cgv_tile_*bwd_input* (and their pair / two-source forms): the LDS-staged stream-K
                                   kernel (csrc/streamk_gemm.hip: 128 x 128 tiles, the (tile, 32-deep slab) units of a launch cut into equal
                                   ranges, one block per CU; needs the workspace of cgv_tile_bwd_input_split).  0 (default) where it
                                   wins: single launches of >= 1536 rows with <= 640 outputs over a >= 1200-deep reduction (see
                                   tile_gemm.hip: sk_wanted); 1 never; 2 / 3 wherever the operands allow, with one / two blocks per CU
                                   (tests, A-B); >= 16: a grid of exactly that many blocks (experiments) */
  CGV_OPT_COUNT = 22
};
/* Measurement: store the GPU wall clock (cgv_timestamp_hz ticks per second) into *slot, in stream order; capturable. */
int cgv_timestamp(uint64_t* slot /*device*/, void* stream);
int cgv_timestamp_hz(void);
/* Measurement: the shader clock sustained under packed fp32 FMAs on `blocks` x 4 waves: out[0] = shader cycles, out[1] =
 * wall-clock ticks of one wave's span of iters x 32 FMAs (the fused message forward's peak is quoted at 2.4 GHz; it runs at this). */
int cgv_sustained_clock_probe(uint64_t* out /*device [2]*/, float* sink /*device [1]*/, int blocks, int iters, void* stream);
int cgv_set_option(int option, int value);   /* 0, or CGV_E_BADARG for an unknown option */
int cgv_get_option(int option);              /* current value (INT32_MIN for an unknown option) */
int cgv_reset_options(void);                 /* every option back to its default */

/* n_rbf values with a compiled kernel: returns 1 if supported. */
int cgv_rbf_supported(int n_rbf);
/* floats per edge-geometry record for this n_rbf, and the offset of its unit-vector block. */
int cgv_geom_stride(int n_rbf);
int cgv_geom_unit_offset(int n_rbf);

/* ---------------------------------------------------------------------------------------
 * K0  radius graph -- replaces get_neighbor_list, CoarseGrainingVAE/data.py:65-82, batched
 * over frames (the Python loop of data.py:207-252).  Pair (i,j) of one frame is an edge iff
 *   s = (dx*dx + dy*dy) + dz*dz  <=  s_star          (no FMA contraction, fp32)
 * where s_star is the largest fp32 whose host sqrt is <= cutoff (computed by the caller with
 * the host's own sqrt, see graph.py) -- this makes membership bit-identical to the reference's
 * `sqrt(s) <= cutoff` without depending on the device sqrt.  Output order is the reference's
 * (torch.nonzero, row-major: i ascending, then j ascending), node ids are batch-global like
 * CG_collate produces (data.py:262-270).  undirected != 0 keeps j > i only (data.py:79-80).
 * Two calls because the edge count is data dependent:
 *   1. cgv_radius_graph_count -> counts[n_nodes] and offsets[n_nodes+1] (exclusive scan;
 *      offsets[n_nodes] = E).  The caller reads E back and allocates.
 *   2. cgv_radius_graph_emit  -> nbr_out[E][2] int64.
 * ------------------------------------------------------------------------------------- */
int cgv_radius_graph_count(const float* xyz /*[n_nodes,3]*/, const int32_t* frame_ptr /*[n_frames+1]*/,
                           int n_frames, int n_nodes, float s_star, int undirected,
                           int32_t* counts /*[n_nodes]*/, int32_t* offsets /*[n_nodes+1]*/, void* stream);
int cgv_radius_graph_emit(const float* xyz, const int32_t* frame_ptr, int n_frames, int n_nodes, float s_star,
                          int undirected, const int32_t* offsets /*[n_nodes+1]*/, int64_t* nbr_out /*[E,2]*/,
                          void* stream);

/* ---------------------------------------------------------------------------------------
 * K7  CSR plan of a directed edge list.  The reference scatters with an UNSORTED index
 * (nbrs[:,0] after make_directed, conv.py:10-20,553-561); the kernels here reduce over
 * destination-sorted segments instead (no atomics, deterministic).  Builds, with a stable
 * sort (ties keep the original edge order):
 *   dst-sorted view : rowptr_d[Nd+1], eid_d[E] (original edge id), dst_d[E], src_d[E]
 *   src-sorted view : rowptr_s[Ns+1], eid_s[E], dst_s[E], src_s[E]      (for backward)
 * dst/src are int64 arrays read with element stride `stride` (nbrs[E,2]: dst=nbrs,
 * src=nbrs+1, stride=2).  src == NULL means src[e] = e (atom->bead contraction with
 * dst = CG_mapping, conv.py:725-731).
 * Construction: by rows -- count, scan, drop, one block per row ranks its unique (partner, edge id) keys: 5 launches per
 * view -- when the workspace (size it with cgv_csr_workspace_bytes(max(E, Nd, Ns) + 1)) has room for the row counters;
 * otherwise two stable radix passes per view.  Both give the same arrays.
 * ------------------------------------------------------------------------------------- */
size_t cgv_csr_workspace_bytes(int n_edges);
int cgv_csr_build(const int64_t* dst, const int64_t* src, int stride, int n_edges, int n_dst, int n_src,
                  int32_t* rowptr_d, int32_t* eid_d, int32_t* dst_d, int32_t* src_d,
                  int32_t* rowptr_s, int32_t* eid_s, int32_t* dst_s, int32_t* src_s,
                  void* workspace, size_t workspace_bytes, void* stream);
/* K7b  receiver-group order of the dst-sorted view, for the shared-source forward (cgv_equi_msg_fwd_grouped):
 * rb consecutive receivers form a group; the group's edges (one contiguous range of the dst-sorted view, so
 * rowptr_d still delimits it) are re-ordered by (source, receiver).  Outputs, all [E] in group order:
 *   dst_g, src_g  receiver / source of the edge      pos_g  its position in the dst-sorted view
 *   meta_g [E,2]  { slot | head << 8 | last << 9 | mask << 16 , source of the group's NEXT step }: slot = receiver - group
 *                 base; a step = a maximal run of edges of one (group, source) pair with strictly increasing receivers
 *                 (a duplicated edge opens a new step); mask = the step's slots; head = first edge of the step; last =
 *                 the step is its group's last one (it names its own source as the next).
 * One launch: a block per group ranks the group's edges by (source, receiver, position) -- at most 2 x degree keys,
 * in LDS up to 4096 per group.  Edge records for this order come from cgv_edge_geometry_grouped (below), which folds
 * meta_g into them. */
size_t cgv_group_plan_workspace_bytes(int n_edges);
int cgv_group_plan_build(const int32_t* rowptr_d /*[Nd+1]*/, const int32_t* dst_d, const int32_t* src_d, int n_edges,
                         int n_dst, int n_src, int rb, int32_t* dst_g, int32_t* src_g, int32_t* pos_g,
                         int32_t* meta_g /*[E,2]*/, void* workspace, size_t workspace_bytes, void* stream);
/* The same order from two stable radix passes over all edges (~20 launches against one): kept as the independent
 * construction the tests compare with. */
size_t cgv_group_plan_radix_workspace_bytes(int n_edges);
int cgv_group_plan_build_radix(const int32_t* dst_d, const int32_t* src_d, int n_edges, int n_dst, int n_src, int rb,
                               int32_t* dst_g, int32_t* src_g, int32_t* pos_g /*or NULL*/, int32_t* meta_g /*[E,2]*/,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * K6  edge geometry -- replaces preprocess_r (conv.py:25-29), PainnRadialBasis
 * (modules.py:148-172) and CosineEnvelope (modules.py:52-58), computed ONCE per graph and
 * cutoff and shared by every layer (the reference recomputes it in each block).
 * For sorted position p:  r = r_edges[eid[p]]                      if r_edges != NULL
 *                         r = pos_src[src[p]] - pos_dst[dst[p]]    otherwise
 * coef[n] = (n+1)*pi/cutoff and pi_f = (float)pi exactly as the host computes them.
 * ------------------------------------------------------------------------------------- */
int cgv_edge_geometry(const float* r_edges /*[E,3] or NULL*/, const int32_t* eid /*[E]*/,
                      const float* pos_dst /*[Nd,3]*/, const float* pos_src /*[Ns,3]*/,
                      const int32_t* dst /*[E]*/, const int32_t* src /*[E]*/, int n_edges, int n_rbf,
                      float cutoff, const float* coef /*[R]*/, float* geom /*[E,stride]*/, void* stream);
/* Records of the receiver-group order (K7b), meta words folded in; even n_rbf; stride cgv_geom_group_stride(R) floats:
 *   [0,R) a_n ; [R] env ; [R+1] meta.x (int bits) ; [R+2,R+5) ux,uy,uz ; [R+5] meta.y (int bits) ; zero padding
 * (n_rbf = 10: 16 floats = one aligned 64-byte scalar load per edge).  Same fp32 expressions as cgv_edge_geometry. */
int cgv_geom_group_stride(int n_rbf);
int cgv_geom_group_unit_offset(int n_rbf);
int cgv_edge_geometry_grouped(const float* pos_dst /*[Nd,3]*/, const float* pos_src /*[Ns,3]*/, const int32_t* dst_g,
                              const int32_t* src_g, const int32_t* meta_g /*[E,2]*/, int n_edges, int n_rbf,
                              float cutoff, const float* coef /*[R]*/, float* geom_g /*[E,group stride]*/, void* stream);

/* ---------------------------------------------------------------------------------------
 * K1  segment reduction -- replaces torch_scatter.scatter_add / scatter_mean (requirements.txt:18;
 * call sites cgvae.py:297-298,479; conv.py:553-561 when used unfused).
 *   out[s, :] = sum_{p in [rowptr[s], rowptr[s+1])} src[perm ? perm[p] : p, :]   (/ max(len,1) if mean)
 * `src` is [n_rows, C]; perm (eid_d of the CSR plan) handles an unsorted index.
 * ------------------------------------------------------------------------------------- */
int cgv_segment_reduce(const float* src, const int32_t* rowptr, const int32_t* perm, int n_seg, int channels,
                       int mean, float* out /*[n_seg,C]*/, void* stream);
/* Two reductions over one index in one launch: the encoder's H = scatter_mean(h), V = scatter_mean(v) (cgvae.py:297-298). */
int cgv_segment_reduce2(const float* src_a, int channels_a, float* out_a, const float* src_b, int channels_b, float* out_b,
                        const int32_t* rowptr, const int32_t* perm, int n_seg, int mean, void* stream);
/* backward of the above: gsrc[perm?perm[p]:p, :] = gout[seg(p), :] (* 1/max(len,1) if mean) */
int cgv_segment_broadcast(const float* gout, const int32_t* rowptr, const int32_t* perm, int n_seg, int channels,
                          int mean, float* gsrc /*[n_rows,C]*/, void* stream);
/* The two heads of a (mu, sigma) pair -- nn.Sequential(Linear, act, Linear) each, cgvae.py:366-371 / run_ala.py:184-189,
 * applied to the same features (cgvae.py:500-503, 226-229) -- as PAIRS of launches: layer j of both heads in one launch
 * (two Dense layers of one shape; x0 may equal x1), forward and backward-input, instead of one launch (forward) or two
 * (backward: row-split product + reduction) per layer.  Rows <= 16 (forward) / 64 (backward).  With gx1 == NULL the
 * backward returns the SUM of both products in gx0 (both layers read the same input: autograd's accumulation add is
 * part of the reduction launch).  ws: 2 x cgv_skinny_bwd_input_workspace_bytes(M, N, K). */
int cgv_pair_linear_fwd(const float* x0, const float* x1, const float* W0, const float* W1, const float* bias0,
                        const float* bias1, float* y0, float* y1, float* z0, float* z1, int act0, int act1, int n_rows, int N,
                        int K, void* stream);
int cgv_pair_linear_bwd_input(const float* gy0, const float* gy1, const float* z0, const float* z1, const float* W0,
                              const float* W1, int act0, int act1, float* gx0, float* gx1, int M, int N, int K, void* ws,
                              size_t ws_bytes, void* stream);
/* The same for up to cgv_multi_linear_max() = 4 layers of one shape per launch -- the prior's (mu, sigma) heads
 * (cgvae.py:398-401) and the encoder's (cgvae.py:500-503) are independent of each other too: layer j of all four heads in
 * one launch.  Tables are HOST arrays of n device pointers (read at call time).  bwd_input: ``group`` = 1 gives n outputs
 * gx[j]; ``group`` = 2 sums the products of problems 2o, 2o + 1 (two layers that read one input) into gx[o].
 * ws: n x max(cgv_skinny_bwd_input_workspace_bytes(M, N, K), 4 M K) bytes. */
int cgv_multi_linear_max(void);
int cgv_multi_linear_fwd(int n, const float* const* x, const float* const* W, const float* const* bias, float* const* y,
                         float* const* z, const int* act, int n_rows, int N, int K, void* stream);
int cgv_multi_linear_bwd_input(int n, int group, const float* const* gy, const float* const* z, const float* const* W,
                               const int* act, float* const* gx, int M, int N, int K, void* ws, size_t ws_bytes, void* stream);
/* CGequiVAE.reparametrize (cgvae.py:445-449: eps = randn_like(sigma); z = mu + eps * sigma) with the noise drawn in the
 * launch: z = mu + sigma * eps, eps ~ N(0, 1) from Philox4x32-10 + Box-Muller, stored (the backward pass needs it:
 * dz/dsigma = eps).  rng: 3 x uint64 in device memory {seed, draw number, 0}; the launch advances the draw number, so a
 * replayed hipGraph draws fresh noise each step.  Same distribution as torch.randn_like, different numbers: runs that
 * must reproduce a given eps pass it in and do not call this. */
int cgv_reparam_sample(const float* mu, const float* sigma, float* eps, float* z, int64_t n, uint64_t* rng, void* stream);
/* Its backward with the KL term's gradients (cgv_elbo_fwd's g_mu / g_sigma: the other consumer of mu and sigma,
 * scripts/utils.py:121) folded in: g_mu = g + k_mu, g_sigma = g * eps + k_sigma; n % 4 == 0. */
int cgv_reparam_bwd(const float* g, const float* eps, const float* k_mu, const float* k_sigma, float* g_mu, float* g_sigma,
                    int64_t n, void* stream);
/* nn.Embedding lookup (cgvae.py:268, 381) with the ids read from a float column (nxyz[:, 0], element stride id_stride):
 * out[i, :] = weight[(int) ids[i * id_stride], :]; ids are clamped to [0, n_types). */
int cgv_embedding_rows(const float* weight /*[n_types,C]*/, const float* ids_f32, int id_stride, int n_rows, int n_types,
                       int channels, float* out /*[n_rows,C]*/, void* stream);
/* The two embedding lookups of a step (encoder: atom types, cgvae.py:268; prior: bead types, cgvae.py:381) in one launch,
 * and their weight gradients (two segment sums over the type-id plans) in one launch. */
int cgv_embedding_rows2(const float* weight_a, const float* ids_a, int id_stride_a, int n_rows_a, int n_types_a, float* out_a,
                        const float* weight_b, const float* ids_b, int id_stride_b, int n_rows_b, int n_types_b, float* out_b,
                        int channels, void* stream);
int cgv_segment_reduce_pair(const float* src_a, const int32_t* rowptr_a, const int32_t* perm_a, int n_seg_a, float* out_a,
                            const float* src_b, const int32_t* rowptr_b, const int32_t* perm_b, int n_seg_b, float* out_b,
                            int channels, void* stream);

/* ---------------------------------------------------------------------------------------
 * K2 / K4  fused EquiMessageBlock (conv.py:505-563 incl. InvariantMessage 63-75 and
 * DistanceEmbed modules.py:192-197) and, with the atom->bead plan, ContractiveMessageBlock
 * (conv.py:703-733).  phi = inv_dense(s) is computed by the caller (node-level GEMMs).
 *   m_k(e,f) = phi[src(e), kF+f] * w[kF+f](e)
 *   ds[i,f]   = sum_{e: dst(e)=i} m_1
 *   dv[i,f,:] = sum_e ( m_2 * unit_e + m_0 * v[src(e), f, :] )
 * No [E, .] tensor is ever written.  with_dv = 0 skips the vector channel (explicit option;
 * the encoder never consumes it -- SURVEY 8a note a12) and leaves dv untouched.
 * s_res / v_res (optional): the outputs become s_res + ds and v_res + dv, i.e. the residual adds
 * of cgvae.py:287-288, 309-310, 391-392 fused into the store.
 * n_rows_hint = number of rows of the gathered arrays (Ns forward, Nd backward; 0 = unknown): enables
 * the buffer-descriptor gather path when every row lies within 2 GiB of its base.
 * n_edges_hint (the edge count, or 0) only selects the launch shape: several waves share a
 * receiver when the average degree is high.  Results do not depend on it beyond fp32
 * summation order.
 * ------------------------------------------------------------------------------------- */
int cgv_equi_msg_fwd(const float* phi /*[Ns,3F]*/, const float* v /*[Ns,F,3]*/, const float* geom_d,
                     const int32_t* rowptr_d, const int32_t* src_d, const float* Wd /*[3F,R]*/,
                     const float* bd /*[3F]*/, float* ds /*[Nd,F]*/, float* dv /*[Nd,F,3]*/, int n_dst,
                     int n_feat, int n_rbf, int with_dv, int64_t n_edges_hint, int64_t n_rows_hint,
                     const float* s_res /*[Nd,F] or NULL*/, const float* v_res /*[Nd,F,3] or NULL*/, void* stream);
/* The same forward (with the vector channel) as a shared-source walk over the receiver-group order
 * (cgv_group_plan_build with the same rb; geom_g = cgv_edge_geometry_grouped records of that order): a wave keeps rb
 * accumulator sets and gathers every source row once per group instead of once per edge -- for graphs whose consecutive receivers share
 * most of their neighbours (molecules: always).  Needs even n_feat / n_rbf, rb in {2, 4}, 8-byte aligned operands
 * (Wd, geom_g 16-byte), all n_rows rows of phi / v within 2 GiB.  Same results up to fp32 summation order. */
int cgv_equi_msg_grouped_supported(int n_feat, int n_rbf, int rb);
int cgv_equi_msg_fwd_grouped(const float* phi /*[Ns,3F]*/, const float* v /*[Ns,F,3]*/, const float* geom_g,
                             const int32_t* rowptr_d, const int32_t* src_g,
                             const float* Wd /*[3F,R]*/, const float* bd /*[3F]*/, float* ds /*[Nd,F]*/,
                             float* dv /*[Nd,F,3]*/, int n_dst, int n_feat, int n_rbf, int rb, int64_t n_rows,
                             int64_t n_edges /* records in geom_g */, const float* s_res /*[Nd,F] or NULL*/, const float* v_res /*[Nd,F,3] or NULL*/,
                             void* stream);
/* cgv_equi_msg_fwd_grouped with `parts` (1..4) blocks per (group, channel tile): the group's edge range is cut into
 * parts x 4 wave slices instead of 4 -- shorter blocks in larger number, which the dispatcher spreads evenly over the CUs
 * where one block per item gives 3.25 blocks per CU (chignolin graph: a quarter of the CUs carry a fourth block) or
 * blocks that live 20-145 us depending on their group's degree (2000-atom graph: the launch ends with a few long ones).
 * The parts' sums meet in the workspace: every block leaves its own in its slot, the block that draws the group's last
 * ticket adds the slots in part order and stores -- results do not depend on timing.  rb = 2, default kernel only
 * (otherwise parts is taken as 1).  workspace: cgv_equi_msg_grouped_workspace_bytes bytes, 16-byte aligned, ZERO-FILLED
 * once by the caller (self-resetting tickets at its head); one workspace serves the launches of one stream. */
size_t cgv_equi_msg_grouped_workspace_bytes(int n_dst, int n_feat, int rb, int parts);
int cgv_equi_msg_fwd_grouped_parts(const float* phi /*[Ns,3F]*/, const float* v /*[Ns,F,3]*/, const float* geom_g,
                                   const int32_t* rowptr_d, const int32_t* src_g,
                                   const float* Wd /*[3F,R]*/, const float* bd /*[3F]*/, float* ds /*[Nd,F]*/,
                                   float* dv /*[Nd,F,3]*/, int n_dst, int n_feat, int n_rbf, int rb, int64_t n_rows,
                                   int64_t n_edges, const float* s_res /*[Nd,F] or NULL*/,
                                   const float* v_res /*[Nd,F,3] or NULL*/, int parts, void* workspace,
                                   size_t workspace_bytes, void* stream);
/* The same shared-source walk with the work cut into EQUAL EDGE RANGES instead of (group, channel tile) blocks:
 * every wave walks the same number of edges of its channel tile's group-ordered edge array, whatever the groups'
 * sizes, on a grid that is resident at once (cgv_equi_msg_fwd_grouped: 3.25 blocks per CU on the chignolin graph, block lifetimes 20-145 us on the 2000-atom
 * graph).  A group cut by a range boundary is finished by the last contributing wave to arrive, which adds the
 * contributors' partial sums (workspace) in range order: results do not depend on timing.  rb = 2 only; dst_g = the
 * plan's receiver ids in group order.  The edge count is read from rowptr_d[n_dst] on the device.
 * workspace: cgv_equi_msg_balanced_workspace_bytes bytes, 16-byte aligned, ZERO-FILLED once by the caller (its head holds
 * self-resetting tickets); one workspace serves every launch of one stream, launches on concurrent streams need their own. */
int cgv_equi_msg_balanced_supported(int n_feat, int n_rbf, int rb);   /* even n_feat / n_rbf, rb = 2 */
size_t cgv_equi_msg_balanced_workspace_bytes(int n_dst, int n_feat, int rb);
int cgv_equi_msg_fwd_balanced(const float* phi /*[Ns,3F]*/, const float* v /*[Ns,F,3]*/, const float* geom_g,
                              const int32_t* rowptr_d, const int32_t* src_g, const int32_t* dst_g,
                              const float* Wd /*[3F,R]*/, const float* bd /*[3F]*/, float* ds /*[Nd,F]*/,
                              float* dv /*[Nd,F,3]*/, int n_dst, int n_feat, int n_rbf, int rb, int64_t n_rows,
                              const float* s_res /*[Nd,F] or NULL*/, const float* v_res /*[Nd,F,3] or NULL*/,
                              void* workspace, size_t workspace_bytes, void* stream);
/* Backward.  gs / gv are the upstream gradients at the receivers (gv == NULL when dv is not
 * consumed).  Traverses the src-sorted view; writes g_phi [Ns,3F], g_v [Ns,F,3] (only if gv),
 * gWd [3F,R], gbd [3F] completely (zeros where nothing flows).  Deterministic two-stage
 * reduction for gWd/gbd through `workspace`. */
size_t cgv_equi_msg_bwd_workspace_bytes(int n_src, int n_feat, int n_rbf);
int cgv_equi_msg_bwd(const float* phi, const float* v, const float* geom_s, const int32_t* rowptr_s,
                     const int32_t* dst_s, const float* Wd, const float* bd, const float* gs /*[Nd,F] or NULL*/,
                     const float* gv /*[Nd,F,3] or NULL*/, float* g_phi, float* g_v, float* gWd, float* gbd,
                     int n_src, int n_feat, int n_rbf, int64_t n_edges_hint, int64_t n_rows_hint, void* workspace,
                     size_t workspace_bytes, void* stream);
/* The same without its second stage, for a training step that wants the filter gradients only at its end: g_phi / g_v
 * are final, gWd / gbd stay as per-chunk partial sums in `workspace` (keep it alive) and are finished -- for up to
 * cgv_filter_reduce_jobs_max() such launches at once -- by cgv_filter_reduce_jobs.  jobs_host: records
 * {const float* part; float* gWd; float* gbd; int n_chunks, K, R, F;} with part = the launch's workspace, n_chunks /
 * K = what it returned in *n_chunks / *k_live, R = n_rbf, F = n_feat.  (The filter gradients feed the optimiser only:
 * their seven reduction launches per chignolin step were seven links in the backward chain.) */
int cgv_equi_msg_bwd_deferred(const float* phi, const float* v, const float* geom_s, const int32_t* rowptr_s,
                              const int32_t* dst_s, const float* Wd, const float* bd, const float* gs, const float* gv,
                              float* g_phi, float* g_v, int n_src, int n_feat, int n_rbf, int64_t n_edges_hint,
                              int64_t n_rows_hint, void* workspace, size_t workspace_bytes, int* n_chunks, int* k_live,
                              void* stream);
int cgv_filter_reduce_jobs_max(void);
int cgv_filter_reduce_job_bytes(void);
int cgv_filter_reduce_jobs(const void* jobs_host, int n_jobs, void* stream);

/* ---------------------------------------------------------------------------------------
 * Per-batch graph work over job tables (csrc/batch_plans.hip): what cgv_csr_build and cgv_edge_geometry[_grouped] do,
 * for ALL sorted views / record arrays of a batch in 4 + 1 launches (replaces make_directed's consumers and the
 * per-block preprocess_r / rbf / envelope of conv.py:10-29, modules.py:148-197 for a whole batch at once).
 * jobs_host [host]: n records with the layout of cgv::PlanJob / cgv::GeomJob (csrc/batch_plans.hip; sizes from
 * cgv_plan_job_bytes / cgv_geom_job_bytes), device pointers inside, copied into the kernel arguments by value.
 * PlanJob.count must be ZERO on entry ([n_rows + 1] ints) and is zero again on exit.
 * ------------------------------------------------------------------------------------- */
/* Rows [n][4] = (type id, x, y, z) of a prepared batch's atoms and beads into the captured step's tensors and, as
 * contiguous [n][3] coordinates, into the graph bundle: one launch (replaces the reference DataLoader's hand-over of the
 * next batch, cgvae.py:486 / scripts/utils.py:131). */
int cgv_batch_load_rows(const float* src_atoms, float* dst_atoms, float* xyz_atoms, int n_atoms, const float* src_beads,
                        float* dst_beads, float* xyz_beads, int n_beads, void* stream);
int cgv_plan_jobs_max(void);
int cgv_plan_job_bytes(void);
int cgv_plan_jobs_build(const void* jobs_host, int n_jobs, void* stream);
int cgv_geom_jobs_max(void);
int cgv_geom_job_bytes(void);
int cgv_geom_jobs_build(const void* jobs_host, int n_jobs, void* stream);

/* ---------------------------------------------------------------------------------------
 * K3  fused EquiMessagePsuedo (conv.py:180-242), i = dst (receiver), j = src, q_k = phi[j,kF+f] w_k
 * with k = 0..8 (phi is [N,9F]), u = unit_e:
 *   dh    = sum q0 s_i                 dhbar = sum (v_i . vbar_j)      (no filter, conv.py:206)
 *   dv    = sum q1 u + q2 v_j + q3 (v_i x vbar_j) + q4 sbar_i vbar_j
 *   dvbar = sum q5 vbar_j + q6 sbar_i v_j + q7 (v_i x v_j) + q8 (vbar_i x vbar_j)
 * Backward needs both CSR views (receiver-side sums on the dst-sorted one, source-side sums,
 * g_phi and the filter gradients on the src-sorted one); any upstream gradient may be NULL.
 * All outputs are written completely.  Derivation: csrc/pseudo_msg.hip.
 * residual != 0: the outputs are the UPDATED states s + dh, sbar + dhbar, v + dv, vbar + dvbar
 * (the residual adds of cgvae.py:108-111 fused in), and the backward adds the pass-through term.
 * ------------------------------------------------------------------------------------- */
int cgv_pseudo_msg_fwd(const float* phi /*[N,9F]*/, const float* s, const float* sbar, const float* v,
                       const float* vbar, const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d,
                       const float* Wd /*[9F,R]*/, const float* bd /*[9F]*/, float* dh, float* dhbar, float* dv,
                       float* dvbar, int n_nodes, int n_feat, int n_rbf, int residual, void* stream);
/* The same launch; dv_rows (or NULL) [3 N, F] additionally receives dv as rows 3 i + xyz, the operand layout of
 * UpdateBlock's u_mat / v_mat products (conv.py:591) -- saves the transpose launch in front of every decoder update. */
int cgv_pseudo_msg_fwd_rows(const float* phi, const float* s, const float* sbar, const float* v, const float* vbar,
                            const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* Wd,
                            const float* bd, float* dh, float* dhbar, float* dv, float* dvbar, float* dv_rows,
                            int n_nodes, int n_feat, int n_rbf, int residual, int64_t n_edges_hint /*0: unknown; dispatch only*/,
                            void* stream);
size_t cgv_pseudo_msg_bwd_workspace_bytes(int n_nodes, int n_feat, int n_rbf);
int cgv_pseudo_msg_bwd(const float* phi, const float* s, const float* sbar, const float* v, const float* vbar,
                       const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d,
                       const float* geom_s, const int32_t* rowptr_s, const int32_t* dst_s,
                       const float* Wd, const float* bd,
                       const float* gh, const float* ghbar, const float* gv, const float* gvbar,
                       float* g_phi /*[N,9F]*/, float* g_s, float* g_sbar, float* g_v, float* g_vbar,
                       float* gWd /*[9F,R]*/, float* gbd /*[9F]*/, int n_nodes, int n_feat, int n_rbf,
                       int residual, int64_t n_edges_hint /*0: unknown; dispatch only*/, void* workspace,
                       size_t workspace_bytes, void* stream);
/* cgv_pseudo_msg_bwd without its last launch (the filter-gradient reduction over chunks): the partial sums stay in
 * `workspace`; one cgv_filter_reduce_jobs launch (job K = 9, *n_chunks) finishes them together with the step's other
 * message blocks.  Replaces the same autograd as cgv_pseudo_msg_bwd (conv.py:180-242). */
int cgv_pseudo_msg_bwd_deferred(const float* phi, const float* s, const float* sbar, const float* v, const float* vbar,
                                const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* geom_s,
                                const int32_t* rowptr_s, const int32_t* dst_s, const float* Wd, const float* bd, const float* gh,
                                const float* ghbar, const float* gv, const float* gvbar, float* g_phi, float* g_s, float* g_sbar,
                                float* g_v, float* g_vbar, int n_nodes, int n_feat, int n_rbf, int residual,
                                int64_t n_edges_hint, void* workspace, size_t workspace_bytes, int* n_chunks /*[host]*/, void* stream);

/* ---------------------------------------------------------------------------------------
 * Decoder layer as channel-group kernels (csrc/decoder_layer.hip) -- replaces, for bead graphs of at most 16 nodes, the
 * per-layer launch chain of the pseudo-vector decoder loop cgvae.py:100-123: EquiMessagePsuedo (conv.py:180-242) with
 * its inv_dense.1 product (conv.py:63-75), UpdateBlock (conv.py:588-616) with the u_mat / v_mat and s_dense.1 products,
 * the residual adds (cgvae.py:108-111, 122-123) and their autograd backward.  A block owns 4 channels f0..f0+3 and the
 * weight rows {g F + f} that feed them; grid = F / 4 blocks of 576 threads.
 * Slices (outputs of the backward phases; inputs of the next one): slice s = block s's row-split partial product,
 * QUAD-MAJOR [K/4][rows][4] floats (rows = n, or 3 n behind uv_bwd), cgv_decoder_slice_floats(K, rows) each;
 * n_slices = F/4 (N/4 for dense_bwd).  The message kernels stage the bead graph in LDS: their `n_edges` argument is the
 * number of edge RECORDS TO STAGE (1..cgv_decoder_max_edges(); the record / index arrays must hold that many) -- pass the
 * arrays' capacity, not a batch's edge count: the edge structure itself is read from rowptr on the device, so the same
 * launch (a captured graph node) serves batches with other edge counts; edges beyond the staged records are ignored.
 *   forward   msg_fwd   phi = a1 W2^T + b2 -> message -> stack[:, :F] = S', Sbar', V', Vbar', V' as rows [3n, F]
 *             uv_fwd    UV [3n, 2F] = rows [Wu; Wv]^T ; stack[:, F:] = sqrt(sum_xyz (Vv^2 + 1e-10))
 *             gate_fwd  a [n, 3F] = a0 W1'^T + b1' ; S'' = S' + (U.Vv) a_sv + a_ss ; V'' = V' + U a_vv
 *   backward  gate_bwd  gS = gs_base + sum gs_slices (written to gs_sum) ; ga, gUV (gU | gVv) ; slices = ga W1'
 *             dense_bwd g = sum g_slices (written to g_dense) ; slices = (g act'(z)) W      (W [N, K], grid N/4)
 *             uv_bwd    g_stack = sum slices ; g_s2 = g_stack[:, :F] + gs_res ; gUV_out = [gUV[:, :F] | gUV[:, F:] + g_stack[:, F:] Vv / norm]
 *                       (gUV_out may be gUV itself only with CGV_OPT_DECODER_COLSPLIT = 0) ; slices [.., 48 rows ..] = gUV_out [Wu; Wv]
 *             msg_bwd   gV' = sum gvrows_slices (rows 3i+xyz) + gv_res ; full EquiMessagePsuedo backward (g_s, g_sbar,
 *                       g_v, g_vbar, g_phi, gWd, gbd written completely) ; slices = g_phi W2
 *             slices_to_dense  out [n, F] = base + sum slices (16-row slices)
 * ------------------------------------------------------------------------------------- */
int cgv_decoder_layer_supported(int n_nodes, int n_feat, int n_rbf);
int64_t cgv_decoder_slice_floats(int K, int rows);
int cgv_decoder_max_edges(void);
int cgv_decoder_block_channels(int width);   /* 4 or 8: gate_bwd / dense_bwd / uv_bwd emit width / this slices */
int cgv_decoder_debug_clock(uint64_t* buf /*device, 272 slots, or NULL*/);   /* measurement only */
int cgv_decoder_msg_fwd(const float* a1, const float* W2, const float* b2, const float* s, const float* sbar, const float* v,
                        const float* vbar, const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* Wd,
                        const float* bd, float* phi, float* stack, float* sbar_out, float* v_out, float* vbar_out,
                        float* rows_out, int n_nodes, int n_feat, int n_rbf, int n_edges, void* stream);
/* An EquiMessageBlock layer (conv.py:505-563 with the residual adds of CGprior.forward, cgvae.py:391-392: h += ds, v += dv)
 * on a SMALL graph (<= 16 nodes, <= cgv_decoder_max_edges() edges: the prior's bead graph) in the channel-group scheme of the
 * decoder layer: forward = cgv_decoder_dense_fwd (a1 = swish(h W1^T + b1)) + cgv_prior_msg_fwd (phi = a1 W2^T + b2 for the
 * block's 3 x 4 rows, then the message on its channels: s_out = s + ds, v_out = v + dv; with_dv = 0 leaves v_out = v);
 * backward = cgv_prior_msg_bwd + cgv_decoder_dense_bwd.  The backward is the SCALAR path only: it is for callers that never
 * use v_out's gradient (the prior and the encoder discard the vector channel, cgvae.py:393-396), so g_q0 = g_q2 = 0 and
 * the dead filter slices get explicit zero gradients.  The upstream gradient d loss / d s_out arrives as base (dense, or
 * NULL) + quad-major slices (or NULL) like every slice consumer here; g_h = their sum (dense: the residual path), g_phi
 * [n, 3F] dense for the weight-gradient launch, slices_out: n_feat / 4 slices of cgv_decoder_slice_floats(n_feat, n). */
int cgv_prior_msg_fwd(const float* a1, const float* W2, const float* b2, const float* s, const float* v, const float* geom_d,
                      const int32_t* rowptr_d, const int32_t* src_d, const float* Wd, const float* bd, float* phi, float* s_out,
                      float* v_out, int n_nodes, int n_feat, int n_rbf, int n_edges, int with_dv, void* stream);
int cgv_prior_msg_bwd(const float* phi, const float* geom_s, const int32_t* rowptr_s, const int32_t* dst_s, const float* Wd,
                      const float* bd, const float* gh_base, const float* gh_slices, int gh_n_slices, int64_t gh_slice_floats,
                      const float* W2, float* g_phi, float* g_h, float* gWd, float* gbd, float* slices_out,
                      int64_t out_slice_floats, int n_nodes, int n_feat, int n_rbf, int n_edges, void* stream);
/* y = act(x W^T + b) (z = pre-activation or NULL) for <= 16 rows, N / 4 blocks: the two full-width products of a layer */
int cgv_decoder_dense_fwd(const float* x, const float* W /*[N,K]*/, const float* bias, float* y, float* z, int n_nodes, int N,
                          int K, int act, void* stream);
int cgv_decoder_uv_fwd(const float* rows, const float* Wuv, float* UV, float* stack, int n_nodes, int n_feat, void* stream);
int cgv_decoder_gate_fwd(const float* a0, const float* W1p, const float* b1p, const float* UV, const float* stack,
                         const float* v2, float* a, float* s3, float* v3, int n_nodes, int n_feat, void* stream);
/* UpdateBlock forward (conv.py:593-616) with the element-wise halves in the products' epilogues, for 1..96 bead rows -- the
 * per-block path beyond the channel-group decoder's 16 nodes (cgv_update_rows_fused_supported):
 *   uv_norm   UV [3n, 2F] = rows [u_mat; v_mat]^T ; stack [n, 2F] = [s | sqrt(sum_xyz (Vv^2 + 1e-10))]   (s NULL: first half untouched)
 *   gate      a [n, 3F] = a0 W1^T + b1 ; s_out = (U.Vv) a_sv + a_ss (+ s_res) ; v_out = U a_vv (+ v_res)  (residual adds of
 *             cgvae.py:122-123 when s_res / v_res are given)
 * replacing cgv_skinny / tile_linear_fwd + cgv_update_norm_stack_fwd and + cgv_update_gate_fwd (two launches each). */
int cgv_update_rows_fused_supported(int n_rows, int n_feat);
int cgv_update_uv_norm_fwd_fused(const float* rows, const float* Wuv /*[2F, F]*/, const float* s /*or NULL*/, float* UV, float* stack,
                                 int n_nodes, int n_feat, void* stream);
int cgv_update_gate_fwd_fused(const float* a0, const float* W1 /*[3F, F]*/, const float* b1, const float* UV, const float* s_res /*or NULL*/,
                              const float* v_res /*or NULL*/, float* a, float* s_out, float* v_out, int n_rows, int n_feat, void* stream);
int cgv_decoder_gate_bwd(const float* UV, const float* a, const float* gs_base /*or NULL*/, const float* gs_slices /*or NULL*/,
                         int gs_n_slices, int64_t gs_slice_stride, const float* gv /*or NULL*/, const float* W1p, float* ga,
                         float* gUV, float* gs_sum, float* slices_out, int64_t out_slice_stride, int n_nodes, int n_feat,
                         void* stream);
int cgv_decoder_dense_bwd(const float* g_slices, int g_n_slices, int64_t g_slice_stride, const float* z /*or NULL*/, int act,
                          const float* W, float* g_dense, float* slices_out, int64_t out_slice_stride, int n_nodes, int N, int K,
                          void* stream);
int cgv_decoder_uv_bwd(const float* gstack_slices, int n_slices, int64_t slice_stride, const float* UV, const float* stack,
                       const float* gs_res, const float* Wuv, const float* gUV, float* gUV_out, float* g_s2, float* slices_out,
                       int64_t out_slice_stride, int n_nodes, int n_feat, void* stream);
int cgv_decoder_msg_bwd(const float* phi, const float* s, const float* sbar, const float* v, const float* vbar,
                        const float* geom_d, const int32_t* rowptr_d, const int32_t* src_d, const float* geom_s,
                        const int32_t* rowptr_s, const int32_t* dst_s, const float* Wd, const float* bd, const float* gh,
                        const float* ghb /*or NULL*/, const float* gvrows_slices, int n_slices, int64_t slice_stride,
                        const float* gv_res /*or NULL*/, const float* gvb /*or NULL*/, const float* W2, float* g_phi, float* g_s,
                        float* g_sbar, float* g_v, float* g_vbar, float* gWd, float* gbd, float* slices_out,
                        int64_t out_slice_stride, int n_nodes, int n_feat, int n_rbf, int n_edges, void* stream);
int cgv_decoder_slices_to_dense(const float* base /*or NULL*/, const float* slices, int n_slices, int64_t slice_stride,
                                float* out, int n_nodes, int n_feat, void* stream);

/* ---------------------------------------------------------------------------------------
 * K5  UpdateBlock element-wise core (conv.py:588-616); the K=F products are separate launches
 * (skinny GEMMs below / hipBLASLt).  U, Vv = u_mat / v_mat applied to v as rows r = node*3 + xyz with row
 * stride `ld` floats (ld = F for separate buffers, 2F when both are column halves of one product with
 * the concatenated weights [u_mat; v_mat]); a = s_dense(stack) viewed [N,3,F] = (a_vv, a_sv, a_ss).
 *   rows_from_vec : rows[n,k,f] = v[n,f,k]                                          conv.py:591
 *   vec_from_rows : vec[n,f,k] = rows[n,k,f] (+ res[n,f,k])                         (its backward, + residual pass-through)
 *   norm_stack    : stack[n] = [ s[n,:] | sqrt(sum_k (Vv[n,k,:]^2 + 1e-10)) ]       conv.py:600-601
 *   gate          : dv[n,f,k] = U[n,k,f] a_vv ; ds[n,f] = (sum_k U Vv) a_sv + a_ss  conv.py:607-614
 *                   with s_res / v_res: s + ds and v + dv (cgvae.py:122-123) instead of the deltas
 * and the backward kernels (g_ds / g_dv / g_res may be NULL; norm_stack_bwd can accumulate onto gVv).
 * ------------------------------------------------------------------------------------- */
int cgv_update_rows_from_vec(const float* v /*[N,F,3]*/, float* rows /*[N,3,F]*/, int n_nodes, int n_feat, void* stream);
int cgv_update_vec_from_rows(const float* rows, const float* res /*[N,F,3] or NULL*/, float* vec, int n_nodes, int n_feat,
                             void* stream);
int cgv_update_norm_stack_fwd(const float* s /*[N,F]*/, const float* Vv, float* stack /*[N,2F]*/, int n_nodes, int n_feat,
                              int ld, void* stream);
int cgv_update_norm_stack_bwd(const float* gstack, const float* Vv, const float* stack, const float* g_res /*[N,F] or NULL*/,
                              float* g_s, float* gVv, int n_nodes, int n_feat, int ld, int accumulate, void* stream);
int cgv_update_gate_fwd(const float* U, const float* Vv, const float* a, const float* s_res /*[N,F] or NULL*/,
                        const float* v_res /*[N,F,3] or NULL*/, float* ds /*[N,F]*/, float* dv /*[N,F,3]*/,
                        int n_nodes, int n_feat, int ld, void* stream);
int cgv_update_gate_bwd(const float* U, const float* Vv, const float* a, const float* g_ds, const float* g_dv,
                        float* gU, float* gVv, float* ga, int n_nodes, int n_feat, int ld, void* stream);
/* The same kernels with a slice-sum operand (see cgv_skinny_linear_bwd_input_slices): rows / gstack / g_ds arrive as
 * row-slice partials of the backward-input product that produced them (g_ds: base + slices). */
int cgv_update_vec_from_rows_slices(const float* rows_slices, int n_slices, int64_t slice_stride, const float* res, float* vec,
                                    int n_nodes, int n_feat, void* stream);
int cgv_update_norm_stack_bwd_slices(const float* gstack_slices, int n_slices, int64_t slice_stride, const float* Vv,
                                     const float* stack, const float* g_res_base /*or NULL*/, const float* g_res_slices /*or NULL*/,
                                     int n_res_slices, int64_t res_slice_stride, float* g_s, float* gVv, int n_nodes, int n_feat,
                                     int ld, int accumulate, void* stream);
int cgv_update_gate_bwd_slices(const float* U, const float* Vv, const float* a, const float* g_ds_base /*or NULL*/,
                               const float* g_ds_slices /*or NULL*/, int n_slices, int64_t slice_stride, const float* g_dv,
                               float* gU, float* gVv, float* ga, int n_nodes, int n_feat, int ld, void* stream);

/* ---------------------------------------------------------------------------------------
 * Skinny fp32 GEMMs for the node-level Dense / nn.Linear layers on the bead graph
 * (modules.py:103-114, Swish modules.py:16-21, and their autograd backward): M <=
 * cgv_skinny_max_rows() rows, weight W[N,K] row-major as torch stores it, N % 4 == 0,
 * K % 4 == 0, 16-byte aligned operands.  act: 0 = identity, 1 = Swish, 2 = tanh, 3 = ReLU (nn.Tanh / nn.ReLU of the mu / sigma heads),
 * 4 = 1e-12 + exp(z/2) (sigma head, cgvae.py:503), 5 = 1e-9 + exp(z/2) (prior std, cgvae.py:401).
 *   fwd        z = x W^T + bias ; y = act(z)      (bias may be NULL; z [M,N] is written when act != 0)
 *   bwd_input  gx[M,K] = (gy * act'(z)) W         (deterministic).  With a workspace of
 *              cgv_skinny_bwd_input_workspace_bytes bytes (contents ignored) the weight rows are
 *              split over ~320 blocks whose partial sums a second small launch adds in a fixed
 *              order; with ws = NULL one block per 64 output columns walks all rows (one launch,
 *              slow, same result up to summation order).
 *   grouped weight gradients: the caller queues one 88-byte record per layer
 *       { gy, x, z, gW, gb (pointers), M, N, K, accumulate, act, block_begin, tiles_k, tile_w, seg_rows, seg_stride, pad }
 *     (tiles_k / tile_w / the block count come from cgv_wgrad_plan; block_begin is the running sum
 *     of the block counts) and ONE cgv_grouped_wgrad launch computes, for every record,
 *       gW[N,K] (+)= (gy * act'(z))^T x ,  gb[N] (+)= sum_m (gy * act'(z))[m,:]     (gb may be NULL)
 *     max_lds_floats = max over records of cgv_wgrad_lds_floats(M, tile_w).
 * v_mfma_f32_16x16x4_f32 is an exact fp32 FMA chain, so these match an fmaf loop bit for bit.
 * ------------------------------------------------------------------------------------- */
int cgv_skinny_max_rows(void);
int cgv_skinny_supported(int M, int N, int K);
int cgv_skinny_fwd_supported(int M, int N, int K);   /* cgv_skinny_linear_fwd alone: any row count (row blocks over blockIdx.y) */
int cgv_skinny_linear_fwd(const float* x, const float* W, const float* bias, float* y, float* z /*or NULL*/, int M, int N,
                          int K, int act, void* stream);
int cgv_skinny_bwd_input_supported(int M, int N, int K);      /* bwd_input alone takes up to 128 rows */
size_t cgv_skinny_bwd_input_workspace_bytes(int M, int N, int K);
int cgv_skinny_linear_bwd_input(const float* gy, const float* z /*or NULL*/, const float* W, float* gx, int M, int N, int K,
                                int act, void* ws /*or NULL*/, size_t ws_bytes, void* stream);
/* the same with a second gradient of the input added in the reduction launch of the row-split product; returns
 * CGV_E_UNSUPPORTED when the product has one row slice only (no reduction launch): add separately then. */
int cgv_skinny_linear_bwd_input_add(const float* gy, const float* z /*or NULL*/, const float* W, const float* add, float* gx,
                                    int M, int N, int K, int act, void* workspace, size_t workspace_bytes, void* stream);
/* ... and with the result multiplied by act_out'(z_out), z_out [M, K] the pre-activation of the layer that produced this
 * layer's input (add may be NULL): see cgv_tile_linear_bwd_input_out.  CGV_E_UNSUPPORTED for a single row slice. */
int cgv_skinny_linear_bwd_input_out(const float* gy, const float* z /*or NULL*/, const float* W, const float* add /*or NULL*/,
                                    float* gx, int M, int N, int K, int act, const float* z_out, int act_out, void* workspace,
                                    size_t workspace_bytes, void* stream);
/* Slice sums ("base + row-slice partials"): the split backward-input product leaves one partial [M, K] matrix per row
 * slice of the weight; instead of a reduction launch per product (57 a step on the chignolin config) the NEXT kernel on
 * the autograd chain adds the slices while it loads its operand -- in slice order, so results are deterministic.
 * Replaces the same autograd steps as cgv_skinny_linear_bwd_input (Dense backward, modules.py:103-114) plus the
 * gradient-accumulation adds autograd inserts where a state feeds two consumers (cgvae.py:100-123).
 *   cgv_skinny_bwd_input_plan          number of slices / floats per slice of the product (M <= 64)
 *   cgv_skinny_linear_bwd_input_slices part[s] = (g * act'(z))[:, rows of slice s] W[rows of slice s, :],
 *                                      g = gy_base + sum_s gy_slices[s]; g_dense (or NULL) receives g itself, [M, N]
 *                                      (the weight-gradient launch needs it as a plain matrix)
 *   cgv_slice_sum                      out = base + sum_s slices[s]: the plain reduction, for consumers outside this library */
int cgv_skinny_bwd_input_plan(int M, int N, int K, int* n_slices /*[host]*/, int64_t* slice_floats /*[host]*/);
int cgv_skinny_linear_bwd_input_slices(const float* gy_base /*[M,N] or NULL*/, const float* gy_slices /*or NULL*/,
                                       int gy_n_slices, int64_t gy_slice_stride, float* g_dense /*[M,N] or NULL*/,
                                       const float* z /*or NULL*/, const float* W, float* part /*[n_slices, M, K]*/,
                                       size_t part_bytes, int M, int N, int K, int act, void* stream);
int cgv_slice_sum(const float* base /*or NULL*/, const float* slices, int n_slices, int64_t slice_stride, float* out,
                  int64_t n_floats, void* stream);
/* Dense backward prologue for any row count (the library-GEMM path of the atom-level layers):
 *   g = gy * act'(z)  (stored to g_out [M,N] when act != 0 and g_out != NULL)  and
 *   gb[n] (+)= sum_m g[m,n]  (gb may be NULL) -- one launch, fixed summation order.
 * Replaces the tensor-op chain of modules.py:16-21's autograd backward plus the bias column sum. */
int cgv_dense_grad_prepare(const float* gy, const float* z /*or NULL*/, float* g_out /*or NULL*/, float* gb /*or NULL*/,
                           int M, int N, int act, int accumulate, void* stream);
/* Tiled fp32 GEMMs for the same layers at any row count (atom-level layers, M = atoms of the batch):
 * the reduction axis of every block is split over its 4 waves, operands are loaded from L2 directly in
 * MFMA layout, bias + Swish are fused into the forward epilogue.  Need N % 4 == 0, K % 4 == 0 and
 * 16-byte aligned x / W / g / gx / gW.  g is the activation-corrected upstream gradient
 * (cgv_dense_grad_prepare).  Exact fp32 FMA chains; deterministic. */
int cgv_tile_supported(int M, int N, int K);
int cgv_tile_linear_fwd(const float* x, const float* W, const float* bias /*or NULL*/, float* y, float* z /*or NULL*/, int M,
                        int N, int K, int act, void* stream);
int cgv_tile_linear_bwd_input(const float* g, const float* W, float* gx, int M, int N, int K, void* stream);
/* the same with g = gy * act'(z) formed in the operand loads (saves the cgv_dense_grad_prepare pass when the weight /
 * bias gradients go to the grouped launch, which applies act' and sums the bias itself) */
int cgv_tile_linear_bwd_input_act(const float* gy, const float* z /*or NULL*/, const float* W, float* gx, int M, int N, int K,
                                  int act, void* stream);
/* gx = add + (gy * act'(z)) W  (add [M, K]: a second gradient of the layer's input, e.g. the residual / message-kernel path of
 * a block whose first Dense reads the same state -- conv.py:63-75 + 553-561: the accumulation autograd would do with a
 * separate add launch rides in the store epilogue). */
int cgv_tile_linear_bwd_input_act_add(const float* gy, const float* z /*or NULL*/, const float* W, const float* add, float* gx,
                                      int M, int N, int K, int act, void* stream);
/* ... and a third gradient held as ONE ROW PER SEGMENT of the rows -- the backward of scatter_mean / scatter_add of this
 * very input (cgvae.py:297): gx[m, :] += seg_grad[row2seg[m], :] (/ max(len(segment), 1) when mean) in the store epilogue,
 * instead of cgv_segment_broadcast + an accumulation add.  add may be NULL. */
int cgv_tile_linear_bwd_input_act_add_bcast(const float* gy, const float* z, const float* W, const float* add, const float* seg_grad,
                                            const int64_t* row2seg, const int32_t* seg_rowptr, int mean, float* gx, int M, int N,
                                            int K, int act, void* stream);

/* PAIR launches of the tile kernels: two Dense layers of ONE shape (M, N, K; each with its own activation code) in one grid (the extra grid dimension
 * selects the operands).  Replaces two consecutive layer launches whose inputs are ready at the same time: the first Dense
 * of ContractiveMessageBlock i and of EquiMessageBlock i + 1 read the same atom state (cgvae.py:286-305, conv.py:512-516 /
 * 709-713), their second Dense layers follow together; in backward the two second layers' input gradients.
 * cgv_tile_pair_supported: does this shape run on the register-tile kernels (the LDS-staged ones of big shapes take no pair)? */
int cgv_tile_pair_supported(int M, int N, int K);
int cgv_tile_pair_linear_fwd(const float* x_a, const float* W_a, const float* bias_a, float* y_a, float* z_a, const float* x_b,
                             const float* W_b, const float* bias_b, float* y_b, float* z_b, int M, int N, int K, int act_a,
                             int act_b, void* stream);
int cgv_tile_pair_linear_bwd_input(const float* gy_a, const float* z_a, const float* W_a, const float* add_a, float* gx_a,
                                   const float* gy_b, const float* z_b, const float* W_b, const float* add_b, float* gx_b, int M,
                                   int N, int K, int act_a, int act_b, void* stream);
/* The two entry points above whose OUTPUT is multiplied by act_out'(z_out) -- z_out [M, K] = the pre-activation of the layer
 * that produced this layer's input (modules.py:103-114: y = act(z)), so the stored gx is that layer's g = gy * act'(z)
 * and its own backward runs with act = 0.  Applied once per element in the store epilogue instead of in the operand loads
 * of the producing layer's backward-input and weight-gradient launches (every column-tile block of a row tile evaluates
 * it again there).  add may be NULL; in the pair form either z_out may be NULL (that output is stored as it is). */
/* UpdateBlock.forward's backward (conv.py:600-603) through s_dense.0 AND the norm / stack step in ONE launch: the product
 * g_stack = (gy * act'(z)) W over M beads (K = 2 F columns) is not stored; columns k < F become g_s = g_stack + g_res (g_res
 * may be NULL), columns F + f go through d ||Vv|| / d Vv: gVv[3 m + xyz, f] (+)= g_stack / stack[m, F + f] * Vv[3 m + xyz, f]
 * (rows of ld floats; accumulate != 0 adds to what cgv_update_gate_bwd left there).  Replaces cgv_tile_linear_bwd_input_act +
 * cgv_update_norm_stack_bwd. */
int cgv_tile_linear_bwd_input_norm_stack(const float* gy, const float* z, const float* W, int M, int N, int K, int act,
                                         const float* stack, const float* Vv, const float* g_res, float* g_s, float* gVv, int ld,
                                         int accumulate, void* stream);
/* What cgv_tile_linear_bwd_input* would do for this shape GIVEN a registered workspace (the workspace is per (host thread,
 * stream) state a single call cannot see): *shares = blocks per output tile of the split reduction (1 = unsplit), *streamk = 1 when
 * the stream-K kernel takes the launch (np = 1: single, 2: pair launch).  For callers that choose between these entry points and
 * the row-split kernel of cgv_skinny_linear_bwd_input. */
int cgv_tile_bwd_input_plan(int M, int N, int K, int np, int* shares /*[host]*/, int* streamk /*[host]*/);
/* Few output tiles and a long reduction (96 bead rows x 1800 columns: 60 tiles on 256 CUs): with a workspace registered the
 * backward-input launches of the CALLING host thread on `stream` give such a tile to 2 - 4 blocks, each with a share of the
 * reduction; their partial tiles meet in the workspace and the last block to arrive adds them in share order and runs the
 * store epilogue (results do not depend on timing).  ws: >= 64 KB + 2 MB, 16-byte aligned, ZERO-FILLED once by the caller
 * (self-resetting tickets at its head), used by one stream at a time; NULL unregisters. */
int cgv_tile_bwd_input_split(void* workspace, size_t workspace_bytes, void* stream);
int cgv_tile_linear_bwd_input_out(const float* gy, const float* z, const float* W, const float* add, float* gx, int M, int N,
                                  int K, int act, const float* z_out, int act_out, void* stream);
int cgv_tile_pair_linear_bwd_input_out(const float* gy_a, const float* z_a, const float* W_a, const float* add_a, float* gx_a,
                                       const float* gy_b, const float* z_b, const float* W_b, const float* add_b, float* gx_b,
                                       int M, int N, int K, int act_a, int act_b, const float* z_out_a, int act_out_a,
                                       const float* z_out_b, int act_out_b, void* stream);
/* gx = add + (gy_a * act_a'(z_a)) W_a + (gy_b * act_b'(z_b)) W_b [+ seg_grad spread over the rows]: the input gradient of two
 * layers of one shape that read the SAME input, as one product with two sources (the first Dense of ContractiveMessageBlock i
 * and of EquiMessageBlock i + 1, cgvae.py:286-305); add / seg_grad may be NULL, seg_* as in
 * cgv_tile_linear_bwd_input_act_add_bcast. */
int cgv_tile_linear_bwd_input_sum2(const float* gy_a, const float* z_a, const float* W_a, const float* gy_b, const float* z_b,
                                   const float* W_b, const float* add, const float* seg_grad, const int64_t* row2seg,
                                   const int32_t* seg_rowptr, int mean, float* gx, int M, int N, int K, int act_a, int act_b,
                                   void* stream);
int cgv_tile_linear_wgrad(const float* g, const float* x, float* gW, int M, int N, int K, int accumulate, void* stream);
int cgv_wgrad_record_bytes(void);
int cgv_wgrad_plan(int M, int N, int K, int* tiles_k /*[host]*/, int* tile_w /*[host]*/, int* n_blocks /*[host]*/);
int cgv_wgrad_lds_floats(int M, int tile_w);
int cgv_grouped_wgrad(const void* table_dev, int n_problems, int total_blocks, int max_lds_floats, void* stream);
/* Data-parallel OPERAND exchange for the same layers.  A weight gradient g^T x has rank <= rows; the bead-level
 * layers have 12 rows per GPU against 0.36 - 3.2 M weights, so instead of all-reducing gW (270 MB per step on the
 * chignolin config) the ranks all-gather their operand rows and each forms the global gradient -- exactly what one
 * process computes on the concatenated batch (the reference's single-process step, scripts/utils.py:110-157).
 *   cgv_pack_operands: one launch copies, for every 64-byte record
 *       { gy, z, x, dst_g, dst_x (pointers), M, N, K, act, block_begin, pad }
 *     dst_g[M,N] = gy * act'(z), dst_x[M,K] = x into the caller's send buffer (block counts: cgv_pack_plan).
 *   cgv_grouped_wgrad_gathered: the 88-byte record of cgv_grouped_wgrad with act = 0, z = NULL and
 *     seg_rows (rows per rank, a multiple of 4) / seg_stride (floats between the rank segments of the gathered
 *     buffer): row m of the problem is row m % seg_rows of segment m / seg_rows; M = ranks * seg_rows (any size).
 *     tiles_k / the block count come from cgv_wgrad_gathered_plan; tile_w is unused.  Output tiles of 64 rows x 128 columns, exact
 *     fp32 MFMA chains in a fixed order: every rank obtains bit-identical gradients. */
int cgv_wgrad_gathered_plan(int M, int N, int K, int seg_rows, int* tiles_k /*[host]*/, int* n_blocks /*[host]*/);
int cgv_grouped_wgrad_gathered(const void* table_dev, int n_problems, int total_blocks, void* stream);
/* tile = 64 | 128: edge of the output tiles (records planned with the same tile); 128 x 128 halves the operand traffic
 * per gW element but measured slower on the shapes of this model -- an opt-in variant */
int cgv_wgrad_gathered_plan_tile(int M, int N, int K, int seg_rows, int tile, int* tiles_k /*[host]*/, int* n_blocks /*[host]*/);
int cgv_grouped_wgrad_gathered_tile(const void* table_dev, int n_problems, int total_blocks, int tile, void* stream);
/* The same grouped weight gradients (table and plan of tile = 128) on the bf16 matrix path with SPLIT operands: every fp32
 * operand value as the exact sum of three bf16 terms, six bf16 MFMA products per fp32 product, fp32 accumulation --
 * fp32-class accuracy (dropped terms < 2^-23 of a product) at 3/8 of the fp32 MFMA time.  Replaces the autograd weight
 * gradients of nn.Linear / Dense (modules.py Dense, conv.py:505-563 inv_dense) on layers with more than 128 operand rows. */
int cgv_grouped_wgrad_split(const void* table_dev, int n_problems, int total_blocks, void* stream);
/* Rank update over gathered operand rows with MFMA tiles -- for layers whose gathered row count is beyond the range in
 * which the FMA-per-row kernel (cgv_grouped_wgrad_adam) pays (~40 rows: 4+ data-parallel ranks of 12 bead rows, or the
 * 36-row [u_mat; v_mat] layers at 2+).  Same records and plan as cgv_grouped_wgrad_gathered (tile 64; accumulate = 0).
 * _sumsq: the tiles are formed and squared, never stored: sumsq[i] = ||gW_i||_F^2 (block partials in `partial`:
 * total_blocks doubles, summed per record in block order); the bias gradients are written.  _adam: the tiles are
 * formed again and go, clipped, through the Adam update of their weights (state from cgv_optim_prepare_extra over
 * those norms); every gW must lie inside the gradient arena, p / m / v are addressed through its offset there.
 * Replaces, for those layers and data-parallel training, Dense's weight gradient (modules.py:103-114) +
 * clip_grad_norm_ + Adam.step (scripts/utils.py:150-157) on the all-gathered batch. */
int cgv_grouped_wgrad_gathered_sumsq(const void* table_dev, int n_problems, int total_blocks, double* partial, double* sumsq,
                                     void* stream);
int cgv_grouped_wgrad_gathered_adam(const void* table_dev, int n_problems, int total_blocks, const float* arena_g,
                                    float* arena_p, float* arena_m, float* arena_v, float lr, float beta1, float beta2,
                                    float eps, const float* state, void* stream);
/* Strip layout of the three gathered launches, for records of at most cgv_wgrad_strip_max_rows() (128) operand rows:
 * one block per 64 ROWS of gW, which stages its g columns once and walks the strip's K / 64 column tiles with the next
 * x tile loading under the current tile's MFMAs (the tile layout above gives each 64 x 64 tile a block of its own that
 * spends most of its life on its first loads).  Records as above with block_begin counted in strip blocks
 * (cgv_wgrad_strip_plan: ceil(N / 64)); tiles_k / tile_w unused.  max_rows = the largest M of the table (sizes the LDS).
 * Bit-identical to the tile layout: same MFMA chains in the same row order. */
int cgv_wgrad_strip_max_rows(void);
int cgv_wgrad_strip_plan(int M, int N, int K, int seg_rows, int* n_blocks /*[host]*/);
int cgv_grouped_wgrad_strip(const void* table_dev, int n_problems, int total_blocks, int max_rows, void* stream);
/* The strip layout on the bf16 matrix path with split operands (fp32-class accuracy, see cgv_grouped_wgrad_split): for
 * records of at most cgv_wgrad_strip_split_max_rows() (96) rows.  x is split ONCE per problem into three bf16 planes in `ws`
 * (record field `pad` = the problem's offset in ws in 256-byte units; cgv_wgrad_strip_split_plane_bytes(M, K) bytes each),
 * g = gy * act'(z) once per strip in registers; the strips then walk their column tiles without arithmetic on the operands.
 * max_rows / max_k: the largest M / K of the table.  modules.py:103-114 (autograd of Dense: gW = g^T x, gb = sum_m g). */
int cgv_wgrad_strip_split_max_rows(void);
size_t cgv_wgrad_strip_split_plane_bytes(int M, int K);
int cgv_grouped_wgrad_strip_split(const void* table_dev, int n_problems, int total_blocks, int max_rows, int max_k, void* ws,
                                  size_t ws_bytes, void* stream);
int cgv_grouped_wgrad_strip_sumsq(const void* table_dev, int n_problems, int total_blocks, int max_rows, double* partial,
                                  double* sumsq, void* stream);
int cgv_grouped_wgrad_strip_adam(const void* table_dev, int n_problems, int total_blocks, int max_rows, const float* arena_g,
                                 float* arena_p, float* arena_m, float* arena_v, float lr, float beta1, float beta2,
                                 float eps, const float* state, void* stream);
int cgv_pack_record_bytes(void);
int cgv_pack_plan(int M, int N, int K, int* n_blocks /*[host]*/);
int cgv_pack_operands(const void* table_dev, int n_problems, int total_blocks, void* stream);

/* ---------------------------------------------------------------------------------------
 * Decoder tail (cgvae.py:462-481):  xyz[a] = v[mapping[a], chan[a], :] - [offset] mean over the bead of the
 * same + cg_xyz[mapping[a]].  rowptr [n_beads+1] / atom [n_atoms]: bead -> atoms CSR (the dst-sorted view of
 * the contraction plan: cgv_csr_build on mapping, rowptr_d / eid_d); chan int64 [n_atoms] = rank of the atom
 * inside its bead (CG2ChannelIdx, cgvae.py:451-460), chan < n_feat.  bwd writes ALL of g_v [n_beads,F,3]
 * (zeros outside the addressed slots) and, when not NULL, g_cg_xyz [n_beads,3].
 * ------------------------------------------------------------------------------------- */
int cgv_reconstruct_fwd(const float* v, const float* cg_xyz, const int32_t* rowptr, const int32_t* atom, const int64_t* chan,
                        int n_beads, int n_feat, int offset, float* xyz /*[n_atoms,3]*/, void* stream);
int cgv_reconstruct_bwd(const float* g_xyz, const int32_t* rowptr, const int32_t* atom, const int64_t* chan, int n_beads,
                        int n_feat, int offset, float* g_v, float* g_cg_xyz /*or NULL*/, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused ELBO loss -- replaces KL (scripts/utils.py:81-86, incl. its (mu1-mu2)^2 / std2 term) and the
 * loss assembly of scripts/utils.py:117-141:  loss = recon + beta*KL + gamma*graph with
 *   recon = mean((xr - x)^2),  graph = mean_b((|xr_a - xr_b|_eps - |x_a - x_b|_eps)^2), eps = 1e-6 inside the sqrt.
 * out4 = { loss, KL, recon, graph }.  The same launch stores d loss / d{mu, sigma, prior_mu, prior_std,
 * xyz_recon}; cgv_elbo_scale multiplies them by the upstream scalar (device pointer) in backward.
 * bonds: int64 [n_bonds, 2] batch-global atom ids (bond_edge_list of CG_collate).
 * ------------------------------------------------------------------------------------- */
/* workspace of cgv_elbo_workspace_bytes bytes (0 for small bead batches): lets the KL terms of a large batch run on
 * their own multi-block launch; with NULL everything runs in the one-block kernel (same result up to summation order). */
size_t cgv_elbo_workspace_bytes(int n_beads, int n_feat);
int cgv_elbo_fwd(const float* mu, const float* sigma, const float* prior_mu, const float* prior_std, const float* xyz,
                 const float* xyz_recon, const int64_t* bonds, int n_beads, int n_feat, int n_atoms, int n_bonds,
                 float beta, float gamma, float* out4, float* loss_out /*[1] or NULL: the loss once more, as a tensor of
                 its own (an autograd output that is not a view of the non-differentiable terms: saves the clone)*/,
                 float* g_mu, float* g_sigma, float* g_prior_mu, float* g_prior_std, float* g_xyz_recon,
                 void* workspace /*or NULL*/, size_t workspace_bytes, void* stream);
int cgv_elbo_scale(const float* g_loss, float* g_mu, float* g_sigma, float* g_prior_mu, float* g_prior_std, int n_bead_elems,
                   float* g_xyz_recon, int n_atom_elems, void* stream);
/* Decoder tail (cgv_reconstruct_fwd: cgvae.py:462-481) + the ELBO above + BOTH their gradients in ONE launch, one block
 * per bead: xyz_recon is an OUTPUT here, and besides d loss / d{mu, sigma, prior_mu, prior_std, xyz_recon} the launch
 * leaves g_V [n_beads, F, 3] = d loss / d V (what cgv_reconstruct_bwd would compute from g_xyz_recon).  rowptr / atom_of /
 * bead_of: the atom -> bead plan (CSR by bead, atom ids and bead ids in bead-sorted order).  Sums in double, block
 * partials added in block order by the block that arrives last (device-scope ticket): deterministic.
 * workspace: cgv_loss_tail_workspace_bytes(n_beads) bytes, 16-byte aligned, ZERO before the first launch (every launch
 * leaves its ticket word at zero again).  Limits: cgv_loss_tail_supported (atoms + beads <= 3072: both coordinate sets live in LDS). */
int cgv_loss_tail_supported(int n_beads, int n_feat, int n_atoms, int n_bonds);
size_t cgv_loss_tail_workspace_bytes(int n_beads);
int cgv_loss_tail(const float* V, const float* cg_xyz, const int32_t* rowptr, const int32_t* atom_of, const int32_t* bead_of,
                  const int64_t* chan, const float* mu, const float* sigma, const float* prior_mu, const float* prior_std,
                  const float* xyz, const int64_t* bonds, int n_beads, int n_feat, int n_atoms, int n_bonds, int offset,
                  float beta, float gamma, float* xyz_recon, float* out4, float* loss_out, float* g_mu, float* g_sigma,
                  float* g_prior_mu, float* g_prior_std, float* g_xyz_recon, float* g_V, void* workspace, size_t workspace_bytes,
                  void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused optimiser step over a flat fp32 arena of the parameters that receive gradients --
 * replaces the skip rule, clip_grad_norm_(params, 0.01) and Adam.step() of
 * scripts/utils.py:145-157 (torch.optim.Adam defaults: no amsgrad, no weight decay).
 *   skip   = loss >= skip_threshold || isnan(loss)         (loss == NULL: never skip)
 *   norm   = |grad_scale| * ||g||_2 ;  coef = min(1, max_norm / (norm + 1e-6))
 *   g' = g * coef * grad_scale ; m,v,p updated as Adam does ; step counter += 1
 * `state` is cgv_optim_state_floats() device floats, zero-initialised by the caller once:
 *   [0] step  [1] last grad norm  [2] clip*scale  [3] 1-b1^t  [4] sqrt(1-b2^t)  [5] skipped?  [6] #skipped
 * `partial` is cgv_optim_partial_floats() device floats of scratch, ZEROED once by the caller (its last 16 floats hold the
 * ticket word of the one-launch norm + decision pass; every launch leaves it zero).  No host synchronisation.
 * ------------------------------------------------------------------------------------- */
int cgv_optim_state_floats(void);
int cgv_optim_partial_floats(void);
int cgv_adam_clip_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                       float eps, float max_norm, float grad_scale, const float* loss /*[1] or NULL*/,
                       float skip_threshold, float* state, float* partial, void* stream);
/* Its two halves: cgv_optim_prepare = global norm, skip decision, clip coefficient, bias corrections -> `state`;
 * cgv_adam_apply = the parameter / moment pass over ONE contiguous range (pointers already offset, 16-byte aligned),
 * reading `state`.  prepare + apply over the whole arena == cgv_adam_clip_step; apply may be issued per range, on
 * another stream, and after further kernels (the trainer overlaps the decoder range's update with the next forward). */
int cgv_optim_prepare(const float* g, int64_t n, float beta1, float beta2, float max_norm, float grad_scale,
                      const float* loss /*[1] device or NULL*/, float skip_threshold, float* state, float* partial,
                      void* stream);
int cgv_adam_apply(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                   const float* state, void* stream);
/* torch.optim.SGD defaults (scripts/run_ala.py:43, `-optimizer sgd`): p -= lr * clip * g over one range, clip coefficient and
 * skip decision taken from the state cgv_optim_prepare wrote (scripts/utils.py:145-157). */
int cgv_sgd_apply(float* p, const float* g, int64_t n, float lr, const float* state, void* stream);

/* Rank-update layers (bead-level Dense / nn.Linear weights: M <= 64 operand rows against 0.36 - 3.2 M weights).  Their
 * weight gradient gW = g^T x is never written: cgv_wgrad_gram gives its squared Frobenius norm from the operands
 * (sum over row pairs of (g_a . g_b)(x_a . x_b), in double; sumsq[i] for table record i) and writes the bias gradients;
 * cgv_optim_prepare_extra is cgv_optim_prepare over the materialised part of the arena plus those norms; and
 * cgv_grouped_wgrad_adam re-forms each gW tile (the table and tiling of cgv_grouped_wgrad) and applies the clipped Adam
 * update to the weights, moments addressed through gW's offset in the gradient arena.  Replaces, for these layers,
 * Dense's autograd weight gradient (CoarseGrainingVAE/modules.py:103-114) + clip_grad_norm_ + Adam.step
 * (scripts/utils.py:150-157) with 24 instead of 36 bytes of HBM traffic per weight.  Records must have accumulate = 0.
 * The records may address GATHERED operands (seg_rows / seg_stride as for cgv_grouped_wgrad_gathered: the rows of all
 * data-parallel ranks, M = world x rows per rank): the update every rank then applies is the whole batch's, and no
 * rank ever materialises these gradients either.  max_rows: the largest M in the table (sizes the kernel's LDS). */
int cgv_wgrad_gram(const void* table_dev, int n_problems, int max_rows, double* sumsq, void* workspace,
                   size_t workspace_bytes, void* stream);
size_t cgv_wgrad_gram_workspace_bytes(int n_problems);
/* cgv_wgrad_gram for records of up to cgv_wgrad_gram_mfma_max_rows() = 128 rows (gathered rows of 4 - 8 ranks, bead rows
 * of a large batch): the Gram matrices come from fp64 MFMA tiles instead of a walk over row pairs (same doubles). */
int cgv_wgrad_gram_mfma(const void* table_dev, int n_problems, int max_rows, double* sumsq, void* workspace,
                        size_t workspace_bytes, void* stream);
size_t cgv_wgrad_gram_mfma_workspace_bytes(int n_problems, int max_rows);
int cgv_wgrad_gram_mfma_max_rows(void);
int cgv_rank_update_supported(int M, int N, int K);   /* 1 when a layer with M operand rows can take this path (M <= 64) */
int cgv_optim_prepare_extra(const float* g, int64_t n, const double* extra, int n_extra, float beta1, float beta2,
                            float max_norm, float grad_scale, const float* loss, float skip_threshold, float* state,
                            float* partial, void* stream);
int cgv_grouped_wgrad_adam(const void* table_dev, int n_problems, int total_blocks, int max_lds_floats,
                           const float* arena_g, float* arena_p, float* arena_m, float* arena_v, float lr, float beta1,
                           float beta2, float eps, const float* state, void* stream);
/* cgv_grouped_wgrad_adam with the FLAT block layout, for records of at most 16 operand rows: the update is elementwise over
 * a weight's contiguous [N, K] array, so a block takes a contiguous range of q4 float4 of it (whole 128-byte lines of
 * p / m / v, each read and written exactly once) with x for all K columns and the g rows of its range in LDS, instead of
 * 64 rows x one k tile (row segments at a stride of K floats, whose border lines neighbouring blocks fetch twice).  Same
 * sum order over the operand rows: results are bit-identical to cgv_grouped_wgrad_adam.
 *   cgv_rank_flat_quantum   default q4 (float4 per block; any multiple of 2048 is accepted, 0 = this default)
 *   cgv_rank_flat_plan      blocks / LDS floats of one record; CGV_E_UNSUPPORTED when the shape does not take the layout
 *                           (more than 16 rows, or x [M, K] + g beyond the LDS budget) -- the launch then stays with
 *                           cgv_grouped_wgrad_adam.  A table's block_begin is the prefix of THESE block counts.
 * Replaces, like cgv_grouped_wgrad_adam, the autograd weight gradient of the bead-level nn.Linear / Dense layers
 * (modules.py Dense) + torch.optim.Adam.step on them (scripts/run_ala.py:147, utils.py:157). */
int cgv_rank_flat_quantum(void);
int cgv_rank_flat_plan(int M, int N, int K, int q4, int* n_blocks /*[host]*/, int* lds_floats /*[host]*/);
int cgv_grouped_wgrad_adam_flat(const void* table_dev, int n_problems, int total_blocks, int max_lds_floats, int q4,
                                const float* arena_g, float* arena_p, float* arena_m, float* arena_v, float lr, float beta1,
                                float beta2, float eps, const float* state, void* stream);
/* Both layouts in ONE launch: records [0, n_flat) flat (flat_blocks blocks, block prefix from 0), records [n_flat, n_problems)
 * tiled as for cgv_grouped_wgrad_adam (tiled_blocks blocks, their own prefix from 0).  The tiled records are layers of
 * more operand rows (36: three stacked heads), bound by the FMAs that form a tile rather than by p / m / v: their blocks are
 * dealt evenly among the flat ones, so they run beside blocks that wait for memory.  max_lds_floats: the larger request. */
int cgv_grouped_wgrad_adam_mixed(const void* table_dev, int n_flat, int n_problems, int flat_blocks, int tiled_blocks,
                                 int max_lds_floats, int q4, const float* arena_g, float* arena_p, float* arena_m,
                                 float* arena_v, float lr, float beta1, float beta2, float eps, const float* state,
                                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CGVAE_HIP_H */
