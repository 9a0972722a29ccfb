/* hidenn_fem.h -- C ABI of libhidenn_hip.so (MI355X / gfx950 HIP kernels for the
 * HiDeNN-FEM element-evaluation + energy hot path).
 *
 * The reference (achraf-15/HiDeNN-FEM) is pure Python/PyTorch and has no FFI of
 * its own; the boundary below is what a ctypes binding inside the reference's
 * src/models.py / src/loss.py would call instead of its ATen op chains.  Each
 * entry point cites the reference code it replaces (paths are relative to the
 * reference repo root).
 *
 * Conventions
 *   - plain pointers and sizes only; every device buffer is allocated and owned
 *     by the caller (torch tensors on the ROCm device), contiguous, 16-B aligned;
 *   - every call takes the device ordinal and a hipStream_t (as void*); work is
 *     enqueued on that stream, nothing synchronises, results are stream-ordered;
 *   - return 0 = ok, <0 = argument error, >0 = hipError_t; message via
 *     hfem_last_error() (thread-local).  No exceptions, no exit();
 *   - degenerate elements are not errors: divisions by a tiny det / clamp(1e-10)
 *     spans behave as in the reference and NaN/Inf propagate;
 *   - dtype: fp64 (the parity contract of BASELINE.md);
 *   - no thread-affine state: PyTorch runs backward on another thread.
 */
#ifndef HIDENN_FEM_H
#define HIDENN_FEM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HFEM_VERSION 114   /* 0.1.1: round 4 -- plan blobs, sharded L-BFGS, fp32 arithmetic; append-only since 113 */

int hfem_version(void);
const char *hfem_last_error(void);
/* number of HIP devices visible, or -1 (used by the loader to fail loudly) */
int hfem_device_count(void);

/* ------------------------------------------------------------------ TRI3 + EDGE2
 * Plane-stress material: mat = {c11, c12, c22, c33} of C (src/loss.py:29-32).
 * W  = sum_q w_q of the triangle rule (src/utils.py:13-81; 0.25 for the default
 *      gauss_order=4).
 * Bk = [3][2] body-force table  B_k = sum_q w_q N_k(xi_q) b(xi_q), b evaluated
 *      at the REFERENCE points as src/loss.py:80 does (zeros by default).
 * Traction: Tconst[4] = {c_i t, c_j t} with c_i = sum_q w_q (1-xi_q), c_j = sum_q
 *      w_q xi_q over the raw Legendre nodes (src/loss.py:96-110), or a
 *      per-edge table T[Ned][4] = {T_i, T_j} when t depends on the point.      */

/* Planless variant: element-parallel, fp64 global atomics.  Replaces
 * EnergyLoss2D.domain_energy + its autograd backward (src/loss.py:55-88,
 * src/models.py:316-357) on ASSEMBLED arrays X,U [Nn][2].
 * ACCUMULATES: loss_acc[0] += E_domain, gX += dE/dX, gU += dE/dU (caller zeroes).
 * gX/gU may be NULL (forward only).  Elements [e_begin, e_end).               */
int hfem_tri3_energy_atomic(int device, const double *X, const double *U,
                            const int32_t *conn, int64_t e_begin, int64_t e_end, int64_t nn,
                            const double mat[4], double W, const double Bk[6],
                            double *loss_acc, double *gX, double *gU, void *stream);

/* Planless Neumann-edge work; replaces EnergyLoss2D.edge_energy + backward
 * (src/loss.py:91-110, src/models.py:359-376).  ACCUMULATES the NEGATIVE work:
 * loss_acc[0] -= sum ds*m, gX/gU += d(-work).  edges [Ned][2] int32, i<j.     */
int hfem_edge2_energy_atomic(int device, const double *X, const double *U,
                             const int32_t *edges, int64_t ned, const double *T,
                             const double Tconst[4], double *loss_acc, double *gX,
                             double *gU, void *stream);

/* Planned variant -- the fast path.  A plan is an owner-computes tiling of the
 * mesh (host-side preprocessing, once per mesh): elements are sorted along a
 * Hilbert curve of their centroids and cut into tiles; every node is owned by
 * exactly one tile; a tile evaluates its home elements plus the halo elements
 * that touch its owned nodes, with node data staged in LDS and gradients
 * accumulated in LDS, so each gradient row is written exactly once with a plain
 * store (no global atomics, no zero fill).  TRI3 plans whose elements pair up
 * along shared fan edges (about 62 % of them or more: structured splits with fixed or random diagonals)
 * store two elements per slot, A = (n,b,c) and B = (n,c,d), each in its own
 * local node order, and run the paired-slot kernel (16 instead of 24 LDS
 * atomics per pair); other meshes keep one element per slot.  The choice is
 * made at plan creation ("plan_elem_order" -1 = auto) and is invisible in the
 * results up to fp64 summation order.
 *
 * It also fuses the free/fixed parameter assembly of src/models.py:292-305:
 *   x_src[n] >= 0 -> coords of node n = x_free[x_src[n]]   (node_coords_free row)
 *   x_src[n] <  0 -> coords of node n = x_fixed[-1-x_src[n]]
 *   u_src likewise for u_free / u_fixed rows (Dirichlet rows).
 * NULL maps mean identity (x_free = X, u_free = U are the full arrays).
 * coords_hint [Nn][2] (host) is only used for the locality sort.
 * device < 0 builds a host-only plan (no HIP call; for CPU tests).            */
typedef struct hfem_plan hfem_plan;

typedef struct hfem_plan_stats {
    int64_t n_elems, n_nodes, n_edges;
    int32_t n_tiles, tile_elems;          /* home elements per tile */
    int64_t tile_elem_total;              /* home + halo elements over all tiles */
    int64_t tile_node_total;              /* owned + halo local nodes over all tiles */
    int32_t max_tile_nodes, max_tile_owned, max_tile_elems, max_tile_edges;
    int64_t device_bytes;                 /* plan arrays resident in HBM */
    int32_t lds_bytes;                    /* dynamic LDS per workgroup */
    int32_t shards;                       /* ranks the tile order was prepared for ("plan_shards"; 1 = not sharded) */
    int32_t threads_per_tile;             /* workgroup size of the tiled kernel this plan takes */
    int32_t paired;                       /* 1: paired slots (two fan-adjacent TRI3 per slot) */
    int32_t slot_rows;                    /* paired plans: slots per thread in the widest tile */
    int32_t store_policy;                 /* gradient stores of this plan: 16 sc1 write-through, 2 nt (meshes of >= 750 k nodes
                                             whose rows have locality), 0 plain -- "store_policy" -1 = this choice */
    int32_t nodes_per_elem;               /* 3 (TRI3) or 4 (QUAD4) -- was reserved0 before version 114 */
    double row_line_factor;               /* distinct 128-byte lines among a tile's coordinate rows / the minimum, mean over the
                                             tiles: ~1.3 = rows stored with locality, up to 8 = random numbering */
} hfem_plan_stats;

int hfem_plan_create(int device, const int64_t *conn, int64_t ne, int64_t nn,
                     const double *coords_hint, const int32_t *x_src, const int32_t *u_src,
                     const int64_t *edges, int64_t ned, int32_t tile_elems,
                     hfem_plan **out);
/* Same with an explicit element type: nodes_per_elem = 3 (TRI3, conn [Ne][3]) or 4 (QUAD4 extension,
 * conn [Ne][4], local nodes CCW).                                                          */
int hfem_plan_create_ex(int device, const int64_t *conn, int64_t ne, int64_t nn, int32_t nodes_per_elem,
                        const double *coords_hint, const int32_t *x_src, const int32_t *u_src,
                        const int64_t *edges, int64_t ned, int32_t tile_elems, hfem_plan **out);
int hfem_plan_destroy(hfem_plan *plan);
int hfem_plan_get_stats(const hfem_plan *plan, hfem_plan_stats *out);
/* Copy a host-side plan array out (for tests).  which: 0 tile_desc [n_tiles][8]
 * i32, 1 elem_pack u32, 2 node_src [.][2] i32, 3 edge_pack u32, 4 edge_gid i32,
 * 5 elem_gid i32 (global element id of every slot's element A; -1 = padding), 6 lab stamps
 * (lab build only), 7 elem_pack_hi u32 (QUAD4 plans: 4th local node of every slot; paired TRI3
 * plans: node d + presence/home bits of element B), 8 tile_chunks (chunked lab order), 9 elem_gid_b
 * i32 (paired plans: global id of every slot's element B; -1 = none), 10 shard_desc [shards][4] i32 = {tile_lo,
 * tile_mid, tile_hi, 0} per rank: the rank's boundary tiles are [tile_lo, tile_mid), its interior tiles [tile_mid, tile_hi),
 * 11 owned_node_ids [Nn] i32: the global node id of every tile's owned nodes in (tile, local) order -- a permutation of the
 * nodes; a caller that stores its parameter rows in THIS order gives every tile one contiguous run of rows (full-line
 * gradient stores, perfectly coalesced gathers).
 * Returns the element count, or <0.  buf may be NULL to query the size.  The per-tile arrays are laid out
 * with uniform strides (tile t's node records start at t * node_stride, its slot records at t * elem_stride, padded;
 * extra records follow the last tile): walk them through tile_desc's offsets and counts, not as dense arrays.    */
int64_t hfem_plan_export(const hfem_plan *plan, int which, void *buf, int64_t cap_elems);

/* A plan as one relocatable byte string: the planner costs ~1 s per 10^6 elements and every rank of a multi-GPU job (and
 * every later run on the same mesh) needs the same plan -- one process builds and serialises, the others deserialise
 * (hidenn_fem_amd/plan.py: TilePlan.to_bytes / from_bytes / the cache_dir of TilePlan and bench.py --gpus N).  The blob
 * holds the host plan, the creation-time decisions (element order, tile shape, store policy, shard layout) and a checksum;
 * it is valid for the library version that wrote it (HFEM_VERSION) on any device.  serialize: buf NULL queries the size;
 * returns bytes or <0.  deserialize: device < 0 = host-only plan.  No reference counterpart (the reference has no plan).  */
int64_t hfem_plan_serialize(const hfem_plan *plan, void *buf, int64_t cap_bytes);
int hfem_plan_deserialize(int device, const void *blob, int64_t n_bytes, hfem_plan **out);

#define HFEM_FLAG_NO_GX 1   /* do not write gx_free (nodes fixed / no r-adaptivity) */
#define HFEM_FLAG_NO_GU 2   /* do not write gu_free */
#define HFEM_FLAG_NO_EDGES 4 /* domain energy only (EnergyLoss2D.domain_energy, src/loss.py:55-88) */
#define HFEM_FLAG_NO_LOSS_SUM 8 /* leave the per-tile partial energies unsummed, loss_out untouched
                                   (gradient-only callers; bench.py's kernel-only roofline leg) */
#define HFEM_FLAG_PHYSICAL_GRAD 64 /* opt-in: grad_u = G Jinv (HFEM_GRAD_PHYSICAL below) instead of the reference's
                                   * G Jinv^T (src/models.py:351).  fp64 TRI3 plan path.            */
#define HFEM_FLAG_DETERMINISTIC 128 /* fixed-order accumulation: gradients (and the loss) are bit-identical run to
                                   * run.  Node-centric gather over the node -> element adjacency (every element
                                   * is re-evaluated by each of its corners: 3x the flops, no atomics); the
                                   * cross-check of the atomic kernels SURVEY section 5 asks for, ~3-4x slower.
                                   * Whole plan only (tile_begin = 0, tile_end = -1 / n_tiles), fp64.          */
#define HFEM_FLAG_SUM_PREVIOUS 32 /* with NO_LOSS_SUM: one extra workgroup of THIS launch sums the tile energies the
                                   * PREVIOUS NO_LOSS_SUM launch on this plan left (the plan keeps two banks) into
                                   * loss_out -- the energy of evaluation k arrives with launch k+1, and the 1-block
                                   * reduction + kernel boundary leave the critical path.  hfem_plan_loss_sum
                                   * delivers the last one.  TRI3 default kernel path only.                    */

#define HFEM_FLAG_PEER_GET 512  /* paired-slot plans (no chained records) and 512-thread one-element-per-slot plans, after hfem_plan_set_peer_get: the launch starts with 8 service workgroups that
                                 * ARE the peer-window get (hfem_peer_iface_get's wait + unpack into this launch's x_free / u_free)
                                 * and the tiles named there (the rank's boundary tiles) wait for them inside the kernel        */
#define HFEM_FLAG_PEER_PUT 2048  /* hfem_tri3_energy_adam_step_ex with HFEM_FLAG_PEER_GET, after hfem_plan_set_peer_put: the launch also PUBLISHES --
                                 * its boundary tiles store the new rows of their interface nodes into every rank's window at
                                 * write-out and the last of them completes the put: one launch per owner-sharded training step */
#define HFEM_FLAG_FP32_MATH 1024 /* hfem_tri3_energy_plan_f32, hfem_tri3_energy_adam_step_ex(dtype 1): fp32 ARITHMETIC as well as fp32 rows -- what the reference itself
                                 * computes in for its default dtype (src/loss.py:16): packed-fp32 element math (the two elements of
                                 * a slot side by side), double LDS accumulators (rows rounded once); tile energies and loss stay fp64.  Paired-slot
                                 * plans; body force, tile ranges, NO_LOSS_SUM / SUM_PREVIOUS / SAME_BANK as the fp64 entry point. */
#define HFEM_FLAG_SAME_BANK 256 /* with NO_LOSS_SUM: another tile range of the SAME evaluation as the previous NO_LOSS_SUM
                                 * launch on this plan (a rank's boundary tiles after its interior tiles): the tile energies
                                 * go to the bank that launch wrote, so one hfem_plan_loss_sum / hfem_plan_iface_pack over
                                 * the union delivers the evaluation's energy.  Not with SUM_PREVIOUS.          */

/* One fwd+bwd "element-eval" pass over tiles [tile_begin, tile_end):
 *   loss_out[0]  = sum_elem A (W psi - beta)  -  sum_edge ds m        (OVERWRITTEN)
 *   gx_free[r]   = dL/d node_coords_free[r]   for rows owned by those tiles (OVERWRITTEN)
 *   gu_free[r]   = dL/d u_free[r]             likewise
 * = EnergyLoss2D.__call__ + loss.backward() (src/loss.py:113-116) including the
 * coords/u_full assembly and its backward.  Rows owned by tiles outside the range
 * are not touched (element sharding).  tile_end = -1: all.   */
int hfem_tri3_energy_plan(hfem_plan *plan, const double *x_free, const double *x_fixed,
                          const double *u_free, const double *u_fixed,
                          const double mat[4], double W, const double Bk[6],
                          const double *T_edge, const double Tconst[4],
                          int32_t tile_begin, int32_t tile_end, double *loss_out,
                          double *gx_free, double *gu_free, int32_t flags, void *stream);
/* Same pass for an fp32 model (the reference's default dtype): x / u rows and the gradient rows are float
 * [rows][2]; they are widened on load and rounded once on store, arithmetic and loss_out stay fp64.
 * Needs a zero body force (Bk NULL or all zero) and the default tile shape; any other plan returns an
 * argument error (the caller then widens to the fp64 entry point).  With HFEM_FLAG_FP32_MATH the arithmetic is fp32 too
 * (csrc/tri3_pair_f32.hip): results within the band the reference's own fp32 run occupies around exact arithmetic
 * (tests/test_gpu_tri3_f32.py), ~1.4x faster; body force allowed.                                     */
int hfem_tri3_energy_plan_f32(hfem_plan *plan, const float *x_free, const float *x_fixed,
                              const float *u_free, const float *u_fixed,
                              const double mat[4], double W, const double Bk[6],
                              const double *T_edge, const double Tconst[4],
                              int32_t tile_begin, int32_t tile_end, double *loss_out,
                              float *gx_free, float *gu_free, int32_t flags, void *stream);
/* One training step in one launch (no reference counterpart; the reference runs `loss.backward(); optimizer.step()`
 * with torch.optim.Adam, examples/example4.py:53-64 commented variant, example1-3): the pass above, but every tile
 * applies Adam's update to the rows it owns instead of storing their gradient.  m_*, v_* [rows][2]: moments, updated
 * in place.  x_out / u_out: the NEW parameter rows, buffers different from x_free / u_free (swap after the launch).
 * bc_dev: device {1 - beta1^step, sqrt(1 - beta2^step)} of this step, written by hfem_adam_prep (which also bumps the
 * device step counter) in stream order just before.  loss_out: energy
 * at the old parameters.  Whole plan, default tile shape, zero body force; flags: HFEM_FLAG_NO_EDGES, _NO_LOSS_SUM. */
int hfem_tri3_energy_adam_step(hfem_plan *plan, const double *x_free, const double *x_fixed,
                               const double *u_free, const double *u_fixed, const double mat[4], double W,
                               const double *T_edge, const double Tconst[4], double *x_out, double *u_out,
                               double *m_x, double *v_x, double *m_u, double *v_u, double lr_x, double lr_u,
                               double beta1, double beta2, double eps, const double *bc_dev,
                               double *loss_out, int32_t flags, void *stream);
/* General form.  dtype: 0 = fp64 rows, 1 = fp32 rows -- x / u / fixed rows, moments and new rows all float (an fp32 model, the
 * reference's default dtype src/loss.py:16, src/models.py:274, trains in one launch per iteration with no widening copy;
 * element arithmetic and loss_out stay fp64, the update is torch.optim.Adam's fp32 arithmetic on the once-rounded
 * gradient).  Bk [3][2]: body-force table as in hfem_tri3_energy_plan (NULL / zeros: none; not with SUM_PREVIOUS).
 * [tile_begin, tile_end) (-1 = all): a tile range updates exactly the rows its tiles own and leaves the other rows of
 * x_out / u_out alone (element sharding: the caller keeps both parameter buffers complete); flags accept
 * HFEM_FLAG_SAME_BANK for the second range of one evaluation.                                                      */
int hfem_tri3_energy_adam_step_ex(hfem_plan *plan, int32_t dtype, const void *x_free, const void *x_fixed,
                                  const void *u_free, const void *u_fixed, const double mat[4], double W,
                                  const double Bk[6], const double *T_edge, const double Tconst[4], void *x_out,
                                  void *u_out, void *m_x, void *v_x, void *m_u, void *v_u, double lr_x, double lr_u,
                                  double beta1, double beta2, double eps, const double *bc_dev, int32_t tile_begin,
                                  int32_t tile_end, double *loss_out, int32_t flags, void *stream);
int hfem_adam_prep(int device, int64_t *step_dev, double beta1, double beta2, double *bc_dev, void *stream);
/* loss_out[0] = sum, in tile order, of the per-tile partial energies that a launch with
 * HFEM_FLAG_NO_LOSS_SUM over the same tile range left in the plan (TRI3 and QUAD4 plans alike).   */
int hfem_plan_loss_sum(hfem_plan *plan, int32_t tile_begin, int32_t tile_end, double *loss_out, void *stream);

/* Span stamps (measurement aid): while dev_buf (n_slots x n_tiles x 2 uint64, caller-owned device memory) is set, launch i
 * of the paired-slot kernel through hfem_tri3_energy_plan writes for every tile it evaluates {tick at which the tile's
 * workgroup started, tick at which its first wave's gradient stores had drained} (s_memrealtime, 100 MHz) at
 * dev_buf[(i % n_slots) * n_tiles * 2 + 2 * tile].  max(end) - min(start) over a launch's tiles = its duration from the
 * first workgroup's start to the last one's end, measurable INSIDE any launch sequence (bench.py's cache regimes).
 * NULL = off (default).  Plans that take another kernel ignore it.                                                */
int hfem_plan_set_span_stamps(hfem_plan *plan, uint64_t *dev_buf, int64_t n_slots);

/* Process-wide DEFAULTS (atomics) that hfem_plan_create captures into the plan it builds; changing one
 * never affects an existing plan, and launches on different plans may run from different threads (a plan
 * serialises its own launches with a mutex).  Product knobs: "tiled_block" (threads per tile of the
 * one-element-per-slot kernels: 256, 512, 1024), "store_policy" (gradient stores: -1 = by mesh size and row locality -- nt
 * from 750 k nodes, sc1 write-through below --, 16, 2, 0), "tiled_fast", "fast_const_caps", "quad4_const_caps",
 * "plan_elem_order" (-1 auto, 3 one element per slot, 5 paired slots, 6 paired slots chained into strips), "plan_node_cap"
 * (home nodes per tile; -1 = the shard-aware policy), "plan_shards" (ranks the tiles will be split over: tiles are sized
 * for the elements PER RANK and every rank's boundary tiles come first in its range), "plan_pair_block" (threads per tile
 * of a paired plan: -1 auto, 256, 512), "plan_chunk_cap", "plan_curve" (0 Morton, 1 Hilbert), "plan_snap" (tile cuts snap
 * back to coarse curve cells by up to that percentage of a tile; 0 = off), "plan_read_pack" (paired slots also packed
 * against ds_read_b128 bank conflicts: number of partner rows examined, default 2; 0 = off).  The ablation / stamp /
 * pipeline knobs exist only in the lab build (libhidenn_hip_lab.so, hfem_get_option("lab_build") == 1); the product
 * library rejects them.
 * hfem_get_option returns the value or -1.                                     */
int hfem_set_option(const char *name, int value);
int hfem_get_option(const char *name);

/* ------------------------------------------------------------------ per-point TRI3
 * Unfused forward of src/models.py:316-357 on assembled X,U: u_h [M][2],
 * detJ [M], grad_u [M][2][2] at reference points x_eval [M][2] of elements
 * elem_id [M] (int64, the reference's dtype).                                  */
int hfem_tri3_eval_fwd(int device, const double *X, const double *U, const int32_t *conn,
                       const double *x_eval, const int64_t *elem_id, int64_t m,
                       double *u_h, double *detJ, double *grad_u, void *stream);
/* Backward of the above: cotangents cu [M][2], cd [M], cg [M][2][2] ->
 * ACCUMULATES gX,gU [Nn][2] (fp64 atomics; caller zeroes).                    */
int hfem_tri3_eval_bwd(int device, const double *X, const double *U, const int32_t *conn,
                       const double *x_eval, const int64_t *elem_id, int64_t m,
                       const double *cu, const double *cd, const double *cg,
                       double *gX, double *gU, void *stream);
/* Same two with an explicit gradient convention.  HFEM_GRAD_REFERENCE: dN_dx = Jinv * dN_dxi exactly as
 * src/models.py:351 computes it (grad_u = G Jinv^T; the parity contract and the default everywhere).
 * HFEM_GRAD_PHYSICAL (opt-in): dN_dx = Jinv^T * dN_dxi, i.e. grad_u = G Jinv -- the physical
 * gradient: exact for linear fields and invariant to the local node order of an element.            */
#define HFEM_GRAD_REFERENCE 0
#define HFEM_GRAD_PHYSICAL 1
int hfem_tri3_eval_fwd_conv(int device, const double *X, const double *U, const int32_t *conn,
                            const double *x_eval, const int64_t *elem_id, int64_t m,
                            double *u_h, double *detJ, double *grad_u, int32_t convention, void *stream);
int hfem_tri3_eval_bwd_conv(int device, const double *X, const double *U, const int32_t *conn,
                            const double *x_eval, const int64_t *elem_id, int64_t m,
                            const double *cu, const double *cd, const double *cg,
                            double *gX, double *gU, int32_t convention, void *stream);
/* Edge branch of src/models.py:359-376: u_h [M][2], ds [M]; and its backward. */
int hfem_edge2_eval_fwd(int device, const double *X, const double *U, const int32_t *edges,
                        const double *xi, const int64_t *edge_id, int64_t m,
                        double *u_h, double *ds, void *stream);
int hfem_edge2_eval_bwd(int device, const double *X, const double *U, const int32_t *edges,
                        const double *xi, const int64_t *edge_id, int64_t m,
                        const double *cu, const double *cds, double *gX, double *gU,
                        void *stream);

/* ------------------------------------------------------------------ QUAD4-iso (extension)
 * The reference has no isoparametric quadrilateral; these follow the SURVEY section 8a
 * spec with the reference's conventions (J[i][j] = d x_i/d xi_j, dN_dx = Jinv*D_N as
 * src/models.py:339-351, abs(detJ) as src/loss.py:84): reference square [-1,1]^2, local nodes CCW from
 * (-1,-1), 2x2 Gauss (+-1/sqrt(3), weights 1), zero body force.  conn4 [Ne][4] int32.
 * Planless (fp64 global atomics): ACCUMULATES loss_acc[0], gX, gU (caller zeroes); gX/gU may be NULL. */
int hfem_quad4_energy_atomic(int device, const double *X, const double *U, const int32_t *conn4,
                             int64_t e_begin, int64_t e_end, int64_t nn, const double mat[4],
                             double *loss_acc, double *gX, double *gU, void *stream);
/* Tiled QUAD4 (plan from hfem_plan_create_ex(..., 4, ...)): same contract as hfem_tri3_energy_plan --
 * loss_out / gx_free / gu_free rows of the tile range OVERWRITTEN, free/fixed row maps fused.      */
int hfem_quad4_energy_plan(hfem_plan *plan, const double *x_free, const double *x_fixed,
                           const double *u_free, const double *u_fixed, const double mat[4],
                           const double *T_edge, const double Tconst[4], int32_t tile_begin,
                           int32_t tile_end, double *loss_out, double *gx_free, double *gu_free,
                           int32_t flags, void *stream);
/* Same with a body force: Bq [4][2] = b at the 2x2 Gauss points in REFERENCE coordinates (the reference's triangle
 * path hands b_force the reference points, src/loss.py:60,80 -- SURVEY F6), order (-,-) (+,-) (-,+) (+,+);
 * e -= sum_q |detJ_q| u_h(xi_q).b_q.  NULL / all zero = hfem_quad4_energy_plan.                        */
int hfem_quad4_energy_plan_body(hfem_plan *plan, const double *x_free, const double *x_fixed,
                                const double *u_free, const double *u_fixed, const double mat[4],
                                const double Bq[8], const double *T_edge, const double Tconst[4],
                                int32_t tile_begin, int32_t tile_end, double *loss_out, double *gx_free,
                                double *gu_free, int32_t flags, void *stream);
/* General form.  dtype: 0 = fp64 rows, 1 = fp32 rows (x / u / fixed rows and the gradient rows are float [rows][2]: an
 * fp32 model, the reference's default dtype src/loss.py:16, without widening copies -- widened on load, rounded once on
 * store, arithmetic and loss_out fp64).  flags additionally accept HFEM_FLAG_PHYSICAL_GRAD (grad_u = G Jinv instead of the
 * reference convention's G Jinv^T) and, for fp64 rows on the whole plan, HFEM_FLAG_DETERMINISTIC (node-centric fixed-order
 * kernel: bit-identical run to run, several times slower).  Bq as hfem_quad4_energy_plan_body (NULL: none).           */
int hfem_quad4_energy_plan_ex(hfem_plan *plan, int32_t dtype, const void *x_free, const void *x_fixed,
                              const void *u_free, const void *u_fixed, const double mat[4], const double Bq[8],
                              const double *T_edge, const double Tconst[4], int32_t tile_begin, int32_t tile_end,
                              double *loss_out, void *gx_free, void *gu_free, int32_t flags, void *stream);
/* Per-point forward/backward with the (x_ref, element_id) contract of src/models.py:316:
 * x_eval [M][2] in [-1,1]^2 -> u_h [M][2], detJ [M], grad_u [M][2][2]; backward ACCUMULATES gX,gU. */
int hfem_quad4_eval_fwd(int device, const double *X, const double *U, const int32_t *conn4,
                        const double *x_eval, const int64_t *elem_id, int64_t m,
                        double *u_h, double *detJ, double *grad_u, void *stream);
int hfem_quad4_eval_bwd(int device, const double *X, const double *U, const int32_t *conn4,
                        const double *x_eval, const int64_t *elem_id, int64_t m,
                        const double *cu, const double *cd, const double *cg,
                        double *gX, double *gU, void *stream);

/* ------------------------------------------------------------------ fused optimiser step
 * One launch of torch.optim.Adam's update (examples/example1.py:31, example2.py:37, example3.py:89;
 * default betas/eps, no weight decay, no amsgrad) on a flat tensor of n parameters:
 * m += (g-m)(1-b1); v = v b2 + (1-b2) g g; p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps).
 * dtype: 0 = fp64, 1 = fp32 (p, g, m, v all of that type).  step counts from 1.            */
int hfem_adam_step(int device, void *p, const void *g, void *m, void *v, int64_t n, int32_t dtype,
                   double lr, double beta1, double beta2, double eps, int64_t step, void *stream);
/* hipGraph-capturable form: the (1-based) step count is read from device memory when the kernel runs;
 * hfem_counter_add bumps such a counter in stream order (call it once before the step's kernels).   */
int hfem_adam_step_dev(int device, void *p, const void *g, void *m, void *v, int64_t n, int32_t dtype,
                       double lr, double beta1, double beta2, double eps, const int64_t *step_dev,
                       void *stream);
int hfem_counter_add(int device, int64_t *counter, int64_t inc, void *stream);
/* Multi-tensor form: EVERY parameter tensor of the optimiser in ONE launch (the reference's models have 1-3 parameter
 * tensors, examples/example4.py:54-64; per-tensor launches cost a kernel boundary each).  table_dev[n_tensors] lives in
 * DEVICE memory; tensor t is updated by blocks [block_begin_t, block_begin_{t+1}) of 256 threads, block b of a tensor taking
 * the 16-byte vectors [b * chunk_vecs, (b + 1) * chunk_vecs); n_blocks = total.  dtype per tensor: 0 fp64, 1 fp32.
 * Step count: step_dev (may be NULL) holds the number of COMPLETED steps; the update uses step_dev[0] + step_offset
 * (step_offset alone when step_dev is NULL: a host-side count).  With step_dev the kernel itself bumps the counter once
 * all blocks have read it (ticket_dev: HFEM_ADAM_TICKET_INTS int32 of device memory, zero before the first call; a
 * two-level last-arriver count) -- no hfem_counter_add.  Keep n_blocks at a few per CU (<= 2048): every block ends with
 * one device-scope atomic.                                                                                            */
#define HFEM_ADAM_TICKET_INTS 1056
typedef struct hfem_adam_tensor {
    void *p; const void *g; void *m; void *v;
    int64_t n;
    double lr, beta1, beta2, eps;
    int32_t dtype, block_begin;
} hfem_adam_tensor;
int hfem_adam_multi_dev(int device, const hfem_adam_tensor *table_dev, int32_t n_tensors, int32_t n_blocks,
                        int64_t chunk_vecs, int64_t *step_dev, int64_t step_offset, int32_t *ticket_dev, void *stream);

/* ------------------------------------------------------------------ row gather/scatter
 * src/models.py:292-305 as index lists instead of bool-mask index_put (which
 * runs aten::nonzero on every call): dst[idx[r]][0..w) = src[r][0..w).         */
int hfem_scatter_rows(int device, const double *src, const int32_t *idx, int64_t rows,
                      int32_t width, double *dst, void *stream);
int hfem_gather_rows(int device, const double *src, const int32_t *idx, int64_t rows,
                     int32_t width, double *dst, void *stream);

/* ------------------------------------------------------------------ L-BFGS on the flat parameter vector
 * (torch.optim.LBFGS as examples/example4.py:68-78 uses it: fixed step, no line search).
 * State (history ring of `history` (s, y) pairs, Gram blocks, direction d, step t) lives on the device.
 * One inner iteration of torch's loop is:
 *     hfem_lbfgs_direction(g)  ->  hfem_lbfgs_apply(param, offset, numel) per parameter tensor
 *     ->  closure (energy kernel) -> hfem_lbfgs_check(g_new, loss_new, after_update = 1) -> host reads status.
 * dtype: 0 fp64, 1 fp32 (vectors); dots and the recursion are fp64 either way.
 * status[8] (host): loss, flags (bit0 max|g| <= tol_grad, bit1 max|t d| <= tol_change,
 * bit2 |loss - prev_loss| < tol_change, bit3 g.d > -tol_change: nothing was applied), max|g|, g.d, t,
 * history count, n_iter, H_diag.  hfem_lbfgs_check synchronises the stream; the others only enqueue.      */
typedef struct hfem_lbfgs hfem_lbfgs;
int hfem_lbfgs_create(int device, int64_t n, int32_t history, int32_t dtype, hfem_lbfgs **out);
int hfem_lbfgs_destroy(hfem_lbfgs *opt);
int hfem_lbfgs_check(hfem_lbfgs *opt, const void *g, const double *loss, int32_t after_update, double tol_grad,
                     double tol_change, double *status_host, void *stream);
int hfem_lbfgs_direction(hfem_lbfgs *opt, const void *g, double lr, double tol_change, void *stream);
int hfem_lbfgs_apply(hfem_lbfgs *opt, void *p, int64_t offset, int64_t numel, void *stream);
void *hfem_lbfgs_direction_ptr(hfem_lbfgs *opt);
/* NODE-SHARDED L-BFGS (hidenn_fem_amd/optim.py ShardedLBFGS; no reference counterpart -- the reference runs example 4 on one
 * device).  The optimiser object holds the history, gradient copy and direction of the parameter rows ONE RANK owns (n = 2 x
 * its owned rows); per inner iteration every rank
 *     hfem_lbfgs_shard_gather (its gradient rows -> flat local vector)
 *     hfem_lbfgs_shard_local  (speculative pair + ONE pass over its history -> a payload of
 *                              hfem_lbfgs_shard_payload_doubles() doubles: per-slot dots, y.s, y.y, gradient statistics,
 *                              max|d|, its partial energy)
 *     [the caller gathers the payloads of all ranks to every rank, rank order: one small all_gather / peer-window put]
 *     hfem_lbfgs_shard_finish (sums in rank order -> torch's break tests -> memory update, recursion, its part of d; status)
 *     hfem_lbfgs_shard_apply  (x[rows], u[rows] += t d)
 * so the two passes over the history shrink by the number of ranks and every rank takes bit-identical decisions.  With
 * world = 1 (gathered = the payload itself) the flow is torch.optim.LBFGS / hfem_lbfgs_direction on one device.           */
int64_t hfem_lbfgs_shard_payload_doubles(const hfem_lbfgs *opt);
int hfem_lbfgs_shard_gather(hfem_lbfgs *opt, const void *gx, const int32_t *rows_x, int64_t nx, const void *gu,
                            const int32_t *rows_u, int64_t nu, void *out, void *stream);
int hfem_lbfgs_shard_local(hfem_lbfgs *opt, const void *g, const double *loss_local_dev, double *payload_dev, void *stream);
int hfem_lbfgs_shard_finish(hfem_lbfgs *opt, const void *g, const double *gathered_dev, int32_t world, int32_t after_update,
                            int32_t want_direction, double lr, double tol_grad, double tol_change, double *status_host,
                            void *stream);
/* hfem_lbfgs_shard_finish with status_host = NULL only enqueues (the record goes to a pinned buffer of the optimiser): a whole
 * steady-state iteration -- apply, energy, gather, local, the exchange, finish -- is then capturable in ONE hipGraph;
 * hfem_lbfgs_shard_status synchronises the stream and hands the record out.                                          */
int hfem_lbfgs_shard_status(hfem_lbfgs *opt, double *status_host, void *stream);
int hfem_lbfgs_shard_apply(hfem_lbfgs *opt, void *x, const int32_t *rows_x, int64_t nx, void *u, const int32_t *rows_u,
                           int64_t nu, void *stream);

/* ------------------------------------------------------------------ post-processing
 * hfem_tri3_von_mises: per element, grad_u at the centroid (constant on a P1 triangle; reference
 * convention, src/models.py:351-355) -> strain -> plane-stress stress -> von Mises, exactly the chain of
 * src/plots.py:183-198 (sigma_xy = E/(1+nu) eps_xy).  von_mises[Ne]; grad_u[Ne][2][2] optional (NULL).
 * hfem_line2_slopes: out[i][c] = (u[i+1][c] - u[i][c]) / (grid[i+1] - grid[i]), i < n_nodes - 1: the
 * per-element du/dx that src/plots.py:5-27 obtains with one autograd call per element.               */
int hfem_tri3_von_mises(int device, const double *X, const double *U, const int32_t *conn, int64_t ne,
                        double E, double nu, double *von_mises, double *grad_u, void *stream);
int hfem_line2_slopes(int device, const double *grid, const double *u, int64_t n_nodes, int32_t dim_u,
                      double *out, void *stream);

/* ------------------------------------------------------------------ multi-GPU interface exchange
 * Owner-sharded mode (no reference counterpart): one all_gather per step of a fixed-size
 * payload per rank, in double2 units:  [x rows | u rows | padding][loss partial, 0].
 * hfem_iface_pack: out[i] = x_free[rows[i]] (i < n_x), u_free[rows[i]] (n_x <= i < n_x + n_u).
 * hfem_iface_unpack: x_free[dst[i]] = recv[src[i]] (i < n_x), u_free[dst[i]] = recv[src[i]] (next n_u);
 * src indexes the gathered buffer (world payloads of `stride` double2 each); loss_out (may be NULL)
 * = sum over ranks, in rank order, of recv[r * stride + loss_slot].x.                               */
int hfem_iface_pack(int device, const double *x_free, const double *u_free, const int32_t *rows,
                    int32_t n_x, int32_t n_u, double *out, void *stream);
int hfem_iface_unpack(int device, const double *recv, const int32_t *src, const int32_t *dst, int32_t n_x,
                      int32_t n_u, double *x_free, double *u_free, int32_t world, int64_t stride,
                      int64_t loss_slot, double *loss_out, void *stream);

/* hfem_iface_pack + the rank's energy + the optimiser's step counter in ONE launch (the owner-sharded training step is a
 * chain of small dependent launches, each worth a kernel boundary): out as hfem_iface_pack; out[loss_slot] = {sum, in tile
 * order, of the tile energies that the HFEM_FLAG_NO_LOSS_SUM launch(es) over [tile_begin, tile_end) left in the plan, 0};
 * counter (may be NULL) += 1; bc_next (may be NULL; needs counter) = {1 - beta1^(c + 1), sqrt(1 - beta2^(c + 1))} with c the
 * bumped count: the bias corrections of the NEXT step, for its fused energy + Adam launch (no hfem_adam_prep launch).
 * loss_slot >= n_x + n_u, in double2 units.                                                                       */
int hfem_plan_iface_pack(hfem_plan *plan, int32_t tile_begin, int32_t tile_end, const double *x_free,
                         const double *u_free, const int32_t *rows, int32_t n_x, int32_t n_u, double *out,
                         int64_t loss_slot, int64_t *counter, double beta1, double beta2, double *bc_next, void *stream);
/* The exchange for fp32 models (the reference's default dtype; `_f32` as for the energy entry points): parameter rows are
 * float2, the payload stays double2 -- widened on the way out, rounded back (losslessly) on the way in.               */
int hfem_plan_iface_pack_f32(hfem_plan *plan, int32_t tile_begin, int32_t tile_end, const float *x_free,
                             const float *u_free, const int32_t *rows, int32_t n_x, int32_t n_u, double *out,
                             int64_t loss_slot, int64_t *counter, double beta1, double beta2, double *bc_next, void *stream);
int hfem_iface_unpack_f32(int device, const double *recv, const int32_t *src, const int32_t *dst, int32_t n_x,
                          int32_t n_u, float *x_free, float *u_free, int32_t world, int64_t stride,
                          int64_t loss_slot, double *loss_out, void *stream);

/* Peer-window exchange (csrc/peer.hip): the interface payload of hfem_plan_iface_pack written BY THE PACK KERNEL into a
 * receive window on every rank -- stores over xGMI into IPC-mapped device memory -- with arrival flags, instead of an
 * all_gather: no collective, no second stream, capturable.  One process per GPU (at most 16 ranks).  Set-up, once:
 *   hfem_peer_create(device, rank, world, stride, &peer)   stride = interface rows + 1 (double2 units, as the all_gather
 *                                                          payload); allocates and zeroes this rank's window
 *   hfem_peer_ipc_handle(peer, h64)                        64-byte hipIpcMemHandle_t of the window; exchange the handles of
 *                                                          all ranks by any side channel (torch.distributed, a file, MPI)
 *   hfem_peer_connect(peer, handles[world][64])            maps the other ranks' windows (own entry ignored); world == 1
 *                                                          needs neither call.  Barrier before the first put.
 * Per step, stream-ordered, STRICTLY alternating on every rank:
 *   hfem_plan_iface_put(...)   = hfem_plan_iface_pack with out = slot (puts so far) & 1 of my lane in every window, then a
 *                                system-scope fence and flag = puts + 1 in every window
 *   hfem_peer_iface_get(...)   = waits until every rank's flag of that slot has arrived in MY window, then
 *                                hfem_iface_unpack from it (src indices as for the gathered [world][stride] payload).
 * The wait is bounded: after timeout_ticks (100 MHz: 1e8 = 1 s) without a flag the kernel sets the sticky status bit 1 and
 * goes on (later gets no longer wait) -- a lost peer is an error the host reads with hfem_peer_status (which synchronises
 * with the device), never a hung GPU.  hfem_peer_status: status_out = sticky bits, puts_out = completed puts (either may
 * be NULL).                                                                                                       */
typedef struct hfem_peer hfem_peer;
int hfem_peer_create(int device, int32_t rank, int32_t world, int64_t stride, hfem_peer **out);
int hfem_peer_ipc_handle(hfem_peer *peer, void *handle_out_64_bytes);
int hfem_peer_connect(hfem_peer *peer, const void *handles_world_x_64_bytes);
int hfem_peer_destroy(hfem_peer *peer);
int hfem_peer_status(hfem_peer *peer, int32_t *status_out, int64_t *puts_out);
/* The get INSIDE the next energy launch (one launch less per step, and no tile that does not need foreign rows waits):
 * hfem_peer_attach_get stores hfem_peer_iface_get's tables / loss slot / timeout in device memory, hfem_plan_set_peer_get
 * binds them to a plan together with the tile range [wait_begin, wait_end) that must wait (the rank's boundary tiles:
 * hfem_plan_export id 10), and hfem_tri3_energy_plan / hfem_tri3_energy_adam_step_ex with HFEM_FLAG_PEER_GET then run the
 * get as their first workgroups.  Status bit 2: a boundary tile gave up waiting (after 2 x timeout_ticks).           */
int hfem_peer_attach_get(hfem_peer *peer, const int32_t *src, const int32_t *dst, int32_t n_x, int32_t n_u,
                         int64_t loss_slot, double *loss_out, int64_t timeout_ticks);
int hfem_plan_set_peer_get(hfem_plan *plan, hfem_peer *peer, int32_t wait_begin, int32_t wait_end);
/* The put INSIDE the fused energy + Adam launch (round 4: ONE launch per owner-sharded training step).  hfem_peer_attach_put
 * stores in device memory where every parameter row sits in this rank's payload lane (pos_x / pos_u: per row of x_free /
 * u_free the double2 index, -1 = not an interface row), the loss slot and the optimiser's step counter + betas;
 * hfem_plan_set_peer_put binds them to a paired-slot plan with the TWO bias-correction buffers the steps alternate between.
 * hfem_tri3_energy_adam_step_ex(..., HFEM_FLAG_PEER_GET | HFEM_FLAG_PEER_PUT | HFEM_FLAG_NO_LOSS_SUM) then: the boundary tiles
 * [wait_begin, wait_end) of hfem_plan_set_peer_get store the NEW rows of their interface nodes into every rank's window at
 * write-out; the last of them to finish writes the rank's energy of the PREVIOUS evaluation into the loss slot (this one's is
 * not complete yet: the global energy a get delivers therefore lags one step more; hfem_plan_iface_put flushes the last),
 * bumps the step counter, writes the next step's bias corrections into the other buffer and raises the flags.  Protocol,
 * slots and bounded waits as hfem_plan_iface_put / the in-launch get.                                                */
int hfem_peer_attach_put(hfem_peer *peer, const int32_t *pos_x, const int32_t *pos_u, int64_t loss_slot, int64_t *counter,
                         double beta1, double beta2);
int hfem_plan_set_peer_put(hfem_plan *plan, hfem_peer *peer, double *bc_a, double *bc_b);
int hfem_plan_iface_put(hfem_plan *plan, hfem_peer *peer, int32_t tile_begin, int32_t tile_end, const double *x_free,
                        const double *u_free, const int32_t *rows, int32_t n_x, int32_t n_u, int64_t loss_slot,
                        int64_t *counter, double beta1, double beta2, double *bc_next, void *stream);
int hfem_peer_iface_get(hfem_peer *peer, const int32_t *src, const int32_t *dst, int32_t n_x, int32_t n_u, double *x_free,
                        double *u_free, int64_t loss_slot, double *loss_out, int64_t timeout_ticks, void *stream);
int hfem_plan_iface_put_f32(hfem_plan *plan, hfem_peer *peer, int32_t tile_begin, int32_t tile_end, const float *x_free,
                            const float *u_free, const int32_t *rows, int32_t n_x, int32_t n_u, int64_t loss_slot,
                            int64_t *counter, double beta1, double beta2, double *bc_next, void *stream);
int hfem_peer_iface_get_f32(hfem_peer *peer, const int32_t *src, const int32_t *dst, int32_t n_x, int32_t n_u, float *x_free,
                            float *u_free, int64_t loss_slot, double *loss_out, int64_t timeout_ticks, void *stream);

/* In-library collectives: one RCCL communicator per rank (one process per GPU).  Rank 0 calls
 * hfem_mg_unique_id and broadcasts the 128 bytes by any side channel (torch.distributed, a file, MPI); every rank
 * then calls hfem_mg_comm_create.  The collectives only ENQUEUE on the caller's stream -- in stream order right
 * after the energy kernel -- so a whole multi-GPU step can be captured into one hipGraph.  fp64, sum.
 * hfem_mg_allgather: recv = [rank 0's `count` doubles | rank 1's | ...].  send == recv + rank * count is allowed.
 * RCCL is bound at run time (dlopen, in this order: the path given to hfem_mg_load, $HFEM_RCCL_PATH, a librccl already
 * mapped into the process, the ROCm installation); without it these calls return an error, the rest works.   */
typedef struct hfem_mg_comm hfem_mg_comm;
int hfem_mg_load(const char *librccl_path);
int hfem_mg_unique_id(void *id_out_128_bytes);
int hfem_mg_comm_create(int device, int rank, int world, const void *id_128_bytes, hfem_mg_comm **out);
int hfem_mg_comm_destroy(hfem_mg_comm *comm);
int hfem_mg_allreduce_sum(hfem_mg_comm *comm, const double *send, double *recv, int64_t count, void *stream);
int hfem_mg_allgather(hfem_mg_comm *comm, const double *send, double *recv, int64_t count, void *stream);
/* Adam on the ROWS a rank owns (owner-sharded mode: a rank updates exactly the parameter rows its tiles own):
 * hfem_adam_step_dev restricted to rows[n_rows] of [.][2] fp64 arrays p, g, m, v.                          */
int hfem_adam_step_rows_dev(int device, double *p, const double *g, double *m, double *v, const int32_t *rows,
                            int64_t n_rows, double lr, double beta1, double beta2, double eps,
                            const int64_t *step_dev, void *stream);

/* Both parameter tensors of the triangular model in one launch: rows_x[n_x] of (px, gx, mx, vx) with lr_x and rows_u[n_u]
 * of (pu, gu, mu, vu) with lr_u.  The (1-based) step of the bias corrections is step_dev[0] + step_offset: 0 when the
 * counter was bumped before (hfem_counter_add), 1 when it is bumped after (hfem_plan_iface_pack).                */
int hfem_adam_step_rows2_dev(int device, double *px, const double *gx, double *mx, double *vx, const int32_t *rows_x,
                             int64_t n_x, double lr_x, double *pu, const double *gu, double *mu, double *vu,
                             const int32_t *rows_u, int64_t n_u, double lr_u, double beta1, double beta2, double eps,
                             const int64_t *step_dev, int64_t step_offset, void *stream);
/* The same on float rows (an fp32 model): torch's fp32 arithmetic. */
int hfem_adam_step_rows2_dev_f32(int device, float *px, const float *gx, float *mx, float *vx, const int32_t *rows_x,
                                 int64_t n_x, double lr_x, float *pu, const float *gu, float *mu, float *vu,
                                 const int32_t *rows_u, int64_t n_u, double lr_u, double beta1, double beta2, double eps,
                                 const int64_t *step_dev, int64_t step_offset, void *stream);

/* ------------------------------------------------------------------ 1D / structured
 * Grid parametrisation softplus -> clamp(1e-6) -> cumsum -> renormalise
 * (src/models.py:45-56, 146-168): p[n] -> grid[n+1]; backward ggrid[n+1] -> gp[n].
 * mask (uint8[n+1], may be NULL) + initial[n+1]: grid = where(mask, initial, grid)
 * (src/models.py:165-166); masked entries get no gradient.                     */
int hfem_grid_param_fwd(int device, const double *p, int64_t n, double x0, double xN,
                        const uint8_t *mask, const double *initial, double *grid,
                        void *stream);
int hfem_grid_param_bwd(int device, const double *p, int64_t n, double x0, double xN,
                        const uint8_t *mask, const double *ggrid, double *gp, void *stream);
/* Same for long grids, over all CUs (three small launches each way): ws = hfem_grid_param_ws_elems(n) doubles
 * of scratch; cum[n] is written by the forward (running sums of the clamped softplus, cum[n-1] = total) and
 * read by the backward.                                                                              */
int64_t hfem_grid_param_ws_elems(int64_t n);
int hfem_grid_param_fwd_ws(int device, const double *p, int64_t n, double x0, double xN, const uint8_t *mask,
                           const double *initial, double *grid, double *cum, double *ws, void *stream);
int hfem_grid_param_bwd_ws(int device, const double *p, int64_t n, double x0, double xN, const uint8_t *mask,
                           const double *ggrid, const double *cum, double *gp, double *ws, void *stream);

/* LINE2 hat-function interpolation, src/models.py:70-90: grid[n], u[n] full
 * arrays, x_eval[m] physical points -> pred[m] and dudx[m] = (u_{e+1}-u_e)/h of the
 * element e = clamp(searchsorted(grid,x)-1, 0, n-2) (either output may be NULL).
 * dudx is the explicit, differentiable replacement for the reference's
 * autograd.grad(u, xq, create_graph=True) (examples/example3.py:56).           */
int hfem_line2_eval_fwd(int device, const double *grid, const double *u, int64_t n,
                        const double *x_eval, int64_t m, double *pred, double *dudx,
                        void *stream);
/* Backward: cot[m] (d/dpred), cot_dudx[m] (d/d dudx, may be NULL) -> ACCUMULATES
 * ggrid[n], gu[n]; writes gx_eval[m] (may be NULL).                            */
int hfem_line2_eval_bwd(int device, const double *grid, const double *u, int64_t n,
                        const double *x_eval, int64_t m, const double *cot,
                        const double *cot_dudx, double *ggrid, double *gu, double *gx_eval,
                        void *stream);
/* Fused 1D bar energy of examples/example3.py:27-70 with detached quadrature
 * (SURVEY F8): xq,wq,bq [npts] constants.  ONE launch; loss_acc[0], ggrid[n],
 * gu[n] are all ACCUMULATED (caller zeroes; ggrid/gu may be NULL).              */
int hfem_bar_energy(int device, const double *grid, const double *u, int64_t n,
                    const double *xq, const double *wq, const double *bq, int64_t npts,
                    double E, double *loss_acc, double *ggrid, double *gu, void *stream);
/* Fused L2-projection loss mean((pred-target)^2) of examples/example1.py:38 with
 * its backward in one launch: loss_acc, ggrid, gu ACCUMULATED (caller zeroes).  */
int hfem_line2_mse(int device, const double *grid, const double *u, int64_t n,
                   const double *x_eval, const double *target, int64_t m, double *loss_acc,
                   double *ggrid, double *gu, void *stream);

/* RECT-Q4 bilinear interpolation on the tensor-product grid, src/models.py:180-212:
 * gx[nx], gy[ny], u[nx][ny] (u_full, i.e. after where(node_mask,u_fixed,u)),
 * x_eval [m][2] physical -> pred[m].                                           */
int hfem_rectq4_eval_fwd(int device, const double *gx, int64_t nx, const double *gy,
                         int64_t ny, const double *u, const double *x_eval, int64_t m,
                         double *pred, void *stream);
/* Backward: cot[m] -> ACCUMULATES ggx[nx], ggy[ny], gu[nx][ny]; writes
 * gx_eval[m][2] (may be NULL).                                                 */
int hfem_rectq4_eval_bwd(int device, const double *gx, int64_t nx, const double *gy,
                         int64_t ny, const double *u, const double *x_eval, int64_t m,
                         const double *cot, double *ggx, double *ggy, double *gu,
                         double *gx_eval, void *stream);
/* Fused 2D L2-projection loss mean((pred-target)^2) of examples/example2.py:45-46
 * with its backward in one launch: loss_acc, ggx, ggy, gu ACCUMULATED.         */
int hfem_rectq4_mse(int device, const double *gx, int64_t nx, const double *gy, int64_t ny,
                    const double *u, const double *x_eval, const double *target, int64_t m,
                    double *loss_acc, double *ggx, double *ggy, double *gu, void *stream);

/* Float-row twins of the 1D / structured entry points above, for models in the reference's default dtype
 * (torch.float32: src/models.py:36-40, 142; examples 1-3 as shipped): every array argument is float -- parameters,
 * points, targets, per-point outputs, gradient accumulators and the loss scalar.  Inputs are widened on load, the
 * arithmetic in between is fp64, per-point outputs are rounded once on store.  The ACCUMULATED outputs (ggrid, gu, ggx,
 * ggy, loss_acc) are float atomics: every contribution is rounded when it is added and the order varies -- the last
 * bits are not reproducible (unlike hfem_tri3_energy_plan_f32, whose gradient rows are summed in fp64 and rounded once).
 * No widening copies on either side of the call.  The scratch of the *_ws forms (cum, ws) stays fp64.           */
int hfem_grid_param_fwd_f32(int device, const float *p, int64_t n, double x0, double xN, const uint8_t *mask,
                            const float *initial, float *grid, void *stream);
int hfem_grid_param_bwd_f32(int device, const float *p, int64_t n, double x0, double xN, const uint8_t *mask,
                            const float *ggrid, float *gp, void *stream);
int hfem_grid_param_fwd_ws_f32(int device, const float *p, int64_t n, double x0, double xN, const uint8_t *mask,
                               const float *initial, float *grid, double *cum, double *ws, void *stream);
int hfem_grid_param_bwd_ws_f32(int device, const float *p, int64_t n, double x0, double xN, const uint8_t *mask,
                               const float *ggrid, const double *cum, float *gp, double *ws, void *stream);
int hfem_line2_eval_fwd_f32(int device, const float *grid, const float *u, int64_t n, const float *x_eval, int64_t m,
                            float *pred, float *dudx, void *stream);
int hfem_line2_eval_bwd_f32(int device, const float *grid, const float *u, int64_t n, const float *x_eval, int64_t m,
                            const float *cot, const float *cot_dudx, float *ggrid, float *gu, float *gx_eval, void *stream);
int hfem_bar_energy_f32(int device, const float *grid, const float *u, int64_t n, const float *xq, const float *wq,
                        const float *bq, int64_t npts, double E, float *loss_acc, float *ggrid, float *gu, void *stream);
int hfem_line2_mse_f32(int device, const float *grid, const float *u, int64_t n, const float *x_eval, const float *target,
                       int64_t m, float *loss_acc, float *ggrid, float *gu, void *stream);
int hfem_rectq4_eval_fwd_f32(int device, const float *gx, int64_t nx, const float *gy, int64_t ny, const float *u,
                             const float *x_eval, int64_t m, float *pred, void *stream);
int hfem_rectq4_eval_bwd_f32(int device, const float *gx, int64_t nx, const float *gy, int64_t ny, const float *u,
                             const float *x_eval, int64_t m, const float *cot, float *ggx, float *ggy, float *gu,
                             float *gx_eval, void *stream);
int hfem_rectq4_mse_f32(int device, const float *gx, int64_t nx, const float *gy, int64_t ny, const float *u,
                        const float *x_eval, const float *target, int64_t m, float *loss_acc, float *ggx, float *ggy,
                        float *gu, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HIDENN_FEM_H */
