/*
 * raht.h -- C ABI of the MI355X-native RAHT attribute codec (libraht_hip.so, gfx950).
 *
 * Drop-in boundary: the three-entry operator table every reference driver dispatches through,
 *     raht_fn = {"RAHT", "iRAHT", "RAHT_param"}      (reference python/encode_3dgs.py:23-27,
 *                                                      encode_ply.py:20-24, encode_dataset.py:20-24)
 * plus the voxelizer that produces its input (python/voxelize_pc.py:62-172) and the driver-inline
 * quantize / reorder arithmetic (python/encode_3dgs.py:204-217, 261-268).
 *
 * Conventions
 *   - every function returns an int status: 0 = RAHT_OK, negative = error; raht_last_error()
 *     returns a thread-local, human-readable description of the last failure;
 *   - no C++ exceptions, no torch types: plain pointers and sizes only;
 *   - all data pointers are DEVICE pointers (HBM) unless a parameter says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream). Transform entry points
 *     (raht_fwd*, raht_inv*, raht_quant*, raht_dequant*) only enqueue kernels on `stream`: no
 *     host synchronisation and, after the first call for a given (element type, D) or after
 *     raht_plan_prepare, no allocation: safe to capture in a hipGraph. Plan construction and
 *     raht_voxelize allocate and synchronise `stream` (they size outputs from device counts);
 *   - the caller owns every data buffer; a plan is an opaque handle owning its own HBM;
 *   - rows of C / T are the points in Morton order, row-major, `ld*` = row stride in ELEMENTS;
 *   - N < 2^31 rows; element offsets are 64-bit;
 *   - devices and threads: every call works on the calling thread's current HIP device, which must be the
 *     device the plan and the buffers live on (one process per GPU is the intended use, but a process may
 *     hold plans on several devices: a plan remembers its device, all cached device memory is kept per
 *     device, and an entry point called with a plan while ANOTHER device is current returns
 *     RAHT_ERR_INVALID instead of mixing memory of two GPUs; raht_plan_destroy alone may be called with any
 *     device current). Different plans
 *     may be used from different host threads at once; ONE plan runs one transform at a time (it owns its
 *     workspaces), i.e. calls on the same plan must be ordered on one stream or serialised by the caller.
 *     Temporary device blocks are pooled per host thread and reused in call order: a host thread that
 *     switches streams between calls must order those streams itself (event or synchronise).
 */
#ifndef RAHT_H
#define RAHT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAHT_VERSION 300

enum raht_status {
    RAHT_OK = 0,
    RAHT_ERR_INVALID = -1,      /* bad argument (NULL, N < 1, depth out of range, D < 1 ...)          */
    RAHT_ERR_UNSORTED = -2,     /* Morton keys not strictly increasing (unsorted or duplicate voxels):
                                   the reference silently mis-pairs these (SURVEY 8a iii/iv)          */
    RAHT_ERR_BOUNDS = -3,       /* a coordinate is < 0 or >= 2^depth (reference RAHT_param.py:26-27
                                   raises ValueError in the slow path, the fast path does not check) */
    RAHT_ERR_HIP = -4,          /* a HIP runtime call failed                                          */
    RAHT_ERR_NOMEM = -5,
    RAHT_ERR_UNSUPPORTED = -6
};

enum raht_dtype { RAHT_F32 = 0, RAHT_F64 = 1, RAHT_I32 = 2, RAHT_I64 = 3 };

/* Which kernel family executes a transform.
 *   RAHT_ENGINE_TILE  : LDS-tile-fused stages (default): a block stages a run of Morton-contiguous
 *                       rows in LDS and performs every butterfly whose subtree lies inside the run;
 *                       the few surviving low-pass rows are merged by later, much smaller stages.
 *   RAHT_ENGINE_LEVEL : one launch per binary level (= per octree level per axis), rows gathered
 *                       from HBM pairwise. Kept as the simple formulation and as a cross-check. */
enum raht_engine { RAHT_ENGINE_TILE = 0, RAHT_ENGINE_LEVEL = 1 };

typedef struct raht_plan raht_plan;
typedef void *raht_stream_t;

const char *raht_last_error(void);
int raht_version(void);

/* ------------------------------------------------------------------------------------------------
 * Plan ("RAHT_param").  Replaces RAHT_param_reorder_fast(V, minV, width, depth)
 * (reference python/RAHT_param.py:190-279).
 *
 * V: N x 3 (x, y, z) coordinates holding integers, already Morton-sorted and duplicate-free, in
 * v_dtype. Vint = floor((V - minV) / (width / 2^depth)); key = sum_k (z_k + 2 y_k + 4 x_k) << 3k.
 * Instead of the reference's per-level List / Flags / weights the plan stores three arrays of
 * length N derived from the sorted keys:
 *     lvl[i] = msb(key[i] ^ key[i-1])    the one binary level at which row i is a right sibling
 *     wl[i]  = i - i0                    weight (leaf count) of its left sibling, i0 = partner row
 *     wr[i]  = nxt(i) - i                weight of the subtree rooted at row i
 * plus order_RAGFT (coarse-to-fine coefficient permutation, RAHT_param.py:251-274).
 * Errors: RAHT_ERR_UNSORTED, RAHT_ERR_BOUNDS (both checked on device; the reference checks neither).
 * N == 1 yields order = [0] (the reference returns None and its drivers crash).
 */
int raht_plan_create(const void *V, int v_dtype, int64_t N, const double minV[3], double width,
                     int depth, raht_stream_t stream, raht_plan **out);

/* Same, from strictly increasing Morton keys (device, uint64) of `nbits` significant bits (<= 63).
 * leaf_weights (device int64[N], may be NULL = all ones) gives every row an occupancy weight: used
 * when rows are themselves roots of already-transformed subtrees (multi-GPU top levels). */
int raht_plan_create_from_keys(const uint64_t *keys_sorted, int64_t N, int nbits,
                               const int64_t *leaf_weights, raht_stream_t stream, raht_plan **out);
/* The same without the 8 N-byte copy of the keys: the plan reads the CALLER's array for as long as it lives (the
 * caller keeps `keys_sorted` allocated and unchanged until raht_plan_destroy). What a per-frame pipeline wants: the
 * voxelizer's sorted key buffer outlives the frame's plan anyway. */
int raht_plan_create_from_keys_borrowed(const uint64_t *keys_sorted, int64_t N, int nbits,
                                        const int64_t *leaf_weights, raht_stream_t stream, raht_plan **out);

/* Truncated trees (Morton-prefix sharded scenes): butterflies at binary levels >= top_level are
 * NOT performed by this plan's transforms; the rows that still carry a low-pass value afterwards
 * ("roots": row 0 and every row whose level is >= top_level, i.e. the first row of every
 * top-level prefix node) are stitched by the caller's top stage. Default top_level = 64 (whole
 * tree, one root = the DC row). Call before the first transform; rebuilds the tile schedules.
 *   raht_plan_roots           : number of roots and (optionally) their rows, ascending, into a
 *                               DEVICE int64[n_roots] buffer;
 *   raht_plan_set_root_buffer : DEVICE buffer of n_roots x D elements of the transform's type
 *                               (row stride D), or NULL. When set, forward transforms write the roots'
 *                               low-pass rows there (and, in the fused-quantization entry, leave
 *                               their Q rows untouched), inverse transforms read the roots from
 *                               there instead of from T / Q. */
int raht_plan_set_top_level(raht_plan *plan, int top_level, raht_stream_t stream);
int raht_plan_roots(const raht_plan *plan, int64_t *n_roots, int64_t *rows_dev, raht_stream_t stream);
int raht_plan_set_root_buffer(raht_plan *plan, void *buf_dev);

/* Row map (small plans only: N <= 8192, they run as ONE launch): plan row i lives in row map[i] of the
 * matrices handed to raht_fwd* / raht_inv* (which then have n_matrix_rows rows; rows outside the map are not
 * touched). map_dev: DEVICE int64[N], copied; NULL removes the map. This is how the replicated top tree of a
 * Morton-prefix sharded scene works IN PLACE on the padded all-gather buffer (world x max_roots rows) instead
 * of on a compacted copy of it. Not available with the fused-quantization entries or with w != NULL. */
int raht_plan_set_row_map(raht_plan *plan, const int64_t *map_dev, int64_t n_matrix_rows, raht_stream_t stream);

int raht_plan_destroy(raht_plan *plan);
int64_t raht_plan_size(const raht_plan *plan);          /* N */
int raht_plan_nbits(const raht_plan *plan);             /* 3 * depth */

/* Engine / tile-size selection (tile_rows = 0 keeps the automatic choice). tile_rows applies to
 * stage 0 (the HBM-heavy launch); raht_plan_set_tail_tile sets the geometry of the later, small
 * stages: rows per tile, channels per chunk (rounded so that every chunk holds at least 16 bytes of
 * channels), and `final_rows`: once at most that many entries are left, ONE launch (the top stage)
 * finishes the tree (0 = automatic: 4096; at most 8192). */
int raht_plan_set_engine(raht_plan *plan, int engine, int tile_rows);
int raht_plan_set_tail_tile(raht_plan *plan, int tail_rows, int tail_channels, int final_rows);
/* Upper bound on the number of stages (= kernel launches per direction) of a tile schedule, 1..64,
 * default 24. A scene whose schedule would need more falls back to the one-launch-per-level engine (same
 * results): with >= 64 rows per tile every stage makes progress, so the bound is what decides. Mainly a
 * testing aid for that fallback; call before the first transform (drops the cached schedules). */
int raht_plan_set_max_stages(raht_plan *plan, int max_stages);

/* Plans, schedules and workspaces take their device memory from a process-wide cache of freed blocks
 * (a codec that builds one plan per frame stops paying hipMalloc / hipFree once the cache is warm;
 * at most RAHT_POOL_MAX_BYTES, default 8 GiB, stay cached). This returns the cached blocks to HIP. */
int raht_release_cached_memory(void);

/* Reference-shaped views, for parity tests and the drivers' DEBUG save_lists
 * (reference python/encode_3dgs.py:165). HOST output buffers.
 *   raht_plan_levels       = len(Flags) of the reference
 *   raht_plan_export_level : List[l] (int64), Flags[l] (uint8 0/1), weights[l] (int64); any of the
 *                            three may be NULL; *n receives the length of level l. */
int raht_plan_levels(const raht_plan *plan);
int raht_plan_export_level(const raht_plan *plan, int level, int64_t *list, uint8_t *flags,
                           int64_t *weights, int64_t *n);
/* order_RAGFT into a DEVICE int64[N] buffer (usable by index_select / argsort as the drivers do,
 * reference python/encode_3dgs.py:210,267). */
int raht_plan_order(const raht_plan *plan, int64_t *order_dev, raht_stream_t stream);
/* Device pointers of the internal arrays (valid until destroy), for inspection / tests. */
int raht_plan_arrays(const raht_plan *plan, const uint64_t **keys, const uint8_t **lvl,
                     const int32_t **wl, const int32_t **wr);
/* Copy one internal array into a caller-provided DEVICE buffer of N elements:
 * which = 0 keys (uint64), 1 lvl (uint8), 2 wl (int32), 3 wr (int32). */
int raht_plan_copy_array(const raht_plan *plan, int which, void *dst_dev, raht_stream_t stream);
/* Tile-engine schedule statistics: number of stages and active rows per stage (host int64[]). */
int raht_plan_stage_stats(raht_plan *plan, int elem_size, int D, int *n_stages, int64_t *rows_per_stage,
                          int max_stages, int *tile_rows);

/* ------------------------------------------------------------------------------------------------
 * Forward transform ("RAHT").  Replaces RAHT2_optimized(C, List, Flags, weights) -> (T, w)
 * (reference python/RAHT.py:252-336): T[i0] = a x0 + b x1, T[i1] = -b x0 + a x1 with
 * a = sqrt(w0/(w0+w1)), b = sqrt(w1/(w0+w1)) (weights are integer leaf counts; a, b are evaluated
 * in float64 and rounded once). C is never modified; T may not alias C. w (N, may be NULL) receives
 * the node weights of RAHT.py:325-328.
 * f32: fp32 arithmetic, tolerance vs the float64 reference is stated in DESIGN.md / tests;
 * f64: float64 arithmetic (the reference's own precision), rtol = atol = 1e-12.
 * Row strides (ld*, in elements) may be anything >= D; strides above 2^18 elements and rows shorter
 * than 16 bytes take the one-launch-per-level engine instead of the LDS-tile engine (same results).
 */
int raht_fwd(const raht_plan *plan, const float *C, int64_t ldc, int D, float *T, int64_t ldt,
             float *w, raht_stream_t stream);
int raht_fwd_f64(const raht_plan *plan, const double *C, int64_t ldc, int D, double *T, int64_t ldt,
                 double *w, raht_stream_t stream);

/* Inverse transform ("iRAHT").  Replaces inverse_RAHT_optimized(T, List, Flags, weights) -> C
 * (reference python/iRAHT.py:40-114): x0 = a T0 - b T1, x1 = b T0 + a T1, levels top-down. */
int raht_inv(const raht_plan *plan, const float *T, int64_t ldt, int D, float *C, int64_t ldc,
             raht_stream_t stream);
int raht_inv_f64(const raht_plan *plan, const double *T, int64_t ldt, int D, double *C, int64_t ldc,
                 raht_stream_t stream);

/* Fused variants for the float32 tile engine: the coefficient matrix T is never materialised.
 *   raht_fwd_quant   = raht_fwd + raht_quant_reorder      (encode_3dgs.py:159,204,210,215)
 *   raht_dequant_inv = raht_dequant_unreorder + raht_inv  (encode_3dgs.py:261,267-268,274)
 * Same results as the two-call sequences (identical float32 arithmetic).
 *
 * PRECISION OF THE float32 INTEGERS (against the reference's float64 floor(T / step + 0.5)). A float32 coefficient carries
 * ~1e-7 of its own magnitude in error, so an integer differs from the reference's when a rounding boundary falls inside
 * |dT| / step: on unit-range attribute channels that is +-1 in about 1e-8 (step 1) to 1e-6 (step 0.01) of the integers
 * (measured and bounded in tests/test_gpu_fullsize.py). The quotient itself has 24 significant bits: a channel whose
 * coefficients reach |T| / step >= 2^23 -- the xyz columns of a 59-channel frame at steps below ~0.1: |T| is ~1e6 there --
 * cannot be integer-exact in float32 (differences of tens of units at step 0.01). Callers that need the reference's integers
 * on such channels use raht_fwd_quant_f64 / raht_dequant_inv_f64 (same kernels in float64, Q bit-identical to the
 * float64 two-call sequence) or a per-channel step table with coarser steps on those channels. */
int raht_fwd_quant(const raht_plan *plan, const float *C, int64_t ldc, int D, const float *steps,
                   int n_steps, int32_t *Q, int64_t ldq, raht_stream_t stream);
int raht_dequant_inv(const raht_plan *plan, const int32_t *Q, int64_t ldq, int D, const float *steps,
                     int n_steps, float *C, int64_t ldc, raht_stream_t stream);

/* ONE forward pass, k quantizations: the drivers quantize one coefficient matrix at nine steps (python/encode_3dgs.py:28 colorStep,
 * :199-217), i.e. the forward transform of a frame is the same for every step. Q[i] = floor(T / steps[i] + 0.5), reordered, for
 * i < k -- each bit-identical to raht_fwd_quant(plan, C, ..., &steps[i], 1, Q[i], ...) -- from one pass over C: the write-back of
 * every finalised row quantizes it k times. steps: HOST float[k] (one step for all channels each); Q: HOST array of k DEVICE
 * matrices (N x D int32, row stride ldq, distinct). Level engine / row-mapped / truncated plans: k single calls inside. */
int raht_fwd_quant_multi(const raht_plan *plan, const float *C, int64_t ldc, int D, const float *steps, int k,
                         int32_t *const *Q, int64_t ldq, raht_stream_t stream);

/* raht_dequant_inv fused with the drivers' distortion measurement (python/encode_3dgs.py:274 C_rec = iRAHT(...), then :298-310
 * torch.mean((C - C_rec) ** 2) over all / quats / scales / opacity / colour columns): the stage-0 kernel of the fused inverse
 * compares every row it reconstructs with the ORIGINAL attributes C_ref on its way out.
 *   sqdiff : DEVICE double[D]: sum over rows of (C_rec[i, c] - C_ref[i, c])^2 -- differences in float32, squares and sums in
 *            float64, summed in a fixed order (deterministic); what raht_sqdiff_columns(C_ref, C_rec) returns, up to the order
 *            of the float64 additions;
 *   C_rec  : the reconstruction, bit-identical to raht_dequant_inv's -- or NULL: then it is never written (the PSNR columns
 *            need the sums only), and a decode step is one pass over Q and C_ref instead of Q -> C_rec, then C_ref and C_rec.
 * Shapes outside the fused path (D > 64, trees of one launch, level engine, truncated plans) run the two calls inside. */
int raht_dequant_inv_sqdiff(const raht_plan *plan, const int32_t *Q, int64_t ldq, int D, const float *steps, int n_steps,
                            const float *C_ref, int64_t ld_ref, float *C_rec, int64_t ldc, double *sqdiff, raht_stream_t stream);

/* The same two at the reference's own precision (python/encode_3dgs.py:82-83: float64 coefficients are what
 * :204 quantizes): the float64 tile kernels with the float64 quantizer (IEEE double division, float64 steps) in their
 * write-back / row gather -- one pass, no N x D temporary. Q is bit-identical to raht_fwd_f64 + raht_quant_reorder_f64,
 * i.e. to floor(T64 / step + 0.5) of the float64 transform. (Level engine / D < 2: the two-call sequence.) */
int raht_fwd_quant_f64(const raht_plan *plan, const double *C, int64_t ldc, int D, const double *steps,
                       int n_steps, int32_t *Q, int64_t ldq, raht_stream_t stream);
int raht_dequant_inv_f64(const raht_plan *plan, const int32_t *Q, int64_t ldq, int D, const double *steps,
                         int n_steps, double *C, int64_t ldc, raht_stream_t stream);

/* MIXED PRECISION: float32 rows whose first n_wide channels (1..4) are carried in float64 through the whole transform, in the
 * SAME launches as the float32 channels (csrc/transform_mx.hip). What it is for: the reference quantizes float64 coefficients
 * (python/encode_3dgs.py:82-83,204), and on a 59-column frame whose columns 0-2 are the voxel coordinates
 * (python/voxelize_pc.py:155) those three columns' coefficients reach 1e6, i.e. |T| / step > 2^24 at the steps the attribute
 * channels need -- float32 integers are tens of units off there (see above) while the 56 attribute columns are fine.
 *   channels [0, n_wide)  : input converted exactly to float64, float64 butterflies with float64 a / b, IEEE double division by
 *                           the float64 step: Q bit-identical to raht_fwd_quant_f64 on those columns, i.e. the reference's
 *                           integers (up to 1-ulp transform noise on exact rounding ties); the inverse rounds once, on output;
 *   channels [n_wide, D)  : float32 arithmetic with (float) steps[c]: bit-identical to raht_fwd_quant / raht_dequant_inv.
 * steps: HOST float64[n_steps], n_steps == 1 or D. D - n_wide >= 4 and D <= 68 run the mixed tile kernels (one pass, ~5 % slower
 * than the float32 kernels); other shapes, and plans switched to the level engine, run the float32 path followed by a float64
 * pass over the n_wide columns (same results up to rounding ties). Not available for row-mapped or truncated plans. */
int raht_fwd_quant_mixed(const raht_plan *plan, const float *C, int64_t ldc, int D, const double *steps, int n_steps,
                         int n_wide, int32_t *Q, int64_t ldq, raht_stream_t stream);
int raht_dequant_inv_mixed(const raht_plan *plan, const int32_t *Q, int64_t ldq, int D, const double *steps, int n_steps,
                           int n_wide, float *C, int64_t ldc, raht_stream_t stream);
/* Tile rows / stage sizes the mixed kernels use for (D, n_wide); *tile_rows = 0 when the shape takes the two-pass path. */
int raht_plan_mixed_stats(raht_plan *plan, int D, int n_wide, int *tile_rows, int *n_stages, int64_t *rows_per_stage,
                          int max_stages);

/* SEVERAL SCENES IN ONE SET OF LAUNCHES (BASELINE configs[3] is a batch of scenes; the frames of a dynamic sequence are
 * one too). Scene i = (plans[i], C[i] / Q[i] with row strides ldc[i] / ldq[i]); all scenes share D and the step table.
 * Stage k of every scene runs in ONE tile-kernel launch (a workgroup finds its scene from its index) and the top stages in
 * one launch with blockIdx.y = scene: a frame of ~1 M Gaussians on its own fills the 768 workgroup slots of the chip two
 * and a half times and then spends a third of its step in three latency-bound tail launches; in a batch the partial rounds
 * of the scenes fill each other and all tails are one launch per stage. Results are BIT-IDENTICAL to n calls of the
 * single-scene entry points (same kernels, same tiles: tests/test_gpu_parity.py::test_batch_*). Any n >= 1 (launches carry
 * up to 8 scenes each); float32 tile engine -- scenes that need the level engine (D < 4, strides > 2^18, pathological key
 * patterns) or a row map run through their single-scene entry point inside the same call. Plans must be distinct objects
 * (a plan owns its workspaces) on the current device. Arrays of pointers / strides are HOST arrays. */
int raht_fwd_batch(int n, raht_plan *const *plans, const float *const *C, const int64_t *ldc, int D,
                   float *const *T, const int64_t *ldt, raht_stream_t stream);
int raht_inv_batch(int n, raht_plan *const *plans, const float *const *T, const int64_t *ldt, int D,
                   float *const *C, const int64_t *ldc, raht_stream_t stream);
int raht_fwd_quant_batch(int n, raht_plan *const *plans, const float *const *C, const int64_t *ldc, int D,
                         const float *steps, int n_steps, int32_t *const *Q, const int64_t *ldq, raht_stream_t stream);
int raht_dequant_inv_batch(int n, raht_plan *const *plans, const int32_t *const *Q, const int64_t *ldq, int D,
                           const float *steps, int n_steps, float *const *C, const int64_t *ldc, raht_stream_t stream);

/* Pre-build the tile schedule and the per-stage workspaces for (elem_size in {4, 8}, D). The
 * first transform with a new (element type, D) does this implicitly (allocating and synchronising
 * once); after raht_plan_prepare the transform entry points only enqueue kernels (hipGraph-safe).
 * A plan owns its workspaces: do not run transforms of one plan concurrently on several streams. */
int raht_plan_prepare(raht_plan *plan, int elem_size, int D, raht_stream_t stream);

/* A plan owns ONE set of per-stage workspaces, so its transforms must be ordered on one stream (above). With this switched on it
 * keeps one set PER DIRECTION: one forward-direction call (raht_fwd*, raht_fwd_quant*) and one inverse-direction call (raht_inv*,
 * raht_dequant_inv*) of the same plan may then be in flight at the same time on two streams -- the shape of the drivers' loop
 * over quantization steps (python/encode_3dgs.py:199-275): the forward of step s + 1 does not depend on the inverse of step s,
 * and one direction's latency-bound tail stages then run under the other's HBM-bound first stage. Costs a second copy of the
 * workspaces (~5 % of a coefficient matrix); they are re-allocated by the next transform or raht_plan_prepare. Two calls of the
 * SAME direction still have to be ordered by the caller. */
int raht_plan_set_concurrent_directions(raht_plan *plan, int on);

/* Profiling aid: HIP events (hipEvent_t, created by the caller with timing enabled) recorded on the
 * launch stream immediately before and after the STAGE-0 kernel of every following transform of this
 * plan, so that the dominant kernel can be timed inside a real step (hipEventElapsedTime after the
 * stream has been synchronised). NULL, NULL switches it off. */
int raht_plan_set_stage0_events(raht_plan *plan, void *ev_before, void *ev_after);

/* Profiling aid: enqueue ONE stage of the float32 tile schedule (stage 0 is the HBM-heavy launch).
 * Not a transform by itself; bench.py uses it to time the dominant kernel with HIP events.
 *   Q == NULL : plain kernels   (forward: mat = C in, mat2 = T out;  inverse: mat = T in, mat2 = C out)
 *   Q != NULL : fused-quantization kernels (forward: mat = C in, Q out;  inverse: Q in, mat2 = C out)
 * ablate: 0 = the real kernel; 1 = skip the butterflies; 2 = also skip merge resolution (a pure
 * staged copy) -- timing experiments only, the output is then not a transform. Ablations exist only in a
 * profiling build of the library (make ABLATE=1); the product build returns RAHT_ERR_UNSUPPORTED for
 * ablate != 0 and its kernels carry no ablation branches. */
int raht_debug_run_stage(const raht_plan *plan, int inverse, int stage, const float *mat, int64_t ld_mat,
                         int D, float *mat2, int64_t ld_mat2, int32_t *Q, int64_t ldq, float step,
                         int ablate, raht_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Quantize + reorder / dequantize + un-reorder (driver-inline in the reference,
 * python/encode_3dgs.py:204 floor(x/step+0.5), :210 index_select(0, order_RAGFT), :215 int32;
 * :261 x*step, :267-268 gather by argsort(order_RAGFT)).
 *   Q[k, c] = (int32) floor(T[order[k], c] / step_c + 0.5)        T[order[k], c] = Q[k, c] * step_c
 * steps: HOST array of n_steps values, n_steps == 1 (one step for all channels) or == D. */
int raht_quant_reorder(const raht_plan *plan, const float *T, int64_t ldt, int D, const float *steps,
                       int n_steps, int32_t *Q, int64_t ldq, raht_stream_t stream);
int raht_dequant_unreorder(const raht_plan *plan, const int32_t *Q, int64_t ldq, int D,
                           const float *steps, int n_steps, float *T, int64_t ldt,
                           raht_stream_t stream);
/* float64 coefficients and steps (the reference's precision): IEEE double division, as torch on the CPU. */
int raht_quant_reorder_f64(const raht_plan *plan, const double *T, int64_t ldt, int D, const double *steps,
                           int n_steps, int32_t *Q, int64_t ldq, raht_stream_t stream);
int raht_dequant_unreorder_f64(const raht_plan *plan, const int32_t *Q, int64_t ldq, int D,
                               const double *steps, int n_steps, double *T, int64_t ldt,
                               raht_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Voxelizer.  Replaces voxelize_pc_batched(PC, vmin, width, J) (reference
 * python/voxelize_pc.py:62-172) and get_morton_code (:25-59): shift by vmin, Vint =
 * clamp(floor((V - vmin) / (width / 2^J)), 0, 2^J - 1) in float32 as torch does, 3J-bit Morton
 * key, STABLE LSD radix sort on device, voxel starts where the key changes, per-voxel attribute
 * mean (sequential in sorted order, so bit-reproducible). The division is the IEEE float32 division torch performs on
 * the CPU (where the reference's golden vectors come from); a GPU run of the reference multiplies by 1 / voxel_size
 * instead and may put a point that sits within 1 ulp of a voxel face into the neighbouring voxel (fixtures
 * tests/golden/vox_boundary*_j9.npz: width 7.3, points at k * voxel_size and one float32 step either side).
 *   PC      : N x (3 + d) float32, row stride ldpc elements
 *   vmin_in : HOST float[3] or NULL (-> per-axis minimum);  width_in < 0 -> max over axes of V - vmin
 * Outputs (DEVICE, caller-allocated for N rows; any may be NULL):
 *   keys_sorted uint64[N], sort_idx int64[N], voxel_indices int64[<=N], PCvox float[<=N x (3+d)]
 *   (integer voxel coordinates as floats, then mean attributes), Vvox int64[<=N x 3].
 * HOST outputs: *n_vox, vmin_out[3], *width_out, *voxel_size_out. */
int raht_voxelize(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in,
                  double width_in, int J, uint64_t *keys_sorted, int64_t *sort_idx,
                  int64_t *voxel_indices, float *PCvox, int64_t *Vvox, int64_t *n_vox,
                  float vmin_out[3], double *width_out, double *voxel_size_out,
                  raht_stream_t stream);

/* voxelize_pc_batched with ALL its outputs in one call (reference python/voxelize_pc.py:62-172 returns PCvox, PCsorted,
 * voxel_indices, DeltaPC together): raht_voxelize plus PCsorted (N x (3+d), may be NULL) and DeltaPC (N x (3+d), may be NULL;
 * needs PCvox) as raht_voxelize_residuals defines them, produced by the SAME pass over the gathered rows that forms the
 * per-voxel means (clouds with >= 5 attribute columns; narrower ones run the two-call sequence inside and then need sort_idx).
 * Bit-identical to raht_voxelize followed by raht_voxelize_residuals. */
int raht_voxelize_all(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in, double width_in,
                      int J, uint64_t *keys_sorted, int64_t *sort_idx, int64_t *voxel_indices, float *PCvox,
                      int64_t *Vvox, float *PCsorted, float *DeltaPC, int64_t *n_vox, float vmin_out[3],
                      double *width_out, double *voxel_size_out, raht_stream_t stream);

/* One frame's prelude in one call: raht_voxelize, then the RAHT plan built STRAIGHT from the voxelizer's own sorted,
 * unique voxel keys (the reference goes through a PLY file and re-derives the same Morton keys from the voxel coordinates:
 * voxelize_pc_batched, python/voxelize_pc.py:62-172, then RAHT_param_reorder_fast, python/RAHT_param.py:190-279).
 *   voxel_keys : DEVICE uint64[N], caller-allocated, REQUIRED: the first n_vox entries receive the voxels' keys and are
 *                BORROWED by the plan (as raht_plan_create_from_keys_borrowed: keep them alive and unchanged while the plan lives)
 *   other arguments as raht_voxelize; PCvox[:, 3:] is the attribute matrix in the plan's row order.
 * *plan is NULL on failure. */
int raht_voxelize_plan(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in, double width_in,
                       int J, uint64_t *voxel_keys, int64_t *voxel_indices, float *PCvox, int64_t *n_vox,
                       float vmin_out[3], double *width_out, double *voxel_size_out, raht_stream_t stream,
                       raht_plan **plan);

/* The voxelizer's secondary outputs (reference python/voxelize_pc.py:103-111, 147-156), from the primary ones of
 * raht_voxelize: PCsorted[k, :] = PC[sort_idx[k], :] (N x (3+d), may be NULL) and the residuals DeltaPC (N x (3+d)):
 * positions V0 - voxel_size * floor(V0 / voxel_size) with V0 = V - vmin, attributes minus their voxel's mean (PCvox).
 * DEVICE pointers except vmin (HOST float[3]); only enqueues kernels on `stream`. */
int raht_voxelize_residuals(const float *PC, int64_t ldpc, int64_t N, int d, const uint64_t *keys_sorted,
                            const int64_t *sort_idx, const float *PCvox, const float vmin[3], double voxel_size,
                            float *PCsorted, float *DeltaPC, raht_stream_t stream);

/* The voxelizer's first phase alone: the (unsorted) 3J-bit Morton key of every point for a GIVEN bounding box
 * (vmin: HOST float[3]; width > 0), same arithmetic as raht_voxelize. A Morton-prefix sharded front end buckets
 * points by the top key bits with it before the all-to-all (sharded.exchange_by_prefix). keys: DEVICE uint64[N]. */
int raht_voxel_keys(const float *PC, int64_t ldpc, int64_t N, const float vmin[3], double width, int J,
                    uint64_t *keys, raht_stream_t stream);

/* Morton keys of integer coordinates (get_morton_code, voxelize_pc.py:25-59). V: N x 3 int64. */
int raht_morton(const int64_t *V, int64_t N, int J, uint64_t *keys, raht_stream_t stream);

/* Stable LSD radix sort of uint64 keys (nbits significant) with an int64 index payload
 * (idx_out[k] = original position). DEVICE buffers. */
int raht_sort_keys(const uint64_t *keys_in, int64_t N, int nbits, uint64_t *keys_out,
                   int64_t *idx_out, raht_stream_t stream);
/* The sort runs one launch per digit pass whose tiles hand their offsets to each other inside the launch (csrc/scan_sort.hip).
 * A tile waits for the tiles numbered before it; tiles are numbered by workgroup index, which cannot deadlock as long as the
 * lowest-numbered unfinished workgroup is resident -- how this GPU dispatches, but not a documented guarantee -- so every wait is
 * bounded by the wall clock (0.2 s): a tile that gives up raises an error word, nothing is written through stale offsets, and
 * the call (raht_sort_keys, raht_voxelize*) REPEATS the sort with one launch chain per digit, which needs no such property.
 * The caller sees a correct result and a slower call; this counter (process-wide, monotonic) says how often it happened.
 * Expected 0: tools/check_big.py and the GPU suite assert it. RAHT_SORT_TICKET=1 numbers the tiles by an atomic ticket
 * instead (acyclic whatever the dispatch order; +4 us per pass). */
int64_t raht_sort_fallbacks(void);

/* ------------------------------------------------------------------------------------------------
 * Per-voxel Gaussian merge. Replaces merge_clusters_cuda / merge_weighted_mean_kernel
 * (reference cuda/merge_cluster_wrapper.cu:11-116, cuda/merge_cluster.cu:2-111): cluster c owns the
 * Gaussians cluster_indices[cluster_offsets[c] .. cluster_offsets[c+1]); weight = opacity (or 1);
 * means / scales / colours = weighted means, quaternion = normalised weighted sum (identity
 * (0,0,0,1) if the norm is 0), opacity = min(sum, 1); empty clusters produce zeros. DEVICE pointers,
 * int32 indices / offsets, float32 data, row-major [N,3] [N,4] [N,3] [N] [N,color_dim]. */
int raht_merge_clusters(const int32_t *cluster_indices, const int32_t *cluster_offsets, int64_t num_clusters,
                        const float *means, const float *quats, const float *scales, const float *opacities,
                        const float *colors, int color_dim, int weight_by_opacity, float *merged_means,
                        float *merged_quats, float *merged_scales, float *merged_opacities,
                        float *merged_colors, raht_stream_t stream);

/* VOXELIZE + MERGE in one call and one pass over the rows (the reference runs its voxelizer and then its CUDA merge
 * kernel back to back: python/test_voxelize_3dgs.py:203-257, cuda/merge_cluster.cu:2-111 -- the sort permutation and the voxel
 * starts are the merge's cluster indices / offsets, :225-233).
 *   G    : N whole Gaussians, row-major [x y z | quat(4) | scale(3) | opacity | colour(color_dim)], row stride ldg >= 11 + color_dim
 *   Gvox : <= N rows of 11 + color_dim: the voxel's INTEGER coordinates as floats (like PCvox), then the merged attributes --
 *          the frame the RAHT path takes (59 columns at color_dim = 48); merged_means (<= N x 3, may be NULL): the merged positions
 * Per column exactly raht_merge_clusters' arithmetic on the members in sorted order (weight = opacity, or 1): bit-identical to
 * raht_voxelize (sort_idx, voxel_indices) followed by raht_merge_clusters on the five split arrays. Other arguments as raht_voxelize. */
int raht_voxelize_merge(const float *G, int64_t ldg, int64_t N, int color_dim, int weight_by_opacity, const float *vmin_in,
                        double width_in, int J, uint64_t *keys_sorted, int64_t *sort_idx, int64_t *voxel_indices, float *Gvox,
                        float *merged_means, int64_t *n_vox, float vmin_out[3], double *width_out, double *voxel_size_out,
                        raht_stream_t stream);

/* A few rows at explicit positions: Q[pos[i], :] = floor(X[i, :] / step + 0.5) and X[i, :] =
 * Q[pos[i], :] * step (pos: DEVICE int64[n], NULL = identity). Used for the <= 512 top coefficients of
 * a Morton-prefix sharded scene, which the shard-local fused kernels leave to the caller (no
 * counterpart in the reference; same arithmetic as python/encode_3dgs.py:204,215,261). */
int raht_quant_rows(const float *X, int64_t ldx, int64_t n, int D, const float *steps, int n_steps,
                    const int64_t *pos, int32_t *Q, int64_t ldq, raht_stream_t stream);
int raht_dequant_rows(const int32_t *Q, int64_t ldq, const int64_t *pos, int64_t n, int D, const float *steps,
                      int n_steps, float *X, int64_t ldx, raht_stream_t stream);

/* Rows at explicit positions, no arithmetic (elem_size 4 or 8 bytes per element; pos: DEVICE int64[n]):
 *   raht_rows_gather :  dst[i, :] = src[pos[i], :]        raht_rows_scatter :  dst[pos[i], :] = src[i, :]
 * One launch each; the sharded driver moves a shard's <= 512 root rows between the coefficient matrix and the
 * all-gather buffers with them. */
int raht_rows_gather(const void *src, int64_t ld_src, const int64_t *pos, int64_t n, int D, int elem_size, void *dst,
                     int64_t ld_dst, raht_stream_t stream);
int raht_rows_scatter(const void *src, int64_t ld_src, const int64_t *pos, int64_t n, int D, int elem_size, void *dst,
                      int64_t ld_dst, raht_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * RLGR entropy stage: adaptive Run-Length / Golomb-Rice coder, byte-exact with the
 * reference's vendored PyRLGR (python/PyRLGR/src/libs/rlgr/membuf.cpp:258-423, call sites
 * python/encode_3dgs.py:229-245). HOST pointers throughout: the coder is sequential and stays on
 * the CPU (as in the reference), but works on strided int32 columns of the quantized coefficient
 * matrix and codes the D channels on a pool of host threads.
 *   raht_rlgr_bound           : bytes that always suffice for n symbols
 *   raht_rlgr_encode          : == m = membuf(); m.rlgrWrite(seq, flag); m.close(); m.get_buffer()
 *   raht_rlgr_decode          : == membuf(buf).rlgrRead(n, flag)
 *   raht_rlgr_encode_channels : symbol n of channel c is Q[n * sym_stride + c * chan_stride] (row-major
 *                               N x D matrix: sym_stride = ld, chan_stride = 1; channel-major D x N as
 *                               produced by raht_transpose_i32: sym_stride = 1, chan_stride = N);
 *                               stream of channel c -> out + c * cap_per_channel, nbytes[c];
 *                               nthreads <= 0 = all host cores
 *   raht_rlgr_decode_channels : the inverse.
 *   raht_transpose_i32        : DEVICE int32 transpose (rows x cols, ld_in) -> (cols x rows, ld_out)
 *                               so that the host coder gets contiguous channels after the D2H copy. */
int64_t raht_rlgr_bound(int64_t n);
int raht_rlgr_encode(const int32_t *seq, int64_t n, int64_t stride, int flag_signed, uint8_t *out,
                     int64_t cap, int64_t *nbytes);
int raht_rlgr_decode(const uint8_t *buf, int64_t nbytes, int64_t n, int flag_signed, int32_t *seq,
                     int64_t stride);
int raht_rlgr_encode_channels(const int32_t *Q, int64_t N, int D, int64_t sym_stride, int64_t chan_stride,
                              int flag_signed, uint8_t *out, int64_t cap_per_channel, int64_t *nbytes,
                              int nthreads);
int raht_rlgr_decode_channels(const uint8_t *bufs, int64_t cap_per_channel, const int64_t *nbytes,
                              int64_t N, int D, int flag_signed, int32_t *Q, int64_t sym_stride,
                              int64_t chan_stride, int nthreads);
int raht_transpose_i32(const int32_t *in, int64_t ld_in, int64_t rows, int64_t cols, int32_t *out,
                       int64_t ld_out, raht_stream_t stream);
/* ------------------------------------------------------------------------------------------------
 * DIRECT all-gather for Morton-prefix sharded scenes (one process per GPU; SURVEY.md 5 / 8e: "a direct all-gather -- each
 * GPU writes its <= 15 KB slice to its 7 peers over the 7 point-to-point links -- beats a ring"). Opt-in alternative to
 * the RCCL all_gather_into_tensor of raht_3dgs_codec_amd.sharded (ShardedRaht(direct=True)); the reference has no
 * multi-GPU path at all.
 *   raht_xchg_alloc   : this rank's exchange block (fine-grained device memory: 2 x world x slot_bytes of data, double
 *                       buffered, + flags) and its 64-byte hipIpc handle, which the caller sends to the peers (once)
 *   raht_xchg_open    : map a peer's block from its handle;  raht_xchg_close / raht_xchg_free: undo
 *   raht_xchg_gather  : ONE launch (one workgroup per peer): write `send` into slot `rank` of buffer (seq & 1) of every
 *                       peer's block, raise the peer's flag, wait until every peer's slot has landed here. seq = 1, 2, 3, ...
 *                       identical on all ranks. Waits are bounded (20 s): a missing peer ends in status 1, not a hung GPU.
 *   raht_xchg_buffer  : device address of buffer (seq & 1): world slots of slot_bytes, slot r = rank r's rows
 *   raht_xchg_status  : (synchronises the stream) 0 = fine, 1 = a wait timed out
 * slot_bytes: a multiple of 16; world <= 8. */
int raht_xchg_bytes(int world, int64_t slot_bytes, int64_t *total);
int raht_xchg_alloc(int world, int64_t slot_bytes, void **base, void *handle64);
int raht_xchg_open(const void *handle64, void **base);
int raht_xchg_close(void *base);
int raht_xchg_free(void *base);
int raht_xchg_gather(const void *send, int64_t slot_bytes, void *const *peers, int rank, int world, uint32_t seq,
                     raht_stream_t stream);
int raht_xchg_buffer(void *base, int world, int64_t slot_bytes, uint32_t seq, void **buf);
int raht_xchg_status(void *base, int world, int64_t slot_bytes, raht_stream_t stream, int *status);

/* ------------------------------------------------------------------------------------------------
 * The RLGR stage ON THE GPU, segmented (csrc/rlgr_seg.hip; SURVEY.md 8f-1 "GPU-segmented"). Every channel is cut into
 * segments of seg_len symbols, every segment is an independent RLGR stream -- byte-identical to what the reference's
 * membuf::rlgrWrite (python/PyRLGR/src/libs/rlgr/membuf.cpp:340-423) emits for that slice of the channel, so any RLGR decoder
 * reads it -- and one lane codes one segment. The container (lengths + offsets + concatenated streams, 4-byte slots) is this
 * library's own: the reference codes a channel as ONE stream (raht_rlgr_encode_channels does exactly that, on the host).
 *   Q        : DEVICE int32, channel-major: symbol n of channel c at Q[c * chan_stride + n] (raht_transpose_i32's output)
 *   seg_bytes: DEVICE uint32[D * nseg], nseg = ceil(N / seg_len): exact stream length of segment c * nseg + s
 *   seg_off  : DEVICE uint32[D * nseg + 1]: its offset in `out` (slots padded to 4 bytes); the last entry = bytes used
 *   out      : DEVICE, 4-byte aligned, cap bytes. RAHT_ERR_NOMEM (and *total_bytes = what is needed) when cap is too small.
 * raht_rlgr_seg_encode synchronises the stream (it returns the size); raht_rlgr_seg_decode does not. */
int raht_rlgr_seg_encode(const int32_t *Q, int64_t N, int D, int64_t chan_stride, int seg_len, int flag_signed,
                         uint32_t *seg_bytes, uint32_t *seg_off, uint8_t *out, int64_t cap, int64_t *total_bytes,
                         raht_stream_t stream);
int raht_rlgr_seg_decode(const uint8_t *in, int64_t in_bytes, const uint32_t *seg_off, const uint32_t *seg_bytes, int64_t N,
                         int D, int seg_len, int flag_signed, int32_t *Q, int64_t chan_stride, uint32_t *bad_dev,
                         raht_stream_t stream);

/* The same coder on ANY of the two layouts of the quantized coefficients: channel-major (sym_stride = 1, chan_stride >= N: the
 * two functions above) or ROW-MAJOR (chan_stride = 1, sym_stride = ld >= D: symbol n of channel c at Q[n * ld + c], i.e. Q exactly
 * as raht_fwd_quant leaves it and raht_dequant_inv takes it -- no raht_transpose_i32 in front of the encoder or behind the
 * decoder: the lanes of a wave are then neighbouring channels at the same position of their segments, so every step of the wave
 * reads / writes one contiguous piece of a row). Same segments, same container, byte for byte. The container is limited to 4 GiB
 * (32-bit segment offsets): inputs whose worst case could exceed it are refused (RAHT_ERR_INVALID). */
int raht_rlgr_seg_encode_strided(const int32_t *Q, int64_t N, int D, int64_t sym_stride, int64_t chan_stride, int seg_len,
                                 int flag_signed, uint32_t *seg_bytes, uint32_t *seg_off, uint8_t *out, int64_t cap,
                                 int64_t *total_bytes, raht_stream_t stream);
int raht_rlgr_seg_decode_strided(const uint8_t *in, int64_t in_bytes, const uint32_t *seg_off, const uint32_t *seg_bytes, int64_t N,
                                 int D, int seg_len, int flag_signed, int32_t *Q, int64_t sym_stride, int64_t chan_stride,
                                 uint32_t *bad_dev, raht_stream_t stream);

/* k frames of ONE shape in one set of launches -- the quantization steps of a frame (python/encode_3dgs.py:199-275 codes them one
 * after the other; nothing connects them). One frame of 3 M x 56 symbols is 1.25 waves per SIMD of independent streams, and the
 * coder's time is that of one wave walking its segments; k frames behind blockIdx.y fill the chip's wave slots. Arrays of k
 * HOST pointers / sizes; every frame keeps its own tables and container and gets EXACTLY the bytes the one-frame entry points
 * above leave for it (same layouts, same limits per frame, 1 <= k <= RAHT_RLGR_BATCH_MAX). The encoder synchronises and
 * fills total_bytes[k] (RAHT_ERR_NOMEM: some cap[j] < total_bytes[j]); the decoder does not synchronise and sets bit j of
 * *bad_dev when a table entry of frame j reached outside in[j]. */
#define RAHT_RLGR_BATCH_MAX 12
int raht_rlgr_seg_encode_batch(int k, const int32_t *const *Q, int64_t N, int D, int64_t sym_stride, int64_t chan_stride, int seg_len,
                               int flag_signed, uint32_t *const *seg_bytes, uint32_t *const *seg_off, uint8_t *const *out,
                               const int64_t *cap, int64_t *total_bytes, raht_stream_t stream);
int raht_rlgr_seg_decode_batch(int k, const uint8_t *const *in, const int64_t *in_bytes, const uint32_t *const *seg_off,
                               const uint32_t *const *seg_bytes, int64_t N, int D, int seg_len, int flag_signed, int32_t *const *Q,
                               int64_t sym_stride, int64_t chan_stride, uint32_t *bad_dev, raht_stream_t stream);
/* The same, ROW-MAJOR frames only (chan_stride = 1), with the drivers' round-trip assertion (python/encode_3dgs.py:242-245: what was
 * decoded equals what was encoded) inside the decoder: expect[j] (DEVICE, layout and strides of Q[j]) is compared symbol by symbol
 * on the way out -- in this layout one more contiguous read per iteration of a wave -- and bit 16 + j of *bad_dev (required) is set
 * when frame j differs. Replaces a comparison pass over two N x D arrays per frame. */
int raht_rlgr_seg_decode_batch_check(int k, const uint8_t *const *in, const int64_t *in_bytes, const uint32_t *const *seg_off,
                                     const uint32_t *const *seg_bytes, int64_t N, int D, int seg_len, int flag_signed,
                                     int32_t *const *Q, const int32_t *const *expect, int64_t sym_stride, int64_t chan_stride,
                                     uint32_t *bad_dev, raht_stream_t stream);

/* How the decoders' symbols leave the lanes: -1 = chosen by the number of lanes in flight (default: one 4-byte store per symbol
 * below 200 000 lanes, where the L2 still gathers a lane's line; above, a 16-word LDS column per lane written out as aligned
 * 64-byte pieces -- with the steps of a frame decoded together the one-word stores cost 5.9 x the symbols' bytes in HBM writes),
 * 0 = words, 1 = 16-byte register groups, 2 = LDS columns. Same output in every mode (tests walk all of them); a tuning and
 * testing knob, process-wide. Returns the previous setting. */
int raht_debug_rlgr_decode_out(int mode);
/* The same for the words of the batched ENCODER's streams on their way into its slots: -1 = by the lanes in flight, 0 = one
 * 4-byte store per word, 2 = LDS columns, 64-byte pieces (9 steps of a 3 M x 56 frame: 0.79 -> 0.71 ms per step). */
int raht_debug_rlgr_encode_out(int mode);

/* out[c] = sum over rows of (A[i, c] - B[i, c])^2, DEVICE double[D]: what the drivers' five PSNR columns are made of
 * (python/encode_3dgs.py:298-310: torch.mean((C - C_rec) ** 2) over all / quats / scales / opacity / colour columns -- each a
 * sum of these D numbers divided by the element count). A, B: N x D DEVICE matrices of dtype RAHT_F32 or RAHT_F64 (differences
 * in that type, squares and sums in float64, deterministic). Two launches, no host round trip. */
int raht_sqdiff_columns(const void *A, int64_t lda, const void *B, int64_t ldb, int64_t N, int D, int dtype, double *out,
                        raht_stream_t stream);

/* Host: are two contiguous int32 arrays equal? *first_diff = index of the first difference or -1. Threaded (the drivers'
 * round-trip assertion, python/encode_3dgs.py:242-245, on 10^8 symbols). */
int raht_i32_equal(const int32_t *a, const int32_t *b, int64_t n, int nthreads, int64_t *first_diff);

#ifdef __cplusplus
}
#endif
#endif /* RAHT_H */
