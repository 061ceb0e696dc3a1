/*
 * raht_cpu.h -- host twins of the product's C ABI (TEST INFRASTRUCTURE ONLY, part of the oracle).
 *
 * SURVEY.md 8(b), last row: "host twins raht_cpu_* with identical signatures". Every function below has the
 * parameter list of its namesake in include/raht.h (raht_cpu_X <-> raht_X, raht_cpu_plan <-> raht_plan), so
 * a host can swap the CPU restatement and the MI355X path by symbol name; tests/test_cpu_twins.py checks the
 * prototypes textually against include/raht.h and the results against the golden vectors.
 * Differences by construction: every pointer is a HOST pointer, `stream` is ignored, and the float32 entry
 * points compute in float64 (the oracle's arithmetic, i.e. the reference's) and round the result once.
 * Built by oracle/Makefile into oracle/_build/libraht_cpu.so on top of raht_oracle.c. The product package
 * never loads it and keeps refusing CPU tensors.
 */
#ifndef RAHT_CPU_H
#define RAHT_CPU_H
#include <stdint.h>
#include "raht.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct raht_cpu_plan raht_cpu_plan;

const char *raht_cpu_last_error(void);
int raht_cpu_version(void);

int raht_cpu_plan_create(const void *V, int v_dtype, int64_t N, const double minV[3], double width,
                     int depth, raht_stream_t stream, raht_cpu_plan **out);
int raht_cpu_plan_create_from_keys(const uint64_t *keys_sorted, int64_t N, int nbits,
                               const int64_t *leaf_weights, raht_stream_t stream, raht_cpu_plan **out);
int raht_cpu_plan_create_from_keys_borrowed(const uint64_t *keys_sorted, int64_t N, int nbits,
                                        const int64_t *leaf_weights, raht_stream_t stream, raht_cpu_plan **out);
int raht_cpu_plan_destroy(raht_cpu_plan *plan);
int64_t raht_cpu_plan_size(const raht_cpu_plan *plan);          /* N */
int raht_cpu_plan_nbits(const raht_cpu_plan *plan);             /* 3 * depth */
int raht_cpu_plan_levels(const raht_cpu_plan *plan);
int raht_cpu_plan_export_level(const raht_cpu_plan *plan, int level, int64_t *list, uint8_t *flags,
                           int64_t *weights, int64_t *n);
int raht_cpu_plan_order(const raht_cpu_plan *plan, int64_t *order_dev, raht_stream_t stream);

int raht_cpu_fwd(const raht_cpu_plan *plan, const float *C, int64_t ldc, int D, float *T, int64_t ldt,
             float *w, raht_stream_t stream);
int raht_cpu_fwd_f64(const raht_cpu_plan *plan, const double *C, int64_t ldc, int D, double *T, int64_t ldt,
                 double *w, raht_stream_t stream);
int raht_cpu_inv(const raht_cpu_plan *plan, const float *T, int64_t ldt, int D, float *C, int64_t ldc,
             raht_stream_t stream);
int raht_cpu_inv_f64(const raht_cpu_plan *plan, const double *T, int64_t ldt, int D, double *C, int64_t ldc,
                 raht_stream_t stream);
int raht_cpu_fwd_quant(const raht_cpu_plan *plan, const float *C, int64_t ldc, int D, const float *steps,
                   int n_steps, int32_t *Q, int64_t ldq, raht_stream_t stream);
int raht_cpu_dequant_inv(const raht_cpu_plan *plan, const int32_t *Q, int64_t ldq, int D, const float *steps,
                     int n_steps, float *C, int64_t ldc, raht_stream_t stream);
int raht_cpu_fwd_quant_f64(const raht_cpu_plan *plan, const double *C, int64_t ldc, int D, const double *steps,
                       int n_steps, int32_t *Q, int64_t ldq, raht_stream_t stream);
int raht_cpu_dequant_inv_f64(const raht_cpu_plan *plan, const int32_t *Q, int64_t ldq, int D, const double *steps,
                         int n_steps, double *C, int64_t ldc, raht_stream_t stream);
int raht_cpu_fwd_batch(int n, raht_cpu_plan *const *plans, const float *const *C, const int64_t *ldc, int D,
                   float *const *T, const int64_t *ldt, raht_stream_t stream);
int raht_cpu_inv_batch(int n, raht_cpu_plan *const *plans, const float *const *T, const int64_t *ldt, int D,
                   float *const *C, const int64_t *ldc, raht_stream_t stream);
int raht_cpu_fwd_quant_batch(int n, raht_cpu_plan *const *plans, const float *const *C, const int64_t *ldc, int D,
                         const float *steps, int n_steps, int32_t *const *Q, const int64_t *ldq, raht_stream_t stream);
int raht_cpu_dequant_inv_batch(int n, raht_cpu_plan *const *plans, const int32_t *const *Q, const int64_t *ldq, int D,
                           const float *steps, int n_steps, float *const *C, const int64_t *ldc, raht_stream_t stream);
int raht_cpu_quant_reorder(const raht_cpu_plan *plan, const float *T, int64_t ldt, int D, const float *steps,
                       int n_steps, int32_t *Q, int64_t ldq, raht_stream_t stream);
int raht_cpu_dequant_unreorder(const raht_cpu_plan *plan, const int32_t *Q, int64_t ldq, int D,
                           const float *steps, int n_steps, float *T, int64_t ldt,
                           raht_stream_t stream);
int raht_cpu_quant_reorder_f64(const raht_cpu_plan *plan, const double *T, int64_t ldt, int D, const double *steps,
                           int n_steps, int32_t *Q, int64_t ldq, raht_stream_t stream);
int raht_cpu_dequant_unreorder_f64(const raht_cpu_plan *plan, const int32_t *Q, int64_t ldq, int D,
                               const double *steps, int n_steps, double *T, int64_t ldt,
                               raht_stream_t stream);

int raht_cpu_voxelize(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in,
                  double width_in, int J, uint64_t *keys_sorted, int64_t *sort_idx,
                  int64_t *voxel_indices, float *PCvox, int64_t *Vvox, int64_t *n_vox,
                  float vmin_out[3], double *width_out, double *voxel_size_out,
                  raht_stream_t stream);
int raht_cpu_voxelize_residuals(const float *PC, int64_t ldpc, int64_t N, int d, const uint64_t *keys_sorted,
                            const int64_t *sort_idx, const float *PCvox, const float vmin[3], double voxel_size,
                            float *PCsorted, float *DeltaPC, raht_stream_t stream);
int raht_cpu_voxelize_all(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in, double width_in,
                      int J, uint64_t *keys_sorted, int64_t *sort_idx, int64_t *voxel_indices, float *PCvox,
                      int64_t *Vvox, float *PCsorted, float *DeltaPC, int64_t *n_vox, float vmin_out[3],
                      double *width_out, double *voxel_size_out, raht_stream_t stream);
int raht_cpu_voxelize_plan(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in, double width_in,
                       int J, uint64_t *voxel_keys, int64_t *voxel_indices, float *PCvox, int64_t *n_vox,
                       float vmin_out[3], double *width_out, double *voxel_size_out, raht_stream_t stream,
                       raht_cpu_plan **plan);
int raht_cpu_morton(const int64_t *V, int64_t N, int J, uint64_t *keys, raht_stream_t stream);
int raht_cpu_sort_keys(const uint64_t *keys_in, int64_t N, int nbits, uint64_t *keys_out,
                   int64_t *idx_out, raht_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RAHT_CPU_H */
