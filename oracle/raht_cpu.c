/*
 * raht_cpu.c -- host twins of the product's C ABI on top of the oracle (TEST INFRASTRUCTURE ONLY).
 * See raht_cpu.h. Each function cites the product entry it twins; the arithmetic is raht_oracle.c's
 * (the literal restatement of the reference), so this file is only argument plumbing.
 */
#include "raht_cpu.h"
#include "raht_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct raht_cpu_plan {
    int64_t N;
    int nbits;
    orc_param *p;          /* List / Flags / weights / order_RAGFT (ref_quirks = 0: a true permutation) */
};

static __thread char g_err[256] = "";
static void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char *raht_cpu_last_error(void) { return g_err; }
int raht_cpu_version(void) { return RAHT_VERSION; }

/* raht_plan_create (include/raht.h): same validation and error codes as the device path */
int raht_cpu_plan_create(const void *V, int v_dtype, int64_t N, const double minV[3], double width,
                         int depth, raht_stream_t stream, raht_cpu_plan **out)
{
    (void)stream;
    if (!V || !out || !minV) { set_err("raht_cpu_plan_create: NULL argument"); return RAHT_ERR_INVALID; }
    if (N < 1 || N >= ((int64_t)1 << 31) || depth < 1 || depth > 21 || !(width > 0)) { set_err("raht_cpu_plan_create: bad N / depth / width"); return RAHT_ERR_INVALID; }
    double *Vd = (double *)malloc(sizeof(double) * 3 * (size_t)N);
    if (!Vd) return RAHT_ERR_NOMEM;
    for (int64_t i = 0; i < 3 * N; ++i) {
        switch (v_dtype) {
        case RAHT_F64: Vd[i] = ((const double *)V)[i]; break;
        case RAHT_F32: Vd[i] = (double)((const float *)V)[i]; break;
        case RAHT_I32: Vd[i] = (double)((const int32_t *)V)[i]; break;
        case RAHT_I64: Vd[i] = (double)((const int64_t *)V)[i]; break;
        default: free(Vd); set_err("raht_cpu_plan_create: bad v_dtype %d", v_dtype); return RAHT_ERR_INVALID;
        }
    }
    const double Q = width / (double)((uint64_t)1 << depth);
    const int64_t hi = (int64_t)1 << depth;
    int64_t *Vi = (int64_t *)malloc(sizeof(int64_t) * 3 * (size_t)N);
    uint64_t *mc = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N);
    if (!Vi || !mc) { free(Vd); free(Vi); free(mc); return RAHT_ERR_NOMEM; }
    int rc = RAHT_OK;
    for (int64_t i = 0; i < N && rc == RAHT_OK; ++i)
        for (int a = 0; a < 3; ++a) {
            const int64_t q = (int64_t)floor((Vd[3 * i + a] - minV[a]) / Q);     /* RAHT_param.py:205-206 */
            if (q < 0 || q >= hi) { set_err("coordinate out of [0, 2^%d) at row %lld", depth, (long long)i); rc = RAHT_ERR_BOUNDS; break; }
            Vi[3 * i + a] = q;
        }
    if (rc == RAHT_OK) {
        orc_morton(Vi, N, depth, mc);
        for (int64_t i = 1; i < N; ++i)
            if (mc[i] <= mc[i - 1]) { set_err("Morton keys are not strictly increasing at row %lld", (long long)i); rc = RAHT_ERR_UNSORTED; break; }
    }
    free(Vi); free(mc);
    raht_cpu_plan *pl = NULL;
    if (rc == RAHT_OK) {
        pl = (raht_cpu_plan *)calloc(1, sizeof(*pl));
        if (!pl || orc_param_build(Vd, N, minV, width, depth, 0, &pl->p) != 0) { free(pl); pl = NULL; rc = RAHT_ERR_NOMEM; }
        else { pl->N = N; pl->nbits = 3 * depth; }
    }
    free(Vd);
    if (rc == RAHT_OK) *out = pl;
    return rc;
}

/* raht_plan_create_from_keys: de-interleave the keys into coordinates, then as above */
int raht_cpu_plan_create_from_keys(const uint64_t *keys_sorted, int64_t N, int nbits,
                                   const int64_t *leaf_weights, raht_stream_t stream, raht_cpu_plan **out)
{
    if (!keys_sorted || !out) { set_err("raht_cpu_plan_create_from_keys: NULL argument"); return RAHT_ERR_INVALID; }
    if (N < 1 || nbits < 1 || nbits > 63) { set_err("raht_cpu_plan_create_from_keys: bad N / nbits"); return RAHT_ERR_INVALID; }
    if (leaf_weights) { set_err("raht_cpu_plan_create_from_keys: weighted plans have no CPU twin"); return RAHT_ERR_UNSUPPORTED; }
    const int depth = (nbits + 2) / 3;
    double *V = (double *)malloc(sizeof(double) * 3 * (size_t)N);
    if (!V) return RAHT_ERR_NOMEM;
    for (int64_t i = 0; i < N; ++i) {
        const uint64_t k = keys_sorted[i];
        if (nbits < 64 && (k >> nbits) != 0) { free(V); set_err("key out of bounds at row %lld", (long long)i); return RAHT_ERR_BOUNDS; }
        int64_t x = 0, y = 0, z = 0;
        for (int b = 0; b < depth; ++b) {
            const uint64_t dg = (k >> (3 * b)) & 7u;          /* digit = z + 2 y + 4 x, voxelize_pc.py:50-57 */
            z |= (int64_t)(dg & 1u) << b; y |= (int64_t)((dg >> 1) & 1u) << b; x |= (int64_t)((dg >> 2) & 1u) << b;
        }
        V[3 * i] = (double)x; V[3 * i + 1] = (double)y; V[3 * i + 2] = (double)z;
    }
    const double zero[3] = {0, 0, 0};
    const int rc = raht_cpu_plan_create(V, RAHT_F64, N, zero, (double)((uint64_t)1 << depth), depth, stream, out);
    free(V);
    if (rc == RAHT_OK) (*out)->nbits = nbits;
    return rc;
}

/* raht_plan_create_from_keys_borrowed: the twin has nothing to borrow (it turns the keys into coordinates at once) */
int raht_cpu_plan_create_from_keys_borrowed(const uint64_t *keys_sorted, int64_t N, int nbits,
                                            const int64_t *leaf_weights, raht_stream_t stream, raht_cpu_plan **out)
{
    return raht_cpu_plan_create_from_keys(keys_sorted, N, nbits, leaf_weights, stream, out);
}

int raht_cpu_plan_destroy(raht_cpu_plan *plan)
{
    if (plan) { orc_param_free(plan->p); free(plan); }
    return RAHT_OK;
}

int64_t raht_cpu_plan_size(const raht_cpu_plan *plan) { return plan ? plan->N : -1; }
int raht_cpu_plan_nbits(const raht_cpu_plan *plan) { return plan ? plan->nbits : -1; }
int raht_cpu_plan_levels(const raht_cpu_plan *plan) { return plan ? orc_param_levels(plan->p) : -1; }

int raht_cpu_plan_export_level(const raht_cpu_plan *plan, int level, int64_t *list, uint8_t *flags,
                               int64_t *weights, int64_t *n)
{
    if (!plan || !n || level < 0 || level >= orc_param_levels(plan->p)) { set_err("raht_cpu_plan_export_level: bad argument"); return RAHT_ERR_INVALID; }
    const int64_t len = orc_param_level_len(plan->p, level);
    if (list) memcpy(list, orc_param_list(plan->p, level), sizeof(int64_t) * (size_t)len);
    if (flags) memcpy(flags, orc_param_flags(plan->p, level), (size_t)len);
    if (weights) memcpy(weights, orc_param_weights(plan->p, level), sizeof(int64_t) * (size_t)len);
    *n = len;
    return RAHT_OK;
}

int raht_cpu_plan_order(const raht_cpu_plan *plan, int64_t *order_dev, raht_stream_t stream)
{
    (void)stream;
    if (!plan || !order_dev) { set_err("raht_cpu_plan_order: NULL argument"); return RAHT_ERR_INVALID; }
    memcpy(order_dev, orc_param_order(plan->p), sizeof(int64_t) * (size_t)plan->N);
    return RAHT_OK;
}

/* strided matrix <-> dense float64 */
static double *dense_from_f32(const float *X, int64_t ld, int64_t N, int D)
{
    double *M = (double *)malloc(sizeof(double) * (size_t)N * (size_t)D);
    if (M) for (int64_t i = 0; i < N; ++i) for (int c = 0; c < D; ++c) M[i * D + c] = (double)X[i * ld + c];
    return M;
}
static double *dense_from_f64(const double *X, int64_t ld, int64_t N, int D)
{
    double *M = (double *)malloc(sizeof(double) * (size_t)N * (size_t)D);
    if (M) for (int64_t i = 0; i < N; ++i) memcpy(M + i * D, X + i * ld, sizeof(double) * (size_t)D);
    return M;
}

static int xform(const raht_cpu_plan *pl, double *in, int D, int inverse, double **out, double *w)
{
    double *o = (double *)malloc(sizeof(double) * (size_t)pl->N * (size_t)D);
    if (!in || !o) { free(in); free(o); return RAHT_ERR_NOMEM; }
    const int rc = inverse ? orc_raht_inv(in, pl->N, D, pl->p, o) : orc_raht_fwd(in, pl->N, D, pl->p, o, w);
    free(in);
    if (rc != 0) { free(o); return RAHT_ERR_INVALID; }
    *out = o;
    return RAHT_OK;
}

#define CHECK_XF(what) if (!plan || !src || !dst || D < 1 || ld_src < D || ld_dst < D) { set_err(what ": bad argument"); return RAHT_ERR_INVALID; }

/* raht_fwd / raht_fwd_f64 / raht_inv / raht_inv_f64 */
int raht_cpu_fwd(const raht_cpu_plan *plan, const float *src, int64_t ld_src, int D, float *dst, int64_t ld_dst, float *w, raht_stream_t stream)
{
    (void)stream;
    CHECK_XF("raht_cpu_fwd");
    double *T = NULL, *wd = w ? (double *)malloc(sizeof(double) * (size_t)plan->N) : NULL;
    int rc = xform(plan, dense_from_f32(src, ld_src, plan->N, D), D, 0, &T, wd);
    if (rc == RAHT_OK) {
        for (int64_t i = 0; i < plan->N; ++i) for (int c = 0; c < D; ++c) dst[i * ld_dst + c] = (float)T[i * D + c];
        if (w) for (int64_t i = 0; i < plan->N; ++i) w[i] = (float)wd[i];
    }
    free(T); free(wd);
    return rc;
}
int raht_cpu_fwd_f64(const raht_cpu_plan *plan, const double *src, int64_t ld_src, int D, double *dst, int64_t ld_dst, double *w, raht_stream_t stream)
{
    (void)stream;
    CHECK_XF("raht_cpu_fwd_f64");
    double *T = NULL;
    int rc = xform(plan, dense_from_f64(src, ld_src, plan->N, D), D, 0, &T, w);
    if (rc == RAHT_OK) for (int64_t i = 0; i < plan->N; ++i) memcpy(dst + i * ld_dst, T + i * D, sizeof(double) * (size_t)D);
    free(T);
    return rc;
}
int raht_cpu_inv(const raht_cpu_plan *plan, const float *src, int64_t ld_src, int D, float *dst, int64_t ld_dst, raht_stream_t stream)
{
    (void)stream;
    CHECK_XF("raht_cpu_inv");
    double *Cm = NULL;
    int rc = xform(plan, dense_from_f32(src, ld_src, plan->N, D), D, 1, &Cm, NULL);
    if (rc == RAHT_OK) for (int64_t i = 0; i < plan->N; ++i) for (int c = 0; c < D; ++c) dst[i * ld_dst + c] = (float)Cm[i * D + c];
    free(Cm);
    return rc;
}
int raht_cpu_inv_f64(const raht_cpu_plan *plan, const double *src, int64_t ld_src, int D, double *dst, int64_t ld_dst, raht_stream_t stream)
{
    (void)stream;
    CHECK_XF("raht_cpu_inv_f64");
    double *Cm = NULL;
    int rc = xform(plan, dense_from_f64(src, ld_src, plan->N, D), D, 1, &Cm, NULL);
    if (rc == RAHT_OK) for (int64_t i = 0; i < plan->N; ++i) memcpy(dst + i * ld_dst, Cm + i * D, sizeof(double) * (size_t)D);
    free(Cm);
    return rc;
}

/* quantize + reorder / dequantize + un-reorder (encode_3dgs.py:204,210,215 / :261,:267-268). float32 twins use
 * float32 arithmetic here (x / step in float32, like torch on a float32 tensor), float64 twins float64. */
static int steps_ok(const void *steps, int n_steps, int D) { return steps && (n_steps == 1 || n_steps == D); }

int raht_cpu_quant_reorder(const raht_cpu_plan *plan, const float *T, int64_t ldt, int D, const float *steps, int n_steps, int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    (void)stream;
    if (!plan || !T || !Q || D < 1 || ldt < D || ldq < D || !steps_ok(steps, n_steps, D)) { set_err("raht_cpu_quant_reorder: bad argument"); return RAHT_ERR_INVALID; }
    const int64_t *order = orc_param_order(plan->p);
    for (int64_t k = 0; k < plan->N; ++k) for (int c = 0; c < D; ++c)
        Q[k * ldq + c] = (int32_t)floorf(T[order[k] * ldt + c] / steps[n_steps == 1 ? 0 : c] + 0.5f);
    return RAHT_OK;
}
int raht_cpu_dequant_unreorder(const raht_cpu_plan *plan, const int32_t *Q, int64_t ldq, int D, const float *steps, int n_steps, float *T, int64_t ldt, raht_stream_t stream)
{
    (void)stream;
    if (!plan || !T || !Q || D < 1 || ldt < D || ldq < D || !steps_ok(steps, n_steps, D)) { set_err("raht_cpu_dequant_unreorder: bad argument"); return RAHT_ERR_INVALID; }
    const int64_t *order = orc_param_order(plan->p);
    for (int64_t k = 0; k < plan->N; ++k) for (int c = 0; c < D; ++c)
        T[order[k] * ldt + c] = (float)Q[k * ldq + c] * steps[n_steps == 1 ? 0 : c];
    return RAHT_OK;
}
int raht_cpu_quant_reorder_f64(const raht_cpu_plan *plan, const double *T, int64_t ldt, int D, const double *steps, int n_steps, int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    (void)stream;
    if (!plan || !T || !Q || D < 1 || ldt < D || ldq < D || !steps_ok(steps, n_steps, D)) { set_err("raht_cpu_quant_reorder_f64: bad argument"); return RAHT_ERR_INVALID; }
    const int64_t *order = orc_param_order(plan->p);
    for (int64_t k = 0; k < plan->N; ++k) for (int c = 0; c < D; ++c)
        Q[k * ldq + c] = (int32_t)floor(T[order[k] * ldt + c] / steps[n_steps == 1 ? 0 : c] + 0.5);
    return RAHT_OK;
}
int raht_cpu_dequant_unreorder_f64(const raht_cpu_plan *plan, const int32_t *Q, int64_t ldq, int D, const double *steps, int n_steps, double *T, int64_t ldt, raht_stream_t stream)
{
    (void)stream;
    if (!plan || !T || !Q || D < 1 || ldt < D || ldq < D || !steps_ok(steps, n_steps, D)) { set_err("raht_cpu_dequant_unreorder_f64: bad argument"); return RAHT_ERR_INVALID; }
    const int64_t *order = orc_param_order(plan->p);
    for (int64_t k = 0; k < plan->N; ++k) for (int c = 0; c < D; ++c)
        T[order[k] * ldt + c] = (double)Q[k * ldq + c] * steps[n_steps == 1 ? 0 : c];
    return RAHT_OK;
}

/* fused entries = the two-call sequences (the product guarantees the same) */
int raht_cpu_fwd_quant(const raht_cpu_plan *plan, const float *C, int64_t ldc, int D, const float *steps, int n_steps, int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    if (!plan || D < 1) { set_err("raht_cpu_fwd_quant: bad argument"); return RAHT_ERR_INVALID; }
    float *T = (float *)malloc(sizeof(float) * (size_t)plan->N * (size_t)D);
    if (!T) return RAHT_ERR_NOMEM;
    int rc = raht_cpu_fwd(plan, C, ldc, D, T, D, NULL, stream);
    if (rc == RAHT_OK) rc = raht_cpu_quant_reorder(plan, T, D, D, steps, n_steps, Q, ldq, stream);
    free(T);
    return rc;
}
int raht_cpu_dequant_inv(const raht_cpu_plan *plan, const int32_t *Q, int64_t ldq, int D, const float *steps, int n_steps, float *C, int64_t ldc, raht_stream_t stream)
{
    if (!plan || D < 1) { set_err("raht_cpu_dequant_inv: bad argument"); return RAHT_ERR_INVALID; }
    float *T = (float *)malloc(sizeof(float) * (size_t)plan->N * (size_t)D);
    if (!T) return RAHT_ERR_NOMEM;
    int rc = raht_cpu_dequant_unreorder(plan, Q, ldq, D, steps, n_steps, T, D, stream);
    if (rc == RAHT_OK) rc = raht_cpu_inv(plan, T, D, D, C, ldc, stream);
    free(T);
    return rc;
}
int raht_cpu_fwd_quant_f64(const raht_cpu_plan *plan, const double *C, int64_t ldc, int D, const double *steps, int n_steps, int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    if (!plan || D < 1) { set_err("raht_cpu_fwd_quant_f64: bad argument"); return RAHT_ERR_INVALID; }
    double *T = (double *)malloc(sizeof(double) * (size_t)plan->N * (size_t)D);
    if (!T) return RAHT_ERR_NOMEM;
    int rc = raht_cpu_fwd_f64(plan, C, ldc, D, T, D, NULL, stream);
    if (rc == RAHT_OK) rc = raht_cpu_quant_reorder_f64(plan, T, D, D, steps, n_steps, Q, ldq, stream);
    free(T);
    return rc;
}
int raht_cpu_dequant_inv_f64(const raht_cpu_plan *plan, const int32_t *Q, int64_t ldq, int D, const double *steps, int n_steps, double *C, int64_t ldc, raht_stream_t stream)
{
    if (!plan || D < 1) { set_err("raht_cpu_dequant_inv_f64: bad argument"); return RAHT_ERR_INVALID; }
    double *T = (double *)malloc(sizeof(double) * (size_t)plan->N * (size_t)D);
    if (!T) return RAHT_ERR_NOMEM;
    int rc = raht_cpu_dequant_unreorder_f64(plan, Q, ldq, D, steps, n_steps, T, D, stream);
    if (rc == RAHT_OK) rc = raht_cpu_inv_f64(plan, T, D, D, C, ldc, stream);
    free(T);
    return rc;
}

/* raht_voxelize / raht_morton / raht_sort_keys */
int raht_cpu_voxelize(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in, double width_in, int J,
                      uint64_t *keys_sorted, int64_t *sort_idx, int64_t *voxel_indices, float *PCvox, int64_t *Vvox,
                      int64_t *n_vox, float vmin_out[3], double *width_out, double *voxel_size_out, raht_stream_t stream)
{
    (void)stream;
    if (!PC || N < 1 || d < 0 || ldpc < 3 + d || J < 1 || J > 21 || !n_vox) { set_err("raht_cpu_voxelize: bad argument"); return RAHT_ERR_INVALID; }
    const int ld = 3 + d;
    float *P = (float *)malloc(sizeof(float) * (size_t)N * (size_t)ld);
    uint64_t *ks = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N);
    int64_t *si = (int64_t *)malloc(sizeof(int64_t) * (size_t)N), *vi = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *vv = (int64_t *)malloc(sizeof(int64_t) * 3 * (size_t)N);
    float *pv = (float *)malloc(sizeof(float) * (size_t)N * (size_t)ld);
    int rc = RAHT_ERR_NOMEM;
    if (P && ks && si && vi && vv && pv) {
        for (int64_t i = 0; i < N; ++i) memcpy(P + i * ld, PC + i * ldpc, sizeof(float) * (size_t)ld);
        float vm[3]; double w = 0, vs = 0; int64_t nv = 0;
        rc = orc_voxelize(P, N, d, vmin_in, width_in, J, ks, si, vi, pv, vv, &nv, vm, &w, &vs) == 0 ? RAHT_OK : RAHT_ERR_INVALID;
        if (rc == RAHT_OK) {
            if (keys_sorted) memcpy(keys_sorted, ks, sizeof(uint64_t) * (size_t)N);
            if (sort_idx) memcpy(sort_idx, si, sizeof(int64_t) * (size_t)N);
            if (voxel_indices) memcpy(voxel_indices, vi, sizeof(int64_t) * (size_t)nv);
            if (PCvox) memcpy(PCvox, pv, sizeof(float) * (size_t)nv * (size_t)ld);
            if (Vvox) memcpy(Vvox, vv, sizeof(int64_t) * 3 * (size_t)nv);
            *n_vox = nv;
            if (vmin_out) memcpy(vmin_out, vm, sizeof(vm));
            if (width_out) *width_out = w;
            if (voxel_size_out) *voxel_size_out = vs;
        }
    }
    free(P); free(ks); free(si); free(vi); free(vv); free(pv);
    return rc;
}

/* raht_voxelize_residuals */
int raht_cpu_voxelize_residuals(const float *PC, int64_t ldpc, int64_t N, int d, const uint64_t *keys_sorted,
                                const int64_t *sort_idx, const float *PCvox, const float vmin[3], double voxel_size,
                                float *PCsorted, float *DeltaPC, raht_stream_t stream)
{
    (void)stream;
    if (!PC || !keys_sorted || !sort_idx || !vmin || !DeltaPC || N < 1 || d < 0 || ldpc < 3 + d || !(voxel_size > 0) || (d > 0 && !PCvox)) {
        set_err("raht_cpu_voxelize_residuals: bad argument");
        return RAHT_ERR_INVALID;
    }
    const int ld = 3 + d;
    float *P = (float *)malloc(sizeof(float) * (size_t)N * (size_t)ld);
    int64_t *vi = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    if (!P || !vi) { free(P); free(vi); return RAHT_ERR_NOMEM; }
    for (int64_t i = 0; i < N; ++i) memcpy(P + i * ld, PC + i * ldpc, sizeof(float) * (size_t)ld);
    int64_t nv = 0;
    for (int64_t i = 0; i < N; ++i) if (i == 0 || keys_sorted[i] != keys_sorted[i - 1]) vi[nv++] = i;      /* voxelize_pc.py:114-118 */
    const int rc = orc_voxel_residuals(P, N, d, sort_idx, vi, nv, PCvox, vmin, voxel_size, PCsorted, DeltaPC);
    free(P); free(vi);
    return rc == 0 ? RAHT_OK : RAHT_ERR_INVALID;
}

/* raht_voxelize_all: the two calls in sequence (the product produces all outputs from one pass; the results are the same) */
int raht_cpu_voxelize_all(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in, double width_in,
                          int J, uint64_t *keys_sorted, int64_t *sort_idx, int64_t *voxel_indices, float *PCvox,
                          int64_t *Vvox, float *PCsorted, float *DeltaPC, int64_t *n_vox, float vmin_out[3],
                          double *width_out, double *voxel_size_out, raht_stream_t stream)
{
    if (!PC || N < 1 || d < 0 || !n_vox) { set_err("raht_cpu_voxelize_all: bad argument"); return RAHT_ERR_INVALID; }
    if (DeltaPC && d > 0 && !PCvox) { set_err("raht_cpu_voxelize_all: DeltaPC needs PCvox"); return RAHT_ERR_INVALID; }
    const int ld = 3 + d;
    uint64_t *ks = keys_sorted ? keys_sorted : (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N);
    int64_t *si = sort_idx ? sort_idx : (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    float vm[3]; double w = 0, vs = 0;
    int rc = (ks && si) ? raht_cpu_voxelize(PC, ldpc, N, d, vmin_in, width_in, J, ks, si, voxel_indices, PCvox, Vvox, n_vox, vm, &w, &vs, stream)
                        : RAHT_ERR_NOMEM;
    if (rc == RAHT_OK && DeltaPC) rc = raht_cpu_voxelize_residuals(PC, ldpc, N, d, ks, si, PCvox, vm, vs, PCsorted, DeltaPC, stream);
    else if (rc == RAHT_OK && PCsorted)
        for (int64_t k = 0; k < N; ++k) memcpy(PCsorted + k * ld, PC + si[k] * ldpc, sizeof(float) * (size_t)ld);      /* voxelize_pc.py:103-108 */
    if (rc == RAHT_OK) {
        if (vmin_out) memcpy(vmin_out, vm, sizeof(vm));
        if (width_out) *width_out = w;
        if (voxel_size_out) *voxel_size_out = vs;
    }
    if (!keys_sorted) free(ks);
    if (!sort_idx) free(si);
    return rc;
}

/* raht_voxelize_plan: voxelize, the keys of the voxels' first points, the plan from those keys (borrowed) */
int raht_cpu_voxelize_plan(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in, double width_in,
                           int J, uint64_t *voxel_keys, int64_t *voxel_indices, float *PCvox, int64_t *n_vox,
                           float vmin_out[3], double *width_out, double *voxel_size_out, raht_stream_t stream,
                           raht_cpu_plan **plan)
{
    if (!voxel_keys || !plan) { set_err("raht_cpu_voxelize_plan: NULL argument"); return RAHT_ERR_INVALID; }
    *plan = NULL;
    if (N < 1) { set_err("raht_cpu_voxelize_plan: bad argument"); return RAHT_ERR_INVALID; }
    uint64_t *ks = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N);
    int64_t *vi = voxel_indices ? voxel_indices : (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t nv = 0;
    int rc = (ks && vi) ? raht_cpu_voxelize(PC, ldpc, N, d, vmin_in, width_in, J, ks, NULL, vi, PCvox, NULL, &nv, vmin_out, width_out, voxel_size_out, stream)
                        : RAHT_ERR_NOMEM;
    if (rc == RAHT_OK) {
        for (int64_t v = 0; v < nv; ++v) voxel_keys[v] = ks[vi[v]];
        if (n_vox) *n_vox = nv;
        rc = raht_cpu_plan_create_from_keys_borrowed(voxel_keys, nv, 3 * J, NULL, stream, plan);
    }
    free(ks);
    if (!voxel_indices) free(vi);
    return rc;
}

int raht_cpu_morton(const int64_t *V, int64_t N, int J, uint64_t *keys, raht_stream_t stream)
{
    (void)stream;
    if (!V || !keys || N < 0 || J < 1 || J > 21) { set_err("raht_cpu_morton: bad argument"); return RAHT_ERR_INVALID; }
    return orc_morton(V, N, J, keys) == 0 ? RAHT_OK : RAHT_ERR_INVALID;
}

typedef struct { uint64_t k; int64_t i; } ki_t;
static int ki_cmp(const void *a, const void *b)
{
    const ki_t *x = (const ki_t *)a, *y = (const ki_t *)b;
    if (x->k != y->k) return x->k < y->k ? -1 : 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);          /* stable: ties keep their original order */
}
int raht_cpu_sort_keys(const uint64_t *keys_in, int64_t N, int nbits, uint64_t *keys_out, int64_t *idx_out, raht_stream_t stream)
{
    (void)stream;
    if (!keys_in || !keys_out || N < 0 || nbits < 1 || nbits > 64) { set_err("raht_cpu_sort_keys: bad argument"); return RAHT_ERR_INVALID; }
    ki_t *t = (ki_t *)malloc(sizeof(ki_t) * (size_t)(N > 0 ? N : 1));
    if (!t) return RAHT_ERR_NOMEM;
    const uint64_t mask = nbits >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << nbits) - 1);
    for (int64_t i = 0; i < N; ++i) { t[i].k = keys_in[i] & mask; t[i].i = i; }
    qsort(t, (size_t)N, sizeof(ki_t), ki_cmp);
    for (int64_t i = 0; i < N; ++i) { keys_out[i] = keys_in[t[i].i]; if (idx_out) idx_out[i] = t[i].i; }
    free(t);
    return RAHT_OK;
}

/* raht_*_batch: scene by scene (the product runs stage k of all scenes in one launch; the results are the same) */
int raht_cpu_fwd_batch(int n, raht_cpu_plan *const *plans, const float *const *C, const int64_t *ldc, int D,
                       float *const *T, const int64_t *ldt, raht_stream_t stream)
{
    if (n < 1 || !plans || !C || !ldc || !T || !ldt) { set_err("raht_cpu_fwd_batch: bad argument"); return RAHT_ERR_INVALID; }
    for (int i = 0; i < n; ++i) { const int rc = raht_cpu_fwd(plans[i], C[i], ldc[i], D, T[i], ldt[i], NULL, stream); if (rc != RAHT_OK) return rc; }
    return RAHT_OK;
}

int raht_cpu_inv_batch(int n, raht_cpu_plan *const *plans, const float *const *T, const int64_t *ldt, int D,
                       float *const *C, const int64_t *ldc, raht_stream_t stream)
{
    if (n < 1 || !plans || !C || !ldc || !T || !ldt) { set_err("raht_cpu_inv_batch: bad argument"); return RAHT_ERR_INVALID; }
    for (int i = 0; i < n; ++i) { const int rc = raht_cpu_inv(plans[i], T[i], ldt[i], D, C[i], ldc[i], stream); if (rc != RAHT_OK) return rc; }
    return RAHT_OK;
}

int raht_cpu_fwd_quant_batch(int n, raht_cpu_plan *const *plans, const float *const *C, const int64_t *ldc, int D,
                             const float *steps, int n_steps, int32_t *const *Q, const int64_t *ldq, raht_stream_t stream)
{
    if (n < 1 || !plans || !C || !ldc || !Q || !ldq) { set_err("raht_cpu_fwd_quant_batch: bad argument"); return RAHT_ERR_INVALID; }
    for (int i = 0; i < n; ++i) { const int rc = raht_cpu_fwd_quant(plans[i], C[i], ldc[i], D, steps, n_steps, Q[i], ldq[i], stream); if (rc != RAHT_OK) return rc; }
    return RAHT_OK;
}

int raht_cpu_dequant_inv_batch(int n, raht_cpu_plan *const *plans, const int32_t *const *Q, const int64_t *ldq, int D,
                               const float *steps, int n_steps, float *const *C, const int64_t *ldc, raht_stream_t stream)
{
    if (n < 1 || !plans || !C || !ldc || !Q || !ldq) { set_err("raht_cpu_dequant_inv_batch: bad argument"); return RAHT_ERR_INVALID; }
    for (int i = 0; i < n; ++i) { const int rc = raht_cpu_dequant_inv(plans[i], Q[i], ldq[i], D, steps, n_steps, C[i], ldc[i], stream); if (rc != RAHT_OK) return rc; }
    return RAHT_OK;
}
