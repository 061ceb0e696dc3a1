/*
 * raht_oracle.c -- CPU restatement of the reference RAHT hot path. TEST INFRASTRUCTURE ONLY:
 * see raht_oracle.h for the rule on who may load this. Scalar, single-threaded, float64 like the
 * reference (python/encode_3dgs.py:82-83 runs everything in torch.float64).
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks every function here against golden
 * vectors produced by the reference's own Python (tests/golden/gen_golden.py).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no FMA contraction so that float64 results
 * are the same IEEE operations torch performs: mul, mul, add).
 */
#include "raht_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

struct orc_param {
    int nlevels;        /* == len(Flags) == len(weights) == len(List) in the reference */
    int64_t N;
    int64_t *len;       /* len[l]   = List[l].numel()                                   */
    int64_t **list;     /* list[l]  = List[l]   (node start rows, 0-based)              */
    uint8_t **flags;    /* flags[l] = Flags[l]  (node k has a right sibling k+1)        */
    int64_t **weights;  /* weights[l] = run lengths                                     */
    int64_t order_len;  /* -1 == None */
    int64_t *order;
    uint64_t *mc;
};

/* ---------------------------------------------------------------- voxelize_pc.py:25-59 */
int orc_morton(const int64_t *V, int64_t N, int J, uint64_t *mc)
{
    if (J < 0 || J > 21) return -1;
    for (int64_t n = 0; n < N; ++n) {
        uint64_t m = 0;
        for (int i = 1; i <= J; ++i) {                       /* :46 */
            uint64_t bx = (uint64_t)(V[3 * n + 0] >> (i - 1)) & 1u;  /* :48 */
            uint64_t by = (uint64_t)(V[3 * n + 1] >> (i - 1)) & 1u;
            uint64_t bz = (uint64_t)(V[3 * n + 2] >> (i - 1)) & 1u;
            uint64_t digit = bz + (by << 1) + (bx << 2);      /* :52-54 */
            m |= digit << (3 * (i - 1));                      /* :57 */
        }
        mc[n] = m;
    }
    return 0;
}

/* ---------------------------------------------------------------- RAHT_param.py:190-279 */
static int push_level(orc_param *p, const int64_t *list, const uint8_t *flags, const int64_t *w,
                      int64_t n)
{
    int l = p->nlevels;
    p->len = (int64_t *)realloc(p->len, sizeof(int64_t) * (size_t)(l + 1));
    p->list = (int64_t **)realloc(p->list, sizeof(int64_t *) * (size_t)(l + 1));
    p->flags = (uint8_t **)realloc(p->flags, sizeof(uint8_t *) * (size_t)(l + 1));
    p->weights = (int64_t **)realloc(p->weights, sizeof(int64_t *) * (size_t)(l + 1));
    size_t nn = (size_t)(n > 0 ? n : 1);
    p->list[l] = (int64_t *)malloc(sizeof(int64_t) * nn);
    p->flags[l] = (uint8_t *)malloc(nn);
    p->weights[l] = (int64_t *)malloc(sizeof(int64_t) * nn);
    memcpy(p->list[l], list, sizeof(int64_t) * (size_t)n);
    memcpy(p->flags[l], flags, (size_t)n);
    memcpy(p->weights[l], w, sizeof(int64_t) * (size_t)n);
    p->len[l] = n;
    p->nlevels = l + 1;
    return 0;
}

/* nonzero(mask) appended to the ac_list store */
typedef struct { int64_t *v; int64_t n; } group_t;

static group_t nonzero_xor(const uint8_t *a, const uint8_t *b, int64_t N, int invert_a_only)
{
    group_t g; g.n = 0;
    g.v = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    for (int64_t i = 0; i < N; ++i) {
        int bit = invert_a_only ? !a[i] : (a[i] ^ b[i]);
        if (bit) g.v[g.n++] = i;
    }
    return g;
}

int orc_param_build(const double *V, int64_t N, const double minV[3], double width, int depth,
                    int ref_quirks, orc_param **out)
{
    if (N < 1 || depth < 1 || depth > 21) return -1;
    orc_param *p = (orc_param *)calloc(1, sizeof(orc_param));
    p->N = N;
    p->order_len = -1;

    /* :205-212  Q, Vint, Morton (digit = z + 2y + 4x) */
    const double Q = width / (double)((uint64_t)1 << depth);
    int64_t *Vint = (int64_t *)malloc(sizeof(int64_t) * 3 * (size_t)N);
    for (int64_t n = 0; n < N; ++n)
        for (int a = 0; a < 3; ++a)
            Vint[3 * n + a] = (int64_t)floor((V[3 * n + a] - minV[a]) / Q);
    p->mc = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N);
    orc_morton(Vint, N, depth, p->mc);
    free(Vint);
    const uint64_t *MC = p->mc;

    const int Nbits = 3 * depth;                                  /* :216 */
    int64_t *curr = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *w = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    uint8_t *flag = (uint8_t *)malloc((size_t)N);
    uint8_t *indices = (uint8_t *)calloc((size_t)N, 1);           /* :220 */
    uint8_t *pre_indices = (uint8_t *)calloc((size_t)N, 1);       /* :221 */
    int64_t ncur = N;
    for (int64_t i = 0; i < N; ++i) curr[i] = i;                  /* :214 */

    group_t ac[80]; int nac = 0;                                  /* ac_list */
    int have_order = 0;

    for (int j = 1; j <= 64; ++j) {                               /* :226 */
        for (int64_t k = 0; k < ncur; ++k)                        /* :228-230 */
            w[k] = (k + 1 < ncur ? curr[k + 1] : N) - curr[k];
        if (ncur == 1) {                                          /* :234-236 */
            flag[0] = 0;
            push_level(p, curr, flag, w, 1);
            break;
        }
        /* :224 mask_table[j-1] = 2^Nbits - 2^j  (int64 wrap-around arithmetic) */
        const uint64_t mask = ((uint64_t)1 << Nbits) - (j < 64 ? ((uint64_t)1 << j) : 0);
        for (int64_t k = 0; k + 1 < ncur; ++k)                    /* :238-243 */
            flag[k] = (((MC[curr[k]] ^ MC[curr[k + 1]]) & mask) == 0);
        flag[ncur - 1] = 0;
        push_level(p, curr, flag, w, ncur);

        int64_t ntmp = 0;                                         /* :246-248 */
        for (int64_t k = 0; k < ncur; ++k) {
            int prev = (k > 0) ? flag[k - 1] : 0;
            if (!prev) tmp[ntmp++] = curr[k];
        }

        if (j % 3 == 0 && j > 2) {                                /* :251-262 */
            memset(indices, 0, (size_t)N);
            for (int64_t k = 0; k < ntmp; ++k) indices[tmp[k]] = 1;
            if (j == 3) ac[nac++] = nonzero_xor(indices, NULL, N, 1);
            else        ac[nac++] = nonzero_xor(indices, pre_indices, N, 0);
            memcpy(pre_indices, indices, (size_t)N);
        }

        if (ntmp == 1 || j >= Nbits) {                            /* :265-274 */
            memset(indices, 0, (size_t)N);
            for (int64_t k = 0; k < ntmp; ++k) indices[tmp[k]] = 1;
            if (!ref_quirks && j < 3) {
                /* intended behaviour: everything that is not a surviving root */
                ac[nac++] = nonzero_xor(indices, NULL, N, 1);
            } else {
                ac[nac++] = nonzero_xor(indices, pre_indices, N, 0);
            }
            group_t root; root.n = ntmp;
            root.v = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ntmp > 0 ? ntmp : 1));
            memcpy(root.v, tmp, sizeof(int64_t) * (size_t)ntmp);
            ac[nac++] = root;
            have_order = 1;
            break;
        }
        memcpy(curr, tmp, sizeof(int64_t) * (size_t)ntmp);        /* :276-277 */
        ncur = ntmp;
    }

    if (have_order) {                                             /* :272-273 reversed cat */
        int64_t tot = 0;
        for (int g = 0; g < nac; ++g) tot += ac[g].n;
        p->order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(tot > 0 ? tot : 1));
        int64_t o = 0;
        for (int g = nac - 1; g >= 0; --g) {
            memcpy(p->order + o, ac[g].v, sizeof(int64_t) * (size_t)ac[g].n);
            o += ac[g].n;
        }
        p->order_len = tot;
    } else if (!ref_quirks) {                                     /* N == 1 */
        p->order = (int64_t *)malloc(sizeof(int64_t));
        p->order[0] = 0;
        p->order_len = 1;
    }
    for (int g = 0; g < nac; ++g) free(ac[g].v);
    free(curr); free(tmp); free(w); free(flag); free(indices); free(pre_indices);
    *out = p;
    return 0;
}

int orc_param_levels(const orc_param *p) { return p->nlevels; }
int64_t orc_param_level_len(const orc_param *p, int l) { return p->len[l]; }
const int64_t *orc_param_list(const orc_param *p, int l) { return p->list[l]; }
const uint8_t *orc_param_flags(const orc_param *p, int l) { return p->flags[l]; }
const int64_t *orc_param_weights(const orc_param *p, int l) { return p->weights[l]; }
int64_t orc_param_order_len(const orc_param *p) { return p->order_len; }
const int64_t *orc_param_order(const orc_param *p) { return p->order; }
const uint64_t *orc_param_morton(const orc_param *p) { return p->mc; }

void orc_param_free(orc_param *p)
{
    if (!p) return;
    for (int l = 0; l < p->nlevels; ++l) { free(p->list[l]); free(p->flags[l]); free(p->weights[l]); }
    free(p->list); free(p->flags); free(p->weights); free(p->len); free(p->order); free(p->mc);
    free(p);
}

/* ---------------------------------------------------------------- RAHT.py:252-336 / iRAHT.py:40-114
 * One level = gather ALL sibling rows, butterfly, then scatter (the reference's index_select /
 * scatter_ batch semantics: every read of a level happens before any write of that level). */
static int level_pass(double *T, double *wv, int D, const orc_param *p, int l, int inverse)
{
    const int64_t n = p->len[l];
    const int64_t *list = p->list[l];
    const uint8_t *fl = p->flags[l];
    const int64_t *wt = p->weights[l];
    int64_t M = 0;
    for (int64_t k = 0; k < n; ++k) M += fl[k] ? 1 : 0;       /* left_mask.sum() */
    if (M == 0) return 0;                                      /* RAHT.py:304-305 */
    /* right_mask = [False, Flags[:-1]]  ->  k-th left pairs with the k-th right (RAHT.py:297-302) */
    int64_t *i0 = (int64_t *)malloc(sizeof(int64_t) * (size_t)M);
    int64_t *i1 = (int64_t *)malloc(sizeof(int64_t) * (size_t)M);
    double *a = (double *)malloc(sizeof(double) * (size_t)M);
    double *b = (double *)malloc(sizeof(double) * (size_t)M);
    int64_t m0 = 0, m1 = 0;
    double *w0 = (double *)malloc(sizeof(double) * (size_t)M);
    double *w1 = (double *)malloc(sizeof(double) * (size_t)M);
    for (int64_t k = 0; k < n; ++k) {
        if (fl[k]) { i0[m0] = list[k]; w0[m0] = (double)wt[k]; ++m0; }
        if (k > 0 && fl[k - 1]) { i1[m1] = list[k]; w1[m1] = (double)wt[k]; ++m1; }
    }
    for (int64_t m = 0; m < M; ++m) {                          /* RAHT.py:317-322 */
        double denom = w0[m] + w1[m];
        a[m] = sqrt(w0[m] / denom);
        b[m] = sqrt(w1[m] / denom);
    }
    double *x0 = (double *)malloc(sizeof(double) * (size_t)M * (size_t)D);
    double *x1 = (double *)malloc(sizeof(double) * (size_t)M * (size_t)D);
    for (int64_t m = 0; m < M; ++m) {                          /* RAHT.py:311-312 */
        memcpy(x0 + m * D, T + i0[m] * D, sizeof(double) * (size_t)D);
        memcpy(x1 + m * D, T + i1[m] * D, sizeof(double) * (size_t)D);
    }
    if (wv && !inverse) {                                      /* RAHT.py:325-328 */
        double *nw = (double *)malloc(sizeof(double) * (size_t)M);
        for (int64_t m = 0; m < M; ++m) nw[m] = wv[i0[m]] + wv[i1[m]];
        for (int64_t m = 0; m < M; ++m) wv[i0[m]] = nw[m];
        for (int64_t m = 0; m < M; ++m) wv[i1[m]] = nw[m];
        free(nw);
    }
    for (int64_t m = 0; m < M; ++m) {
        double *r0 = T + i0[m] * D, *r1 = T + i1[m] * D;
        const double am = a[m], bm = b[m];
        if (!inverse) {                                        /* RAHT.py:331-334 */
            for (int c = 0; c < D; ++c) r0[c] = am * x0[m * D + c] + bm * x1[m * D + c];
        } else {                                               /* iRAHT.py:108,111 */
            for (int c = 0; c < D; ++c) r0[c] = am * x0[m * D + c] - bm * x1[m * D + c];
        }
        (void)r1;
    }
    for (int64_t m = 0; m < M; ++m) {
        double *r1 = T + i1[m] * D;
        const double am = a[m], bm = b[m];
        if (!inverse) {
            for (int c = 0; c < D; ++c) r1[c] = (-bm) * x0[m * D + c] + am * x1[m * D + c];
        } else {                                               /* iRAHT.py:109,112 */
            for (int c = 0; c < D; ++c) r1[c] = bm * x0[m * D + c] + am * x1[m * D + c];
        }
    }
    free(i0); free(i1); free(a); free(b); free(w0); free(w1); free(x0); free(x1);
    return 0;
}

int orc_raht_fwd(const double *C, int64_t N, int D, const orc_param *p, double *T, double *w)
{
    if (N != p->N) return -1;
    memcpy(T, C, sizeof(double) * (size_t)N * (size_t)D);       /* RAHT.py:285 */
    if (w) for (int64_t i = 0; i < N; ++i) w[i] = 1.0;          /* RAHT.py:286 */
    for (int l = 0; l < p->nlevels; ++l) level_pass(T, w, D, p, l, 0);   /* RAHT.py:293 */
    return 0;
}

int orc_raht_inv(const double *T, int64_t N, int D, const orc_param *p, double *C)
{
    if (N != p->N) return -1;
    memcpy(C, T, sizeof(double) * (size_t)N * (size_t)D);       /* iRAHT.py:69 */
    for (int l = p->nlevels - 1; l >= 0; --l) level_pass(C, NULL, D, p, l, 1);  /* iRAHT.py:76 */
    return 0;
}

/* ---------------------------------------------------------------- encode_3dgs.py:204-217,261-268 */
int orc_quant_reorder(const double *T, int64_t N, int D, double step, const int64_t *order, int32_t *Q)
{
    for (int64_t k = 0; k < N; ++k) {
        const double *src = T + order[k] * D;                   /* :210 index_select(0, order) */
        for (int c = 0; c < D; ++c)
            Q[k * D + c] = (int32_t)floor(src[c] / step + 0.5); /* :204, :215 */
    }
    return 0;
}

int orc_dequant_unreorder(const int32_t *Q, int64_t N, int D, double step, const int64_t *order, double *T)
{
    /* :267-268  Coeff_dec[argsort(order)]  <=>  T[order[k]] = Qf[k]  when order is a permutation */
    for (int64_t k = 0; k < N; ++k) {
        double *dst = T + order[k] * D;
        for (int c = 0; c < D; ++c) dst[c] = (double)Q[k * D + c] * step;   /* :261 */
    }
    return 0;
}

/* ---------------------------------------------------------------- voxelize_pc.py:62-172 */
static void stable_sort_u64(const uint64_t *keys, int64_t N, uint64_t *keys_out, int64_t *idx_out)
{
    uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(N > 0 ? N : 1));
    uint64_t *kb = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(N > 0 ? N : 1));
    int64_t *ia = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    int64_t *ib = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    memcpy(ka, keys, sizeof(uint64_t) * (size_t)N);
    for (int64_t i = 0; i < N; ++i) ia[i] = i;
    for (int pass = 0; pass < 8; ++pass) {
        int64_t cnt[257]; memset(cnt, 0, sizeof(cnt));
        const int sh = 8 * pass;
        for (int64_t i = 0; i < N; ++i) cnt[((ka[i] >> sh) & 0xff) + 1]++;
        for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < N; ++i) {
            int64_t pos = cnt[(ka[i] >> sh) & 0xff]++;
            kb[pos] = ka[i]; ib[pos] = ia[i];
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        int64_t *ti = ia; ia = ib; ib = ti;
    }
    memcpy(keys_out, ka, sizeof(uint64_t) * (size_t)N);
    memcpy(idx_out, ia, sizeof(int64_t) * (size_t)N);
    free(ka); free(kb); free(ia); free(ib);
}

int orc_voxelize(const float *PC, int64_t N, int d, const float *vmin_in, double width_in, int J,
                 uint64_t *keys_sorted, int64_t *sort_idx, int64_t *voxel_indices, float *PCvox,
                 int64_t *Vvox, int64_t *Nvox, float vmin_out[3], double *width_out,
                 double *voxel_size_out)
{
    if (N < 1 || J < 1 || J > 21) return -1;
    const int ld = 3 + d;
    float vmin[3];
    if (vmin_in) { vmin[0] = vmin_in[0]; vmin[1] = vmin_in[1]; vmin[2] = vmin_in[2]; }
    else {                                                       /* :88 */
        for (int a = 0; a < 3; ++a) {
            float m = PC[a];
            for (int64_t n = 1; n < N; ++n) { float v = PC[n * ld + a]; if (v < m) m = v; }
            vmin[a] = m;
        }
    }
    double width = width_in;
    if (width_in < 0) {                                          /* :95  V0.max().item() */
        float mx = PC[0] - vmin[0];
        for (int64_t n = 0; n < N; ++n)
            for (int a = 0; a < 3; ++a) { float v = PC[n * ld + a] - vmin[a]; if (v > mx) mx = v; }
        width = (double)mx;
    }
    const double voxel_size = width / (double)((uint64_t)1 << J);   /* :97 python float */
    const float vs = (float)voxel_size;       /* tensor(float32) / python scalar -> float32 divide */
    const int64_t hi = ((int64_t)1 << J) - 1;
    int64_t *Vint = (int64_t *)malloc(sizeof(int64_t) * 3 * (size_t)N);
    for (int64_t n = 0; n < N; ++n)
        for (int a = 0; a < 3; ++a) {
            float v0 = PC[n * ld + a] - vmin[a];                 /* :92 */
            int64_t q = (int64_t)floorf(v0 / vs);                /* :98 */
            if (q < 0) q = 0;
            if (q > hi) q = hi;
            Vint[3 * n + a] = q;
        }
    uint64_t *M = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)N);
    orc_morton(Vint, N, J, M);                                   /* :100 */
    stable_sort_u64(M, N, keys_sorted, sort_idx);                /* :101 (stable restatement) */
    free(M);

    int64_t nv = 0;                                              /* :114-118 */
    for (int64_t i = 0; i < N; ++i)
        if (i == 0 || keys_sorted[i] != keys_sorted[i - 1]) voxel_indices[nv++] = i;
    *Nvox = nv;

    for (int64_t v = 0; v < nv; ++v) {
        const int64_t s = voxel_indices[v], e = (v + 1 < nv) ? voxel_indices[v + 1] : N;
        const int64_t src = sort_idx[s];
        for (int a = 0; a < 3; ++a) {                            /* :152/:159, :155 */
            Vvox[3 * v + a] = Vint[3 * src + a];
            PCvox[v * ld + a] = (float)Vint[3 * src + a];
        }
        const float cnt = (float)(e - s);                        /* :137 */
        for (int c = 0; c < d; ++c) {                            /* :140-144, sequential in sorted order */
            float acc = 0.0f;
            for (int64_t i = s; i < e; ++i) acc += PC[sort_idx[i] * ld + 3 + c];
            PCvox[v * ld + 3 + c] = acc / cnt;
        }
    }
    free(Vint);
    vmin_out[0] = vmin[0]; vmin_out[1] = vmin[1]; vmin_out[2] = vmin[2];
    *width_out = width;
    *voxel_size_out = voxel_size;
    return 0;
}

/* voxelize_pc.py:103-111, 147-156: the secondary outputs of the voxelizer, from its primary ones.
 * PCsorted[k] = PC[sort_idx[k]];  DeltaV = V0 - voxel_size * floor(V0 / voxel_size), V0 = V - vmin (float32
 * arithmetic, the Python-float voxel_size rounded to float32 as torch does for tensor / scalar);
 * DeltaC = C0 - Cvox[voxel of k]. */
int orc_voxel_residuals(const float *PC, int64_t N, int d, const int64_t *sort_idx, const int64_t *voxel_indices,
                        int64_t Nvox, const float *PCvox, const float vmin[3], double voxel_size, float *PCsorted,
                        float *DeltaPC)
{
    const int ld = 3 + d;
    const float vs = (float)voxel_size;
    int64_t v = 0;
    for (int64_t k = 0; k < N; ++k) {
        while (v + 1 < Nvox && voxel_indices[v + 1] <= k) ++v;            /* :129-132 voxel_id */
        const float *src = PC + sort_idx[k] * ld;
        if (PCsorted) memcpy(PCsorted + k * ld, src, sizeof(float) * (size_t)ld);      /* :103-108 */
        for (int a = 0; a < 3; ++a) {
            const float v0 = src[a] - vmin[a];                            /* :92, :103 */
            const float vox = vs * floorf(v0 / vs);                       /* :110 */
            DeltaPC[k * ld + a] = v0 - vox;                               /* :111 */
        }
        for (int c = 0; c < d; ++c) DeltaPC[k * ld + 3 + c] = src[3 + c] - PCvox[v * ld + 3 + c];   /* :147-148 */
    }
    return 0;
}

/* ---------------------------------------------------------------- PyRLGR membuf.cpp
 * Literal restatement of the bit buffer (membuf.cpp:74-189) and of the coder (:228-423). */
#define ORC_L 4
#define ORC_U0 3
#define ORC_D0 1
#define ORC_U1 2
#define ORC_D1 1
#define ORC_MASK(k) ((((uint64_t)1) << (k)) - 1)

typedef struct {
    uint64_t data;
    uint8_t bits;
    uint8_t *buf;          /* write: output, read: input */
    int64_t cap, size, pos;
    int overflow;
} orc_membuf;

static void mb_flush(orc_membuf *m)                         /* :74-86 */
{
    while (m->bits >= 8) {
        m->bits -= 8;
        uint8_t byte = (uint8_t)((m->data >> m->bits) & 0xff);
        if (m->size < m->cap) m->buf[m->size] = byte; else m->overflow = 1;
        m->size++;
        m->pos++;
    }
}

static void mb_fill(orc_membuf *m)                          /* :89-104 */
{
    while (m->bits <= 56) {
        if (m->pos < m->size) {
            uint8_t byte = m->buf[m->pos++];
            m->data = (m->data << 8) + byte;
            m->bits += 8;
        } else return;
    }
}

static uint8_t mb_read1(orc_membuf *m)                      /* :106-112 */
{
    if (!m->bits) mb_fill(m);
    m->bits--;
    return (uint8_t)((m->data >> m->bits) & 1u);
}

static uint64_t mb_read(orc_membuf *m, uint8_t bits)        /* :114-126 */
{
    if (bits > 56) {
        uint64_t d = mb_read(m, (uint8_t)(bits - 32)) << 32;
        return d + mb_read(m, 32);
    }
    mb_fill(m);
    m->bits -= bits;
    return (m->data >> m->bits) & ORC_MASK(bits);
}

static void mb_write1(orc_membuf *m, uint8_t bit)           /* :161-170 */
{
    m->data <<= 1;
    if (bit) m->data++;
    m->bits++;
    if (m->bits >= 8) mb_flush(m);
}

static void mb_write(orc_membuf *m, uint64_t data, uint8_t bits)   /* :172-184 */
{
    if (bits > 56) {
        mb_write(m, data >> 32, (uint8_t)(bits - 32));
        mb_write(m, data & ORC_MASK(32), 32);
        return;
    }
    m->data = (m->data << bits) + data;
    m->bits += bits;
    mb_flush(m);
}

static void mb_close(orc_membuf *m)                         /* :47-58 */
{
    uint8_t r = m->bits % 8;
    if (r) mb_write(m, 0, (uint8_t)(8 - r)); else mb_flush(m);
}

static uint64_t mb_gr_read(orc_membuf *m, uint8_t bits)     /* :228-240 */
{
    uint64_t p = 0;
    while (mb_read1(m)) {
        p++;
        if (p >= 32) return mb_read(m, 32);
    }
    return (p << bits) + mb_read(m, bits);
}

static void mb_gr_write(orc_membuf *m, uint64_t data, uint8_t bits)   /* :242-256 */
{
    uint64_t p = data >> bits;
    if (p < 32) {
        mb_write(m, ORC_MASK(p + 1) - 1, (uint8_t)(p + 1));
        mb_write(m, data & ORC_MASK(bits), bits);
    } else {
        mb_write(m, ORC_MASK(32), 32);
        mb_write(m, data, 32);
    }
}

static uint64_t s2u(int64_t v) { return v < 0 ? (((uint64_t)(-v)) << 1) - 1 : ((uint64_t)v) << 1; }   /* :4-14 */
static int64_t u2s(uint64_t v) { int64_t d = (int64_t)(v >> 1); return (v & 1) ? -d - 1 : d; }          /* :16-23 */

#define ADAPT_KRP(p)                                                         \
    do {                                                                     \
        if (p) { k_RP += (p) - 1; if (k_RP > 32 * ORC_L) k_RP = 32 * ORC_L; } \
        else { if (k_RP < 2) k_RP = 0; else k_RP -= 2; }                     \
    } while (0)

int64_t orc_rlgr_encode(const int64_t *seq, int64_t N, int flag_signed, uint8_t *out, int64_t cap)
{
    orc_membuf mb; memset(&mb, 0, sizeof(mb));
    mb.buf = out; mb.cap = cap;
    uint64_t u = 0, k_P = 0, k_RP = 2 * ORC_L, m = 0, k = 0, k_R, p;        /* :343-349 */
    for (int64_t n = 0; n < N; ++n) {                                      /* :351 */
        u = flag_signed ? s2u(seq[n]) : (uint64_t)seq[n];
        k = k_P / ORC_L;
        k_R = k_RP / ORC_L;
        if (k) {                                                           /* run mode :356-383 */
            if (u) {
                u--;
                mb_write1(&mb, 0);
                mb_write(&mb, m, (uint8_t)k);
                mb_gr_write(&mb, u, (uint8_t)k_R);
                p = u >> k_R;
                ADAPT_KRP(p);
                if (k_P < ORC_D1) k_P = 0; else k_P -= ORC_D1;
                m = 0;
            } else {
                m++;
                if (m == ((uint64_t)1 << k)) {
                    mb_write1(&mb, 1);
                    k_P += ORC_U1;
                    m = 0;
                }
            }
        } else {                                                           /* no-run mode :384-407 */
            mb_gr_write(&mb, u, (uint8_t)k_R);
            p = u >> k_R;
            ADAPT_KRP(p);
            if (u) { if (k_P < ORC_D0) k_P = 0; else k_P -= ORC_D0; }
            else k_P += ORC_U0;
            m = 0;
        }
    }
    if (k && !u) {                                                         /* :410-413 */
        mb_write1(&mb, 0);
        mb_write(&mb, m, (uint8_t)(k_P / ORC_L));
    }
    mb_close(&mb);
    return mb.overflow ? -1 : mb.size;
}

int orc_rlgr_decode(const uint8_t *buf, int64_t nbytes, int64_t N, int flag_signed, int64_t *seq)
{
    orc_membuf mb; memset(&mb, 0, sizeof(mb));
    mb.buf = (uint8_t *)buf; mb.size = nbytes;
    uint64_t u, k_P = 0, k_RP = 2 * ORC_L, m = 0, k, k_R, p;                /* :261-268 */
    int64_t n = 0;
    while (n < N) {                                                        /* :270 */
        k = k_P / ORC_L;
        k_R = k_RP / ORC_L;
        if (k) {                                                           /* run mode :274-307 */
            m = 0;
            while (mb_read1(&mb)) {
                m += (uint64_t)1 << k;
                k_P += ORC_U1;
                k = k_P / ORC_L;
            }
            m += mb_read(&mb, (uint8_t)k);
            while (m-- && n < N) seq[n++] = 0;
            if (n >= N) break;
            u = mb_gr_read(&mb, (uint8_t)k_R);
            seq[n++] = flag_signed ? u2s(u + 1) : (int64_t)(u + 1);
            p = u >> k_R;
            ADAPT_KRP(p);
            if (k_P < ORC_D1) k_P = 0; else k_P -= ORC_D1;
        } else {                                                           /* no-run mode :308-329 */
            u = mb_gr_read(&mb, (uint8_t)k_R);
            seq[n++] = flag_signed ? u2s(u) : (int64_t)u;
            p = u >> k_R;
            ADAPT_KRP(p);
            if (u) { if (k_P < ORC_D0) k_P = 0; else k_P -= ORC_D0; }
            else k_P += ORC_U0;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- cuda/merge_cluster.cu:2-111
 * PARITY UNPINNED (see raht_oracle.h): restated from the kernel text only. */
int orc_merge_clusters(const int32_t *ci, const int32_t *co, int64_t K, const float *means, const float *quats,
                       const float *scales, const float *opac, const float *colors, int cd, int wbo, float *mm,
                       float *mq, float *ms, float *mo, float *mc)
{
    for (int64_t c = 0; c < K; ++c) {
        const int start = co[c], end = co[c + 1];
        if (end - start == 0) {                                   /* :26 -- outputs stay zero */
            for (int k = 0; k < 3; ++k) { mm[c * 3 + k] = 0; ms[c * 3 + k] = 0; }
            for (int k = 0; k < 4; ++k) mq[c * 4 + k] = 0;
            mo[c] = 0;
            for (int k = 0; k < cd; ++k) mc[c * cd + k] = 0;
            continue;
        }
        float ma[3] = {0, 0, 0}, qa[4] = {0, 0, 0, 0}, sa[3] = {0, 0, 0}, osum = 0, tw = 0;
        for (int i = start; i < end; ++i) {                       /* :38-63 */
            const int idx = ci[i];
            const float w = wbo ? opac[idx] : 1.0f;
            tw += w;
            for (int k = 0; k < 3; ++k) ma[k] = fmaf(means[idx * 3 + k], w, ma[k]);
            for (int k = 0; k < 4; ++k) qa[k] = fmaf(quats[idx * 4 + k], w, qa[k]);
            for (int k = 0; k < 3; ++k) sa[k] = fmaf(scales[idx * 3 + k], w, sa[k]);
            osum += opac[idx];
        }
        const float twd = (tw == 0.0f) ? 1.0f : tw;               /* :66-68 */
        for (int k = 0; k < 3; ++k) mm[c * 3 + k] = ma[k] / twd;  /* :71-73 */
        float n2 = qa[0] * qa[0];                                 /* :76-77 */
        n2 = fmaf(qa[1], qa[1], n2); n2 = fmaf(qa[2], qa[2], n2); n2 = fmaf(qa[3], qa[3], n2);
        const float nrm = sqrtf(n2);
        if (nrm > 0.0f) for (int k = 0; k < 4; ++k) mq[c * 4 + k] = qa[k] / nrm;    /* :78-83 */
        else { mq[c * 4 + 0] = 0; mq[c * 4 + 1] = 0; mq[c * 4 + 2] = 0; mq[c * 4 + 3] = 1.0f; }   /* :84-89 */
        for (int k = 0; k < 3; ++k) ms[c * 3 + k] = sa[k] / twd;  /* :91-93 */
        mo[c] = fminf(osum, 1.0f);                                /* :96 */
        for (int k = 0; k < cd; ++k) {                            /* :98-110 */
            float acc = 0.0f;
            for (int i = start; i < end; ++i) {
                const int idx = ci[i];
                const float w = wbo ? opac[idx] : 1.0f;
                acc = fmaf(colors[(int64_t)idx * cd + k], w, acc);
            }
            mc[c * cd + k] = (tw > 0.0f) ? acc / tw : 0.0f;
        }
    }
    return 0;
}
