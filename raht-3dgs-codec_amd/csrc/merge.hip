// merge.hip -- per-voxel Gaussian merge: replaces the reference's only CUDA kernel,
// merge_weighted_mean_kernel (reference cuda/merge_cluster.cu:2-111; host wrapper
// cuda/merge_cluster_wrapper.cu:11-116). Upstream of the RAHT path (test_voxelize_3dgs.py:247-257).
//
// CDNA4 design: the reference gives one THREAD per cluster and walks the members once for the
// geometry and then color_dim more times for the colours (merge_cluster.cu:97-110), every access a
// 4-byte strided load. Here one 64-lane WAVE owns a cluster and lanes map to the 11 + color_dim
// output columns (3 mean, 4 quat, 3 scale, 1 opacity, colours), so every member row is read once, as
// coalesced segments; a wave takes 16 clusters per iteration and keeps 8 first-member rows in
// flight (most voxels hold one Gaussian). Members are accumulated in index order with explicit
// fmaf, so results are bit-reproducible (oracle: orc_merge_clusters).
#include "raht_common.h"

#include <algorithm>

namespace raht {

struct MergeArgs {
    const int32_t *cluster_indices;
    const int32_t *cluster_offsets;
    int64_t num_clusters;
    const float *means, *quats, *scales, *opacities, *colors;
    int color_dim;
    int weight_by_opacity;
    float *m_means, *m_quats, *m_scales, *m_opacities, *m_colors;
};

__global__ __launch_bounds__(256) void merge_kernel(const MergeArgs A)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int ncol = 11 + A.color_dim;
    for (int c0 = 0; c0 < ncol; c0 += 64) {
        // this lane's column: which input array, row stride and offset inside the row
        const int col = c0 + lane;
        const bool act = col < ncol;
        const float *src; float *dst; int stride, off; int kind;   // kind 0 weighted mean, 1 quat, 2 opacity sum, 3 colour
        if (col < 3) { src = A.means; dst = A.m_means; stride = 3; off = col; kind = 0; }
        else if (col < 7) { src = A.quats; dst = A.m_quats; stride = 4; off = col - 3; kind = 1; }
        else if (col < 10) { src = A.scales; dst = A.m_scales; stride = 3; off = col - 7; kind = 0; }
        else if (col == 10) { src = A.opacities; dst = A.m_opacities; stride = 1; off = 0; kind = 2; }
        else { src = A.colors; dst = A.m_colors; stride = A.color_dim; off = act ? col - 11 : 0; kind = 3; }

        for (int64_t v0 = wave * 16; v0 < A.num_clusters; v0 += nwaves * 16) {
            const int64_t vi = v0 + lane;
            const int32_t st = (lane <= 16 && vi <= A.num_clusters) ? A.cluster_offsets[vi] : 0;
            for (int u0 = 0; u0 < 16; u0 += 8) {
                int s[8], e[8], first[8];
                float wfirst[8], xfirst[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    s[u] = __builtin_amdgcn_readlane(st, u0 + u);
                    e[u] = __builtin_amdgcn_readlane(st, u0 + u + 1);
                    const bool has = (v0 + u0 + u < A.num_clusters) && (e[u] > s[u]);
                    first[u] = has ? A.cluster_indices[s[u]] : 0;
                    if (!has) e[u] = s[u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    wfirst[u] = A.weight_by_opacity ? A.opacities[first[u]] : 1.0f;
                    xfirst[u] = act ? src[(int64_t)first[u] * stride + off] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t v = v0 + u0 + u;
                    if (v >= A.num_clusters) break;                          // wave-uniform
                    float acc = 0.0f, tw = 0.0f;
                    if (e[u] > s[u]) {                                       // merge_cluster.cu:38-63
                        tw = wfirst[u];
                        acc = (kind == 2) ? xfirst[u] : __fmaf_rn(xfirst[u], wfirst[u], 0.0f);
                        for (int i = s[u] + 1; i < e[u]; ++i) {
                            const int idx = A.cluster_indices[i];
                            const float w = A.weight_by_opacity ? A.opacities[idx] : 1.0f;
                            tw += w;
                            const float x = act ? src[(int64_t)idx * stride + off] : 0.0f;
                            acc = (kind == 2) ? acc + x : __fmaf_rn(x, w, acc);
                        }
                    }
                    // quaternion norm: lanes 3..6 of the first column chunk (merge_cluster.cu:76-78)
                    float n2 = 0.0f;
                    if (c0 == 0) {
                        const float qx = __shfl(acc, 3, 64), qy = __shfl(acc, 4, 64), qz = __shfl(acc, 5, 64), qw = __shfl(acc, 6, 64);
                        n2 = qx * qx;
                        n2 = __fmaf_rn(qy, qy, n2);
                        n2 = __fmaf_rn(qz, qz, n2);
                        n2 = __fmaf_rn(qw, qw, n2);
                    }
                    if (!act) continue;
                    float r;
                    if (e[u] <= s[u]) r = 0.0f;                              // empty cluster: outputs stay zero (wrapper :66-70)
                    else if (kind == 0) r = __fdiv_rn(acc, tw == 0.0f ? 1.0f : tw);      // :66-73, :91-93
                    else if (kind == 1) {                                    // :76-89
                        const float nrm = __fsqrt_rn(n2);
                        r = (nrm > 0.0f) ? __fdiv_rn(acc, nrm) : (off == 3 ? 1.0f : 0.0f);
                    } else if (kind == 2) r = fminf(acc, 1.0f);              // :96
                    else r = (tw > 0.0f) ? __fdiv_rn(acc, tw) : 0.0f;        // :98-110
                    dst[v * stride + off] = r;
                }
            }
        }
    }
}

}  // namespace raht

using namespace raht;

extern "C" {

int raht_merge_clusters(const int32_t *cluster_indices, const int32_t *cluster_offsets, int64_t num_clusters,
                        const float *means, const float *quats, const float *scales, const float *opacities,
                        const float *colors, int color_dim, int weight_by_opacity, float *merged_means,
                        float *merged_quats, float *merged_scales, float *merged_opacities, float *merged_colors,
                        raht_stream_t stream)
{
    if (num_clusters < 0 || color_dim < 0 || !cluster_offsets || (num_clusters > 0 && (!cluster_indices || !means || !quats || !scales ||
        !opacities || (color_dim > 0 && !colors) || !merged_means || !merged_quats || !merged_scales || !merged_opacities ||
        (color_dim > 0 && !merged_colors)))) {
        set_error("raht_merge_clusters: bad argument");
        return RAHT_ERR_INVALID;
    }
    if (num_clusters == 0) return RAHT_OK;
    MergeArgs A;
    A.cluster_indices = cluster_indices; A.cluster_offsets = cluster_offsets; A.num_clusters = num_clusters;
    A.means = means; A.quats = quats; A.scales = scales; A.opacities = opacities; A.colors = colors;
    A.color_dim = color_dim; A.weight_by_opacity = weight_by_opacity ? 1 : 0;
    A.m_means = merged_means; A.m_quats = merged_quats; A.m_scales = merged_scales; A.m_opacities = merged_opacities;
    A.m_colors = merged_colors;
    const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(num_clusters, 64), 8192);
    hipLaunchKernelGGL(merge_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, A);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

}  // extern "C"
