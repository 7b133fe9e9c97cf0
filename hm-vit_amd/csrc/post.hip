// Detection post-processing of the HM-ViT pipeline, all on the device:
//
//   k_box_decode   VoxelPostprocessor.post_process up to the candidate list
//                  (opencood/data_utils/post_processor/voxel_postprocessor.py:232-330, delta_to_boxes3d :355-396;
//                  opencood/utils/box_utils.py boxes_to_corners_3d :139-184, project_box3d :258-296,
//                  remove_large_pred_bbx :722-751, remove_bbx_abnormal_z :754-772): sigmoid + score threshold,
//                  anchor deltas -> (x, y, z, h, w, l, yaw), the 8 corners, optional rigid projection, the two
//                  sanity filters; survivors are appended to a compact list (order fixed later by the NMS ranking).
//   k_rank_scores  position of every candidate in the descending-score order (ties by anchor index)
//   k_quad_iou     IoU of convex quadrilaterals (first 4 corners, x/y) - the reference's shapely polygon IoU
//                  (opencood/utils/common_utils.py:120-158) restated as Sutherland-Hodgman clipping
//   k_nms_greedy   box_utils.nms_rotated (:575-620): top-1000 by score, greedy suppression at IoU > threshold,
//                  and the final GT_RANGE mask of get_mask_for_boxes_within_range_torch (:326-357)
#include "common.hpp"
#include "kernels.hpp"

namespace hmvit {

// ------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_box_decode(BoxDecodeParams p) {
    const int n_anchor = p.H * p.W * p.A;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_anchor) return;
    const int a = idx % p.A, pix = idx / p.A;
    const size_t HW = (size_t)p.H * p.W;
    const float logit = p.psm[(size_t)a * HW + pix];
    const float prob = 1.f / (1.f + expf(-logit));
    if (!(prob > p.thresh)) return;

    float d[7], an[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        d[k] = p.rm[(size_t)(a * 7 + k) * HW + pix];
        an[k] = p.anchors[(size_t)idx * 7 + k];
    }
    const float diag = sqrtf(an[4] * an[4] + an[5] * an[5]);
    float box[7];
    box[0] = d[0] * diag + an[0];
    box[1] = d[1] * diag + an[1];
    box[2] = d[2] * an[3] + an[2];
    box[3] = expf(d[3]) * an[3];
    box[4] = expf(d[4]) * an[4];
    box[5] = expf(d[5]) * an[5];
    box[6] = d[6] + an[6];
    // boxes_to_corners_3d: 'hwl' order stores (h, w, l); the corner template wants (dx, dy, dz) = (l, w, h)
    const float dx = p.order_hwl ? box[5] : box[3], dy = p.order_hwl ? box[4] : box[5], dz = p.order_hwl ? box[3] : box[4];
    const float c = cosf(box[6]), s = sinf(box[6]);
    const float sx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, sy[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
    float cx[8], cy[8], cz[8];
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float lx = dx * (sx[k] * 0.5f), ly = dy * (sy[k] * 0.5f), lz = dz * (sz[k] * 0.5f);
        // rotate_points_along_z: points @ [[c, s, 0], [-s, c, 0], [0, 0, 1]]
        float x = lx * c - ly * s + box[0], y = lx * s + ly * c + box[1], z = lz + box[2];
        if (p.T) {
            const float* T = p.T;
            const float X = T[0] * x + T[1] * y + T[2] * z + T[3];
            const float Y = T[4] * x + T[5] * y + T[6] * z + T[7];
            const float Z = T[8] * x + T[9] * y + T[10] * z + T[11];
            x = X; y = Y; z = Z;
        }
        cx[k] = x; cy[k] = y; cz[k] = z;
        xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
        ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
        zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
    }
    // remove_large_pred_bbx: x/y extent <= 6 m and (quirk, box_utils.py:743-749) a non-zero y extent used as "z_len"
    const bool keep_size = (xmax - xmin <= 6.f) && (ymax - ymin <= 6.f) && (ymax - ymin != 0.f);
    // remove_bbx_abnormal_z
    const bool keep_z = zmin >= -3.f && zmax <= 1.f;
    if (!(keep_size && keep_z)) return;

    const int slot = atomicAdd(p.count, 1);
    if (slot >= p.capacity) return;
    float* o = p.corners + (size_t)slot * 24;
#pragma unroll
    for (int k = 0; k < 8; ++k) { o[3 * k] = cx[k]; o[3 * k + 1] = cy[k]; o[3 * k + 2] = cz[k]; }
    p.scores[slot] = prob;
    p.index[slot] = idx;
}

int launch_box_decode(const BoxDecodeParams& p, hipStream_t st) {
    const int n = p.H * p.W * p.A;
    if (n <= 0) return HMVIT_OK;
    hipLaunchKernelGGL(k_box_decode, dim3(cdiv(n, 256)), dim3(256), 0, st, p);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// ranking: rank[i] = number of candidates that precede i in (score descending, anchor index ascending) order
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rank_scores(const float* __restrict__ scores, const int* __restrict__ index, int n,
                                                     int* __restrict__ rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float si = scores[i];
    const int ii = index ? index[i] : i;
    int r = 0;
    for (int j = 0; j < n; ++j) {
        const float sj = scores[j];
        const int ij = index ? index[j] : j;
        r += (sj > si) || (sj == si && ij < ii);
    }
    rank[i] = r;
}

// ------------------------------------------------------------------------------------------
// convex quadrilateral IoU
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float poly_area2(const float (&px)[8], const float (&py)[8], int n) {   // twice the signed area
    float a = 0.f;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        a += px[i] * py[j] - px[j] * py[i];
    }
    return a;
}

__device__ float quad_iou(const float* __restrict__ A, const float* __restrict__ B, int stride) {
    // A, B: 4 corners each, element k at A[k * stride + {0, 1}]
    float ax[8], ay[8], bx[4], by[4];
    for (int k = 0; k < 4; ++k) { ax[k] = A[k * stride]; ay[k] = A[k * stride + 1]; bx[k] = B[k * stride]; by[k] = B[k * stride + 1]; }
    float area_a = 0.5f * poly_area2(ax, ay, 4);
    float b8x[8], b8y[8];
    for (int k = 0; k < 4; ++k) { b8x[k] = bx[k]; b8y[k] = by[k]; }
    float area_b = 0.5f * poly_area2(b8x, b8y, 4);
    if (area_b < 0.f) {   // make the clip polygon counter-clockwise
        float t;
        t = bx[1]; bx[1] = bx[3]; bx[3] = t;
        t = by[1]; by[1] = by[3]; by[3] = t;
        area_b = -area_b;
    }
    area_a = fabsf(area_a);
    // Sutherland-Hodgman: clip A by every edge of B (inside = left of the directed edge)
    int n = 4;
    float ox[8], oy[8];
    for (int e = 0; e < 4 && n > 0; ++e) {
        const float ex0 = bx[e], ey0 = by[e], ex1 = bx[(e + 1) & 3], ey1 = by[(e + 1) & 3];
        const float dx = ex1 - ex0, dy = ey1 - ey0;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const float si = dx * (ay[i] - ey0) - dy * (ax[i] - ex0);
            const float sj = dx * (ay[j] - ey0) - dy * (ax[j] - ex0);
            const bool in_i = si >= 0.f, in_j = sj >= 0.f;
            if (in_i && m < 8) { ox[m] = ax[i]; oy[m] = ay[i]; ++m; }
            if (in_i != in_j && m < 8) {
                const float t = si / (si - sj);
                ox[m] = ax[i] + t * (ax[j] - ax[i]);
                oy[m] = ay[i] + t * (ay[j] - ay[i]);
                ++m;
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) { ax[i] = ox[i]; ay[i] = oy[i]; }
    }
    const float inter = n >= 3 ? fabsf(0.5f * poly_area2(ax, ay, n)) : 0.f;
    const float uni = area_a + area_b - inter;
    return uni > 0.f ? inter / uni : 0.f;
}

// iou[i * nb + j] = IoU(a_i, b_j); corner k of box i at a[(i * stride_box) + k * stride_pt + {0, 1}]
__global__ __launch_bounds__(256) void k_quad_iou(const float* __restrict__ a, const float* __restrict__ b, int na, int nb,
                                                  int stride_box, int stride_pt, float* __restrict__ iou) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= na * nb) return;
    const int i = t / nb, j = t - i * nb;
    iou[t] = quad_iou(a + (size_t)i * stride_box, b + (size_t)j * stride_box, stride_pt);
}

int launch_quad_iou(const float* a, const float* b, int na, int nb, int stride_box, int stride_pt, float* iou, hipStream_t st) {
    if (na <= 0 || nb <= 0) return HMVIT_OK;
    hipLaunchKernelGGL(k_quad_iou, dim3(cdiv(na * nb, 256)), dim3(256), 0, st, a, b, na, nb, stride_box, stride_pt, iou);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// rotated NMS: one workgroup walks the top-K candidates in score order
// ------------------------------------------------------------------------------------------
constexpr int kNmsTop = 1000;

// sorted[r] = candidate with rank r (r < K)
__global__ __launch_bounds__(256) void k_nms_gather(const int* __restrict__ rank, int n, int K, int* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && rank[i] < K) sorted[rank[i]] = i;
}

// iou matrix of the K sorted candidates (upper triangle is what the greedy pass reads)
__global__ __launch_bounds__(256) void k_nms_iou(const float* __restrict__ corners, const int* __restrict__ sorted, int K,
                                                 float* __restrict__ iou) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= K * K) return;
    const int i = t / K, j = t - i * K;
    if (j <= i) return;
    iou[t] = quad_iou(corners + (size_t)sorted[i] * 24, corners + (size_t)sorted[j] * 24, 3);
}

__global__ __launch_bounds__(1024) void k_nms_greedy(const float* __restrict__ corners, const int* __restrict__ sorted,
                                                     const float* __restrict__ iou, int K, float thresh, float range_lo_x,
                                                     float range_lo_y, float range_hi_x, float range_hi_y, int* __restrict__ keep,
                                                     int* __restrict__ n_keep) {
    __shared__ unsigned char dead[kNmsTop];
    __shared__ int n_out;
    for (int i = threadIdx.x; i < K; i += blockDim.x) dead[i] = 0;
    if (threadIdx.x == 0) n_out = 0;
    __syncthreads();
    for (int i = 0; i < K; ++i) {
        if (!dead[i]) {   // uniform: read after the barrier of the previous iteration
            for (int j = i + 1 + threadIdx.x; j < K; j += blockDim.x)
                if (iou[(size_t)i * K + j] > thresh) dead[j] = 1;
            if (threadIdx.x == 0) {
                // get_mask_for_boxes_within_range_torch: every corner inside the x / y range (applied after the NMS)
                const float* c = corners + (size_t)sorted[i] * 24;
                bool in = true;
                for (int k = 0; k < 8; ++k)
                    in = in && c[3 * k] >= range_lo_x && c[3 * k] <= range_hi_x && c[3 * k + 1] >= range_lo_y && c[3 * k + 1] <= range_hi_y;
                if (in) keep[n_out++] = sorted[i];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_keep = n_out;
}

int launch_nms_rotated(const float* corners, const float* scores, const int* index, int n, float thresh, const float* range4,
                       int* rank, int* sorted, float* iou, int* keep, int* n_keep, hipStream_t st) {
    if (n <= 0) {
        HMVIT_CHECK_HIP(hipMemsetAsync(n_keep, 0, sizeof(int), st));
        return HMVIT_OK;
    }
    const int K = n < kNmsTop ? n : kNmsTop;
    hipLaunchKernelGGL(k_rank_scores, dim3(cdiv(n, 256)), dim3(256), 0, st, scores, index, n, rank);
    hipLaunchKernelGGL(k_nms_gather, dim3(cdiv(n, 256)), dim3(256), 0, st, rank, n, K, sorted);
    hipLaunchKernelGGL(k_nms_iou, dim3(cdiv(K * K, 256)), dim3(256), 0, st, corners, sorted, K, iou);
    hipLaunchKernelGGL(k_nms_greedy, dim3(1), dim3(1024), 0, st, corners, sorted, iou, K, thresh, range4[0], range4[1], range4[2],
                       range4[3], keep, n_keep);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

}  // namespace hmvit
