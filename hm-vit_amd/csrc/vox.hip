// Pillariser / voxeliser on the device: replaces spconv.utils.Point2VoxelCPU3d as used by
// opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:34-57.  spconv is a third-party dependency that is not under
// /root/reference (spconv-cu113, version unpinned in the reference's README); its published sequential algorithm is:
//
//   for every point, in input order: c = floor((p - range_min) / voxel_size) per axis, drop the point if any c is outside
//   [0, grid); if the voxel is new: drop the point when max_voxels voxels exist already, else open voxel number
//   n_voxels++ with coordinates (z, y, x); append the point to its voxel unless it already holds max_points points.
//
// The same result without a sequential pass: voxel order = order of the first point of every cell (an exclusive scan over
// "this point is the first of its cell"), and the slot of a point inside its voxel = its rank by input index among the
// points of the cell, obtained by max_points rounds of "atomicMin over the not yet placed points of every cell" - fully
// deterministic, no sort.
#include "common.hpp"
#include "kernels.hpp"

namespace hmvit {

__global__ __launch_bounds__(256) void k_vox_cell(VoxParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_points) return;
    const float4 pt = *reinterpret_cast<const float4*>(p.points + (size_t)i * 4);
    const int cx = (int)floorf((pt.x - p.rmin[0]) / p.vsize[0]);
    const int cy = (int)floorf((pt.y - p.rmin[1]) / p.vsize[1]);
    const int cz = (int)floorf((pt.z - p.rmin[2]) / p.vsize[2]);
    int cell = -1;
    if (cx >= 0 && cx < p.nx && cy >= 0 && cy < p.ny && cz >= 0 && cz < p.nz) {
        cell = (cz * p.ny + cy) * p.nx + cx;
        atomicMin(&p.first[cell], i);
        atomicAdd(&p.count[cell], 1);
    }
    p.cell[i] = cell;
    p.placed[i] = cell < 0 ? 1 : 0;
}

// flag[i] = 1 when point i opens a voxel
__global__ __launch_bounds__(256) void k_vox_flags(VoxParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_points) return;
    const int cell = p.cell[i];
    p.scan[i] = (cell >= 0 && p.first[cell] == i) ? 1 : 0;
}

// exclusive prefix sum of scan[0..n) in place, one workgroup; total -> *n_voxels (capped at max_voxels)
__global__ __launch_bounds__(1024) void k_vox_scan(VoxParams p) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < p.n_points; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < p.n_points ? p.scan[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < p.n_points) p.scan[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *p.n_voxels = carry < p.max_voxels ? carry : p.max_voxels;
}

// first points open their voxel: id, coordinates (z, y, x), number of points
__global__ __launch_bounds__(256) void k_vox_open(VoxParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_points) return;
    const int cell = p.cell[i];
    if (cell < 0 || p.first[cell] != i) return;
    const int v = p.scan[i];
    if (v >= p.max_voxels) { p.vox_id[cell] = -1; return; }
    p.vox_id[cell] = v;
    const int cx = cell % p.nx, cy = (cell / p.nx) % p.ny, cz = cell / (p.nx * p.ny);
    p.coords[(size_t)v * 3 + 0] = cz;
    p.coords[(size_t)v * 3 + 1] = cy;
    p.coords[(size_t)v * 3 + 2] = cx;
    const int c = p.count[cell];
    p.num_points[v] = c < p.max_points ? c : p.max_points;
}

// one slot round: (a) every unplaced point bids with its index, (b) the lowest index of a cell takes slot `round`
__global__ __launch_bounds__(256) void k_vox_bid(VoxParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_points || p.placed[i]) return;
    atomicMin(&p.cmin[p.cell[i]], i);
}
__global__ __launch_bounds__(256) void k_vox_place(VoxParams p, int round) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_points || p.placed[i]) return;
    const int cell = p.cell[i];
    if (p.cmin[cell] != i) return;
    p.cmin[cell] = 0x7fffffff;     // only the winner writes; the others compare against their own index
    p.placed[i] = 1;
    const int v = p.vox_id[cell];
    if (v >= 0)
        *reinterpret_cast<float4*>(p.voxels + ((size_t)v * p.max_points + round) * 4) =
            *reinterpret_cast<const float4*>(p.points + (size_t)i * 4);
}

__global__ __launch_bounds__(256) void k_vox_fill(int* a, int* b, int n, int va, int vb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = va; b[i] = vb; }
}

int launch_voxelize(const VoxParams& p, hipStream_t st) {
    const int n_cells = p.nx * p.ny * p.nz;
    HMVIT_CHECK_HIP(hipMemsetAsync(p.count, 0, (size_t)n_cells * sizeof(int), st));
    HMVIT_CHECK_HIP(hipMemsetAsync(p.voxels, 0, (size_t)p.max_voxels * p.max_points * 4 * sizeof(float), st));
    HMVIT_CHECK_HIP(hipMemsetAsync(p.num_points, 0, (size_t)p.max_voxels * sizeof(int), st));
    HMVIT_CHECK_HIP(hipMemsetAsync(p.coords, 0, (size_t)p.max_voxels * 3 * sizeof(int), st));
    hipLaunchKernelGGL(k_vox_fill, dim3(cdiv(n_cells, 256)), dim3(256), 0, st, p.first, p.cmin, n_cells, 0x7fffffff, 0x7fffffff);
    if (p.n_points > 0) {
        const dim3 g(cdiv(p.n_points, 256)), b(256);
        hipLaunchKernelGGL(k_vox_cell, g, b, 0, st, p);
        hipLaunchKernelGGL(k_vox_flags, g, b, 0, st, p);
        hipLaunchKernelGGL(k_vox_scan, dim3(1), dim3(1024), 0, st, p);
        hipLaunchKernelGGL(k_vox_open, g, b, 0, st, p);
        for (int r = 0; r < p.max_points; ++r) {
            hipLaunchKernelGGL(k_vox_bid, g, b, 0, st, p);
            hipLaunchKernelGGL(k_vox_place, g, b, 0, st, p, r);
        }
    } else {
        HMVIT_CHECK_HIP(hipMemsetAsync(p.n_voxels, 0, sizeof(int), st));
    }
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

}  // namespace hmvit
