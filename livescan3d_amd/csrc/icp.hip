// icp.hip -- device-resident point-to-point ICP for gfx950, replacing src/NativeUtils/icp.cpp:18-177.
//
// The reference (per iteration): nanoflann kd-tree over the target rebuilt every iteration + OpenMP queries
// (icp.cpp:18-32), a sequential one-to-one matching scan (:95-126), 2.5-sigma rejection on squared distances
// (:34-73,:128), and an uncentred Kabsch step through OpenCV (mean difference, 3x3 SVD, apply; :138-168).
//
// Here (everything stays in HBM, no host synchronisation inside the iteration loop):
//   * exact nearest neighbour -- either an LDS-tiled brute force (nn_mode 0) or a voxel grid over the target built
//     ONCE per call (the target never moves): counting sort by cell, cells ordered super-block (16^3 cells) ->
//     block (4^3 cells) -> cell so that every block and super-block is one contiguous range of the sorted points,
//     tight AABBs per block and super-block.  A query first scans the 27 cells around it; that answer is final when
//     the best f32 distance is provably inside the scanned neighbourhood.  The remaining ("far") queries are
//     compacted and resolved by a second kernel that walks the two-level AABB hierarchy with exact f32 lower bounds
//     (a box is skipped only when its minimum distance exceeds the best so far), so the result is the exact NN at
//     any distance without ever falling back to O(n1) work per query.  Distances are evaluated exactly like
//     PointCloud::kdtree_distance (include/NativeUtils/icp.h:40-47): (d0*d0 + d1*d1) + d2*d2, no FMA.
//     Equal distances resolve to the lowest target index (nanoflann's tie order is traversal dependent).
//   * one-to-one matching -- a 64-bit atomicMin per target on (dist_bits << 32 | ~i): minimum distance wins, the
//     LATER source index wins ties, which is what the sequential scan at icp.cpp:95-126 ends with.
//   * statistics / Kabsch sums -- wavefront shuffle reductions -> LDS -> one partial per workgroup, combined in a
//     fixed order in double (deterministic, run-to-run bit-identical).  The accumulators are sum(d), sum((d-mean)^2),
//     count, sum(m1), sum(m2), sum(m2 m1^T): the reference's solver is closed-form Kabsch, not a 6x6 normal matrix.
//   * 3x3 SVD -- one-sided Jacobi in double in a single-thread kernel (U*Vt is the polar factor of M: independent
//     of the SVD algorithm up to rounding), then the same f32 products, det test and updates as icp.cpp:155-168.
//   * apply -- one fused translate+rotate pass over the source cloud in the reference's f32 operation order.
// This file is compiled with -ffp-contract=off.
#include "lsn_common.hpp"

#include <mutex>
#include <vector>

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;      // partial-sum slots
constexpr int kMaxCells = 1 << 22;    // dense grid capacity (16 MiB of cell starts)
constexpr int kScanItems = 16;        // cells per thread in the scan kernels
constexpr int kScanBlock = kThreads * kScanItems;  // 4096
constexpr int kSuper = 16;            // cells per super-block edge
constexpr int kMaxSupers = kMaxCells / (kSuper * kSuper * kSuper);  // 1024
constexpr int kMaxBlocks3 = kMaxCells / 64;                          // 4^3-cell blocks
constexpr float kBoundSlack = 0.999f; // shrinks the geometric bound: covers the rounding of the cell arithmetic
constexpr int kBfTile = 1024;         // targets per LDS tile in the brute-force kernel

struct GridParams {
    float ox, oy, oz;   // origin (bbox min)
    float h, inv_h;
    int nx, ny, nz;     // cells per axis (multiples of 16)
    int nsx, nsy, nsz;  // super-blocks per axis
    int ncells;         // nx * ny * nz
    int n_points;
};

struct Box {  // tight AABB of a block / super-block; empty: lo = +inf, hi = -inf
    float lx, ly, lz, hx, hy, hz, pad0, pad1;
};

// per-iteration device state
struct IcpState {
    float T[3];
    float Rn[9];
    float mean, stddev, thresh;
    int m, mk;
};

// ---- small helpers ------------------------------------------------------------------------------------------

__device__ __forceinline__ float dist2(float qx, float qy, float qz, float px, float py, float pz)
{
    const float d0 = qx - px;
    const float d1 = qy - py;
    const float d2 = qz - pz;
    return d0 * d0 + d1 * d1 + d2 * d2;  // (d0*d0 + d1*d1) + d2*d2, contraction is off
}

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_min_f(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Sums NV doubles per thread over the workgroup (4 waves) into out[] (valid in every thread).
template <int NV>
__device__ __forceinline__ void block_sum_d(double (&v)[NV], double *lds /* [4*NV] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        double s = wave_sum_d(v[i]);
        if (lane == 0) lds[wave * NV + i] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = (lds[i] + lds[NV + i]) + (lds[2 * NV + i] + lds[3 * NV + i]);
    __syncthreads();
}

// Deterministic sum of n_parts partial vectors (stride NV) by one workgroup; result valid in every thread.
template <int NV>
__device__ __forceinline__ void reduce_partials(const double *parts, int n_parts, double (&out)[NV], double *lds)
{
#pragma unroll
    for (int i = 0; i < NV; i++) out[i] = 0;
    for (int b = threadIdx.x; b < n_parts; b += kThreads) {
#pragma unroll
        for (int i = 0; i < NV; i++) out[i] += parts[(size_t)b * NV + i];
    }
    block_sum_d<NV>(out, lds);
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_h, int n)
{
    float f = floorf((v - o) * inv_h);
    int c = (f >= 0.0f) ? ((f < (float)n) ? (int)f : n - 1) : 0;  // NaN -> 0
    return c;
}

// Cell order: super-block (16^3 cells) major, then block (4^3 cells), then cell -- every block / super-block is a
// contiguous range of cell indices, hence of the cell-sorted point array.
__device__ __forceinline__ int cell_index(int cx, int cy, int cz, const GridParams &g)
{
    const int s = ((cz >> 4) * g.nsy + (cy >> 4)) * g.nsx + (cx >> 4);
    const int b = ((((cz >> 2) & 3) * 4) + ((cy >> 2) & 3)) * 4 + ((cx >> 2) & 3);
    const int l = (((cz & 3) * 4) + (cy & 3)) * 4 + (cx & 3);
    return (s << 12) | (b << 6) | l;
}

// Exact f32 lower bound of dist2(q, p) over every p inside the box: component-wise |q - p| >= the clamped gap, and
// rounding is monotone, so the same operation order as dist2 keeps the inequality in floating point.
__device__ __forceinline__ float box_min_dist2(float qx, float qy, float qz, const Box &b)
{
    const float dx = fmaxf(0.0f, fmaxf(b.lx - qx, qx - b.hx));
    const float dy = fmaxf(0.0f, fmaxf(b.ly - qy, qy - b.hy));
    const float dz = fmaxf(0.0f, fmaxf(b.lz - qz, qz - b.hz));
    return dx * dx + dy * dy + dz * dz;
}

// Upper bound of the NN distance offered by a non-empty box: its farthest corner (every box holds >= 1 point).
__device__ __forceinline__ float box_max_dist2(float qx, float qy, float qz, const Box &b)
{
    const float dx = fmaxf(fabsf(qx - b.lx), fabsf(qx - b.hx));
    const float dy = fmaxf(fabsf(qy - b.ly), fabsf(qy - b.hy));
    const float dz = fmaxf(fabsf(qz - b.lz), fabsf(qz - b.hz));
    const float d = dx * dx + dy * dy + dz * dz;
    return b.lx <= b.hx ? d * 1.0001f : INFINITY;  // slack: the bound is only used to prune, never reported
}

// ---- grid build -----------------------------------------------------------------------------------------------

__global__ __launch_bounds__(kThreads) void bbox_partial_kernel(const float *pts, int n, float *part /* [blocks][6] */)
{
    __shared__ float lds[4 * 6];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float v = pts[3 * (size_t)i + c];
            mn[c] = fminf(mn[c], v);  // fminf/fmaxf ignore NaN operands
            mx[c] = fmaxf(mx[c], v);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = wave_min_f(mn[c]), b = wave_max_f(mx[c]);
        if (lane == 0) {
            lds[wave * 6 + c] = a;
            lds[wave * 6 + 3 + c] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = lds[threadIdx.x];
        for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? fminf(v, lds[w * 6 + threadIdx.x]) : fmaxf(v, lds[w * 6 + threadIdx.x]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(64) void grid_setup_kernel(const float *part, int n_parts, int n_points, float cell_override, GridParams *gp)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = threadIdx.x; b < n_parts; b += 64)
        for (int c = 0; c < 3; c++) {
            mn[c] = fminf(mn[c], part[b * 6 + c]);
            mx[c] = fmaxf(mx[c], part[b * 6 + 3 + c]);
        }
    for (int c = 0; c < 3; c++) {
        mn[c] = wave_min_f(mn[c]);
        mx[c] = wave_max_f(mx[c]);
    }
    if (threadIdx.x != 0) return;
    float ext[3];
    for (int c = 0; c < 3; c++) {
        if (!(mn[c] <= mx[c])) { mn[c] = 0; mx[c] = 0; }  // all-NaN axis
        ext[c] = fminf(mx[c] - mn[c], 1e30f);
    }
    float L = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
    if (!(L > 0.0f) || !(L < INFINITY)) L = 1.0f;
    // cell edge ~ twice the point spacing of a surface-like cloud spread over the bbox faces
    float area = ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2];
    float h = 2.0f * sqrtf(fmaxf(area, 1e-12f) / (float)(n_points > 0 ? n_points : 1));
    h = fminf(fmaxf(h, L / 512.0f), L / 4.0f);
    if (cell_override > 0.0f) h = cell_override;
    int nsx = 1, nsy = 1, nsz = 1;
    for (int guard = 0; guard < 2000; guard++) {
        // cells needed per axis, rounded up to whole super-blocks
        nsx = (int)fminf(floorf(ext[0] / h / kSuper) + 1.0f, 4096.0f);
        nsy = (int)fminf(floorf(ext[1] / h / kSuper) + 1.0f, 4096.0f);
        nsz = (int)fminf(floorf(ext[2] / h / kSuper) + 1.0f, 4096.0f);
        if ((long long)nsx * nsy * nsz <= kMaxSupers && (float)(nsx * kSuper) * h > ext[0] * 1.001f &&
            (float)(nsy * kSuper) * h > ext[1] * 1.001f && (float)(nsz * kSuper) * h > ext[2] * 1.001f)
            break;
        h *= 1.26f;
        nsx = nsy = nsz = 1;
    }
    if ((long long)nsx * nsy * nsz > kMaxSupers) nsx = nsy = nsz = 1;
    const int nx = nsx * kSuper, ny = nsy * kSuper, nz = nsz * kSuper;
    gp->ox = mn[0]; gp->oy = mn[1]; gp->oz = mn[2];
    gp->h = h;
    gp->inv_h = 1.0f / h;
    gp->nx = nx; gp->ny = ny; gp->nz = nz;
    gp->nsx = nsx; gp->nsy = nsy; gp->nsz = nsz;
    gp->ncells = nx * ny * nz;
    gp->n_points = n_points;
}

__global__ __launch_bounds__(kThreads) void cell_count_kernel(const float *pts, int n, const GridParams *gp, int *cell_of,
                                                              int *cell_cnt)
{
    const GridParams g = *gp;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        int cx = cell_coord(pts[3 * (size_t)i], g.ox, g.inv_h, g.nx);
        int cy = cell_coord(pts[3 * (size_t)i + 1], g.oy, g.inv_h, g.ny);
        int cz = cell_coord(pts[3 * (size_t)i + 2], g.oz, g.inv_h, g.nz);
        int c = cell_index(cx, cy, cz, g);
        cell_of[i] = c;
        atomicAdd(&cell_cnt[c], 1);
    }
}

// exclusive scan of cell_cnt[0..ncells) into cell_start[0..ncells], three small kernels over the fixed capacity
__global__ __launch_bounds__(kThreads) void scan_block_sums_kernel(const int *cnt, const GridParams *gp, int *block_sums)
{
    __shared__ int lds[4];
    const int ncells = gp->ncells;
    const int base = blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
    int s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++)
        if (base + k < ncells) s += cnt[base + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ __launch_bounds__(1024) void scan_top_kernel(int *block_sums, int n_blocks)
{
    // n_blocks <= 1024: one element per thread, Hillis-Steele in LDS
    __shared__ int lds[1024];
    int v = threadIdx.x < n_blocks ? block_sums[threadIdx.x] : 0;
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int t = threadIdx.x >= off ? lds[threadIdx.x - off] : 0;
        __syncthreads();
        lds[threadIdx.x] += t;
        __syncthreads();
    }
    if (threadIdx.x < n_blocks) block_sums[threadIdx.x] = lds[threadIdx.x] - v;  // exclusive
    if (threadIdx.x == n_blocks - 1) block_sums[n_blocks] = lds[threadIdx.x];     // grand total for the tail block
}

__global__ __launch_bounds__(kThreads) void scan_finish_kernel(const int *cnt, const GridParams *gp, const int *block_sums,
                                                               int *cell_start)
{
    __shared__ int lds[4];
    const int ncells = gp->ncells;
    const int base = blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
    if (blockIdx.x * kScanBlock > ncells) return;
    int v[kScanItems];
    int s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        v[k] = (base + k < ncells) ? cnt[base + k] : 0;
        s += v[k];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int pre = block_sums[blockIdx.x];
    for (int w = 0; w < wave; w++) pre += lds[w];
    int run = pre + incl - s;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        if (base + k <= ncells) cell_start[base + k] = run;  // also writes cell_start[ncells] = n_points
        run += v[k];
    }
}

__global__ __launch_bounds__(kThreads) void cell_scatter_kernel(const float *pts, int n, const int *cell_of, const int *cell_start,
                                                                int *cell_cnt, float4 *sorted)
{
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const int c = cell_of[i];
        const int slot = cell_start[c] + atomicSub(&cell_cnt[c], 1) - 1;  // counts run back down to zero
        sorted[slot] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __int_as_float(i));
    }
}

// ---- nearest neighbour ----------------------------------------------------------------------------------------

__device__ __forceinline__ void claim_target(unsigned long long *keys, int k, float d, int i)
{
    // icp.cpp:95-126: smallest distance keeps the target; on equal distance the later source index replaces
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)i);
    atomicMin(&keys[k], key);
}

__device__ __forceinline__ void scan_range(const float4 *sorted, int s, int e, float qx, float qy, float qz, float &best,
                                           int &best_i)
{
    for (int j = s; j < e; j++) {
        const float4 p = sorted[j];
        const float d = dist2(qx, qy, qz, p.x, p.y, p.z);
        const int k = __float_as_int(p.w);
        if (d < best || (d == best && k < best_i)) {
            best = d;
            best_i = k;
        }
    }
}

// Tight AABB of every 4^3-cell block (one wave per block; its points are one contiguous range).
__global__ __launch_bounds__(kThreads) void block_box_kernel(const GridParams *gp, const int *cell_start, const float4 *sorted, Box *boxes)
{
    const int lane = threadIdx.x & 63;
    const int n_blocks = gp->ncells / 64;
    for (int b = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6); b < n_blocks; b += gridDim.x * (kThreads / 64)) {
        Box bx = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY, 0, 0};
        const int s = cell_start[b * 64], e = cell_start[b * 64 + 64];
        if (s < e) {  // wave-uniform
            for (int j = s + lane; j < e; j += 64) {
                const float4 p = sorted[j];
                bx.lx = fminf(bx.lx, p.x); bx.ly = fminf(bx.ly, p.y); bx.lz = fminf(bx.lz, p.z);
                bx.hx = fmaxf(bx.hx, p.x); bx.hy = fmaxf(bx.hy, p.y); bx.hz = fmaxf(bx.hz, p.z);
            }
            bx.lx = wave_min_f(bx.lx); bx.ly = wave_min_f(bx.ly); bx.lz = wave_min_f(bx.lz);
            bx.hx = wave_max_f(bx.hx); bx.hy = wave_max_f(bx.hy); bx.hz = wave_max_f(bx.hz);
        }
        if (lane == 0) boxes[b] = bx;
    }
}

// AABB of every super-block = union of its 64 block boxes (one wave per super-block).
__global__ __launch_bounds__(64) void super_box_kernel(const GridParams *gp, const Box *boxes, Box *supers)
{
    const int s = blockIdx.x;
    if (s >= gp->ncells / 4096) return;
    const Box b = boxes[s * 64 + threadIdx.x];
    Box r;
    r.lx = wave_min_f(b.lx); r.ly = wave_min_f(b.ly); r.lz = wave_min_f(b.lz);
    r.hx = wave_max_f(b.hx); r.hy = wave_max_f(b.hy); r.hz = wave_max_f(b.hz);
    r.pad0 = r.pad1 = 0;
    if (threadIdx.x == 0) supers[s] = r;
}

// One thread per query: scan the 27 cells around the (clamped) query cell.  Every unscanned target sits in a cell at
// Chebyshev distance >= 2, i.e. at least one cell edge h away, so the answer is final once best <= (h * slack)^2
// (slack covers the rounding of the cell arithmetic).  Otherwise the query joins the far list, carrying what it found.
__global__ __launch_bounds__(kThreads) void nn_grid_kernel(const float *queries, int n2, const GridParams *gp, const int *cell_start,
                                                           const float4 *sorted, int *idx, float *dist, unsigned long long *keys,
                                                           int *far_list, int *n_far, const float *seed_targets, int n1)
{
    const GridParams g = *gp;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n2) return;
    const float qx = queries[3 * (size_t)i], qy = queries[3 * (size_t)i + 1], qz = queries[3 * (size_t)i + 2];
    const int cx = cell_coord(qx, g.ox, g.inv_h, g.nx);
    const int cy = cell_coord(qy, g.oy, g.inv_h, g.ny);
    const int cz = cell_coord(qz, g.oz, g.inv_h, g.nz);
    float best = INFINITY;
    int best_i = 0x7FFFFFFF;
    if (seed_targets) {
        // ICP iterations after the first: the query moved a little, so its previous neighbour (idx[i], a real target
        // point) is an excellent candidate.  Starting from it changes nothing in the result -- the answer is still the
        // lexicographic (distance, index) minimum over every point that is not provably farther -- but the far pass
        // begins with a tight bound and prunes almost the whole hierarchy at once.
        const int k = idx[i];
        if ((unsigned int)k < (unsigned int)n1) {
            const float d = dist2(qx, qy, qz, seed_targets[3 * (size_t)k], seed_targets[3 * (size_t)k + 1], seed_targets[3 * (size_t)k + 2]);
            if (d == d) {  // not NaN
                best = d;
                best_i = k;
            }
        }
        // A candidate farther than a cell edge cannot be proven by the 27 cells around the query whatever they hold, so
        // the query goes to the far pass at once: that pass scans every block that can hold something nearer anyway.
        const float seed_bound = g.h * kBoundSlack;
        if (best < INFINITY && best > seed_bound * seed_bound) {
            idx[i] = best_i;
            dist[i] = best;
            far_list[atomicAdd(n_far, 1)] = i;
            return;
        }
    }
    // This kernel is a chain of dependent memory round trips per query, not arithmetic: fetch all 27 cell ranges at once
    // (54 independent loads in flight), then walk them with the point loads issued four at a time.
    int rs[27], re[27];
#pragma unroll
    for (int k = 0; k < 27; k++) {
        const int x = cx + (k % 3) - 1, y = cy + ((k / 3) % 3) - 1, z = cz + (k / 9) - 1;
        const bool in = (unsigned int)x < (unsigned int)g.nx && (unsigned int)y < (unsigned int)g.ny && (unsigned int)z < (unsigned int)g.nz;
        const int c = in ? cell_index(x, y, z, g) : 0;
        rs[k] = in ? cell_start[c] : 0;
        re[k] = in ? cell_start[c + 1] : 0;
    }
#pragma unroll
    for (int k = 0; k < 27; k++) {
        int j = rs[k];
        for (; j + 4 <= re[k]; j += 4) {
            const float4 p0 = sorted[j], p1 = sorted[j + 1], p2 = sorted[j + 2], p3 = sorted[j + 3];
            const float d0 = dist2(qx, qy, qz, p0.x, p0.y, p0.z), d1 = dist2(qx, qy, qz, p1.x, p1.y, p1.z);
            const float d2 = dist2(qx, qy, qz, p2.x, p2.y, p2.z), d3 = dist2(qx, qy, qz, p3.x, p3.y, p3.z);
            const int k0 = __float_as_int(p0.w), k1 = __float_as_int(p1.w), k2 = __float_as_int(p2.w), k3 = __float_as_int(p3.w);
            if (d0 < best || (d0 == best && k0 < best_i)) { best = d0; best_i = k0; }
            if (d1 < best || (d1 == best && k1 < best_i)) { best = d1; best_i = k1; }
            if (d2 < best || (d2 == best && k2 < best_i)) { best = d2; best_i = k2; }
            if (d3 < best || (d3 == best && k3 < best_i)) { best = d3; best_i = k3; }
        }
        scan_range(sorted, j, re[k], qx, qy, qz, best, best_i);
    }
    const float bound = g.h * kBoundSlack;
    if (best <= bound * bound) {
        idx[i] = best_i;
        dist[i] = best;
        if (keys) claim_target(keys, best_i, best, i);
    } else {
        idx[i] = best_i;  // what the neighbourhood offered (0x7FFFFFFF / inf when it was empty): the far pass starts there
        dist[i] = best;
        far_list[atomicAdd(n_far, 1)] = i;
    }
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

// Far queries (compacted list): exact NN through the two-level AABB hierarchy, ONE WAVE PER QUERY -- the hierarchy is
// 64-ary on purpose: the 64 lanes test 64 super-block boxes at a time, then the 64 block boxes of a super-block in one
// step, then scan a block's points 64 at a time; (distance, index) pairs are reduced lexicographically with wave
// shuffles.  Super-blocks are visited nearest-first, so the bound (min of: what the 27-cell scan found, the smallest
// farthest-corner distance of any box seen, the best point so far) tightens after the first visit and almost every
// other box is skipped.  A box is skipped only when its exact f32 minimum distance EXCEEDS the bound, so equal-distance
// candidates are still seen and the lowest index wins, like everywhere else.
__global__ __launch_bounds__(kThreads) void nn_far_kernel(const float *queries, const GridParams *gp, const int *cell_start,
                                                          const float4 *sorted, const Box *boxes, const Box *supers, const int *far_list,
                                                          const int *n_far, int *idx, float *dist, unsigned long long *keys)
{
    // every super-block's minimum distance lives in LDS (lane l owns entries l, 64 + l, ...): the loops below then run
    // over the chunks that exist (3 for a typical 2-sensor scene) instead of a compile-time 16
    __shared__ float s_md[kThreads / 64][kMaxSupers];
    const int lane = threadIdx.x & 63;
    float *md = s_md[threadIdx.x >> 6];
    const int slot = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);  // one query per wave
    if (slot >= *n_far) return;                                          // wave-uniform
    const int i = far_list[slot];
    const int n_supers = gp->ncells / 4096;
    const float qx = queries[3 * (size_t)i], qy = queries[3 * (size_t)i + 1], qz = queries[3 * (size_t)i + 2];
    float best = dist[i];  // uniform across the wave from here on
    int best_i = idx[i];

    const int n_chunks = (n_supers + 63) >> 6;
    float ub = INFINITY;
    for (int c = 0; c < n_chunks; c++) {
        const int s = c * 64 + lane;
        float m = INFINITY;
        if (s < n_supers) {
            const Box b = supers[s];
            m = box_min_dist2(qx, qy, qz, b);
            ub = fminf(ub, box_max_dist2(qx, qy, qz, b));
        }
        md[s] = m;
    }
    float bound = fminf(best, wave_min_f(ub));

    for (int visits = 0; visits <= kMaxSupers; visits++) {
        // nearest unvisited super-block
        float m = INFINITY;
        int mc = 0;
        for (int c = 0; c < n_chunks; c++) {
            const float v = md[c * 64 + lane];
            if (v < m) {
                m = v;
                mc = c;
            }
        }
        const float wm = wave_min_f(m);
        if (!(wm <= bound)) break;  // nothing left that could hold a point at distance <= bound (also ends on NaN)
        const int src = __ffsll((long long)__ballot(m == wm)) - 1;
        const int s = __shfl(mc, src, 64) * 64 + src;
        if (lane == src) md[s] = INFINITY;  // visited (only this lane ever reads the entry again)

        // its 64 blocks, one per lane; every lane also fetches its block's point range now, so that the candidate loop
        // below starts on the points without another dependent round trip per block
        const Box bb = boxes[s * 64 + lane];
        const int range0 = cell_start[(s * 64 + lane) * 64], range1 = cell_start[(s * 64 + lane) * 64 + 64];
        const float mb = box_min_dist2(qx, qy, qz, bb);
        bound = fminf(bound, wave_min_f(box_max_dist2(qx, qy, qz, bb)));
        unsigned long long cand = __ballot(mb <= bound);
        float lbest = best;
        int lbi = best_i;
        while (cand) {
            const int b = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            const int e = __shfl(range1, b, 64);
            for (int j = __shfl(range0, b, 64) + lane; j < e; j += 64) {
                const float4 p = sorted[j];
                const float d = dist2(qx, qy, qz, p.x, p.y, p.z);
                const int k = __float_as_int(p.w);
                if (d < lbest || (d == lbest && k < lbi)) {
                    lbest = d;
                    lbi = k;
                }
            }
        }
        // lexicographic (distance, index) minimum over the wave; non-negative floats order like their bit patterns
        const unsigned long long key = wave_min_u64(((unsigned long long)__float_as_uint(lbest) << 32) | (unsigned int)lbi);
        const float kd = __uint_as_float((unsigned int)(key >> 32));
        if (kd < best || (kd == best && (int)(unsigned int)key < best_i)) {
            best = kd;
            best_i = (int)(unsigned int)key;
        }
        bound = fminf(bound, best);
    }
    if (lane == 0) {
        if (best_i == 0x7FFFFFFF) best_i = 0;  // every distance was NaN: keep the index in range
        idx[i] = best_i;
        dist[i] = best;
        if (keys) claim_target(keys, best_i, best, i);
    }
}

// Far queries that come with a good candidate (ICP iterations after the first: the previous neighbour seeds the search):
// the job is no longer to FIND a near point but to PROVE that nothing is nearer.  Still one wave per query (point scans
// must stay 64 wide: a single lane walking a block's points is a chain of dependent loads), but without the machinery of
// the exploring kernel above -- no nearest-first ordering, no LDS table, one reduction at the very end: the lanes test the
// super-block boxes 64 at a time against the candidate's distance, the (few) boxes that do not exceed it are opened in
// index order, lanes keep private (distance, index) minima and only the bound is refreshed per super-block.  Same pruning
// rule as everywhere: a box is skipped only when its exact f32 minimum distance EXCEEDS the bound; ties go to the lowest index.
__global__ __launch_bounds__(kThreads) void nn_far_seeded_kernel(const float *queries, const GridParams *gp, const int *cell_start,
                                                                 const float4 *sorted, const Box *boxes, const Box *supers,
                                                                 const int *far_list, const int *n_far, int *idx, float *dist,
                                                                 unsigned long long *keys, int *far2_list, int *n_far2)
{
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);  // one query per wave
    if (slot >= *n_far) return;                                          // wave-uniform
    const int i = far_list[slot];
    const float qx = queries[3 * (size_t)i], qy = queries[3 * (size_t)i + 1], qz = queries[3 * (size_t)i + 2];
    const float seed = dist[i];
    const int seed_i = idx[i];
    if (!(seed < INFINITY)) {  // inf / NaN: nothing to prove against -> the exploring kernel searches from scratch
        if (lane == 0) far2_list[atomicAdd(n_far2, 1)] = i;
        return;
    }
    float bound = seed;                 // wave-uniform
    float lbest = seed;                 // per-lane running minimum, lexicographic with lbi
    int lbi = seed_i;
    const int n_supers = gp->ncells / 4096;
    for (int c0 = 0; c0 < n_supers; c0 += 64) {
        const int sl = c0 + lane;
        float m = INFINITY;
        if (sl < n_supers) m = box_min_dist2(qx, qy, qz, supers[sl]);
        unsigned long long open = __ballot(m <= bound);
        while (open) {
            const int s = c0 + __ffsll((long long)open) - 1;
            open &= open - 1;
            // the super-block's 64 blocks, one per lane, with their point ranges
            const Box bb = boxes[s * 64 + lane];
            const int range0 = cell_start[(s * 64 + lane) * 64], range1 = cell_start[(s * 64 + lane) * 64 + 64];
            unsigned long long cand = __ballot(box_min_dist2(qx, qy, qz, bb) <= bound);
            while (cand) {
                const int b = __ffsll((long long)cand) - 1;
                cand &= cand - 1;
                const int e = __shfl(range1, b, 64);
                for (int j = __shfl(range0, b, 64) + lane; j < e; j += 64) {
                    const float4 p = sorted[j];
                    const float d = dist2(qx, qy, qz, p.x, p.y, p.z);
                    const int k = __float_as_int(p.w);
                    if (d < lbest || (d == lbest && k < lbi)) {
                        lbest = d;
                        lbi = k;
                    }
                }
            }
            bound = wave_min_f(lbest);
            // super-blocks of this chunk that the tighter bound rules out need not be opened
            open &= __ballot(m <= bound);
        }
    }
    const unsigned long long key = wave_min_u64(((unsigned long long)__float_as_uint(lbest) << 32) | (unsigned int)lbi);
    if (lane == 0) {
        idx[i] = (int)(unsigned int)key;
        dist[i] = __uint_as_float((unsigned int)(key >> 32));
        if (keys) claim_target(keys, (int)(unsigned int)key, __uint_as_float((unsigned int)(key >> 32)), i);
    }
}

// LDS-tiled brute force: a workgroup owns 256 queries (directly, or through the `list` of unresolved queries) and
// streams the whole target cloud through LDS; every lane reads the same LDS address (broadcast, conflict-free).
// Targets are visited in index order with a strict '<', so the lowest index wins ties.
__global__ __launch_bounds__(kThreads) void nn_brute_kernel(const float *targets, int n1, const float *queries, int n2,
                                                            const int *list, const int *n_list, int *idx, float *dist,
                                                            unsigned long long *keys)
{
    __shared__ float tx[kBfTile], ty[kBfTile], tz[kBfTile];
    const int nq = list ? *n_list : n2;
    if (blockIdx.x * kThreads >= nq) return;  // uniform per workgroup
    const int slot = blockIdx.x * kThreads + threadIdx.x;
    const bool active = slot < nq;
    const int i = active ? (list ? list[slot] : slot) : 0;
    float qx = 0, qy = 0, qz = 0;
    if (active) {
        qx = queries[3 * (size_t)i];
        qy = queries[3 * (size_t)i + 1];
        qz = queries[3 * (size_t)i + 2];
    }
    float best = INFINITY;
    int best_i = 0x7FFFFFFF;
    for (int t0 = 0; t0 < n1; t0 += kBfTile) {
        const int nt = min(kBfTile, n1 - t0);
        __syncthreads();
        for (int j = threadIdx.x; j < nt; j += kThreads) {
            tx[j] = targets[3 * (size_t)(t0 + j)];
            ty[j] = targets[3 * (size_t)(t0 + j) + 1];
            tz[j] = targets[3 * (size_t)(t0 + j) + 2];
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < nt; j++) {
            const float d = dist2(qx, qy, qz, tx[j], ty[j], tz[j]);
            if (d < best) {
                best = d;
                best_i = t0 + j;
            }
        }
    }
    if (active) {
        if (best_i == 0x7FFFFFFF) best_i = 0;
        idx[i] = best_i;
        dist[i] = best;
        if (keys) claim_target(keys, best_i, best, i);
    }
}

// ---- matching statistics and Kabsch sums ----------------------------------------------------------------------

__device__ __forceinline__ bool is_winner(const unsigned long long *keys, const int *idx, int i)
{
    return (unsigned int)(keys[idx[i]] & 0xFFFFFFFFull) == 0xFFFFFFFFu - (unsigned int)i;
}

// pass 1: m = number of one-to-one matches, sum of their squared distances
__global__ __launch_bounds__(kThreads) void stats1_kernel(const int *idx, const float *dist, const unsigned long long *keys, int n2,
                                                          double *part /* [blocks][2] */)
{
    __shared__ double lds[4 * 2];
    double v[2] = {0, 0};
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n2; i += gridDim.x * kThreads) {
        if (is_winner(keys, idx, i)) {
            v[0] += 1.0;
            v[1] += (double)dist[i];
        }
    }
    block_sum_d<2>(v, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2] = v[0];
        part[blockIdx.x * 2 + 1] = v[1];
    }
}

// pass 2: sum (d - mean)^2 with mean rounded to f32 like GetStandardDeviation (icp.cpp:36-49)
__global__ __launch_bounds__(kThreads) void stats2_kernel(const int *idx, const float *dist, const unsigned long long *keys, int n2,
                                                          const double *part1, int n_part1, double *part /* [blocks][1] */,
                                                          IcpState *st)
{
    __shared__ double lds[4 * 2];
    double s1[2];
    reduce_partials<2>(part1, n_part1, s1, lds);
    const float mean = (float)(s1[1] / s1[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->m = (int)s1[0];
        st->mean = mean;
    }
    double v[1] = {0};
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n2; i += gridDim.x * kThreads) {
        if (is_winner(keys, idx, i)) {
            const float df = dist[i] - mean;  // f32 difference, squared in double (pow(float,int) -> double)
            v[0] += (double)df * (double)df;
        }
    }
    block_sum_d<1>(v, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = v[0];
}

// pass 3: reject d > 2.5*std (icp.cpp:56-73), accumulate count, sum m1, sum m2, sum m2 m1^T over the kept matches
__global__ __launch_bounds__(kThreads) void accum_kernel(const float *verts1, const float *verts2, const int *idx, const float *dist,
                                                         const unsigned long long *keys, int n2, const double *part2, int n_part2,
                                                         double *part /* [blocks][16] */, IcpState *st)
{
    __shared__ double lds[4 * 16];
    double s2[1];
    reduce_partials<1>(part2, n_part2, s2, lds);
    const float m_f = (float)st->m;  // written by stats2_kernel of the same iteration (previous launch)
    float sd = (float)(s2[0]);       // the reference keeps the running sum in a float
    sd = sd / m_f;
    sd = sqrtf(sd);
    const float thresh = 2.5f * sd;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->stddev = sd;
        st->thresh = thresh;
    }
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = 0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n2; i += gridDim.x * kThreads) {
        if (!is_winner(keys, idx, i)) continue;
        if (dist[i] > thresh) continue;  // NaN distances are kept, like the reference's comparison
        const int k = idx[i];
        const double a0 = verts1[3 * (size_t)k], a1 = verts1[3 * (size_t)k + 1], a2 = verts1[3 * (size_t)k + 2];
        const double b0 = verts2[3 * (size_t)i], b1 = verts2[3 * (size_t)i + 1], b2 = verts2[3 * (size_t)i + 2];
        v[0] += 1.0;
        v[1] += a0; v[2] += a1; v[3] += a2;
        v[4] += b0; v[5] += b1; v[6] += b2;
        v[7] += b0 * a0; v[8] += b0 * a1; v[9] += b0 * a2;
        v[10] += b1 * a0; v[11] += b1 * a1; v[12] += b1 * a2;
        v[13] += b2 * a0; v[14] += b2 * a1; v[15] += b2 * a2;
    }
    block_sum_d<16>(v, lds);
    if (threadIdx.x < 16) {
        // v[] is valid in every thread; thread k stores component k
        double out = 0;
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (threadIdx.x == k) out = v[k];
        part[blockIdx.x * 16 + threadIdx.x] = out;
    }
}

// 3x3 SVD by one-sided Jacobi (double).  A = U diag(w) V^T, singular values sorted descending.
__device__ void svd3(const double A[9], double U[9], double w[3], double V[9])
{
    double B[9];
    for (int i = 0; i < 9; i++) { B[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; sweep++) {
        int rotations = 0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double a = 0, b = 0, c = 0;
                for (int k = 0; k < 3; k++) {
                    a += B[3 * k + p] * B[3 * k + p];
                    b += B[3 * k + q] * B[3 * k + q];
                    c += B[3 * k + p] * B[3 * k + q];
                }
                if (fabs(c) <= 1e-300 || c * c <= 1e-32 * (a * b)) continue;  // columns already orthogonal to ~1e-16
                rotations++;
                const double zeta = (b - a) / (2.0 * c);
                const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                for (int k = 0; k < 3; k++) {
                    const double bp = B[3 * k + p], bq = B[3 * k + q];
                    B[3 * k + p] = cs * bp - sn * bq;
                    B[3 * k + q] = sn * bp + cs * bq;
                    const double vp = V[3 * k + p], vq = V[3 * k + q];
                    V[3 * k + p] = cs * vp - sn * vq;
                    V[3 * k + q] = sn * vp + cs * vq;
                }
            }
        if (rotations == 0) break;
    }
    for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += B[3 * k + j] * B[3 * k + j];
        w[j] = sqrt(s);
    }
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (w[ord[j]] > w[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double Bs[9], Vs[9], ws[3];
    for (int j = 0; j < 3; j++) {
        ws[j] = w[ord[j]];
        for (int k = 0; k < 3; k++) { Bs[3 * k + j] = B[3 * k + ord[j]]; Vs[3 * k + j] = V[3 * k + ord[j]]; }
    }
    for (int i = 0; i < 9; i++) V[i] = Vs[i];
    for (int j = 0; j < 3; j++) w[j] = ws[j];
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) U[3 * k + j] = (w[j] > 1e-300) ? Bs[3 * k + j] / w[j] : 0.0;
    if (!(w[0] > 1e-300)) {
        for (int i = 0; i < 9; i++) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    if (!(w[1] > 1e-12 * w[0])) {
        const double u0[3] = {U[0], U[3], U[6]};
        int m = 0;
        if (fabs(u0[1]) < fabs(u0[m])) m = 1;
        if (fabs(u0[2]) < fabs(u0[m])) m = 2;
        double e[3] = {0, 0, 0};
        e[m] = 1;
        const double d = u0[m];
        const double v[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
        const double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        U[1] = v[0] / nv; U[4] = v[1] / nv; U[7] = v[2] / nv;
    }
    if (!(w[2] > 1e-12 * w[0])) {
        const double a0 = U[0], a1 = U[3], a2 = U[6], b0 = U[1], b1 = U[4], b2 = U[7];
        const double c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;
        const double detV = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
        const double s = detV < 0 ? -1.0 : 1.0;
        U[2] = s * c0; U[5] = s * c1; U[8] = s * c2;
    }
}

// icp.cpp:141 (T), :152-163 (M, SVD, tempR), :167-168 (t, R update).  One workgroup; thread 0 does the 3x3 work.
__global__ __launch_bounds__(kThreads) void solve_kernel(const double *part3, int n_part3, float *R, float *t, IcpState *st, float *trace,
                                                         int iter)
{
    __shared__ double lds[4 * 16];
    double s[16];
    reduce_partials<16>(part3, n_part3, s, lds);
    if (threadIdx.x != 0) return;
    const int mk = (int)s[0];
    st->mk = mk;
    float T[3] = {0, 0, 0};
    float Rn[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (mk > 0) {
        // T = mean(m1 - m2); M = sum (m2 + T) m1^T = sum m2 m1^T + T (sum m1)^T
        for (int c = 0; c < 3; c++) T[c] = (float)((s[1 + c] - s[4 + c]) / (double)mk);
        double M[9];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) M[3 * a + b] = (double)(float)(s[7 + 3 * a + b] + (double)T[a] * s[1 + b]);
        double U[9], w[3], V[9];
        svd3(M, U, w, V);
        float Uf[9], Vtf[9];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                Uf[3 * r + c] = (float)U[3 * r + c];
                Vtf[3 * r + c] = (float)V[3 * c + r];
            }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Rn[3 * r + c] = Uf[3 * r] * Vtf[c] + Uf[3 * r + 1] * Vtf[3 + c] + Uf[3 * r + 2] * Vtf[6 + c];
        const double det = (double)Rn[0] * ((double)Rn[4] * Rn[8] - (double)Rn[5] * Rn[7]) -
                           (double)Rn[1] * ((double)Rn[3] * Rn[8] - (double)Rn[5] * Rn[6]) +
                           (double)Rn[2] * ((double)Rn[3] * Rn[7] - (double)Rn[4] * Rn[6]);
        if (det < 0) {
            for (int r = 0; r < 3; r++) Uf[3 * r + 2] = -Uf[3 * r + 2];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) Rn[3 * r + c] = Uf[3 * r] * Vtf[c] + Uf[3 * r + 1] * Vtf[3 + c] + Uf[3 * r + 2] * Vtf[6 + c];
        }
        // matT += tempT * matR.t() (R before the update), matR = matR * tempR
        float add[3];
        for (int c = 0; c < 3; c++) add[c] = T[0] * R[3 * c] + T[1] * R[3 * c + 1] + T[2] * R[3 * c + 2];
        for (int c = 0; c < 3; c++) t[c] += add[c];
        float Rnew[9];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Rnew[3 * r + c] = R[3 * r] * Rn[c] + R[3 * r + 1] * Rn[3 + c] + R[3 * r + 2] * Rn[6 + c];
        for (int k = 0; k < 9; k++) R[k] = Rnew[k];
    }
    for (int c = 0; c < 3; c++) st->T[c] = T[c];
    for (int k = 0; k < 9; k++) st->Rn[k] = Rn[k];
    if (trace) {
        float *tr = trace + 16 * iter;
        tr[0] = (float)st->m;
        tr[1] = (float)mk;
        tr[2] = st->mean;
        tr[3] = st->stddev;
        for (int c = 0; c < 3; c++) tr[4 + c] = T[c];
        for (int k = 0; k < 9; k++) tr[7 + k] = Rn[k];
    }
}

// icp.cpp:143-146 + :165: v = (v + T) * Rn, row vectors, f32, one rounding per operation
__global__ __launch_bounds__(kThreads) void apply_kernel(float *verts2, int n2, const IcpState *st)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (st->mk <= 0) return;
    const float T0 = st->T[0], T1 = st->T[1], T2 = st->T[2];
    const float r0 = st->Rn[0], r1 = st->Rn[1], r2 = st->Rn[2], r3 = st->Rn[3], r4 = st->Rn[4], r5 = st->Rn[5], r6 = st->Rn[6],
                r7 = st->Rn[7], r8 = st->Rn[8];
    if (i >= n2) return;
    float x = verts2[3 * (size_t)i], y = verts2[3 * (size_t)i + 1], z = verts2[3 * (size_t)i + 2];
    x = x + T0;
    y = y + T1;
    z = z + T2;
    verts2[3 * (size_t)i] = x * r0 + y * r3 + z * r6;
    verts2[3 * (size_t)i + 1] = x * r1 + y * r4 + z * r7;
    verts2[3 * (size_t)i + 2] = x * r2 + y * r5 + z * r8;
}

}  // namespace

// -------------------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------------------

struct LsnIcp {
    int device = 0;
    int max_n1 = 0, max_n2 = 0;
    float cell_override = 0.0f;
    lsn::DevBuf gp, bbox_part, cell_of, cell_cnt, cell_start, block_sums, sorted, boxes, supers;
    lsn::DevBuf idx, dist, keys, unresolved, unresolved2, counters, part1, part2, part3, state, trace;
    int trace_iters = 0;
    bool seed_nn = true;   // $LSN_ICP_NO_SEED=1 turns the previous-neighbour seeding off (ablation)
    std::mutex mu;
};

static constexpr int kTraceCap = 1024;

extern "C" LsnIcp *lsnIcpCreate(int device, int max_n1, int max_n2)
{
    lsn::clear_error();
    if (max_n1 <= 0 || max_n2 <= 0) {
        lsn::set_error("lsnIcpCreate: bad capacities (%d, %d)", max_n1, max_n2);
        return nullptr;
    }
    LSN_HIP_NULL(hipSetDevice(device));
    LsnIcp *w = new (std::nothrow) LsnIcp();
    if (!w) return nullptr;
    w->device = device;
    w->max_n1 = max_n1;
    w->max_n2 = max_n2;
    if (const char *e2 = getenv("LSN_ICP_NO_SEED")) w->seed_nn = atoi(e2) == 0;
    const char *env = getenv("LSN_ICP_CELL");
    if (env) w->cell_override = (float)atof(env);
    bool bad = false;
    bad |= w->gp.reserve(sizeof(GridParams)) != 0;
    bad |= w->bbox_part.reserve(sizeof(float) * 6 * kMaxBlocks) != 0;
    bad |= w->cell_of.reserve(sizeof(int) * (size_t)max_n1) != 0;
    bad |= w->cell_cnt.reserve(sizeof(int) * (size_t)kMaxCells) != 0;
    bad |= w->cell_start.reserve(sizeof(int) * ((size_t)kMaxCells + kScanBlock)) != 0;
    bad |= w->block_sums.reserve(sizeof(int) * 2048) != 0;
    bad |= w->sorted.reserve(sizeof(float4) * (size_t)max_n1) != 0;
    bad |= w->boxes.reserve(sizeof(Box) * (size_t)kMaxBlocks3) != 0;
    bad |= w->supers.reserve(sizeof(Box) * (size_t)kMaxSupers) != 0;
    bad |= w->idx.reserve(sizeof(int) * (size_t)max_n2) != 0;
    bad |= w->dist.reserve(sizeof(float) * (size_t)max_n2) != 0;
    bad |= w->keys.reserve(sizeof(unsigned long long) * (size_t)max_n1) != 0;
    bad |= w->unresolved.reserve(sizeof(int) * (size_t)max_n2) != 0;
    bad |= w->unresolved2.reserve(sizeof(int) * (size_t)max_n2) != 0;
    bad |= w->counters.reserve(64) != 0;
    bad |= w->part1.reserve(sizeof(double) * 2 * kMaxBlocks) != 0;
    bad |= w->part2.reserve(sizeof(double) * kMaxBlocks) != 0;
    bad |= w->part3.reserve(sizeof(double) * 16 * kMaxBlocks) != 0;
    bad |= w->state.reserve(sizeof(IcpState)) != 0;
    bad |= w->trace.reserve(sizeof(float) * 16 * kTraceCap) != 0;
    if (bad) {
        delete w;
        return nullptr;
    }
    return w;
}

extern "C" void lsnIcpDestroy(LsnIcp *w)
{
    if (!w) return;
    (void)hipSetDevice(w->device);
    delete w;
}

static inline int blocks_for(int n) { return (n + kThreads - 1) / kThreads; }
static inline int capped_blocks(int n) { int b = blocks_for(n); return b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b); }

// Builds the voxel grid over the target cloud (stream ordered, no host synchronisation).
static int build_grid(LsnIcp *w, const float *d_verts1, int n1, hipStream_t s)
{
    const int nb = capped_blocks(n1);
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(nb), dim3(kThreads), 0, s, d_verts1, n1, w->bbox_part.as<float>());
    hipLaunchKernelGGL(grid_setup_kernel, dim3(1), dim3(64), 0, s, w->bbox_part.as<float>(), nb, n1, w->cell_override,
                       w->gp.as<GridParams>());
    LSN_HIP(hipMemsetAsync(w->cell_cnt.p, 0, sizeof(int) * (size_t)kMaxCells, s));
    hipLaunchKernelGGL(cell_count_kernel, dim3(nb), dim3(kThreads), 0, s, d_verts1, n1, w->gp.as<GridParams>(), w->cell_of.as<int>(),
                       w->cell_cnt.as<int>());
    const int sb = kMaxCells / kScanBlock;  // 1024
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(sb), dim3(kThreads), 0, s, w->cell_cnt.as<int>(), w->gp.as<GridParams>(),
                       w->block_sums.as<int>());
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, s, w->block_sums.as<int>(), sb);
    hipLaunchKernelGGL(scan_finish_kernel, dim3(sb + 1), dim3(kThreads), 0, s, w->cell_cnt.as<int>(), w->gp.as<GridParams>(),
                       w->block_sums.as<int>(), w->cell_start.as<int>());
    hipLaunchKernelGGL(cell_scatter_kernel, dim3(nb), dim3(kThreads), 0, s, d_verts1, n1, w->cell_of.as<int>(), w->cell_start.as<int>(),
                       w->cell_cnt.as<int>(), w->sorted.as<float4>());
    hipLaunchKernelGGL(block_box_kernel, dim3(2048), dim3(kThreads), 0, s, (const GridParams *)w->gp.as<GridParams>(),
                       (const int *)w->cell_start.as<int>(), (const float4 *)w->sorted.as<float4>(), w->boxes.as<Box>());
    hipLaunchKernelGGL(super_box_kernel, dim3(kMaxSupers), dim3(64), 0, s, (const GridParams *)w->gp.as<GridParams>(),
                       (const Box *)w->boxes.as<Box>(), w->supers.as<Box>());
    LSN_HIP(hipGetLastError());
    return 0;
}

static int run_nn(LsnIcp *w, const float *d_verts1, int n1, const float *d_verts2, int n2, int *d_idx, float *d_dist,
                  unsigned long long *keys, int nn_mode, hipStream_t s, bool seeded = false)
{
    if (nn_mode == 0) {
        hipLaunchKernelGGL(nn_brute_kernel, dim3(blocks_for(n2)), dim3(kThreads), 0, s, d_verts1, n1, d_verts2, n2, (const int *)nullptr,
                           (const int *)nullptr, d_idx, d_dist, keys);
    } else {
        int *n_unres = w->counters.as<int>();
        LSN_HIP(hipMemsetAsync(n_unres, 0, sizeof(int), s));
        hipLaunchKernelGGL(nn_grid_kernel, dim3(blocks_for(n2)), dim3(kThreads), 0, s, d_verts2, n2, w->gp.as<GridParams>(),
                           w->cell_start.as<int>(), w->sorted.as<float4>(), d_idx, d_dist, keys, w->unresolved.as<int>(), n_unres,
                           seeded ? d_verts1 : (const float *)nullptr, n1);
        const int *far = w->unresolved.as<int>();
        const int *n_far = n_unres;
        if (seeded) {
            // every far query carries its previous neighbour: the proving kernel takes them; only queries without a
            // finite candidate go on to the exploring kernel
            LSN_HIP(hipMemsetAsync(n_unres + 1, 0, sizeof(int), s));
            hipLaunchKernelGGL(nn_far_seeded_kernel, dim3((n2 + 3) / 4), dim3(kThreads), 0, s, d_verts2, (const GridParams *)w->gp.as<GridParams>(),
                               (const int *)w->cell_start.as<int>(), (const float4 *)w->sorted.as<float4>(), (const Box *)w->boxes.as<Box>(),
                               (const Box *)w->supers.as<Box>(), far, n_far, d_idx, d_dist, keys, w->unresolved2.as<int>(), n_unres + 1);
            far = w->unresolved2.as<int>();
            n_far = n_unres + 1;
        }
        // the queries nothing above could prove; workgroups beyond the list length exit at once
        hipLaunchKernelGGL(nn_far_kernel, dim3((n2 + 3) / 4), dim3(kThreads), 0, s, d_verts2, (const GridParams *)w->gp.as<GridParams>(),
                           (const int *)w->cell_start.as<int>(), (const float4 *)w->sorted.as<float4>(), (const Box *)w->boxes.as<Box>(),
                           (const Box *)w->supers.as<Box>(), far, n_far, d_idx, d_dist, keys);
    }
    LSN_HIP(hipGetLastError());
    static const bool debug = getenv("LSN_ICP_DEBUG") != nullptr;   // dev aid: synchronises, prints the far-list sizes and the grid
    if (debug && nn_mode != 0) {
        int c[2] = {0, 0};
        GridParams g;
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(c, w->counters.p, sizeof(c), hipMemcpyDeviceToHost);
        (void)hipMemcpy(&g, w->gp.p, sizeof(g), hipMemcpyDeviceToHost);
        fprintf(stderr, "[lsn icp] n2=%d far=%d far2=%d h=%g grid=%dx%dx%d\n", n2, c[0], seeded ? c[1] : -1, (double)g.h, g.nx, g.ny, g.nz);
    }
    return 0;
}

static int check_sizes(LsnIcp *w, int n1, int n2, const char *who)
{
    if (!w) {
        lsn::set_error("%s: null workspace", who);
        return -1;
    }
    if (n1 <= 0 || n2 <= 0 || n1 > w->max_n1 || n2 > w->max_n2) {
        lsn::set_error("%s: sizes (%d, %d) outside the workspace capacity (%d, %d)", who, n1, n2, w->max_n1, w->max_n2);
        return -1;
    }
    return 0;
}

extern "C" int lsnIcpNearest(LsnIcp *w, const float *d_verts1, int n1, const float *d_verts2, int n2, int *d_idx, float *d_dist2,
                             int nn_mode, void *stream)
{
    lsn::clear_error();
    if (check_sizes(w, n1, n2, "lsnIcpNearest")) return -1;
    if (!d_verts1 || !d_verts2 || !d_idx || !d_dist2) {
        lsn::set_error("lsnIcpNearest: null argument");
        return -1;
    }
    std::lock_guard<std::mutex> g(w->mu);
    LSN_HIP(hipSetDevice(w->device));
    hipStream_t s = lsn::as_stream(stream);
    if (nn_mode != 0 && build_grid(w, d_verts1, n1, s)) return -1;
    return run_nn(w, d_verts1, n1, d_verts2, n2, d_idx, d_dist2, nullptr, nn_mode, s);
}

extern "C" int lsnIcpRun(LsnIcp *w, const float *d_verts1, int n1, float *d_verts2, int n2, float *d_R, float *d_t, int maxIter,
                         int nn_mode, void *stream)
{
    lsn::clear_error();
    if (check_sizes(w, n1, n2, "lsnIcpRun")) return -1;
    if (!d_verts1 || !d_verts2 || !d_R || !d_t) {
        lsn::set_error("lsnIcpRun: null argument");
        return -1;
    }
    std::lock_guard<std::mutex> g(w->mu);
    LSN_HIP(hipSetDevice(w->device));
    hipStream_t s = lsn::as_stream(stream);
    w->trace_iters = maxIter < kTraceCap ? (maxIter > 0 ? maxIter : 0) : kTraceCap;
    if (maxIter <= 0) return 0;

    if (nn_mode != 0 && build_grid(w, d_verts1, n1, s)) return -1;  // the target is fixed: one build for all iterations

    const int nb = capped_blocks(n2);
    unsigned long long *keys = w->keys.as<unsigned long long>();
    IcpState *st = w->state.as<IcpState>();
    for (int iter = 0; iter < maxIter; iter++) {
        LSN_HIP(hipMemsetAsync(keys, 0xFF, sizeof(unsigned long long) * (size_t)n1, s));
        // from the second iteration on idx[] still holds every query's previous neighbour: the search is seeded with it
        if (run_nn(w, d_verts1, n1, d_verts2, n2, w->idx.as<int>(), w->dist.as<float>(), keys, nn_mode, s, iter > 0 && w->seed_nn)) return -1;
        hipLaunchKernelGGL(stats1_kernel, dim3(nb), dim3(kThreads), 0, s, w->idx.as<int>(), w->dist.as<float>(), keys, n2,
                           w->part1.as<double>());
        hipLaunchKernelGGL(stats2_kernel, dim3(nb), dim3(kThreads), 0, s, w->idx.as<int>(), w->dist.as<float>(), keys, n2,
                           w->part1.as<double>(), nb, w->part2.as<double>(), st);
        hipLaunchKernelGGL(accum_kernel, dim3(nb), dim3(kThreads), 0, s, d_verts1, (const float *)d_verts2, w->idx.as<int>(),
                           w->dist.as<float>(), keys, n2, w->part2.as<double>(), nb, w->part3.as<double>(), st);
        hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(kThreads), 0, s, w->part3.as<double>(), nb, d_R, d_t, st,
                           iter < kTraceCap ? w->trace.as<float>() : (float *)nullptr, iter);
        hipLaunchKernelGGL(apply_kernel, dim3(blocks_for(n2)), dim3(kThreads), 0, s, d_verts2, n2, (const IcpState *)st);
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

extern "C" int lsnIcpTrace(LsnIcp *w, float *out, int max_iters, void *stream)
{
    lsn::clear_error();
    if (!w || !out) return -1;
    LSN_HIP(hipSetDevice(w->device));
    LSN_HIP(hipStreamSynchronize(lsn::as_stream(stream)));
    int n = w->trace_iters < max_iters ? w->trace_iters : max_iters;
    if (n > 0) LSN_HIP(hipMemcpy(out, w->trace.p, sizeof(float) * 16 * (size_t)n, hipMemcpyDeviceToHost));
    return n;
}

// refineWorker_DoWork (LiveScanServer/MainWindowForm.cs:330-410) with every cloud resident in HBM for the whole
// Gauss-Seidel loop: the reference re-uploads "all other sensors" and the sensor's own cloud for each of the
// n_sensors x n_refine_iters ICP calls; here each cloud goes up once and comes back once.  The pose composition at the
// end repeats the C# loops literally, including their in-place update of worldTransforms[i].R while later rows still
// read it (:398-406).
extern "C" int lsnRefine(int device, int n_sensors, float *const *clouds, const int *counts, int n_refine_iters, int n_icp_iters,
                         float *world_R, float *world_t, float *Rs_out, float *Ts_out)
{
    lsn::clear_error();
    if (n_sensors <= 0 || !clouds || !counts) {
        lsn::set_error("lsnRefine: bad arguments");
        return -1;
    }
    long long total = 0;
    int max_n = 0, min_n = 0x7FFFFFFF;
    for (int i = 0; i < n_sensors; i++) {
        if (counts[i] < 0 || (counts[i] > 0 && !clouds[i])) {
            lsn::set_error("lsnRefine: bad cloud %d", i);
            return -1;
        }
        total += counts[i];
        max_n = counts[i] > max_n ? counts[i] : max_n;
        min_n = counts[i] < min_n ? counts[i] : min_n;
    }
    std::vector<float> Rt((size_t)n_sensors * 12, 0.0f);
    for (int i = 0; i < n_sensors; i++)
        for (int j = 0; j < 3; j++) Rt[(size_t)i * 12 + j + j * 3] = 1.0f;   // Rs[i] = I, Ts[i] = 0 (:330-344)
    const bool runnable = n_sensors >= 2 && min_n > 0 && total - min_n <= 0x7FFFFFFFll && n_refine_iters > 0 && n_icp_iters > 0;
    if (runnable) {
        LSN_HIP(hipSetDevice(device));
        hipStream_t s = nullptr;
        LSN_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        lsn::DevBuf d_all, d_others, d_Rt;
        LsnIcp *ws = lsnIcpCreate(device, (int)(total - min_n), max_n);
        int rc = ws ? 0 : -1;
        if (!rc) rc = d_all.reserve(sizeof(float) * 3 * (size_t)total) || d_others.reserve(sizeof(float) * 3 * (size_t)(total - min_n)) ||
                      d_Rt.reserve(sizeof(float) * Rt.size());
        std::vector<long long> off(n_sensors + 1, 0);
        for (int i = 0; i < n_sensors; i++) off[i + 1] = off[i] + counts[i];
        for (int i = 0; i < n_sensors && !rc; i++)
            rc = hipMemcpyAsync(d_all.as<float>() + 3 * off[i], clouds[i], sizeof(float) * 3 * (size_t)counts[i], hipMemcpyHostToDevice, s) != hipSuccess;
        if (!rc) rc = hipMemcpyAsync(d_Rt.p, Rt.data(), sizeof(float) * Rt.size(), hipMemcpyHostToDevice, s) != hipSuccess;
        for (int it = 0; it < n_refine_iters && !rc; it++) {                  // :347
            for (int i = 0; i < n_sensors && !rc; i++) {                      // :349
                long long pos = 0;                                             // :352-357 all other sensors' current clouds
                for (int j = 0; j < n_sensors && !rc; j++) {
                    if (j == i) continue;
                    rc = hipMemcpyAsync(d_others.as<float>() + 3 * pos, d_all.as<float>() + 3 * off[j], sizeof(float) * 3 * (size_t)counts[j],
                                        hipMemcpyDeviceToDevice, s) != hipSuccess;
                    pos += counts[j];
                }
                if (!rc)
                    rc = lsnIcpRun(ws, d_others.as<float>(), (int)pos, d_all.as<float>() + 3 * off[i], counts[i], d_Rt.as<float>() + 12 * i,
                                   d_Rt.as<float>() + 12 * i + 9, n_icp_iters, 1, s);     // :370
            }
        }
        // results land in scratch first: the caller's arrays are only touched when everything worked
        std::vector<float> back((size_t)total * 3);
        if (!rc) rc = hipMemcpyAsync(back.data(), d_all.p, sizeof(float) * 3 * (size_t)total, hipMemcpyDeviceToHost, s) != hipSuccess;
        if (!rc) rc = hipMemcpyAsync(Rt.data(), d_Rt.p, sizeof(float) * Rt.size(), hipMemcpyDeviceToHost, s) != hipSuccess;
        if (!rc) rc = hipStreamSynchronize(s) != hipSuccess;
        if (ws) lsnIcpDestroy(ws);
        (void)hipStreamDestroy(s);
        if (rc) {
            if (lsn::last_error().empty()) lsn::set_error("lsnRefine: %s", hipGetErrorString(hipGetLastError()));
            return -1;
        }
        for (int i = 0; i < n_sensors; i++) memcpy(clouds[i], back.data() + 3 * off[i], sizeof(float) * 3 * (size_t)counts[i]);
    }
    // :382-410 pose composition, the C# loops as written
    if (world_R && world_t) {
        for (int i = 0; i < n_sensors; i++) {
            float *WR = world_R + 9 * i, *Wt = world_t + 3 * i;
            const float *Ri = Rt.data() + 12 * (size_t)i, *Ti = Ri + 9;
            float tempT[3] = {0, 0, 0};
            float tempR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < 3; j++) {
                for (int k = 0; k < 3; k++) tempT[j] += Ti[k] * WR[3 * k + j];
                Wt[j] += tempT[j];
            }
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) {
                    for (int l = 0; l < 3; l++) tempR[3 * j + k] += Ri[l * 3 + j] * WR[3 * l + k];
                    WR[3 * j + k] = tempR[3 * j + k];
                }
        }
    }
    for (int i = 0; i < n_sensors; i++) {
        if (Rs_out) memcpy(Rs_out + 9 * i, Rt.data() + 12 * (size_t)i, 9 * sizeof(float));
        if (Ts_out) memcpy(Ts_out + 3 * i, Rt.data() + 12 * (size_t)i + 9, 3 * sizeof(float));
    }
    return 0;
}
