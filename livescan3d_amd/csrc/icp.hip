// icp.hip -- device-resident point-to-point ICP for gfx950, replacing src/NativeUtils/icp.cpp:18-177.
//
// The reference (per iteration): nanoflann kd-tree over the target rebuilt every iteration + OpenMP queries
// (icp.cpp:18-32), a sequential one-to-one matching scan (:95-126), 2.5-sigma rejection on squared distances
// (:34-73,:128), and an uncentred Kabsch step through OpenCV (mean difference, 3x3 SVD, apply; :138-168).
//
// Here (everything stays in HBM, no host synchronisation inside the iteration loop):
//   * exact nearest neighbour -- either a brute force (nn_mode 0: one query per lane, the targets streamed through scalar loads /
//     SGPRs, no LDS) or a voxel grid over the target built
//     ONCE per call (the target never moves): counting sort by cell, cells ordered super-block (16^3 cells) ->
//     block (4^3 cells) -> cell so that every block and super-block is one contiguous range of the sorted points,
//     tight AABBs per block and super-block.  The source cloud is sorted the same way once per call, so 64 consecutive
//     queries form a compact patch ("query group") that shares its candidates: the hierarchy is culled per group and
//     every surviving block is streamed through LDS for all 64 queries at once, as independent work items spread over
//     the whole device (see the NN section below).  A box is passed over only when its exact f32 minimum distance
//     exceeds every query's bound, so the result is the exact NN at any distance.  Distances are evaluated exactly like
//     PointCloud::kdtree_distance (include/NativeUtils/icp.h:40-47): (d0*d0 + d1*d1) + d2*d2, no FMA.
//     Equal distances resolve to the lowest target index (nanoflann's tie order is traversal dependent).
//     In front of the group search sits the NEAR PATH (round 6): a query that knows a real target point near it -- its previous
//     neighbour, or what a probe of the 27 cells around it found -- walks the grid cells the ball of that distance touches by itself
//     and is settled; only the rest forms the groups' search.  Same bits either way (see the NEAR PATH section).
//     The grid is built with one atomic per RUN of equal cells among a wave's lanes (consecutive points are raster neighbours), whose
//     return value ranks the run's points: the scatter needs no atomics.
//   * one-to-one matching -- a 64-bit atomicMin per target on (dist_bits << 32 | ~i): minimum distance wins, the
//     LATER source index wins ties, which is what the sequential scan at icp.cpp:95-126 ends with; a workgroup first
//     combines its claims in LDS (scene clouds pile thousands of claims onto a few rim targets).
//   * statistics / Kabsch sums -- wavefront shuffle reductions -> LDS -> one partial per workgroup, combined in a
//     fixed order in double over the queries' ORIGINAL order (deterministic, run-to-run bit-identical, independent of
//     the sort).  The accumulators are count, sum(d), sum(d^2), sum(m1), sum(m2), sum(m2 m1^T): the reference's
//     solver is closed-form Kabsch, not a 6x6 normal matrix.
//   * 3x3 SVD -- one-sided Jacobi in double in a single-thread kernel (U*Vt is the polar factor of M: independent
//     of the SVD algorithm up to rounding), then the same f32 products, det test and updates as icp.cpp:155-168.
//   * apply -- the translate+rotate of the source cloud in the reference's f32 operation order rides in the first
//     kernel of the next iteration's NN step (one separate pass after the last iteration).
// This file is compiled with -ffp-contract=off.
#include "lsn_common.hpp"

#include <algorithm>
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
constexpr int kChunk = 256;            // points per scan item (4 LDS batches of 64)
constexpr int kSeedBlocks = 4;        // blocks nearest to a query patch that are scanned first when it has no candidates yet

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
    // warm start of the next iteration's Jacobi SVD: the right singular vectors of this iteration's cross-covariance (the
    // clouds barely move between iterations, so the next matrix is almost diagonalised by them); v_valid is cleared at the
    // start of every lsnIcpRun, which keeps a call's result independent of what the workspace ran before
    int v_valid;
    // the next seeded NN step takes the near path: set by accum_kernel when at least half of this step's queries lie within one
    // target cell edge of their neighbour (below that the walk costs the step more than the shorter group search gives back:
    // configs[1], 18 % settled, +6 us per iteration; configs[2], 98 %, -26 us).  A matter of speed only: the results are the same bits.
    int near_next;
    double Vprev[9];
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

// Deterministic sum of n_parts partial vectors (stride NVP, a power of two >= NV) by one workgroup; result valid in every
// thread.  Thread t sums component t % NVP of the partials t / NVP, t / NVP + G, ... (G = kThreads / NVP groups; NVP
// consecutive threads read one partial: coalesced, all loads independent), the G group sums meet in LDS and thread c adds
// those of component c in a fixed order.  (One partial vector per thread and a shuffle tree over 16 doubles took 5.8 us in
// solve_kernel and in EVERY workgroup of accum_kernel: strided loads, 192 dependent cross-lane moves.)
template <int NV, int NVP>
__device__ __forceinline__ void reduce_partials(const double *parts, int n_parts, double (&out)[NV], double *lds /* [kThreads + NVP] */)
{
    constexpr int G = kThreads / NVP;
    const int c = threadIdx.x & (NVP - 1), g = threadIdx.x / NVP;
    // eight loads in flight per thread (a plain loop would wait for each load before issuing the next: n_parts / G round trips)
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    int p = g;
    for (; p + 7 * G < n_parts; p += 8 * G) {
        const double v0 = parts[(size_t)p * NVP + c], v1 = parts[(size_t)(p + G) * NVP + c], v2 = parts[(size_t)(p + 2 * G) * NVP + c],
                     v3 = parts[(size_t)(p + 3 * G) * NVP + c], v4 = parts[(size_t)(p + 4 * G) * NVP + c], v5 = parts[(size_t)(p + 5 * G) * NVP + c],
                     v6 = parts[(size_t)(p + 6 * G) * NVP + c], v7 = parts[(size_t)(p + 7 * G) * NVP + c];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3; a4 += v4; a5 += v5; a6 += v6; a7 += v7;
    }
    {   // the tail, predicated: still eight independent loads
        const double v0 = p < n_parts ? parts[(size_t)p * NVP + c] : 0.0, v1 = p + G < n_parts ? parts[(size_t)(p + G) * NVP + c] : 0.0;
        const double v2 = p + 2 * G < n_parts ? parts[(size_t)(p + 2 * G) * NVP + c] : 0.0, v3 = p + 3 * G < n_parts ? parts[(size_t)(p + 3 * G) * NVP + c] : 0.0;
        const double v4 = p + 4 * G < n_parts ? parts[(size_t)(p + 4 * G) * NVP + c] : 0.0, v5 = p + 5 * G < n_parts ? parts[(size_t)(p + 5 * G) * NVP + c] : 0.0;
        const double v6 = p + 6 * G < n_parts ? parts[(size_t)(p + 6 * G) * NVP + c] : 0.0;
        a0 += v0; a1 += v1; a2 += v2; a3 += v3; a4 += v4; a5 += v5; a6 += v6;
    }
    const double acc = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    lds[threadIdx.x] = acc;  // [g][c]
    __syncthreads();
    if (threadIdx.x < NVP) {
        double t = 0;
        for (int k = 0; k < G; k++) t += lds[k * NVP + threadIdx.x];
        lds[kThreads + threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) out[i] = lds[kThreads + i];
    __syncthreads();
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

// One point per thread: its cell and its rank inside the cell (the value the cell's counter had: cell_scatter_kernel then needs no
// atomics).  Consecutive points of a cloud are raster neighbours and mostly share a cell, so a run of equal cells among the wave's
// lanes costs ONE atomicAdd (measured on configs[2]'s 738 k targets: an atomic per point 47 us, per run see EXPERIMENTS.md R6-1).
__global__ __launch_bounds__(kThreads) void cell_count_kernel(const float *pts, int n, const GridParams *gp, int *cell_of, int *rank_of,
                                                              int *cell_cnt)
{
    const GridParams g = *gp;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int c = -1 - lane;   // past the end: a run of its own, no atomic
    if (i < n) {
        const int cx = cell_coord(pts[3 * (size_t)i], g.ox, g.inv_h, g.nx);
        const int cy = cell_coord(pts[3 * (size_t)i + 1], g.oy, g.inv_h, g.ny);
        const int cz = cell_coord(pts[3 * (size_t)i + 2], g.oz, g.inv_h, g.nz);
        c = cell_index(cx, cy, cz, g);
    }
    const int prev = __shfl_up(c, 1, 64);
    const unsigned long long starts = __ballot(lane == 0 || c != prev);
    const int first = 63 - __clzll((long long)(starts & (~0ull >> (63 - lane))));                 // the run's first lane (<= lane)
    const unsigned long long later = lane == 63 ? 0ull : starts & (~0ull << (lane + 1));
    const int end = later ? __ffsll((long long)later) - 1 : 64;                                    // one past the run's last lane
    int base = 0;
    if (lane == first && c >= 0) base = atomicAdd(&cell_cnt[c], end - first);
    base = __shfl(base, first, 64);
    if (i < n) {
        cell_of[i] = c;
        rank_of[i] = base + (lane - first);
    }
}

// exclusive scan of cell_cnt[0..ncells) into cell_start[0..ncells], three small kernels over the fixed capacity
// (ncells is a multiple of 16^3 = kScanBlock: a scan block lies inside the grid or outside it)
__global__ __launch_bounds__(kThreads) void scan_block_sums_kernel(const int *cnt, const GridParams *gp, int *block_sums)
{
    __shared__ int lds[4];
    const int ncells = gp->ncells;
    const int base = blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
    int s = 0;
    if (blockIdx.x * kScanBlock < ncells) {
        const int4 *v = reinterpret_cast<const int4 *>(cnt + base);
#pragma unroll
        for (int k = 0; k < kScanItems / 4; k++) {
            const int4 a = v[k];
            s += (a.x + a.y) + (a.z + a.w);
        }
    }
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
    if (blockIdx.x * kScanBlock >= ncells) {
        if (blockIdx.x * kScanBlock == ncells && threadIdx.x == 0) cell_start[ncells] = block_sums[blockIdx.x];   // = n_points
        return;
    }
    int v[kScanItems];
    int s = 0;
    {
        const int4 *v4 = reinterpret_cast<const int4 *>(cnt + base);
#pragma unroll
        for (int k = 0; k < kScanItems / 4; k++) {
            const int4 a = v4[k];
            v[4 * k] = a.x; v[4 * k + 1] = a.y; v[4 * k + 2] = a.z; v[4 * k + 3] = a.w;
            s += (a.x + a.y) + (a.z + a.w);
        }
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
    int4 *o4 = reinterpret_cast<int4 *>(cell_start + base);
#pragma unroll
    for (int k = 0; k < kScanItems / 4; k++) {
        int4 o;
        o.x = run; run += v[4 * k];
        o.y = run; run += v[4 * k + 1];
        o.z = run; run += v[4 * k + 2];
        o.w = run; run += v[4 * k + 3];
        o4[k] = o;
    }
}

__global__ __launch_bounds__(kThreads) void cell_scatter_kernel(const float *pts, int n, const int *cell_of, const int *rank_of, const int *cell_start,
                                                                int *cell_cnt, float4 *sorted)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int c = cell_of[i];
    sorted[cell_start[c] + rank_of[i]] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __int_as_float(i));
    cell_cnt[c] = 0;   // every touched counter back to zero (plain stores of the same value): one memset per workspace, not per build
}

// ---- nearest neighbour ----------------------------------------------------------------------------------------

// The claims of a whole workgroup.  In a scene most queries of a patch share a handful of targets (the rim of the other
// sensor's surface): one global atomicMin per query then piles thousands of atomics onto single addresses, which the L2
// serialises (measured: 100 us for 108 k claims).  The workgroup therefore first combines its claims in LDS -- a
// direct-mapped table of kClaimSlots (target, smallest key) pairs filled with LDS atomics; a query whose slot is taken by
// another target claims in global memory directly -- and then issues one global atomicMin per occupied slot, skipped
// when the target's key is already smaller.  Same result: a minimum of minima.  Must be reached by every thread.
constexpr int kClaimSlots = 1024;
__device__ __forceinline__ void claim_targets(unsigned long long *keys, bool active, int k, float d, int i, int *slot_k, unsigned long long *slot_v)
{
    for (int t = threadIdx.x; t < kClaimSlots; t += kThreads) {
        slot_k[t] = -1;
        slot_v[t] = ~0ull;
    }
    __syncthreads();
    if (active) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)i);
        const int slot = k & (kClaimSlots - 1);
        const int owner = atomicCAS(&slot_k[slot], -1, k);
        if (owner == -1 || owner == k)
            atomicMin(&slot_v[slot], key);
        else if (keys[k] > key)
            atomicMin(&keys[k], key);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kClaimSlots; t += kThreads) {
        const int tk = slot_k[t];
        if (tk >= 0) {
            const unsigned long long v = slot_v[t];
            if (keys[tk] > v) atomicMin(&keys[tk], v);
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
        // the point range rides in the two spare words: one 32-byte load per block in the query kernel
        bx.pad0 = __int_as_float(s);
        bx.pad1 = __int_as_float(e);
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

// Lower bound of dist2(q, p) for EVERY q inside the query box W = [wl, wh] and every p inside b, evaluated with the same
// operations as box_min_dist2: q <= wh gives b.l - wh <= b.l - q exactly, rounding is monotone, so the per-axis gap never
// exceeds the one box_min_dist2 computes for any q of W, and the squares and sums keep the order.  Empty boxes
// (lo = +inf, hi = -inf) come out as +inf.
__device__ __forceinline__ float boxbox_min_dist2(float wlx, float wly, float wlz, float whx, float why, float whz, const Box &b)
{
    const float dx = fmaxf(0.0f, fmaxf(b.lx - whx, wlx - b.hx));
    const float dy = fmaxf(0.0f, fmaxf(b.ly - why, wly - b.hy));
    const float dz = fmaxf(0.0f, fmaxf(b.lz - whz, wlz - b.hz));
    return dx * dx + dy * dy + dz * dz;
}

// In-wave LDS hand-over: LDS instructions of one wave execute in order, so a ds_write of all lanes followed by ds_reads of
// the same wave needs no hardware wait -- only the compiler must not move the accesses across this point.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct alignas(16) WaveStage {  // 64 candidate points of one wave, structure of arrays: 4 consecutive points = one 16-B LDS read
    float x[64], y[64], z[64];
    int i[64];
};

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));

// dist2 of one query against four points, as packed f32 pairs (v_pk_add_f32 with negated operand, v_pk_mul_f32): one
// rounding per operation and the order (d0*d0 + d1*d1) + d2*d2 of PointCloud::kdtree_distance (icp.h:40-47); contraction is off.
__device__ __forceinline__ f4v dist2x4(f2v qx, f2v qy, f2v qz, f4v X, f4v Y, f4v Z)
{
    const f2v ax = qx - X.xy, bx = qx - X.zw;
    const f2v ay = qy - Y.xy, by = qy - Y.zw;
    const f2v az = qz - Z.xy, bz = qz - Z.zw;
    const f2v a = ax * ax + ay * ay + az * az;
    const f2v b = bx * bx + by * by + bz * bz;
    f4v d;
    d.xy = a;
    d.zw = b;
    return d;
}

__device__ __forceinline__ void scan_batch(const float4 &p, int cnt, WaveStage &st, int lane, f2v qx2, f2v qy2, f2v qz2, float &best, int &best_i)
{
    st.x[lane] = p.x;
    st.y[lane] = p.y;
    st.z[lane] = p.z;
    st.i[lane] = __float_as_int(p.w);
    wave_lds_fence();
    for (int t = 0; t < cnt; t += 8) {  // two independent groups of four per step (fills the issue slots between dependent packed ops)
        const f4v d = dist2x4(qx2, qy2, qz2, *reinterpret_cast<const f4v *>(&st.x[t]), *reinterpret_cast<const f4v *>(&st.y[t]),
                              *reinterpret_cast<const f4v *>(&st.z[t]));
        const f4v e = dist2x4(qx2, qy2, qz2, *reinterpret_cast<const f4v *>(&st.x[t + 4]), *reinterpret_cast<const f4v *>(&st.y[t + 4]),
                              *reinterpret_cast<const f4v *>(&st.z[t + 4]));
        const float m = fminf(fminf(fminf(d.x, d.y), fminf(d.z, d.w)), fminf(fminf(e.x, e.y), fminf(e.z, e.w)));  // fminf skips NaN
        if (__ballot(m <= best)) {                                                                                 // wave-uniform branch
            // some lane is improved or tied: the step's candidate of a lane is (m, lowest index among its points at distance m),
            // branch-free (16 compares / selects and a min tree instead of eight conditional updates)
            const i4v K = *reinterpret_cast<const i4v *>(&st.i[t]);
            const i4v L = *reinterpret_cast<const i4v *>(&st.i[t + 4]);
            const int none = 0x7FFFFFFF;
            const int k0 = d.x == m ? K.x : none, k1 = d.y == m ? K.y : none, k2 = d.z == m ? K.z : none, k3 = d.w == m ? K.w : none;
            const int k4 = e.x == m ? L.x : none, k5 = e.y == m ? L.y : none, k6 = e.z == m ? L.z : none, k7 = e.w == m ? L.w : none;
            const int km = min(min(min(k0, k1), min(k2, k3)), min(min(k4, k5), min(k6, k7)));
            const bool take = (m < best) | ((m == best) & (km < best_i));
            best = take ? m : best;
            best_i = take ? km : best_i;
        }
    }
    wave_lds_fence();
}

__device__ __forceinline__ float4 load_point_or_pad(const float4 *__restrict__ sorted, int j, int je)
{
    float4 p = make_float4(NAN, NAN, NAN, __int_as_float(0x7FFFFFFF));  // padding: its distance is NaN, never taken
    if (j < je) p = sorted[j];
    return p;
}

// Every lane (one query each) against the points sorted[js, je): the wave loads 64 points at a time with one coalesced
// 16-byte load per lane -- up to four such batches in flight at once, one memory round trip per kChunk points -- parks a
// batch in its LDS stage and then walks it eight points at a time: all lanes read the same LDS address (broadcast), the
// differences / products / sums run as packed f32 pairs (one rounding per operation, the order of
// PointCloud::kdtree_distance, icp.h:40-47).  Only the minimum of the eight distances is compared with the lane's best;
// the (distance, index) bookkeeping runs in the rare case that some lane of the wave is improved or tied.
__device__ __forceinline__ void scan_points(const float4 *__restrict__ sorted, int js, int je, WaveStage &st, int lane, float qx, float qy,
                                            float qz, float &best, int &best_i)
{
    const f2v qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
    for (int base = js; base < je; base += kChunk) {
        const float4 p0 = load_point_or_pad(sorted, base + lane, je);
        const float4 p1 = load_point_or_pad(sorted, base + 64 + lane, je);
        const float4 p2 = load_point_or_pad(sorted, base + 128 + lane, je);
        const float4 p3 = load_point_or_pad(sorted, base + 192 + lane, je);
        const int left = je - base;  // wave-uniform
        scan_batch(p0, min(64, left), st, lane, qx2, qy2, qz2, best, best_i);
        if (left > 64) scan_batch(p1, min(64, left - 64), st, lane, qx2, qy2, qz2, best, best_i);
        if (left > 128) scan_batch(p2, min(64, left - 128), st, lane, qx2, qy2, qz2, best, best_i);
        if (left > 192) scan_batch(p3, min(64, left - 192), st, lane, qx2, qy2, qz2, best, best_i);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Exact nearest neighbour for 64 SPATIALLY SORTED QUERIES AT A TIME ("query group" = one wave's worth).  The source cloud
// is sorted once per call along its own voxel order, so a group is a compact patch (about one 4^3-cell block) and -- the
// motion being rigid -- stays one for every iteration.  A patch shares its candidates: instead of 64 private tree walks
// (round 1: one wave per far query, each a chain of dependent loads) the two-level box hierarchy is culled once per
// group against the box W of its queries, 64 boxes per step, one per lane, and every surviving block is streamed through
// LDS for all 64 queries at once.
//
// The search is cut into INDEPENDENT work items so that no wave ever runs a chain of dependent memory round trips and
// no slow wave holds the launch up (measured: the one-kernel form below, kept as the overflow fallback, was bound by its
// slowest wave -- ~200 us for ~50 dependent loads -- while the average wave needed 370 candidate points):
//   nn_cull_kernel    one wave per group: seed every query with its previous neighbour (a REAL target point, so the
//                     result does not depend on it), W, the largest bound; emits (group, super-block) items
//   nn_blocks_kernel  one wave per item: the super-block's 64 block boxes, one per lane, against W, then against every
//                     query's own bound; emits (group, point range) items, at most 128 points each
//   nn_scan_kernel    one wave per item: 64 queries x the range's points through LDS, packed f32; the lexicographic
//                     (distance, index) minimum is merged into the query's 64-bit key with atomicMin
//   nn_finish_kernel  one thread per query: index / distance out (original order) + the one-to-one claim
// Without seeds (first ICP iteration, lsnIcpNearest) nn_probe_kernel first lets every query look around its own cell and emits
// the blocks nearest to the queries that found nothing as scan items; their minima are the seeds.
//
// Exactness: a query's key only ever takes (distance, index) pairs of real target points, and a box is passed over only
// when no query of the group can need it: the group test opens a box when boxbox_min_dist2(W, box) <= max_l bound_l,
// which holds whenever some lane's own box_min_dist2(q_l, box) <= bound_l (see boxbox_min_dist2); the per-lane test
// skips a box only when its exact f32 lower bound EXCEEDS every lane's bound, so equal-distance candidates are still
// evaluated and the lowest index wins, like everywhere else.  The bounds are fixed for the whole step (the seeds); they
// are distances of real points, so every point that could beat or tie them is inside an opened box.
// The item lists have a fixed capacity; if one overflows (pathological inputs: no usable seeds, far outliers everywhere)
// the finish kernel runs the complete walk (wave_search) for every group instead -- slower, same result.
//
// Queries with a non-finite coordinate do not take part in the culling (they would open every box); they end with
// index 0 / distance +inf, which is what the search gives them when target 0 is finite.

constexpr unsigned long long kNoKey = 0x7F8000007FFFFFFFull;  // (+inf, no index)

struct GroupInfo {  // per query group, written by nn_cull_kernel
    float wlx, wly, wlz, whx, why, whz;  // W: the box of the group's queries that still search
    float rmax;                          // largest bound among them (-1: nobody searches)
    int pad;
    unsigned long long resolved;         // lanes whose key is already final (the near path below): they open no box
};

constexpr int kSegs = 64;        // the work lists are cut into 64 segments with a counter each: an append is one atomicAdd per
                                 // wave on the counter of segment (group mod 64) -- a single counter serialised ~10^4 atomics
                                 // per step on one address (measured: 30-50 us per launch)
constexpr int kSegStride = 32;  // ints between two segments' counters: one 128-byte line each -- atomics on one line serialise in its L2 channel
                                // whatever the address (measured: 8 k appends on 64 adjacent counters took as long as on one)
constexpr int kCntA = 0, kCntB = 1, kCntSeed = 2;          // a segment's counters inside its line: list A, list B, seed-round part of B
constexpr int kOverflow = kSegs * kSegStride;              // the bank's overflow flag
constexpr int kBankInts = kSegs * kSegStride + kSegStride; // ints per bank

struct NnWork {  // device work lists of one ICP workspace
    uint2 *list_a;      // (group, super-block), segment s = entries [s * seg_a, (s + 1) * seg_a)
    int4 *list_b;       // (group, first point, end point, -)
    int seg_a, seg_b;   // segment capacities
    int item_points;    // points per scan item (a multiple of 64, <= kChunk)
    int *counters;      // two banks of kBankInts
};

// Consumers: wave w of a launch serves segment (w mod 64), entries w / 64, w / 64 + waves / 64, ...: the segment's length
// and the wave's first entry are independent loads (one memory round trip before the work starts, no prefix sums).

__device__ __forceinline__ unsigned long long pack_key(float d, int k)
{
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)k;
}
__device__ __forceinline__ float key_dist(unsigned long long key) { return __uint_as_float((unsigned int)(key >> 32)); }
__device__ __forceinline__ int key_index(unsigned long long key) { return (int)(unsigned int)key; }

__device__ __forceinline__ bool finite3(float x, float y, float z) { return fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY; }

__device__ __forceinline__ int wave_excl_scan_i(int v, int lane)
{
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    return incl - v;
}

// Appends the point range [js, je) of every lane with `take` set to list B as (group, range) items of <= item_points points.
// `spread` picks the segment: a heavy group's ranges are spread over the segments by its super-blocks.
__device__ __forceinline__ void emit_ranges(const NnWork &wk, int *bank, int which, int g, int spread, bool take, int js, int je, int lane)
{
    const int n_items = take ? (je - js + wk.item_points - 1) / wk.item_points : 0;
    const int before = wave_excl_scan_i(n_items, lane);
    const int total = __shfl(before + n_items, 63, 64);
    if (total == 0) return;  // wave-uniform
    const int seg = spread & (kSegs - 1);
    int base = 0;
    if (lane == 0) base = atomicAdd(bank + seg * kSegStride + which, total);
    base = __shfl(base, 0, 64);
    if (base + total > wk.seg_b) {  // (the seed round's items are consumed before the search's are written: both start at 0)
        if (lane == 0) atomicExch(bank + kOverflow, 1);
        return;
    }
    for (int c = 0; c < n_items; c++) {
        const int a = js + c * wk.item_points;
        wk.list_b[(size_t)seg * wk.seg_b + base + before + c] = make_int4(g, a, min(a + wk.item_points, je), 0);
    }
}

// The complete walk for one group in one wave: seeds the search from the nearest blocks when some query has no candidate,
// then opens every super-block / block that a query may need, tightening the bounds as it goes.  Exact for any input;
// used when the item lists overflow.
__device__ void wave_search(const GridParams *__restrict__ gp, const float4 *__restrict__ sorted, const Box *__restrict__ boxes,
                            const Box *__restrict__ supers, WaveStage &st, int lane, float qx, float qy, float qz, bool part, float &best,
                            int &best_i)
{
    if (!__ballot(part)) return;
    const float wlx = wave_min_f(part ? qx : INFINITY), wly = wave_min_f(part ? qy : INFINITY), wlz = wave_min_f(part ? qz : INFINITY);
    const float whx = wave_max_f(part ? qx : -INFINITY), why = wave_max_f(part ? qy : -INFINITY), whz = wave_max_f(part ? qz : -INFINITY);
    const int n_supers = gp->ncells / 4096;
    int seed_super = -1;
    unsigned long long seed_done = 0;
    if (__ballot(part && !(best < INFINITY))) {
        // some query has no candidate yet: scan the blocks nearest to the patch first (any real point bounds the search)
        float m_best = INFINITY;
        int s_best = 0;
        for (int c0 = 0; c0 < n_supers; c0 += 64) {
            const int sl = c0 + lane;
            if (sl < n_supers) {
                const float m = boxbox_min_dist2(wlx, wly, wlz, whx, why, whz, supers[sl]);
                if (m < m_best) {
                    m_best = m;
                    s_best = sl;
                }
            }
        }
        const float wm = wave_min_f(m_best);
        if (wm < INFINITY) {
            const int src_lane = __ffsll((long long)__ballot(m_best == wm)) - 1;
            seed_super = __shfl(s_best, src_lane, 64);
            const Box bb = boxes[seed_super * 64 + lane];
            float mb = boxbox_min_dist2(wlx, wly, wlz, whx, why, whz, bb);
            for (int t = 0; t < kSeedBlocks; t++) {
                const float wmb = wave_min_f(mb);
                if (!(wmb < INFINITY)) break;
                const int b = __ffsll((long long)__ballot(mb == wmb)) - 1;
                const int js = __builtin_amdgcn_readlane(__float_as_int(bb.pad0), b), je = __builtin_amdgcn_readlane(__float_as_int(bb.pad1), b);
                scan_points(sorted, js, je, st, lane, qx, qy, qz, best, best_i);
                seed_done |= 1ull << b;
                if (lane == b) mb = INFINITY;
            }
        }
    }
    float bound = part ? best : -1.0f;  // a lane outside the search never opens a box
    float rmax = wave_max_f(bound);
    for (int c0 = 0; c0 < n_supers; c0 += 64) {
        const int sl = c0 + lane;
        float m = INFINITY;
        if (sl < n_supers) m = boxbox_min_dist2(wlx, wly, wlz, whx, why, whz, supers[sl]);
        unsigned long long open = __ballot(m <= rmax && m < INFINITY);
        while (open) {
            const int s = c0 + __ffsll((long long)open) - 1;
            open &= open - 1;
            // the super-block's 64 blocks, one per lane, with their point ranges
            const Box bb = boxes[s * 64 + lane];
            const float mb = boxbox_min_dist2(wlx, wly, wlz, whx, why, whz, bb);
            unsigned long long cand = __ballot(mb <= rmax && mb < INFINITY);
            if (s == seed_super) cand &= ~seed_done;
            while (cand) {
                const int b = __ffsll((long long)cand) - 1;
                cand &= cand - 1;
                Box xb;  // block b's box, broadcast from lane b
                xb.lx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb.lx), b));
                xb.ly = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb.ly), b));
                xb.lz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb.lz), b));
                xb.hx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb.hx), b));
                xb.hy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb.hy), b));
                xb.hz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb.hz), b));
                if (!__ballot(box_min_dist2(qx, qy, qz, xb) <= bound)) continue;  // no lane can find anything nearer or tied in it
                const int js = __builtin_amdgcn_readlane(__float_as_int(bb.pad0), b), je = __builtin_amdgcn_readlane(__float_as_int(bb.pad1), b);
                scan_points(sorted, js, je, st, lane, qx, qy, qz, best, best_i);
                bound = part ? best : -1.0f;
            }
            rmax = wave_max_f(bound);
            open &= __ballot(m <= rmax);  // super-blocks of this chunk that the tighter bounds rule out need not be opened
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The NEAR PATH: a query that knows a REAL target point near it walks the target grid's cells around it by itself -- a
// handful of candidates instead of its group's ~800 -- and its key is final.  What the kd-tree does for such a query
// (include/nanoflann.h:1200-1247: the leaf the query falls in and the few leaves its ball touches), with the grid's cells
// as the leaves.  Two forms:
//   seeded (nn_cull_kernel<true>)  the bound B is the f32 squared distance of the query's previous neighbour; the cells the ball
//                                  of B touches are walked, and the result is final;
//   probe (nn_probe_kernel)        no bound yet: the 27 cells around the query's own cell are walked; the result is final when
//                                  the ball of the best distance found stays inside those cells, a seed for the group search otherwise.
//
// Exactness.  dist2 = fl(fl(fl(d0*d0) + fl(d1*d1)) + fl(d2*d2)), d_a = fl(q_a - p_a).  Every term is >= 0 and rounding is
// monotone, so fl(d_a*d_a) <= dist2 for each axis.  With r such that fl(r*r) > B (checked, not assumed), any p with
// dist2(q, p) <= B has |d_a| < r on every axis; fl(q_a - p_a) is off from q_a - p_a by <= 2^-24 relative, so
// |q_a - p_a| < r (1 + 2^-23), and with r2 = r * 1.0001 + |q_a| * 2.4e-7 the f32 values lo = fl(q_a - r2), hi = fl(q_a + r2)
// satisfy lo <= p_a <= hi (the two margins cover the roundings of r2 and of the subtraction / addition).  cell_coord is
// monotone in its first argument (an f32 subtraction of o, a multiplication by inv_h > 0, floor, clamp), and it is the
// very function -- same GridParams -- that binned the targets: every such p lies in a cell of
// [cell(lo), cell(hi)]^3.  A walk over a box of cells that contains those takes (d < best) or (d == best and a lower index)
// starting from (B, that real point's index): the result is the lexicographic (distance, index) minimum over ALL targets,
// like the hierarchy's.  A target with a NaN coordinate never wins anywhere (its distance is NaN).  The path is taken only
// when the box spans <= kNearSpan cells per axis and holds <= max_pts points; everything else goes through the query
// groups below.
//
// Work split (measured, EXPERIMENTS.md R6-1: a lane -- or a team of 4 / 8 lanes -- walking its query's cells alone leaves the launch
// waiting for the few workgroups whose queries sit in crowded cells: 9-15 us of walk at the 95th percentile against 0.2 us at the
// median): the workgroup is one query group and shares ALL its candidates out evenly.  Wave 0 prepares the 64 queries (lane =
// query); a team of 4 consecutive lanes per query fetches the cell ranges of the box's runs (a row of cells along x is one run
// of consecutive cell indices inside a 4-cell block row, two when it crosses into the next block) and appends them to the
// workgroup's list in LDS as chunks of <= 8 consecutive points; then every thread takes chunks from that list -- eight
// loads in flight per chunk -- and merges the chunk's (distance, index) minimum into its query's key with
// a 64-bit LDS atomicMin (the lexicographic minimum: distances are >= +0, so their bit patterns order like their values).
constexpr int kNearSpan = 3;
constexpr int kNearRows = kNearSpan * kNearSpan;
constexpr int kNearRuns = 2 * kNearRows;     // runs per query at most
constexpr int kTeam = 4;                     // lanes per query in the team phase; the workgroup is 64 x kTeam threads
constexpr int kCullThreads = 64 * kTeam;
constexpr int kChunkPts = 8;                 // points per chunk = loads in flight per thread and chunk
constexpr int kNearCapMax = 128;             // largest candidate cap per query
constexpr int kMaxChunks = 64 * (kNearCapMax / kChunkPts + kNearRuns);   // a run of n points is ceil(n / 8) chunks

#ifdef LSN_CULL_STAMPS   // dev aid: per-workgroup clock stamps of nn_cull_kernel<true> (tools/cull_stamps.py)
__device__ long long g_cull_stamps[8 * 8192];
#define LSN_STAMP(i) do { if (APPLY && threadIdx.x == 0 && blockIdx.x < 8192) g_cull_stamps[8 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define LSN_STAMP(i) do { } while (0)
#endif

struct NearBox {   // a query's box of cells (inclusive), filled by lane = query, read by its team
    int xl, xh, yl, yh, zl, zh;
    int want;      // the query takes the near path (after near_enqueue: and its box held <= the cap)
    int pad;
};

struct NearShared {   // LDS of one query group
    float4 q[64];              // x, y, z, bound
    unsigned long long key[64];
    NearBox box[64];
    int flag[64];              // seeded: resolved; probe: bit 0 resolved, bit 1 has a key
    int n_chunks;
    int2 chunks[kMaxChunks];   // (first point, query | points << 8)
};

__device__ __forceinline__ bool near_radius(float B, const GridParams &g, float &r)
{
    r = fmaxf(sqrtf(B) * 1.00001f, 1e-18f);
    return r * r > B && r < 1e18f && g.inv_h > 0.0f && g.inv_h < INFINITY;
}

__device__ __forceinline__ void near_axis(float q, float r, float o, float inv_h, int n, int &lo, int &hi)
{
    const float r2 = r * 1.0001f + fabsf(q) * 2.4e-7f;
    lo = cell_coord(q - r2, o, inv_h, n);
    hi = cell_coord(q + r2, o, inv_h, n);
}

// lane = query: the cells the ball of B around q touches; want = it is small enough for the near path
__device__ __forceinline__ NearBox near_ball_box(const GridParams &g, bool want, float qx, float qy, float qz, float B)
{
    NearBox b = {0, 0, 0, 0, 0, 0, 0, 0};
    float r;
    if (want && near_radius(B, g, r)) {
        near_axis(qx, r, g.ox, g.inv_h, g.nx, b.xl, b.xh);
        near_axis(qy, r, g.oy, g.inv_h, g.ny, b.yl, b.yh);
        near_axis(qz, r, g.oz, g.inv_h, g.nz, b.zl, b.zh);
        b.want = b.xh - b.xl < kNearSpan && b.yh - b.yl < kNearSpan && b.zh - b.zl < kNearSpan;
    }
    return b;
}

// Team phase (all kCullThreads threads; thread = (query ql, lane sub of its team); sh.n_chunks is zero and the boxes are in LDS):
// the runs of the query's box become chunks of the workgroup's list when the box holds <= max_pts (<= kNearCapMax) points;
// sh.box[ql].want says so afterwards.
template <bool PROBE>
__device__ __forceinline__ void near_enqueue(NearShared &sh, const GridParams &g, const int *__restrict__ cell_start, int max_pts, int ql, int sub)
{
    constexpr int kMine = (kNearRuns + kTeam - 1) / kTeam;
    const NearBox bx = sh.box[ql];
    bool want = bx.want != 0;
    if (!__ballot(want)) return;   // wave-uniform
    const int xb = min(bx.xh, bx.xl | 3);
    const bool two = bx.xh > xb;
    const int nyr = bx.yh - bx.yl + 1, n_rows = nyr * (bx.zh - bx.zl + 1);
    const int n_runs = want ? (two ? 2 * n_rows : n_rows) : 0;
    int s[kMine], e[kMine];
    int total = 0;
#pragma unroll
    for (int m = 0; m < kMine; m++) {
        const int k = sub + m * kTeam;
        s[m] = e[m] = 0;
        if (k < n_runs) {
            const int row = two ? k >> 1 : k;
            const bool second = two && (k & 1);
            const int dz = (row >= nyr) + (row >= 2 * nyr), dy = row - dz * nyr;
            const int c = cell_index(second ? xb + 1 : bx.xl, bx.yl + dy, bx.zl + dz, g);
            s[m] = cell_start[c];
            e[m] = cell_start[c + (second ? bx.xh - xb : xb - bx.xl + 1)];
        }
        total += e[m] - s[m];
    }
#pragma unroll
    for (int off = 1; off < kTeam; off <<= 1) total += __shfl_xor(total, off, 64);
    const bool fits = total <= min(max_pts, kNearCapMax);
    // a crowded box: left to the group search; the probe still takes a few of its points -- any real point is a seed (want = 2)
    if (want && !fits && sub == 0) sh.box[ql].want = PROBE ? 2 : 0;
    if (PROBE && want && !fits && e[0] > s[0]) {
        const int base = atomicAdd(&sh.n_chunks, 1);
        sh.chunks[base] = make_int2(s[0], ql | (min(kChunkPts, e[0] - s[0]) << 8));
    }
    want = want && fits;
#pragma unroll
    for (int m = 0; m < kMine; m++) {
        const int n = want ? (e[m] - s[m] + kChunkPts - 1) / kChunkPts : 0;
        if (n > 0) {
            const int base = atomicAdd(&sh.n_chunks, n);   // <= kMaxChunks in total: every query's runs hold <= kNearCapMax points
            for (int c = 0; c < n; c++) {
                const int j = s[m] + kChunkPts * c;
                sh.chunks[base + c] = make_int2(j, ql | (min(kChunkPts, e[m] - j) << 8));
            }
        }
    }
}

// All threads, behind a barrier: the workgroup's chunks; sh.key[q] ends as the lexicographic minimum of
// its start value and every point of the query's box.
__device__ __forceinline__ void near_consume(NearShared &sh, const float4 *__restrict__ sorted)
{
    const int n = sh.n_chunks;
    auto fetch = [&](int2 ck, float4 (&p)[kChunkPts]) {
#pragma unroll
        for (int u = 0; u < kChunkPts; u++)
            if (u < (ck.y >> 8)) p[u] = sorted[ck.x + u];
    };
    auto merge = [&](int2 ck, const float4 (&p)[kChunkPts]) {
        const int q = ck.y & 63, cnt = ck.y >> 8;
        const float4 qq = sh.q[q];
        float b = INFINITY;
        int bi = 0x7FFFFFFF;
#pragma unroll
        for (int u = 0; u < kChunkPts; u++) {
            const float d = dist2(qq.x, qq.y, qq.z, p[u].x, p[u].y, p[u].z);
            const int i = __float_as_int(p[u].w);
            const bool take = (u < cnt) & ((d < b) | ((d == b) & (i < bi)));   // NaN: never
            b = take ? d : b;
            bi = take ? i : bi;
        }
        const unsigned long long key = pack_key(b, bi);
        if (cnt > 0 && b < INFINITY && key < sh.key[q]) atomicMin(&sh.key[q], key);
    };
    for (int c = threadIdx.x; c < n; c += kCullThreads) {
        const int2 ck = sh.chunks[c];
        float4 p[kChunkPts];
        fetch(ck, p);
        merge(ck, p);
    }
}

// The kSeedBlocks blocks nearest to the box W of a group's queries (inside the nearest super-block) become scan items of the seed
// round; what they yield seeds the real search (any real point bounds it).  One wave, lane = query; `need`: the lane takes part.
__device__ __forceinline__ void emit_seed_blocks(const GridParams *__restrict__ gp, const Box *__restrict__ boxes, const Box *__restrict__ supers,
                                                 const NnWork &wk, int bank, int g, int lane, bool need, float qx, float qy, float qz)
{
    if (!__ballot(need)) return;
    const float wlx = wave_min_f(need ? qx : INFINITY), wly = wave_min_f(need ? qy : INFINITY), wlz = wave_min_f(need ? qz : INFINITY);
    const float whx = wave_max_f(need ? qx : -INFINITY), why = wave_max_f(need ? qy : -INFINITY), whz = wave_max_f(need ? qz : -INFINITY);
    const int n_supers = gp->ncells / 4096;
    float m_best = INFINITY;
    int s_best = 0;
    for (int c0 = 0; c0 < n_supers; c0 += 64) {
        const int sl = c0 + lane;
        if (sl < n_supers) {
            const float m = boxbox_min_dist2(wlx, wly, wlz, whx, why, whz, supers[sl]);
            if (m < m_best) {
                m_best = m;
                s_best = sl;
            }
        }
    }
    const float wm = wave_min_f(m_best);
    if (!(wm < INFINITY)) return;  // no target point with comparable coordinates at all
    const int seed_super = __shfl(s_best, __ffsll((long long)__ballot(m_best == wm)) - 1, 64);
    const Box bb = boxes[seed_super * 64 + lane];
    float mb = boxbox_min_dist2(wlx, wly, wlz, whx, why, whz, bb);
    unsigned long long chosen = 0;
    for (int t = 0; t < kSeedBlocks; t++) {
        const float wmb = wave_min_f(mb);
        if (!(wmb < INFINITY)) break;
        const int b = __ffsll((long long)__ballot(mb == wmb)) - 1;
        chosen |= 1ull << b;
        if (lane == b) mb = INFINITY;
    }
    emit_ranges(wk, wk.counters + kBankInts * bank, kCntSeed, g, g, (chosen >> lane) & 1, __float_as_int(bb.pad0), __float_as_int(bb.pad1), lane);
}

// No seeds yet (first ICP iteration, lsnIcpNearest).  One workgroup per query group.  Every query's key starts empty; the probe
// (near_pts > 0) walks the 27 cells around the query's own cell: a query whose best find's ball stays inside them is settled
// (groups[g].resolved, read by nn_cull_kernel<false>), any other find is the query's seed.  When some query of the group still has
// no key, the blocks nearest to those queries become scan items of the seed round.
__global__ __launch_bounds__(kCullThreads) __attribute__((amdgpu_waves_per_eu(7, 8))) void nn_probe_kernel(const float4 *__restrict__ src, int n2, const GridParams *__restrict__ gp,
                                                                const Box *__restrict__ boxes, const Box *__restrict__ supers,
                                                                unsigned long long *best_key, GroupInfo *groups, NnWork wk, int bank,
                                                                const int *__restrict__ cell_start, const float4 *__restrict__ sorted, int near_pts)
{
    __shared__ NearShared sh;
    const int g = blockIdx.x;
    if (g * 64 >= n2) return;  // workgroup-uniform
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const GridParams grid = *gp;
    if (wave == 0) {
        const int j = g * 64 + lane;
        float4 q4 = make_float4(NAN, NAN, NAN, 0.0f);
        if (j < n2) q4 = src[j];
        const bool part = j < n2 && finite3(q4.x, q4.y, q4.z);
        NearBox b = {0, 0, 0, 0, 0, 0, 0, 0};
        if (near_pts > 0 && part && grid.inv_h > 0.0f && grid.inv_h < INFINITY) {
            const int cx = cell_coord(q4.x, grid.ox, grid.inv_h, grid.nx), cy = cell_coord(q4.y, grid.oy, grid.inv_h, grid.ny),
                      cz = cell_coord(q4.z, grid.oz, grid.inv_h, grid.nz);
            b.xl = max(cx - 1, 0); b.xh = min(cx + 1, grid.nx - 1);
            b.yl = max(cy - 1, 0); b.yh = min(cy + 1, grid.ny - 1);
            b.zl = max(cz - 1, 0); b.zh = min(cz + 1, grid.nz - 1);
            b.want = 1;
        }
        sh.q[lane] = make_float4(q4.x, q4.y, q4.z, part ? 0.0f : -1.0f);
        sh.box[lane] = b;
        sh.flag[lane] = 0;
        sh.key[lane] = kNoKey;
        if (lane == 0) sh.n_chunks = 0;
    }
    __syncthreads();
    if (near_pts > 0) {
        near_enqueue<true>(sh, grid, cell_start, near_pts, threadIdx.x / kTeam, threadIdx.x % kTeam);
        __syncthreads();
        near_consume(sh, sorted);
        __syncthreads();
        if (wave == 0 && sh.box[lane].want && sh.key[lane] != kNoKey) {
            // final when every target within the distance found lies in a cell that was walked
            const float4 q = sh.q[lane];
            const NearBox w = sh.box[lane];
            const NearBox n = near_ball_box(grid, true, q.x, q.y, q.z, key_dist(sh.key[lane]));
            const bool done = n.want && n.xl >= w.xl && n.xh <= w.xh && n.yl >= w.yl && n.yh <= w.yh && n.zl >= w.zl && n.zh <= w.zh;
            sh.flag[lane] = done && w.want == 1 ? 3 : 2;
        }
    }
    if (wave != 0) return;
    const int j = g * 64 + lane;
    const float4 me = sh.q[lane];
    const int flag = sh.flag[lane];
    if (j < n2) best_key[j] = sh.key[lane];
    const unsigned long long resolved = __ballot((flag & 1) != 0);
    if (lane == 0) {
        GroupInfo gi = {0, 0, 0, 0, 0, 0, -1.0f, 0, resolved};   // the rest is nn_cull_kernel<false>'s to fill
        groups[g] = gi;
    }
    emit_seed_blocks(gp, boxes, supers, wk, bank, g, lane, me.w >= 0.0f && !(flag & 2), me.x, me.y, me.z);
}

// One workgroup of 64 x kTeam threads per query group.
// Phase A, wave 0 with lane = query: (APPLY) move the query by the previous iteration's (T, Rn) -- icp.cpp:143-146 + :165:
// v = (v + T) * Rn, row vectors, f32, one rounding per operation -- in the sorted working copy and in the caller's array (same
// three floats at the query's original position); the query's seed (its previous neighbour's distance from where the query is
// now; without seed_targets the key the probe / the seed round left) and the cells its ball touches.  All threads (APPLY): clear
// the match keys for this iteration's claims.  The team phases: the near path (seeded form; after a probe its verdicts are
// taken from groups[g].resolved instead).  Phase B, every wave with lane = query: the box and the largest bound of the queries
// that still search, and one (group, super-block) item per super-block they may need -- the waves share out the chunks of
// super-blocks.
template <bool APPLY>
__global__ __launch_bounds__(kCullThreads) __attribute__((amdgpu_waves_per_eu(7, 8))) void nn_cull_kernel(float4 *src, float *verts2, int n2, const IcpState *st, unsigned long long *keys,
                                                               int n_keys, const GridParams *__restrict__ gp, const Box *__restrict__ supers,
                                                               const float *__restrict__ seed_targets, int n1, const int *idx,
                                                               unsigned long long *best_key, GroupInfo *groups, NnWork wk, int bank,
                                                               const int *__restrict__ cell_start, const float4 *__restrict__ sorted, int near_arg)
{
    __shared__ NearShared sh;
    // near_arg: 0 = no near path; n > 0 = candidate cap n, seeded steps when the previous step said so (st->near_next); n < 0 = cap -n, always
    const int near_pts = near_arg < 0 ? -near_arg : (!APPLY || st->near_next ? near_arg : 0);
    if (APPLY)
        for (int k = blockIdx.x * kCullThreads + threadIdx.x; k < n_keys; k += gridDim.x * kCullThreads) keys[k] = ~0ull;
    const int g = blockIdx.x;
    if (g * 64 >= n2) return;  // workgroup-uniform
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LSN_STAMP(0);
    const GridParams grid = *gp;
    if (wave == 0) {
        const int j = g * 64 + lane;
        const bool active = j < n2;
        float4 q4 = make_float4(NAN, NAN, NAN, 0.0f);
        if (active) q4 = src[j];
        const int orig = __float_as_int(q4.w);
        if (APPLY && active && st->mk > 0) {
            const float x = q4.x + st->T[0], y = q4.y + st->T[1], z = q4.z + st->T[2];
            q4.x = x * st->Rn[0] + y * st->Rn[3] + z * st->Rn[6];
            q4.y = x * st->Rn[1] + y * st->Rn[4] + z * st->Rn[7];
            q4.z = x * st->Rn[2] + y * st->Rn[5] + z * st->Rn[8];
            src[j] = q4;
            const size_t o = 3 * (size_t)orig;
            verts2[o] = q4.x;
            verts2[o + 1] = q4.y;
            verts2[o + 2] = q4.z;
        }
        const float qx = q4.x, qy = q4.y, qz = q4.z;
        const bool part = active && finite3(qx, qy, qz);
        unsigned long long key = kNoKey;
        bool settled = false;   // by the probe
        if (seed_targets) {
            if (part) {
                const int k = idx[j];   // the sorted-order copy of the previous neighbours (nn_finish_kernel): loaded together with src[j]
                if ((unsigned int)k < (unsigned int)n1) {
                    const float d = dist2(qx, qy, qz, seed_targets[3 * (size_t)k], seed_targets[3 * (size_t)k + 1], seed_targets[3 * (size_t)k + 2]);
                    if (d == d) key = pack_key(d, k);  // not NaN
                }
            }
        } else {
            if (active) key = best_key[j];
            settled = near_pts > 0 && ((groups[g].resolved >> lane) & 1);
        }
        const float B = key_dist(key);
        sh.q[lane] = make_float4(qx, qy, qz, part && !settled ? B : -1.0f);
        sh.key[lane] = key;
        sh.box[lane] = near_ball_box(grid, seed_targets && near_pts > 0 && part && B < INFINITY, qx, qy, qz, B);
        sh.flag[lane] = settled;
        if (lane == 0) sh.n_chunks = 0;
        LSN_STAMP(1);
    }
    __syncthreads();
    LSN_STAMP(2);
    if (seed_targets && near_pts > 0) {
        near_enqueue<false>(sh, grid, cell_start, near_pts, threadIdx.x / kTeam, threadIdx.x % kTeam);
        __syncthreads();
        LSN_STAMP(3);
        near_consume(sh, sorted);
        __syncthreads();
        LSN_STAMP(4);
        if (wave == 0 && sh.box[lane].want) {   // the walk covered the ball of the bound: the key is final
            sh.q[lane].w = -1.0f;
            sh.flag[lane] = 1;
        }
        __syncthreads();
    }
    const float4 me = sh.q[lane];
    if (wave == 0 && seed_targets && g * 64 + lane < n2) best_key[g * 64 + lane] = sh.key[lane];
    const bool search = me.w >= 0.0f;   // (a bound is a squared distance or +inf)
    GroupInfo gi;
    gi.wlx = wave_min_f(search ? me.x : INFINITY); gi.wly = wave_min_f(search ? me.y : INFINITY); gi.wlz = wave_min_f(search ? me.z : INFINITY);
    gi.whx = wave_max_f(search ? me.x : -INFINITY); gi.why = wave_max_f(search ? me.y : -INFINITY); gi.whz = wave_max_f(search ? me.z : -INFINITY);
    gi.rmax = wave_max_f(me.w);
    gi.pad = 0;
    gi.resolved = __ballot(sh.flag[lane] != 0);
    if (threadIdx.x == 0) groups[g] = gi;
    LSN_STAMP(5);
    if (!(gi.rmax >= 0.0f)) return;
    const int n_supers = grid.ncells / 4096;
    int *cnt = wk.counters + kBankInts * bank;
    // two passes over the wave's super-block boxes (count, then write) so that the append costs the wave ONE atomicAdd; the boxes of
    // four chunks (256 super-blocks) are loaded at once: one memory round trip per four chunks instead of one per chunk
    auto chunk_dist = [&](int c0, float (&m)[4]) {
        float4 lo[4];   // lx, ly, lz, hx
        float2 hi[4];   // hy, hz (the two spare words of a super-block's box stay where they are)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int sl = c0 + 64 * k + lane;
            if (sl < n_supers) {
                lo[k] = *reinterpret_cast<const float4 *>(&supers[sl].lx);
                hi[k] = *reinterpret_cast<const float2 *>(&supers[sl].hy);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int sl = c0 + 64 * k + lane;
            const Box b = {lo[k].x, lo[k].y, lo[k].z, lo[k].w, hi[k].x, hi[k].y, 0.0f, 0.0f};
            m[k] = sl < n_supers ? boxbox_min_dist2(gi.wlx, gi.wly, gi.wlz, gi.whx, gi.why, gi.whz, b) : INFINITY;
        }
    };
    int total = 0;
    for (int c0 = 256 * wave; c0 < n_supers; c0 += 256 * kTeam) {
        float m[4];
        chunk_dist(c0, m);
#pragma unroll
        for (int k = 0; k < 4; k++) total += __popcll(__ballot(m[k] <= gi.rmax && m[k] < INFINITY));
    }
    if (total == 0) return;
    const int seg = (g + wave) & (kSegs - 1);
    int base = 0;
    if (lane == 0) base = atomicAdd(cnt + seg * kSegStride + kCntA, total);
    base = __shfl(base, 0, 64);
    if (base + total > wk.seg_a) {
        if (lane == 0) atomicExch(cnt + kOverflow, 1);
        return;
    }
    uint2 *out = wk.list_a + (size_t)seg * wk.seg_a + base;
    for (int c0 = 256 * wave; c0 < n_supers; c0 += 256 * kTeam) {
        float m[4];
        chunk_dist(c0, m);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool open = m[k] <= gi.rmax && m[k] < INFINITY;
            const unsigned long long mask = __ballot(open);
            if (open) out[__popcll(mask & ((1ull << lane) - 1))] = make_uint2((unsigned int)g, (unsigned int)(c0 + 64 * k + lane));
            out += __popcll(mask);
        }
    }
    LSN_STAMP(6);
}

// One wave per (group, super-block) item: the super-block's 64 blocks, one per lane, against the group's box, then each
// surviving block against every query's own bound; what is left becomes scan items.
__global__ __launch_bounds__(64) void nn_blocks_kernel(const float4 *__restrict__ src, int n2, const Box *__restrict__ boxes,
                                                             const unsigned long long *__restrict__ best_key,
                                                             const GroupInfo *__restrict__ groups, NnWork wk, int bank)
{
    const int lane = threadIdx.x & 63;
    int *cnt = wk.counters + kBankInts * bank;
    if (cnt[kOverflow]) return;  // a list overflowed: the finish kernel searches from scratch
    const int wave_id = blockIdx.x, stride = gridDim.x >> 6;  // one wave per workgroup: a slot on the CU is free again as soon as its wave is done
    const int seg = wave_id & (kSegs - 1);
    const uint2 *list = wk.list_a + (size_t)seg * wk.seg_a;
    int slot = wave_id >> 6;
    uint2 item = list[min(slot, wk.seg_a - 1)];  // speculative: issued together with the length
    const int n_items = min(cnt[seg * kSegStride + kCntA], wk.seg_a);
    for (; slot < n_items; slot += stride, item = list[min(slot, wk.seg_a - 1)]) {
        const int g = (int)item.x, s = (int)item.y;
        const int j = g * 64 + lane;
        const Box bb = boxes[s * 64 + lane];
        const GroupInfo gi = groups[g];
        float4 q4 = make_float4(NAN, NAN, NAN, 0.0f);
        float bound = -1.0f;
        if (j < n2) {
            q4 = src[j];
            if (finite3(q4.x, q4.y, q4.z) && !((gi.resolved >> lane) & 1)) bound = key_dist(best_key[j]);
        }
        const float mb = boxbox_min_dist2(gi.wlx, gi.wly, gi.wlz, gi.whx, gi.why, gi.whz, bb);
        unsigned long long cand = __ballot(mb <= gi.rmax && mb < INFINITY);
        unsigned long long take = 0;
        while (cand) {
            const int b = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            Box xb;  // block b's box, broadcast from lane b
            xb.lx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb.lx), b));
            xb.ly = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb.ly), b));
            xb.lz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb.lz), b));
            xb.hx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb.hx), b));
            xb.hy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb.hy), b));
            xb.hz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bb.hz), b));
            if (__ballot(box_min_dist2(q4.x, q4.y, q4.z, xb) <= bound)) take |= 1ull << b;  // some query may find something nearer or tied in it
        }
        emit_ranges(wk, cnt, kCntB, g, g + s, (take >> lane) & 1, __float_as_int(bb.pad0), __float_as_int(bb.pad1), lane);
    }
}

// One wave per (group, point range) item: the group's 64 queries against the range's points; a query whose (distance,
// index) pair improved merges it into its key -- atomicMin on (distance bits << 32 | index) is the lexicographic minimum.
__global__ __launch_bounds__(64) void nn_scan_kernel(const float4 *__restrict__ src, int n2, const float4 *__restrict__ sorted,
                                                           unsigned long long *best_key, NnWork wk, int bank, int which)
{
    __shared__ WaveStage st;
    const int lane = threadIdx.x;
    const int *cnt = wk.counters + kBankInts * bank;
    if (cnt[kOverflow]) return;
    const int wave_id = blockIdx.x, stride = gridDim.x >> 6;  // one wave per workgroup, like nn_blocks_kernel
    const int seg = wave_id & (kSegs - 1);
    const int4 *list = wk.list_b + (size_t)seg * wk.seg_b;
    int slot = wave_id >> 6;
    int4 item = list[min(slot, wk.seg_b - 1)];  // speculative: issued together with the length
    const int n_items = min(cnt[seg * kSegStride + which], wk.seg_b);
    for (; slot < n_items; slot += stride, item = list[min(slot, wk.seg_b - 1)]) {
        const int j = item.x * 64 + lane;
        float4 q4 = make_float4(NAN, NAN, NAN, 0.0f);
        unsigned long long key = kNoKey;
        if (j < n2) {
            q4 = src[j];
            key = best_key[j];
        }
        float best = key_dist(key);
        int best_i = key_index(key);
        scan_points(sorted, item.y, item.z, st, lane, q4.x, q4.y, q4.z, best, best_i);
        const unsigned long long now = pack_key(best, best_i);
        if (j < n2 && now < key) atomicMin(&best_key[j], now);
    }
}

// One thread per query: the neighbour index and squared distance at the query's ORIGINAL position, and the one-to-one claim.
// When an item list overflowed, every group first runs the complete walk (seeded with whatever its keys hold).
__global__ __launch_bounds__(kThreads) void nn_finish_kernel(const float4 *__restrict__ src, int n2, const GridParams *__restrict__ gp,
                                                             const float4 *__restrict__ sorted, const Box *__restrict__ boxes,
                                                             const Box *__restrict__ supers, const unsigned long long *best_key, int *idx,
                                                             float *dist, int *idx_sorted, unsigned long long *keys, NnWork wk, int bank)
{
    __shared__ WaveStage s_stage[kThreads / 64];
    __shared__ int s_claim_k[kClaimSlots];
    __shared__ unsigned long long s_claim_v[kClaimSlots];
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * kThreads + threadIdx.x;
    int *cnt = wk.counters + kBankInts * bank;
    const bool overflow = cnt[kOverflow] != 0;
    // the other bank served the previous step and is read by nobody any more: clear it for the next step
    if (blockIdx.x == 0)
        for (int t = threadIdx.x; t < kBankInts; t += kThreads) wk.counters[kBankInts * (1 - bank) + t] = 0;
    const bool active = j < n2;
    float4 q4 = make_float4(NAN, NAN, NAN, 0.0f);
    unsigned long long key = kNoKey;
    if (active) {
        q4 = src[j];
        key = best_key[j];
    }
    float best = key_dist(key);
    int best_i = key_index(key);
    if (overflow) wave_search(gp, sorted, boxes, supers, s_stage[threadIdx.x >> 6], lane, q4.x, q4.y, q4.z, active && finite3(q4.x, q4.y, q4.z), best, best_i);
    const int orig = __float_as_int(q4.w);
    if (active) {
        if (best_i == 0x7FFFFFFF) best_i = 0;  // nothing comparable (NaN everywhere): keep the index in range
        idx[orig] = best_i;
        dist[orig] = best;
        if (idx_sorted) idx_sorted[j] = best_i;   // the next iteration's seeds, where nn_cull_kernel reads them with the queries
    }
    if (keys) claim_targets(keys, active, best_i, best, orig, s_claim_k, s_claim_v);
}

// Brute force (nn_mode 0, the ablation leg).  A lane owns one query; the targets of a step are wave-uniform, so they are fetched
// with scalar loads (s_load_dwordx16 + x8 = 8 points of the AoS cloud as it stands) and enter the VALU ops as SGPR operands:
// no tile loads, no barriers, no LDS traffic in the loop.  Targets are visited in index order and a lane is only updated on
// a strictly smaller distance, so the lowest index wins ties inside a slice; the launch cuts the target cloud into slices
// (grid.y) whose (distance, index) keys are combined with a 64-bit atomicMin (lexicographic: smaller distance, then lower
// index) and nn_brute_finish_kernel writes the results.
// Measured on the way (configs[1], 1.2e10 pairs): LDS-tiled with broadcast ds_read_b128 and packed f32, one workgroup per 256
// queries over all targets 2.79 ms (435 workgroups = 1.7 waves per SIMD, the kernel lasting as long as its busiest CU);
// the same in 5 slices with the next step's LDS reads prefetched 2.40; scalar loads 2.13; ~75 slices (32 k workgroups of
// ~1500 points) 1.77 ms = 54 TFLOP/s.  Ceiling (tools/pk_rate.hip): packed or not, dependent f32 add / mul chains retire at
// 60-61 TFLOP/s chip-wide -- without FMA (contraction is off by contract) half of the nominal 157 is out of reach.
__global__ __launch_bounds__(kThreads) void nn_brute_kernel(const float *__restrict__ targets, int n1, const float4 *__restrict__ src, int n2,
                                                            int slice_points /* a multiple of 8 */, unsigned long long *best_key)
{
    const int j = blockIdx.x * kThreads + threadIdx.x;
    const float4 q4 = j < n2 ? src[j] : make_float4(NAN, NAN, NAN, 0.0f);
    const float qx = q4.x, qy = q4.y, qz = q4.z;
    float best = INFINITY;
    int best_i = 0x7FFFFFFF;
    const int first = (int)min((long long)n1, (long long)blockIdx.y * slice_points);
    const int last = (int)min((long long)n1, (long long)first + slice_points);
    const int full = first + ((last - first) & ~7);   // whole steps of 8 points
    const float *__restrict__ tp = targets + 3 * (size_t)first;
    // two register (SGPR) buffers used alternately, so the prefetched step needs no copy
    float bufA[24], bufB[24];
    const int last_step = max(full - 8, first);
    auto fetch = [&](float (&buf)[24], int t) {   // unconditional (a prefetch past the end re-reads the last step): no wait lands behind the loads
        const float *__restrict__ p = tp + 3 * (min(t, last_step) - first);
#pragma unroll
        for (int c = 0; c < 24; c++) buf[c] = p[c];
    };
    auto step = [&](const float (&buf)[24], int t) {
        float d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {   // the order (dx*dx + dy*dy) + dz*dz of PointCloud::kdtree_distance (icp.h:40-47); contraction is off
            const float dx = qx - buf[3 * k], dy = qy - buf[3 * k + 1], dz = qz - buf[3 * k + 2];
            d[k] = dx * dx + dy * dy + dz * dz;
        }
        const float m = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));   // fminf skips NaN
        if (__ballot(m < best)) {   // wave-uniform; rare once the running minima have settled
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (d[k] < best) { best = d[k]; best_i = t + k; }
        }
    };
    if (first < full) fetch(bufA, first);
    for (int t = first; t < full; t += 16) {
        // scalar loads return out of order, so the only wait is "all of them": a buffer's first use must come BEFORE the next
        // fetch is issued, or that wait would cover the fresh loads too (the empty asm pins that order)
        fetch(bufB, t + 8);
        step(bufA, t);
        asm volatile("" ::"s"(bufB[0]));
        fetch(bufA, t + 16);
        if (t + 8 < full) step(bufB, t + 8);
        asm volatile("" ::"s"(bufA[0]));
    }
    for (int t = full; t < last; t++) {   // the cloud's last < 8 points
        const float dx = qx - targets[3 * (size_t)t], dy = qy - targets[3 * (size_t)t + 1], dz = qz - targets[3 * (size_t)t + 2];
        const float d = dx * dx + dy * dy + dz * dz;
        if (d < best) { best = d; best_i = t; }
    }
    if (j < n2 && best_i != 0x7FFFFFFF) atomicMin(&best_key[j], pack_key(best, best_i));
}

__global__ __launch_bounds__(kThreads) void nn_brute_init_kernel(unsigned long long *best_key, int n2)
{
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j < n2) best_key[j] = kNoKey;
}

// after a sliced brute-force search: results to the queries' original positions + the one-to-one claims
__global__ __launch_bounds__(kThreads) void nn_brute_finish_kernel(const float4 *__restrict__ src, int n2, const unsigned long long *best_key, int *idx,
                                                                   float *dist, unsigned long long *keys)
{
    __shared__ int s_claim_k[kClaimSlots];
    __shared__ unsigned long long s_claim_v[kClaimSlots];
    const int j = blockIdx.x * kThreads + threadIdx.x;
    const bool active = j < n2;
    int orig = 0, best_i = 0;
    float best = INFINITY;
    if (active) {
        const unsigned long long key = best_key[j];
        orig = __float_as_int(src[j].w);
        best = key_dist(key);
        best_i = key_index(key);
        if (best_i == 0x7FFFFFFF) best_i = 0;
        idx[orig] = best_i;
        dist[orig] = best;
    }
    if (keys) claim_targets(keys, active, best_i, best, orig, s_claim_k, s_claim_v);
}

// ---- matching statistics and Kabsch sums ----------------------------------------------------------------------

__device__ __forceinline__ bool is_winner(const unsigned long long *keys, const int *idx, int i)
{
    return (unsigned int)(keys[idx[i]] & 0xFFFFFFFFull) == 0xFFFFFFFFu - (unsigned int)i;
}

// pass 1: m = number of one-to-one matches, the sum of their squared distances and the sum of the squares of those
// (fourth sum, for the next step's choice of path: the queries within one target cell edge of their neighbour; gp null: brute force)
__global__ __launch_bounds__(kThreads) void stats_kernel(const int *idx, const float *dist, const unsigned long long *keys, int n2,
                                                         const GridParams *gp, double *part /* [blocks][4]: count, sum d, sum d^2, near */)
{
    __shared__ double lds[4 * 4];
    const float h2 = gp ? gp->h * gp->h : -1.0f;
    double v[4] = {0, 0, 0, 0};
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n2; i += gridDim.x * kThreads) {
        const float df = dist[i];
        if (df <= h2) v[3] += 1.0;
        if (is_winner(keys, idx, i)) {
            const double d = (double)df;
            v[0] += 1.0;
            v[1] += d;
            v[2] += d * d;
        }
    }
    block_sum_d<4>(v, lds);
    if (threadIdx.x < 4) {
        double out = v[0];
        if (threadIdx.x == 1) out = v[1];
        if (threadIdx.x == 2) out = v[2];
        if (threadIdx.x == 3) out = v[3];
        part[blockIdx.x * 4 + threadIdx.x] = out;
    }
}

// pass 2: mean and standard deviation of the matches' squared distances (GetStandardDeviation, icp.cpp:34-54: the mean is
// rounded to f32, the deviations are taken from that f32 mean, the sum is kept in a float and divided by the count),
// reject d > 2.5*std (icp.cpp:56-73), accumulate count, sum m1, sum m2, sum m2 m1^T over the kept matches.
// sum (d - mean)^2 = sum d^2 - 2 mean sum d + m mean^2 is evaluated in double from pass 1's three sums (mean = the f32
// value): one pass over the matches instead of two; the cancellation costs ~1e-15 relative, far below the f32 result.
__global__ __launch_bounds__(kThreads) void accum_kernel(const float *verts1, const float *verts2, const int *idx, const float *dist,
                                                         const unsigned long long *keys, int n2, const double *part1, int n_part1,
                                                         double *part /* [blocks][16] */, IcpState *st)
{
    __shared__ double lds[4 * 16];
    __shared__ double red[kThreads + 4];
    double s1[4];
    reduce_partials<4, 4>(part1, n_part1, s1, red);
    const float mean = (float)(s1[1] / s1[0]);
    const float m_f = (float)(int)s1[0];
    const double dev = s1[2] - 2.0 * (double)mean * s1[1] + s1[0] * (double)mean * (double)mean;
    float sd = (float)(dev > 0.0 ? dev : 0.0);  // the reference keeps the running sum in a float
    sd = sd / m_f;
    sd = sqrtf(sd);
    const float thresh = 2.5f * sd;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->m = (int)s1[0];
        st->mean = mean;
        st->stddev = sd;
        st->thresh = thresh;
        st->near_next = 2.0 * s1[3] >= (double)n2;
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
// V0 (nullable): an orthonormal matrix to start from (B = A V0): any orthonormal start gives the same decomposition up to
// rounding; a good one saves sweeps.
__device__ void svd3(const double A[9], double U[9], double w[3], double V[9], const double *V0)
{
    double B[9];
    for (int i = 0; i < 9; i++) { B[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    if (V0) {
        for (int i = 0; i < 9; i++) V[i] = V0[i];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) B[3 * r + c] = A[3 * r] * V0[c] + A[3 * r + 1] * V0[3 + c] + A[3 * r + 2] * V0[6 + c];
    }
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
                // tan of the rotation: sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) with zeta = (b - a) / (2c), multiplied through
                // by 2|c| -- one division and one square root per rotation instead of three and two (this thread is the
                // critical path of an ICP iteration)
                const double ba = b - a;
                const double tt = ((ba >= 0) == (c >= 0) ? 2.0 : -2.0) * fabs(c) / (fabs(ba) + sqrt(ba * ba + 4.0 * c * c));
                const double cs = rsqrt(1.0 + tt * tt), sn = cs * tt;
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
    for (int j = 0; j < 3; j++) {
        const double inv = (w[j] > 1e-300) ? 1.0 / w[j] : 0.0;
        for (int k = 0; k < 3; k++) U[3 * k + j] = Bs[3 * k + j] * inv;
    }
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
    __shared__ double red[kThreads + 16];
    double s[16];
    reduce_partials<16, 16>(part3, n_part3, s, red);
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
        svd3(M, U, w, V, st->v_valid == 1 ? st->Vprev : nullptr);
        for (int i = 0; i < 9; i++) st->Vprev[i] = V[i];
        st->v_valid = 1;
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

// icp.cpp:143-146 + :165: v = (v + T) * Rn, row vectors, f32, one rounding per operation.  Thread j moves query j of the
// sorted working copy and stores the same three floats into the caller's array at the query's original position; the
// same launch clears the match keys for the next iteration's atomicMin.
__global__ __launch_bounds__(kThreads) void apply_kernel(float4 *src, float *verts2, int n2, const IcpState *st, unsigned long long *keys,
                                                         int n_keys)
{
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j < n_keys) keys[j] = ~0ull;
    if (st->mk <= 0) return;
    const float T0 = st->T[0], T1 = st->T[1], T2 = st->T[2];
    const float r0 = st->Rn[0], r1 = st->Rn[1], r2 = st->Rn[2], r3 = st->Rn[3], r4 = st->Rn[4], r5 = st->Rn[5], r6 = st->Rn[6],
                r7 = st->Rn[7], r8 = st->Rn[8];
    if (j >= n2) return;
    float4 q = src[j];
    const float x = q.x + T0, y = q.y + T1, z = q.z + T2;
    q.x = x * r0 + y * r3 + z * r6;
    q.y = x * r1 + y * r4 + z * r7;
    q.z = x * r2 + y * r5 + z * r8;
    src[j] = q;
    const size_t o = 3 * (size_t)__float_as_int(q.w);
    verts2[o] = q.x;
    verts2[o + 1] = q.y;
    verts2[o + 2] = q.z;
}

// What lsnIcpRun clears before its first iteration.
// seeds (nullable): neighbours handed in from outside, by the queries' ORIGINAL index (lsnRefine: the previous Gauss-Seidel pass's) -- they go
// to the sorted order the first NN step reads; that step then is a seeded one with no motion to apply (mk = 0).
__global__ __launch_bounds__(kThreads) void run_init_kernel(unsigned long long *keys, int n_keys, int *counters, int n_counters, IcpState *st,
                                                            const int *__restrict__ seeds, const float4 *__restrict__ src, int n2, int *idx_sorted)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n_keys) keys[i] = ~0ull;
    if (i < n_counters) counters[i] = 0;
    if (i == 0) {
        st->v_valid = 0;
        st->mk = 0;
        if (seeds) st->near_next = 1;   // (the first step has no statistics of a previous one: seeds from a converged pass are near)
    }
    if (seeds && i < n2) idx_sorted[i] = seeds[__float_as_int(src[i].w)];
}

}  // namespace

// -------------------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------------------

// One voxel grid over a cloud: the cell-sorted copy (x, y, z, original index) and, for a target, the box hierarchy.
struct GridBufs {
    lsn::DevBuf gp, cell_of, rank_of, cell_cnt, cell_start, sorted, boxes, supers;
    bool counts_clear = false;   // cell_cnt is all zero (cell_scatter_kernel writes every touched count back to zero: one memset per workspace, not per build)
    int reserve(int max_n, bool with_boxes)
    {
        int bad = 0;
        bad |= gp.reserve(sizeof(GridParams));
        bad |= cell_of.reserve(sizeof(int) * (size_t)max_n);
        bad |= rank_of.reserve(sizeof(int) * (size_t)max_n);
        bad |= cell_cnt.reserve(sizeof(int) * (size_t)kMaxCells);
        bad |= cell_start.reserve(sizeof(int) * ((size_t)kMaxCells + kScanBlock));
        bad |= sorted.reserve(sizeof(float4) * (size_t)max_n);
        if (with_boxes) {
            bad |= boxes.reserve(sizeof(Box) * (size_t)kMaxBlocks3);
            bad |= supers.reserve(sizeof(Box) * (size_t)kMaxSupers);
        }
        return bad;
    }
};

struct LsnIcp {
    int device = 0;
    int max_n1 = 0, max_n2 = 0;
    float cell_override = 0.0f;
    GridBufs tgt, src;    // tgt: cell-sorted target + boxes; src.sorted: the spatially sorted working copy of the source
    lsn::DevBuf bbox_part, block_sums;
    lsn::DevBuf idx, dist, keys, counters, part1, part3, state, trace;
    lsn::DevBuf best_key, groups, list_a, list_b, idx_sorted;   // the NN step's per-query keys, per-group boxes and work lists
    int seg_a = 0, seg_b = 0;   // capacity of one list segment
    int item_points = 128;      // points per scan item; $LSN_ICP_ITEM (tuning): 64, 128, 192 or 256.  128 instead of 256: 0.094 -> 0.089 ms/iteration (configs[1])
    // optional phase timing of lsnIcpRun (lsnIcpSetProfiling): HIP events on the caller's stream around
    // [0] grid build + source sort, [1] the NN steps (incl. the fused apply), [2] statistics + Kabsch sums + solve, [3] final apply
    bool profiling = false;
    std::vector<hipEvent_t> events;
    std::vector<int> event_phase;   // phase that ENDS at event k (event 0 opens the run)
    size_t n_events = 0;
    int trace_iters = 0;
    bool seed_nn = true;   // $LSN_ICP_NO_SEED=1 turns the previous-neighbour seeding off (ablation)
    int near_mode = 1;     // $LSN_ICP_NEAR: 0 = no near path, 1 = the probe always, seeded steps when the previous step's distances say it pays, 2 = always
    int near_pts = 128;    // the near path's candidate cap per query (<= kNearCapMax); $LSN_ICP_NEAR=0 turns the path off (A/B, tests), $LSN_ICP_NEAR_PTS sets the cap
    int last_groups = 0;   // query groups of the last grid NN step (lsnIcpNearResolved)
    std::mutex mu;
};

static constexpr int kTraceCap = 1024;

static LsnIcp * lsnIcpCreate_impl(int device, int max_n1, int max_n2)
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
    if (const char *e3 = getenv("LSN_ICP_NEAR_PTS")) w->near_pts = std::max(1, std::min(kNearCapMax, atoi(e3)));
    if (const char *e4 = getenv("LSN_ICP_NEAR")) w->near_mode = std::max(0, std::min(2, atoi(e4)));
    const char *env = getenv("LSN_ICP_CELL");
    if (env) w->cell_override = (float)atof(env);
    bool bad = false;
    bad |= w->tgt.reserve(max_n1, true) != 0;
    bad |= w->src.reserve(max_n2, false) != 0;
    bad |= w->bbox_part.reserve(sizeof(float) * 6 * kMaxBlocks) != 0;
    bad |= w->block_sums.reserve(sizeof(int) * 2048) != 0;
    bad |= w->idx.reserve(sizeof(int) * (size_t)max_n2) != 0;
    bad |= w->dist.reserve(sizeof(float) * (size_t)max_n2) != 0;
    bad |= w->keys.reserve(sizeof(unsigned long long) * (size_t)max_n1) != 0;
    bad |= w->counters.reserve(sizeof(int) * 2 * kBankInts) != 0;
    {
        const int n_groups = (max_n2 + 63) / 64;
        // generous: a seeded group needs ~5 super-blocks and ~10 point ranges; $LSN_ICP_TINY_LISTS=1 forces the overflow path (tests)
        const bool tiny = getenv("LSN_ICP_TINY_LISTS") && atoi(getenv("LSN_ICP_TINY_LISTS")) != 0;
        if (const char *e = getenv("LSN_ICP_ITEM")) {
            const int v = atoi(e);
            if (v == 64 || v == 128 || v == 192 || v == 256) w->item_points = v;
        }
        w->seg_a = tiny ? 2 : 1024 + n_groups / 4;   // x 64 segments: 64 k + 16 per group
        w->seg_b = tiny ? 2 : 8192 + 2 * n_groups;   // x 64 segments: 512 k + 128 per group
        bad |= w->best_key.reserve(sizeof(unsigned long long) * (size_t)max_n2) != 0;
        bad |= w->idx_sorted.reserve(sizeof(int) * (size_t)max_n2) != 0;
        bad |= w->groups.reserve(sizeof(GroupInfo) * (size_t)n_groups) != 0;
        bad |= w->list_a.reserve(sizeof(uint2) * (size_t)w->seg_a * kSegs) != 0;
        bad |= w->list_b.reserve(sizeof(int4) * (size_t)w->seg_b * kSegs) != 0;
    }
    bad |= w->part1.reserve(sizeof(double) * 4 * kMaxBlocks) != 0;
    bad |= w->part3.reserve(sizeof(double) * 16 * kMaxBlocks) != 0;
    bad |= w->state.reserve(sizeof(IcpState)) != 0;
    bad |= w->trace.reserve(sizeof(float) * 16 * kTraceCap) != 0;
    if (bad) {
        delete w;
        return nullptr;
    }
    return w;
}

extern "C" LsnIcp * lsnIcpCreate(int device, int max_n1, int max_n2)
{
    return lsn::guarded<LsnIcp *>("lsnIcpCreate", static_cast<LsnIcp *>(nullptr), [&]() { return lsnIcpCreate_impl(device, max_n1, max_n2); });
}

static void lsnIcpDestroy_impl(LsnIcp *w)
{
    if (!w) return;
    (void)hipSetDevice(w->device);
    for (hipEvent_t e : w->events) (void)hipEventDestroy(e);
    delete w;
}

extern "C" void lsnIcpDestroy(LsnIcp *w)
{
    lsn::guarded_void("lsnIcpDestroy", [&]() { lsnIcpDestroy_impl(w); });
}

static int lsnIcpSetProfiling_impl(LsnIcp *w, int on)
{
    lsn::clear_error();
    if (!w) return -1;
    std::lock_guard<std::mutex> g(w->mu);
    w->profiling = on != 0;
    w->n_events = 0;
    return 0;
}

extern "C" int lsnIcpSetProfiling(LsnIcp *w, int on)
{
    return lsn::guarded<int>("lsnIcpSetProfiling", static_cast<int>(-1), [&]() { return lsnIcpSetProfiling_impl(w, on); });
}

// records the end of `phase` on the stream (profiling only)
static void mark(LsnIcp *w, int phase, hipStream_t s)
{
    if (!w->profiling) return;
    if (w->n_events == w->events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        w->events.push_back(e);
        w->event_phase.push_back(0);
    }
    w->event_phase[w->n_events] = phase;
    (void)hipEventRecord(w->events[w->n_events++], s);
}

static int lsnIcpProfile_impl(LsnIcp *w, float *ms4, void *stream)
{
    lsn::clear_error();
    if (!w || !ms4) return -1;
    std::lock_guard<std::mutex> g(w->mu);
    LSN_HIP(hipSetDevice(w->device));
    LSN_HIP(hipStreamSynchronize(lsn::as_stream(stream)));
    for (int k = 0; k < 4; k++) ms4[k] = 0.0f;
    for (size_t k = 1; k < w->n_events; k++) {
        float ms = 0.0f;
        LSN_HIP(hipEventElapsedTime(&ms, w->events[k - 1], w->events[k]));
        ms4[w->event_phase[k] & 3] += ms;
    }
    return (int)w->n_events;
}

extern "C" int lsnIcpProfile(LsnIcp *w, float *ms4, void *stream)
{
    return lsn::guarded<int>("lsnIcpProfile", static_cast<int>(-1), [&]() { return lsnIcpProfile_impl(w, ms4, stream); });
}

static inline int blocks_for(int n) { return (n + kThreads - 1) / kThreads; }
static inline int capped_blocks(int n) { int b = blocks_for(n); return b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b); }

// Builds the voxel grid over a cloud (stream ordered, no host synchronisation): g.sorted = the points in cell order with
// their original index; with_boxes adds the block / super-block AABBs the query kernel culls with.
static int build_grid(LsnIcp *w, GridBufs &g, const float *d_pts, int n, bool with_boxes, hipStream_t s)
{
    const int nb = capped_blocks(n);
    GridParams *gp = g.gp.as<GridParams>();
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(nb), dim3(kThreads), 0, s, d_pts, n, w->bbox_part.as<float>());
    hipLaunchKernelGGL(grid_setup_kernel, dim3(1), dim3(64), 0, s, w->bbox_part.as<float>(), nb, n, w->cell_override, gp);
    if (!g.counts_clear) LSN_HIP(hipMemsetAsync(g.cell_cnt.p, 0, sizeof(int) * (size_t)kMaxCells, s));
    g.counts_clear = false;   // dirty until the scatter below has been enqueued
    const int nall = std::max(1, blocks_for(n));   // one point per thread: a capped grid makes every thread a chain of dependent rounds
    hipLaunchKernelGGL(cell_count_kernel, dim3(nall), dim3(kThreads), 0, s, d_pts, n, gp, g.cell_of.as<int>(), g.rank_of.as<int>(), g.cell_cnt.as<int>());
    const int sb = kMaxCells / kScanBlock;  // 1024
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(sb), dim3(kThreads), 0, s, g.cell_cnt.as<int>(), gp, w->block_sums.as<int>());
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, s, w->block_sums.as<int>(), sb);
    hipLaunchKernelGGL(scan_finish_kernel, dim3(sb + 1), dim3(kThreads), 0, s, g.cell_cnt.as<int>(), gp, w->block_sums.as<int>(),
                       g.cell_start.as<int>());
    hipLaunchKernelGGL(cell_scatter_kernel, dim3(nall), dim3(kThreads), 0, s, d_pts, n, g.cell_of.as<int>(), g.rank_of.as<int>(), g.cell_start.as<int>(),
                       g.cell_cnt.as<int>(), g.sorted.as<float4>());
    LSN_HIP(hipGetLastError());   // (a failed launch leaves counts_clear false: the next build clears them again)
    g.counts_clear = true;
    if (with_boxes) {
        hipLaunchKernelGGL(block_box_kernel, dim3(2048), dim3(kThreads), 0, s, (const GridParams *)gp, (const int *)g.cell_start.as<int>(),
                           (const float4 *)g.sorted.as<float4>(), g.boxes.as<Box>());
        hipLaunchKernelGGL(super_box_kernel, dim3(kMaxSupers), dim3(64), 0, s, (const GridParams *)gp, (const Box *)g.boxes.as<Box>(),
                           g.supers.as<Box>());
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

static NnWork work_of(LsnIcp *w)
{
    NnWork wk;
    wk.list_a = w->list_a.as<uint2>();
    wk.list_b = w->list_b.as<int4>();
    wk.seg_a = w->seg_a;
    wk.seg_b = w->seg_b;
    wk.item_points = w->item_points;
    wk.counters = w->counters.as<int>();
    return wk;
}


// One NN step over the sorted working copy of the source (w->src.sorted, n2 queries); results land at the queries'
// ORIGINAL positions in d_idx / d_dist.  seeded: d_idx holds every query's previous neighbour (ICP iterations > 0); the
// first launch then also applies the previous iteration's motion (st) to the source and clears `keys`.
// `bank` alternates between consecutive steps on a stream (the counters of the other bank are cleared meanwhile).
static int run_nn(LsnIcp *w, const float *d_verts1, int n1, float *d_verts2, int n2, int *d_idx, float *d_dist, unsigned long long *keys,
                  int nn_mode, hipStream_t s, bool seeded, const IcpState *st, int bank)
{
    float4 *src = w->src.sorted.as<float4>();
    if (nn_mode == 0) {
        // slices of the target cloud: ~32 k workgroups of a few hundred to a few thousand points each (measured best at configs[1]
        // and [2]: 2 k workgroups 2.13 ms, 8 k 1.80, 16-32 k 1.77, 64 k 1.83)
        static const int want_wgs = getenv("LSN_ICP_BRUTE_WGS") ? std::max(1, atoi(getenv("LSN_ICP_BRUTE_WGS"))) : 32768;
        const int qblocks = blocks_for(n2);
        int slices = std::max(1, std::min(65535, (want_wgs + qblocks - 1) / qblocks));
        const int slice_points = std::max(64, (((n1 + slices - 1) / slices) + 7) & ~7);
        slices = std::max(1, (n1 + slice_points - 1) / slice_points);
        unsigned long long *best_key = w->best_key.as<unsigned long long>();
        hipLaunchKernelGGL(nn_brute_init_kernel, dim3(qblocks), dim3(kThreads), 0, s, best_key, n2);
        hipLaunchKernelGGL(nn_brute_kernel, dim3(qblocks, slices), dim3(kThreads), 0, s, d_verts1, n1, (const float4 *)src, n2, slice_points, best_key);
        hipLaunchKernelGGL(nn_brute_finish_kernel, dim3(qblocks), dim3(kThreads), 0, s, (const float4 *)src, n2,
                           (const unsigned long long *)best_key, d_idx, d_dist, keys);
        LSN_HIP(hipGetLastError());
        return 0;
    }
    const GridParams *gp = w->tgt.gp.as<GridParams>();
    const float4 *sorted = w->tgt.sorted.as<float4>();
    const Box *boxes = w->tgt.boxes.as<Box>(), *supers = w->tgt.supers.as<Box>();
    const int *cell_start = w->tgt.cell_start.as<int>();
    unsigned long long *best_key = w->best_key.as<unsigned long long>();
    GroupInfo *groups = w->groups.as<GroupInfo>();
    w->last_groups = (n2 + 63) / 64;
    const NnWork wk = work_of(w);
    const int n_groups = (n2 + 63) / 64;
    // consumer launches (one wave per workgroup, a multiple of 64 of them): a seeded group needs ~3-5 super-blocks and ~5-8
    // point ranges; longer lists are served by looping.  Waves without an item leave after one load.
    // waves per query group of the two list consumers.  The scan list of a seeded step holds ~6 (configs[1]) to ~11 (configs[2]) items
    // per group, unevenly over the 64 segments: with fewer waves than the fullest segment has items, some waves serve a second item
    // behind their first -- a second chain of dependent round trips that sets the launch time ($LSN_ICP_SCAN_WAVES, A/B in EXPERIMENTS.md)
    static const int scan_waves = getenv("LSN_ICP_SCAN_WAVES") ? std::max(1, atoi(getenv("LSN_ICP_SCAN_WAVES"))) : 12;
    static const int block_waves = getenv("LSN_ICP_BLOCK_WAVES") ? std::max(1, atoi(getenv("LSN_ICP_BLOCK_WAVES"))) : 8;
    const dim3 blocks_grid(64 * ((block_waves * n_groups + 63) / 64)), scan_grid(64 * ((scan_waves * n_groups + 63) / 64));
    if (seeded) {
        hipLaunchKernelGGL(nn_cull_kernel<true>, dim3(n_groups), dim3(kCullThreads), 0, s, src, d_verts2, n2, st, keys, n1, gp, supers, d_verts1, n1,
                           (const int *)w->idx_sorted.as<int>(), best_key, groups, wk, bank, cell_start, sorted,
                           w->near_mode == 0 ? 0 : (w->near_mode == 2 ? -w->near_pts : w->near_pts));
    } else {
        hipLaunchKernelGGL(nn_probe_kernel, dim3(n_groups), dim3(kCullThreads), 0, s, (const float4 *)src, n2, gp, boxes, supers, best_key, groups, wk, bank,
                           cell_start, sorted, w->near_mode == 0 ? 0 : w->near_pts);
        hipLaunchKernelGGL(nn_scan_kernel, scan_grid, dim3(64), 0, s, (const float4 *)src, n2, sorted, best_key, wk, bank, kCntSeed);
        hipLaunchKernelGGL(nn_cull_kernel<false>, dim3(n_groups), dim3(kCullThreads), 0, s, src, (float *)nullptr, n2, (const IcpState *)nullptr,
                           (unsigned long long *)nullptr, 0, gp, supers, (const float *)nullptr, n1, (const int *)nullptr, best_key, groups, wk, bank,
                           cell_start, sorted, w->near_mode == 0 ? 0 : w->near_pts);
    }
    hipLaunchKernelGGL(nn_blocks_kernel, blocks_grid, dim3(64), 0, s, (const float4 *)src, n2, boxes,
                       (const unsigned long long *)best_key, (const GroupInfo *)groups, wk, bank);
    hipLaunchKernelGGL(nn_scan_kernel, scan_grid, dim3(64), 0, s, (const float4 *)src, n2, sorted, best_key, wk, bank, kCntB);
    hipLaunchKernelGGL(nn_finish_kernel, dim3(blocks_for(n2)), dim3(kThreads), 0, s, (const float4 *)src, n2, gp, sorted, boxes, supers,
                       (const unsigned long long *)best_key, d_idx, d_dist, keys ? w->idx_sorted.as<int>() : (int *)nullptr, keys, wk, bank);
    LSN_HIP(hipGetLastError());
    static const bool debug = getenv("LSN_ICP_DEBUG") != nullptr;   // dev aid: synchronises and prints the work-list sizes and the grid
    if (debug) {
        int c[2 * kBankInts];
        GridParams g;
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(c, w->counters.p, sizeof(c), hipMemcpyDeviceToHost);   // (the other bank has been cleared by now)
        (void)hipMemcpy(&g, w->tgt.gp.p, sizeof(g), hipMemcpyDeviceToHost);
        const int *b = c + kBankInts * bank;
        long long na = 0, nb = 0, ns = 0;
        int ma = 0, mb = 0;
        for (int k = 0; k < kSegs; k++) {
            const int *c3 = b + k * kSegStride;
            na += c3[kCntA]; nb += c3[kCntB]; ns += c3[kCntSeed];
            ma = c3[kCntA] > ma ? c3[kCntA] : ma;
            mb = c3[kCntB] > mb ? c3[kCntB] : mb;
            mb = c3[kCntSeed] > mb ? c3[kCntSeed] : mb;
        }
        fprintf(stderr, "[lsn icp] n1=%d n2=%d groups=%d seeded=%d h=%g grid=%dx%dx%d supers=%d; items: super-blocks %lld (fullest segment %d of %d), ranges %lld + seed ranges %lld (fullest segment %d of %d), overflow %d\n",
                n1, n2, n_groups, (int)seeded, (double)g.h, g.nx, g.ny, g.nz, g.ncells / 4096, na, ma, w->seg_a, nb, ns, mb, w->seg_b, b[kOverflow]);
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

static int lsnIcpNearest_impl(LsnIcp *w, const float *d_verts1, int n1, const float *d_verts2, int n2, int *d_idx, float *d_dist2,
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
    if (nn_mode != 0 && build_grid(w, w->tgt, d_verts1, n1, true, s)) return -1;
    if (build_grid(w, w->src, d_verts2, n2, false, s)) return -1;
    LSN_HIP(hipMemsetAsync(w->counters.p, 0, sizeof(int) * 2 * kBankInts, s));
    return run_nn(w, d_verts1, n1, nullptr, n2, d_idx, d_dist2, nullptr, nn_mode, s, false, nullptr, 0);
}

extern "C" int lsnIcpNearest(LsnIcp *w, const float *d_verts1, int n1, const float *d_verts2, int n2, int *d_idx, float *d_dist2,
                             int nn_mode, void *stream)
{
    return lsn::guarded<int>("lsnIcpNearest", static_cast<int>(-1), [&]() { return lsnIcpNearest_impl(w, d_verts1, n1, d_verts2, n2, d_idx, d_dist2, nn_mode, stream); });
}

// d_seeds (nullable, lsnRefine): n2 target indices by the queries' original index -- any index < n1 is a valid seed (a real point bounds the
// search; the result never depends on it), good ones make the first step as cheap as the later ones.
static int lsnIcpRun_impl(LsnIcp *w, const float *d_verts1, int n1, float *d_verts2, int n2, float *d_R, float *d_t, int maxIter,
                         int nn_mode, void *stream, const int *d_seeds = nullptr)
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

    w->n_events = 0;
    mark(w, 0, s);
    // the target is fixed: one grid for all iterations; the source is sorted once (rigid motion keeps neighbours together)
    if (nn_mode != 0 && build_grid(w, w->tgt, d_verts1, n1, true, s)) return -1;
    if (build_grid(w, w->src, d_verts2, n2, false, s)) return -1;

    const int nb = capped_blocks(n2);
    unsigned long long *keys = w->keys.as<unsigned long long>();
    IcpState *st = w->state.as<IcpState>();
    // one launch instead of three fills (each a kernel of its own, ~4.5 us): the match keys, the NN step's list counters, and
    // v_valid -- the first iteration's SVD starts cold
    const bool seeded0 = d_seeds && w->seed_nn && nn_mode != 0;
    hipLaunchKernelGGL(run_init_kernel, dim3(blocks_for(std::max(std::max(n1, n2), 2 * kBankInts))), dim3(kThreads), 0, s, keys, n1, w->counters.as<int>(),
                       2 * kBankInts, st, seeded0 ? d_seeds : (const int *)nullptr, (const float4 *)w->src.sorted.as<float4>(), n2, w->idx_sorted.as<int>());
    for (int iter = 0; iter < maxIter; iter++) {
        // from the second iteration on idx[] still holds every query's previous neighbour: the search is seeded with it, and
        // its first launch also carries out the previous iteration's motion and clears the match keys
        const bool fused = (iter > 0 || seeded0) && w->seed_nn && nn_mode != 0;
        if (iter > 0 && !fused)
            hipLaunchKernelGGL(apply_kernel, dim3(blocks_for(n2 > n1 ? n2 : n1)), dim3(kThreads), 0, s, w->src.sorted.as<float4>(), d_verts2, n2,
                               (const IcpState *)st, keys, n1);
        if (iter == 0) mark(w, 0, s);
        if (run_nn(w, d_verts1, n1, d_verts2, n2, w->idx.as<int>(), w->dist.as<float>(), keys, nn_mode, s, fused, st, iter & 1)) return -1;
        mark(w, 1, s);
        hipLaunchKernelGGL(stats_kernel, dim3(nb), dim3(kThreads), 0, s, w->idx.as<int>(), w->dist.as<float>(), keys, n2,
                           nn_mode != 0 ? (const GridParams *)w->tgt.gp.as<GridParams>() : (const GridParams *)nullptr, w->part1.as<double>());
        hipLaunchKernelGGL(accum_kernel, dim3(nb), dim3(kThreads), 0, s, d_verts1, (const float *)d_verts2, w->idx.as<int>(),
                           w->dist.as<float>(), keys, n2, w->part1.as<double>(), nb, w->part3.as<double>(), st);
        hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(kThreads), 0, s, w->part3.as<double>(), nb, d_R, d_t, st,
                           iter < kTraceCap ? w->trace.as<float>() : (float *)nullptr, iter);
        mark(w, 2, s);
    }
    hipLaunchKernelGGL(apply_kernel, dim3(blocks_for(n2)), dim3(kThreads), 0, s, w->src.sorted.as<float4>(), d_verts2, n2, (const IcpState *)st,
                       keys, 0);
    mark(w, 3, s);
    LSN_HIP(hipGetLastError());
    return 0;
}

extern "C" int lsnIcpRun(LsnIcp *w, const float *d_verts1, int n1, float *d_verts2, int n2, float *d_R, float *d_t, int maxIter,
                         int nn_mode, void *stream)
{
    return lsn::guarded<int>("lsnIcpRun", static_cast<int>(-1), [&]() { return lsnIcpRun_impl(w, d_verts1, n1, d_verts2, n2, d_R, d_t, maxIter, nn_mode, stream); });
}

// How many queries of the last voxel-grid NN step the near path settled (diagnostic: synchronises `stream`, reads the groups' masks back).
static int lsnIcpNearResolved_impl(LsnIcp *w, void *stream)
{
    lsn::clear_error();
    if (!w) return -1;
    std::lock_guard<std::mutex> g(w->mu);
    LSN_HIP(hipSetDevice(w->device));
    LSN_HIP(hipStreamSynchronize(lsn::as_stream(stream)));
    std::vector<GroupInfo> gi((size_t)w->last_groups);
    if (w->last_groups > 0) LSN_HIP(hipMemcpy(gi.data(), w->groups.p, sizeof(GroupInfo) * gi.size(), hipMemcpyDeviceToHost));
    long long n = 0;
    for (const GroupInfo &x : gi) n += __builtin_popcountll(x.resolved);
    return (int)n;
}

extern "C" int lsnIcpNearResolved(LsnIcp *w, void *stream)
{
    return lsn::guarded<int>("lsnIcpNearResolved", static_cast<int>(-1), [&]() { return lsnIcpNearResolved_impl(w, stream); });
}

#ifdef LSN_CULL_STAMPS
extern "C" int lsnDevCullStamps(long long *out, int n_blocks)
{
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cull_stamps), sizeof(long long) * 8 * (size_t)(n_blocks < 8192 ? n_blocks : 8192)) == hipSuccess ? 0 : -1;
}
#endif

static int lsnIcpTrace_impl(LsnIcp *w, float *out, int max_iters, void *stream)
{
    lsn::clear_error();
    if (!w || !out) return -1;
    LSN_HIP(hipSetDevice(w->device));
    LSN_HIP(hipStreamSynchronize(lsn::as_stream(stream)));
    int n = w->trace_iters < max_iters ? w->trace_iters : max_iters;
    if (n > 0) LSN_HIP(hipMemcpy(out, w->trace.p, sizeof(float) * 16 * (size_t)n, hipMemcpyDeviceToHost));
    return n;
}

extern "C" int lsnIcpTrace(LsnIcp *w, float *out, int max_iters, void *stream)
{
    return lsn::guarded<int>("lsnIcpTrace", static_cast<int>(-1), [&]() { return lsnIcpTrace_impl(w, out, max_iters, stream); });
}

// refineWorker_DoWork (LiveScanServer/MainWindowForm.cs:330-410) with every cloud resident in HBM for the whole
// Gauss-Seidel loop: the reference re-uploads "all other sensors" and the sensor's own cloud for each of the
// n_sensors x n_refine_iters ICP calls; here each cloud goes up once and comes back once.  The pose composition at the
// end repeats the C# loops literally, including their in-place update of worldTransforms[i].R while later rows still
// read it (:398-406).
struct RefineState {   // what a refine pass keeps on the device between calls
    std::mutex mu;
    int device = -1;
    LsnIcp *ws = nullptr;
    hipStream_t s = nullptr;
    lsn::DevBuf d_all, d_others, d_Rt, d_seeds;
    void drop()
    {
        if (device >= 0) (void)hipSetDevice(device);
        if (ws) lsnIcpDestroy(ws);
        ws = nullptr;
        if (s) (void)hipStreamDestroy(s);
        s = nullptr;
        d_all.release(); d_others.release(); d_Rt.release(); d_seeds.release();
    }
    ~RefineState() { drop(); }
};

static RefineState &refine_state()
{
    static RefineState *st = new RefineState();   // never destroyed: no HIP calls from static destructors at process exit
    return *st;
}

static int lsnRefine_impl(int device, int n_sensors, float *const *clouds, const int *counts, int n_refine_iters, int n_icp_iters,
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
        // the pass's device state (workspace, cloud buffers, stream) is kept between calls: allocating it was 2-3 ms of a 20 ms pass.
        // A second pass running at the same time gets a state of its own.
        RefineState *rs = &refine_state();
        std::unique_lock<std::mutex> hold(rs->mu, std::try_to_lock);
        RefineState own;
        if (!hold.owns_lock()) rs = &own;
        const int need1 = (int)(total - min_n), need2 = max_n;
        if (rs->ws && (rs->device != device || rs->ws->max_n1 < need1 || rs->ws->max_n2 < need2)) rs->drop();
        if (!rs->ws) {
            rs->device = device;
            rs->ws = lsnIcpCreate(device, need1, need2);
        }
        int rc = rs->ws ? 0 : -1;
        if (!rc && !rs->s) rc = hipStreamCreateWithFlags(&rs->s, hipStreamNonBlocking) != hipSuccess;
        hipStream_t s = rs->s;
        LsnIcp *ws = rs->ws;
        lsn::DevBuf &d_all = rs->d_all, &d_others = rs->d_others, &d_Rt = rs->d_Rt, &d_seeds = rs->d_seeds;
        if (!rc) rc = d_all.reserve(sizeof(float) * 3 * (size_t)total) || d_others.reserve(sizeof(float) * 3 * (size_t)(total - min_n)) ||
                      d_Rt.reserve(sizeof(float) * Rt.size()) || d_seeds.reserve(sizeof(int) * (size_t)total);
        static const bool carry_seeds = !(getenv("LSN_REFINE_SEEDS") && atoi(getenv("LSN_REFINE_SEEDS")) == 0);   // A/B
        std::vector<long long> off(n_sensors + 1, 0);
        for (int i = 0; i < n_sensors; i++) off[i + 1] = off[i] + counts[i];
        for (int i = 0; i < n_sensors && !rc; i++)
            rc = hipMemcpyAsync(d_all.as<float>() + 3 * off[i], clouds[i], sizeof(float) * 3 * (size_t)counts[i], hipMemcpyHostToDevice, s) != hipSuccess;
        if (!rc) rc = hipMemcpyAsync(d_Rt.p, Rt.data(), sizeof(float) * Rt.size(), hipMemcpyHostToDevice, s) != hipSuccess;
        for (int it = 0; it < n_refine_iters && !rc; it++) {                  // :347
            for (int i = 0; i < n_sensors && !rc; i++) {                      // :349
                // :352-357 all other sensors' current clouds, in sensor order = everything before sensor i's block and everything behind it: two copies
                const long long pos = total - counts[i];
                if (off[i] > 0)
                    rc = hipMemcpyAsync(d_others.as<float>(), d_all.as<float>(), sizeof(float) * 3 * (size_t)off[i], hipMemcpyDeviceToDevice, s) != hipSuccess;
                if (!rc && off[i + 1] < total)
                    rc = hipMemcpyAsync(d_others.as<float>() + 3 * off[i], d_all.as<float>() + 3 * off[i + 1], sizeof(float) * 3 * (size_t)(total - off[i + 1]),
                                        hipMemcpyDeviceToDevice, s) != hipSuccess;
                // From the second pass on the first NN step of a call is seeded with the neighbours the sensor's call of the previous pass ended
                // with ("all other sensors" is the same concatenation in every pass, so the indices still name real points; the others have moved a
                // little, which only makes the seeds a little less tight): ~65 us instead of ~150 for that step, same result.
                if (!rc)
                    rc = lsn::guarded<int>("lsnRefine", -1, [&]() {
                        return lsnIcpRun_impl(ws, d_others.as<float>(), (int)pos, d_all.as<float>() + 3 * off[i], counts[i], d_Rt.as<float>() + 12 * i,
                                              d_Rt.as<float>() + 12 * i + 9, n_icp_iters, 1, s,
                                              carry_seeds && it > 0 ? d_seeds.as<int>() + off[i] : (const int *)nullptr);     // :370
                    });
                if (!rc && carry_seeds && it + 1 < n_refine_iters)
                    rc = hipMemcpyAsync(d_seeds.as<int>() + off[i], ws->idx.p, sizeof(int) * (size_t)counts[i], hipMemcpyDeviceToDevice, s) != hipSuccess;
            }
        }
        // results land in scratch first: the caller's arrays are only touched when everything worked
        std::vector<float> back((size_t)total * 3);
        if (!rc) rc = hipMemcpyAsync(back.data(), d_all.p, sizeof(float) * 3 * (size_t)total, hipMemcpyDeviceToHost, s) != hipSuccess;
        if (!rc) rc = hipMemcpyAsync(Rt.data(), d_Rt.p, sizeof(float) * Rt.size(), hipMemcpyDeviceToHost, s) != hipSuccess;
        if (!rc && s) rc = hipStreamSynchronize(s) != hipSuccess;
        if (rc) {
            if (!lsn::has_error()) lsn::set_error("lsnRefine: %s", hipGetErrorString(hipGetLastError()));
            if (s) (void)hipStreamSynchronize(s);
            rs->drop();   // nothing of a failed pass is kept
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

extern "C" int lsnRefine(int device, int n_sensors, float *const *clouds, const int *counts, int n_refine_iters, int n_icp_iters,
                         float *world_R, float *world_t, float *Rs_out, float *Ts_out)
{
    return lsn::guarded<int>("lsnRefine", static_cast<int>(-1), [&]() { return lsnRefine_impl(device, n_sensors, clouds, counts, n_refine_iters, n_icp_iters, world_R, world_t, Rs_out, Ts_out); });
}
