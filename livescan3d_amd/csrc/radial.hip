// radial.hip -- depthMapAndColorSetRadialCorrection: forward warp + raster-order hole closing, lsnFusionRadialCorrect.
// Shares the plan of fusion.hip (fusion_shared.hpp).
#include "fusion_shared.hpp"

namespace {

// ---- radial correction (the step before the fusion path on every tick) ---------------------------------------------
//
// Replaces depthMapAndColorRadialCorrection (src/NativeUtils/depthprocessing.cpp:191-261) and its export (:1794-1815):
//   1. forward warp of every valid pixel to (x_corr, y_corr); the reference's raster-order loop lets the LAST source
//      pixel win a collision -> atomicMax of the source index per destination, then a gather;
//   2. hole closing, which the reference does IN PLACE in raster order: a pixel filled earlier in the pass is seen by
//      its right / lower neighbours.  Those dependencies (left, up-left, up, up-right) are honoured exactly by a skewed
//      wavefront: one thread per row, row y runs two columns behind row y-1, one workgroup barrier per step.
// All arithmetic in the reference's order (contraction off); (int) follows the x86-64 cvttss2si the reference is built
// with: NaN / out-of-range -> INT_MIN, which then fails the >= 0 test.

struct RadialParams { float cx, cy, fx, fy, r2, r4, r6, pad; };

__device__ __forceinline__ int f2i_x86(float v)
{
    return (v > -2147483904.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000;
}

__global__ __launch_bounds__(kThreads) void radial_warp_kernel(const FrameDesc *frames, const TileDesc *tiles, const RadialParams *rp,
                                                               const unsigned short *depth, unsigned int *winner, int tiles_per_tick,
                                                               long long tick_pix_stride)
{
    const int tick = blockIdx.x / tiles_per_tick;
    const int tile = blockIdx.x - tick * tiles_per_tick;
    const TileDesc td = tiles[tile];
    const FrameDesc fd = frames[td.frame];
    const RadialParams P = rp[td.frame];
    const unsigned short *dep = depth + tick * tick_pix_stride + fd.depth_off;
    unsigned int *win = winner + tick * tick_pix_stride + fd.depth_off;
    const int p0 = (tile - fd.tile_start) * kTile;
    for (int i = threadIdx.x; i < kTile; i += kThreads) {  // consecutive lanes -> consecutive pixels
        const int p = p0 + i;
        if (p >= fd.npix) break;
        if (dep[p] == 0) continue;                                             // :202-203
        const int y = p / fd.w, x = p - y * fd.w;
        const float u = ((float)x - P.cx) / P.fx;                              // :204
        const float v = ((float)y - P.cy) / P.fy;                              // :205
        const float r = u * u + v * v;                                         // :206
        const float d = 1 - P.r2 * r - P.r4 * r * r - P.r6 * r * r * r;        // :207
        const int x_corr = f2i_x86(u * d * P.fx + P.cx);                       // :209
        const int y_corr = f2i_x86(v * d * P.fy + P.cy);                       // :210
        if (x_corr >= 0 && y_corr >= 0 && x_corr < fd.w && y_corr < fd.h)      // :212
            atomicMax(&win[x_corr + (long long)y_corr * fd.w], (unsigned int)p + 1u);  // later source pixel wins (:214-215)
    }
}

__global__ __launch_bounds__(kThreads) void radial_gather_kernel(const FrameDesc *frames, const TileDesc *tiles, const unsigned short *depth,
                                                                 const unsigned char *rgb, const unsigned int *winner,
                                                                 unsigned short *map_copy, unsigned char *colors_copy, int tiles_per_tick,
                                                                 long long tick_pix_stride)
{
    const int tick = blockIdx.x / tiles_per_tick;
    const int tile = blockIdx.x - tick * tiles_per_tick;
    const TileDesc td = tiles[tile];
    const FrameDesc fd = frames[td.frame];
    const long long fb = tick * tick_pix_stride + fd.depth_off;
    const int p0 = (tile - fd.tile_start) * kTile;
    for (int i = threadIdx.x; i < kTile; i += kThreads) {
        const int p = p0 + i;
        if (p >= fd.npix) break;
        const unsigned int wsrc = winner[fb + p];
        unsigned short d = 0;
        unsigned char c0 = 0, c1 = 0, c2 = 0;
        if (wsrc) {
            const long long s = fb + (long long)(wsrc - 1u);
            d = depth[s];
            c0 = rgb[3 * s]; c1 = rgb[3 * s + 1]; c2 = rgb[3 * s + 2];
        }
        map_copy[fb + p] = d;
        colors_copy[3 * (fb + p)] = c0;
        colors_copy[3 * (fb + p) + 1] = c1;
        colors_copy[3 * (fb + p) + 2] = c2;
    }
}

// The warp target of a pixel depends on the intrinsics only, not on the depth values: per calibration, every destination
// pixel gets the (at most four) source pixels that map onto it, highest index first -- the reference's raster-order loop
// lets the LAST valid source win (:200-218).  A tick then needs no atomics, no winner array and no memset: the corrected
// pixel is the first candidate whose depth is not zero.  Destinations with more than four sources (a pathologically
// contracting calibration) raise the overflow flag and the batch takes the atomicMax path above instead.
__global__ __launch_bounds__(kThreads) void radial_cand_fill_kernel(const FrameDesc *frames, const TileDesc *tiles, const RadialParams *rp,
                                                                    unsigned int *count, unsigned int *cand, int *overflow)
{
    const int tile = blockIdx.x;
    const TileDesc td = tiles[tile];
    const FrameDesc fd = frames[td.frame];
    const RadialParams P = rp[td.frame];
    const int p0 = (tile - fd.tile_start) * kTile;
    for (int i = threadIdx.x; i < kTile; i += kThreads) {
        const int p = p0 + i;
        if (p >= fd.npix) break;
        const int y = p / fd.w, x = p - y * fd.w;
        const float u = ((float)x - P.cx) / P.fx;                              // :204
        const float v = ((float)y - P.cy) / P.fy;                              // :205
        const float r = u * u + v * v;                                         // :206
        const float d = 1 - P.r2 * r - P.r4 * r * r - P.r6 * r * r * r;        // :207
        const int x_corr = f2i_x86(u * d * P.fx + P.cx);                       // :209
        const int y_corr = f2i_x86(v * d * P.fy + P.cy);                       // :210
        if (x_corr >= 0 && y_corr >= 0 && x_corr < fd.w && y_corr < fd.h) {    // :212
            const long long dst = fd.depth_off + x_corr + (long long)y_corr * fd.w;
            const unsigned int slot = atomicAdd(&count[dst], 1u);
            if (slot < 4) cand[4 * dst + slot] = (unsigned int)p + 1u;
            else atomicOr(overflow, 1);
        }
    }
}

__global__ __launch_bounds__(kThreads) void radial_cand_sort_kernel(uint4 *cand, long long n)
{
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    uint4 c = cand[i];
    // descending, empty slots (0) last: a 4-element sorting network
    auto cswap = [](unsigned int &a, unsigned int &b) { const unsigned int hi = max(a, b), lo = min(a, b); a = hi; b = lo; };
    cswap(c.x, c.y); cswap(c.z, c.w); cswap(c.x, c.z); cswap(c.y, c.w); cswap(c.y, c.z);
    cand[i] = c;
}

// Aligned copy of n bytes that sit in LDS at lds[lead ..), lead = (address of dst) mod 16, to dst: whole 16-byte chunks as one store
// each, the ragged ends element by element (the twin of store_run in exchange.hip).
template <int ELEM, typename T>
__device__ __forceinline__ void store_tile_run(T *dst, const T *lds, int lead, int n)
{
    static_assert(sizeof(T) == ELEM, "element size");
    const int end = lead + n;
    const int c0 = lead ? 1 : 0, c1 = end >> 4;
    uint4 *g16 = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(dst) - lead);
    const uint4 *l16 = reinterpret_cast<const uint4 *>(lds);
    for (int j = c0 + (int)threadIdx.x; j < c1; j += kThreads) g16[j] = l16[j];
    const int head = lead ? min(n, 16 - lead) : 0;
    const int tail0 = max(head, 16 * c1 - lead);
    const int t = (int)threadIdx.x * ELEM;
    if (t < head) dst[threadIdx.x] = lds[lead / ELEM + threadIdx.x];
    if (tail0 + t < n) dst[tail0 / ELEM + threadIdx.x] = lds[(lead + tail0) / ELEM + threadIdx.x];
}

__global__ __launch_bounds__(kThreads) void radial_gather_cand_kernel(const FrameDesc *__restrict__ frames, const TileDesc *__restrict__ tiles,
                                                                      const unsigned short *__restrict__ depth, const unsigned char *__restrict__ rgb,
                                                                      const uint4 *__restrict__ cand, unsigned short *__restrict__ map_copy,
                                                                      unsigned char *__restrict__ colors_copy, int tiles_per_tick, long long tick_pix_stride)
{
    const int tick = blockIdx.x / tiles_per_tick;
    const int tile = blockIdx.x - tick * tiles_per_tick;
    const TileDesc td = tiles[tile];
    const FrameDesc fd = frames[td.frame];
    const long long fb = tick * tick_pix_stride + fd.depth_off;
    const int p0 = (tile - fd.tile_start) * kTile;
    // the tile's results are staged in LDS and leave as 16-byte stores (four narrow stores per pixel before)
    __shared__ alignas(16) unsigned short s_d[kTile + 8];
    __shared__ alignas(16) unsigned char s_c[3 * kTile + 16];
    unsigned short *gd = map_copy + fb + p0;
    unsigned char *gc = colors_copy + 3 * (fb + p0);
    const int lead_d = (int)(reinterpret_cast<uintptr_t>(gd) & 15), lead_c = (int)(reinterpret_cast<uintptr_t>(gc) & 15);
    // A pixel is a chain of three dependent loads (candidates -> their depths -> the winner's colour); the kernel is bound by
    // that latency (PMC: 87 % of the wave-cycles waiting), so a thread keeps FOUR pixels in flight: all candidate loads, then
    // all depth loads, then all colour loads (the restrict qualifiers let the compiler keep them that way).
    constexpr int kFly = 4;
    for (int i0 = threadIdx.x; i0 < kTile; i0 += kFly * kThreads) {
        uint4 c[kFly];
        bool in[kFly];
#pragma unroll
        for (int k = 0; k < kFly; k++) {
            const int p = p0 + i0 + k * kThreads;
            in[k] = i0 + k * kThreads < kTile && p < fd.npix;
            c[k] = in[k] ? cand[fd.depth_off + p] : make_uint4(0, 0, 0, 0);
        }
        unsigned short d0[kFly], d1[kFly], d2[kFly], d3[kFly];
#pragma unroll
        for (int k = 0; k < kFly; k++) {
            // all four candidate depths are fetched at once (independent loads); the first non-zero one wins
            d0[k] = c[k].x ? depth[fb + (c[k].x - 1u)] : 0;
            d1[k] = c[k].y ? depth[fb + (c[k].y - 1u)] : 0;
            d2[k] = c[k].z ? depth[fb + (c[k].z - 1u)] : 0;
            d3[k] = c[k].w ? depth[fb + (c[k].w - 1u)] : 0;
        }
        unsigned int src[kFly];
        unsigned short d[kFly];
        unsigned char r0[kFly], r1[kFly], r2[kFly];
#pragma unroll
        for (int k = 0; k < kFly; k++) {
            src[k] = 0;
            d[k] = 0;
            if (d0[k]) { src[k] = c[k].x; d[k] = d0[k]; }
            else if (d1[k]) { src[k] = c[k].y; d[k] = d1[k]; }
            else if (d2[k]) { src[k] = c[k].z; d[k] = d2[k]; }
            else if (d3[k]) { src[k] = c[k].w; d[k] = d3[k]; }
            r0[k] = r1[k] = r2[k] = 0;
            if (src[k]) {
                const long long sidx = fb + (long long)(src[k] - 1u);
                r0[k] = rgb[3 * sidx]; r1[k] = rgb[3 * sidx + 1]; r2[k] = rgb[3 * sidx + 2];
            }
        }
#pragma unroll
        for (int k = 0; k < kFly; k++) {
            if (!in[k]) continue;
            const int i = i0 + k * kThreads;
            s_d[lead_d / 2 + i] = d[k];
            unsigned char *c3 = s_c + lead_c + 3 * i;
            c3[0] = r0[k];
            c3[1] = r1[k];
            c3[2] = r2[k];
        }
    }
    __syncthreads();
    const int n_px = min(kTile, fd.npix - p0);
    store_tile_run<2>(gd, s_d, lead_d, 2 * n_px);
    store_tile_run<1>(gc, s_c, lead_c, 3 * n_px);
}

// One workgroup per sensor-frame, one thread per row (bands of blockDim rows when h is larger).  At step t the thread of
// row y handles column x = 1 + t - 2 (y - band0): the pixels it reads from row y-1 (x-1, x, x+1) were finished at least
// one barrier ago, its own left neighbour one step ago, everything to the right and below is still original -- exactly
// the state the reference's raster-order in-place loop sees (:223-256).
// Everything a step touches lives in LDS rings of 32 columns per row (4 chunks of 8; u16 depth and packed RGB): a row's
// thread streams its row through the rings two chunks ahead of where it works (the global loads are issued 8 steps
// before their data is needed) and overwrites a slot when it fills a hole, so the row below reads finals, the row above
// reads originals, no step waits for global memory, and the step barrier only has to order LDS traffic.  Filled pixels
// are also stored to the global maps, fire-and-forget.
constexpr int kRing = 32;

struct RingChunk { unsigned int d[4]; unsigned int c[8]; };  // 8 pixels: depth u16 x 8, colour 0x00BBGGRR x 8

// Loads chunk `chunk` (columns 8 chunk .. 8 chunk + 7) of `row`; anything outside the frame reads as 0.
__device__ __forceinline__ void ring_load_chunk(const unsigned short *map, const unsigned char *col, int w, int h, int row, int chunk,
                                                RingChunk &reg)
{
    const bool row_ok = row >= 0 && row < h;
    if ((w & 7) == 0) {
        // aligned rows: one 16-B depth load and 24 B of colour (three 8-B loads)
        uint4 dv = make_uint4(0, 0, 0, 0);
        uint2 c0 = make_uint2(0, 0), c1 = c0, c2 = c0;
        if (row_ok && chunk >= 0 && chunk * 8 < w) {
            const long long p = (long long)row * w + chunk * 8;
            dv = *reinterpret_cast<const uint4 *>(map + p);
            const uint2 *cp = reinterpret_cast<const uint2 *>(col + 3 * p);
            c0 = cp[0]; c1 = cp[1]; c2 = cp[2];
        }
        reg.d[0] = dv.x; reg.d[1] = dv.y; reg.d[2] = dv.z; reg.d[3] = dv.w;
        const unsigned int cw[6] = {c0.x, c0.y, c1.x, c1.y, c2.x, c2.y};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int b = 3 * j;
            const unsigned int lo = cw[b >> 2], hi = cw[(b >> 2) + 1 < 6 ? (b >> 2) + 1 : 5];
            reg.c[j] = __funnelshift_r(lo, hi, (b & 3) * 8) & 0x00FFFFFFu;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int c0 = chunk * 8 + j;
        unsigned int dv = 0, cv = 0;
        if (row_ok && c0 >= 0 && c0 < w) {
            const long long p = (long long)row * w + c0;
            dv = map[p];
            cv = col[3 * p] | (col[3 * p + 1] << 8) | (col[3 * p + 2] << 16);
        }
        if (j & 1) reg.d[j >> 1] |= dv << 16;
        else reg.d[j >> 1] = dv;
        reg.c[j] = cv;
    }
}

__device__ __forceinline__ void ring_store_chunk(unsigned short *dring_row, unsigned int *cring_row, int chunk, const RingChunk &reg)
{
    const int s0 = (chunk * 8) & (kRing - 1);
    unsigned int *dd = reinterpret_cast<unsigned int *>(dring_row + s0);
#pragma unroll
    for (int j = 0; j < 4; j++) dd[j] = reg.d[j];
#pragma unroll
    for (int j = 0; j < 8; j++) cring_row[s0 + j] = reg.c[j];
}

// A streamed row: its ring slot and the two chunks in flight.
struct RingFeed {
    int ring, row;
    RingChunk p0, p1;
};

// only_if (nullable): close only the frames whose flag is set -- the fall-back of the two-pass closing below -- and copy the closed
// frame to out_d / out_c.
__global__ __launch_bounds__(768) void radial_close_kernel(const FrameDesc *frames, int n_frames, unsigned short *map_copy,
                                                            unsigned char *colors_copy, long long tick_pix_stride, const int *only_if,
                                                            unsigned short *out_d, unsigned char *out_c)
{
    extern __shared__ unsigned int ring_mem[];  // colours: (blockDim.x + 2) x kRing u32, then depths: the same count of u16
    if (only_if && only_if[blockIdx.x] == 0) return;   // uniform over the workgroup
    const int rows = blockDim.x;
    unsigned int *cring = ring_mem;
    unsigned short *dring = reinterpret_cast<unsigned short *>(ring_mem + (rows + 2) * kRing);
    const int tick = blockIdx.x / n_frames;
    const int f = blockIdx.x - tick * n_frames;
    const FrameDesc fd = frames[f];
    const int w = fd.w, h = fd.h;
    unsigned short *map = map_copy + tick * tick_pix_stride + fd.depth_off;
    unsigned char *col = colors_copy + 3 * (tick * tick_pix_stride + fd.depth_off);
    const int r = threadIdx.x;
    for (int band0 = 1; band0 < h - 1; band0 += rows) {
        const int y = band0 + r;
        // Rows this thread streams through the rings (two named feeds, no runtime-indexed arrays -- those would live in
        // scratch memory): A = its own row (ring index r + 1); B = a ghost row: the row above the band for thread 0
        // (index 0), the row below it for the last thread (index rows + 1).
        const bool has_a = y <= h - 1;
        const bool has_b = (r == 0) || (r == rows - 1);
        RingFeed A, B;
        A.ring = r + 1; A.row = y;
        B.ring = r == 0 ? 0 : rows + 1; B.row = r == 0 ? band0 - 1 : y + 1;
        // Every 16 steps ALL lanes publish the two chunks they fetched 16 steps earlier and fetch the next two, so the
        // wave's global loads are consumed a full round after they were issued.  During the round that starts at column
        // x0 the neighbours touch columns x0 - 3 .. x0 + 18 of this row: chunks (x0 - 3) >> 3 .. (x0 + 18) >> 3, at most
        // four -- exactly the ring.
        const int x_start = 1 - 2 * r;
        auto top_chunk = [](int x0) { return (x0 + 18) >> 3; };  // arithmetic shift: floor for negative columns too
        __syncthreads();  // the previous band is done with the rings (and its fills have reached the global maps)
        {
            const int P = top_chunk(x_start);
            if (has_a) {
                for (int c = P - 3; c <= P; c++) {
                    ring_load_chunk(map, col, w, h, A.row, c, A.p0);
                    ring_store_chunk(dring + A.ring * kRing, cring + A.ring * kRing, c, A.p0);
                }
                ring_load_chunk(map, col, w, h, A.row, P + 1, A.p0);
                ring_load_chunk(map, col, w, h, A.row, P + 2, A.p1);
            }
            if (has_b) {
                for (int c = P - 3; c <= P; c++) {
                    ring_load_chunk(map, col, w, h, B.row, c, B.p0);
                    ring_store_chunk(dring + B.ring * kRing, cring + B.ring * kRing, c, B.p0);
                }
                ring_load_chunk(map, col, w, h, B.row, P + 1, B.p0);
                ring_load_chunk(map, col, w, h, B.row, P + 2, B.p1);
            }
        }
        __syncthreads();
        const unsigned short *d_up = dring + r * kRing, *d_below = dring + (r + 2) * kRing;
        unsigned short *d_mine = dring + (r + 1) * kRing;
        const unsigned int *c_up = cring + r * kRing, *c_below = cring + (r + 2) * kRing;
        unsigned int *c_mine = cring + (r + 1) * kRing;
        const int steps = (w - 2) + 2 * (rows - 1);
        for (int t = 0; t < steps; t++) {
            const int x = x_start + t;
            if (t > 0 && (t & 15) == 0) {  // uniform over the workgroup
                const int P = top_chunk(x);
                if (has_a) {
                    ring_store_chunk(dring + A.ring * kRing, cring + A.ring * kRing, P - 1, A.p0);
                    ring_store_chunk(dring + A.ring * kRing, cring + A.ring * kRing, P, A.p1);
                    ring_load_chunk(map, col, w, h, A.row, P + 1, A.p0);
                    ring_load_chunk(map, col, w, h, A.row, P + 2, A.p1);
                }
                if (has_b) {
                    ring_store_chunk(dring + B.ring * kRing, cring + B.ring * kRing, P - 1, B.p0);
                    ring_store_chunk(dring + B.ring * kRing, cring + B.ring * kRing, P, B.p1);
                    ring_load_chunk(map, col, w, h, B.row, P + 1, B.p0);
                    ring_load_chunk(map, col, w, h, B.row, P + 2, B.p1);
                }
                // the new chunks must be in place before any neighbour reads them in this very step
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            }
            if (y < h - 1 && x >= 1 && x < w - 1 && d_mine[x & (kRing - 1)] == 0) {        // :229-234
                const int xm = (x - 1) & (kRing - 1), x0 = x & (kRing - 1), xp = (x + 1) & (kRing - 1);
                const int nb[8] = {d_up[xm], d_up[x0], d_up[xp], d_mine[xm], d_mine[xp], d_below[xm], d_below[x0], d_below[xp]};
                // the acceptance chain of :241-248, branch-free: lane-mask logic and selects instead of eight nested branches
                int n = 0, sum = 0, prev_val = -1;
                unsigned int accepted = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const bool ok = (nb[i] > 0) & ((prev_val == -1) | (abs(nb[i] - prev_val) < 30));  // :241
                    prev_val = ok ? nb[i] : prev_val;
                    n += ok ? 1 : 0;
                    sum += ok ? nb[i] : 0;
                    accepted |= (ok ? 1u : 0u) << i;
                }
                if (n > 4) {                                                                // :250-256
                    const unsigned int nc[8] = {c_up[xm], c_up[x0], c_up[xp], c_mine[xm], c_mine[xp], c_below[xm], c_below[x0], c_below[xp]};
                    int sR = 0, sG = 0, sB = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const unsigned int c = (accepted >> i) & 1u ? nc[i] : 0u;
                        sR += c & 0xFF; sG += (c >> 8) & 0xFF; sB += (c >> 16) & 0xFF;
                    }
                    // n is 5..8 and the sums stay below 2^20: a float reciprocal and one correction step divide exactly
                    const float rn = 1.0f / (float)n;
                    auto div_n = [&](int v) {
                        int q = (int)((float)v * rn);
                        const int r = v - q * n;
                        q += r >= n ? 1 : 0;
                        q -= r < 0 ? 1 : 0;
                        return (unsigned int)q;
                    };
                    const unsigned int fd_ = div_n(sum);
                    const unsigned int fR = div_n(sR), fG = div_n(sG), fB = div_n(sB);
                    d_mine[x0] = (unsigned short)fd_;
                    c_mine[x0] = fR | (fG << 8) | (fB << 16);
                    const long long pos = x + (long long)y * w;
                    map[pos] = (unsigned short)fd_;
                    col[pos * 3] = (unsigned char)fR;
                    col[pos * 3 + 1] = (unsigned char)fG;
                    col[pos * 3 + 2] = (unsigned char)fB;
                }
            }
            // Step barrier on LDS traffic only: a plain __syncthreads() would also wait for the chunk prefetches and
            // the fire-and-forget fills (a global round trip per step).
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
    }
    if (only_if) {
        __syncthreads();   // every fill has reached the global maps
        unsigned short *od = out_d + tick * tick_pix_stride + fd.depth_off;
        unsigned char *oc = out_c + 3 * (tick * tick_pix_stride + fd.depth_off);
        for (int i = threadIdx.x; i < fd.npix; i += blockDim.x) {
            od[i] = map[i];
            oc[3 * i] = col[3 * i];
            oc[3 * i + 1] = col[3 * i + 1];
            oc[3 * i + 2] = col[3 * i + 2];
        }
    }
}


// ---- hole closing in two passes (the default; LSN_RADIAL_CLOSE=wavefront selects the kernel above) ------------------------
//
// Only a hole BEHIND A FILLED PREDECESSOR (up-left, up, up-right, left) depends on the raster order of the reference's in-place
// loop: everything else sees the un-closed map on all eight sides.  Measured on the CPU restatement (512x424): hash-noise
// frames fill no hole at all, scene frames 5 376 of 107 807, the longest chain of fills feeding fills is 262.  So:
//   1. close_first_kernel, streaming: every pixel is copied to the output; every hole is evaluated against the un-closed map
//      (exact unless one of its predecessors gets filled) and every fill lists its hole successors (right, down-left, down,
//      down-right) in the frame's work list;
//   2. close_fix_kernel, one workgroup per frame: re-evaluates the listed holes with their predecessors read from the OUTPUT
//      (current values) and their successors from the un-closed map; a pixel whose value changes lists its own hole successors
//      for the next round.  The dependency graph is acyclic (raster order), every change re-triggers its dependants, so the
//      rounds end -- after at most the longest chain -- in the unique state the sequential loop reaches (:223-256), whatever
//      the order inside a round.
// A frame whose lists overflow is closed by the wavefront kernel instead (flag per frame, no host round trip).
__device__ __forceinline__ void accept_chain(const int (&nb)[8], int &n, int &sum, unsigned int &accepted)
{
    n = 0;
    sum = 0;
    accepted = 0;
    int prev_val = -1;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const bool ok = (nb[i] > 0) & ((prev_val == -1) | (abs(nb[i] - prev_val) < 30));  // :241
        prev_val = ok ? nb[i] : prev_val;
        n += ok ? 1 : 0;
        sum += ok ? nb[i] : 0;
        accepted |= (ok ? 1u : 0u) << i;
    }
}

// v / n for n = 5..8 and v < 2^20: a float reciprocal and one correction step divide exactly
__device__ __forceinline__ unsigned int div_small(int v, int n)
{
    const float rn = 1.0f / (float)n;
    int q = (int)((float)v * rn);
    const int r = v - q * n;
    q += r >= n ? 1 : 0;
    q -= r < 0 ? 1 : 0;
    return (unsigned int)q;
}

// the average colour of the accepted neighbours (:244-246, :252-255) as 0x00BBGGRR; nc[i] = neighbour i's packed colour
__device__ __forceinline__ unsigned int average_colour(const unsigned int (&nc)[8], unsigned int accepted, int n)
{
    int sR = 0, sG = 0, sB = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const unsigned int c = (accepted >> i) & 1u ? nc[i] : 0u;
        sR += c & 0xFF; sG += (c >> 8) & 0xFF; sB += (c >> 16) & 0xFF;
    }
    return div_small(sR, n) | (div_small(sG, n) << 8) | (div_small(sB, n) << 16);
}

__device__ __forceinline__ unsigned int load_rgb24(const unsigned char *c) { return c[0] | (c[1] << 8) | (c[2] << 16); }

struct CloseArgs {
    const FrameDesc *frames;
    const TileDesc *tiles;
    const unsigned short *orig_d;   // the warped, un-closed maps
    const unsigned char *orig_c;
    unsigned short *out_d;          // the caller's maps: the result
    unsigned char *out_c;
    unsigned int *work;             // [n_ticks * n_frames][work_cap] pixel indices inside the frame
    int *work_cnt;                  // [n_ticks * n_frames] x kCntStride: entries listed (may exceed work_cap: overflow)
    int *flags;                     // [n_ticks * n_frames] 1 = close this frame with the wavefront kernel
    int tiles_per_tick, n_frames, work_cap;
    long long tick_pix_stride;
};

// orders a wave's LDS writes before its own later LDS reads (other lanes' data), no barrier: one wave, program order
__device__ __forceinline__ void wave_lds_fence_r()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kDx[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, kDy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};   // the neighbour order of :225

// Appends the wave's entries to the frame's list: one atomicAdd per wave.  `mine` = this lane's count; returns its first slot.
__device__ __forceinline__ int wave_reserve(int *counter, int mine)
{
    const int lane = threadIdx.x & 63;
    const int incl = wave_inclusive_scan(mine, lane);
    const int total = __shfl(incl, 63, 64);
    int base = 0;
    if (total > 0) {
        if (lane == 0) base = atomicAdd(counter, total);
        base = __shfl(base, 0, 64);
    }
    return base + incl - mine;
}

// The same for a whole workgroup (every thread must call it): one atomicAdd per workgroup -- a frame's tiles run side by side
// and all append to the frame's one counter.
__device__ __forceinline__ int block_reserve(int *counter, int mine, int *s /* [8] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_inclusive_scan(mine, lane);
    if (lane == 63) s[wave] = incl;
    __syncthreads();
    const int total = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) s[4] = total > 0 ? atomicAdd(counter, total) : 0;
    __syncthreads();
    int off = s[4];
    for (int i = 0; i < wave; i++) off += s[i];
    return off + incl - mine;
}

constexpr int kCntStride = 32;   // ints between two frames' counters: one 128-byte line each (atomics on one line serialise in its L2 channel)

template <bool VEC>
__global__ __launch_bounds__(kThreads, 8) void close_first_kernel(const CloseArgs a)
{
    __shared__ int s_res[16];
    const int tick = blockIdx.x / a.tiles_per_tick;
    const int tile = blockIdx.x - tick * a.tiles_per_tick;
    const TileDesc td = a.tiles[tile];
    const FrameDesc fd = a.frames[td.frame];
    const int w = fd.w, h = fd.h;
    const long long fb = tick * a.tick_pix_stride + fd.depth_off;
    const int tf = tick * a.n_frames + td.frame;
    unsigned int *work = a.work + (long long)tf * a.work_cap;
    const int p0 = (tile - fd.tile_start) * kTile + (int)threadIdx.x * kPxPerLane;
    const bool in_frame = p0 < fd.npix;
    __shared__ unsigned int s_fill[VEC ? kThreads / 64 : 1][VEC ? 64 * kPxPerLane : 1];   // a wave's fills: pixel | accepted << 24, then their colours
    unsigned int own_c[6] = {0, 0, 0, 0, 0, 0};   // the lane's 24-byte colour group (VEC)
    unsigned int push[kPxPerLane];   // per pixel: bit s = successor s (right, down-left, down, down-right) goes on the list
    unsigned int fill[kPxPerLane];   // per pixel: the accepted-neighbour mask of a fill (0: not filled; a fill accepts >= 5)
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) push[k] = fill[k] = 0;
    int px[kPxPerLane], py[kPxPerLane];
    if (VEC) {
        // w % 8 == 0: the lane's 8 pixels share a row; rows y-1 .. y+1, columns x0-1 .. x0+8 live in registers
        const int v = td.x0 + (int)threadIdx.x * kPxPerLane;
        int q = (int)((float)v * fd.inv_w);
        int x0 = v - q * w;
        if (x0 < 0) { q--; x0 += w; }
        if (x0 >= w) { q++; x0 -= w; }
        const int y = td.y0 + q;
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) { px[k] = x0 + k; py[k] = y; }
        if (in_frame) {
            int D[3][kPxPerLane + 2];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int yy = y - 1 + r;
                const bool row_in = yy >= 0 && yy < h;
                const unsigned short *row = a.orig_d + fb + (long long)(row_in ? yy : y) * w;
                uint4 c = *reinterpret_cast<const uint4 *>(row + x0);
                unsigned int left = x0 > 0 ? row[x0 - 1] : 0u, right = x0 + 8 < w ? row[x0 + 8] : 0u;
                if (!row_in) { c = make_uint4(0, 0, 0, 0); left = right = 0; }
                const unsigned int cw[4] = {c.x, c.y, c.z, c.w};
                D[r][0] = (int)left;
#pragma unroll
                for (int k = 0; k < kPxPerLane; k++) D[r][1 + k] = (int)((cw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu);
                D[r][kPxPerLane + 1] = (int)right;
            }
            const long long pos0 = fb + (long long)y * w + x0;
            const uint2 *cp = reinterpret_cast<const uint2 *>(a.orig_c + 3 * pos0);
            const uint2 c0 = cp[0], c1 = cp[1], c2 = cp[2];
            unsigned int cw[6] = {c0.x, c0.y, c1.x, c1.y, c2.x, c2.y};
            unsigned int od[kPxPerLane];
#pragma unroll
            for (int k = 0; k < kPxPerLane; k++) od[k] = (unsigned int)D[1][k + 1];
#pragma unroll
            for (int k = 0; k < kPxPerLane; k++) {
                const int x = x0 + k;
                if (D[1][k + 1] == 0 && y >= 1 && y < h - 1 && x >= 1 && x < w - 1) {                 // :229-234
                    const int nb[8] = {D[0][k], D[0][k + 1], D[0][k + 2], D[1][k], D[1][k + 2], D[2][k], D[2][k + 1], D[2][k + 2]};
                    int n, sum;
                    unsigned int accepted;
                    accept_chain(nb, n, sum, accepted);
                    if (n > 4) {                                                                       // :250-256
                        od[k] = div_small(sum, n);
                        fill[k] = accepted;   // the colour average follows below, for all the wave's fills at once
                        // its hole successors now depend on the order: onto the list (interior pixels only, :223-224)
                        const bool below = y + 1 < h - 1;
                        push[k] = ((D[1][k + 2] == 0 && x + 1 < w - 1) ? 1u : 0u) | ((D[2][k] == 0 && x - 1 >= 1 && below) ? 2u : 0u) |
                                  ((D[2][k + 1] == 0 && below) ? 4u : 0u) | ((D[2][k + 2] == 0 && x + 1 < w - 1 && below) ? 8u : 0u);
                    }
                }
            }
            *reinterpret_cast<uint4 *>(a.out_d + pos0) = make_uint4(od[0] | (od[1] << 16), od[2] | (od[3] << 16), od[4] | (od[5] << 16), od[6] | (od[7] << 16));
            own_c[0] = cw[0]; own_c[1] = cw[1]; own_c[2] = cw[2]; own_c[3] = cw[3]; own_c[4] = cw[4]; own_c[5] = cw[5];
        }
        // The fills' colours.  A wave's fills (a handful, spread over its lanes and over the 8 pixels of a lane) are handed out one
        // per lane through LDS, so all their neighbour colours are fetched in ONE round trip (evaluated where they arise, the
        // wave would wait for a load eight times over); the averages come back the same way and are patched into the owners'
        // 24-byte colour groups before those are stored.
        {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            unsigned int *mine_l = s_fill[wave];
            int n_fill = 0;
#pragma unroll
            for (int k = 0; k < kPxPerLane; k++) n_fill += fill[k] ? 1 : 0;
            const int incl = wave_inclusive_scan(n_fill, lane);
            const int total = __shfl(incl, 63, 64);
            int slot = incl - n_fill;
#pragma unroll
            for (int k = 0; k < kPxPerLane; k++)
                if (fill[k]) mine_l[slot++] = (unsigned int)(py[k] * w + px[k]) | (fill[k] << 24);
            wave_lds_fence_r();
            for (int i = lane; i < total; i += 64) {
                const unsigned int e = mine_l[i];
                const int p = (int)(e & 0xFFFFFFu);
                const unsigned int accepted = e >> 24;
                unsigned int nc[8];
#pragma unroll
                for (int j = 0; j < 8; j++) nc[j] = (accepted >> j) & 1u ? load_rgb24(a.orig_c + 3 * (fb + p + kDy[j] * w + kDx[j])) : 0u;
                mine_l[i] = average_colour(nc, accepted, __popc(accepted));
            }
            wave_lds_fence_r();
            slot = incl - n_fill;
#pragma unroll
            for (int k = 0; k < kPxPerLane; k++) {
                if (fill[k]) {
                    const unsigned int rgb = mine_l[slot++];
                    // the pixel's three bytes start at byte 3k of the lane's 24-byte group
                    const int b = 3 * k, wi = b >> 2, sh = (b & 3) * 8;
                    const unsigned long long m = 0xFFFFFFull << sh, v = (unsigned long long)rgb << sh;
                    own_c[wi] = (own_c[wi] & ~(unsigned int)m) | (unsigned int)v;
                    if (wi + 1 < 6 && (m >> 32)) own_c[wi + 1] = (own_c[wi + 1] & ~(unsigned int)(m >> 32)) | (unsigned int)(v >> 32);
                }
            }
        }
        if (in_frame) {
            uint2 *op = reinterpret_cast<uint2 *>(a.out_c + 3 * (fb + (long long)py[0] * w + px[0]));
            op[0] = make_uint2(own_c[0], own_c[1]);
            op[1] = make_uint2(own_c[2], own_c[3]);
            op[2] = make_uint2(own_c[4], own_c[5]);
        }
    } else {
        // any width: pixel by pixel
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) {
            const int p = p0 + k;
            px[k] = py[k] = 0;
            if (p >= fd.npix) continue;
            const int y = p / w, x = p - y * w;
            px[k] = x; py[k] = y;
            const long long pos = fb + p;
            unsigned int d = a.orig_d[pos], rgb = load_rgb24(a.orig_c + 3 * pos);
            if (d == 0 && y >= 1 && y < h - 1 && x >= 1 && x < w - 1) {
                int nb[8];
#pragma unroll
                for (int i = 0; i < 8; i++) nb[i] = a.orig_d[pos + kDy[i] * w + kDx[i]];
                int n, sum;
                unsigned int accepted;
                accept_chain(nb, n, sum, accepted);
                if (n > 4) {
                    unsigned int nc[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) nc[i] = (accepted >> i) & 1u ? load_rgb24(a.orig_c + 3 * (pos + kDy[i] * w + kDx[i])) : 0u;
                    d = div_small(sum, n);
                    rgb = average_colour(nc, accepted, n);
                    const bool below = y + 1 < h - 1;
                    push[k] = ((nb[4] == 0 && x + 1 < w - 1) ? 1u : 0u) | ((nb[5] == 0 && x - 1 >= 1 && below) ? 2u : 0u) |
                              ((nb[6] == 0 && below) ? 4u : 0u) | ((nb[7] == 0 && x + 1 < w - 1 && below) ? 8u : 0u);
                }
            }
            a.out_d[pos] = (unsigned short)d;
            a.out_c[3 * pos] = (unsigned char)rgb;
            a.out_c[3 * pos + 1] = (unsigned char)(rgb >> 8);
            a.out_c[3 * pos + 2] = (unsigned char)(rgb >> 16);
        }
    }
    int mine = 0;
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) mine += __popc(push[k]);
    int slot = block_reserve(a.work_cnt + kCntStride * tf, mine, s_res);
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) {
#pragma unroll
        for (int sidx = 0; sidx < 4; sidx++) {
            if ((push[k] >> sidx) & 1u) {
                const int i = 4 + sidx;   // neighbours 4..7 are the successors
                if (slot < a.work_cap) work[slot] = (unsigned int)((py[k] + kDy[i]) * w + px[k] + kDx[i]);
                slot++;
            }
        }
    }
}

constexpr int kFixList = 8192;   // entries per round list (LDS, two lists: 64 KB -- a few scene frames listed more than 4096 in a round)
constexpr int kFixRounds = 1 << 16;

__global__ __launch_bounds__(kThreads) void close_fix_kernel(const CloseArgs a)
{
    __shared__ unsigned int lists[2][kFixList];
    __shared__ int s_n[2];
    __shared__ int s_overflow;
    const int tf = blockIdx.x;
    const int tick = tf / a.n_frames, f = tf - tick * a.n_frames;
    const FrameDesc fd = a.frames[f];
    const int w = fd.w, h = fd.h;
    const long long fb = tick * a.tick_pix_stride + fd.depth_off;
    const unsigned short *orig_d = a.orig_d + fb;
    const unsigned char *orig_c = a.orig_c + 3 * fb;
    // written and re-read by the waves of this workgroup only, a round apart: they share the CU's L1, and the barrier between two
    // rounds (workgroup-scope release / acquire) orders the accesses -- no cache bypass needed
    unsigned short *out_d = a.out_d + fb;
    unsigned char *out_c = a.out_c + 3 * fb;
    const int n0 = a.work_cnt[kCntStride * tf];
    if (threadIdx.x == 0) {
        s_n[0] = s_n[1] = 0;
        s_overflow = n0 > a.work_cap ? 1 : 0;
        a.flags[tf] = 0;
    }
    __syncthreads();
    const unsigned int *glist = a.work + (long long)tf * a.work_cap;
    int cur = 0;   // lists[cur] is read, lists[1 - cur] is filled; round 0 reads the global list instead
    for (int round = 0; !s_overflow; round++) {
        const int n_items = round == 0 ? n0 : s_n[cur];
        if (n_items == 0) break;
        if (round >= kFixRounds) {   // cannot happen (the chains are finite); never leave a frame half closed
            if (threadIdx.x == 0) s_overflow = 1;
            __syncthreads();
            break;
        }
        for (int i0 = 0; i0 < n_items; i0 += kThreads) {
            const int i = i0 + (int)threadIdx.x;
            int mine = 0;
            unsigned int succ = 0;
            int x = 0, y = 0;
            if (i < n_items) {
                const int p = (int)(round == 0 ? glist[i] : lists[cur][i]);
                y = p / w;
                x = p - y * w;
                // predecessors (neighbours 0..3) as they are now, successors (4..7) as the sequential loop would still see them
                int nb[8];
#pragma unroll
                for (int k = 0; k < 4; k++) nb[k] = out_d[p + kDy[k] * w + kDx[k]];
#pragma unroll
                for (int k = 4; k < 8; k++) nb[k] = orig_d[p + kDy[k] * w + kDx[k]];
                // every colour the verdict may need is fetched together with the depths (one memory round trip per round, not two)
                unsigned int nc[8];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const long long q = 3 * (long long)(p + kDy[k] * w + kDx[k]);
                    nc[k] = (unsigned int)out_c[q] | ((unsigned int)out_c[q + 1] << 8) | ((unsigned int)out_c[q + 2] << 16);
                }
#pragma unroll
                for (int k = 4; k < 8; k++) nc[k] = load_rgb24(orig_c + 3 * (long long)(p + kDy[k] * w + kDx[k]));
                const unsigned int od = out_d[p];
                const unsigned int orgb = (unsigned int)out_c[3 * (long long)p] | ((unsigned int)out_c[3 * (long long)p + 1] << 8) |
                                          ((unsigned int)out_c[3 * (long long)p + 2] << 16);
                unsigned int nd = 0, nrgb = load_rgb24(orig_c + 3 * (long long)p);   // an unfilled hole keeps what the warp left (:250)
                int n, sum;
                unsigned int accepted;
                accept_chain(nb, n, sum, accepted);
                if (n > 4) {
                    nd = div_small(sum, n);
                    nrgb = average_colour(nc, accepted, n);
                }
                if (nd != od || nrgb != orgb) {
                    out_d[p] = (unsigned short)nd;
                    out_c[3 * (long long)p] = (unsigned char)nrgb;
                    out_c[3 * (long long)p + 1] = (unsigned char)(nrgb >> 8);
                    out_c[3 * (long long)p + 2] = (unsigned char)(nrgb >> 16);
                    const bool below = y + 1 < h - 1;
                    succ = ((nb[4] == 0 && x + 1 < w - 1) ? 1u : 0u) | ((nb[5] == 0 && x - 1 >= 1 && below) ? 2u : 0u) |
                           ((nb[6] == 0 && below) ? 4u : 0u) | ((nb[7] == 0 && x + 1 < w - 1 && below) ? 8u : 0u);
                    mine = __popc(succ);
                }
            }
            int slot = wave_reserve(&s_n[1 - cur], mine);
#pragma unroll
            for (int sidx = 0; sidx < 4; sidx++) {
                if ((succ >> sidx) & 1u) {
                    const int k = 4 + sidx;
                    if (slot < kFixList) lists[1 - cur][slot] = (unsigned int)((y + kDy[k]) * w + x + kDx[k]);
                    else s_overflow = 1;
                    slot++;
                }
            }
        }
        __threadfence_block();
        __syncthreads();   // this round's writes are in place before the next round reads them
        cur = 1 - cur;
        if (threadIdx.x == 0) s_n[1 - cur] = 0;
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_overflow) a.flags[tf] = 1;
}

}  // namespace

extern "C" int lsnFusionRadialCorrect(LsnFusion *p, const float *intr_params, void *d_depth, void *d_colors, void *stream)
{
    lsn::clear_error();
    if (!p || !intr_params || !d_depth || !d_colors) {
        lsn::set_error("lsnFusionRadialCorrect: null argument");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    hipStream_t s = lsn::as_stream(stream);
    const size_t npix = (size_t)p->cap * p->n_ticks;
    if (p->winner.reserve(4 * npix) || p->map_copy.reserve(2 * npix) || p->colors_copy.reserve(3 * npix) ||
        p->radial.reserve(sizeof(RadialParams) * p->n_maps))
        return -1;
    std::vector<RadialParams> rp(p->n_maps);
    for (int i = 0; i < p->n_maps; i++) {
        const float *ip = intr_params + 7 * i;  // IntrinsicCameraParameters(float*), include/NativeUtils/depthprocessing.h:96-97
        rp[i] = RadialParams{ip[0], ip[1], ip[2], ip[3], ip[4], ip[5], ip[6], 0.0f};
    }
    const int grid = p->tiles_per_tick * p->n_ticks;
    const bool same_intr = p->cand_valid && p->radial_intr.size() == 7 * (size_t)p->n_maps &&
                           memcmp(p->radial_intr.data(), intr_params, sizeof(float) * 7 * p->n_maps) == 0;
    if (!same_intr) {
        LSN_HIP(hipMemcpyAsync(p->radial.p, rp.data(), sizeof(RadialParams) * p->n_maps, hipMemcpyHostToDevice, s));
        LSN_HIP(hipStreamSynchronize(s));  // rp is a local
        // the warp candidates of this calibration (one tick's worth of pixels; `winner` serves as the per-destination counter)
        if (p->cand.reserve(16 * (size_t)p->cap)) return -1;
        LSN_HIP(hipMemsetAsync(p->winner.p, 0, 4 * (size_t)p->cap, s));
        LSN_HIP(hipMemsetAsync(p->cand.p, 0, 16 * (size_t)p->cap, s));
        LSN_HIP(hipMemsetAsync(p->misc.as<char>() + 64, 0, sizeof(int), s));
        int *overflow = reinterpret_cast<int *>(p->misc.as<char>() + 64);
        hipLaunchKernelGGL(radial_cand_fill_kernel, dim3(p->tiles_per_tick), dim3(kThreads), 0, s, p->frames.as<FrameDesc>(),
                           p->tile_frame.as<TileDesc>(), p->radial.as<RadialParams>(), p->winner.as<unsigned int>(), p->cand.as<unsigned int>(), overflow);
        hipLaunchKernelGGL(radial_cand_sort_kernel, dim3((unsigned)((p->cap + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p->cand.as<uint4>(),
                           p->cap);
        int ov = 0;
        LSN_HIP(hipMemcpyAsync(&ov, overflow, sizeof(int), hipMemcpyDeviceToHost, s));
        LSN_HIP(hipStreamSynchronize(s));
        p->cand_overflow = ov != 0;
        p->radial_intr.assign(intr_params, intr_params + 7 * (size_t)p->n_maps);
        p->cand_valid = true;
    }
    const char *force = getenv("LSN_RADIAL_FORCE_ATOMIC");  // tests: take the atomicMax path even when the table did not overflow
    if (!p->cand_overflow && !(force && atoi(force) != 0)) {
        hipLaunchKernelGGL(radial_gather_cand_kernel, dim3(grid), dim3(kThreads), 0, s, p->frames.as<FrameDesc>(), p->tile_frame.as<TileDesc>(),
                           static_cast<const unsigned short *>(d_depth), static_cast<const unsigned char *>(d_colors), (const uint4 *)p->cand.as<uint4>(),
                           p->map_copy.as<unsigned short>(), p->colors_copy.as<unsigned char>(), p->tiles_per_tick, p->cap);
    } else {
        LSN_HIP(hipMemsetAsync(p->winner.p, 0, 4 * npix, s));
        hipLaunchKernelGGL(radial_warp_kernel, dim3(grid), dim3(kThreads), 0, s, p->frames.as<FrameDesc>(), p->tile_frame.as<TileDesc>(),
                           p->radial.as<RadialParams>(), static_cast<const unsigned short *>(d_depth), p->winner.as<unsigned int>(),
                           p->tiles_per_tick, p->cap);
        hipLaunchKernelGGL(radial_gather_kernel, dim3(grid), dim3(kThreads), 0, s, p->frames.as<FrameDesc>(), p->tile_frame.as<TileDesc>(),
                           static_cast<const unsigned short *>(d_depth), static_cast<const unsigned char *>(d_colors),
                           (const unsigned int *)p->winner.as<unsigned int>(), p->map_copy.as<unsigned short>(), p->colors_copy.as<unsigned char>(),
                           p->tiles_per_tick, p->cap);
    }
    int max_h = 1;
    for (int v : p->h) max_h = v > max_h ? v : max_h;
    int rows = max_h - 2 < 64 ? 64 : ((max_h - 2 + 63) / 64) * 64;
    if (rows > 768) rows = 768;  // (rows + 2) x 32 columns x 6 B of LDS rings must fit in 160 KB
    // One band per frame is the shortest chain of steps, but (rows + 2) x 192 B of LDS per workgroup then allows a single
    // frame per CU.  With more frames than CUs, 256-row bands (49.5 KB: three frames per CU) win: 2.75 vs 3.37 ms for
    // 512 frames of 512x424 on MI355X.
    if ((long long)p->n_maps * p->n_ticks > 256 && rows > 256) rows = 256;
    if (const char *env = getenv("LSN_RADIAL_ROWS")) {  // tuning: rows per band (multiple of 64, <= 768)
        const int v = atoi(env);
        if (v >= 64 && v <= 768 && v % 64 == 0) rows = v;
    }
    const size_t ring_bytes = (sizeof(unsigned int) + sizeof(unsigned short)) * kRing * (rows + 2);
    const int n_tf = p->n_maps * p->n_ticks;
    // Two-pass closing: its work lists live in `winner` (free once the gather is done): counters, flags, then work_cap entries per frame
    // (read on every call, like LSN_RADIAL_FORCE_ATOMIC: the tests switch them inside one process)
    const char *close_env = getenv("LSN_RADIAL_CLOSE"), *tiny_env = getenv("LSN_RADIAL_TINY_LISTS");
    const bool wavefront_only = close_env && !strcmp(close_env, "wavefront");
    const bool tiny_lists = tiny_env && atoi(tiny_env) != 0;   // tests: force the fall-back
    long long work_cap = ((long long)npix - (kCntStride + 1ll) * n_tf - 64) / n_tf;
    if (work_cap > 16384) work_cap = 16384;
    if (tiny_lists && work_cap > 8) work_cap = 8;
    long long max_npix = 0;
    for (int i = 0; i < p->n_maps; i++) max_npix = std::max(max_npix, (long long)p->w[i] * p->h[i]);
    // (a wave's fill entries carry the pixel index in 24 bits)
    if (!wavefront_only && work_cap >= 8 && max_npix < (1ll << 24)) {
        CloseArgs ca;
        ca.frames = p->frames.as<FrameDesc>();
        ca.tiles = p->tile_frame.as<TileDesc>();
        ca.orig_d = p->map_copy.as<unsigned short>();
        ca.orig_c = p->colors_copy.as<unsigned char>();
        ca.out_d = static_cast<unsigned short *>(d_depth);
        ca.out_c = static_cast<unsigned char *>(d_colors);
        ca.work_cnt = p->winner.as<int>();
        ca.flags = ca.work_cnt + (size_t)kCntStride * n_tf;
        ca.work = p->winner.as<unsigned int>() + ((size_t)kCntStride + 1) * n_tf + 64;
        ca.tiles_per_tick = p->tiles_per_tick;
        ca.n_frames = p->n_maps;
        ca.work_cap = (int)work_cap;
        ca.tick_pix_stride = p->cap;
        LSN_HIP(hipMemsetAsync(ca.work_cnt, 0, sizeof(int) * kCntStride * (size_t)n_tf, s));
        const bool vec = p->vec_ok && ((uintptr_t)d_depth & 15) == 0 && ((uintptr_t)d_colors & 7) == 0 && (p->cap % 8) == 0;
        if (vec) hipLaunchKernelGGL(close_first_kernel<true>, dim3(grid), dim3(kThreads), 0, s, ca);
        else     hipLaunchKernelGGL(close_first_kernel<false>, dim3(grid), dim3(kThreads), 0, s, ca);
        hipLaunchKernelGGL(close_fix_kernel, dim3((unsigned)n_tf), dim3(kThreads), 0, s, ca);
        // frames whose lists overflowed (flag set by close_fix_kernel): the ordered pass on the un-closed maps, copied out
        hipLaunchKernelGGL(radial_close_kernel, dim3((unsigned)n_tf), dim3(rows), ring_bytes, s, p->frames.as<FrameDesc>(), p->n_maps,
                           p->map_copy.as<unsigned short>(), p->colors_copy.as<unsigned char>(), p->cap, (const int *)ca.flags,
                           static_cast<unsigned short *>(d_depth), static_cast<unsigned char *>(d_colors));
        LSN_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(radial_close_kernel, dim3((unsigned)n_tf), dim3(rows), ring_bytes, s, p->frames.as<FrameDesc>(), p->n_maps,
                       p->map_copy.as<unsigned short>(), p->colors_copy.as<unsigned char>(), p->cap, (const int *)nullptr,
                       (unsigned short *)nullptr, (unsigned char *)nullptr);
    LSN_HIP(hipGetLastError());
    // :259-260 the corrected maps replace the inputs
    LSN_HIP(hipMemcpyAsync(d_depth, p->map_copy.p, 2 * npix, hipMemcpyDeviceToDevice, s));
    LSN_HIP(hipMemcpyAsync(d_colors, p->colors_copy.p, 3 * npix, hipMemcpyDeviceToDevice, s));
    return 0;
}

