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

// ---- the compact warp table ------------------------------------------------------------------------------------------------
//
// The candidate table above costs 16 bytes per destination and tick (the 8 x 512x424 rig: 27.8 MB, streamed from the Infinity
// Cache by every tick -- the round-2 PMC pass showed 2 x 255 MB fetched for 139 MB of frames).  A radial map moves a pixel by
// a few columns / rows, and almost every destination has one or two sources, so the table the ticks read is ONE dword per
// destination: two candidates as (source pixel - destination pixel) + 32768 in 16 bits each (indices inside the frame; 0 = none),
// highest source index first -- a candidate then decodes with one add.  A destination with three or four sources, or one more than
// 32767 pixels away, holds kWide and is looked up in the full table (a handful of pixels per frame for Kinect-like intrinsics).
// 6.9 MB for the 8-sensor rig: it stays in the L2s.
constexpr unsigned int kWide = 0xFFFFFFFFu;

__global__ __launch_bounds__(kThreads) void radial_cand_pack_kernel(const FrameDesc *frames, const TileDesc *tiles, const uint4 *cand,
                                                                    unsigned int *ctab)
{
    const int tile = blockIdx.x;
    const TileDesc td = tiles[tile];
    const FrameDesc fd = frames[td.frame];
    const int p0 = (tile - fd.tile_start) * kTile;
    for (int i = threadIdx.x; i < kTile; i += kThreads) {
        const int p = p0 + i;
        if (p >= fd.npix) break;
        const uint4 c = cand[fd.depth_off + p];
        auto enc = [&](unsigned int src, unsigned int &code) {
            code = 0;
            if (!src) return true;
            const int rel = (int)(src - 1u) - p;
            if (rel < -32767 || rel > 32767) return false;
            code = (unsigned int)(rel + 32768);
            return true;
        };
        unsigned int c0, c1;
        const bool ok0 = enc(c.x, c0), ok1 = enc(c.y, c1);
        ctab[fd.depth_off + p] = (ok0 && ok1 && c.z == 0) ? (c0 | (c1 << 16)) : kWide;
    }
}

typedef unsigned int u32_ua __attribute__((aligned(1)));
typedef unsigned short u16_ua __attribute__((aligned(1)));
typedef unsigned long long u64_ua __attribute__((aligned(1)));

struct WarpSrc {
    const unsigned short *depth;   // the frames as they came in (or, for the closing alone, the warped un-closed maps)
    const unsigned char *rgb;
    const unsigned int *ctab;      // [pixels per tick] compact table
    const uint4 *cand;             // [pixels per tick] full table (kWide entries)
    long long last_px;             // index of the last pixel of the whole batch (its colour is not read as a dword)
};


// Warped value of K destination pixels p[k] of one frame (fb = first pixel of the frame in the batch, doff = in its tick): a
// chain of dependent loads per pixel -- table, the candidates' depths (both at once; with SPEC the first candidate's colour too),
// the winner's colour -- issued for all K pixels level by level.  Every load is unconditional (a pixel that needs none reads its own
// position and drops the value): no exec-mask branches, so the K loads of a level really are in flight together.  pv: any valid
// pixel of the frame (stands in for p[k] where in[k] is false).
// (buffer addressing: a 128-bit resource in SGPRs + one 32-bit byte offset per lane -- no 64-bit vector arithmetic per load, and an
// offset past the frame reads 0, which is what a candidate that does not exist has to read)
constexpr unsigned int kRsrcWord3 = 0x00027000u;   // gfx9 raw buffer, 32-bit elements
constexpr unsigned int kNowhere = 0xFFFFFFF0u;     // a byte offset outside every frame

template <int K, bool SPEC = false>
__device__ __forceinline__ void gather_batch(const WarpSrc &S, long long fb, long long doff, int npix, int pv, const int (&p)[K], const bool (&in)[K],
                                             unsigned int (&d)[K], unsigned int (&rgb)[K])
{
    (void)pv;
    const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int *>(S.ctab + doff), 0, 4 * npix, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t dep = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(S.depth + fb), 0, 2 * npix, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t col = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(S.rgb + 3 * fb), 0, 3 * npix, kRsrcWord3);
    const unsigned int last = (unsigned int)npix - 1u;
    auto colour = [&](unsigned int q, bool real) {     // 0x00BBGGRR of the frame's pixel q: one unaligned dword (the byte behind the three belongs
        const unsigned int adj = q == last ? 1u : 0u;   // to the next pixel; the frame's last pixel is read one byte early instead); !real reads 0
        const unsigned int v = __builtin_amdgcn_raw_buffer_load_b32(col, real ? 3u * q - adj : kNowhere, 0, 0);
        return (v >> (8u * adj)) & 0x00FFFFFFu;         // (a frame of ONE pixel has no byte in front of it either: one_pixel_colour() below)
    };
    unsigned int e[K];
#pragma unroll
    for (int k = 0; k < K; k++) e[k] = __builtin_amdgcn_raw_buffer_load_b32(tab, in[k] ? 4u * (unsigned int)p[k] : kNowhere, 0, 0);
    unsigned int s0[K], s1[K], d0[K], d1[K], c0[K];
    bool any_wide = false;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const unsigned int a0 = e[k] & 0xFFFFu, a1 = e[k] >> 16;
        const bool wide = e[k] == kWide;
        any_wide |= wide;
        const bool v0 = a0 != 0 && !wide, v1 = a1 != 0 && !wide;
        s0[k] = (unsigned int)p[k] + a0 - 32768u;
        s1[k] = (unsigned int)p[k] + a1 - 32768u;
        d0[k] = __builtin_amdgcn_raw_buffer_load_b16(dep, v0 ? 2u * s0[k] : kNowhere, 0, 0);
        d1[k] = __builtin_amdgcn_raw_buffer_load_b16(dep, v1 ? 2u * s1[k] : kNowhere, 0, 0);
        if (SPEC) c0[k] = colour(s0[k], v0);               // the first candidate's colour rides with the depths
    }
    bool second = false;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const unsigned int src = d0[k] ? s0[k] : s1[k];     // the highest valid source wins (:200-218: the last one in raster order)
        d[k] = d0[k] ? d0[k] : d1[k];
        if (SPEC) {
            rgb[k] = d0[k] ? c0[k] : 0u;
            second |= !d0[k] && d1[k];
        } else {
            rgb[k] = colour(src, d[k] != 0);
        }
    }
    if (SPEC && __any(second)) {                            // a zero-depth first candidate in front of a valid second one
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool need = !d0[k] && d1[k];
            const unsigned int v = colour(s1[k], need);
            if (need) rgb[k] = v;
        }
    }
    if (__any(any_wide)) {                                  // three or four sources, or a far one: the full table (a few pixels per frame)
        const unsigned short *gdep = S.depth + fb;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (e[k] == kWide) {
                const uint4 c = S.cand[doff + p[k]];
                const unsigned int cs[4] = {c.x, c.y, c.z, c.w};
                int sw = -1;
                unsigned int dd = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (sw < 0 && cs[j]) {
                        const unsigned int v = gdep[cs[j] - 1u];
                        if (v) { sw = (int)(cs[j] - 1u); dd = v; }
                    }
                }
                d[k] = dd;
                rgb[k] = colour((unsigned int)(sw >= 0 ? sw : 0), sw >= 0);
            }
        }
    }
}

// gather_batch on a frame of ONE pixel: its colour would be read "one byte before the last pixel", i.e. in front of the frame, and
// comes back 0 (found by tests/test_fuzz_gpu.py).  The depth it returned is right, and the only source a 1 x 1 frame's pixel can have is
// itself: its colour is the frame's three bytes if that depth is not zero.  Called once per such frame, outside the gather loops (a
// test inside colour() cost radial_band_kernel 30 %: 456 -> 590 us per 512 frames).
__device__ __forceinline__ unsigned int one_pixel_colour(const WarpSrc &S, long long fb, unsigned int depth)
{
    const unsigned char *px = S.rgb + 3 * fb;
    return depth ? ((unsigned int)px[0] | ((unsigned int)px[1] << 8) | ((unsigned int)px[2] << 16)) : 0u;
}

// The warp alone, to the un-closed scratch maps (the in-place entry point: the closing then reads those).
__global__ __launch_bounds__(kThreads) void radial_gather_pack_kernel(const FrameDesc *__restrict__ frames, const TileDesc *__restrict__ tiles,
                                                                      const WarpSrc S, unsigned short *__restrict__ map_copy,
                                                                      unsigned char *__restrict__ colors_copy, int tiles_per_tick, long long tick_pix_stride)
{
    const int tick = blockIdx.x / tiles_per_tick;
    const int tile = blockIdx.x - tick * tiles_per_tick;
    const TileDesc td = tiles[tile];
    const FrameDesc fd = frames[td.frame];
    const long long fb = tick * tick_pix_stride + fd.depth_off;
    const int p0 = (tile - fd.tile_start) * kTile;
    // the tile's results are staged in LDS and leave as 16-byte stores
    __shared__ alignas(16) unsigned short s_d[kTile + 8];
    __shared__ alignas(16) unsigned char s_c[3 * kTile + 16];
    unsigned short *gd = map_copy + fb + p0;
    unsigned char *gc = colors_copy + 3 * (fb + p0);
    const int lead_d = (int)(reinterpret_cast<uintptr_t>(gd) & 15), lead_c = (int)(reinterpret_cast<uintptr_t>(gc) & 15);
    constexpr int kFly = 4;
    for (int i0 = threadIdx.x; i0 < kTile; i0 += kFly * kThreads) {
        int p[kFly];
        bool in[kFly];
#pragma unroll
        for (int k = 0; k < kFly; k++) {
            p[k] = p0 + i0 + k * kThreads;
            in[k] = i0 + k * kThreads < kTile && p[k] < fd.npix;
        }
        unsigned int d[kFly], c[kFly];
        gather_batch<kFly>(S, fb, fd.depth_off, fd.npix, p0, p, in, d, c);
#pragma unroll
        for (int k = 0; k < kFly; k++) {
            if (!in[k]) continue;
            const int i = i0 + k * kThreads;
            s_d[lead_d / 2 + i] = (unsigned short)d[k];
            unsigned char *c3 = s_c + lead_c + 3 * i;
            c3[0] = (unsigned char)c[k];
            c3[1] = (unsigned char)(c[k] >> 8);
            c3[2] = (unsigned char)(c[k] >> 16);
        }
    }
    if (fd.npix == 1 && threadIdx.x == 0) {
        const unsigned int c1 = one_pixel_colour(S, fb, s_d[lead_d / 2]);
        unsigned char *c3 = s_c + lead_c;
        c3[0] = (unsigned char)c1;
        c3[1] = (unsigned char)(c1 >> 8);
        c3[2] = (unsigned char)(c1 >> 16);
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

// (LSN_RADIAL_CLOSE=wavefront: the ordered kernel of round 1, kept as an ablation of the band + rounds closing below)
__global__ __launch_bounds__(768) void radial_close_kernel(const FrameDesc *frames, int n_frames, unsigned short *map_copy,
                                                            unsigned char *colors_copy, long long tick_pix_stride)
{
    extern __shared__ unsigned int ring_mem[];  // colours: (blockDim.x + 2) x kRing u32, then depths: the same count of u16
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
}


// ---- warp + hole closing, one band of rows per workgroup (the default) -------------------------------------------------------
//
// Only a hole BEHIND A FILLED PREDECESSOR (up-left, up, up-right, left) depends on the raster order of the reference's in-place
// loop (:223-256): everything else sees the un-closed map on all eight sides.  On the 512x424 scene frames (CPU restatement):
// 108 k holes per frame, 5.2 k of them with the five valid neighbours a fill needs at all, 4.8 k filled; the re-evaluation of the
// holes behind a fill settles in 13-25 rounds of 3.0 k, 2.4 k, 0.9 k, 0.4 k ... pixels.  So:
//   1. radial_band_kernel, one workgroup per band of rows: the warped band and one halo row either side are built IN LDS -- the
//      un-closed map never goes to memory (GATHER), or it is read from the scratch maps a separate warp left (in-place entry point,
//      atomicMax path).  Every interior hole with at least five valid neighbours goes on a list in LDS; the list is evaluated one
//      candidate per lane against the un-closed band (exact unless one of the hole's predecessors gets filled); the fills are patched
//      into the band; the band leaves as 16-byte stores together with one bit per pixel "was a hole before the closing", and every
//      fill's hole successors (right, down-left, down, down-right) go on the frame's work list;
//   2. close_fix_kernel, one workgroup per frame: re-evaluates the listed holes with their predecessors as they are NOW and their
//      successors as the un-closed map had them (the bit says "hole": 0, else the value in place, which a non-hole never changes); a
//      pixel whose value changes lists its own hole successors for the next round.  The dependency graph is acyclic (raster order)
//      and every change re-triggers its dependants, so the rounds end -- after at most the longest chain -- in the unique state the
//      sequential loop reaches, whatever the order inside a round.  A round list that outgrows LDS switches the frame to full sweeps
//      over all its holes until nothing changes: the same fixed point, no list.
// The colour of a hole is zero in the un-closed map (map_copy / colors_copy start zeroed, :193-194, and only valid pixels are warped).
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

constexpr int kDx[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, kDy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};   // the neighbour order of :225

// Appends the wave's entries to a list: one atomicAdd per wave.  `mine` = this lane's count; returns its first slot.
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

constexpr int kCntStride = 32;   // ints between two frames' counters: one 128-byte line each (atomics on one line serialise in its L2 channel)
#ifndef LSN_BAND_THREADS
#define LSN_BAND_THREADS 256
#endif
constexpr int kBandThreads = LSN_BAND_THREADS;   // (A/B knob: tools/ab_band.sh)

struct BandDesc { int frame, y0; };

// first bit of frame f in a tick's hole bitmap: 64-bit aligned, frames never share a word
__device__ __forceinline__ long long hole_base_bit(const FrameDesc &fd, int f) { return ((fd.depth_off + 63) & ~63ll) + 64ll * f; }

struct BandArgs {
    const FrameDesc *frames;
    const BandDesc *bands;           // the bands of one tick
    WarpSrc src;
    unsigned short *out_d;           // the closed maps
    unsigned char *out_c;
    unsigned char *holes;            // [n_ticks][holes_tick_bytes] one bit per pixel: a hole of the un-closed map
    unsigned int *work;              // [n_ticks][2 * pixels per tick]: frame f's list starts at 2 * depth_off, 2 * npix entries
    int *work_cnt;                   // [n_ticks * n_frames] x kCntStride
    int bands_per_tick, n_frames, rows;
    long long tick_pix_stride, holes_tick_bytes;
};

constexpr int kBandList = 8192;   // most candidates one pass over the band's rows lists (16-bit local pixel indices); taller / wider bands go in chunks of rows

// entries of the candidate list of a band
__host__ __device__ inline int band_list_entries(int rows, int w)
{
    const int all = rows * w;
    return all <= kBandList ? all : (kBandList > w ? kBandList : w);
}

// LDS of one band workgroup: (rows + 2) x w depths and colours (each behind up to 15 bytes of lead so that the rows that leave
// sit at their destination's address modulo 16), the candidate list, a counter
__host__ __device__ inline int band_lds_bytes(int rows, int w)
{
    const int band_px = (rows + 2) * w;
    return 16 + ((2 * band_px + 15) & ~15) + 16 + ((3 * band_px + 15) & ~15) + ((2 * band_list_entries(rows, w) + 15) & ~15) + 16;
}

// copies n bytes from LDS to global memory; (lds offset from the 16-byte aligned LDS base) = (global address) modulo 16
__device__ __forceinline__ void store_band_run(unsigned char *dst, const unsigned char *lds, int n)
{
    const int lead = (int)(reinterpret_cast<uintptr_t>(dst) & 15);
    const int head = lead ? min(n, 16 - lead) : 0;
    const int chunks = (n - head) >> 4;
    const uint4 *l16 = reinterpret_cast<const uint4 *>(lds + head);
    uint4 *g16 = reinterpret_cast<uint4 *>(dst + head);
    for (int j = threadIdx.x; j < chunks; j += kBandThreads) g16[j] = l16[j];
    const int tail0 = head + 16 * chunks;
    if ((int)threadIdx.x < head) dst[threadIdx.x] = lds[threadIdx.x];
    if (tail0 + (int)threadIdx.x < n) dst[tail0 + threadIdx.x] = lds[tail0 + threadIdx.x];
}

// both 16-bit halves of x replaced by min(half, 1): one v_pk_min_u16 (the compiler splits the vector form into two scalar ones)
__device__ __forceinline__ unsigned int pk_min1(unsigned int x)
{
    unsigned int r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(0x00010001u));
    return r;
}

template <bool GATHER, bool VEC>
__global__ __launch_bounds__(kBandThreads) void radial_band_kernel(const BandArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int tick = blockIdx.x / a.bands_per_tick;
    const int band = blockIdx.x - tick * a.bands_per_tick;
    const BandDesc bd = a.bands[band];
    const FrameDesc fd = a.frames[bd.frame];
    const int w = fd.w, h = fd.h, y0 = bd.y0;
    const int nrows = min(a.rows, h - y0);
    const long long fb = tick * a.tick_pix_stride + fd.depth_off;
    const int tf = tick * a.n_frames + bd.frame;
    const int band_px = (a.rows + 2) * w;
    unsigned short *gd = a.out_d + fb + (long long)y0 * w;
    unsigned char *gc = a.out_c + 3 * (fb + (long long)y0 * w);
    // local pixel 0 is pixel (y0 - 1) * w of the frame; the rows that leave start at local pixel w
    const int lead_d = (int)((reinterpret_cast<uintptr_t>(gd) - 2 * (uintptr_t)w) & 15);
    const int lead_c = (int)((reinterpret_cast<uintptr_t>(gc) - 3 * (uintptr_t)w) & 15);
    const int off_c = 16 + ((2 * band_px + 15) & ~15);
    const int off_l = off_c + 16 + ((3 * band_px + 15) & ~15);
    unsigned short *s_d = reinterpret_cast<unsigned short *>(smem + lead_d);
    unsigned char *s_c = smem + off_c + lead_c;
    unsigned short *s_list = reinterpret_cast<unsigned short *>(smem + off_l);
    const int list_entries = band_list_entries(a.rows, w);
    int *s_n = reinterpret_cast<int *>(smem + off_l + ((2 * list_entries + 15) & ~15));
    const int pl0 = (y0 - 1) * w;                       // frame pixel of local pixel 0 (negative for the first band)
    const int lo = pl0 < 0 ? -pl0 : 0;                  // local pixels [lo, hi) exist in the frame
    const int hi = min((nrows + 2) * w, fd.npix - pl0);

    // ---- 1. the un-closed band into LDS ----
    if (GATHER) {
#ifndef LSN_BAND_FLY
#define LSN_BAND_FLY 8
#endif
        constexpr int kFly = LSN_BAND_FLY;
        for (int i0 = lo + tid; i0 < hi; i0 += kFly * kBandThreads) {
            int p[kFly];
            bool in[kFly];
#pragma unroll
            for (int k = 0; k < kFly; k++) {
                p[k] = pl0 + i0 + k * kBandThreads;
                in[k] = i0 + k * kBandThreads < hi;
            }
            unsigned int d[kFly], c[kFly];
#ifndef LSN_BAND_SPEC
#define LSN_BAND_SPEC false
#endif
            gather_batch<kFly, LSN_BAND_SPEC>(a.src, fb, fd.depth_off, fd.npix, pl0 + lo, p, in, d, c);
#pragma unroll
            for (int k = 0; k < kFly; k++) {
                if (!in[k]) continue;
                const int i = i0 + k * kBandThreads;
                s_d[i] = (unsigned short)d[k];
                unsigned char *c3 = s_c + 3 * i;
                c3[0] = (unsigned char)c[k];
                c3[1] = (unsigned char)(c[k] >> 8);
                c3[2] = (unsigned char)(c[k] >> 16);
            }
        }
        if (fd.npix == 1 && tid == 0) {   // local pixel `lo` is the frame's only pixel, and this thread stored it
            const unsigned int c1 = one_pixel_colour(a.src, fb, s_d[lo]);
            unsigned char *c3 = s_c + 3 * lo;
            c3[0] = (unsigned char)c1;
            c3[1] = (unsigned char)(c1 >> 8);
            c3[2] = (unsigned char)(c1 >> 16);
        }
    } else if (VEC) {
        const unsigned short *sd = a.src.depth + fb + pl0;
        const unsigned char *sc = a.src.rgb + 3 * (fb + pl0);
        for (int i = lo + 8 * tid; i < hi; i += 8 * kBandThreads) {    // w % 8 == 0: lo, hi and every frame start are multiples of 8
            *reinterpret_cast<uint4 *>(s_d + i) = *reinterpret_cast<const uint4 *>(sd + i);
            const uint2 *cp = reinterpret_cast<const uint2 *>(sc + 3 * i);
            uint2 *lp = reinterpret_cast<uint2 *>(s_c + 3 * i);
            const uint2 c0 = cp[0], c1 = cp[1], c2 = cp[2];
            lp[0] = c0; lp[1] = c1; lp[2] = c2;
        }
    } else {
        const unsigned short *sd = a.src.depth + fb + pl0;
        const unsigned char *sc = a.src.rgb + 3 * (fb + pl0);
        for (int i = lo + tid; i < hi; i += kBandThreads) {
            s_d[i] = sd[i];
            s_c[3 * i] = sc[3 * i];
            s_c[3 * i + 1] = sc[3 * i + 1];
            s_c[3 * i + 2] = sc[3 * i + 2];
        }
    }
    __syncthreads();

    // ---- 2. the band leaves as it is (un-closed); the fills follow below as single stores, behind the barrier that ends this phase ----
    store_band_run(reinterpret_cast<unsigned char *>(gd), reinterpret_cast<const unsigned char *>(s_d + w), 2 * nrows * w);
    store_band_run(gc, s_c + 3 * w, 3 * nrows * w);

    // ---- 3. the holes, a few rows at a time: one bit per pixel for the second pass; the candidates (interior holes with >= 5 valid
    //         neighbours) go on a short list in LDS and are evaluated one per lane against the un-closed band (which nothing modifies:
    //         exact unless one of the hole's predecessors gets filled); a fill is stored straight to the output and lists its hole
    //         successors (right, down-left, down, down-right) on the frame's work list ----
    const long long hole_bit0 = hole_base_bit(fd, bd.frame);
    unsigned char *holes = a.holes + tick * a.holes_tick_bytes;
    unsigned int *work = a.work + 2 * fb;
    int *work_cnt = a.work_cnt + kCntStride * tf;
    const int rows_per_chunk = max(1, list_entries / w);   // a chunk's candidates always fit the list
    for (int r0 = 0; r0 < nrows; r0 += rows_per_chunk) {
        const int r1 = min(nrows, r0 + rows_per_chunk);
        __syncthreads();                                 // the previous chunk's list has been consumed (first chunk: the band's stores are done)
        if (tid == 0) *s_n = 0;
        __syncthreads();
        if (VEC) {
            const int gw = w >> 3, ngroups = (r1 - r0) * gw;
            for (int g0 = 0; g0 < ngroups; g0 += kBandThreads) {
                const int g = g0 + tid;
                unsigned int cb = 0;
                int li = 0;
                if (g < ngroups) {
                    const int r = r0 + g / gw, x0 = (g % gw) << 3, y = y0 + r;
                    li = (r + 1) * w + x0;
                    unsigned int m[3];
#pragma unroll
                    for (int rr = 0; rr < 3; rr++) {
                        const unsigned short *row = s_d + li + (rr - 1) * w;
                        const uint4 c = *reinterpret_cast<const uint4 *>(row);
                        const unsigned int left = x0 > 0 ? row[-1] : 0u, right = x0 + 8 < w ? row[8] : 0u;
                        // min(depth, 1) of two pixels at a time (packed u16), then the 0/1 halves of four dwords gathered into eight bits
                        const unsigned int p0 = pk_min1(c.x) | (pk_min1(c.y) << 2), p1 = pk_min1(c.z) | (pk_min1(c.w) << 2);   // bits 0, 2 | 16, 18
                        const unsigned int q0 = (p0 | (p0 >> 15)) & 0xFu, q1 = (p1 | (p1 >> 15)) & 0xFu;                        // pixels 0..3, 4..7
                        m[rr] = (left ? 1u : 0u) | ((q0 | (q1 << 4)) << 1) | (right ? (1u << 9) : 0u);   // bit c: column x0 - 1 + c holds a valid depth
                    }
                    const unsigned int hb = ~(m[1] >> 1) & 0xFFu;      // bit k: pixel x0 + k is a hole
                    holes[(hole_bit0 + (long long)y * w + x0) >> 3] = (unsigned char)hb;
                    if (y >= 1 && y < h - 1) {                         // :223-224 interior pixels only
                        // at least five of the eight neighbours valid, for the eight pixels at once: bit k of each word below is one neighbour
                        // of pixel x0 + k, a bit-sliced adder counts them (total = t0 + 2 v0 + 4 w0 + 8 g1)
                        const unsigned int n0 = m[0], n1 = m[0] >> 1, n2 = m[0] >> 2, n3 = m[1], n4 = m[1] >> 2, n5 = m[2], n6 = m[2] >> 1, n7 = m[2] >> 2;
                        const unsigned int s1 = n0 ^ n1 ^ n2, c1 = (n0 & n1) | (n2 & (n0 ^ n1));
                        const unsigned int s2 = n5 ^ n6 ^ n7, c2 = (n5 & n6) | (n7 & (n5 ^ n6));
                        const unsigned int s3 = n3 ^ n4, c3 = n3 & n4;
                        const unsigned int t0 = s1 ^ s2 ^ s3, d1 = (s1 & s2) | (s3 & (s1 ^ s2));
                        const unsigned int u0 = c1 ^ c2 ^ c3, e1 = (c1 & c2) | (c3 & (c1 ^ c2));
                        const unsigned int v0 = u0 ^ d1, f1 = u0 & d1;
                        const unsigned int w0 = e1 ^ f1, g1 = e1 & f1;
                        unsigned int inside = 0xFFu;                                   // 1 <= x < w - 1
                        if (x0 == 0) inside &= ~1u;
                        if (x0 + 8 == w) inside &= ~0x80u;
                        cb = (g1 | (w0 & (v0 | t0))) & hb & inside;
                    }
                }
                int slot = wave_reserve(s_n, __popc(cb));
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if ((cb >> k) & 1u) s_list[slot++] = (unsigned short)(li + k);
            }
        } else {
            // any width: pixel by pixel (the bitmap was cleared by the host)
            unsigned int *hw = reinterpret_cast<unsigned int *>(holes);
            const int n_px = (r1 - r0) * w;
            for (int i0 = 0; i0 < n_px; i0 += kBandThreads) {
                const int i = r0 * w + i0 + tid;
                bool cand = false;
                const int li = w + i;
                if (i0 + tid < n_px && s_d[li] == 0) {
                    const int r = i / w, x = i - r * w, y = y0 + r;
                    const long long bit = hole_bit0 + (long long)y * w + x;
                    atomicOr(&hw[bit >> 5], 1u << (bit & 31));
                    if (y >= 1 && y < h - 1 && x >= 1 && x < w - 1) {
                        int nv = 0;
#pragma unroll
                        for (int j = 0; j < 8; j++) nv += s_d[li + kDy[j] * w + kDx[j]] != 0 ? 1 : 0;
                        cand = nv >= 5;
                    }
                }
                const int slot = wave_reserve(s_n, cand ? 1 : 0);
                if (cand) s_list[slot] = (unsigned short)li;
            }
        }
        __syncthreads();
        const int n_cand = *s_n;
        for (int i0 = 0; i0 < n_cand; i0 += kBandThreads) {
            const int i = i0 + tid;
            if (i0 + (tid & ~63) >= n_cand) continue;                 // nothing left for this wave (wave-uniform)
            unsigned int succ = 0;
            int x = 0, y = 0;
            if (i < n_cand) {
                const int q = (int)s_list[i];
                int nb[8];
#pragma unroll
                for (int j = 0; j < 8; j++) nb[j] = s_d[q + kDy[j] * w + kDx[j]];
                int n, sum;
                unsigned int accepted;
                accept_chain(nb, n, sum, accepted);
                if (n > 4) {                                                                       // :250-256
                    unsigned int nc[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const unsigned char *c3 = s_c + 3 * (q + kDy[j] * w + kDx[j]);
                        nc[j] = (accepted >> j) & 1u ? ((unsigned int)c3[0] | ((unsigned int)c3[1] << 8) | ((unsigned int)c3[2] << 16)) : 0u;
                    }
                    const unsigned int rgb = average_colour(nc, accepted, n);
                    const long long pos = fb + pl0 + q;
                    a.out_d[pos] = (unsigned short)div_small(sum, n);
                    a.out_c[3 * pos] = (unsigned char)rgb;
                    a.out_c[3 * pos + 1] = (unsigned char)(rgb >> 8);
                    a.out_c[3 * pos + 2] = (unsigned char)(rgb >> 16);
                    int yl = (int)((float)q * fd.inv_w);
                    x = q - yl * w;
                    if (x < 0) { yl--; x += w; }
                    if (x >= w) { yl++; x -= w; }
                    y = y0 - 1 + yl;
                    // its hole successors now depend on the order: onto the frame's list (interior pixels only, :223-224)
                    const bool below = y + 1 < h - 1;
                    succ = ((nb[4] == 0 && x + 1 < w - 1) ? 1u : 0u) | ((nb[5] == 0 && x - 1 >= 1 && below) ? 2u : 0u) |
                           ((nb[6] == 0 && below) ? 4u : 0u) | ((nb[7] == 0 && x + 1 < w - 1 && below) ? 8u : 0u);
                }
            }
            int slot = wave_reserve(work_cnt, __popc(succ));
#pragma unroll
            for (int sidx = 0; sidx < 4; sidx++) {
                if ((succ >> sidx) & 1u) {
                    const int j = 4 + sidx;   // neighbours 4..7 are the successors
                    if (slot < 2 * fd.npix) work[slot] = (unsigned int)((y + kDy[j]) * w + x + kDx[j]);   // (a frame lists < 24/13 npix entries)
                    slot++;
                }
            }
        }
    }
}

#ifndef LSN_FIX_THREADS
#define LSN_FIX_THREADS 256
#endif
constexpr int kFixThreads = LSN_FIX_THREADS;   // measured on 512 scene frames: 1024 threads 377 us, 512: 274, 256: 245 -- the rounds are short, idle waves only add barrier time
constexpr int kFixList = 8192;   // entries per round list (LDS, two lists: 64 KB)

struct FixArgs {
    const FrameDesc *frames;
    unsigned short *out_d;
    unsigned char *out_c;
    const unsigned char *holes;
    const unsigned int *work;        // the list the per-frame kernel starts from: 2 * npix entries per frame at 2 * (first pixel of the frame)
    int *work_cnt;                   // [3][n_ticks * n_frames] x kCntStride: band kernel -> first grid-wide round -> second -> per-frame kernel
    int n_frames, list_cap;          // list_cap: entries a round list may hold (kFixList; the tests shrink it to force the sweeps)
    int n_tf, cnt_index;             // cnt_index: which of the three counter arrays the per-frame kernel starts from
    long long tick_pix_stride, holes_tick_bytes;
};

// Re-evaluates hole p of a frame (out_d / out_c / holes point at the frame, bit0 = its first bit in `holes`): predecessors as they
// are now, successors as the un-closed map had them.  Returns true and writes the pixel when its value changed; succ = its hole
// successors (bit s = neighbour 4 + s), which then have to be looked at again.
__device__ __forceinline__ bool fix_pixel(unsigned short *out_d, unsigned char *out_c, const unsigned char *holes, long long bit0, int p, int w, int h,
                                          float inv_w, unsigned int &succ, int &x, int &y)
{
    y = (int)((float)p * inv_w);
    x = p - y * w;
    if (x < 0) { y--; x += w; }
    if (x >= w) { y++; x -= w; }
    // three rows of three pixels: depths as dword + word, colours as 8 + 1 bytes (nothing is read outside the nine pixels)
    unsigned int dv[3][3], cv[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int q = p + (r - 1) * w - 1;
        const unsigned int d01 = *reinterpret_cast<const u32_ua *>(out_d + q);
        dv[r][0] = d01 & 0xFFFFu;
        dv[r][1] = d01 >> 16;
        dv[r][2] = out_d[q + 2];
        const unsigned long long c8 = *reinterpret_cast<const u64_ua *>(out_c + 3 * (long long)q);
        cv[r][0] = (unsigned int)c8 & 0xFFFFFFu;
        cv[r][1] = (unsigned int)(c8 >> 24) & 0xFFFFFFu;
        cv[r][2] = (unsigned int)(c8 >> 48) | ((unsigned int)out_c[3 * (long long)q + 8] << 16);
    }
    // the hole bits of the four successors: p + 1, and p + w - 1 .. p + w + 1
    const long long b1 = bit0 + p + 1, b2 = bit0 + p + w - 1;
    const unsigned int h1 = (holes[b1 >> 3] >> (b1 & 7)) & 1u;
    const unsigned int h2 = ((unsigned int)*reinterpret_cast<const u16_ua *>(holes + (b2 >> 3)) >> (b2 & 7)) & 7u;
    const int nb[8] = {(int)dv[0][0], (int)dv[0][1], (int)dv[0][2], (int)dv[1][0], h1 ? 0 : (int)dv[1][2],
                       (h2 & 1u) ? 0 : (int)dv[2][0], (h2 & 2u) ? 0 : (int)dv[2][1], (h2 & 4u) ? 0 : (int)dv[2][2]};
    const unsigned int nc[8] = {cv[0][0], cv[0][1], cv[0][2], cv[1][0], cv[1][2], cv[2][0], cv[2][1], cv[2][2]};
    const unsigned int od = dv[1][1], orgb = cv[1][1];
    unsigned int nd = 0, nrgb = 0;   // an unfilled hole keeps what the warp left (:250): nothing
    int n, sum;
    unsigned int accepted;
    accept_chain(nb, n, sum, accepted);
    if (n > 4) {
        nd = div_small(sum, n);
        nrgb = average_colour(nc, accepted, n);
    }
    succ = 0;
    if (nd == od && nrgb == orgb) return false;
    out_d[p] = (unsigned short)nd;
    out_c[3 * (long long)p] = (unsigned char)nrgb;
    out_c[3 * (long long)p + 1] = (unsigned char)(nrgb >> 8);
    out_c[3 * (long long)p + 2] = (unsigned char)(nrgb >> 16);
    const bool below = y + 1 < h - 1;
    succ = ((h1 && x + 1 < w - 1) ? 1u : 0u) | (((h2 & 1u) && x - 1 >= 1 && below) ? 2u : 0u) | (((h2 & 2u) && below) ? 4u : 0u) |
           (((h2 & 4u) && x + 1 < w - 1 && below) ? 8u : 0u);
    return true;
}

constexpr int kOverflow = 0x40000000;   // a list counter at or above this: the list overflowed somewhere up the chain, sweep the frame

// One re-evaluation round over ALL frames at once (the first rounds hold thousands of pixels per frame: one workgroup per frame would
// walk them twelve at a time... 256 at a time, a dozen dependent memory round trips).  Reads frame tf's list (cnt_in entries at
// list_in + in_stride * first pixel), appends the hole successors of every changed pixel to its list in list_out.  Pixels of one
// round that are neighbours may see each other half-written; whoever changes a pixel re-lists its successors, so the next round (behind
// a kernel boundary) evaluates them again with everything in place.
__global__ __launch_bounds__(kThreads) void close_fix_round_kernel(const FixArgs a, const unsigned int *list_in, const int *cnt_in, int in_stride,
                                                                   int in_cap_num, int in_cap_den, unsigned int *list_out, int *cnt_out,
                                                                   int out_stride, int out_cap_num, int out_cap_den, int blocks_per_frame)
{
    const int tf = blockIdx.x / blocks_per_frame, b = blockIdx.x - tf * blocks_per_frame;
    const int tick = tf / a.n_frames, f = tf - tick * a.n_frames;
    const FrameDesc fd = a.frames[f];
    const int w = fd.w, h = fd.h;
    const long long fb = tick * a.tick_pix_stride + fd.depth_off;
    const int n_in = cnt_in[kCntStride * tf];
    const int in_cap = (int)((long long)fd.npix * in_cap_num / in_cap_den), out_cap = (int)((long long)fd.npix * out_cap_num / out_cap_den);
    if (n_in > in_cap) {                                   // overflowed: hand the verdict on, the per-frame kernel sweeps
        if (b == 0 && threadIdx.x == 0) cnt_out[kCntStride * tf] = kOverflow;
        return;
    }
    unsigned short *out_d = a.out_d + fb;
    unsigned char *out_c = a.out_c + 3 * fb;
    const unsigned char *holes = a.holes + tick * a.holes_tick_bytes;
    const long long bit0 = hole_base_bit(fd, f);
    const unsigned int *lin = list_in + in_stride * fb;
    unsigned int *lout = list_out + out_stride * fb;
    for (int i0 = b * kThreads; i0 < n_in; i0 += blocks_per_frame * kThreads) {
        const int i = i0 + (int)threadIdx.x;
        if (i0 + (int)(threadIdx.x & ~63u) >= n_in) continue;   // nothing left for this wave (wave-uniform)
        unsigned int succ = 0;
        int x = 0, y = 0;
        if (i < n_in) fix_pixel(out_d, out_c, holes, bit0, (int)lin[i], w, h, fd.inv_w, succ, x, y);
        int slot = wave_reserve(cnt_out + kCntStride * tf, __popc(succ));
#pragma unroll
        for (int sidx = 0; sidx < 4; sidx++) {
            if ((succ >> sidx) & 1u) {
                const int k = 4 + sidx;
                if (slot < out_cap) lout[slot] = (unsigned int)((y + kDy[k]) * w + x + kDx[k]);
                slot++;
            }
        }
    }
}

__global__ __launch_bounds__(kFixThreads) void close_fix_kernel(const FixArgs a)
{
    __shared__ unsigned int lists[2][kFixList];
    __shared__ int s_n[2];
    __shared__ int s_flag;
    const int tf = blockIdx.x;
    const int tick = tf / a.n_frames, f = tf - tick * a.n_frames;
    const FrameDesc fd = a.frames[f];
    const int w = fd.w, h = fd.h;
    const long long fb = tick * a.tick_pix_stride + fd.depth_off;
    // written and re-read by the waves of this workgroup only, a round apart: they share the CU's L1, and the barrier between two
    // rounds (workgroup-scope release / acquire) orders the accesses -- no cache bypass needed
    unsigned short *out_d = a.out_d + fb;
    unsigned char *out_c = a.out_c + 3 * fb;
    const unsigned char *holes = a.holes + tick * a.holes_tick_bytes;
    const long long bit0 = hole_base_bit(fd, f);
    int *cnt = a.work_cnt + kCntStride * tf;
    const size_t cnt_step = (size_t)kCntStride * a.n_tf;
    const int n_listed = cnt[a.cnt_index * cnt_step];
    const int n0 = n_listed > 2 * fd.npix ? 0 : n_listed;   // (a frame whose list overflowed is swept instead)
    __syncthreads();                                   // everybody has read the counter ...
    if (threadIdx.x == 0) {
        cnt[0] = cnt[cnt_step] = cnt[2 * cnt_step] = 0;   // ... and all three are left cleared for the next call
        s_n[0] = s_n[1] = 0;
        s_flag = 0;
    }
    __syncthreads();
    const unsigned int *glist = a.work + 2 * fb;
    int cur = 0;   // lists[cur] is read, lists[1 - cur] is filled; round 0 reads the global list instead
    bool sweep = n_listed > 2 * fd.npix;
    for (int round = 0;; round++) {
        const int n_items = round == 0 ? n0 : s_n[cur];
        if (n_items == 0) break;
        for (int i0 = 0; i0 < n_items; i0 += kFixThreads) {
            const int i = i0 + (int)threadIdx.x;
            if (i0 + (int)(threadIdx.x & ~63u) >= n_items) continue;   // nothing left for this wave (wave-uniform)
            unsigned int succ = 0;
            int x = 0, y = 0;
            if (i < n_items) fix_pixel(out_d, out_c, holes, bit0, (int)(round == 0 ? glist[i] : lists[cur][i]), w, h, fd.inv_w, succ, x, y);
            int slot = wave_reserve(&s_n[1 - cur], __popc(succ));
#pragma unroll
            for (int sidx = 0; sidx < 4; sidx++) {
                if ((succ >> sidx) & 1u) {
                    const int k = 4 + sidx;
                    if (slot < a.list_cap) lists[1 - cur][slot] = (unsigned int)((y + kDy[k]) * w + x + kDx[k]);
                    else s_flag = 1;
                    slot++;
                }
            }
        }
        __threadfence_block();
        __syncthreads();   // this round's writes are in place before the next round reads them
        if (s_flag) { sweep = true; break; }
        cur = 1 - cur;
        if (threadIdx.x == 0) s_n[1 - cur] = 0;
        __syncthreads();
    }
    // A list outgrew LDS: sweep over every interior hole of the frame until a whole sweep changes nothing.  Same fixed point (a pixel
    // is final once its predecessors are and it has been looked at again), no list; slow, and only adversarial frames get here.
    while (sweep) {
        __syncthreads();
        if (threadIdx.x == 0) s_flag = 0;
        __syncthreads();
        bool changed = false;
        for (int p = w + 1 + (int)threadIdx.x; p < fd.npix - w - 1; p += kFixThreads) {
            const long long b = bit0 + p;
            if (!((holes[b >> 3] >> (b & 7)) & 1u)) continue;
            int x, y;
            {
                y = (int)((float)p * fd.inv_w);
                x = p - y * w;
                if (x < 0) { y--; x += w; }
                if (x >= w) { y++; x -= w; }
            }
            if (x < 1 || x >= w - 1) continue;
            unsigned int succ;
            changed |= fix_pixel(out_d, out_c, holes, bit0, p, w, h, fd.inv_w, succ, x, y);
        }
        if (changed) s_flag = 1;
        __threadfence_block();
        __syncthreads();
        sweep = s_flag != 0;
    }
}

// The bands of one tick for `rows` rows per band; returns their number.
static int make_bands(const LsnFusion *p, int rows, std::vector<BandDesc> &bands)
{
    bands.clear();
    for (int f = 0; f < p->n_maps; f++)
        for (int y0 = 0; y0 < p->h[f]; y0 += rows) bands.push_back(BandDesc{f, y0});
    return (int)bands.size();
}

}  // namespace

static int radial_correct_on(LsnFusion *p, const float *intr_params, const void *d_depth_in, const void *d_colors_in, void *d_depth, void *d_colors,
                             hipStream_t s);

// The plan's radial scratch -- warp tables, band list, hole bitmap, work lists and their counters -- is shared by all calls on the plan: a
// call on ANOTHER stream than the previous one's first waits for that chain's end (an event recorded behind every chain), so that nothing
// of it is still counting, listing or reading when this call clears and refills the scratch.  Calls on one stream are ordered by the stream.
static int radial_correct(LsnFusion *p, const float *intr_params, const void *d_depth_in, const void *d_colors_in, void *d_depth, void *d_colors,
                          hipStream_t s)
{
    LSN_HIP(hipSetDevice(p->device));
    if (!p->radial_done) LSN_HIP(hipEventCreateWithFlags(&p->radial_done, hipEventDisableTiming));
    if (p->radial_chain_open && p->work_cnt_stream != s) LSN_HIP(hipStreamWaitEvent(s, p->radial_done, 0));
    const int rc = radial_correct_on(p, intr_params, d_depth_in, d_colors_in, d_depth, d_colors, s);
    // (also behind a call that failed half-way: whatever it did enqueue is what the next stream has to wait for)
    if (hipEventRecord(p->radial_done, s) == hipSuccess) p->radial_chain_open = true;
    else (void)hipGetLastError();
    p->work_cnt_stream = s;
    return rc;
}

static int radial_correct_on(LsnFusion *p, const float *intr_params, const void *d_depth_in, const void *d_colors_in, void *d_depth, void *d_colors,
                             hipStream_t s)
{
    const bool in_place = d_depth_in == d_depth && d_colors_in == d_colors;
    if (!in_place && (d_depth_in == d_depth || d_colors_in == d_colors)) {
        lsn::set_error("lsnFusionRadialCorrectTo: depth and colours must both be in place or both out of place");
        return -1;
    }
    const size_t npix = (size_t)p->cap * p->n_ticks;
    if (!in_place) {
        // out of place the bands warp straight from the input (no scratch copy): an output range that overlaps an input range in part
        // would be overwritten by one band while another still reads it as a warp source
        auto overlap = [](const void *a, size_t na, const void *b, size_t nb) {
            const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
            return x < y + nb && y < x + na;
        };
        if (overlap(d_depth_in, 2 * npix, d_depth, 2 * npix) || overlap(d_colors_in, 3 * npix, d_colors, 3 * npix) ||
            overlap(d_depth_in, 2 * npix, d_colors, 3 * npix) || overlap(d_colors_in, 3 * npix, d_depth, 2 * npix)) {
            lsn::set_error("lsnFusionRadialCorrectTo: the output buffers overlap the input buffers in part (pass the same pointers for an in-place correction)");
            return -1;
        }
    }
    if (p->radial.reserve(sizeof(RadialParams) * p->n_maps)) return -1;
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
        // the warp candidates of this calibration (one tick's worth of pixels), then their compact form
        if (p->cand.reserve(16 * (size_t)p->cap) || p->ctab.reserve(4 * (size_t)p->cap + 16)) return -1;
        unsigned int *count = p->ctab.as<unsigned int>();   // the per-destination counters of the fill pass live where the compact table will
        LSN_HIP(hipMemsetAsync(count, 0, 4 * (size_t)p->cap, s));
        LSN_HIP(hipMemsetAsync(p->cand.p, 0, 16 * (size_t)p->cap, s));
        LSN_HIP(hipMemsetAsync(p->misc.as<char>() + 64, 0, sizeof(int), s));
        int *overflow = reinterpret_cast<int *>(p->misc.as<char>() + 64);
        hipLaunchKernelGGL(radial_cand_fill_kernel, dim3(p->tiles_per_tick), dim3(kThreads), 0, s, p->frames.as<FrameDesc>(),
                           p->tile_frame.as<TileDesc>(), p->radial.as<RadialParams>(), count, p->cand.as<unsigned int>(), overflow);
        hipLaunchKernelGGL(radial_cand_sort_kernel, dim3((unsigned)((p->cap + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p->cand.as<uint4>(),
                           p->cap);
        hipLaunchKernelGGL(radial_cand_pack_kernel, dim3(p->tiles_per_tick), dim3(kThreads), 0, s, p->frames.as<FrameDesc>(),
                           p->tile_frame.as<TileDesc>(), (const uint4 *)p->cand.as<uint4>(), p->ctab.as<unsigned int>());
        int ov = 0;
        LSN_HIP(hipMemcpyAsync(&ov, overflow, sizeof(int), hipMemcpyDeviceToHost, s));
        LSN_HIP(hipStreamSynchronize(s));
        p->cand_overflow = ov != 0;
        p->radial_intr.assign(intr_params, intr_params + 7 * (size_t)p->n_maps);
        p->cand_valid = true;
    }
    // (read on every call: the tests switch them inside one process)
    const char *force = getenv("LSN_RADIAL_FORCE_ATOMIC");  // tests: take the atomicMax path even when the table did not overflow
    const char *close_env = getenv("LSN_RADIAL_CLOSE"), *tiny_env = getenv("LSN_RADIAL_TINY_LISTS");
    const bool atomic_warp = p->cand_overflow || (force && atoi(force) != 0);
    const bool wavefront = close_env && !strcmp(close_env, "wavefront");
    const bool tiny_lists = tiny_env && atoi(tiny_env) != 0;   // tests: force the sweeps of the second pass
    const bool vec_ptrs = ((uintptr_t)d_depth & 15) == 0 && ((uintptr_t)d_colors & 7) == 0 && ((uintptr_t)d_depth_in & 15) == 0 &&
                          ((uintptr_t)d_colors_in & 7) == 0 && (p->cap % 8) == 0;
    const bool vec = p->vec_ok && vec_ptrs;
    WarpSrc src;
    src.depth = static_cast<const unsigned short *>(d_depth_in);
    src.rgb = static_cast<const unsigned char *>(d_colors_in);
    src.ctab = p->ctab.as<unsigned int>();
    src.cand = p->cand.as<uint4>();
    src.last_px = (long long)npix - 1;
    // The un-closed maps go through memory when the closing cannot warp by itself: in place (a band would overwrite another band's
    // sources), after the atomicMax warp, and for the wavefront kernel.
    const bool scratch = in_place || atomic_warp || wavefront;
    if (scratch) {
        if (p->map_copy.reserve(2 * npix + 16) || p->colors_copy.reserve(3 * npix + 16)) return -1;
        if (!atomic_warp) {
            hipLaunchKernelGGL(radial_gather_pack_kernel, dim3(grid), dim3(kThreads), 0, s, p->frames.as<FrameDesc>(), p->tile_frame.as<TileDesc>(), src,
                               p->map_copy.as<unsigned short>(), p->colors_copy.as<unsigned char>(), p->tiles_per_tick, p->cap);
        } else {
            if (p->winner.reserve(4 * npix)) return -1;
            LSN_HIP(hipMemsetAsync(p->winner.p, 0, 4 * npix, s));
            hipLaunchKernelGGL(radial_warp_kernel, dim3(grid), dim3(kThreads), 0, s, p->frames.as<FrameDesc>(), p->tile_frame.as<TileDesc>(),
                               p->radial.as<RadialParams>(), src.depth, p->winner.as<unsigned int>(), p->tiles_per_tick, p->cap);
            hipLaunchKernelGGL(radial_gather_kernel, dim3(grid), dim3(kThreads), 0, s, p->frames.as<FrameDesc>(), p->tile_frame.as<TileDesc>(),
                               src.depth, src.rgb, (const unsigned int *)p->winner.as<unsigned int>(), p->map_copy.as<unsigned short>(),
                               p->colors_copy.as<unsigned char>(), p->tiles_per_tick, p->cap);
        }
        src.depth = p->map_copy.as<unsigned short>();
        src.rgb = p->colors_copy.as<unsigned char>();
    }
    const int n_tf = p->n_maps * p->n_ticks;
    if (wavefront) {
        int max_h = 1;
        for (int v : p->h) max_h = v > max_h ? v : max_h;
        int rows = max_h - 2 < 64 ? 64 : ((max_h - 2 + 63) / 64) * 64;
        if (rows > 768) rows = 768;  // (rows + 2) x 32 columns x 6 B of LDS rings must fit in 160 KB
        // One band per frame is the shortest chain of steps, but (rows + 2) x 192 B of LDS per workgroup then allows a single
        // frame per CU.  With more frames than CUs, 256-row bands (49.5 KB: three frames per CU) win.
        if ((long long)n_tf > 256 && rows > 256) rows = 256;
        if (const char *env = getenv("LSN_RADIAL_ROWS")) {  // tuning: rows per band (multiple of 64, <= 768)
            const int v = atoi(env);
            if (v >= 64 && v <= 768 && v % 64 == 0) rows = v;
        }
        const size_t ring_bytes = (sizeof(unsigned int) + sizeof(unsigned short)) * kRing * (rows + 2);
        hipLaunchKernelGGL(radial_close_kernel, dim3((unsigned)n_tf), dim3(rows), ring_bytes, s, p->frames.as<FrameDesc>(), p->n_maps,
                           p->map_copy.as<unsigned short>(), p->colors_copy.as<unsigned char>(), p->cap);
        LSN_HIP(hipGetLastError());
        // :259-260 the corrected maps replace the inputs
        LSN_HIP(hipMemcpyAsync(d_depth, p->map_copy.p, 2 * npix, hipMemcpyDeviceToDevice, s));
        LSN_HIP(hipMemcpyAsync(d_colors, p->colors_copy.p, 3 * npix, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    // bands: as many rows as fit the LDS budget of two workgroups per CU for the widest frame
    int max_w = 1, max_h = 1;
    for (int i = 0; i < p->n_maps; i++) { max_w = std::max(max_w, p->w[i]); max_h = std::max(max_h, p->h[i]); }
    // 256 threads on 6 rows of a 512-wide frame: 27 KB, six workgroups of four waves per CU.  Round 6 (512 scene frames, radial_band_kernel
    // alone, profiles/r06_ab_band.txt): 512 threads x 12 rows (rounds 3-5: three workgroups of eight waves) 462-467 us, 256 x 6 417-419,
    // 256 x 4 426, 256 x 7 432, 256 x 5 470, 256 x 8 489, 256 x 12 551, 128 x 4 496, 1024 x 12 718 -- the same 24 waves per CU, but six
    // groups marching through their phases (gather: memory; closing: VALU + LDS) on their own fill each other's waits better than three
    // (eight pixels in flight per thread: the 8 x 512 pixels of such a band are two full trips for 256 threads -- 378-391 us; six in flight:
    // 416-419, 16: 468.)  Wider frames: fewer rows, down to two, to stay near that footprint -- 16 x 1024x1024 x 8 scene ticks, whole
    // correction: 5 rows (the 52 KB budget of rounds 3-5) 1.31 ms, 2 / 3 / 4 rows 1.22-1.25, 6 rows 1.26, 8 rows 1.46.
    int rows = 6;
    while (rows > 2 && band_lds_bytes(rows, max_w) > 32 * 1024) rows--;
    if (const char *env = getenv("LSN_RADIAL_BAND_ROWS")) {  // tuning
        const int v = atoi(env);
        if (v >= 1 && band_lds_bytes(v, max_w) <= 160 * 1024) rows = v;
    }
    if (rows > max_h) rows = max_h;
    if (band_lds_bytes(rows, max_w) > 160 * 1024 || (long long)(rows + 2) * max_w > 65535) {
        lsn::set_error("lsnFusionRadialCorrect: a frame of width %d does not fit the closing kernel's LDS band", max_w);
        return -1;
    }
    if (p->band_rows != rows) {
        std::vector<BandDesc> bands;
        p->bands_per_tick = make_bands(p, rows, bands);
        if (p->bands.reserve(sizeof(BandDesc) * bands.size())) return -1;
        LSN_HIP(hipMemcpyAsync(p->bands.p, bands.data(), sizeof(BandDesc) * bands.size(), hipMemcpyHostToDevice, s));
        LSN_HIP(hipStreamSynchronize(s));  // bands is a local
        p->band_rows = rows;
    }
    const long long holes_tick_bytes = (((p->cap + 64ll * (p->n_maps + 1)) / 8) + 31) & ~15ll;
    const size_t cnt_bytes = 3 * sizeof(int) * kCntStride * (size_t)n_tf;
    if (p->holes.reserve((size_t)holes_tick_bytes * p->n_ticks + 64) || p->work.reserve(8 * npix + 64) || p->work2.reserve(4 * npix + 64)) return -1;
    // The second pass leaves every counter cleared -- when it has run to its end.  A call that failed between the band kernel and the
    // per-frame kernel (a launch error), or one that took another closing route after the band kernel, leaves counts behind: the
    // counters are cleared here unless the previous chain is known to have been enqueued completely.
    if (p->work_cnt.bytes < cnt_bytes) {
        if (p->work_cnt.reserve(cnt_bytes)) return -1;
        p->work_cnt_clean = false;
    }
    // (a call on another stream than the previous one's has waited for that chain's end: radial_correct)
    if (!p->work_cnt_clean) LSN_HIP(hipMemsetAsync(p->work_cnt.p, 0, p->work_cnt.bytes, s));
    p->work_cnt_clean = false;
    if (!vec) LSN_HIP(hipMemsetAsync(p->holes.p, 0, (size_t)holes_tick_bytes * p->n_ticks, s));   // the pixel-by-pixel pass only sets bits
    BandArgs ba;
    ba.frames = p->frames.as<FrameDesc>();
    ba.bands = p->bands.as<BandDesc>();
    ba.src = src;
    ba.out_d = static_cast<unsigned short *>(d_depth);
    ba.out_c = static_cast<unsigned char *>(d_colors);
    ba.holes = p->holes.as<unsigned char>();
    ba.work = p->work.as<unsigned int>();
    ba.work_cnt = p->work_cnt.as<int>();
    ba.bands_per_tick = p->bands_per_tick;
    ba.n_frames = p->n_maps;
    ba.rows = rows;
    ba.tick_pix_stride = p->cap;
    ba.holes_tick_bytes = holes_tick_bytes;
    const size_t lds = (size_t)band_lds_bytes(rows, max_w);
    const dim3 bgrid((unsigned)((long long)p->bands_per_tick * p->n_ticks));
    if (!p->band_attr_set) {
        // more than 64 KB of dynamic LDS has to be asked for, once per kernel
        LSN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&radial_band_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        LSN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&radial_band_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        LSN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&radial_band_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        LSN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&radial_band_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        p->band_attr_set = true;
    }
    if (!scratch) {
        if (vec) hipLaunchKernelGGL((radial_band_kernel<true, true>), bgrid, dim3(kBandThreads), lds, s, ba);
        else     hipLaunchKernelGGL((radial_band_kernel<true, false>), bgrid, dim3(kBandThreads), lds, s, ba);
    } else {
        if (vec) hipLaunchKernelGGL((radial_band_kernel<false, true>), bgrid, dim3(kBandThreads), lds, s, ba);
        else     hipLaunchKernelGGL((radial_band_kernel<false, false>), bgrid, dim3(kBandThreads), lds, s, ba);
    }
    // (tick.hip: the other half of a tick batch starts its own band kernel here, beside this half's closing rounds)
    if (p->after_band) LSN_HIP(hipEventRecord(p->after_band, s));
    FixArgs fa;
    fa.frames = ba.frames;
    fa.out_d = ba.out_d;
    fa.out_c = ba.out_c;
    fa.holes = ba.holes;
    fa.work = ba.work;
    fa.work_cnt = ba.work_cnt;
    fa.n_frames = p->n_maps;
    fa.list_cap = tiny_lists ? 8 : kFixList;
    fa.n_tf = n_tf;
    fa.tick_pix_stride = p->cap;
    fa.holes_tick_bytes = holes_tick_bytes;
    // A few frames (a live tick): the first two rounds -- thousands of pixels per frame -- over all frames at once, the tail one workgroup
    // per frame (8 frames: 130 -> 102 us).  A large batch is bound by the scattered lines those rounds touch, not by their latency, and
    // the two extra launches only cost (512 frames: 242 -> 305 us): there the per-frame kernel does it all.
    fa.cnt_index = 0;
    if (n_tf <= 128) {
        fa.cnt_index = 2;
        int *c0 = ba.work_cnt, *c1 = c0 + (size_t)kCntStride * n_tf, *c2 = c1 + (size_t)kCntStride * n_tf;
        const int bpf = 12;
        const dim3 rgrid((unsigned)((long long)n_tf * bpf));
        hipLaunchKernelGGL(close_fix_round_kernel, rgrid, dim3(kThreads), 0, s, fa, (const unsigned int *)p->work.as<unsigned int>(), (const int *)c0, 2, 2, 1,
                           p->work2.as<unsigned int>(), c1, 1, 1, tiny_lists ? (1 << 30) : 1, bpf);
        hipLaunchKernelGGL(close_fix_round_kernel, rgrid, dim3(kThreads), 0, s, fa, (const unsigned int *)p->work2.as<unsigned int>(), (const int *)c1, 1, 1,
                           tiny_lists ? (1 << 30) : 1, p->work.as<unsigned int>(), c2, 2, 2, 1, bpf);
    }
    hipLaunchKernelGGL(close_fix_kernel, dim3((unsigned)n_tf), dim3(kFixThreads), 0, s, fa);
    LSN_HIP(hipGetLastError());
    p->work_cnt_clean = true;
    return 0;
}

// Test hook: how many of the closing chain's work counters are not zero once `stream` has drained.  The chain leaves them all cleared when
// it has run to its end (that is what lets the next call skip its memset); tests/test_radial_gpu.py holds it to that on every closing route.
static int lsnFusionRadialCountersLeft_impl(LsnFusion *p, void *stream)
{
    lsn::clear_error();
    if (!p) return -1;
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    LSN_HIP(hipStreamSynchronize(lsn::as_stream(stream)));
    if (!p->work_cnt.p || p->work_cnt.bytes == 0) return 0;
    std::vector<int> host(p->work_cnt.bytes / sizeof(int));
    LSN_HIP(hipMemcpy(host.data(), p->work_cnt.p, host.size() * sizeof(int), hipMemcpyDeviceToHost));
    int left = 0;
    for (int v : host) left += v != 0;
    return left;
}

extern "C" int lsnFusionRadialCountersLeft(LsnFusion *p, void *stream)
{
    return lsn::guarded<int>("lsnFusionRadialCountersLeft", static_cast<int>(-1), [&]() { return lsnFusionRadialCountersLeft_impl(p, stream); });
}

static int lsnFusionRadialCorrect_impl(LsnFusion *p, const float *intr_params, void *d_depth, void *d_colors, void *stream)
{
    lsn::clear_error();
    if (!p || !intr_params || !d_depth || !d_colors) {
        lsn::set_error("lsnFusionRadialCorrect: null argument");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    return radial_correct(p, intr_params, d_depth, d_colors, d_depth, d_colors, lsn::as_stream(stream));
}

extern "C" int lsnFusionRadialCorrect(LsnFusion *p, const float *intr_params, void *d_depth, void *d_colors, void *stream)
{
    return lsn::guarded<int>("lsnFusionRadialCorrect", static_cast<int>(-1), [&]() { return lsnFusionRadialCorrect_impl(p, intr_params, d_depth, d_colors, stream); });
}

static int lsnFusionRadialCorrectTo_impl(LsnFusion *p, const float *intr_params, const void *d_depth_in, const void *d_colors_in, void *d_depth_out,
                                        void *d_colors_out, void *stream)
{
    lsn::clear_error();
    if (!p || !intr_params || !d_depth_in || !d_colors_in || !d_depth_out || !d_colors_out) {
        lsn::set_error("lsnFusionRadialCorrectTo: null argument");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    return radial_correct(p, intr_params, d_depth_in, d_colors_in, d_depth_out, d_colors_out, lsn::as_stream(stream));
}

extern "C" int lsnFusionRadialCorrectTo(LsnFusion *p, const float *intr_params, const void *d_depth_in, const void *d_colors_in, void *d_depth_out,
                                        void *d_colors_out, void *stream)
{
    return lsn::guarded<int>("lsnFusionRadialCorrectTo", static_cast<int>(-1), [&]() { return lsnFusionRadialCorrectTo_impl(p, intr_params, d_depth_in, d_colors_in, d_depth_out, d_colors_out, stream); });
}
