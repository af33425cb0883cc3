// fusion_shared.hpp -- what the translation units of the fusion plan share: the device-side building blocks of the fused
// unproject + transform + crop + compaction kernels (fusion.hip holds the design notes), the scan kernel, and the host-side
// plan object behind the opaque LsnFusion handle.  Device code lives in an anonymous namespace: every .hip file gets its own copy.
#pragma once

#include "lsn_common.hpp"

#include <mutex>
#include <vector>

namespace lsn {

struct FrameDesc {
    int w, h, npix, tile_start;  // tile_start: first tile of this frame inside its tick
    long long depth_off;         // u16 elements from the tick's depth base
    long long rgb_off;           // bytes from the tick's colour base
    int xtab_off, ytab_off;      // this sensor's rows of the unprojection tables (floats)
    float inv_w;                 // 1 / w, for the (corrected, exact) float division of small pixel offsets
    int pad1;
};

struct TileDesc {  // one per tile of a tick
    int frame;     // sensor-frame the tile belongs to
    int x0, y0;    // column / row of the tile's first pixel (host-computed: no integer division on the device)
    int pad;
};

struct SensorParams {  // 16 floats, wave-uniform -> scalar loads
    float cx, cy, fx, fy;
    float t0, t1, t2;
    float r00, r01, r02, r10, r11, r12, r20, r21, r22;
};

struct FuseArgs {
    const FrameDesc *frames;
    const TileDesc *tiles;             // tile (within tick) -> frame and first-pixel coordinates
    const SensorParams *params;
    const float *xtab;  // [(x - cx) / fx] per sensor column
    const float *ytab;  // [(cy - y) / fy] per sensor row
    const unsigned short *depth;
    const unsigned char *rgb;
    uint4 *out;
    int *tile_counts;                // mode 0: [n_ticks * tiles_per_tick] counts, then exclusive prefixes
    unsigned long long *run_state;   // mode 1: [n_ticks * tiles_per_tick] {flag:2 | value}, indexed by run
    unsigned int *ticket;            // mode 1: per-tick run tickets, 32 words apart
    int *offsets;                    // [n_ticks][n_frames + 1]
    int *pixmap;                     // optional [n_ticks][pixels per tick]: vertex index inside the tick's cloud, -1 = none (rigs the 8-pixel lanes do not fit)
    int *pm_first;                   // ... or its compact form, per lane of 8 pixels: index of the lane's first vertex ...
    unsigned char *pm_mask;          // ... and which of the 8 pixels have one (0.625 instead of 4 bytes per pixel)
    const unsigned short *depth_next;  // streamed mode (MODE 3): the NEXT batch's depth, counted in the shadow of this write
    int *tile_counts_next;             // ... and where its per-tile counts go
    int *error_flag;                 // mode 1: set when a bounded spin gives up (sticky until read)
    const unsigned int *thr;         // optional [pixels per tick]: the depth interval each pixel survives in (thresh_kernel), null = none
    int n_frames;
    int tiles_per_tick;
    int n_ticks;
    int tiles_per_run;               // mode 1
    unsigned int epoch;              // mode 2: tag of this launch's look-back words (run_state is never cleared)
    int chunk;                       // write pass block order: 0 = tick-major; C > 0 = chunks of C consecutive tiles, all ticks of a chunk before the next chunk
    int reverse_ticks;               // the write pass takes the ticks last to first (what the count pass read last is nearest in cache); $LSN_WRITE_FORWARD=1: first to last
    int tile0;                       // one-tick plans only: the launch covers tiles [tile0, tile0 + gridDim.x) of the tick (a group of sensors, run_frames)
    int host_out;                    // mode 2: `out` is pinned host memory (plain, destination-aligned stores; see stage_and_store)
    int *group_end_mirror;           // mode 2, optional (pinned host memory): where this launch's vertices end inside the tick, stored by its last tile
    int *offsets_mirror;             // mode 2, optional: the offset table entries are also stored here (pinned host memory), [n_frames + 1] = give-up flag
    int runs_per_tick;               // mode 1
    long long tick_depth_stride;  // u16 elements
    long long tick_rgb_stride;    // bytes
    long long tick_vert_stride;   // vertices
    float minX, minY, minZ, maxX, maxY, maxZ;
};

// Kernel arguments of the triangulation passes (mesh.hip: tri_kernel<0 / 1>).
struct TriArgs {
    const FrameDesc *frames;
    const TileDesc *tiles;
    const unsigned short *depth;
    const int *pixmap;   // [n_ticks][pixels per tick] (rigs whose widths are not multiples of 8) ...
    const int *pm_first; // ... or [n_ticks][pixels per tick / 8]: the first vertex index of every lane of 8 pixels ...
    const unsigned char *pm_mask;   // ... and the mask of its pixels that have a vertex
    int *tri;            // [n_ticks][tri_cap][3]
    int *tile_counts;    // [n_ticks * tiles_per_tick] counts, then exclusive prefixes (scan_kernel)
    unsigned int *codes; // [n_ticks * tiles_per_tick * 256] per-lane 4-bit-per-pixel triangle codes: count pass -> write pass
    int tiles_per_tick;
    int win;                    // triangles staged per LDS round of the write pass
    int host_out;               // `tri` is pinned host memory: the launch picks the HOST form of the write pass
    int index_base;             // added to every vertex index a triangle names: the tick's vertices start there in the caller's cloud (a call
                                // sharded over devices, host_flows.hip: formMesh's rebase across devices); 0 everywhere else
    long long tick_pix_stride;  // pixels per tick
    long long tick_tri_stride;  // triangles per tick (capacity)
};

}  // namespace lsn
using lsn::FrameDesc;
using lsn::FuseArgs;
using lsn::TriArgs;
using lsn::SensorParams;
using lsn::TileDesc;

namespace {

constexpr int kThreads = 256;
constexpr int kPxPerLane = 8;
constexpr int kTile = kThreads * kPxPerLane;  // 2048 pixels per workgroup step
constexpr int kWin = 1152;                    // survivors staged per LDS round (9/16 of a tile)
// The merged cloud is written once and not read again by the launch sequence: streaming (nt) stores keep the 15 MB per tick
// out of L2 / Infinity Cache, where the depth frames and the threshold table live between the count and the write pass
// (measured: 0.335 -> 0.310 ms per 64-tick step; nt loads of the inputs in the write pass changed nothing).
constexpr bool kNontemporalStores = true;

// (FrameDesc, TileDesc, SensorParams and FuseArgs are declared above, in namespace lsn: they appear in the signatures of
// the host helpers shared between the translation units and therefore need external linkage)


typedef float f2 __attribute__((ext_vector_type(2)));

// Z = float(d) / 1000.0f (depthprocessing.cpp:149-150) for two pixels without the ~13-instruction IEEE division:
// with r = fl32(1/1000) = 0x3a83126f, q0 = d*r, e = fma(-q0, 1000, d), q = fma(e, r, q0) is the correctly rounded
// quotient for EVERY u16 d -- proven exhaustively with exact rational arithmetic in tests/test_fast_division.py.
__device__ __forceinline__ f2 depth_to_metres2(f2 d)
{
    const f2 r = {0x1.0624dep-10f, 0x1.0624dep-10f};
    const f2 k = {1000.0f, 1000.0f};
    const f2 q0 = d * r;
    const f2 e = __builtin_elementwise_fma(-q0, k, d);
    return __builtin_elementwise_fma(e, r, q0);
}

// createVertices' per-pixel arithmetic (depthprocessing.cpp:149-163) on TWO pixels at once, one rounding per
// operation (contraction is off, so a*b+c stays a packed multiply and a packed add: v_pk_mul_f32 / v_pk_add_f32 do two
// f32 lanes' worth per issue slot, which halves the VALU time of this VALU-heavy kernel).
// xfac = (float(x) - cx) / fx and yfac = (cy - float(y)) / fy (:151-152) depend on the column / row only; they come
// from per-sensor tables filled on the device with the same IEEE operations (table_kernel), so the per-pixel work
// has no division left.
__device__ __forceinline__ void unproject2(f2 d, f2 xfac, f2 yfac, const SensorParams &P, f2 &ox, f2 &oy, f2 &oz)
{
    f2 Z = depth_to_metres2(d);
    f2 X = xfac * Z;
    f2 Y = yfac * Z;
    X = X + P.t0;
    Y = Y + P.t1;
    Z = Z + P.t2;
    ox = X * P.r00 + Y * P.r01 + Z * P.r02;
    oy = X * P.r10 + Y * P.r11 + Z * P.r12;
    oz = X * P.r20 + Y * P.r21 + Z * P.r22;
}

// The inclusive AABB test with the reference's own comparisons (:162), so that a NaN coordinate is kept exactly like
// the reference keeps it (non-short-circuit '|': six compares and lane-mask ORs, no divergent branches).
__device__ __forceinline__ bool inside_box(float ox, float oy, float oz, const FuseArgs &a)
{
    const bool rejected = (ox < a.minX) | (ox > a.maxX) | (oy < a.minY) | (oy > a.maxY) | (oz < a.minZ) | (oz > a.maxZ);
    return !rejected;
}

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int n = __shfl_up(v, off, 64);
        if (lane >= off) v += n;
    }
    return v;
}

// Number of survivors among the lower lanes of the wave, and in the whole wave, straight from the keep predicates'
// lane masks: v_mbcnt per mask for the lanes below, s_bcnt1 (SALU) for the total -- no shuffles, no per-lane counters.
__device__ __forceinline__ void rank_from_masks(const bool (&keep)[kPxPerLane], int &below, int &wave_total)
{
    below = 0;
    wave_total = 0;
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) {
        const unsigned long long m = __ballot(keep[k]);
        below = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, below));
        wave_total += __popcll(m);
    }
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- one tile: where it is, its inputs, its arithmetic --------------------------------------------------------

struct Tile {  // wave-uniform (SGPRs)
    long long pix_base;      // index of the frame's first pixel inside its tick (= depth_off)
    int f, w, h, npix, px0;  // frame, its size, first pixel of the tile inside the frame
    int x0, y0;              // that pixel's column / row
    float inv_w;
    bool frame_start;
    const unsigned short *dptr;
    const unsigned char *cptr;
    const float *xt, *yt;
};

__device__ __forceinline__ Tile locate(const FuseArgs &a, int tick, int tile)
{
    Tile t;
    const TileDesc td = a.tiles[tile];
    t.f = td.frame;
    t.x0 = td.x0;
    t.y0 = td.y0;
    const FrameDesc fd = a.frames[t.f];
    t.inv_w = fd.inv_w;
    t.pix_base = fd.depth_off;
    t.w = fd.w;
    t.h = fd.h;
    t.npix = fd.npix;
    t.px0 = (tile - fd.tile_start) * kTile;
    t.frame_start = tile == fd.tile_start;
    t.dptr = a.depth + tick * a.tick_depth_stride + fd.depth_off;
    t.cptr = a.rgb + tick * a.tick_rgb_stride + fd.rgb_off;
    t.xt = a.xtab + fd.xtab_off;
    t.yt = a.ytab + fd.ytab_off;
    return t;
}

struct Inputs {  // one lane's 8 pixels
    unsigned int dw[4];  // 8 x u16 depth
    unsigned int cw[6];  // 8 x RGB8
};

// VEC: every frame has w % 8 == 0 and the buffers are 16-B aligned -> one 16-B depth load, 24 B of colour per lane.
template <bool VEC, bool RGB>
__device__ __forceinline__ void load_inputs(const Tile &t, Inputs &in)
{
    const int p0 = t.px0 + threadIdx.x * kPxPerLane;
#pragma unroll
    for (int i = 0; i < 4; i++) in.dw[i] = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) in.cw[i] = 0;
    if (VEC) {
        if (p0 < t.npix) {
            const uint4 dv = *reinterpret_cast<const uint4 *>(t.dptr + p0);
            in.dw[0] = dv.x; in.dw[1] = dv.y; in.dw[2] = dv.z; in.dw[3] = dv.w;
            if (RGB) {
                const uint2 *cp = reinterpret_cast<const uint2 *>(t.cptr + 3ll * p0);
                const uint2 c0 = cp[0], c1 = cp[1], c2 = cp[2];
                in.cw[0] = c0.x; in.cw[1] = c0.y; in.cw[2] = c1.x; in.cw[3] = c1.y; in.cw[4] = c2.x; in.cw[5] = c2.y;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) {
            if (p0 + k < t.npix) {
                const unsigned int d = t.dptr[p0 + k];
                in.dw[k >> 1] |= d << ((k & 1) * 16);
                if (RGB) {
                    const unsigned char *c = t.cptr + 3ll * (p0 + k);
                    const unsigned int rgb = c[0] | (c[1] << 8) | (c[2] << 16);
                    const int b = 3 * k;  // the pixel's 3 bytes start at byte 3k of the lane's 24-byte group
                    in.cw[b >> 2] |= rgb << ((b & 3) * 8);
                    if ((b & 3) > 1) in.cw[(b >> 2) + 1] |= rgb >> ((4 - (b & 3)) * 8);
                }
            }
        }
    }
}

// The colours alone (the lazy-colour write pass loads them after the keep predicates are known, for the lanes that kept anything).
template <bool VEC>
__device__ __forceinline__ void load_rgb(const Tile &t, Inputs &in)
{
    const int p0 = t.px0 + threadIdx.x * kPxPerLane;
    if (VEC) {
        if (p0 < t.npix) {
            const uint2 *cp = reinterpret_cast<const uint2 *>(t.cptr + 3ll * p0);
            const uint2 c0 = cp[0], c1 = cp[1], c2 = cp[2];
            in.cw[0] = c0.x; in.cw[1] = c0.y; in.cw[2] = c1.x; in.cw[3] = c1.y; in.cw[4] = c2.x; in.cw[5] = c2.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) {
            if (p0 + k < t.npix) {
                const unsigned char *c = t.cptr + 3ll * (p0 + k);
                const unsigned int rgb = c[0] | (c[1] << 8) | (c[2] << 16);
                const int b = 3 * k;
                in.cw[b >> 2] |= rgb << ((b & 3) * 8);
                if ((b & 3) > 1) in.cw[(b >> 2) + 1] |= rgb >> ((4 - (b & 3)) * 8);
            }
        }
    }
}

// {R, G, B, A = 255} of pixel k of the lane (depthprocessing.cpp:1598-1601)
__device__ __forceinline__ unsigned int rgba_of(const Inputs &in, int k)
{
    const int b = 3 * k;
    const unsigned int lo = in.cw[b >> 2];
    const unsigned int hi = in.cw[(b >> 2) + 1 < 6 ? (b >> 2) + 1 : 5];
    return (__funnelshift_r(lo, hi, (b & 3) * 8) & 0x00FFFFFFu) | 0xFF000000u;
}

// Keep predicates (wave-wide lane masks in SGPR pairs) and, when WRITE, the assembled vertices of a lane's 8 pixels.
// Branch-free: a zero depth (invalid pixel, :144, or a lane past the frame end) is computed and then dropped.
// (column, row) of the lane's first pixel: the tile starts at (x0, y0) and the lane is v = x0 + 8*tid < w + 2048 columns
// further; v / w by a float multiply and an exact +-1 correction (v < 2^23, so the estimate is off by at most one).
__device__ __forceinline__ void lane_origin(const Tile &t, int &x, int &y)
{
    const int p0 = t.px0 + threadIdx.x * kPxPerLane;
    const int v = t.x0 + (int)threadIdx.x * kPxPerLane;
    int q = (int)((float)v * t.inv_w);
    x = v - q * t.w;
    if (x < 0) { q--; x += t.w; }
    if (x >= t.w) { q++; x -= t.w; }
    y = t.y0 + q;
    if (p0 >= t.npix) { x = 0; y = 0; }
}

// streaming 16-byte store of one vertex (the merged cloud is written once and not read again by the launch sequence)
__device__ __forceinline__ void store_vertex(uint4 *dst, const uint4 v)
{
    if (kNontemporalStores) {
        __builtin_nontemporal_store(v.x, &dst->x);
        __builtin_nontemporal_store(v.y, &dst->y);
        __builtin_nontemporal_store(v.z, &dst->z);
        __builtin_nontemporal_store(v.w, &dst->w);
    } else {
        *dst = v;
    }
}

// The column / row factors of a lane's 8 pixels (they depend on the tile geometry only, not on the tick or the batch).
template <bool VEC>
__device__ __forceinline__ void tile_factors(const Tile &t, float (&xf)[kPxPerLane], float (&yf)[kPxPerLane])
{
    int x, y;
    lane_origin(t, x, y);
    float yfac = t.yt[y];
    if (VEC) {
        // w % 8 == 0: the lane's 8 pixels share a row and their columns are 8 consecutive, 32-B aligned table entries
        const float4 x0 = *reinterpret_cast<const float4 *>(t.xt + x);
        const float4 x1 = *reinterpret_cast<const float4 *>(t.xt + x + 4);
        xf[0] = x0.x; xf[1] = x0.y; xf[2] = x0.z; xf[3] = x0.w;
        xf[4] = x1.x; xf[5] = x1.y; xf[6] = x1.z; xf[7] = x1.w;
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) yf[k] = yfac;
    } else {
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) {
            xf[k] = t.xt[x];
            yf[k] = yfac;
            // rows may end inside a lane's 8 pixels when w % 8 != 0
            x++;
            if (x == t.w) {
                x = 0;
                y = y + 1 < t.h ? y + 1 : y;
                yfac = t.yt[y];
            }
        }
    }
}

template <bool WRITE>
__device__ __forceinline__ void compute_pixels(const FuseArgs &a, const SensorParams &P, const Inputs &in, const float (&xf)[kPxPerLane],
                                               const float (&yf)[kPxPerLane], bool (&keep)[kPxPerLane], uint4 (&vert)[kPxPerLane])
{
#pragma unroll
    for (int k = 0; k < kPxPerLane; k += 2) {
        const unsigned int d0 = in.dw[k >> 1] & 0xFFFFu, d1 = in.dw[k >> 1] >> 16;
        f2 ox, oy, oz;
        unproject2(f2{(float)d0, (float)d1}, f2{xf[k], xf[k + 1]}, f2{yf[k], yf[k + 1]}, P, ox, oy, oz);
        keep[k] = inside_box(ox.x, oy.x, oz.x, a) && d0 != 0;
        keep[k + 1] = inside_box(ox.y, oy.y, oz.y, a) && d1 != 0;
        if (WRITE) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                vert[k + j].x = rgba_of(in, k + j);  // A = 255 (:1601)
                vert[k + j].y = __float_as_uint(j ? ox.y : ox.x);
                vert[k + j].z = __float_as_uint(j ? oy.y : oy.x);
                vert[k + j].w = __float_as_uint(j ? oz.y : oz.x);
            }
        }
    }
}

template <bool VEC, bool WRITE>
__device__ __forceinline__ void compute_tile(const FuseArgs &a, const Tile &t, const Inputs &in, bool (&keep)[kPxPerLane],
                                             uint4 (&vert)[kPxPerLane])
{
    const SensorParams P = a.params[t.f];
    float xf[kPxPerLane], yf[kPxPerLane];
    tile_factors<VEC>(t, xf, yf);
    compute_pixels<WRITE>(a, P, in, xf, yf, keep, vert);
}

// Where rank q of a staged window lives: q + q / 8 (one 16-byte pad per 8 ranks).  LSN_STAGE_PAD_SHIFT (build-time, A/B only):
// 4 = one pad per 16 ranks, 31 = no padding.
#ifndef LSN_STAGE_PAD_SHIFT
#define LSN_STAGE_PAD_SHIFT 3
#endif
__device__ __forceinline__ int stage_slot16(int q) { return q + (q >> LSN_STAGE_PAD_SHIFT); }
constexpr int kStageSlots = kWin + (kWin >> LSN_STAGE_PAD_SHIFT) + 1;

// Stages a tile's survivors in LDS in rank order, window by window, and copies them out with consecutive lanes writing
// consecutive 16-B vertices.  Rank q of a window lives at slot q + q/8: a lane's 8 consecutive ranks then start 9 slots
// (144 B) apart, which keeps the 16-B LDS writes of neighbouring lanes on different bank groups (stride 128 B is an
// 8-way conflict).  A typical tile (crop + invalid pixels) fits in one window of kWin; the 20.7 KB footprint (instead of
// 36.9 KB for a whole tile) lets 7 workgroups share a CU.  Every thread must call this (barriers inside); rank0 is the
// lane's first rank inside the tile, tile_tot is uniform.  On return the LDS window is free again.
// host_dst: the destination is pinned host memory (the exports' output block, host_flows.hip) -- the stores cross PCIe, where plain stores
// move ~5 % more than streaming ones and a wave whose 1 KB starts on a 1 KB boundary ~3 % more than one that straddles lines
// (tools/link_probe.hip), so the copy-out is shifted to the destination's alignment.
__device__ __forceinline__ void stage_and_store(uint4 *stage, const bool (&keep)[kPxPerLane], const uint4 (&vert)[kPxPerLane], int rank0,
                                                int tile_tot, uint4 *dst, bool host_dst = false)
{
    for (int w0 = 0; w0 < tile_tot; w0 += kWin) {
        int r = rank0 - w0;
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) {
            if (keep[k]) {
                if ((unsigned int)r < (unsigned int)kWin) stage[stage_slot16(r)] = vert[k];
                r++;
            }
        }
        __syncthreads();
        const int n = min(kWin, tile_tot - w0);
        if (host_dst) {
            const int mis = (int)((reinterpret_cast<uintptr_t>(dst + w0) >> 4) & 63);
            for (int i = (int)threadIdx.x - mis; i < n; i += kThreads)
                if (i >= 0) dst[w0 + i] = stage[stage_slot16(i)];
        } else
        for (int i = threadIdx.x; i < n; i += kThreads) {
            const uint4 v = stage[stage_slot16(i)];
            if (kNontemporalStores) {   // written once, never read again by this launch sequence
                __builtin_nontemporal_store(v.x, &dst[w0 + i].x);
                __builtin_nontemporal_store(v.y, &dst[w0 + i].y);
                __builtin_nontemporal_store(v.z, &dst[w0 + i].z);
                __builtin_nontemporal_store(v.w, &dst[w0 + i].w);
            } else {
                dst[w0 + i] = v;
            }
        }
        __syncthreads();
    }
}

// Mode 0, between the count and the write launch: one workgroup per tick turns that tick's tile counts into exclusive
// prefixes in place and fills the per-sensor offset table (offsets[tick][f] = first vertex of sensor f, [n_frames] = total).
// 1024 threads x 8 consecutive counts each: a tick of up to 8192 tiles (16 x 1024x1024) is ONE round -- two 16-byte loads, a serial
// prefix in registers, a wave scan of the lane totals, 16 wave totals through LDS -- where 256 threads x 1 count walked it in 32
// rounds of three barriers each (the config-5 shape has 8 such ticks per step: 8 workgroups on 256 CUs, all of them that slow).
// mirror (optional): the offset table is also stored there -- pinned host memory, so the host has the counts when the stream
// is idle without a copy of its own.
constexpr int kScanThreads = 1024;
constexpr int kScanItems = 8;
constexpr int kScanWaves = kScanThreads / 64;

__device__ __forceinline__ void scan_tick(int *tc, int tiles_per_tick, const FrameDesc *frames, int n_frames, int *off, int *mirror_row,
                                          int (&s_wave)[kScanWaves])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;   // uniform: every thread adds the same round totals
    for (int c0 = 0; c0 < tiles_per_tick; c0 += kScanThreads * kScanItems) {
        const int i0 = c0 + (int)threadIdx.x * kScanItems;
        int v[kScanItems];
        const bool whole = i0 + kScanItems <= tiles_per_tick && (reinterpret_cast<uintptr_t>(tc + i0) & 15) == 0;
        if (whole) {
            const int4 a = reinterpret_cast<const int4 *>(tc + i0)[0], b = reinterpret_cast<const int4 *>(tc + i0)[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < kScanItems; k++) v[k] = i0 + k < tiles_per_tick ? tc[i0 + k] : 0;
        }
        int t = 0;
#pragma unroll
        for (int k = 0; k < kScanItems; k++) {   // v[k] becomes the exclusive prefix inside the lane
            const int x = v[k];
            v[k] = t;
            t += x;
        }
        const int incl = wave_inclusive_scan(t, lane);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int pre = carry, round_tot = 0;
#pragma unroll
        for (int w = 0; w < kScanWaves; w++) {
            const int x = s_wave[w];
            if (w < wave) pre += x;
            round_tot += x;
        }
        pre += incl - t;
        if (whole) {
            reinterpret_cast<int4 *>(tc + i0)[0] = make_int4(pre + v[0], pre + v[1], pre + v[2], pre + v[3]);
            reinterpret_cast<int4 *>(tc + i0)[1] = make_int4(pre + v[4], pre + v[5], pre + v[6], pre + v[7]);
        } else {
#pragma unroll
            for (int k = 0; k < kScanItems; k++)
                if (i0 + k < tiles_per_tick) tc[i0 + k] = pre + v[k];
        }
        carry += round_tot;
        __syncthreads();   // s_wave is rewritten by the next round; the prefixes above are visible to the workgroup below
    }
    // frames are few: thread f looks up the prefix at its first tile (written above by this workgroup)
    for (int f = threadIdx.x; f <= n_frames; f += kScanThreads) {
        const int v = f < n_frames ? tc[frames[f].tile_start] : carry;
        off[f] = v;
        if (mirror_row) mirror_row[f] = v;
    }
}

// Launch with dim3(n_ticks) x dim3(kScanThreads).
__attribute__((unused)) __global__ __launch_bounds__(kScanThreads) void scan_kernel(int *tile_counts, int tiles_per_tick, const FrameDesc *frames, int n_frames,
                                                        int *offsets, int *mirror)
{
    __shared__ int s_wave[kScanWaves];
    const int tick = blockIdx.x;
    scan_tick(tile_counts + (long long)tick * tiles_per_tick, tiles_per_tick, frames, n_frames, offsets + (long long)tick * (n_frames + 1),
                     mirror ? mirror + (long long)tick * (n_frames + 1) : nullptr, s_wave);
}


}  // namespace

// -------------------------------------------------------------------------------------------------------------
// host side: the plan
// -------------------------------------------------------------------------------------------------------------

struct LsnFusion {
    int device = 0;
    int n_ticks = 0, n_maps = 0;
    std::vector<int> w, h;
    std::vector<int> tile_start;  // first tile of every frame inside a tick, [n_maps] = tiles_per_tick
    long long cap = 0;  // vertices per tick
    long long tick_depth_elems = 0, tick_rgb_bytes = 0;
    int tiles_per_tick = 0;
    bool vec_ok = false;
    bool params_set = false;
    int mode = 0;
    unsigned int epoch = 0;          // mode 2: launches so far (tags the look-back words)
    int tiles_per_run_override = 0;  // $LSN_TILES_PER_RUN (tuning / tests)
    bool want_pixmap = false;        // the run in progress also fills the pixel -> vertex map (set and cleared under mu by run_locked)
    float bounds[6] = {0, 0, 0, 0, 0, 0};
    lsn::DevBuf frames, tile_frame, params, tile_counts, tile_state, misc;  // misc: error flag (word 0) + tickets
    lsn::DevBuf xtab, ytab;
    lsn::DevBuf pixmap, pm_first, pm_mask, tri_counts, tri_codes;  // triangulation scratch, allocated on first use
    bool pixmap_compact = false;     // which form of the pixel -> vertex map the last run wrote
    lsn::DevBuf winner, map_copy, colors_copy, radial;  // radial-correction scratch, allocated on first use
    lsn::DevBuf cand;                                   // [pixels per tick][4] warp candidates of the current intrinsics
    lsn::DevBuf ctab;                                   // [pixels per tick] their compact form (one dword per destination)
    lsn::DevBuf bands, holes, work, work2, work_cnt;    // hole closing: band list, hole bitmap, per-frame work lists (two, used in turn) and their counters
    int band_rows = 0, bands_per_tick = 0;              // what `bands` was built for
    hipStream_t work_cnt_stream = nullptr;              // the stream of the last radial call ...
    hipEvent_t radial_done = nullptr;                   // ... and the end of its chain: a call on another stream waits for it (radial.hip radial_correct)
    bool radial_chain_open = false;                     // radial_done has been recorded at least once
    hipEvent_t after_band = nullptr;                    // not owned: recorded behind the band kernel of a radial call (set by lsnTickRun around its calls)
    bool work_cnt_clean = false;                        // the closing chain of the last call was enqueued to its end (it leaves work_cnt zeroed)
    bool band_attr_set = false;
    std::vector<float> radial_intr;                     // the intrinsics `cand` was built for
    bool cand_valid = false, cand_overflow = false;
    // pipelined mode: the count + scan of call k+1 run on a side stream while the write kernel of call k is still busy
    bool pipelined = false;
    hipStream_t side = nullptr;
    hipEvent_t ev_counted = nullptr, ev_written[2] = {nullptr, nullptr};
    lsn::DevBuf tile_counts_b, offs_int;  // second count buffer, internal offsets [2][n_ticks][n_maps+1]
    unsigned long long calls = 0;
    // streamed mode: which batch the "other" half of the count scratch was counted for
    const void *counted_for = nullptr;
    unsigned long long counted_gen = 0, params_gen = 1;
    int stream_half = 0;
    // per-pixel depth thresholds (thresh_kernel): built once the same parameters are used for a second run
    lsn::DevBuf thr;
    bool thr_valid = false;
    bool thr_enabled = true;             // $LSN_NO_THRESHOLDS=1 keeps the arithmetic count pass (ablation / tests)
    bool one_tick_single_pass = false;   // a one-tick plan of <= 2048 tiles takes the single pass (fuse_kernel<4>) instead of count -> scan -> write; $LSN_ONE_TICK_SINGLE_PASS=0 / 1 forces
    bool lazy_rgb = true;                // the write pass loads colours only where a lane kept a pixel; $LSN_LAZY_RGB=0 loads them with the depth (ablation)
    int runs_with_params = 0;
    std::vector<float> last_intr, last_wt;
    float thr_build_ms = 0;
    // dominant-kernel timing
    bool profile = false;
    int profile_every = 1;               // ... of every launch, or of every n-th (lsnFusionProfile(plan, n)): two event records cost ~2 us each on the stream
    unsigned long long profile_seq = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t ev_used = 0;
    double acc_ms = 0;
    long long launches = 0;
    const char *timed_kernel = nullptr;  // which kernel the event pairs bracket (set by the entry point that records them)
    std::mutex mu;
};

namespace lsn {
// Kernel arguments of one call (everything but the per-mode scratch selection).
void fill_args(LsnFusion *p, FuseArgs &a, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets);
// Called at the top of every run (p->mu held): builds the per-pixel depth thresholds on the second run with unchanged parameters.
int ensure_thresholds(LsnFusion *p, hipStream_t s);
// The count pass of one batch into a.tile_counts: from the thresholds when they exist, else arithmetically.
void launch_count(LsnFusion *p, bool vec, hipStream_t s, const FuseArgs &a);
// Next HIP-event pair of the dominant-kernel timer (profiling on).
int next_event_pair(LsnFusion *p, hipEvent_t &e0, hipEvent_t &e1);
// whether the dominant kernel of the launch sequence being queued is timed (profiling on, and this launch's turn)
inline bool timed_launch(LsnFusion *p) { return p->profile && (p->profile_every <= 1 || p->profile_seq++ % (unsigned long long)p->profile_every == 0); }
}  // namespace lsn
using lsn::ensure_thresholds;
using lsn::fill_args;
using lsn::launch_count;
using lsn::next_event_pair;
using lsn::timed_launch;
