// fusion.hip -- fused depth-unproject + R(p+t) + AABB crop + raster-order compaction + AoS pack for gfx950.
//
// Replaces, for N sensors x T ticks in one launch sequence, the reference's
//   createVertices                 src/NativeUtils/depthprocessing.cpp:122-187  (+ RotatePoint :109-120)
//   generateVerticesFromDepthMaps  :708-733   (std::thread per sensor -> one grid over every pixel of every sensor)
//   formMesh (vertex part)         :1578-1608 (SoA -> VertexC4ubV3f repack + concatenation in sensor order)
//
// Arithmetic contract (bit-exact with the reference, verified against oracle/): every f32 operation is rounded
// on its own in the reference's order -- this file MUST be compiled with -ffp-contract=off and without any
// fast-math flag.  The only fused operations are the two explicit FMAs of the exact d/1000 sequence.
//
// Data layout in HBM (all little-endian, as LiveScanServer packs it, KinectServer.cs:453-498):
//   depth   [tick][sensor][h][w] u16       -- 2 B / pixel, read with one 16-B load per lane (8 pixels)
//   colours [tick][sensor][h][w][3] u8     -- 3 B / pixel, read with 24 B per lane (8 pixels)
//   cloud   [tick][capacity] {u8 R,G,B,A; f32 X,Y,Z} -- 16 B / surviving vertex, written as whole 16-B lanes,
//                                             sensor-major then raster order (formMesh order)
// Algorithmic bytes per sensor-frame: 2 P + 19 V  (P pixels, V survivors).  MFMA has no role here (no contraction;
// it would also change the rounding); the per-sensor pose/intrinsics are wave-uniform and live in SGPRs (scalar
// loads), LDS is used for what it is needed for: turning the per-lane compaction into fully coalesced 16-B stores.
//
// Measured character on MI355X (profiles/): the per-pixel arithmetic is NOT free -- ~35-45 VALU instructions per
// pixel at ~4 cycles per wave instruction make the count pass VALU-bound and the write pass ~50 % VALU-busy -- so the
// kernels are built to (1) spend as few VALU issue slots as possible (no divisions, packed f32 pairs, SALU lane-mask
// logic) and (2) keep VALU-heavy and HBM-heavy work resident on a CU at the same time.
//
// Compaction: a 256-thread workgroup owns tiles of 2048 consecutive pixels of one sensor-frame (8 per lane, so all
// global loads are wide and coalesced).  Lanes count their survivors, a wave scan (cross-lane shuffles) and a 4-entry
// LDS exchange give every survivor its rank inside the tile, survivors are staged in LDS in rank order, and the tile
// is copied out with consecutive lanes writing consecutive vertices.  Two ways to get a tile's offset in its tick:
//   mode 0  count kernel -> scan kernel (one workgroup per tick) -> write kernel            (three launches)
//   mode 1  ONE launch: a workgroup owns a RUN of consecutive tiles of one tick; it first counts the run (depth only),
//           publishes the run's aggregate, resolves its base with a decoupled look-back over the runs of its tick,
//           then recomputes and writes the run.  Counting (VALU) and writing (HBM) phases of different workgroups
//           overlap on every CU, the second depth read comes from cache, and there is one look-back per run, not per tile.
#include "lsn_common.hpp"

#include <mutex>
#include <vector>

namespace {

constexpr int kThreads = 256;
constexpr int kPxPerLane = 8;
constexpr int kTile = kThreads * kPxPerLane;  // 2048 pixels per workgroup step
constexpr int kWin = 1152;                    // survivors staged per LDS round (9/16 of a tile)
// The merged cloud is written once and not read again by the launch sequence: streaming (nt) stores keep the 15 MB per tick
// out of L2 / Infinity Cache, where the depth frames and the threshold table live between the count and the write pass
// (measured: 0.335 -> 0.310 ms per 64-tick step; nt loads of the inputs in the write pass changed nothing).
constexpr bool kNontemporalStores = true;

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
    int *pixmap;                     // optional [n_ticks][pixels per tick]: vertex index inside the tick's cloud, -1 = none
    const unsigned short *depth_next;  // streamed mode (MODE 3): the NEXT batch's depth, counted in the shadow of this write
    int *tile_counts_next;             // ... and where its per-tile counts go
    int *error_flag;                 // mode 1: set when a bounded spin gives up (sticky until read)
    const unsigned int *thr;         // optional [pixels per tick]: the depth interval each pixel survives in (thresh_kernel), null = none
    int n_frames;
    int tiles_per_tick;
    int n_ticks;
    int tiles_per_run;               // mode 1
    int runs_per_tick;               // mode 1
    long long tick_depth_stride;  // u16 elements
    long long tick_rgb_stride;    // bytes
    long long tick_vert_stride;   // vertices
    float minX, minY, minZ, maxX, maxY, maxZ;
};

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

// Keep predicates (wave-wide lane masks in SGPR pairs) and, when WRITE, the assembled vertices of a lane's 8 pixels.
// Branch-free: a zero depth (invalid pixel, :144, or a lane past the frame end) is computed and then dropped.
// The column / row factors of a lane's 8 pixels (they depend on the tile geometry only, not on the tick or the batch).
template <bool VEC>
__device__ __forceinline__ void tile_factors(const Tile &t, float (&xf)[kPxPerLane], float (&yf)[kPxPerLane])
{
    const int p0 = t.px0 + threadIdx.x * kPxPerLane;
    const bool in_frame = p0 < t.npix;
    // (x, y) of the lane's first pixel: the tile starts at (x0, y0) and the lane is v = x0 + 8*tid < w + 2048 columns
    // further; v / w by a float multiply and an exact +-1 correction (v < 2^23, so the estimate is off by at most one).
    const int v = t.x0 + (int)threadIdx.x * kPxPerLane;
    int q = (int)((float)v * t.inv_w);
    int x = v - q * t.w;
    if (x < 0) { q--; x += t.w; }
    if (x >= t.w) { q++; x -= t.w; }
    int y = t.y0 + q;
    if (!in_frame) { x = 0; y = 0; }
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
                const int b = 3 * (k + j);
                const unsigned int lo = in.cw[b >> 2];
                const unsigned int hi = in.cw[(b >> 2) + 1 < 6 ? (b >> 2) + 1 : 5];
                vert[k + j].x = (__funnelshift_r(lo, hi, (b & 3) * 8) & 0x00FFFFFFu) | 0xFF000000u;  // A = 255 (:1601)
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

// Stages a tile's survivors in LDS in rank order, window by window, and copies them out with consecutive lanes writing
// consecutive 16-B vertices.  Rank q of a window lives at slot q + q/8: a lane's 8 consecutive ranks then start 9 slots
// (144 B) apart, which keeps the 16-B LDS writes of neighbouring lanes on different bank groups (stride 128 B is an
// 8-way conflict).  A typical tile (crop + invalid pixels) fits in one window of kWin; the 20.7 KB footprint (instead of
// 36.9 KB for a whole tile) lets 7 workgroups share a CU.  Every thread must call this (barriers inside); rank0 is the
// lane's first rank inside the tile, tile_tot is uniform.  On return the LDS window is free again.
__device__ __forceinline__ void stage_and_store(uint4 *stage, const bool (&keep)[kPxPerLane], const uint4 (&vert)[kPxPerLane], int rank0,
                                                int tile_tot, uint4 *dst)
{
    for (int w0 = 0; w0 < tile_tot; w0 += kWin) {
        int r = rank0 - w0;
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) {
            if (keep[k]) {
                if ((unsigned int)r < (unsigned int)kWin) stage[r + (r >> 3)] = vert[k];
                r++;
            }
        }
        __syncthreads();
        const int n = min(kWin, tile_tot - w0);
        for (int i = threadIdx.x; i < n; i += kThreads) {
            const uint4 v = stage[i + (i >> 3)];
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

// ---- per-pixel depth thresholds: the count pass without the arithmetic ------------------------------------------------------
// With the calibration fixed (it only changes when the user recalibrates) a pixel's fate depends on its depth value alone.
// thresh_kernel finds, per pixel, the set of d in [1, 65535] the reference keeps and stores it as {lo, count} when it is one
// interval (it is, for every sane calibration); the count pass then is "(d - lo) < count" -- 3 VALU instructions per
// pixel instead of ~38 -- and becomes what it should be: a depth-only, HBM-bound sweep.
//
// Exactness.  Every f32 operation of the reference's pipeline is monotone in each operand (rounding to nearest is
// monotone), so evaluating the SAME operations on the two ends of a depth range [a, b] (ordinary interval arithmetic, but
// with the pipeline's own roundings) encloses the result of every d in the range.  A range whose enclosure lies inside the
// box is kept as a whole, one whose enclosure is beyond a face on some axis is rejected as a whole, anything else is
// bisected down to single depths, which are evaluated exactly like the write pass does.  No analytic error bound is
// involved.  Pixels whose survivors are not one interval, or whose parameters are not finite, get lo = 0 and are evaluated
// arithmetically by the count pass as before.
struct Iv { float lo, hi; };
__device__ __forceinline__ Iv iv_scale(Iv a, float c) { const float x = a.lo * c, y = a.hi * c; return c >= 0.0f ? Iv{x, y} : Iv{y, x}; }
__device__ __forceinline__ Iv iv_shift(Iv a, float c) { return Iv{a.lo + c, a.hi + c}; }
__device__ __forceinline__ Iv iv_sum(Iv a, Iv b) { return Iv{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ bool iv_finite(Iv a) { return isfinite(a.lo) && isfinite(a.hi); }

__device__ __forceinline__ float depth_to_metres1(float d)
{
    const f2 z = depth_to_metres2(f2{d, d});
    return z.x;
}

// 0 = every depth in [a, b] is rejected, 1 = every depth is kept, 2 = undecided
__device__ __forceinline__ int classify_range(int a, int b, float xfac, float yfac, const SensorParams &P, const float (&bx)[6])
{
    if (a == b) {
        f2 ox, oy, oz;
        unproject2(f2{(float)a, (float)a}, f2{xfac, xfac}, f2{yfac, yfac}, P, ox, oy, oz);
        const bool rejected = (ox.x < bx[0]) | (ox.x > bx[3]) | (oy.x < bx[1]) | (oy.x > bx[4]) | (oz.x < bx[2]) | (oz.x > bx[5]);
        return rejected ? 0 : 1;
    }
    const Iv z = {depth_to_metres1((float)a), depth_to_metres1((float)b)};
    const Iv X = iv_shift(iv_scale(z, xfac), P.t0), Y = iv_shift(iv_scale(z, yfac), P.t1), Z = iv_shift(z, P.t2);
    const Iv ox = iv_sum(iv_sum(iv_scale(X, P.r00), iv_scale(Y, P.r01)), iv_scale(Z, P.r02));
    const Iv oy = iv_sum(iv_sum(iv_scale(X, P.r10), iv_scale(Y, P.r11)), iv_scale(Z, P.r12));
    const Iv oz = iv_sum(iv_sum(iv_scale(X, P.r20), iv_scale(Y, P.r21)), iv_scale(Z, P.r22));
    if (!(iv_finite(ox) && iv_finite(oy) && iv_finite(oz))) return 2;
    if (ox.hi < bx[0] || ox.lo > bx[3] || oy.hi < bx[1] || oy.lo > bx[4] || oz.hi < bx[2] || oz.lo > bx[5]) return 0;
    if (ox.lo >= bx[0] && ox.hi <= bx[3] && oy.lo >= bx[1] && oy.hi <= bx[4] && oz.lo >= bx[2] && oz.hi <= bx[5]) return 1;
    return 2;
}

__global__ __launch_bounds__(kThreads) void thresh_kernel(const FrameDesc *frames, const SensorParams *params, int n_frames, const float *xtab,
                                                          const float *ytab, unsigned int *thr, float minX, float minY, float minZ, float maxX,
                                                          float maxY, float maxZ)
{
    const int f = blockIdx.y;
    if (f >= n_frames) return;
    const FrameDesc fd = frames[f];
    const SensorParams P = params[f];
    const float bx[6] = {minX, minY, minZ, maxX, maxY, maxZ};
    const bool params_ok = isfinite(P.t0) && isfinite(P.t1) && isfinite(P.t2) && isfinite(P.r00) && isfinite(P.r01) && isfinite(P.r02) &&
                           isfinite(P.r10) && isfinite(P.r11) && isfinite(P.r12) && isfinite(P.r20) && isfinite(P.r21) && isfinite(P.r22) &&
                           !(isnan(minX) || isnan(minY) || isnan(minZ) || isnan(maxX) || isnan(maxY) || isnan(maxZ));
    for (int p = blockIdx.x * kThreads + threadIdx.x; p < fd.npix; p += gridDim.x * kThreads) {
        const int y = p / fd.w, x = p - y * fd.w;
        const float xfac = xtab[fd.xtab_off + x], yfac = ytab[fd.ytab_off + y];
        unsigned int code = 0;                                  // lo = 0: "evaluate arithmetically"
        if (params_ok && isfinite(xfac) && isfinite(yfac)) {
            int state = 0, lo = 1, hi = 0;                      // 0 = before the interval, 1 = inside, 2 = after, 3 = not an interval
            int d = 1;
            while (d <= 65535 && state != 3) {
                int len = d & -d;                               // largest aligned power-of-two block that starts at d
                while (d + len - 1 > 65535) len >>= 1;
                int c = classify_range(d, d + len - 1, xfac, yfac, P, bx);
                while (c == 2) {                                // len == 1 always decides
                    len >>= 1;
                    c = classify_range(d, d + len - 1, xfac, yfac, P, bx);
                }
                if (c == 1) {
                    if (state == 0) { lo = d; state = 1; }
                    else if (state == 2) state = 3;
                    hi = d + len - 1;
                } else if (state == 1) {
                    state = 2;
                }
                d += len;
            }
            if (state != 3) code = (unsigned int)lo | ((unsigned int)(hi - lo + 1) << 16);   // empty: lo = 1, count = 0
        }
        thr[fd.depth_off + p] = code;
    }
}

// Thresholds of a lane's 8 pixels; `flagged` = lanes/pixels that must be evaluated arithmetically.
template <bool VEC>
__device__ __forceinline__ void load_thresholds(const FuseArgs &a, const Tile &t, unsigned int (&lo)[kPxPerLane], unsigned int (&cnt)[kPxPerLane],
                                                bool &any_flagged)
{
    const int p0 = t.px0 + threadIdx.x * kPxPerLane;
    unsigned int c[kPxPerLane];
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) c[k] = 1u;             // past the frame end: lo = 1, count = 0 -> never kept
    const unsigned int *tp = a.thr + t.pix_base + p0;
    if (VEC) {
        if (p0 < t.npix) {
            const uint4 u = reinterpret_cast<const uint4 *>(tp)[0], v = reinterpret_cast<const uint4 *>(tp)[1];
            c[0] = u.x; c[1] = u.y; c[2] = u.z; c[3] = u.w; c[4] = v.x; c[5] = v.y; c[6] = v.z; c[7] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++)
            if (p0 + k < t.npix) c[k] = tp[k];
    }
    bool fl = false;
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) {
        lo[k] = c[k] & 0xFFFFu;
        cnt[k] = c[k] >> 16;
        fl |= lo[k] == 0;
    }
    any_flagged = __any(fl);
}

// Survivors of a lane's 8 pixels among the wave, from the thresholds (flagged pixels: from the arithmetic, keep_exact).
__device__ __forceinline__ int wave_count_thr(const Inputs &in, const unsigned int (&lo)[kPxPerLane], const unsigned int (&cnt)[kPxPerLane])
{
    int wt = 0;
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) {
        const unsigned int d = (k & 1) ? in.dw[k >> 1] >> 16 : in.dw[k >> 1] & 0xFFFFu;
        wt += __popcll(__ballot(d - lo[k] < cnt[k]));
    }
    return wt;
}

// Count pass from the thresholds: a workgroup owns one tile position for kTickGroup consecutive ticks, so the thresholds
// are loaded once per 8 ticks and the 8 depth loads of a lane are in flight together.
template <bool VEC, int kTickGroup>
__global__ __launch_bounds__(kThreads) void count_thr_kernel(const FuseArgs a)
{
    __shared__ int s_cnt[4][kTickGroup];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tg = blockIdx.x / a.tiles_per_tick;
    const int tile = blockIdx.x - tg * a.tiles_per_tick;
    const int tick0 = tg * kTickGroup;
    const int nt = min(kTickGroup, a.n_ticks - tick0);
    const Tile t = locate(a, tick0, tile);
    unsigned int lo[kPxPerLane], cnt[kPxPerLane];
    bool any_flagged;
    load_thresholds<VEC>(a, t, lo, cnt, any_flagged);
    Inputs in[kTickGroup];
#pragma unroll
    for (int i = 0; i < kTickGroup; i++) {
        Tile ti = t;
        ti.dptr = t.dptr + (long long)(i < nt ? i : 0) * a.tick_depth_stride;
        load_inputs<VEC, false>(ti, in[i]);
    }
    if (!any_flagged) {
#pragma unroll
        for (int i = 0; i < kTickGroup; i++) {
            const int wt = wave_count_thr(in[i], lo, cnt);
            if (lane == 0) s_cnt[wave][i] = wt;
        }
    } else {
        // some pixel of this wave has no interval (non-finite calibration, ...): the arithmetic decides, as in fuse_kernel<0>
        const SensorParams P = a.params[t.f];
        float xf[kPxPerLane], yf[kPxPerLane];
        tile_factors<VEC>(t, xf, yf);
#pragma unroll 1
        for (int i = 0; i < nt; i++) {
            // reloaded here on purpose: indexing in[] with a run-time i would move the whole array to scratch memory
            Tile ti = t;
            ti.dptr = t.dptr + (long long)i * a.tick_depth_stride;
            Inputs ix;
            load_inputs<VEC, false>(ti, ix);
            bool keep[kPxPerLane];
            uint4 unused[kPxPerLane];
            compute_pixels<false>(a, P, ix, xf, yf, keep, unused);
            int wt = 0;
#pragma unroll
            for (int k = 0; k < kPxPerLane; k++) wt += __popcll(__ballot(keep[k]));
            if (lane == 0) s_cnt[wave][i] = wt;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nt)
        a.tile_counts[(long long)(tick0 + threadIdx.x) * a.tiles_per_tick + tile] =
            s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
}

// ---- mode 0: count kernel, scan kernel, write kernel ------------------------------------------------------------

// MODE 0 = count only (writes tile_counts), 1 = write with offsets from tile_counts (exclusive prefixes by then),
// 3 = streamed: mode 1 for this batch AND the count of the same tile of the NEXT batch (depth_next) in one workgroup --
// the count pass is VALU-bound, the write pass HBM-bound, and inside one kernel they share every CU all the time.
template <int MODE, bool VEC>
__global__ __launch_bounds__(kThreads) void fuse_kernel(const FuseArgs a)
{
    constexpr bool kWrite = MODE != 0;
    __shared__ uint4 stage[kWrite ? (kWin + kWin / 8) : 1];
    __shared__ int s_wave_tot[4];
    __shared__ int s_wave_next[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tick = blockIdx.x / a.tiles_per_tick;
    const int tile = blockIdx.x - tick * a.tiles_per_tick;

    const Tile t = locate(a, tick, tile);
    Inputs in;
    load_inputs<VEC, kWrite>(t, in);
    Inputs nx;
    if (MODE == 3) {
        // the next batch's tile: same geometry, other depth buffer; its loads fly together with this tile's
        Tile tn = t;
        tn.dptr = a.depth_next + tick * a.tick_depth_stride + t.pix_base;
        load_inputs<VEC, false>(tn, nx);
    }
    const SensorParams P = a.params[t.f];
    float xf[kPxPerLane], yf[kPxPerLane];
    tile_factors<VEC>(t, xf, yf);
    bool keep[kPxPerLane];
    uint4 vert[kPxPerLane];
    compute_pixels<kWrite>(a, P, in, xf, yf, keep, vert);
    if (MODE == 3) {
        int wt = 0;
        bool arithmetic = true;
        if (a.thr) {
            // the next batch's survivors straight from the per-pixel depth thresholds: no arithmetic at all
            unsigned int lo[kPxPerLane], cnt[kPxPerLane];
            bool any_flagged;
            load_thresholds<VEC>(a, t, lo, cnt, any_flagged);
            arithmetic = any_flagged;
            if (!any_flagged) wt = wave_count_thr(nx, lo, cnt);
        }
        if (arithmetic) {
            // same sensor, same tile geometry: the column / row factors and the pose are shared with this batch's tile
            bool keep_n[kPxPerLane];
            uint4 unused[kPxPerLane];
            compute_pixels<false>(a, P, nx, xf, yf, keep_n, unused);
#pragma unroll
            for (int k = 0; k < kPxPerLane; k++) wt += __popcll(__ballot(keep_n[k]));
        }
        if (lane == 0) s_wave_next[wave] = wt;
    }

    int below, wave_total;
    rank_from_masks(keep, below, wave_total);
    if (lane == 0) s_wave_tot[wave] = wave_total;
    int base = 0;
    if (MODE == 1 || MODE == 3) base = a.tile_counts[blockIdx.x];  // scan_kernel left the exclusive prefix inside the tick here
    __syncthreads();
    int wave_off = 0, tile_tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = s_wave_tot[i];
        if (i < wave) wave_off += v;
        tile_tot += v;
    }
    if (MODE == 0) {
        if (threadIdx.x == 0) a.tile_counts[blockIdx.x] = tile_tot;
        return;
    }
    if (MODE == 3 && threadIdx.x == 0) a.tile_counts_next[blockIdx.x] = s_wave_next[0] + s_wave_next[1] + s_wave_next[2] + s_wave_next[3];
    if (a.pixmap) {
        // depth_to_vertices_map (depthprocessing.cpp:166), already rebased to the tick's merged cloud like formMesh
        // rebases triangle indices (:1614-1626): what the triangulation pass reads
        const int p0 = t.px0 + threadIdx.x * kPxPerLane;
        if (p0 < t.npix) {
            int *pm = a.pixmap + tick * a.tick_depth_stride + t.pix_base + p0;
            int r = base + wave_off + below;
            int v[kPxPerLane];
#pragma unroll
            for (int k = 0; k < kPxPerLane; k++) {
                v[k] = keep[k] ? r : -1;
                r += keep[k] ? 1 : 0;
            }
            if (VEC) {
                reinterpret_cast<int4 *>(pm)[0] = make_int4(v[0], v[1], v[2], v[3]);
                reinterpret_cast<int4 *>(pm)[1] = make_int4(v[4], v[5], v[6], v[7]);
            } else {
#pragma unroll
                for (int k = 0; k < kPxPerLane; k++)
                    if (p0 + k < t.npix) pm[k] = v[k];
            }
        }
    }
    stage_and_store(stage, keep, vert, wave_off + below, tile_tot, a.out + tick * a.tick_vert_stride + base);
}

// Mode 0, between the count and the write launch: one workgroup per tick turns that tick's tile counts into exclusive
// prefixes in place and fills the per-sensor offset table (offsets[tick][f] = first vertex of sensor f, [n_frames] = total).
__global__ __launch_bounds__(kThreads) void scan_kernel(int *tile_counts, int tiles_per_tick, const FrameDesc *frames, int n_frames,
                                                        int *offsets)
{
    __shared__ int s_wave[4];
    __shared__ int s_carry;
    const int tick = blockIdx.x;
    int *tc = tile_counts + (long long)tick * tiles_per_tick;
    int *off = offsets + (long long)tick * (n_frames + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < tiles_per_tick; c0 += kThreads) {
        const int i = c0 + threadIdx.x;
        const int v = i < tiles_per_tick ? tc[i] : 0;
        const int incl = wave_inclusive_scan(v, lane);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int pre = s_carry;
        for (int w = 0; w < wave; w++) pre += s_wave[w];
        if (i < tiles_per_tick) tc[i] = pre + incl - v;
        __syncthreads();
        if (threadIdx.x == kThreads - 1) s_carry = pre + incl;
        __syncthreads();
    }
    // frames are few: thread f looks up the prefix at its first tile (written above by this workgroup)
    for (int f = threadIdx.x; f <= n_frames; f += kThreads) off[f] = f < n_frames ? tc[frames[f].tile_start] : s_carry;
}


// ---- mode 1: one launch, runs of tiles with a decoupled look-back per run ---------------------------------------

// Look-back status word: bits 63..62 flag (0 = empty, 1 = run aggregate, 2 = inclusive prefix), low 32 bits value.
// One naturally aligned 8-byte word carries flag AND value, written by one agent-scope (sc1, write-through) store and
// polled with agent-scope loads: nothing else is handed off, so no fence is needed and the result cannot depend on
// where the producing workgroup ran.
constexpr unsigned long long kFlagAggregate = 1ull << 62;
constexpr unsigned long long kFlagPrefix = 2ull << 62;
constexpr int kSpinLimit = 1 << 20;

template <bool VEC>
__global__ __launch_bounds__(kThreads) void run_kernel(const FuseArgs a)
{
    __shared__ uint4 stage[kWin + kWin / 8];
    __shared__ int s_wave_tot[4];
    __shared__ int s_run;
    __shared__ int s_base;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // Every tick is its own compaction domain with its own ticket counter (128 B apart: one returning atomic on a
    // single word saturates near 88 tickets/us).  Workgroup b serves tick b % n_ticks and draws that tick's next run:
    // the runs it will look back at were drawn earlier, so they are running or done, and none of them waits for
    // anything before publishing its aggregate -- no deadlock whatever the dispatch order.
    const int tick = blockIdx.x % a.n_ticks;
    if (threadIdx.x == 0) s_run = (int)atomicAdd(a.ticket + 32 * tick, 1u);
    __syncthreads();
    const int run = s_run;
    if (run >= a.runs_per_tick) return;  // cannot happen with grid = n_ticks * runs_per_tick; keeps a bad launch harmless
    const int t0 = run * a.tiles_per_run;
    const int t1 = min(t0 + a.tiles_per_run, a.tiles_per_tick);

    // ---- phase 1: count the run (depth only; the next tile's depth is in flight while this one is evaluated) -------
    int wave_total = 0;
    {
        Tile t = locate(a, tick, t0);
        Inputs in;
        load_inputs<VEC, false>(t, in);
        for (int tile = t0; tile < t1; tile++) {
            Tile tn = t;
            Inputs nx = in;
            if (tile + 1 < t1) {
                tn = locate(a, tick, tile + 1);
                load_inputs<VEC, false>(tn, nx);
            }
            bool counted = false;
            if (a.thr) {
                // the per-pixel depth thresholds make the run's count a handful of integer compares (see count_thr_kernel)
                unsigned int lo[kPxPerLane], cnt[kPxPerLane];
                bool any_flagged;
                load_thresholds<VEC>(a, t, lo, cnt, any_flagged);
                if (!any_flagged) {
                    wave_total += wave_count_thr(in, lo, cnt);
                    counted = true;
                }
            }
            if (!counted) {
                bool keep[kPxPerLane];
                uint4 unused[kPxPerLane];
                compute_tile<VEC, false>(a, t, in, keep, unused);
#pragma unroll
                for (int k = 0; k < kPxPerLane; k++) wave_total += __popcll(__ballot(keep[k]));  // SALU: lane mask popcount
            }
            t = tn;
            in = nx;
        }
    }
    if (lane == 0) s_wave_tot[wave] = wave_total;
    __syncthreads();
    const int run_tot = s_wave_tot[0] + s_wave_tot[1] + s_wave_tot[2] + s_wave_tot[3];

    // inputs of the first tile of phase 2 fly while the look-back polls
    Tile t = locate(a, tick, t0);
    Inputs in;
    load_inputs<VEC, true>(t, in);

    // ---- decoupled look-back over the runs of this tick (wave 0) ---------------------------------------------------
    if (wave == 0) {
        unsigned long long *st = a.run_state + (long long)tick * a.tiles_per_tick;
        if (lane == 0) {
            const unsigned long long w = (run == 0 ? kFlagPrefix : kFlagAggregate) | (unsigned int)run_tot;
            __hip_atomic_store(&st[run], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int acc = 0;
        int pos = run - 1 - lane;  // lane 0 looks at the nearest predecessor
        bool done = run == 0;
        int spins = 0;
        while (!done) {
            unsigned long long w = kFlagPrefix;  // before the first run: prefix 0
            if (pos >= 0) w = __hip_atomic_load(&st[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int flag = (unsigned int)(w >> 62);
            const unsigned long long pref = __ballot(flag == 2);
            // only the predecessors up to and including the nearest prefix holder matter
            const int first = pref ? __ffsll((long long)pref) - 1 : 63;
            const unsigned long long relevant = first >= 63 ? ~0ull : ((2ull << first) - 1ull);
            const unsigned long long empty = __ballot(flag == 0) & relevant;
            if (empty != 0) {
                // one of them has not published yet: poll again (bounded, so the grid always drains)
                if (++spins > kSpinLimit) {
                    if (lane == 0) atomicExch(a.error_flag, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            const int v = (lane <= first) ? (int)(unsigned int)w : 0;
            acc += wave_sum(v);
            if (pref) done = true;
            else pos -= 64;
        }
        if (lane == 0) {
            if (run != 0) {
                const unsigned long long w = kFlagPrefix | (unsigned int)(acc + run_tot);
                __hip_atomic_store(&st[run], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_base = acc;
        }
    }
    __syncthreads();
    int running = s_base;

    // ---- phase 2: recompute and write the run (depth now comes from cache; next tile's inputs in flight) ------------
    int *off = a.offsets + (long long)tick * (a.n_frames + 1);
    for (int tile = t0; tile < t1; tile++) {
        Tile tn = t;
        Inputs nx = in;
        if (tile + 1 < t1) {
            tn = locate(a, tick, tile + 1);
            load_inputs<VEC, true>(tn, nx);
        }
        bool keep[kPxPerLane];
        uint4 vert[kPxPerLane];
        compute_tile<VEC, true>(a, t, in, keep, vert);
        int below, wave_tot2;
        rank_from_masks(keep, below, wave_tot2);
        if (lane == 0) s_wave_tot[wave] = wave_tot2;
        __syncthreads();
        int wave_off = 0, tile_tot = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int v = s_wave_tot[i];
            if (i < wave) wave_off += v;
            tile_tot += v;
        }
        if (threadIdx.x == 0 && t.frame_start) off[t.f] = running;
        if (tile_tot > 0) {
            stage_and_store(stage, keep, vert, wave_off + below, tile_tot, a.out + tick * a.tick_vert_stride + running);
        } else {
            __syncthreads();  // s_wave_tot is rewritten by the next tile
        }
        running += tile_tot;
        t = tn;
        in = nx;
    }
    if (threadIdx.x == 0 && t1 == a.tiles_per_tick) off[a.n_frames] = running;
}


// ---- triangulation (the "next" row after the vertex path) --------------------------------------------------------
//
// Replaces MeshGenerator::generateTrianglesGradients (src/NativeUtils/meshGenerator.cpp:14-181, driver
// depthprocessing.cpp:1659-1691) and formMesh's triangle part (:1611-1627).  Per pixel with a vertex, a 2x2 stencil
// (P, U = up, UR = up-right, R = right) yields up to two triangles after depth-continuity tests that also look one
// step further along every edge; the reference's 4 row-band threads concatenated in order are plain raster order over
// y in [2, h-2), x in [1, w-2).  Same structure as the vertex path: count -> scan_kernel -> write, 8 pixels per lane,
// the 4 x 11 depth window and the 2 x 9 index window of a lane live in registers.  All integer arithmetic.

struct TriArgs {
    const FrameDesc *frames;
    const TileDesc *tiles;
    const unsigned short *depth;
    const int *pixmap;   // [n_ticks][pixels per tick]
    int *tri;            // [n_ticks][tri_cap][3]
    int *tile_counts;    // [n_ticks * tiles_per_tick] counts, then exclusive prefixes (scan_kernel)
    unsigned int *codes; // [n_ticks * tiles_per_tick * 256] per-lane 4-bit-per-pixel triangle codes: count pass -> write pass
    int tiles_per_tick;
    long long tick_pix_stride;  // pixels per tick
    long long tick_tri_stride;  // triangles per tick (capacity)
};

constexpr int kTriWin = 1536;  // triangles staged per LDS round (18 KB)

// MeshGenerator::checkTriangleConstraints (meshGenerator.cpp:14-61) on a pixel's 4 x 4 depth window W[dy + 2][dx + 1],
// dx in [-1, 2], dy in [-2, 1]; the three corners are compile-time offsets, so are the forward / backward probes.
// depth_thr = (int)((v0 + v1 + v2) / 3.0 * 0.00272 + 7.273) (:26, double) equals (272 s + 2181900) / 300000 in integers
// for EVERY possible sum s of three u16 -- proven exhaustively in tests/test_fast_division.py.
template <int X1, int Y1, int X2, int Y2>
__device__ __forceinline__ bool edge_ok(const int (&W)[4][4], int thr)
{
    const int val1 = W[Y1 + 2][X1 + 1], val2 = W[Y2 + 2][X2 + 1];
    if (abs(val1 - val2) < thr) return true;                                  // :35-36
    constexpr int SX = X2 - X1, SY = Y2 - Y1;
    const int val_forward = W[Y2 + SY + 2][X2 + SX + 1];                      // :39-40
    if (val_forward != 0 && abs(val2 - val1 - (val_forward - val2)) < thr) return true;    // :42-47
    const int val_backward = W[Y1 - SY + 2][X1 - SX + 1];                     // :50
    if (val_backward != 0 && abs(val2 - val1 - (val1 - val_backward)) < thr) return true;  // :51-56
    return false;
}

template <int X1, int Y1, int X2, int Y2, int X3, int Y3>
__device__ __forceinline__ bool tri_ok(const int (&W)[4][4])
{
    const int v0 = W[Y1 + 2][X1 + 1], v1 = W[Y2 + 2][X2 + 1], v2 = W[Y3 + 2][X3 + 1];
    if (v0 == 0 || v1 == 0 || v2 == 0) return false;                          // :22-23
    const int thr = (272 * (v0 + v1 + v2) + 2181900) / 300000;
    return edge_ok<X1, Y1, X2, Y2>(W, thr) && edge_ok<X2, Y2, X3, Y3>(W, thr) && edge_ok<X3, Y3, X1, Y1>(W, thr);
}

// Which of the four candidate triangles of a pixel are emitted: bit i = triangle i of meshGenerator.cpp:101-104
// (0: R,U,P  1: R,UR,U  2: P,UR,U  3: P,R,UR), after the vertex-index checks of :133-134.
__device__ __forceinline__ unsigned int pixel_triangles(const int (&W)[4][4], int mP, int mU, int mUR, int mR)
{
    if (mP == -1) return 0;                                                   // :113-114
    const bool t0 = tri_ok<0, 0, 0, -1, 1, 0>(W);                             // :117
    const bool t1 = tri_ok<1, 0, 0, -1, 1, -1>(W);                            // :118
    bool t2 = false, t3 = false;
    if (!t0 && !t1) {
        t2 = tri_ok<0, 0, 0, -1, 1, -1>(W);                                   // :122
        t3 = tri_ok<0, 0, 1, -1, 1, 0>(W);                                    // :123
    }
    unsigned int m = 0;
    if (t0 && mR != -1 && mU != -1) m |= 1u;
    if (t1 && mR != -1 && mUR != -1 && mU != -1) m |= 2u;
    if (t2 && mUR != -1 && mU != -1) m |= 4u;
    if (t3 && mR != -1 && mUR != -1) m |= 8u;
    return m;
}

// The same verdicts for a lane's 8 consecutive pixels of one row, with every edge evaluated once.  checkTriangleConstraints
// accepts an edge when ANY of three differences is below the triangle's threshold (:35-56), and the three differences do not
// depend on the direction the edge is walked in (walking B->A swaps the roles of the forward and the backward probe), so an
// undirected edge has ONE metric = their minimum and passes for a triangle iff metric < that triangle's threshold.  The four
// candidate triangles of a pixel share 5 edges (and the vertical one with the next pixel): 5 metrics per pixel instead
// of 12 edge walks, no branches.  D: depth rows y-2 .. y+1, columns x0-1 .. x0+9; M: vertex indices of rows y-1, y.
__device__ __forceinline__ int tri_threshold(int s) { return (272 * s + 2181900) / 300000; }   // :26, in integers

// edge_metric with the probes pre-biased: Z = depth + 2^17 for a valid probe pixel, 2^30 for an invalid one (depth 0), so
// that |x - probe| becomes one v_sad_u32 on non-negative operands and an invalid probe yields a difference no threshold
// can reach -- no select per edge.  (2 vB - vA) + 2^17 and (2 vA - vB) + 2^17 lie in [65537, 262142].
constexpr unsigned int kProbeBias = 1u << 17, kProbeInvalid = 1u << 30;

__device__ __forceinline__ unsigned int abs_diff_u32(unsigned int a, unsigned int b)
{
    unsigned int r;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));   // |a - b| in one VALU slot (the compiler has no pattern for it here)
    return r;
}

__device__ __forceinline__ unsigned int edge_metric_biased(unsigned int vA, unsigned int vB, unsigned int zBeyondB, unsigned int zBeyondA)
{
    const unsigned int a = abs_diff_u32(vA, vB);                                         // |vB - vA|                 (:35)
    const unsigned int f = abs_diff_u32(2u * vB + kProbeBias - vA, zBeyondB);            // |d - (beyondB - vB)|      (:39-47)
    const unsigned int b = abs_diff_u32(2u * vA + kProbeBias - vB, zBeyondA);            // |d - (vA - beyondA)|      (:50-56)
    return min(a, min(f, b));
}

__device__ __forceinline__ unsigned int lane_triangles(const int (&D)[4][kPxPerLane + 3], const int (&M)[2][kPxPerLane + 1], int x0, int w)
{
    unsigned int Z[4][kPxPerLane + 3];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < kPxPerLane + 3; c++) Z[r][c] = D[r][c] != 0 ? (unsigned int)D[r][c] + kProbeBias : kProbeInvalid;
    unsigned int ev[kPxPerLane + 1];   // P-U of window column c = 1 .. 9
#pragma unroll
    for (int c = 1; c <= kPxPerLane + 1; c++) ev[c - 1] = edge_metric_biased(D[2][c], D[1][c], Z[0][c], Z[3][c]);
    unsigned int code = 0;
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) {
        const int c = k + 1, x = x0 + k;
        const unsigned int vP = D[2][c], vU = D[1][c], vUR = D[1][c + 1], vR = D[2][c + 1];
        const unsigned int hP = edge_metric_biased(vP, vR, Z[2][c + 2], Z[2][c - 1]);   // P - R
        const unsigned int hU = edge_metric_biased(vU, vUR, Z[1][c + 2], Z[1][c - 1]);  // U - UR
        const unsigned int d1 = edge_metric_biased(vU, vR, Z[3][c + 2], Z[0][c - 1]);   // U - R   (down-right)
        const unsigned int d2 = edge_metric_biased(vP, vUR, Z[0][c + 2], Z[3][c - 1]);  // P - UR  (up-right)
        const unsigned int pu = ev[c - 1], ru = ev[c];                                   // P - U, R - UR
        const bool zP = vP != 0, zU = vU != 0, zUR = vUR != 0, zR = vR != 0;            // :22-23
        const unsigned int sPR = vP + vR, sUUR = vU + vUR;
        const unsigned int th0 = (unsigned int)tri_threshold((int)(sPR + vU)), th1 = (unsigned int)tri_threshold((int)(sUUR + vR));
        const unsigned int th2 = (unsigned int)tri_threshold((int)(sUUR + vP)), th3 = (unsigned int)tri_threshold((int)(sPR + vUR));
        const bool t0 = zR & zU & zP & (d1 < th0) & (pu < th0) & (hP < th0);            // R,U,P   (:117)
        const bool t1 = zR & zUR & zU & (ru < th1) & (hU < th1) & (d1 < th1);           // R,UR,U  (:118)
        const bool alt = !(t0 | t1);                                                    // :120
        const bool t2 = alt & zP & zUR & zU & (d2 < th2) & (hU < th2) & (pu < th2);     // P,UR,U (:122)
        const bool t3 = alt & zP & zR & zUR & (hP < th3) & (ru < th3) & (d2 < th3);     // P,R,UR (:123)
        const bool mP = M[1][k] != -1, mU = M[0][k] != -1, mUR = M[0][k + 1] != -1, mR = M[1][k + 1] != -1;
        const bool in_cols = (x >= 1) & (x < w - 2);                                    // :87-90
        unsigned int m = 0;
        m |= (t0 & mR & mU) ? 1u : 0u;                                                  // :133-134
        m |= (t1 & mR & mUR & mU) ? 2u : 0u;
        m |= (t2 & mUR & mU) ? 4u : 0u;
        m |= (t3 & mR & mUR) ? 8u : 0u;
        code |= ((in_cols & mP) ? m : 0u) << (4 * k);                                   // :113-114
    }
    return code;
}

// MODE 0 = count triangles per tile, 1 = write them at the scanned offsets.
template <int MODE, bool VEC>
__global__ __launch_bounds__(kThreads) void tri_kernel(const TriArgs a)
{
    __shared__ int stage[MODE == 1 ? 3 * kTriWin : 1];
    __shared__ int s_wave_tot[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tick = blockIdx.x / a.tiles_per_tick;
    const int tile = blockIdx.x - tick * a.tiles_per_tick;
    const TileDesc td = a.tiles[tile];
    const FrameDesc fd = a.frames[td.frame];
    const int w = fd.w, h = fd.h;
    const unsigned short *dep = a.depth + tick * a.tick_pix_stride + fd.depth_off;
    const int *map = a.pixmap + tick * a.tick_pix_stride + fd.depth_off;
    const int p0 = (tile - fd.tile_start) * kTile + threadIdx.x * kPxPerLane;
    const bool in_frame = p0 < fd.npix;

    // (x, y) of the lane's first pixel, as in compute_tile
    const int v = td.x0 + (int)threadIdx.x * kPxPerLane;
    int q = (int)((float)v * fd.inv_w);
    int x0 = v - q * w;
    if (x0 < 0) { q--; x0 += w; }
    if (x0 >= w) { q++; x0 -= w; }
    int y0 = td.y0 + q;
    if (!in_frame) { x0 = 0; y0 = 0; }

    unsigned int code = 0;      // 4 bits per pixel: which triangles it emits
    int M[2][kPxPerLane + 1];   // vertex indices: row y-1 (U, UR) and row y (P, R), columns x0 .. x0+8
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c <= kPxPerLane; c++) M[r][c] = -1;

    const size_t code_slot = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (MODE == 1) {
        // the count pass already evaluated every stencil: reload its verdicts, fetch only the vertex indices
        code = a.codes[code_slot];
        if (VEC && code != 0) {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int *mrow = map + (long long)(y0 - 1 + r) * w + x0;
                const int4 m0 = reinterpret_cast<const int4 *>(mrow)[0], m1 = reinterpret_cast<const int4 *>(mrow)[1];
                M[r][0] = m0.x; M[r][1] = m0.y; M[r][2] = m0.z; M[r][3] = m0.w;
                M[r][4] = m1.x; M[r][5] = m1.y; M[r][6] = m1.z; M[r][7] = m1.w;
                M[r][8] = x0 + 8 < w ? mrow[8] : -1;
            }
        }
    } else if (VEC) {
        // w % 8 == 0: the 8 pixels share row y0; window rows y0-2 .. y0+1, columns x0-1 .. x0+9
        const bool row_ok = in_frame && y0 >= 2 && y0 < h - 2;   // :87-90 (bands clamp to [2, h-2))
        if (row_ok) {
            int D[4][kPxPerLane + 3];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const unsigned short *row = dep + (long long)(y0 - 2 + r) * w;
                const uint4 c = *reinterpret_cast<const uint4 *>(row + x0);
                const unsigned int cw[4] = {c.x, c.y, c.z, c.w};
                D[r][0] = x0 > 0 ? row[x0 - 1] : 0;
#pragma unroll
                for (int k = 0; k < kPxPerLane; k++) D[r][1 + k] = (cw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
                unsigned int right = 0;
                if (x0 + 8 < w) right = *reinterpret_cast<const unsigned int *>(row + x0 + 8);
                D[r][9] = right & 0xFFFFu;
                D[r][10] = right >> 16;
            }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int *mrow = map + (long long)(y0 - 1 + r) * w + x0;
                const int4 m0 = reinterpret_cast<const int4 *>(mrow)[0], m1 = reinterpret_cast<const int4 *>(mrow)[1];
                M[r][0] = m0.x; M[r][1] = m0.y; M[r][2] = m0.z; M[r][3] = m0.w;
                M[r][4] = m1.x; M[r][5] = m1.y; M[r][6] = m1.z; M[r][7] = m1.w;
                M[r][8] = x0 + 8 < w ? mrow[8] : -1;
            }
            code = lane_triangles(D, M, x0, w);
        }
    } else {
        // general widths: a lane's pixels may span rows; every pixel fetches its own 4 x 4 window
        int x = x0, y = y0;
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) {
            if (p0 + k < fd.npix && y >= 2 && y < h - 2 && x >= 1 && x < w - 2) {
                int W[4][4];
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++) W[r][c] = dep[(long long)(y - 2 + r) * w + (x - 1 + c)];
                const long long p = (long long)y * w + x;
                M[1][k] = map[p];            // P
                M[0][k] = map[p - w];        // U
                // UR / R of this pixel are kept in the slots the VEC path would use only when they do not collide:
                // the general path re-reads them at emission time instead (see below), the code word is what counts
                code |= pixel_triangles(W, map[p], map[p - w], map[p - w + 1], map[p + 1]) << (4 * k);
            }
            x++;
            if (x == w) { x = 0; y++; }
        }
    }

    // ---- ranks ---------------------------------------------------------------------------------------------------
    const int cnt = __popc(code);
    const int incl = wave_inclusive_scan(cnt, lane);
    if (lane == 63) s_wave_tot[wave] = incl;
    int base = 0;
    if (MODE == 1) base = a.tile_counts[blockIdx.x];
    __syncthreads();
    int wave_off = 0, tile_tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int t = s_wave_tot[i];
        if (i < wave) wave_off += t;
        tile_tot += t;
    }
    if (MODE == 0) {
        a.codes[code_slot] = code;
        if (threadIdx.x == 0) a.tile_counts[blockIdx.x] = tile_tot;
        return;
    }

    // ---- stage in rank order, copy out coalesced (triangles_shifts order, meshGenerator.cpp:101-104) ------------------
    int *dst = a.tri + 3 * (tick * a.tick_tri_stride + base);
    const int rank0 = wave_off + incl - cnt;
    for (int w0 = 0; w0 < tile_tot; w0 += kTriWin) {
        int r = rank0 - w0;
        int x = x0, y = y0;
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) {
            const unsigned int m = (code >> (4 * k)) & 15u;
            if (m) {
                int mP, mU, mUR, mR;
                if (VEC) {
                    mP = M[1][k]; mU = M[0][k]; mUR = M[0][k + 1]; mR = M[1][k + 1];
                } else {
                    const long long p = (long long)y * w + x;
                    mP = map[p]; mU = map[p - w]; mUR = map[p - w + 1]; mR = map[p + 1];
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (m & (1u << i)) {
                        if ((unsigned int)r < (unsigned int)kTriWin) {
                            const int i0 = i == 0 ? mR : (i == 1 ? mR : mP);
                            const int i1 = i == 0 ? mU : (i == 1 ? mUR : (i == 2 ? mUR : mR));
                            const int i2 = i == 0 ? mP : (i == 1 ? mU : (i == 2 ? mU : mUR));
                            stage[3 * r] = i0;
                            stage[3 * r + 1] = i1;
                            stage[3 * r + 2] = i2;
                        }
                        r++;
                    }
                }
            }
            if (!VEC) {
                x++;
                if (x == w) { x = 0; y++; }
            }
        }
        __syncthreads();
        const int n = 3 * min(kTriWin, tile_tot - w0);
        for (int i = threadIdx.x; i < n; i += kThreads) __builtin_nontemporal_store(stage[i], &dst[3 * w0 + i]);   // written once
        __syncthreads();
    }
}


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

__global__ __launch_bounds__(kThreads) void radial_gather_cand_kernel(const FrameDesc *frames, const TileDesc *tiles, const unsigned short *depth,
                                                                      const unsigned char *rgb, const uint4 *cand, unsigned short *map_copy,
                                                                      unsigned char *colors_copy, int tiles_per_tick, long long tick_pix_stride)
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
        const uint4 c = cand[fd.depth_off + p];
        // all four candidate depths are fetched at once (independent loads); the first non-zero one wins
        const unsigned short d0 = c.x ? depth[fb + (c.x - 1u)] : 0, d1 = c.y ? depth[fb + (c.y - 1u)] : 0;
        const unsigned short d2 = c.z ? depth[fb + (c.z - 1u)] : 0, d3 = c.w ? depth[fb + (c.w - 1u)] : 0;
        unsigned int src = 0;
        unsigned short d = 0;
        if (d0) { src = c.x; d = d0; }
        else if (d1) { src = c.y; d = d1; }
        else if (d2) { src = c.z; d = d2; }
        else if (d3) { src = c.w; d = d3; }
        unsigned char c0 = 0, c1 = 0, c2 = 0;
        if (src) {
            const long long sidx = fb + (long long)(src - 1u);
            c0 = rgb[3 * sidx]; c1 = rgb[3 * sidx + 1]; c2 = rgb[3 * sidx + 2];
        }
        map_copy[fb + p] = d;
        colors_copy[3 * (fb + p)] = c0;
        colors_copy[3 * (fb + p) + 1] = c1;
        colors_copy[3 * (fb + p) + 2] = c2;
    }
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

// Fills a sensor's unprojection tables with the reference's own operations (depthprocessing.cpp:151-152):
// xtab[x] = (float(x) - cx) / fx, ytab[y] = (cy - float(y)) / fy -- IEEE subtraction and correctly rounded division.
__global__ __launch_bounds__(kThreads) void table_kernel(const FrameDesc *frames, const SensorParams *params, int n_frames, float *xtab,
                                                         float *ytab)
{
    const int f = blockIdx.y;
    if (f >= n_frames) return;
    const FrameDesc fd = frames[f];
    const SensorParams P = params[f];
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < fd.w + fd.h; i += gridDim.x * kThreads) {
        if (i < fd.w) xtab[fd.xtab_off + i] = ((float)i - P.cx) / P.fx;
        else ytab[fd.ytab_off + (i - fd.w)] = (P.cy - (float)(i - fd.w)) / P.fy;
    }
}

// Merged-cloud assembly after the all-gather (one sensor block per GPU): shard r holds, for every tick, the cloud of
// its own sensors at [r][tick][0 .. count) of a fixed-capacity slab; the merged cloud of a tick is the concatenation
// of the shards in rank order = formMesh's sensor order (depthprocessing.cpp:1594-1608).
struct MergeArgs {
    const uint4 *shards;     // [n_shards][n_ticks][shard_cap]
    const int *shard_off;    // [n_shards][n_ticks][maps_per_shard + 1]  (the gathered lsnFusionRun offsets)
    uint4 *merged;           // [n_ticks][merged_cap]
    int *merged_off;         // [n_ticks][n_shards * maps_per_shard + 1]
    long long shard_cap, merged_cap;
    int n_shards, n_ticks, maps_per_shard;
};

__global__ __launch_bounds__(kThreads) void merge_shards_kernel(const MergeArgs a)
{
    const int tick = blockIdx.y / a.n_shards;
    const int shard = blockIdx.y - tick * a.n_shards;
    const int mps1 = a.maps_per_shard + 1;
    int base = 0;
    for (int r = 0; r < shard; r++) base += a.shard_off[((long long)r * a.n_ticks + tick) * mps1 + a.maps_per_shard];
    const int *my_off = a.shard_off + ((long long)shard * a.n_ticks + tick) * mps1;
    const int count = my_off[a.maps_per_shard];
    const uint4 *src = a.shards + ((long long)shard * a.n_ticks + tick) * a.shard_cap;
    uint4 *dst = a.merged + (long long)tick * a.merged_cap + base;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < count; i += gridDim.x * kThreads) {
        const uint4 v = src[i];                               // streaming stores, like the write kernel (kNontemporalStores)
        __builtin_nontemporal_store(v.x, &dst[i].x);
        __builtin_nontemporal_store(v.y, &dst[i].y);
        __builtin_nontemporal_store(v.z, &dst[i].z);
        __builtin_nontemporal_store(v.w, &dst[i].w);
    }
    if (blockIdx.x == 0 && threadIdx.x <= a.maps_per_shard) {
        int *mo = a.merged_off + (long long)tick * (a.n_shards * a.maps_per_shard + 1);
        if (threadIdx.x < a.maps_per_shard) mo[shard * a.maps_per_shard + threadIdx.x] = base + my_off[threadIdx.x];
        else if (shard == a.n_shards - 1) mo[a.n_shards * a.maps_per_shard] = base + count;
    }
}

// ---- survivor exchange (multi-GPU): ship what the vertices are made of ---------------------------------------------------------
// A vertex is 16 bytes, the inputs it is computed from are 5 (u16 depth + RGB8) plus one bit of "this pixel survived".
// pack_kernel writes a shard's survivors as compact depth / colour streams in vertex order and the survivor mask;
// after the all-gather recon_kernel rebuilds every sensor's vertices on every GPU with the same arithmetic as
// fuse_kernel<1> (same helpers, same rounding) straight into the merged cloud.  Only for plans on the wide-load path
// (all widths multiples of 8) with identically sized sensors; other rigs exchange vertices (merge_shards_kernel).
struct PackArgs {
    unsigned char *mask;      // [n_ticks][cap / 8]      bit (p & 7) of byte p >> 3 = pixel p of the tick survived
    unsigned short *depth_c;  // [n_ticks][cap]          survivors' depth, vertex order
    unsigned char *rgb_c;     // [n_ticks][cap][3]       survivors' colour, vertex order
};

__global__ __launch_bounds__(kThreads) void pack_kernel(const FuseArgs a, const PackArgs pk)
{
    __shared__ int s_wave_tot[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tick = blockIdx.x / a.tiles_per_tick;
    const int tile = blockIdx.x - tick * a.tiles_per_tick;
    const Tile t = locate(a, tick, tile);
    Inputs in;
    load_inputs<true, true>(t, in);
    const SensorParams P = a.params[t.f];
    float xf[kPxPerLane], yf[kPxPerLane];
    tile_factors<true>(t, xf, yf);
    bool keep[kPxPerLane];
    uint4 unused[kPxPerLane];
    compute_pixels<false>(a, P, in, xf, yf, keep, unused);
    int below, wave_total;
    rank_from_masks(keep, below, wave_total);
    if (lane == 0) s_wave_tot[wave] = wave_total;
    const int base = a.tile_counts[blockIdx.x];               // exclusive prefix inside the tick (scan_kernel)
    __syncthreads();
    int wave_off = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i < wave) wave_off += s_wave_tot[i];
    const int p0 = t.px0 + threadIdx.x * kPxPerLane;
    if (p0 >= t.npix) return;
    unsigned int m8 = 0;
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) m8 |= (keep[k] ? 1u : 0u) << k;
    pk.mask[(tick * a.tick_depth_stride + t.pix_base + p0) >> 3] = (unsigned char)m8;
    long long r = tick * a.tick_vert_stride + base + wave_off + below;
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) {
        if (keep[k]) {
            const int b = 3 * k;
            const unsigned int lo = in.cw[b >> 2], hi = in.cw[(b >> 2) + 1 < 6 ? (b >> 2) + 1 : 5];
            const unsigned int rgb = __funnelshift_r(lo, hi, (b & 3) * 8);
            pk.depth_c[r] = (unsigned short)((k & 1) ? in.dw[k >> 1] >> 16 : in.dw[k >> 1] & 0xFFFFu);
            pk.rgb_c[3 * r] = (unsigned char)rgb;
            pk.rgb_c[3 * r + 1] = (unsigned char)(rgb >> 8);
            pk.rgb_c[3 * r + 2] = (unsigned char)(rgb >> 16);
            r++;
        }
    }
}

struct ReconArgs {
    const unsigned char *mask;       // [n_shards][n_ticks][cap_loc / 8]
    const unsigned short *depth_c;   // [n_shards][n_ticks][slab]
    const unsigned char *rgb_c;      // [n_shards][n_ticks][slab][3]
    const int *tile_prefix;          // [n_shards][n_ticks][tiles_loc]
    const int *shard_off;            // [n_shards][n_ticks][maps_per_shard + 1]
    int *merged_off;                 // [n_ticks][n_maps + 1]
    long long slab, cap_loc;
    int tiles_loc, n_shards, maps_per_shard;
};

// `a` describes the WHOLE rig (all sensors, their parameters, a.out = the merged cloud).
__global__ __launch_bounds__(kThreads) void recon_kernel(const FuseArgs a, const ReconArgs r)
{
    __shared__ uint4 stage[kWin + kWin / 8];
    __shared__ int s_wave_tot[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tick = blockIdx.x / a.tiles_per_tick;
    const int tile = blockIdx.x - tick * a.tiles_per_tick;
    const Tile t = locate(a, tick, tile);
    const int shard = t.f / r.maps_per_shard;
    const FrameDesc f0 = a.frames[shard * r.maps_per_shard];   // first sensor of the owning shard
    const long long st = (long long)shard * a.n_ticks + tick;  // (shard, tick) slot in the gathered arrays
    const int local_tile = tile - f0.tile_start;
    const int p0 = t.px0 + threadIdx.x * kPxPerLane;
    const bool in_frame = p0 < t.npix;
    const unsigned int m8 = in_frame ? r.mask[(st * r.cap_loc + (t.pix_base - f0.depth_off) + p0) >> 3] : 0u;
    bool keep[kPxPerLane];
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) keep[k] = (m8 >> k) & 1u;
    int below, wave_total;
    rank_from_masks(keep, below, wave_total);
    if (lane == 0) s_wave_tot[wave] = wave_total;
    const int tile_base = r.tile_prefix[st * r.tiles_loc + local_tile];
    int shard_base = 0;
    for (int q = 0; q < shard; q++) shard_base += r.shard_off[((long long)q * a.n_ticks + tick) * (r.maps_per_shard + 1) + r.maps_per_shard];
    __syncthreads();
    int wave_off = 0, tile_tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = s_wave_tot[i];
        if (i < wave) wave_off += v;
        tile_tot += v;
    }
    // the lane's survivors are consecutive entries of the shard's compact streams
    const unsigned short *dc = r.depth_c + st * r.slab;
    const unsigned char *cc = r.rgb_c + 3 * st * r.slab;
    long long ci = tile_base + wave_off + below;
    unsigned int d[kPxPerLane], c[kPxPerLane];
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) {
        d[k] = 0;
        c[k] = 0;
        if (keep[k]) {
            d[k] = dc[ci];
            c[k] = cc[3 * ci] | (cc[3 * ci + 1] << 8) | (cc[3 * ci + 2] << 16);
            ci++;
        }
    }
    const SensorParams P = a.params[t.f];
    float xf[kPxPerLane], yf[kPxPerLane];
    tile_factors<true>(t, xf, yf);
    uint4 vert[kPxPerLane];
#pragma unroll
    for (int k = 0; k < kPxPerLane; k += 2) {
        f2 ox, oy, oz;
        unproject2(f2{(float)d[k], (float)d[k + 1]}, f2{xf[k], xf[k + 1]}, f2{yf[k], yf[k + 1]}, P, ox, oy, oz);
        vert[k] = make_uint4(c[k] | 0xFF000000u, __float_as_uint(ox.x), __float_as_uint(oy.x), __float_as_uint(oz.x));
        vert[k + 1] = make_uint4(c[k + 1] | 0xFF000000u, __float_as_uint(ox.y), __float_as_uint(oy.y), __float_as_uint(oz.y));
    }
    if (t.frame_start && threadIdx.x == 0) {
        int *mo = r.merged_off + (long long)tick * (a.n_frames + 1);
        mo[t.f] = shard_base + tile_base;                     // the frame's first tile: its prefix is the sensor's offset in the shard
        if (t.f == a.n_frames - 1) {
            int total = 0;
            for (int q = 0; q < r.n_shards; q++) total += r.shard_off[((long long)q * a.n_ticks + tick) * (r.maps_per_shard + 1) + r.maps_per_shard];
            mo[a.n_frames] = total;
        }
    }
    stage_and_store(stage, keep, vert, wave_off + below, tile_tot, a.out + tick * a.tick_vert_stride + shard_base + tile_base);
}

}  // namespace

// -------------------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------------------

struct LsnFusion {
    int device = 0;
    int n_ticks = 0, n_maps = 0;
    std::vector<int> w, h;
    long long cap = 0;  // vertices per tick
    long long tick_depth_elems = 0, tick_rgb_bytes = 0;
    int tiles_per_tick = 0;
    bool vec_ok = false;
    bool params_set = false;
    int mode = 0;
    int tiles_per_run_override = 0;  // $LSN_TILES_PER_RUN (tuning / tests)
    bool want_pixmap = false;        // set by lsnFusionRunMesh around its vertex pass
    float bounds[6] = {0, 0, 0, 0, 0, 0};
    lsn::DevBuf frames, tile_frame, params, tile_counts, tile_state, misc;  // misc: error flag (word 0) + tickets
    lsn::DevBuf xtab, ytab;
    lsn::DevBuf pixmap, tri_counts, tri_codes;  // triangulation scratch, allocated on first use
    lsn::DevBuf winner, map_copy, colors_copy, radial;  // radial-correction scratch, allocated on first use
    lsn::DevBuf cand;                                   // [pixels per tick][4] warp candidates of the current intrinsics
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
    int runs_with_params = 0;
    std::vector<float> last_intr, last_wt;
    float thr_build_ms = 0;
    // dominant-kernel timing
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t ev_used = 0;
    double acc_ms = 0;
    long long launches = 0;
    const char *timed_kernel = nullptr;  // which kernel the event pairs bracket (set by the entry point that records them)
    std::mutex mu;
};

extern "C" LsnFusion *lsnFusionCreate(int device, int n_ticks, int n_maps, const int *widths, const int *heights)
{
    lsn::clear_error();
    if (n_ticks <= 0 || n_maps <= 0 || !widths || !heights) {
        lsn::set_error("lsnFusionCreate: bad arguments (n_ticks=%d n_maps=%d)", n_ticks, n_maps);
        return nullptr;
    }
    LSN_HIP_NULL(hipSetDevice(device));
    LsnFusion *p = new (std::nothrow) LsnFusion();
    if (!p) return nullptr;
    p->device = device;
    if (const char *env = getenv("LSN_TILES_PER_RUN")) p->tiles_per_run_override = atoi(env);
    if (const char *env = getenv("LSN_NO_THRESHOLDS")) p->thr_enabled = atoi(env) == 0;
    p->n_ticks = n_ticks;
    p->n_maps = n_maps;
    std::vector<FrameDesc> fr(n_maps);
    std::vector<TileDesc> tf;
    long long doff = 0, coff = 0;
    int tiles = 0, xoff = 0, yoff = 0;
    bool vec = true;
    for (int i = 0; i < n_maps; i++) {
        if (widths[i] <= 0 || heights[i] <= 0 || (long long)widths[i] * heights[i] > (1ll << 30)) {
            lsn::set_error("lsnFusionCreate: bad frame size %dx%d", widths[i], heights[i]);
            delete p;
            return nullptr;
        }
        p->w.push_back(widths[i]);
        p->h.push_back(heights[i]);
        const int npix = widths[i] * heights[i];
        fr[i].w = widths[i];
        fr[i].h = heights[i];
        fr[i].npix = npix;
        fr[i].tile_start = tiles;
        fr[i].depth_off = doff;
        fr[i].rgb_off = coff;
        fr[i].xtab_off = xoff;
        fr[i].ytab_off = yoff;
        fr[i].pad1 = 0;
        xoff += (widths[i] + 7) & ~7;  // keeps every sensor's row 32-B aligned for the float4 loads
        yoff += heights[i];
        const int nt = (npix + kTile - 1) / kTile;
        fr[i].inv_w = 1.0f / (float)widths[i];
        for (int t = 0; t < nt; t++) {
            const long long px = (long long)t * kTile;
            TileDesc td;
            td.frame = i;
            td.y0 = (int)(px / widths[i]);
            td.x0 = (int)(px % widths[i]);
            td.pad = 0;
            tf.push_back(td);
        }
        tiles += nt;
        doff += npix;
        coff += 3ll * npix;
        // 16-B depth loads / 8-B colour loads need every frame to start 8-pixel aligned and rows not to split a lane
        if (widths[i] % 8 != 0) vec = false;
    }
    p->cap = doff;
    p->tick_depth_elems = doff;
    p->tick_rgb_bytes = coff;
    p->tiles_per_tick = tiles;
    p->vec_ok = vec;
    if (doff > 0x7FFFFFFFll) {
        lsn::set_error("lsnFusionCreate: a tick may not exceed 2^31-1 pixels (Mesh.nVertices is an int)");
        delete p;
        return nullptr;
    }
    if ((long long)tiles * n_ticks > 0x7FFFFFFFll) {
        lsn::set_error("lsnFusionCreate: too many tiles");
        delete p;
        return nullptr;
    }
    const size_t n_tiles_total = (size_t)tiles * n_ticks;
    if (p->frames.reserve(sizeof(FrameDesc) * n_maps) || p->tile_frame.reserve(sizeof(TileDesc) * tf.size()) ||
        p->params.reserve(sizeof(SensorParams) * n_maps) || p->tile_counts.reserve(sizeof(int) * n_tiles_total) ||
        p->tile_state.reserve(sizeof(unsigned long long) * n_tiles_total) || p->misc.reserve(128 * ((size_t)n_ticks + 1)) ||
        p->xtab.reserve(sizeof(float) * (size_t)(xoff + 8)) || p->ytab.reserve(sizeof(float) * (size_t)(yoff + 8))) {
        delete p;
        return nullptr;
    }
    if (hipMemset(p->misc.p, 0, 128 * ((size_t)n_ticks + 1)) != hipSuccess) {
        lsn::set_error("lsnFusionCreate: scratch initialisation failed");
        delete p;
        return nullptr;
    }
    if (hipMemcpy(p->frames.p, fr.data(), sizeof(FrameDesc) * n_maps, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->tile_frame.p, tf.data(), sizeof(TileDesc) * tf.size(), hipMemcpyHostToDevice) != hipSuccess) {
        lsn::set_error("lsnFusionCreate: geometry upload failed");
        delete p;
        return nullptr;
    }
    return p;
}

extern "C" void lsnFusionDestroy(LsnFusion *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (auto &e : p->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (p->side) {
        (void)hipStreamSynchronize(p->side);
        (void)hipStreamDestroy(p->side);
        (void)hipEventDestroy(p->ev_counted);
        (void)hipEventDestroy(p->ev_written[0]);
        (void)hipEventDestroy(p->ev_written[1]);
    }
    delete p;
}

extern "C" long long lsnFusionTickCapacity(const LsnFusion *p) { return p ? p->cap : 0; }

extern "C" int lsnFusionSetParams(LsnFusion *p, const float *intr, const float *wt, const float *bounds6, void *stream)
{
    lsn::clear_error();
    if (!p || !intr || !wt || !bounds6) {
        lsn::set_error("lsnFusionSetParams: null argument");
        return -1;
    }
    LSN_HIP(hipSetDevice(p->device));
    {
        // LiveScanServer passes the same calibration with every call (KinectServer.cs:470-490): nothing to do then
        std::lock_guard<std::mutex> g(p->mu);
        if (p->params_set && p->last_intr.size() == 7 * (size_t)p->n_maps &&
            memcmp(p->last_intr.data(), intr, sizeof(float) * 7 * p->n_maps) == 0 &&
            memcmp(p->last_wt.data(), wt, sizeof(float) * 12 * p->n_maps) == 0 && memcmp(p->bounds, bounds6, sizeof(p->bounds)) == 0)
            return 0;
    }
    std::vector<SensorParams> sp(p->n_maps);
    for (int i = 0; i < p->n_maps; i++) {
        // IntrinsicCameraParameters(float*) / WorldTranformation(float*), include/NativeUtils/depthprocessing.h:56-63,96-97
        const float *ip = intr + 7 * i, *tp = wt + 12 * i;
        SensorParams &s = sp[i];
        s.cx = ip[0]; s.cy = ip[1]; s.fx = ip[2]; s.fy = ip[3];
        s.t0 = tp[0]; s.t1 = tp[1]; s.t2 = tp[2];
        s.r00 = tp[3]; s.r01 = tp[4]; s.r02 = tp[5];
        s.r10 = tp[6]; s.r11 = tp[7]; s.r12 = tp[8];
        s.r20 = tp[9]; s.r21 = tp[10]; s.r22 = tp[11];
    }
    // pageable source: hipMemcpyAsync copies it out before returning, so the local vector may die
    LSN_HIP(hipMemcpyAsync(p->params.p, sp.data(), sizeof(SensorParams) * p->n_maps, hipMemcpyHostToDevice,
                           lsn::as_stream(stream)));
    hipLaunchKernelGGL(table_kernel, dim3(8, (unsigned)p->n_maps), dim3(kThreads), 0, lsn::as_stream(stream), p->frames.as<FrameDesc>(),
                       p->params.as<SensorParams>(), p->n_maps, p->xtab.as<float>(), p->ytab.as<float>());
    LSN_HIP(hipGetLastError());
    LSN_HIP(hipStreamSynchronize(lsn::as_stream(stream)));
    std::lock_guard<std::mutex> g(p->mu);
    memcpy(p->bounds, bounds6, sizeof(p->bounds));
    p->last_intr.assign(intr, intr + 7 * (size_t)p->n_maps);
    p->last_wt.assign(wt, wt + 12 * (size_t)p->n_maps);
    p->params_set = true;
    p->params_gen++;  // counts made ahead with the old parameters are void
    p->thr_valid = false;
    p->runs_with_params = 0;
    return 0;
}

extern "C" int lsnFusionSetMode(LsnFusion *p, int mode)
{
    if (!p || mode < 0 || mode > 1) {
        lsn::set_error("lsnFusionSetMode: mode must be 0 (count/scan/write launches) or 1 (single launch, runs + look-back)");
        return -1;
    }
    p->mode = mode;
    return 0;
}

extern "C" int lsnFusionSetPipelined(LsnFusion *p, int enable)
{
    lsn::clear_error();
    if (!p) return -1;
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    if (enable && !p->side) {
        if (p->tile_counts_b.reserve(sizeof(int) * (size_t)p->tiles_per_tick * p->n_ticks) ||
            p->offs_int.reserve(sizeof(int) * 2 * (size_t)p->n_ticks * (p->n_maps + 1)))
            return -1;
        LSN_HIP(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
        LSN_HIP(hipEventCreateWithFlags(&p->ev_counted, hipEventDisableTiming));
        LSN_HIP(hipEventCreateWithFlags(&p->ev_written[0], hipEventDisableTiming));
        LSN_HIP(hipEventCreateWithFlags(&p->ev_written[1], hipEventDisableTiming));
    }
    if (!enable && p->side) LSN_HIP(hipStreamSynchronize(p->side));
    p->pipelined = enable != 0;
    p->calls = 0;
    return 0;
}

extern "C" int lsnFusionProfile(LsnFusion *p, int enable)
{
    if (!p) return -1;
    p->profile = enable != 0;
    return 0;
}

static int drain_events(LsnFusion *p)
{
    for (size_t i = 0; i < p->ev_used; i++) {
        float ms = 0;
        LSN_HIP(hipEventSynchronize(p->events[i].second));
        LSN_HIP(hipEventElapsedTime(&ms, p->events[i].first, p->events[i].second));
        p->acc_ms += ms;
        p->launches++;
    }
    p->ev_used = 0;
    return 0;
}

extern "C" int lsnFusionKernelStats(LsnFusion *p, double *avg_ms, long long *launches, char *name, int name_len, int reset)
{
    lsn::clear_error();
    if (!p) return -1;
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    if (drain_events(p)) return -1;
    if (avg_ms) *avg_ms = p->launches ? p->acc_ms / (double)p->launches : 0.0;
    if (launches) *launches = p->launches;
    if (name && name_len > 0)
        snprintf(name, (size_t)name_len, "%s", p->timed_kernel ? p->timed_kernel : (p->mode == 0 ? "fuse_kernel<1>" : "run_kernel"));
    if (reset) {
        p->acc_ms = 0;
        p->launches = 0;
    }
    return 0;
}

// Kernel arguments of one call (everything but the per-mode scratch selection).
static void fill_args(LsnFusion *p, FuseArgs &a, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets)
{
    a.frames = p->frames.as<FrameDesc>();
    a.tiles = p->tile_frame.as<TileDesc>();
    a.params = p->params.as<SensorParams>();
    a.xtab = p->xtab.as<float>();
    a.ytab = p->ytab.as<float>();
    a.depth = static_cast<const unsigned short *>(d_depth);
    a.rgb = static_cast<const unsigned char *>(d_colors);
    a.out = static_cast<uint4 *>(d_vertices);
    a.tile_counts = p->tile_counts.as<int>();
    a.run_state = p->tile_state.as<unsigned long long>();
    a.error_flag = p->misc.as<int>();
    a.ticket = p->misc.as<unsigned int>() + 32;
    a.offsets = d_offsets;
    a.pixmap = p->want_pixmap ? p->pixmap.as<int>() : nullptr;
    a.n_frames = p->n_maps;
    a.tiles_per_tick = p->tiles_per_tick;
    a.n_ticks = p->n_ticks;
    // mode 1: runs of consecutive tiles.  Long runs amortise the look-back and the second (cached) depth read, short
    // runs balance the load: aim at >= ~6 runs per workgroup slot (7 per CU x 256 CUs), at most 8 tiles per run.
    {
        const long long total = (long long)p->tiles_per_tick * p->n_ticks;
        long long tpr = total / (6ll * 7 * 256);
        if (p->tiles_per_run_override > 0) tpr = p->tiles_per_run_override;
        if (tpr < 1) tpr = 1;
        if (tpr > 8 && p->tiles_per_run_override <= 0) tpr = 8;
        if (tpr > p->tiles_per_tick) tpr = p->tiles_per_tick;
        a.tiles_per_run = (int)tpr;
        a.runs_per_tick = (p->tiles_per_tick + a.tiles_per_run - 1) / a.tiles_per_run;
    }
    a.tick_depth_stride = p->tick_depth_elems;
    a.tick_rgb_stride = p->tick_rgb_bytes;
    a.tick_vert_stride = p->cap;
    a.minX = p->bounds[0]; a.minY = p->bounds[1]; a.minZ = p->bounds[2];
    a.maxX = p->bounds[3]; a.maxY = p->bounds[4]; a.maxZ = p->bounds[5];

    a.depth_next = nullptr;
    a.tile_counts_next = nullptr;
    a.thr = p->thr_valid ? p->thr.as<unsigned int>() : nullptr;
}

// Called at the top of every run (p->mu held): from the second run with the same parameters on, the count pass uses the
// per-pixel depth thresholds; they are built here, once, on the caller's stream.
static int ensure_thresholds(LsnFusion *p, hipStream_t s)
{
    if (!p->thr_enabled || p->thr_valid) return 0;
    if (++p->runs_with_params < 2) return 0;
    if (p->thr.reserve(sizeof(unsigned int) * (size_t)p->cap)) return -1;
    hipEvent_t e0, e1;
    LSN_HIP(hipEventCreate(&e0));
    LSN_HIP(hipEventCreate(&e1));
    LSN_HIP(hipEventRecord(e0, s));
    hipLaunchKernelGGL(thresh_kernel, dim3(256, (unsigned)p->n_maps), dim3(kThreads), 0, s, p->frames.as<FrameDesc>(), p->params.as<SensorParams>(),
                       p->n_maps, p->xtab.as<float>(), p->ytab.as<float>(), p->thr.as<unsigned int>(), p->bounds[0], p->bounds[1], p->bounds[2],
                       p->bounds[3], p->bounds[4], p->bounds[5]);
    LSN_HIP(hipGetLastError());
    LSN_HIP(hipEventRecord(e1, s));
    LSN_HIP(hipEventSynchronize(e1));
    (void)hipEventElapsedTime(&p->thr_build_ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    p->thr_valid = true;
    return 0;
}

// The count pass of one batch into a.tile_counts: from the thresholds when they exist, else arithmetically.
static void launch_count(LsnFusion *p, bool vec, hipStream_t s, const FuseArgs &a)
{
    if (a.thr) {
        static const int tune = getenv("LSN_TICK_GROUP") ? atoi(getenv("LSN_TICK_GROUP")) : 0;
        int G = tune ? tune : (p->n_ticks >= 8 ? 8 : (p->n_ticks >= 4 ? 4 : 1));
        if (G != 16 && G != 8 && G != 4 && G != 2) G = 1;
        const int grid = p->tiles_per_tick * ((p->n_ticks + G - 1) / G);
#define LSN_COUNT_THR(GG)                                                                                         \
    do {                                                                                                          \
        if (vec) hipLaunchKernelGGL((count_thr_kernel<true, GG>), dim3(grid), dim3(kThreads), 0, s, a);           \
        else     hipLaunchKernelGGL((count_thr_kernel<false, GG>), dim3(grid), dim3(kThreads), 0, s, a);          \
    } while (0)
        if (G == 16) LSN_COUNT_THR(16);
        else if (G == 8) LSN_COUNT_THR(8);
        else if (G == 4) LSN_COUNT_THR(4);
        else if (G == 2) LSN_COUNT_THR(2);
        else LSN_COUNT_THR(1);
#undef LSN_COUNT_THR
    } else {
        const int grid = p->tiles_per_tick * p->n_ticks;
        if (vec) hipLaunchKernelGGL((fuse_kernel<0, true>), dim3(grid), dim3(kThreads), 0, s, a);
        else     hipLaunchKernelGGL((fuse_kernel<0, false>), dim3(grid), dim3(kThreads), 0, s, a);
    }
}

// Next HIP-event pair of the dominant-kernel timer (profiling on).
static int next_event_pair(LsnFusion *p, hipEvent_t &e0, hipEvent_t &e1)
{
    if (p->ev_used == p->events.size()) {
        if (p->events.size() >= 4096) {
            if (drain_events(p)) return -1;
        } else {
            hipEvent_t x, y;
            LSN_HIP(hipEventCreate(&x));
            LSN_HIP(hipEventCreate(&y));
            p->events.emplace_back(x, y);
        }
    }
    e0 = p->events[p->ev_used].first;
    e1 = p->events[p->ev_used].second;
    p->ev_used++;
    return 0;
}

template <int MODE>
static void launch(bool vec, int grid, hipStream_t s, const FuseArgs &a)
{
    if (vec) hipLaunchKernelGGL((fuse_kernel<MODE, true>), dim3(grid), dim3(kThreads), 0, s, a);
    else     hipLaunchKernelGGL((fuse_kernel<MODE, false>), dim3(grid), dim3(kThreads), 0, s, a);
}

extern "C" int lsnFusionRun(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets,
                            void *stream)
{
    lsn::clear_error();
    if (!p || !d_depth || !d_colors || !d_vertices || !d_offsets) {
        lsn::set_error("lsnFusionRun: null argument");
        return -1;
    }
    if (!p->params_set) {
        lsn::set_error("lsnFusionRun: lsnFusionSetParams has not been called");
        return -1;
    }
    if (((uintptr_t)d_vertices & 15) != 0) {
        lsn::set_error("lsnFusionRun: d_vertices must be 16-byte aligned");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    hipStream_t s = lsn::as_stream(stream);

    if (ensure_thresholds(p, s)) return -1;
    FuseArgs a;
    fill_args(p, a, d_depth, d_colors, d_vertices, d_offsets);

    // the wide-load path also needs 16-B aligned buffers and every tick to start 16-B / 8-B aligned
    const bool vec = p->vec_ok && ((uintptr_t)d_depth & 15) == 0 && ((uintptr_t)d_colors & 7) == 0 &&
                     (p->tick_depth_elems % 8) == 0;
    const int grid = p->tiles_per_tick * p->n_ticks;

    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (p->profile && next_event_pair(p, e0, e1)) return -1;

    if (p->pipelined && p->mode == 0 && !p->want_pixmap) {
        // Count + scan of THIS call go to the side stream: they only read the inputs (promised resident by
        // lsnFusionSetPipelined) and write this call's half of the double-buffered scratch, so they overlap with the
        // previous call's write kernel, which is still running on the caller's stream (a VALU-bound kernel beside an
        // HBM-bound one).  The caller's stream then waits for them and runs the write kernel.
        const int b = (int)(p->calls & 1);
        const size_t off_elems = (size_t)p->n_ticks * (p->n_maps + 1);
        a.tile_counts = b ? p->tile_counts_b.as<int>() : p->tile_counts.as<int>();
        int *off_int = p->offs_int.as<int>() + b * off_elems;
        if (p->calls >= 2) LSN_HIP(hipStreamWaitEvent(p->side, p->ev_written[b], 0));  // the write that last read this half
        FuseArgs ac = a;
        ac.offsets = off_int;
        launch_count(p, vec, p->side, ac);
        hipLaunchKernelGGL(scan_kernel, dim3((unsigned)p->n_ticks), dim3(kThreads), 0, p->side, ac.tile_counts, ac.tiles_per_tick, ac.frames,
                           ac.n_frames, off_int);
        LSN_HIP(hipEventRecord(p->ev_counted, p->side));
        LSN_HIP(hipStreamWaitEvent(s, p->ev_counted, 0));
        LSN_HIP(hipMemcpyAsync(d_offsets, off_int, sizeof(int) * off_elems, hipMemcpyDeviceToDevice, s));
        if (e0) LSN_HIP(hipEventRecord(e0, s));
        launch<1>(vec, grid, s, a);
        if (e1) LSN_HIP(hipEventRecord(e1, s));
        LSN_HIP(hipEventRecord(p->ev_written[b], s));
        p->calls++;
    } else if (p->mode == 0 || p->want_pixmap) {
        launch_count(p, vec, s, a);
        hipLaunchKernelGGL(scan_kernel, dim3((unsigned)p->n_ticks), dim3(kThreads), 0, s, a.tile_counts, a.tiles_per_tick, a.frames, a.n_frames,
                           a.offsets);
        if (e0) LSN_HIP(hipEventRecord(e0, s));
        launch<1>(vec, grid, s, a);
        if (e1) LSN_HIP(hipEventRecord(e1, s));
    } else {
        LSN_HIP(hipMemsetAsync(p->tile_state.p, 0, sizeof(unsigned long long) * (size_t)grid, s));
        LSN_HIP(hipMemsetAsync(p->misc.as<char>() + 128, 0, 128 * (size_t)p->n_ticks, s));  // tickets; the error flag is sticky
        if (e0) LSN_HIP(hipEventRecord(e0, s));
        const int rgrid = a.runs_per_tick * p->n_ticks;
        if (vec) hipLaunchKernelGGL((run_kernel<true>), dim3(rgrid), dim3(kThreads), 0, s, a);
        else     hipLaunchKernelGGL((run_kernel<false>), dim3(rgrid), dim3(kThreads), 0, s, a);
        if (e1) LSN_HIP(hipEventRecord(e1, s));
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

// Streamed calls: this batch is written while the NEXT batch (already resident) is counted by the same kernel.
extern "C" int lsnFusionRunStreamed(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets,
                                    const void *d_next_depth, void *stream)
{
    lsn::clear_error();
    if (!p || !d_depth || !d_colors || !d_vertices || !d_offsets) {
        lsn::set_error("lsnFusionRunStreamed: null argument");
        return -1;
    }
    if (!p->params_set) {
        lsn::set_error("lsnFusionRunStreamed: lsnFusionSetParams has not been called");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    hipStream_t s = lsn::as_stream(stream);
    const size_t n_tiles = (size_t)p->tiles_per_tick * p->n_ticks;
    const size_t off_elems = (size_t)p->n_ticks * (p->n_maps + 1);
    if (p->tile_counts_b.reserve(sizeof(int) * n_tiles) || p->offs_int.reserve(sizeof(int) * 2 * off_elems)) return -1;

    if (ensure_thresholds(p, s)) return -1;
    FuseArgs a;
    fill_args(p, a, d_depth, d_colors, d_vertices, d_offsets);
    const bool vec = p->vec_ok && ((uintptr_t)d_depth & 15) == 0 && ((uintptr_t)d_colors & 7) == 0 && (p->tick_depth_elems % 8) == 0 &&
                     (!d_next_depth || ((uintptr_t)d_next_depth & 15) == 0);
    const int grid = (int)n_tiles;
    int *cur = p->stream_half ? p->tile_counts_b.as<int>() : p->tile_counts.as<int>();
    int *nxt = p->stream_half ? p->tile_counts.as<int>() : p->tile_counts_b.as<int>();
    int *off_cur = p->offs_int.as<int>() + (p->stream_half ? off_elems : 0);
    int *off_nxt = p->offs_int.as<int>() + (p->stream_half ? 0 : off_elems);
    a.tile_counts = cur;
    if (p->counted_for != d_depth || p->counted_gen != p->params_gen) {
        // nothing (valid) was counted ahead for this batch: do it now, like mode 0
        a.offsets = off_cur;
        launch_count(p, vec, s, a);
        hipLaunchKernelGGL(scan_kernel, dim3((unsigned)p->n_ticks), dim3(kThreads), 0, s, cur, a.tiles_per_tick, a.frames, a.n_frames, off_cur);
    }
    LSN_HIP(hipMemcpyAsync(d_offsets, off_cur, sizeof(int) * off_elems, hipMemcpyDeviceToDevice, s));
    a.offsets = d_offsets;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (p->profile && next_event_pair(p, e0, e1)) return -1;
    if (e0) LSN_HIP(hipEventRecord(e0, s));
    if (d_next_depth) {
        a.depth_next = static_cast<const unsigned short *>(d_next_depth);
        a.tile_counts_next = nxt;
        launch<3>(vec, grid, s, a);
    } else {
        launch<1>(vec, grid, s, a);
    }
    if (e1) LSN_HIP(hipEventRecord(e1, s));
    if (d_next_depth) {
        hipLaunchKernelGGL(scan_kernel, dim3((unsigned)p->n_ticks), dim3(kThreads), 0, s, nxt, a.tiles_per_tick, a.frames, a.n_frames, off_nxt);
        p->counted_for = d_next_depth;
        p->counted_gen = p->params_gen;
        p->stream_half ^= 1;
    } else {
        p->counted_for = nullptr;
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

// Diagnostics / tests: builds the per-pixel depth thresholds now (if the plan uses them) and copies them out.
extern "C" int lsnFusionThresholds(LsnFusion *p, unsigned int *out_host, float *build_ms, void *stream)
{
    lsn::clear_error();
    if (!p || !p->params_set) {
        lsn::set_error("lsnFusionThresholds: no plan / lsnFusionSetParams has not been called");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    if (!p->thr_enabled) return 1;
    hipStream_t s = lsn::as_stream(stream);
    if (!p->thr_valid) {
        p->runs_with_params = 1;
        if (ensure_thresholds(p, s)) return -1;
    }
    if (build_ms) *build_ms = p->thr_build_ms;
    if (out_host) {
        LSN_HIP(hipMemcpyAsync(out_host, p->thr.p, sizeof(unsigned int) * (size_t)p->cap, hipMemcpyDeviceToHost, s));
        LSN_HIP(hipStreamSynchronize(s));
    }
    return 0;
}

extern "C" long long lsnFusionTickTriangleCapacity(const LsnFusion *p) { return p ? 2 * p->cap : 0; }

extern "C" int lsnFusionRunMesh(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets,
                                void *d_triangles, int *d_tri_offsets, void *stream)
{
    lsn::clear_error();
    if (!p || !d_triangles || !d_tri_offsets) {
        lsn::set_error("lsnFusionRunMesh: null argument");
        return -1;
    }
    {
        std::lock_guard<std::mutex> g(p->mu);
        LSN_HIP(hipSetDevice(p->device));
        if (p->pixmap.reserve(sizeof(int) * (size_t)p->cap * p->n_ticks) ||
            p->tri_counts.reserve(sizeof(int) * (size_t)p->tiles_per_tick * p->n_ticks) ||
            p->tri_codes.reserve(sizeof(unsigned int) * (size_t)p->tiles_per_tick * p->n_ticks * kThreads))
            return -1;
        p->want_pixmap = true;
    }
    // vertices + depth_to_vertices_map (count / scan / write launches)
    const int rc = lsnFusionRun(p, d_depth, d_colors, d_vertices, d_offsets, stream);
    std::lock_guard<std::mutex> g(p->mu);
    p->want_pixmap = false;
    if (rc) return rc;
    hipStream_t s = lsn::as_stream(stream);
    TriArgs t;
    t.frames = p->frames.as<FrameDesc>();
    t.tiles = p->tile_frame.as<TileDesc>();
    t.depth = static_cast<const unsigned short *>(d_depth);
    t.pixmap = p->pixmap.as<int>();
    t.tri = static_cast<int *>(d_triangles);
    t.tile_counts = p->tri_counts.as<int>();
    t.codes = p->tri_codes.as<unsigned int>();
    t.tiles_per_tick = p->tiles_per_tick;
    t.tick_pix_stride = p->cap;
    t.tick_tri_stride = 2 * p->cap;
    const bool vec = p->vec_ok && ((uintptr_t)d_depth & 15) == 0 && (p->tick_depth_elems % 8) == 0;
    const int grid = p->tiles_per_tick * p->n_ticks;
    if (vec) hipLaunchKernelGGL((tri_kernel<0, true>), dim3(grid), dim3(kThreads), 0, s, t);
    else     hipLaunchKernelGGL((tri_kernel<0, false>), dim3(grid), dim3(kThreads), 0, s, t);
    hipLaunchKernelGGL(scan_kernel, dim3((unsigned)p->n_ticks), dim3(kThreads), 0, s, t.tile_counts, t.tiles_per_tick, t.frames, p->n_maps,
                       d_tri_offsets);
    if (vec) hipLaunchKernelGGL((tri_kernel<1, true>), dim3(grid), dim3(kThreads), 0, s, t);
    else     hipLaunchKernelGGL((tri_kernel<1, false>), dim3(grid), dim3(kThreads), 0, s, t);
    LSN_HIP(hipGetLastError());
    return 0;
}

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
    hipLaunchKernelGGL(radial_close_kernel, dim3((unsigned)(p->n_maps * p->n_ticks)), dim3(rows), (sizeof(unsigned int) + sizeof(unsigned short)) * kRing * (rows + 2), s,
                       p->frames.as<FrameDesc>(), p->n_maps,
                       p->map_copy.as<unsigned short>(), p->colors_copy.as<unsigned char>(), p->cap);
    LSN_HIP(hipGetLastError());
    // :259-260 the corrected maps replace the inputs
    LSN_HIP(hipMemcpyAsync(d_depth, p->map_copy.p, 2 * npix, hipMemcpyDeviceToDevice, s));
    LSN_HIP(hipMemcpyAsync(d_colors, p->colors_copy.p, 3 * npix, hipMemcpyDeviceToDevice, s));
    return 0;
}

// Reads back the look-back error flag (diagnostics for tests); synchronises the stream.
extern "C" int lsnFusionLookbackFailed(LsnFusion *p, void *stream)
{
    if (!p) return -1;
    int flag = 0;
    LSN_HIP(hipSetDevice(p->device));
    LSN_HIP(hipStreamSynchronize(lsn::as_stream(stream)));
    LSN_HIP(hipMemcpy(&flag, p->misc.p, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) LSN_HIP(hipMemset(p->misc.p, 0, sizeof(int)));
    return flag;
}

extern "C" int lsnFusionTilesPerTick(const LsnFusion *p) { return p ? p->tiles_per_tick : 0; }

// Survivor exchange, sender side: count + scan as in lsnFusionRun, then the compact streams instead of vertices.
extern "C" int lsnFusionPackSurvivors(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_mask, void *d_depth_c, void *d_rgb_c,
                                      int *d_tile_prefix, int *d_offsets, void *stream)
{
    lsn::clear_error();
    if (!p || !d_depth || !d_colors || !d_mask || !d_depth_c || !d_rgb_c || !d_tile_prefix || !d_offsets) {
        lsn::set_error("lsnFusionPackSurvivors: null argument");
        return -1;
    }
    if (!p->params_set) {
        lsn::set_error("lsnFusionPackSurvivors: lsnFusionSetParams has not been called");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    hipStream_t s = lsn::as_stream(stream);
    const bool vec = p->vec_ok && ((uintptr_t)d_depth & 15) == 0 && ((uintptr_t)d_colors & 7) == 0 && (p->tick_depth_elems % 8) == 0;
    if (!vec) {
        lsn::set_error("lsnFusionPackSurvivors: needs frame widths that are multiples of 8 and 16-byte aligned buffers (exchange vertices instead)");
        return -1;
    }
    if (ensure_thresholds(p, s)) return -1;
    FuseArgs a;
    fill_args(p, a, d_depth, d_colors, nullptr, d_offsets);
    launch_count(p, true, s, a);
    hipLaunchKernelGGL(scan_kernel, dim3((unsigned)p->n_ticks), dim3(kThreads), 0, s, a.tile_counts, a.tiles_per_tick, a.frames, a.n_frames, a.offsets);
    PackArgs pk;
    pk.mask = static_cast<unsigned char *>(d_mask);
    pk.depth_c = static_cast<unsigned short *>(d_depth_c);
    pk.rgb_c = static_cast<unsigned char *>(d_rgb_c);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)(p->tiles_per_tick * p->n_ticks)), dim3(kThreads), 0, s, a, pk);
    LSN_HIP(hipGetLastError());
    LSN_HIP(hipMemcpyAsync(d_tile_prefix, p->tile_counts.p, sizeof(int) * (size_t)p->tiles_per_tick * p->n_ticks, hipMemcpyDeviceToDevice, s));
    return 0;
}

// Survivor exchange, receiver side: `all` is a plan over the WHOLE rig (every sensor, lsnFusionSetParams called with all
// parameters, same n_ticks); the gathered arrays hold n_shards equally shaped shards of maps_per_shard sensors each.
extern "C" int lsnFusionReconstruct(LsnFusion *all, int n_shards, int maps_per_shard, const void *d_masks, const void *d_depth_c,
                                    const void *d_rgb_c, long long slab, const int *d_tile_prefix, const int *d_shard_offsets,
                                    void *d_merged, int *d_merged_offsets, void *stream)
{
    lsn::clear_error();
    if (!all || !d_masks || !d_depth_c || !d_rgb_c || !d_tile_prefix || !d_shard_offsets || !d_merged || !d_merged_offsets) {
        lsn::set_error("lsnFusionReconstruct: null argument");
        return -1;
    }
    if (!all->params_set) {
        lsn::set_error("lsnFusionReconstruct: lsnFusionSetParams has not been called on the whole-rig plan");
        return -1;
    }
    std::lock_guard<std::mutex> g(all->mu);
    if (n_shards <= 0 || maps_per_shard <= 0 || n_shards * maps_per_shard != all->n_maps || slab <= 0) {
        lsn::set_error("lsnFusionReconstruct: %d shards x %d sensors do not make the plan's %d sensors", n_shards, maps_per_shard, all->n_maps);
        return -1;
    }
    for (int i = 1; i < all->n_maps; i++)
        if (all->w[i] != all->w[0] || all->h[i] != all->h[0]) {
            lsn::set_error("lsnFusionReconstruct: the survivor exchange needs identically sized sensors");
            return -1;
        }
    if (!all->vec_ok || (all->tick_depth_elems % 8) != 0 || ((uintptr_t)d_merged & 15) != 0) {
        lsn::set_error("lsnFusionReconstruct: needs frame widths that are multiples of 8 and a 16-byte aligned output");
        return -1;
    }
    LSN_HIP(hipSetDevice(all->device));
    FuseArgs a;
    fill_args(all, a, nullptr, nullptr, d_merged, d_merged_offsets);
    a.thr = nullptr;
    ReconArgs r;
    r.mask = static_cast<const unsigned char *>(d_masks);
    r.depth_c = static_cast<const unsigned short *>(d_depth_c);
    r.rgb_c = static_cast<const unsigned char *>(d_rgb_c);
    r.tile_prefix = d_tile_prefix;
    r.shard_off = d_shard_offsets;
    r.merged_off = d_merged_offsets;
    r.slab = slab;
    r.cap_loc = all->cap / n_shards;
    r.tiles_loc = all->tiles_per_tick / n_shards;
    r.n_shards = n_shards;
    r.maps_per_shard = maps_per_shard;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (all->profile) {
        if (next_event_pair(all, e0, e1)) return -1;
        all->timed_kernel = "recon_kernel";
        LSN_HIP(hipEventRecord(e0, lsn::as_stream(stream)));
    }
    hipLaunchKernelGGL(recon_kernel, dim3((unsigned)(all->tiles_per_tick * all->n_ticks)), dim3(kThreads), 0, lsn::as_stream(stream), a, r);
    if (e1) LSN_HIP(hipEventRecord(e1, lsn::as_stream(stream)));
    LSN_HIP(hipGetLastError());
    return 0;
}

extern "C" int lsnMergeShards(int device, int n_shards, int n_ticks, int maps_per_shard, const void *d_shards, long long shard_cap,
                              const int *d_shard_offsets, void *d_merged, long long merged_cap, int *d_merged_offsets, void *stream)
{
    lsn::clear_error();
    if (n_shards <= 0 || n_ticks <= 0 || maps_per_shard <= 0 || maps_per_shard >= kThreads || !d_shards || !d_shard_offsets || !d_merged ||
        !d_merged_offsets || shard_cap <= 0 || merged_cap < shard_cap) {
        lsn::set_error("lsnMergeShards: bad arguments");
        return -1;
    }
    if ((long long)n_shards * n_ticks > 65535) {
        lsn::set_error("lsnMergeShards: n_shards * n_ticks must not exceed 65535");
        return -1;
    }
    LSN_HIP(hipSetDevice(device));
    MergeArgs a;
    a.shards = static_cast<const uint4 *>(d_shards);
    a.shard_off = d_shard_offsets;
    a.merged = static_cast<uint4 *>(d_merged);
    a.merged_off = d_merged_offsets;
    a.shard_cap = shard_cap;
    a.merged_cap = merged_cap;
    a.n_shards = n_shards;
    a.n_ticks = n_ticks;
    a.maps_per_shard = maps_per_shard;
    long long chunks = (shard_cap + kThreads * 8 - 1) / (kThreads * 8);
    if (chunks > 256) chunks = 256;
    hipLaunchKernelGGL(merge_shards_kernel, dim3((unsigned)chunks, (unsigned)(n_shards * n_ticks)), dim3(kThreads), 0, lsn::as_stream(stream), a);
    LSN_HIP(hipGetLastError());
    return 0;
}
