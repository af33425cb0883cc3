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
//
// Files: fusion_shared.hpp holds the building blocks every translation unit of the plan uses (geometry structs, the
// per-pixel arithmetic, tile loading, ranking, LDS staging, scan_kernel, the LsnFusion plan object); this file holds the
// count / write / streamed / look-back kernels, the per-pixel depth thresholds and the plan's entry points; mesh.hip the
// triangulation, radial.hip the radial correction, exchange.hip the multi-GPU exchange step.


#include "fusion_shared.hpp"

namespace {

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

// Look-back status word: bits 63..62 flag (0 = empty, 1 = run aggregate, 2 = inclusive prefix), low 32 bits value.
// One naturally aligned 8-byte word carries flag AND value, written by one agent-scope (sc1, write-through) store and
// polled with agent-scope loads: nothing else is handed off, so no fence is needed and the result cannot depend on
// where the producing workgroup ran.
constexpr unsigned long long kFlagAggregate = 1ull << 62;
constexpr unsigned long long kFlagPrefix = 2ull << 62;
constexpr int kSpinLimit = 1 << 20;

// Mode 2 (single pass, one look-back per TILE) tags its words instead of clearing them: bits 63..34 = the launch's epoch,
// 33..32 = flag, low 32 bits = value; a word of another epoch reads as "empty".
constexpr int kEpochShift = 34;
#ifndef LSN_LOOK_SLOTS
#define LSN_LOOK_SLOTS 1
#endif
#ifndef LSN_LOOK_AGENT_ONLY
#define LSN_LOOK_AGENT_ONLY 0
#endif
constexpr int kLookSlots = LSN_LOOK_SLOTS;   // predecessor words per lane and round trip of the look-back (x 64 lanes); build-time for A/B runs
constexpr unsigned long long kTileAggregate = 1ull << 32, kTilePrefix = 2ull << 32;

// Exclusive prefix of this tile inside its tick, by decoupled look-back (wave 0 of the workgroup; 64 predecessors per poll).
// Relies on workgroups starting in blockIdx order (a tile only ever waits for lower-numbered tiles of its tick, and the
// lowest unfinished workgroup is always resident); bounded by kSpinLimit so that a launch always drains: returns -1, raises
// error flag 1 and leaves a poisoned word (flag 3: everybody behind gives up at once) when it gives up.
__device__ __forceinline__ int tile_lookback(const FuseArgs &a, int tick, int tile, int tile_tot, int lane)
{
    unsigned long long *st = a.run_state + (long long)tick * a.tiles_per_tick;
    const unsigned long long tag = (unsigned long long)a.epoch << kEpochShift;
    if (lane == 0)
        __hip_atomic_store(&st[tile], tag | (tile == 0 ? kTilePrefix : kTileAggregate) | (unsigned int)tile_tot, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    int acc = 0;
    int pos = tile - 1 - lane;  // lane 0 looks at the nearest predecessor
    bool done = tile == 0;
    int spins = 0;
    static_assert(kLookSlots >= 1 && kLookSlots <= 16, "LSN_LOOK_SLOTS");
    const bool near_polls = a.n_ticks > 1 && !LSN_LOOK_AGENT_ONLY;
    while (!done) {
        // kLookSlots x 64 predecessors per round trip: the words of all slots are in flight together, then slot after slot is evaluated like
        // one 64-wide poll.  Built for one-tick launches, where every tile is resident, publishes its aggregate at about the same time and a
        // tile far down the tick sums hundreds of aggregates in dependent rounds of 64 -- and measured SLOWER the wider the window (8 x 512x424,
        // one tick: 16.1 us with 1 slot, 16.0 with 4, 18.1 with 8, 19.0 with 16; 16 x 1024x1024: 85 / 104 / 128 / 143 us; 64 ticks: 359 / 370 /
        // 416 / 507 us -- profiles/r05_ab_lookback.txt): the polls are uncached 8-byte loads and their number, not the depth of the chain, is
        // what costs.  The default stays 1 slot; -DLSN_LOOK_SLOTS=n rebuilds the A/B.
        unsigned long long ws[kLookSlots];
#pragma unroll
        for (int j = 0; j < kLookSlots; j++) {
            const int q = pos - 64 * j;
            ws[j] = tag | kTilePrefix;  // before the first tile: prefix 0
            if (q >= 0) {
                // most polls take the short way (sc0: past the CU's L1 only; the word is in this XCD's L2 when the producer ran on
                // this XCD); every fourth goes to memory (agent scope), so a producer on another XCD is seen as well -- a stale L2
                // line can only read as "empty" (epoch tag).  A one-tick launch spreads the tick's tiles over all XCDs: every poll goes to memory
                if (near_polls && (spins & 3) != 3) ws[j] = __hip_atomic_load(&st[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else ws[j] = __hip_atomic_load(&st[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        bool again = false;
#pragma unroll
        for (int j = 0; j < kLookSlots; j++) {
            if (done || again) break;
            const unsigned long long w = ws[j];
            const unsigned int flag = (w >> kEpochShift) == a.epoch ? (unsigned int)(w >> 32) & 3u : 0u;
            const unsigned long long pref = __ballot(flag == 2);
            const int first = pref ? __ffsll((long long)pref) - 1 : 63;
            const unsigned long long relevant = first >= 63 ? ~0ull : ((2ull << first) - 1ull);
            const bool poisoned = (__ballot(flag == 3) & relevant) != 0;
            if (poisoned || (__ballot(flag == 0) & relevant) != 0) {
                if (poisoned || ++spins > kSpinLimit) {
                    if (lane == 0) {
                        atomicExch(a.error_flag, 1);
                        if (a.offsets_mirror) a.offsets_mirror[a.n_frames + 1] = 1;
                        __hip_atomic_store(&st[tile], tag | (3ull << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    return -1;
                }
                __builtin_amdgcn_s_sleep(1);
                pos -= 64 * j;   // the slots before this one are summed: the next round starts here
                again = true;
                break;
            }
            acc += wave_sum(lane <= first ? (int)(unsigned int)w : 0);
            if (pref) done = true;
        }
        if (!done && !again) pos -= 64 * kLookSlots;
    }
    if (lane == 0 && tile != 0)
        __hip_atomic_store(&st[tile], tag | kTilePrefix | (unsigned int)(acc + tile_tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return acc;
}

// ---- mode 0: count kernel, scan kernel, write kernel ------------------------------------------------------------

// MODE 0 = count only (writes tile_counts), 1 = write with offsets from tile_counts (exclusive prefixes by then),
// 3 = streamed: mode 1 for this batch AND the count of the same tile of the NEXT batch (depth_next) in one workgroup --
// the count pass is VALU-bound, the write pass HBM-bound, and inside one kernel they share every CU all the time.
// LAZY (write pass only): the colours are loaded after the keep predicates are known, by the lanes that kept a pixel --
// spatially coherent frames (real scenes: background beyond the crop box, invalid regions) then never fetch the colour
// lines of rejected areas; the price is that the colour load no longer flies together with the depth load.
// HOST (write pass of the two-pass form only): `out` is pinned host memory -- plain, destination-aligned stores (stage_and_store's note).  A
// template parameter and not a run-time flag, so that the device-resident instantiations keep their code (and their streaming stores: a
// run-time choice between a streaming and a plain store is folded into one plain store, mesh.hip); mode 4 has the run-time a.host_out.
template <int MODE, bool VEC, bool LAZY = false, bool HOST = false>
__global__ __launch_bounds__(kThreads) void fuse_kernel(const FuseArgs a)
{
    constexpr bool kWrite = MODE != 0;
    __shared__ uint4 stage[kWrite ? kStageSlots : 1];
    __shared__ int s_wave_tot[4];
    __shared__ int s_wave_next[4];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // mode 4 walks the ticks fastest: the ~2000 resident workgroups then are ~30 consecutive tiles of EVERY tick, i.e. as many
    // independent look-back chains as ticks (tick-major order keeps 2-3 chains alive and the launch crawls at their pace:
    // 0.54 ms instead of 0.25); with a tick count that is a multiple of 8 a tick's tiles also share an XCD
    int tick = blockIdx.x / a.tiles_per_tick;
    int tile = blockIdx.x - tick * a.tiles_per_tick;
    if (kWrite && a.chunk > 0) {
        const int per_chunk = a.chunk * a.n_ticks;
        const int chunk = blockIdx.x / per_chunk;
        const int r = blockIdx.x - chunk * per_chunk;
        const int width = min(a.chunk, a.tiles_per_tick - chunk * a.chunk);   // the last chunk may be narrower
        tick = r / width;
        tile = chunk * a.chunk + (r - tick * width);
    }
    tile += a.tile0;   // a launch over a group of sensors of a one-tick plan (run_frames); 0 everywhere else
    // The write pass takes the ticks last to first: what the count pass read last is still in the Infinity Cache (64 ticks of depth are
    // 222 MB against 256 MB of cache, and in the forward order the colours streaming through evict exactly the depth that is needed
    // next).  Worth 0-2 % of the step, never less than nothing (A/B/B/A on three boxes: 0.3007 -> 0.2951 / 0.2966 ms, 0.2946 -> 0.2921, 0.2976 against 0.2973;
    // $LSN_WRITE_FORWARD=1 undoes it).
    if (MODE == 1 && a.reverse_ticks) tick = a.n_ticks - 1 - tick;
    const int lin = tick * a.tiles_per_tick + tile;

    const Tile t = locate(a, tick, tile);
    Inputs in;
    load_inputs<VEC, kWrite && !LAZY>(t, in);
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
    if (kWrite && LAZY) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) any |= keep[k];
        if (any) {
            load_rgb<VEC>(t, in);
#pragma unroll
            for (int k = 0; k < kPxPerLane; k++) vert[k].x = rgba_of(in, k);
        }
    }
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
    int base = 0, counted = 0;
    if (MODE == 1 || MODE == 3) {
        base = a.tile_counts[lin];  // scan_kernel left the exclusive prefix inside the tick here
        // what the count pass saw in this tile = the next tile's prefix (the tick's total after the last tile) minus this one
        counted = (tile + 1 < a.tiles_per_tick ? a.tile_counts[lin + 1] : a.offsets[tick * (a.n_frames + 1) + a.n_frames]) - base;
    }
    __syncthreads();
    int wave_off = 0, tile_tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = s_wave_tot[i];
        if (i < wave) wave_off += v;
        tile_tot += v;
    }
    if (MODE == 0) {
        if (threadIdx.x == 0) a.tile_counts[lin] = tile_tot;
        return;
    }
    if (MODE == 3 && threadIdx.x == 0) a.tile_counts_next[lin] = s_wave_next[0] + s_wave_next[1] + s_wave_next[2] + s_wave_next[3];
    if (MODE == 4) {
        // single pass: the vertices wait in registers while wave 0 resolves the tile's offset from its predecessors
        if (wave == 0) {
            const int b = tile_lookback(a, tick, tile, tile_tot, lane);
            if (lane == 0) s_base = b;
        }
        __syncthreads();
        base = s_base;
        if (base < 0) {
            // the look-back gave up (this tile or one before it): nothing of the tile is written, and the tick's vertex count -- which every
            // consumer reads first -- says so instead of keeping the previous call's value (every tile behind a poisoned one gives up as
            // well, so the tick's last tile always gets here)
            if (threadIdx.x == 0 && tile == a.tiles_per_tick - 1) {
                a.offsets[tick * (a.n_frames + 1) + a.n_frames] = -1;
                if (a.offsets_mirror) a.offsets_mirror[a.n_frames] = -1;
            }
            return;
        }
        if (threadIdx.x == 0) {   // the per-sensor offsets table scan_kernel writes in mode 0
            int *off = a.offsets + tick * (a.n_frames + 1);
            if (t.frame_start) off[t.f] = base;
            if (tile == a.tiles_per_tick - 1) off[a.n_frames] = base + tile_tot;
            if (a.offsets_mirror) {
                if (t.frame_start) a.offsets_mirror[t.f] = base;
                if (tile == a.tiles_per_tick - 1) a.offsets_mirror[a.n_frames] = base + tile_tot;
            }
            if (a.group_end_mirror && blockIdx.x == gridDim.x - 1) *a.group_end_mirror = base + tile_tot;
        }
    }
    if (MODE != 4 && tile_tot != counted) {
        // The inputs are not the ones the count pass read (a caller refilled a buffer that had been counted ahead by
        // lsnFusionRunStreamed, or overwrote the inputs of a call in flight): the prefixes no longer fit, tiles would
        // overlap and the last one would run past its tick's slab.  Write nothing, raise the flag (lsnFusionCheck).
        if (threadIdx.x == 0) atomicExch(a.error_flag, 2);
        return;
    }
    if (VEC ? a.pm_first != nullptr : a.pixmap != nullptr) {
        // depth_to_vertices_map (depthprocessing.cpp:166), already rebased to the tick's merged cloud like formMesh
        // rebases triangle indices (:1614-1626): what the triangulation pass reads.  A lane's 8 pixels get consecutive indices,
        // so the map is the lane's first index + an 8-bit mask (5 bytes per 8 pixels instead of 32); rigs the lanes do not fit
        // (widths that are not multiples of 8) keep one int per pixel.
        const int p0 = t.px0 + threadIdx.x * kPxPerLane;
        if (p0 < t.npix) {
            int r = base + wave_off + below;
            if (VEC) {
                unsigned int m = 0;
#pragma unroll
                for (int k = 0; k < kPxPerLane; k++) m |= keep[k] ? (1u << k) : 0u;
                const long long g = (tick * a.tick_depth_stride + t.pix_base + p0) >> 3;
                a.pm_first[g] = r;
                a.pm_mask[g] = (unsigned char)m;
            } else {
                int *pm = a.pixmap + tick * a.tick_depth_stride + t.pix_base + p0;
#pragma unroll
                for (int k = 0; k < kPxPerLane; k++) {
                    if (p0 + k < t.npix) pm[k] = keep[k] ? r : -1;
                    r += keep[k] ? 1 : 0;
                }
            }
        }
    }
    stage_and_store(stage, keep, vert, wave_off + below, tile_tot, a.out + tick * a.tick_vert_stride + base, HOST || (MODE == 4 && a.host_out != 0));
}

// ---- mode 1: one launch, runs of tiles with a decoupled look-back per run ---------------------------------------


template <bool VEC>
__global__ __launch_bounds__(kThreads) void run_kernel(const FuseArgs a)
{
    __shared__ uint4 stage[kStageSlots];
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
}  // namespace

// -------------------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------------------

static LsnFusion * lsnFusionCreate_impl(int device, int n_ticks, int n_maps, const int *widths, const int *heights)
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
    if (const char *env = getenv("LSN_LAZY_RGB")) p->lazy_rgb = atoi(env) != 0;

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
        p->tile_start.push_back(tiles);
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
    // one-tick plans of up to 2048 tiles take the single pass (run_locked); $LSN_ONE_TICK_SINGLE_PASS=0 / 1 forces the three launches / the single pass
    p->one_tick_single_pass = n_ticks == 1 && tiles <= 2048;
    if (const char *env = getenv("LSN_ONE_TICK_SINGLE_PASS")) p->one_tick_single_pass = atoi(env) != 0;
    p->tile_start.push_back(tiles);
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

extern "C" LsnFusion * lsnFusionCreate(int device, int n_ticks, int n_maps, const int *widths, const int *heights)
{
    return lsn::guarded<LsnFusion *>("lsnFusionCreate", static_cast<LsnFusion *>(nullptr), [&]() { return lsnFusionCreate_impl(device, n_ticks, n_maps, widths, heights); });
}

static void lsnFusionDestroy_impl(LsnFusion *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (auto &e : p->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (p->radial_done) (void)hipEventDestroy(p->radial_done);
    if (p->side) {
        (void)hipStreamSynchronize(p->side);
        (void)hipStreamDestroy(p->side);
        (void)hipEventDestroy(p->ev_counted);
        (void)hipEventDestroy(p->ev_written[0]);
        (void)hipEventDestroy(p->ev_written[1]);
    }
    delete p;
}

extern "C" void lsnFusionDestroy(LsnFusion *p)
{
    lsn::guarded_void("lsnFusionDestroy", [&]() { lsnFusionDestroy_impl(p); });
}

extern "C" long long lsnFusionTickCapacity(const LsnFusion *p) { return p ? p->cap : 0; }

// IntrinsicCameraParameters(float*) / WorldTranformation(float*), include/NativeUtils/depthprocessing.h:56-63,96-97: the caller's 7 and 12
// floats of one sensor as the kernels read them.  The one place the unpack order is written down (tests/test_abi.py holds it against
// the reference's own constructors through lsnPackSensorParams).
static void pack_sensor_params(const float *ip, const float *tp, SensorParams &s)
{
    s.cx = ip[0]; s.cy = ip[1]; s.fx = ip[2]; s.fy = ip[3];
    s.t0 = tp[0]; s.t1 = tp[1]; s.t2 = tp[2];
    s.r00 = tp[3]; s.r01 = tp[4]; s.r02 = tp[5];
    s.r10 = tp[6]; s.r11 = tp[7]; s.r12 = tp[8];
    s.r20 = tp[9]; s.r21 = tp[10]; s.r22 = tp[11];
}

static int lsnPackSensorParams_impl(const float *intr7, const float *wt12, float *out16)
{
    lsn::clear_error();
    if (!intr7 || !wt12 || !out16) {
        lsn::set_error("lsnPackSensorParams: null argument");
        return -1;
    }
    SensorParams s;
    pack_sensor_params(intr7, wt12, s);
    static_assert(sizeof(SensorParams) == 16 * sizeof(float), "SensorParams is 16 floats");
    memcpy(out16, &s, sizeof(s));
    return 0;
}

extern "C" int lsnPackSensorParams(const float *intr7, const float *wt12, float *out16)
{
    return lsn::guarded<int>("lsnPackSensorParams", static_cast<int>(-1), [&]() { return lsnPackSensorParams_impl(intr7, wt12, out16); });
}

static int lsnFusionSetParams_impl(LsnFusion *p, const float *intr, const float *wt, const float *bounds6, void *stream)
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
    for (int i = 0; i < p->n_maps; i++) pack_sensor_params(intr + 7 * i, wt + 12 * i, sp[i]);
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

extern "C" int lsnFusionSetParams(LsnFusion *p, const float *intr, const float *wt, const float *bounds6, void *stream)
{
    return lsn::guarded<int>("lsnFusionSetParams", static_cast<int>(-1), [&]() { return lsnFusionSetParams_impl(p, intr, wt, bounds6, stream); });
}

static int lsnFusionSetMode_impl(LsnFusion *p, int mode)
{
    if (!p || mode < 0 || mode > 2) {
        lsn::set_error("lsnFusionSetMode: mode must be 0 (count/scan/write launches), 1 (single launch, runs + look-back) or 2 (single pass, look-back per tile)");
        return -1;
    }
    p->mode = mode;
    return 0;
}

extern "C" int lsnFusionSetMode(LsnFusion *p, int mode)
{
    return lsn::guarded<int>("lsnFusionSetMode", static_cast<int>(-1), [&]() { return lsnFusionSetMode_impl(p, mode); });
}

static int lsnFusionSetPipelined_impl(LsnFusion *p, int enable)
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

extern "C" int lsnFusionSetPipelined(LsnFusion *p, int enable)
{
    return lsn::guarded<int>("lsnFusionSetPipelined", static_cast<int>(-1), [&]() { return lsnFusionSetPipelined_impl(p, enable); });
}

static int lsnFusionProfile_impl(LsnFusion *p, int enable)
{
    if (!p) return -1;
    p->profile = enable != 0;
    p->profile_every = enable > 1 ? enable : 1;
    p->profile_seq = 0;
    return 0;
}

extern "C" int lsnFusionProfile(LsnFusion *p, int enable)
{
    return lsn::guarded<int>("lsnFusionProfile", static_cast<int>(-1), [&]() { return lsnFusionProfile_impl(p, enable); });
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

static int lsnFusionKernelStats_impl(LsnFusion *p, double *avg_ms, long long *launches, char *name, int name_len, int reset)
{
    lsn::clear_error();
    if (!p) return -1;
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    if (drain_events(p)) return -1;
    if (avg_ms) *avg_ms = p->launches ? p->acc_ms / (double)p->launches : 0.0;
    if (launches) *launches = p->launches;
    if (name && name_len > 0)
        snprintf(name, (size_t)name_len, "%s", p->timed_kernel ? p->timed_kernel : (p->mode == 0 ? "fuse_kernel<1>" : p->mode == 2 ? "fuse_kernel<4>" : "run_kernel"));
    if (reset) {
        p->acc_ms = 0;
        p->launches = 0;
    }
    return 0;
}

extern "C" int lsnFusionKernelStats(LsnFusion *p, double *avg_ms, long long *launches, char *name, int name_len, int reset)
{
    return lsn::guarded<int>("lsnFusionKernelStats", static_cast<int>(-1), [&]() { return lsnFusionKernelStats_impl(p, avg_ms, launches, name, name_len, reset); });
}

// Kernel arguments of one call (everything but the per-mode scratch selection).
void lsn::fill_args(LsnFusion *p, FuseArgs &a, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets)
{
    a.epoch = 0;
    a.chunk = 0;
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
    a.pm_first = p->want_pixmap ? p->pm_first.as<int>() : nullptr;
    a.pm_mask = p->want_pixmap ? p->pm_mask.as<unsigned char>() : nullptr;
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
    a.tile0 = 0;
    {
        static const int fwd = getenv("LSN_WRITE_FORWARD") ? atoi(getenv("LSN_WRITE_FORWARD")) : 0;
        a.reverse_ticks = fwd ? 0 : 1;
    }
    a.host_out = 0;
    a.offsets_mirror = nullptr;
    a.group_end_mirror = nullptr;
}

// Called at the top of every run (p->mu held): from the second run with the same parameters on, the count pass uses the
// per-pixel depth thresholds; they are built here, once, on the caller's stream.
int lsn::ensure_thresholds(LsnFusion *p, hipStream_t s)
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
void lsn::launch_count(LsnFusion *p, bool vec, hipStream_t s, const FuseArgs &a)
{
    if (a.thr) {
        static const int tune = getenv("LSN_TICK_GROUP") ? atoi(getenv("LSN_TICK_GROUP")) : 0;
        int G = tune ? tune : (a.n_ticks >= 8 ? 8 : (a.n_ticks >= 4 ? 4 : 1));
        if (G != 16 && G != 8 && G != 4 && G != 2) G = 1;
        const int grid = p->tiles_per_tick * ((a.n_ticks + G - 1) / G);
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
        const int grid = p->tiles_per_tick * a.n_ticks;
        if (vec) hipLaunchKernelGGL((fuse_kernel<0, true>), dim3(grid), dim3(kThreads), 0, s, a);
        else     hipLaunchKernelGGL((fuse_kernel<0, false>), dim3(grid), dim3(kThreads), 0, s, a);
    }
}

// Next HIP-event pair of the dominant-kernel timer (profiling on).
int lsn::next_event_pair(LsnFusion *p, hipEvent_t &e0, hipEvent_t &e1)
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
static void launch(bool vec, int grid, hipStream_t s, const FuseArgs &a, bool lazy_rgb = false)
{
    if ((MODE == 1 || MODE == 4) && lazy_rgb) {
        if (vec) hipLaunchKernelGGL((fuse_kernel<MODE, true, true>), dim3(grid), dim3(kThreads), 0, s, a);
        else     hipLaunchKernelGGL((fuse_kernel<MODE, false, true>), dim3(grid), dim3(kThreads), 0, s, a);
        return;
    }
    if (vec) hipLaunchKernelGGL((fuse_kernel<MODE, true>), dim3(grid), dim3(kThreads), 0, s, a);
    else     hipLaunchKernelGGL((fuse_kernel<MODE, false>), dim3(grid), dim3(kThreads), 0, s, a);
}

static int lsnFusionRun_impl(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets,
                            void *stream)
{
    lsn::clear_error();
    if (!p || !d_depth || !d_colors || !d_vertices || !d_offsets) {
        lsn::set_error("lsnFusionRun: null argument");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    return lsn::run_locked(p, d_depth, d_colors, d_vertices, d_offsets, lsn::as_stream(stream), false, nullptr);
}

extern "C" int lsnFusionRun(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets,
                            void *stream)
{
    return lsn::guarded<int>("lsnFusionRun", static_cast<int>(-1), [&]() { return lsnFusionRun_impl(p, d_depth, d_colors, d_vertices, d_offsets, stream); });
}

int lsn::run_hooked(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets, hipStream_t s, const RunHooks *hooks)
{
    if (!p || !d_depth || !d_colors || !d_vertices || !d_offsets) {
        lsn::set_error("lsnFusionRun: null argument");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    return lsn::run_locked(p, d_depth, d_colors, d_vertices, d_offsets, s, false, hooks);
}

int lsn::run_locked(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets, hipStream_t s, bool with_pixmap,
                    const RunHooks *hooks)
{
    if (!p->params_set) {
        lsn::set_error("lsnFusionRun: lsnFusionSetParams has not been called");
        return -1;
    }
    if (((uintptr_t)d_vertices & 15) != 0) {
        lsn::set_error("lsnFusionRun: d_vertices must be 16-byte aligned");
        return -1;
    }
    LSN_HIP(hipSetDevice(p->device));
    struct PixmapScope {   // a.pixmap follows p->want_pixmap (fill_args); never left set behind an early return
        LsnFusion *p;
        PixmapScope(LsnFusion *q, bool on) : p(q) { p->want_pixmap = on; }
        ~PixmapScope() { p->want_pixmap = false; }
    } scope(p, with_pixmap);

    if (ensure_thresholds(p, s)) return -1;
    FuseArgs a;
    fill_args(p, a, d_depth, d_colors, d_vertices, d_offsets);

    // the wide-load path also needs 16-B aligned buffers and every tick to start 16-B / 8-B aligned
    const bool vec = p->vec_ok && ((uintptr_t)d_depth & 15) == 0 && ((uintptr_t)d_colors & 7) == 0 &&
                     (p->tick_depth_elems % 8) == 0;
    if (with_pixmap) p->pixmap_compact = vec;   // the triangulation reads the form this run writes
    const int grid = p->tiles_per_tick * p->n_ticks;
    const size_t off_bytes = sizeof(int) * (size_t)p->n_ticks * (p->n_maps + 1);

    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed_launch(p) && next_event_pair(p, e0, e1)) return -1;

    // One-tick plans (what a live device-resident caller holds: one merge per call): count -> scan -> write is three dependent launches
    // around ~5 us of work -- 13.4-15.8 us per call whatever the rig, the launches' own latency -- the single pass one launch whose
    // look-back chain grows with the tick's tiles: 1 x 512x424 (106 tiles) 7.6-8.0 us, 8 x 512x424 (848) 13.5-13.6, 2 x 1024x1024 (2048)
    // 14.2, 16 x 1024x1024 (16384) 82 against 55 (tools/one_tick_driver.py; measured WITHOUT the kernel-timing events, whose two
    // records per call had hidden the difference: 21.4 against 20.0-21.6, profiles/r05_ab_lookback.txt).  So a one-tick plan of up to
    // 2048 tiles takes the single pass (lsnFusionCreate), a bigger one the three launches; $LSN_ONE_TICK_SINGLE_PASS=0 / 1 forces either.
    const bool single_pass = (p->mode == 2 && !with_pixmap && !hooks) || (p->mode == 0 && p->n_ticks == 1 && !hooks && !p->pipelined && p->one_tick_single_pass);
    if (p->mode == 0) p->timed_kernel = single_pass ? "fuse_kernel<4>" : nullptr;   // what the event pair below brackets

    if (p->pipelined && p->mode == 0 && !with_pixmap && !hooks) {
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
        hipLaunchKernelGGL(scan_kernel, dim3((unsigned)p->n_ticks), dim3(kScanThreads), 0, p->side, ac.tile_counts, ac.tiles_per_tick, ac.frames,
                           ac.n_frames, off_int, nullptr);
        LSN_HIP(hipEventRecord(p->ev_counted, p->side));
        LSN_HIP(hipStreamWaitEvent(s, p->ev_counted, 0));
        LSN_HIP(hipMemcpyAsync(d_offsets, off_int, sizeof(int) * off_elems, hipMemcpyDeviceToDevice, s));
        if (e0) LSN_HIP(hipEventRecord(e0, s));
        launch<1>(vec, grid, s, a);
        if (e1) LSN_HIP(hipEventRecord(e1, s));
        LSN_HIP(hipEventRecord(p->ev_written[b], s));
        p->calls++;
    } else if (!single_pass && (p->mode == 0 || with_pixmap || hooks)) {
        launch_count(p, vec, s, a);
        hipLaunchKernelGGL(scan_kernel, dim3((unsigned)p->n_ticks), dim3(kScanThreads), 0, s, a.tile_counts, a.tiles_per_tick, a.frames, a.n_frames,
                           a.offsets, nullptr);
        if (hooks && hooks->h_offsets) LSN_HIP(hipMemcpyAsync(hooks->h_offsets, d_offsets, off_bytes, hipMemcpyDeviceToHost, s));
        if (hooks && hooks->counted) LSN_HIP(hipEventRecord(hooks->counted, s));
        if (hooks && hooks->colours_ready) LSN_HIP(hipStreamWaitEvent(s, hooks->colours_ready, 0));
        if (e0) LSN_HIP(hipEventRecord(e0, s));
        a.chunk = 0;   // tick-major block order (walking the ticks fastest was measured slower: DESIGN.md section 4, mode 2)
        launch<1>(vec, grid, s, a, p->lazy_rgb);
        if (e1) LSN_HIP(hipEventRecord(e1, s));
        if (hooks && hooks->written) LSN_HIP(hipEventRecord(hooks->written, s));
    } else if (single_pass) {
        // single pass: per-tile look-back words tagged with the launch's epoch (30 bits; the words are cleared when it wraps)
        if (p->epoch == 0 || p->epoch >= (1u << 30) - 1) {
            LSN_HIP(hipMemsetAsync(p->tile_state.p, 0, sizeof(unsigned long long) * (size_t)grid, s));
            p->epoch = 0;
        }
        a.epoch = ++p->epoch;
        // block order: chunks of 16 consecutive tiles, every tick's chunk before the next chunk -- the ~2000 resident workgroups
        // then belong to as many look-back chains as there are ticks (measured, 64 ticks x 848 tiles: tick-major 0.54 ms, ticks
        // fastest 0.40, chunks of 4 / 16 / 64 tiles 0.335 / 0.328 / 0.41)
        static const int chunk = getenv("LSN_FUSE_CHUNK") ? atoi(getenv("LSN_FUSE_CHUNK")) : 16;
        a.chunk = std::max(0, chunk);
        if (e0) LSN_HIP(hipEventRecord(e0, s));
        launch<4>(vec, grid, s, a, p->lazy_rgb);
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

// A one-tick plan fused group by group: frames [f0, f1) in ONE launch, single pass (a tile computes its vertices once, publishes
// its count and resolves its offset by look-back over the tiles before it -- fuse_kernel<4>, the mode-2 kernel), so the groups of
// a tick can be launched one after the other as their frames arrive, each continuing where the previous one stopped: the tiles of
// a later launch find the inclusive prefixes of the earlier launches' tiles in place (same epoch).  This is the form the host
// exports use (host_flows.hip): their output block is pinned host memory, the launch is bound by the PCIe link, and any further kernel
// boundary -- a separate count, a scan -- is time in which nothing crosses it.  first_of_tick starts a tick (new epoch).
// offsets_mirror (optional, pinned host memory, n_maps + 2 ints): the offset table as the tiles resolve it, then a give-up flag;
// group_end_mirror (optional, pinned): where this launch's vertices end inside the tick; host_out: d_vertices is pinned host memory.
int lsn::run_frames(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets, int f0, int f1, bool first_of_tick,
                    bool with_pixmap, int *offsets_mirror, int *group_end_mirror, bool host_out, hipStream_t s)
{
    if (!p || !d_depth || !d_colors || !d_vertices || !d_offsets || f0 < 0 || f1 > p->n_maps || f0 >= f1 || p->n_ticks != 1) {
        lsn::set_error("run_frames: bad arguments");
        return -1;
    }
    if (!p->params_set) {
        lsn::set_error("run_frames: lsnFusionSetParams has not been called");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    const bool vec = p->vec_ok && ((uintptr_t)d_depth & 15) == 0 && ((uintptr_t)d_colors & 7) == 0 && (p->tick_depth_elems % 8) == 0;
    if (with_pixmap) {
        if (vec ? (p->pm_first.reserve(sizeof(int) * ((size_t)p->cap / 8 + 2)) || p->pm_mask.reserve((size_t)p->cap / 8 + 2))
                : p->pixmap.reserve(sizeof(int) * (size_t)p->cap))
            return -1;
        p->pixmap_compact = vec;
    }
    struct PixmapScope {
        LsnFusion *p;
        PixmapScope(LsnFusion *q, bool on) : p(q) { p->want_pixmap = on; }
        ~PixmapScope() { p->want_pixmap = false; }
    } scope(p, with_pixmap);
    FuseArgs a;
    fill_args(p, a, d_depth, d_colors, d_vertices, d_offsets);
    if (first_of_tick) {
        if (p->epoch == 0 || p->epoch >= (1u << 30) - 1) {
            LSN_HIP(hipMemsetAsync(p->tile_state.p, 0, sizeof(unsigned long long) * (size_t)p->tiles_per_tick, s));
            p->epoch = 0;
        }
        ++p->epoch;
    }
    a.epoch = p->epoch;
    a.chunk = 0;
    a.tile0 = p->tile_start[f0];
    a.offsets_mirror = offsets_mirror;
    a.group_end_mirror = group_end_mirror;
    a.host_out = host_out ? 1 : 0;
    launch<4>(vec, p->tile_start[f1] - p->tile_start[f0], s, a, p->lazy_rgb);
    LSN_HIP(hipGetLastError());
    return 0;
}

// The two halves of the two-pass form of a ONE-TICK plan, for a caller that has to know the tick's vertex count before it can say where
// the vertices go: the sensor blocks of a call sharded over several devices (host_flows.hip) -- device d's vertices start where the devices
// before it end, inside ONE pinned host block.  run_count: count pass (depth only: it can run while the colours are still on their
// way up) + scan, the offset table also stored to `offsets_mirror` (pinned), `counted` recorded behind it.  run_write: the write pass
// at the scanned offsets, vertices to `vertices` -- pinned host memory when host_out (plain, destination-aligned stores).  The two
// calls must see the same buffers (d_colors decides the wide-load form in both); the plan's prefixes link them, so nothing else may
// run on the plan in between (the lane's lock).
int lsn::run_count(LsnFusion *p, const void *d_depth, const void *d_colors, int *d_offsets, int *offsets_mirror, hipEvent_t counted, hipStream_t s)
{
    if (!p || !d_depth || !d_colors || !d_offsets || p->n_ticks != 1) {
        lsn::set_error("run_count: bad arguments");
        return -1;
    }
    if (!p->params_set) {
        lsn::set_error("run_count: lsnFusionSetParams has not been called");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    if (ensure_thresholds(p, s)) return -1;
    FuseArgs a;
    fill_args(p, a, d_depth, d_colors, nullptr, d_offsets);
    const bool vec = p->vec_ok && ((uintptr_t)d_depth & 15) == 0 && ((uintptr_t)d_colors & 7) == 0 && (p->tick_depth_elems % 8) == 0;
    launch_count(p, vec, s, a);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(kScanThreads), 0, s, a.tile_counts, a.tiles_per_tick, a.frames, a.n_frames, a.offsets, offsets_mirror);
    LSN_HIP(hipGetLastError());
    if (counted) LSN_HIP(hipEventRecord(counted, s));
    return 0;
}

int lsn::run_write(LsnFusion *p, const void *d_depth, const void *d_colors, void *vertices, int *d_offsets, bool with_pixmap, bool host_out, hipStream_t s)
{
    if (!p || !d_depth || !d_colors || !vertices || !d_offsets || p->n_ticks != 1 || ((uintptr_t)vertices & 15) != 0) {
        lsn::set_error("run_write: bad arguments");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    const bool vec = p->vec_ok && ((uintptr_t)d_depth & 15) == 0 && ((uintptr_t)d_colors & 7) == 0 && (p->tick_depth_elems % 8) == 0;
    if (with_pixmap) {
        if (vec ? (p->pm_first.reserve(sizeof(int) * ((size_t)p->cap / 8 + 2)) || p->pm_mask.reserve((size_t)p->cap / 8 + 2))
                : p->pixmap.reserve(sizeof(int) * (size_t)p->cap))
            return -1;
        p->pixmap_compact = vec;
    }
    struct PixmapScope {
        LsnFusion *p;
        PixmapScope(LsnFusion *q, bool on) : p(q) { p->want_pixmap = on; }
        ~PixmapScope() { p->want_pixmap = false; }
    } scope(p, with_pixmap);
    FuseArgs a;
    fill_args(p, a, d_depth, d_colors, vertices, d_offsets);
    const int grid = p->tiles_per_tick;
    if (host_out) {
        if (vec) hipLaunchKernelGGL((fuse_kernel<1, true, true, true>), dim3(grid), dim3(kThreads), 0, s, a);
        else     hipLaunchKernelGGL((fuse_kernel<1, false, true, true>), dim3(grid), dim3(kThreads), 0, s, a);
    } else {
        launch<1>(vec, grid, s, a, p->lazy_rgb);
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

// Streamed calls: this batch is written while the NEXT batch (already resident) is counted by the same kernel.
static int lsnFusionRunStreamed_impl(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets,
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
        hipLaunchKernelGGL(scan_kernel, dim3((unsigned)p->n_ticks), dim3(kScanThreads), 0, s, cur, a.tiles_per_tick, a.frames, a.n_frames, off_cur, nullptr);
    }
    LSN_HIP(hipMemcpyAsync(d_offsets, off_cur, sizeof(int) * off_elems, hipMemcpyDeviceToDevice, s));
    a.offsets = d_offsets;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed_launch(p) && next_event_pair(p, e0, e1)) return -1;
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
        hipLaunchKernelGGL(scan_kernel, dim3((unsigned)p->n_ticks), dim3(kScanThreads), 0, s, nxt, a.tiles_per_tick, a.frames, a.n_frames, off_nxt, nullptr);
        p->counted_for = d_next_depth;
        p->counted_gen = p->params_gen;
        p->stream_half ^= 1;
    } else {
        p->counted_for = nullptr;
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

extern "C" int lsnFusionRunStreamed(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets,
                                    const void *d_next_depth, void *stream)
{
    return lsn::guarded<int>("lsnFusionRunStreamed", static_cast<int>(-1), [&]() { return lsnFusionRunStreamed_impl(p, d_depth, d_colors, d_vertices, d_offsets, d_next_depth, stream); });
}

// Diagnostics / tests: builds the per-pixel depth thresholds now (if the plan uses them) and copies them out.
static int lsnFusionThresholds_impl(LsnFusion *p, unsigned int *out_host, float *build_ms, void *stream)
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

extern "C" int lsnFusionThresholds(LsnFusion *p, unsigned int *out_host, float *build_ms, void *stream)
{
    return lsn::guarded<int>("lsnFusionThresholds", static_cast<int>(-1), [&]() { return lsnFusionThresholds_impl(p, out_host, build_ms, stream); });
}

// Reads back the look-back error flag (diagnostics for tests); synchronises the stream.
extern "C" int lsnFusionCheck(LsnFusion *p, void *stream) { return lsnFusionLookbackFailed(p, stream); }

static int lsnFusionLookbackFailed_impl(LsnFusion *p, void *stream)
{
    if (!p) return -1;
    int flag = 0;
    LSN_HIP(hipSetDevice(p->device));
    LSN_HIP(hipStreamSynchronize(lsn::as_stream(stream)));
    LSN_HIP(hipMemcpy(&flag, p->misc.p, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) LSN_HIP(hipMemset(p->misc.p, 0, sizeof(int)));
    return flag;
}

extern "C" int lsnFusionLookbackFailed(LsnFusion *p, void *stream)
{
    return lsn::guarded<int>("lsnFusionLookbackFailed", static_cast<int>(-1), [&]() { return lsnFusionLookbackFailed_impl(p, stream); });
}

