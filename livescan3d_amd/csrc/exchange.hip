// exchange.hip -- the multi-GPU exchange step: survivor exchange (pack_kernel / recon_kernel) and the vertex exchange's
// packing pass (merge_shards_kernel).  Shares the plan and the per-pixel arithmetic of fusion.hip (fusion_shared.hpp).
#include "fusion_shared.hpp"

namespace {

struct MergeArgs {
    const uint4 *shards;     // [n_shards][n_ticks][shard_cap]
    const int *shard_off;    // [n_shards][n_ticks][maps_per_shard + 1]  (the gathered lsnFusionRun offsets)
    uint4 *merged;           // [n_ticks][merged_cap]
    int *merged_off;         // [n_ticks][n_shards * maps_per_shard + 1]
    long long shard_cap, merged_cap;
    int n_shards, n_ticks, maps_per_shard;
    int tick_major;          // shards laid out [n_ticks][n_shards][shard_cap] instead (what per-tick all-gathers produce)
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
    const uint4 *src = a.shards + (a.tick_major ? (long long)tick * a.n_shards + shard : (long long)shard * a.n_ticks + tick) * a.shard_cap;
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
    const int *tick_base;     // null: tick k's survivors start at k * cap (layout above).  Else [n_ticks]: they start at tick_base[k] --
                              // all ticks back to back, ONE contiguous run per shard: what the RCCL all-gather sends as it is
};

// Aligned copy of n bytes that sit in LDS at lds[lead ..), lead = (address of dst) mod 16, to dst: the 16-byte chunks that lie
// wholly inside the range go out as one store each, the (< 16-byte) pieces at either end element by element (ELEM bytes each) --
// the neighbouring tiles own the rest of those chunks.
template <int ELEM, typename T>
__device__ __forceinline__ void store_run(T *dst, const T *lds, int lead, int n)
{
    static_assert(sizeof(T) == ELEM, "element size");
    const int end = lead + n;                        // in bytes, relative to the aligned start of the first chunk
    const int c0 = lead ? 1 : 0, c1 = end >> 4;       // chunks [c0, c1) are whole
    uint4 *g16 = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(dst) - lead);
    const uint4 *l16 = reinterpret_cast<const uint4 *>(lds);
    for (int j = c0 + (int)threadIdx.x; j < c1; j += kThreads) g16[j] = l16[j];
    const int head = lead ? min(n, 16 - lead) : 0;   // bytes before the first whole chunk
    const int tail0 = max(head, 16 * c1 - lead);     // first byte after the last whole chunk
    const int t = (int)threadIdx.x * ELEM;
    if (t < head) dst[threadIdx.x] = lds[lead / ELEM + threadIdx.x];
    if (tail0 + t < n) dst[tail0 / ELEM + threadIdx.x] = lds[(lead + tail0) / ELEM + threadIdx.x];
}

// The survivors of a tile are staged in LDS in rank order and leave as 16-byte stores (the first version had every lane store
// its own short run: 32 one- and two-byte stores per lane, 0.32 ms per 64 ticks x 8 sensors against 0.14 at the copy rate).
__global__ __launch_bounds__(kThreads) void pack_kernel(const FuseArgs a, const PackArgs pk)
{
    __shared__ alignas(16) unsigned short s_d[kTile + 8];
    __shared__ alignas(16) unsigned char s_c[3 * kTile + 16];
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
    const long long r0 = (pk.tick_base ? (long long)pk.tick_base[tick] : tick * a.tick_vert_stride) + base;   // the tile's first stream entry
    __syncthreads();
    int wave_off = 0, tile_tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = s_wave_tot[i];
        if (i < wave) wave_off += v;
        tile_tot += v;
    }
    const int p0 = t.px0 + threadIdx.x * kPxPerLane;
    unsigned short *gd = pk.depth_c + r0;
    unsigned char *gc = pk.rgb_c + 3 * r0;
    const int lead_d = (int)(reinterpret_cast<uintptr_t>(gd) & 15), lead_c = (int)(reinterpret_cast<uintptr_t>(gc) & 15);   // bytes
    if (p0 < t.npix) {
        unsigned int m8 = 0;
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) m8 |= (keep[k] ? 1u : 0u) << k;
        pk.mask[(tick * a.tick_depth_stride + t.pix_base + p0) >> 3] = (unsigned char)m8;
        int e = wave_off + below;
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) {
            if (keep[k]) {
                const int b = 3 * k;
                const unsigned int lo = in.cw[b >> 2], hi = in.cw[(b >> 2) + 1 < 6 ? (b >> 2) + 1 : 5];
                const unsigned int rgb = __funnelshift_r(lo, hi, (b & 3) * 8);
                s_d[lead_d / 2 + e] = (unsigned short)((k & 1) ? in.dw[k >> 1] >> 16 : in.dw[k >> 1] & 0xFFFFu);
                unsigned char *c = s_c + lead_c + 3 * e;
                c[0] = (unsigned char)rgb;
                c[1] = (unsigned char)(rgb >> 8);
                c[2] = (unsigned char)(rgb >> 16);
                e++;
            }
        }
    }
    __syncthreads();
    store_run<2>(gd, s_d, lead_d, 2 * tile_tot);
    store_run<1>(gc, s_c, lead_c, 3 * tile_tot);
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
    const int *tick_base;            // null: streams laid out [n_shards][n_ticks][slab].  Else [n_shards][n_ticks]: shard q's streams are ONE
                                     // run of `slab` entries, tick k's survivors start at tick_base[q][k] inside it
    int tick0;                       // first tick of this launch (a launch may cover a chunk of the ticks: the streams passed then
                                     // start at tick0's survivors, `slab` entries per shard for the chunk)
};

// Exclusive prefix over the ticks of every shard's per-tick survivor count (the last entry of its offset rows): where each
// tick starts in a shard's back-to-back streams.  One workgroup per shard.
__global__ __launch_bounds__(kThreads) void tick_base_kernel(const int *shard_off /* [n_shards][n_ticks][mps + 1] */, int n_ticks, int mps,
                                                             int *tick_base /* [n_shards][n_ticks] */)
{
    __shared__ int s_wave[4];
    __shared__ int s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int *off = shard_off + (long long)blockIdx.x * n_ticks * (mps + 1);
    int *out = tick_base + (long long)blockIdx.x * n_ticks;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_ticks; c0 += kThreads) {
        const int i = c0 + threadIdx.x;
        const int v = i < n_ticks ? off[(long long)i * (mps + 1) + mps] : 0;
        const int incl = wave_inclusive_scan(v, lane);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int pre = s_carry;
        for (int w = 0; w < wave; w++) pre += s_wave[w];
        if (i < n_ticks) out[i] = pre + incl - v;
        __syncthreads();
        if (threadIdx.x == kThreads - 1) s_carry = pre + incl;
        __syncthreads();
    }
}

// `a` describes the WHOLE rig (all sensors, their parameters, a.out = the merged cloud).
// Two phases per tile.  Pixel-major: the survivor mask gives every lane the ranks of its 8 pixels, and the kept pixels leave their
// (column, row) in LDS at their rank.  Survivor-major: thread i takes survivors i, i + 256, ... -- consecutive threads read
// consecutive entries of the shard's compact depth / colour streams and write consecutive 16-byte vertices, so neither the
// gathers nor the stores need staging (the first version kept the pixel-major layout of the write kernel throughout: every lane
// fetched its own short run of the streams, 32 scattered 1- and 2-byte loads per lane, and staged the vertices through LDS:
// 0.37 ms per 64 ticks x 8 sensors against the 0.2 ms its 1.28 GB take at the copy rate).
struct ReconGeom {   // wave-uniform
    Tile t;
    int tick, shard, local_tile;
    long long st, mask_pix0;
};

__device__ __forceinline__ ReconGeom recon_geom(const FuseArgs &a, const ReconArgs &r, int b)
{
    ReconGeom g;
    g.tick = b / a.tiles_per_tick + r.tick0;
    const int tile = b - (g.tick - r.tick0) * a.tiles_per_tick;
    g.t = locate(a, g.tick, tile);
    g.shard = g.t.f / r.maps_per_shard;
    const FrameDesc f0 = a.frames[g.shard * r.maps_per_shard];   // first sensor of the owning shard
    g.st = (long long)g.shard * a.n_ticks + g.tick;              // (shard, tick) slot in the gathered arrays
    g.local_tile = tile - f0.tile_start;
    g.mask_pix0 = g.st * r.cap_loc + (g.t.pix_base - f0.depth_off) + g.t.px0;
    return g;
}

// Measured and dropped: a persistent form (a workgroup walks tiles b, b + grid, ... and fetches the next tile's mask byte and
// prefixes ahead) -- 0.50-0.66 ms: loads and stores share one in-order counter on this GPU, so the next tile's gathers wait
// for the previous tile's 16-byte stores to be acknowledged.
__global__ __launch_bounds__(kThreads, 8) void recon_kernel(const FuseArgs a, const ReconArgs r)
{
    __shared__ unsigned int s_xy[kTile];   // (column | row << 16) of the tile's survivors, by rank
    __shared__ int s_wave_tot[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const ReconGeom g = recon_geom(a, r, blockIdx.x);
    const Tile &t = g.t;
    const int p0 = t.px0 + threadIdx.x * kPxPerLane;
    const unsigned int m8 = p0 < t.npix ? r.mask[(g.mask_pix0 + threadIdx.x * kPxPerLane) >> 3] : 0u;
    const int tile_base = r.tile_prefix[g.st * r.tiles_loc + g.local_tile];
    int shard_base = 0;
    for (int q = 0; q < g.shard; q++) shard_base += r.shard_off[((long long)q * a.n_ticks + g.tick) * (r.maps_per_shard + 1) + r.maps_per_shard];
    const SensorParams P = a.params[t.f];
    bool keep[kPxPerLane];
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) keep[k] = (m8 >> k) & 1u;
    int below, wave_total;
    rank_from_masks(keep, below, wave_total);
    if (lane == 0) s_wave_tot[wave] = wave_total;
    int x, y;
    lane_origin(t, x, y);
    __syncthreads();
    int wave_off = 0, tile_tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = s_wave_tot[i];
        if (i < wave) wave_off += v;
        tile_tot += v;
    }
    {
        int e = wave_off + below;   // w % 8 == 0: the lane's 8 pixels share row y, columns x .. x + 7
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++)
            if (keep[k]) s_xy[e++] = (unsigned int)(x + k) | ((unsigned int)y << 16);
    }
    if (t.frame_start && threadIdx.x == 0) {
        int *mo = r.merged_off + (long long)g.tick * (a.n_frames + 1);
        mo[t.f] = shard_base + tile_base;             // the frame's first tile: its prefix is the sensor's offset in the shard
        if (t.f == a.n_frames - 1) {
            int total = 0;
            for (int q = 0; q < r.n_shards; q++)
                total += r.shard_off[((long long)q * a.n_ticks + g.tick) * (r.maps_per_shard + 1) + r.maps_per_shard];
            mo[a.n_frames] = total;
        }
    }
    __syncthreads();
    // the tile's survivors are consecutive entries of the shard's compact streams
    const long long run = r.tick_base ? (long long)g.shard * r.slab + (r.tick_base[g.st] - r.tick_base[(long long)g.shard * a.n_ticks + r.tick0])
                                      : g.st * r.slab;
    const unsigned short *dc = r.depth_c + run + tile_base;
    const unsigned char *cc = r.rgb_c + 3 * (run + tile_base);
    uint4 *dst = a.out + g.tick * a.tick_vert_stride + shard_base + tile_base;
    // all of a thread's (<= 8) survivors are fetched before the first one is evaluated: one memory round trip
    unsigned int d[kPxPerLane], c[kPxPerLane];
    float xf[kPxPerLane], yf[kPxPerLane];
#pragma unroll
    for (int i = 0; i < kPxPerLane; i++) {
        const int e = threadIdx.x + i * kThreads;
        d[i] = 0;
        c[i] = 0;
        unsigned int xy = 0;
        if (e < tile_tot) {
            xy = s_xy[e];
            d[i] = dc[e];
            c[i] = cc[3 * e] | (cc[3 * e + 1] << 8) | (cc[3 * e + 2] << 16);
        }
        xf[i] = t.xt[xy & 0xFFFFu];
        yf[i] = t.yt[xy >> 16];
    }
#pragma unroll
    for (int i = 0; i < kPxPerLane; i += 2) {
        f2 ox, oy, oz;
        unproject2(f2{(float)d[i], (float)d[i + 1]}, f2{xf[i], xf[i + 1]}, f2{yf[i], yf[i + 1]}, P, ox, oy, oz);
        const int e0 = threadIdx.x + i * kThreads, e1 = e0 + kThreads;
        if (e0 < tile_tot) store_vertex(dst + e0, make_uint4(c[i] | 0xFF000000u, __float_as_uint(ox.x), __float_as_uint(oy.x), __float_as_uint(oz.x)));
        if (e1 < tile_tot) store_vertex(dst + e1, make_uint4(c[i + 1] | 0xFF000000u, __float_as_uint(ox.y), __float_as_uint(oy.y), __float_as_uint(oz.y)));
    }
}

}  // namespace

extern "C" int lsnFusionTilesPerTick(const LsnFusion *p) { return p ? p->tiles_per_tick : 0; }

static int lsnFusionPackSurvivors_impl(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_mask, void *d_depth_c, void *d_rgb_c,
                                      int *d_tile_prefix, int *d_offsets, void *stream)
{
    lsn::clear_error();
    return lsn::pack_survivors(p, d_depth, d_colors, d_mask, d_depth_c, d_rgb_c, d_tile_prefix, d_offsets, nullptr, stream);
}

extern "C" int lsnFusionPackSurvivors(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_mask, void *d_depth_c, void *d_rgb_c,
                                      int *d_tile_prefix, int *d_offsets, void *stream)
{
    return lsn::guarded<int>("lsnFusionPackSurvivors", static_cast<int>(-1), [&]() { return lsnFusionPackSurvivors_impl(p, d_depth, d_colors, d_mask, d_depth_c, d_rgb_c, d_tile_prefix, d_offsets, stream); });
}

// Survivor exchange, sender side: count + scan as in lsnFusionRun, then the compact streams instead of vertices.
// d_tick_base (nullable, [n_ticks]): filled with every tick's start in the back-to-back layout, which the streams then use.
int lsn::pack_survivors(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_mask, void *d_depth_c, void *d_rgb_c,
                        int *d_tile_prefix, int *d_offsets, int *d_tick_base, void *stream)
{
    if (!p || !d_depth || !d_colors || !d_mask || !d_depth_c || !d_rgb_c || !d_offsets || (!d_tile_prefix && !d_tick_base)) {
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
    hipLaunchKernelGGL(scan_kernel, dim3((unsigned)p->n_ticks), dim3(kScanThreads), 0, s, a.tile_counts, a.tiles_per_tick, a.frames, a.n_frames, a.offsets, nullptr);
    PackArgs pk;
    pk.mask = static_cast<unsigned char *>(d_mask);
    pk.depth_c = static_cast<unsigned short *>(d_depth_c);
    pk.rgb_c = static_cast<unsigned char *>(d_rgb_c);
    pk.tick_base = d_tick_base;
    if (d_tick_base) hipLaunchKernelGGL(tick_base_kernel, dim3(1), dim3(kThreads), 0, s, (const int *)d_offsets, p->n_ticks, p->n_maps, d_tick_base);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)(p->tiles_per_tick * p->n_ticks)), dim3(kThreads), 0, s, a, pk);
    LSN_HIP(hipGetLastError());
    // (LsnShard sends the prefixes straight from the plan's scratch and passes no copy target)
    if (d_tile_prefix)
        LSN_HIP(hipMemcpyAsync(d_tile_prefix, p->tile_counts.p, sizeof(int) * (size_t)p->tiles_per_tick * p->n_ticks, hipMemcpyDeviceToDevice, s));
    return 0;
}

// Survivor exchange, receiver side: `all` is a plan over the WHOLE rig (every sensor, lsnFusionSetParams called with all
// parameters, same n_ticks); the gathered arrays hold n_shards equally shaped shards of maps_per_shard sensors each.
static int lsnFusionReconstruct_impl(LsnFusion *all, int n_shards, int maps_per_shard, const void *d_masks, const void *d_depth_c,
                                    const void *d_rgb_c, long long slab, const int *d_tile_prefix, const int *d_shard_offsets,
                                    void *d_merged, int *d_merged_offsets, void *stream)
{
    lsn::clear_error();
    return lsn::reconstruct(all, n_shards, maps_per_shard, d_masks, d_depth_c, d_rgb_c, slab, d_tile_prefix, d_shard_offsets, d_merged,
                            d_merged_offsets, nullptr, stream);
}

extern "C" int lsnFusionReconstruct(LsnFusion *all, int n_shards, int maps_per_shard, const void *d_masks, const void *d_depth_c,
                                    const void *d_rgb_c, long long slab, const int *d_tile_prefix, const int *d_shard_offsets,
                                    void *d_merged, int *d_merged_offsets, void *stream)
{
    return lsn::guarded<int>("lsnFusionReconstruct", static_cast<int>(-1), [&]() { return lsnFusionReconstruct_impl(all, n_shards, maps_per_shard, d_masks, d_depth_c, d_rgb_c, slab, d_tile_prefix, d_shard_offsets, d_merged, d_merged_offsets, stream); });
}

static int lsnFusionPackSurvivorsRun_impl(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_mask, void *d_depth_c, void *d_rgb_c,
                                         int *d_tile_prefix, int *d_offsets, int *d_tick_base, void *stream)
{
    lsn::clear_error();
    if (!d_tick_base || !d_tile_prefix) {
        lsn::set_error("lsnFusionPackSurvivorsRun: null argument");
        return -1;
    }
    return lsn::pack_survivors(p, d_depth, d_colors, d_mask, d_depth_c, d_rgb_c, d_tile_prefix, d_offsets, d_tick_base, stream);
}

extern "C" int lsnFusionPackSurvivorsRun(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_mask, void *d_depth_c, void *d_rgb_c,
                                         int *d_tile_prefix, int *d_offsets, int *d_tick_base, void *stream)
{
    return lsn::guarded<int>("lsnFusionPackSurvivorsRun", static_cast<int>(-1), [&]() { return lsnFusionPackSurvivorsRun_impl(p, d_depth, d_colors, d_mask, d_depth_c, d_rgb_c, d_tile_prefix, d_offsets, d_tick_base, stream); });
}

static int lsnFusionReconstructRun_impl(LsnFusion *all, int n_shards, int maps_per_shard, const void *d_masks, const void *d_depth_c,
                                       const void *d_rgb_c, long long run_len, const int *d_tile_prefix, const int *d_shard_offsets,
                                       void *d_merged, int *d_merged_offsets, int *d_tick_base_scratch, void *stream)
{
    lsn::clear_error();
    if (!d_tick_base_scratch) {
        lsn::set_error("lsnFusionReconstructRun: null argument");
        return -1;
    }
    return lsn::reconstruct(all, n_shards, maps_per_shard, d_masks, d_depth_c, d_rgb_c, run_len, d_tile_prefix, d_shard_offsets, d_merged,
                            d_merged_offsets, d_tick_base_scratch, stream);
}

extern "C" int lsnFusionReconstructRun(LsnFusion *all, int n_shards, int maps_per_shard, const void *d_masks, const void *d_depth_c,
                                       const void *d_rgb_c, long long run_len, const int *d_tile_prefix, const int *d_shard_offsets,
                                       void *d_merged, int *d_merged_offsets, int *d_tick_base_scratch, void *stream)
{
    return lsn::guarded<int>("lsnFusionReconstructRun", static_cast<int>(-1), [&]() { return lsnFusionReconstructRun_impl(all, n_shards, maps_per_shard, d_masks, d_depth_c, d_rgb_c, run_len, d_tile_prefix, d_shard_offsets, d_merged, d_merged_offsets, d_tick_base_scratch, stream); });
}

// d_tick_base (nullable, scratch [n_shards][n_ticks]): the gathered streams are one back-to-back run of `slab` entries per shard
// (lsn::pack_survivors with a tick base); the per-shard tick starts are recomputed here from the gathered offset tables.
int lsn::reconstruct(LsnFusion *all, int n_shards, int maps_per_shard, const void *d_masks, const void *d_depth_c, const void *d_rgb_c,
                     long long slab, const int *d_tile_prefix, const int *d_shard_offsets, void *d_merged, int *d_merged_offsets,
                     int *d_tick_base, void *stream, int tick0, int n_chunk_ticks, bool fill_tick_base)
{
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
    r.tick_base = d_tick_base;
    r.tick0 = tick0;
    if (n_chunk_ticks <= 0) n_chunk_ticks = all->n_ticks - tick0;
    if (tick0 < 0 || tick0 + n_chunk_ticks > all->n_ticks || (tick0 != 0 && !d_tick_base)) {
        lsn::set_error("lsnFusionReconstruct: bad tick range");
        return -1;
    }
    if (d_tick_base && fill_tick_base)
        hipLaunchKernelGGL(tick_base_kernel, dim3((unsigned)n_shards), dim3(kThreads), 0, lsn::as_stream(stream), d_shard_offsets, all->n_ticks,
                           maps_per_shard, d_tick_base);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed_launch(all)) {
        if (next_event_pair(all, e0, e1)) return -1;
        all->timed_kernel = "recon_kernel";
        LSN_HIP(hipEventRecord(e0, lsn::as_stream(stream)));
    }
    hipLaunchKernelGGL(recon_kernel, dim3((unsigned)(all->tiles_per_tick * n_chunk_ticks)), dim3(kThreads), 0, lsn::as_stream(stream), a, r);
    if (e1) LSN_HIP(hipEventRecord(e1, lsn::as_stream(stream)));
    LSN_HIP(hipGetLastError());
    return 0;
}

static int lsnMergeShards_impl(int device, int n_shards, int n_ticks, int maps_per_shard, const void *d_shards, long long shard_cap,
                              const int *d_shard_offsets, void *d_merged, long long merged_cap, int *d_merged_offsets, void *stream)
{
    lsn::clear_error();
    return lsn::merge_shards(device, n_shards, n_ticks, maps_per_shard, d_shards, shard_cap, d_shard_offsets, d_merged, merged_cap, d_merged_offsets,
                             false, stream);
}

extern "C" int lsnMergeShards(int device, int n_shards, int n_ticks, int maps_per_shard, const void *d_shards, long long shard_cap,
                              const int *d_shard_offsets, void *d_merged, long long merged_cap, int *d_merged_offsets, void *stream)
{
    return lsn::guarded<int>("lsnMergeShards", static_cast<int>(-1), [&]() { return lsnMergeShards_impl(device, n_shards, n_ticks, maps_per_shard, d_shards, shard_cap, d_shard_offsets, d_merged, merged_cap, d_merged_offsets, stream); });
}

int lsn::merge_shards(int device, int n_shards, int n_ticks, int maps_per_shard, const void *d_shards, long long shard_cap,
                      const int *d_shard_offsets, void *d_merged, long long merged_cap, int *d_merged_offsets, bool tick_major, void *stream)
{
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
    a.tick_major = tick_major ? 1 : 0;
    long long chunks = (shard_cap + kThreads * 8 - 1) / (kThreads * 8);
    if (chunks > 256) chunks = 256;
    hipLaunchKernelGGL(merge_shards_kernel, dim3((unsigned)chunks, (unsigned)(n_shards * n_ticks)), dim3(kThreads), 0, lsn::as_stream(stream), a);
    LSN_HIP(hipGetLastError());
    return 0;
}



// -------------------------------------------------------------------------------------------------------------------------
// LsnShard: the whole multi-GPU step behind the C-ABI -- one process per GPU, this rank's block of sensors in, the merged
// cloud of ALL sensors out, RCCL over xGMI in between (include/NativeUtils.h part 2b).
//
// The reference fans createVertices out over one std::thread per sensor and concatenates the per-sensor clouds in sensor
// order (src/NativeUtils/depthprocessing.cpp:708-733, formMesh :1594-1608).  Across GPUs: rank r owns the contiguous block
// [r * S / G, (r + 1) * S / G) of every tick's sensors (rank order = formMesh's sensor order) and ONE exchange step forms the
// merged cloud on every rank.  What crosses the links is what the vertices are made of (5 bytes per survivor + 1 bit per
// pixel instead of 16 bytes per vertex); every rank rebuilds all vertices with the arithmetic of the write kernel.
//
// Per step, on the caller's stream:  pack (count, scan, tick bases, compact streams -- all ticks of the rank back to back,
// so the send buffer is one contiguous run and nothing is staged)  ->  ncclAllGather of the offset tables  ->  the host reads
// them through pinned memory (one event wait: the element count of a collective is a host argument, and all ranks must pass
// the same one, so the largest rank total has to be known on the host)  ->  ONE ncclGroup with the all-gathers of the tile
// prefixes, the survivor masks and the two streams  ->  reconstruct.  $LSN_SHARD_PADDED=1 skips the host read and always
// gathers full-capacity streams (no synchronisation at all, about twice the bytes on the links).
// RCCL is loaded with dlopen on first use: a single-GPU host of this library never maps it.
// -------------------------------------------------------------------------------------------------------------------------
#include <dlfcn.h>

// RCCL is only ever dlopen()ed, so the library builds without the rccl development headers: the handful of types and constants the
// seven entry points need are declared here (the nccl.h ABI: opaque communicator, 128-byte id, ncclSuccess = 0, ncclUint8 = 1,
// ncclInt32 = 2).
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2 } ncclDataType_t;
}

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;      // optional: what the communicator itself reports (lsnShardRanksSeen)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    std::string path;   // the file the entry points came from
};

// ncclGroupStart ... ncclGroupEnd with the end guaranteed: a collective that fails inside a group must not leave the thread's group
// open (every later collective would be queued and never launched -- the next step would hang instead of reporting an error).
struct NcclGroup {
    Rccl *r;
    bool open = false;
    explicit NcclGroup(Rccl *r_) : r(r_) {}
    ncclResult_t start() { const ncclResult_t rc = r->GroupStart(); open = rc == ncclSuccess; return rc; }
    ncclResult_t end() { open = false; return r->GroupEnd(); }
    ~NcclGroup() { if (open) (void)r->GroupEnd(); }
};

Rccl *rccl()
{
    static std::mutex mu;
    static Rccl *r = nullptr;
    std::lock_guard<std::mutex> g(mu);
    if (r) return r;
    void *lib = nullptr;
    // $LSN_RCCL_LIBRARY: an RCCL build outside the loader path -- or the shared-memory test double of tests/fake_rccl, which
    // lets several ranks share one GPU (RCCL itself refuses that)
    const char *custom = getenv("LSN_RCCL_LIBRARY");
    if (custom && *custom) {
        lib = dlopen(custom, RTLD_NOW | RTLD_LOCAL);
    } else {
        // An RCCL that is already mapped in this process comes first (PyTorch-ROCm maps its own bundled librccl.so when
        // torch.distributed initialises the "nccl" backend): one process never holds two RCCL instances.  Only a host without one
        // (LiveScanServer, C/C++ callers) loads the system's.
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (lib) break;
        }
        if (!lib) {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (lib) break;
            }
        }
    }
    if (!lib) {
        lsn::set_error("lsnShard: cannot load %s (%s)", custom && *custom ? custom : "librccl.so.1", dlerror());
        return nullptr;
    }
    Rccl *t = new Rccl();
    t->lib = lib;
    t->GetUniqueId = (decltype(t->GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    t->CommInitRank = (decltype(t->CommInitRank))dlsym(lib, "ncclCommInitRank");
    t->CommDestroy = (decltype(t->CommDestroy))dlsym(lib, "ncclCommDestroy");
    t->AllGather = (decltype(t->AllGather))dlsym(lib, "ncclAllGather");
    t->GroupStart = (decltype(t->GroupStart))dlsym(lib, "ncclGroupStart");
    t->GroupEnd = (decltype(t->GroupEnd))dlsym(lib, "ncclGroupEnd");
    t->GetErrorString = (decltype(t->GetErrorString))dlsym(lib, "ncclGetErrorString");
    t->CommCount = (decltype(t->CommCount))dlsym(lib, "ncclCommCount");
    t->CommUserRank = (decltype(t->CommUserRank))dlsym(lib, "ncclCommUserRank");
    if (!t->GetUniqueId || !t->CommInitRank || !t->CommDestroy || !t->AllGather || !t->GroupStart || !t->GroupEnd || !t->GetErrorString) {
        lsn::set_error("lsnShard: librccl.so.1 lacks an expected entry point");
        delete t;
        return nullptr;
    }
    Dl_info info;
    if (dladdr((void *)t->AllGather, &info) && info.dli_fname) t->path = info.dli_fname;
    r = t;
    return r;
}

#define LSN_NCCL(expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess) {                                                                           \
            lsn::set_error("%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString(_r), __FILE__, __LINE__); \
            return -1;                                                                                     \
        }                                                                                                  \
    } while (0)

}  // namespace

struct LsnShard {
    int device = 0, rank = 0, world = 1;
    int n_ticks = 0, n_maps = 0, mpr = 0;     // all sensors / sensors per rank
    LsnFusion *local = nullptr, *whole = nullptr;
    ncclComm_t comm = nullptr;
    bool padded = false;                       // $LSN_SHARD_PADDED=1
    // Rigs the survivor exchange cannot serve (sensors of different sizes, widths that are not multiples of 8) exchange the
    // 16-byte vertices instead: lsnFusionRun on the rank's block, one all-gather per tick (all ticks in one ncclGroup) of slabs
    // cut to the largest shard of the step, then the packing pass of lsnMergeShards.  Same merged cloud, ~3x the bytes.
    bool vertex_mode = false;
    lsn::DevBuf v_local, v_gathered;           // [n_ticks][cap_loc] vertices of this rank / [n_ticks][world][slab]
    long long cap_loc = 0;                     // vertices per tick of one rank's block
    int tiles_loc = 0;
    lsn::DevBuf mask, depth_c, rgb_c, offsets, tick_base;        // this rank's packed survivors
    lsn::DevBuf g_off, g_tp, g_mask, g_dc, g_cc, g_tick_base;    // gathered, [world][...]
    lsn::DevBuf merged, merged_off;
    int *h_goff = nullptr;                     // pinned copy of the gathered offset tables
    hipEvent_t ev_off = nullptr;
    // the streams travel in `chunks` groups of ticks on a second stream while the reconstruction of the previous group runs
    int chunks = 1;                            // $LSN_SHARD_CHUNKS (1 = one shot on the caller's stream, the default)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_pre = nullptr;
    std::vector<hipEvent_t> ev_chunk;
    long long last_slab = 0, last_bytes_per_rank = 0;
    bool failed = false;                       // a step returned an error: every later step refuses
    std::string failure;
    std::mutex mu;
};

static int lsnShardUniqueId_impl(unsigned char *id128)
{
    lsn::clear_error();
    if (!id128) return -1;
    Rccl *r = rccl();
    if (!r) return -1;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    LSN_NCCL(r->GetUniqueId(&id));
    memcpy(id128, &id, 128);
    return 0;
}

extern "C" int lsnShardUniqueId(unsigned char *id128)
{
    return lsn::guarded<int>("lsnShardUniqueId", static_cast<int>(-1), [&]() { return lsnShardUniqueId_impl(id128); });
}

static void lsnShardDestroy_impl(LsnShard *sh)
{
    if (!sh) return;
    (void)hipSetDevice(sh->device);
    if (sh->comm && rccl()) (void)rccl()->CommDestroy(sh->comm);
    if (sh->local) lsnFusionDestroy(sh->local);
    if (sh->whole) lsnFusionDestroy(sh->whole);
    if (sh->h_goff) (void)hipHostFree(sh->h_goff);
    if (sh->ev_off) (void)hipEventDestroy(sh->ev_off);
    if (sh->ev_pre) (void)hipEventDestroy(sh->ev_pre);
    for (hipEvent_t e : sh->ev_chunk) (void)hipEventDestroy(e);
    if (sh->comm_stream) (void)hipStreamDestroy(sh->comm_stream);
    delete sh;
}

extern "C" void lsnShardDestroy(LsnShard *sh)
{
    lsn::guarded_void("lsnShardDestroy", [&]() { lsnShardDestroy_impl(sh); });
}

static LsnShard * lsnShardPrepare_impl(int device, int rank, int world, int n_ticks, int n_maps, const int *widths, const int *heights)
{
    lsn::clear_error();
    if (world <= 0 || rank < 0 || rank >= world || n_ticks <= 0 || n_maps <= 0 || !widths || !heights || n_maps % world != 0) {
        lsn::set_error("lsnShardPrepare: bad arguments (rank %d of %d, %d sensors must split evenly)", rank, world, n_maps);
        return nullptr;
    }
    bool uniform = true;
    for (int i = 0; i < n_maps; i++) uniform = uniform && widths[i] == widths[0] && heights[i] == heights[0] && widths[i] % 8 == 0;
    const int per_rank = n_maps / world;
    // the vertex exchange needs equally shaped shards as well (an all-gather moves equal blocks): every rank's block must have
    // the same pixel capacity
    long long cap0 = 0;
    for (int q = 0; q < world; q++) {
        long long capq = 0;
        for (int i = 0; i < per_rank; i++) capq += (long long)widths[q * per_rank + i] * heights[q * per_rank + i];
        if (q == 0) cap0 = capq;
        if (capq != cap0) {
            lsn::set_error("lsnShardPrepare: the ranks' sensor blocks must hold the same number of pixels (%lld vs %lld)", cap0, capq);
            return nullptr;
        }
    }
    Rccl *r = rccl();
    if (!r) return nullptr;
    LSN_HIP_NULL(hipSetDevice(device));
    LsnShard *sh = new (std::nothrow) LsnShard();
    if (!sh) return nullptr;
    sh->device = device;
    sh->rank = rank;
    sh->world = world;
    sh->n_ticks = n_ticks;
    sh->n_maps = n_maps;
    sh->mpr = n_maps / world;
    sh->vertex_mode = !uniform || (getenv("LSN_SHARD_VERTICES") && atoi(getenv("LSN_SHARD_VERTICES")) != 0);
    if (const char *e = getenv("LSN_SHARD_PADDED")) sh->padded = atoi(e) != 0;
    // One shot on the caller's stream unless $LSN_SHARD_CHUNKS asks for the pipelined form (collectives on a second stream beside the
    // reconstruction): that form has only ever run with the shared-memory test double and with one real rank -- it stays opt-in until
    // a run on a multi-GPU node has verified it.
    sh->chunks = 1;
    if (const char *e = getenv("LSN_SHARD_CHUNKS")) sh->chunks = atoi(e);
    if (sh->chunks < 1) sh->chunks = 1;
    if (sh->chunks > 16) sh->chunks = 16;
    if (sh->chunks > n_ticks) sh->chunks = n_ticks;
    sh->local = lsnFusionCreate(device, n_ticks, sh->mpr, widths + rank * sh->mpr, heights + rank * sh->mpr);
    sh->whole = lsnFusionCreate(device, n_ticks, n_maps, widths, heights);
    bool bad = !sh->local || !sh->whole;
    if (!bad) {
        sh->cap_loc = sh->local->cap;
        sh->tiles_loc = sh->local->tiles_per_tick;
        const size_t T = (size_t)n_ticks, W = (size_t)world, cap = (size_t)sh->cap_loc;
        if (sh->vertex_mode) {
            bad |= sh->v_local.reserve(T * cap * 16) != 0;
            bad |= sh->v_gathered.reserve(W * T * cap * 16) != 0;
        }
        bad |= sh->mask.reserve(T * cap / 8) != 0;
        // a chunk's send starts at the rank's own tick start and is as long as the LONGEST rank's chunk: room to read past the end
        const size_t chunk_cap = ((T + sh->chunks - 1) / sh->chunks) * cap + 64;
        bad |= sh->depth_c.reserve((T * cap + chunk_cap) * 2 + 64) != 0;
        bad |= sh->rgb_c.reserve((T * cap + chunk_cap) * 3 + 64) != 0;
        bad |= sh->offsets.reserve(sizeof(int) * T * (sh->mpr + 1)) != 0;
        bad |= sh->tick_base.reserve(sizeof(int) * T) != 0;
        bad |= sh->g_off.reserve(sizeof(int) * W * T * (sh->mpr + 1)) != 0;
        bad |= sh->g_tp.reserve(sizeof(int) * W * T * sh->tiles_loc) != 0;
        bad |= sh->g_mask.reserve(W * T * cap / 8) != 0;
        bad |= sh->g_dc.reserve(W * (T * cap + 8 * 16 + 64) * 2) != 0;
        bad |= sh->g_cc.reserve(W * (T * cap + 8 * 16 + 64) * 3) != 0;
        bad |= sh->g_tick_base.reserve(sizeof(int) * W * T) != 0;
        bad |= sh->merged.reserve((size_t)sh->whole->cap * 16 * T) != 0;
        bad |= sh->merged_off.reserve(sizeof(int) * T * (n_maps + 1)) != 0;
        bad |= hipHostMalloc((void **)&sh->h_goff, sizeof(int) * W * T * (sh->mpr + 1), hipHostMallocDefault) != hipSuccess;
        bad |= hipEventCreateWithFlags(&sh->ev_off, hipEventDisableTiming) != hipSuccess;
        bad |= hipEventCreateWithFlags(&sh->ev_pre, hipEventDisableTiming) != hipSuccess;
        bad |= hipStreamCreateWithFlags(&sh->comm_stream, hipStreamNonBlocking) != hipSuccess;
        for (int c = 0; c < sh->chunks && !bad; c++) {
            hipEvent_t e = nullptr;
            bad |= hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess;
            if (e) sh->ev_chunk.push_back(e);
        }
    }
    if (bad) {
        if (!lsn::has_error()) lsn::set_error("lsnShardPrepare: allocation failed: %s", hipGetErrorString(hipGetLastError()));
        lsnShardDestroy(sh);
        return nullptr;
    }
    return sh;
}

extern "C" LsnShard * lsnShardPrepare(int device, int rank, int world, int n_ticks, int n_maps, const int *widths, const int *heights)
{
    return lsn::guarded<LsnShard *>("lsnShardPrepare", static_cast<LsnShard *>(nullptr), [&]() { return lsnShardPrepare_impl(device, rank, world, n_ticks, n_maps, widths, heights); });
}

static int lsnShardConnect_impl(LsnShard *sh, const unsigned char *id128)
{
    lsn::clear_error();
    if (!sh || !id128) {
        lsn::set_error("lsnShardConnect: null argument");
        return -1;
    }
    Rccl *r = rccl();
    if (!r) return -1;
    std::lock_guard<std::mutex> g(sh->mu);
    if (sh->comm) {
        lsn::set_error("lsnShardConnect: already connected");
        return -1;
    }
    LSN_HIP(hipSetDevice(sh->device));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    const ncclResult_t rc = r->CommInitRank(&sh->comm, sh->world, id, sh->rank);   // blocks until every rank of the world has called it
    if (rc != ncclSuccess) {
        lsn::set_error("lsnShardConnect: ncclCommInitRank failed: %s", r->GetErrorString(rc));
        sh->comm = nullptr;
        return -1;
    }
    return 0;
}

extern "C" int lsnShardConnect(LsnShard *sh, const unsigned char *id128)
{
    return lsn::guarded<int>("lsnShardConnect", static_cast<int>(-1), [&]() { return lsnShardConnect_impl(sh, id128); });
}

static LsnShard * lsnShardCreate_impl(int device, int rank, int world, const unsigned char *id128, int n_ticks, int n_maps, const int *widths,
                                    const int *heights)
{
    if (!id128) {
        lsn::clear_error();
        lsn::set_error("lsnShardCreate: null id");
        return nullptr;
    }
    LsnShard *sh = lsnShardPrepare(device, rank, world, n_ticks, n_maps, widths, heights);
    if (!sh) return nullptr;
    if (lsnShardConnect(sh, id128)) {
        const std::string why = lsn::error_buffer();
        lsnShardDestroy(sh);
        lsn::set_error("%s", why.c_str());
        return nullptr;
    }
    return sh;
}

extern "C" LsnShard * lsnShardCreate(int device, int rank, int world, const unsigned char *id128, int n_ticks, int n_maps, const int *widths,
                                    const int *heights)
{
    return lsn::guarded<LsnShard *>("lsnShardCreate", static_cast<LsnShard *>(nullptr), [&]() { return lsnShardCreate_impl(device, rank, world, id128, n_ticks, n_maps, widths, heights); });
}

static int lsnShardRcclPath_impl(char *buf, int len)
{
    Rccl *r = rccl();
    if (!r) return -1;
    if (buf && len > 0) snprintf(buf, (size_t)len, "%s", r->path.c_str());
    return (int)r->path.size();
}

extern "C" int lsnShardRcclPath(char *buf, int len)
{
    return lsn::guarded<int>("lsnShardRcclPath", static_cast<int>(-1), [&]() { return lsnShardRcclPath_impl(buf, len); });
}

// What the connected communicator itself reports: its rank count, or -1 (not connected / the RCCL in use lacks ncclCommCount / the
// communicator's own rank differs from the handle's).  bench.py puts it into the N > 1 line as n_ranks_seen.
static int lsnShardRanksSeen_impl(LsnShard *sh)
{
    lsn::clear_error();
    Rccl *r = rccl();
    if (!sh || !r) return -1;
    std::lock_guard<std::mutex> g(sh->mu);
    if (!sh->comm || !r->CommCount) {
        lsn::set_error("lsnShardRanksSeen: %s", !sh->comm ? "the handle is not connected" : "this RCCL has no ncclCommCount");
        return -1;
    }
    int n = -1, me = sh->rank;
    LSN_NCCL(r->CommCount(sh->comm, &n));
    if (r->CommUserRank) LSN_NCCL(r->CommUserRank(sh->comm, &me));
    if (me != sh->rank) {
        lsn::set_error("lsnShardRanksSeen: the communicator says rank %d, the handle %d", me, sh->rank);
        return -1;
    }
    return n;
}

extern "C" int lsnShardRanksSeen(LsnShard *sh)
{
    return lsn::guarded<int>("lsnShardRanksSeen", static_cast<int>(-1), [&]() { return lsnShardRanksSeen_impl(sh); });
}

extern "C" LsnFusion *lsnShardPlan(LsnShard *sh, int whole) { return sh ? (whole ? sh->whole : sh->local) : nullptr; }
extern "C" long long lsnShardMergedCapacity(const LsnShard *sh) { return sh && sh->whole ? sh->whole->cap : 0; }
extern "C" long long lsnShardLastBytesSent(const LsnShard *sh) { return sh ? sh->last_bytes_per_rank : 0; }

static int lsnShardSetParams_impl(LsnShard *sh, const float *intr_all, const float *wt_all, const float *bounds6, void *stream)
{
    lsn::clear_error();
    if (!sh || !intr_all || !wt_all || !bounds6) {
        lsn::set_error("lsnShardSetParams: null argument");
        return -1;
    }
    if (lsnFusionSetParams(sh->whole, intr_all, wt_all, bounds6, stream)) return -1;
    return lsnFusionSetParams(sh->local, intr_all + 7 * (size_t)sh->rank * sh->mpr, wt_all + 12 * (size_t)sh->rank * sh->mpr, bounds6, stream);
}

extern "C" int lsnShardSetParams(LsnShard *sh, const float *intr_all, const float *wt_all, const float *bounds6, void *stream)
{
    return lsn::guarded<int>("lsnShardSetParams", static_cast<int>(-1), [&]() { return lsnShardSetParams_impl(sh, intr_all, wt_all, bounds6, stream); });
}

static int shard_step(LsnShard *sh, const void *d_depth_local, const void *d_colors_local, void **d_merged, int **d_merged_offsets, void *stream);

static int lsnShardStep_impl(LsnShard *sh, const void *d_depth_local, const void *d_colors_local, void **d_merged, int **d_merged_offsets,
                            void *stream)
{
    lsn::clear_error();
    if (!sh || !d_depth_local || !d_colors_local) {
        lsn::set_error("lsnShardStep: null argument");
        return -1;
    }
    std::lock_guard<std::mutex> g(sh->mu);
    if (!sh->comm) {
        lsn::set_error("lsnShardStep: the handle is not connected (lsnShardConnect)");
        return -1;
    }
    if (sh->failed) {
        // a collective of an earlier step failed: the ranks' communicators are no longer in step, nothing further may be queued on them
        lsn::set_error("lsnShardStep: an earlier step failed (%s); destroy the handle", sh->failure.c_str());
        return -1;
    }
    const int rc = shard_step(sh, d_depth_local, d_colors_local, d_merged, d_merged_offsets, stream);
    if (rc) {
        sh->failed = true;
        sh->failure = lsn::error_buffer();
    }
    return rc;
}

extern "C" int lsnShardStep(LsnShard *sh, const void *d_depth_local, const void *d_colors_local, void **d_merged, int **d_merged_offsets,
                            void *stream)
{
    return lsn::guarded<int>("lsnShardStep", static_cast<int>(-1), [&]() { return lsnShardStep_impl(sh, d_depth_local, d_colors_local, d_merged, d_merged_offsets, stream); });
}

static int shard_step(LsnShard *sh, const void *d_depth_local, const void *d_colors_local, void **d_merged, int **d_merged_offsets, void *stream)
{
    Rccl *r = rccl();
    if (!r) return -1;
    LSN_HIP(hipSetDevice(sh->device));
    hipStream_t s = lsn::as_stream(stream);
    const size_t T = (size_t)sh->n_ticks, W = (size_t)sh->world, cap = (size_t)sh->cap_loc;
    const size_t off_ints = T * (sh->mpr + 1);
    if (sh->vertex_mode) {
        if (lsn::run_hooked(sh->local, d_depth_local, d_colors_local, sh->v_local.p, sh->offsets.as<int>(), s, nullptr)) return -1;
        LSN_NCCL(r->AllGather(sh->offsets.p, sh->g_off.p, off_ints, ncclInt32, sh->comm, s));
        long long slab = (long long)cap;
        if (!sh->padded) {
            LSN_HIP(hipMemcpyAsync(sh->h_goff, sh->g_off.p, sizeof(int) * W * off_ints, hipMemcpyDeviceToHost, s));
            LSN_HIP(hipEventRecord(sh->ev_off, s));
            LSN_HIP(hipEventSynchronize(sh->ev_off));
            long long most = 1;
            for (size_t q = 0; q < W * T; q++) {
                const long long n = sh->h_goff[q * (sh->mpr + 1) + sh->mpr];
                most = n > most ? n : most;
            }
            if (most > slab) {
                lsn::set_error("lsnShardStep: a shard reports %lld vertices, more than its %lld pixels", most, slab);
                return -1;
            }
            slab = most;
        }
        {
            NcclGroup grp(r);
            LSN_NCCL(grp.start());
            for (size_t k = 0; k < T; k++)
                LSN_NCCL(r->AllGather(sh->v_local.as<uint4>() + k * cap, sh->v_gathered.as<uint4>() + k * W * (size_t)slab, (size_t)slab * 16, ncclUint8,
                                      sh->comm, s));
            LSN_NCCL(grp.end());
        }
        sh->last_slab = slab;
        sh->last_bytes_per_rank = (long long)(sizeof(int) * off_ints + T * (size_t)slab * 16);
        if (lsn::merge_shards(sh->device, sh->world, sh->n_ticks, sh->mpr, sh->v_gathered.p, slab, sh->g_off.as<int>(), sh->merged.p, sh->whole->cap,
                              sh->merged_off.as<int>(), true, stream))
            return -1;
        if (d_merged) *d_merged = sh->merged.p;
        if (d_merged_offsets) *d_merged_offsets = sh->merged_off.as<int>();
        return 0;
    }
    if (lsn::pack_survivors(sh->local, d_depth_local, d_colors_local, sh->mask.p, sh->depth_c.p, sh->rgb_c.p, nullptr, sh->offsets.as<int>(),
                            sh->tick_base.as<int>(), stream))
        return -1;
    LSN_NCCL(r->AllGather(sh->offsets.p, sh->g_off.p, off_ints, ncclInt32, sh->comm, s));
    long long slab = (long long)(T * cap);   // entries of one rank's streams that travel
    if (!sh->padded) {
        LSN_HIP(hipMemcpyAsync(sh->h_goff, sh->g_off.p, sizeof(int) * W * off_ints, hipMemcpyDeviceToHost, s));
        LSN_HIP(hipEventRecord(sh->ev_off, s));
        LSN_HIP(hipEventSynchronize(sh->ev_off));
        long long most = 1;
        for (size_t q = 0; q < W; q++) {
            long long tot = 0;
            for (size_t k = 0; k < T; k++) tot += sh->h_goff[(q * T + k) * (sh->mpr + 1) + sh->mpr];
            most = tot > most ? tot : most;
        }
        if (most > slab) {
            lsn::set_error("lsnShardStep: a rank reports %lld survivors, more than its %lld pixels", most, slab);
            return -1;
        }
        slab = (most + 7) & ~7ll;   // the colour stream of every rank then starts 8-byte aligned
    }
    const int C = sh->padded ? 1 : sh->chunks;
    if (C <= 1) {
        sh->last_slab = slab;
        sh->last_bytes_per_rank = (long long)(sizeof(int) * off_ints + sizeof(int) * T * sh->tiles_loc + T * cap / 8 + (size_t)slab * 5);
        {
            NcclGroup grp(r);
            LSN_NCCL(grp.start());
            LSN_NCCL(r->AllGather(sh->local->tile_counts.p, sh->g_tp.p, T * sh->tiles_loc, ncclInt32, sh->comm, s));
            LSN_NCCL(r->AllGather(sh->mask.p, sh->g_mask.p, T * cap / 8, ncclUint8, sh->comm, s));
            LSN_NCCL(r->AllGather(sh->depth_c.p, sh->g_dc.p, (size_t)slab * 2, ncclUint8, sh->comm, s));
            LSN_NCCL(r->AllGather(sh->rgb_c.p, sh->g_cc.p, (size_t)slab * 3, ncclUint8, sh->comm, s));
            LSN_NCCL(grp.end());
        }
        if (lsn::reconstruct(sh->whole, sh->world, sh->mpr, sh->g_mask.p, sh->g_dc.p, sh->g_cc.p, slab, sh->g_tp.as<int>(), sh->g_off.as<int>(),
                             sh->merged.p, sh->merged_off.as<int>(), sh->g_tick_base.as<int>(), stream))
            return -1;
    } else {
        // Pipelined: the tile prefixes and the masks travel first (every chunk needs them); then the streams of tick group c
        // cross the links on the communication stream while the caller's stream reconstructs group c - 1.  Every rank cuts the
        // ticks the same way and derives the same chunk lengths from the same gathered offset tables.
        const int per = (int)((T + C - 1) / C);
        auto count_of = [&](size_t q, size_t k) { return (long long)sh->h_goff[(q * T + k) * (sh->mpr + 1) + sh->mpr]; };
        {
            NcclGroup grp(r);
            LSN_NCCL(grp.start());
            LSN_NCCL(r->AllGather(sh->local->tile_counts.p, sh->g_tp.p, T * sh->tiles_loc, ncclInt32, sh->comm, s));
            LSN_NCCL(r->AllGather(sh->mask.p, sh->g_mask.p, T * cap / 8, ncclUint8, sh->comm, s));
            LSN_NCCL(grp.end());
        }
        LSN_HIP(hipEventRecord(sh->ev_pre, s));                        // pack and the small gathers are done: the streams may be read
        LSN_HIP(hipStreamWaitEvent(sh->comm_stream, sh->ev_pre, 0));
        long long my_start = 0, recv_off = 0, sent = 0;
        bool first = true;
        for (int c = 0; c < C; c++) {
            const int t0 = c * per, t1 = (int)std::min<size_t>(T, (size_t)(c + 1) * per);
            if (t0 >= t1) break;
            long long most = 1, mine = 0;
            for (size_t q = 0; q < W; q++) {
                long long len = 0;
                for (int k = t0; k < t1; k++) len += count_of(q, (size_t)k);
                most = len > most ? len : most;
                if ((int)q == sh->rank) mine = len;
            }
            const long long slab_c = (most + 7) & ~7ll;
            {
                NcclGroup grp(r);
                LSN_NCCL(grp.start());
                LSN_NCCL(r->AllGather(sh->depth_c.as<unsigned short>() + my_start, sh->g_dc.as<unsigned short>() + recv_off, (size_t)slab_c * 2, ncclUint8,
                                      sh->comm, sh->comm_stream));
                LSN_NCCL(r->AllGather(sh->rgb_c.as<unsigned char>() + 3 * my_start, sh->g_cc.as<unsigned char>() + 3 * recv_off, (size_t)slab_c * 3,
                                      ncclUint8, sh->comm, sh->comm_stream));
                LSN_NCCL(grp.end());
            }
            LSN_HIP(hipEventRecord(sh->ev_chunk[c], sh->comm_stream));
            LSN_HIP(hipStreamWaitEvent(s, sh->ev_chunk[c], 0));
            if (lsn::reconstruct(sh->whole, sh->world, sh->mpr, sh->g_mask.p, sh->g_dc.as<unsigned short>() + recv_off,
                                 sh->g_cc.as<unsigned char>() + 3 * recv_off, slab_c, sh->g_tp.as<int>(), sh->g_off.as<int>(), sh->merged.p,
                                 sh->merged_off.as<int>(), sh->g_tick_base.as<int>(), stream, t0, t1 - t0, first))
                return -1;
            first = false;
            my_start += mine;
            recv_off += (long long)W * slab_c;
            sent += slab_c;
        }
        sh->last_slab = sent;
        sh->last_bytes_per_rank = (long long)(sizeof(int) * off_ints + sizeof(int) * T * sh->tiles_loc + T * cap / 8 + (size_t)sent * 5);
    }
    if (d_merged) *d_merged = sh->merged.p;
    if (d_merged_offsets) *d_merged_offsets = sh->merged_off.as<int>();
    return 0;
}
