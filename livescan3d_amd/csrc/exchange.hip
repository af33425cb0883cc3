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

