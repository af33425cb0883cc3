// host_flows.hip -- how a call of the reference's exports on HOST arrays runs: the process-wide context (host_ctx.hpp) and the call flows.
//
// LiveScanServer hands over host arrays (pinned managed arrays / AllocHGlobal blocks, KinectServer.cs:354-374,
// MainWindowForm.cs:364-370) and reads host memory back (Marshal.Copy, KinectServer.cs:383), so these entry points
// add the H2D / D2H hops around the same kernels bench.py drives directly on HBM-resident data.
// One process-wide context (device from $LSN_DEVICE, default 0): the merge / single-sensor / radial calls share one set of
// device buffers and serialise on one lock; ICP has its own buffers, stream and lock, so a multi-millisecond refine call does
// not hold up the merge calls (LiveScanServer runs them on two BackgroundWorkers, MainWindowForm.cs:238,304).
//
// A merge call is PCIe-bound (8 x 512x424: 8.7 MB up, 15-35 MB down, against ~30 us of kernels), so the call is laid out around the
// link (round 4; numbers from tools/link_probe.hip on the MI355X box, DESIGN.md "host path"):
//   * OUTPUT: the write kernels store the vertices and the triangles STRAIGHT INTO the pinned host blocks that become
//     Mesh::vertices / Mesh::triangles (hipHostMalloc memory is device-visible).  16-byte stores of consecutive lanes cross the
//     link at the rate of the copy engine (55 GB/s), but need no length in advance -- so there is no count round trip, no
//     download to issue, and the bytes start to leave as soon as the first sensors have been fused;
//   * INPUT: the caller's arrays are pageable (C# pins them, the runtime does not know).  A pageable hipMemcpy of >= 1 MiB pins the
//     pages in place and runs at ~52 GB/s but keeps the calling thread until it is done; below 1 MiB the runtime stages through a
//     bounce buffer at ~15 GB/s.  So the frames go up in runs of >= 1 MiB (make_schedule), and a GROUP of sensors is launched the
//     moment its frames are there, storing to the host while the next run is on its way up -- both directions of the link busy;
//   * while a kernel streams to host memory NO other kernel completes, on any stream (probe F: a 4-workgroup kernel launched beside a
//     15 MB store kernel finishes with it), and every dependent launch between two storing kernels is time in which nothing
//     crosses the link.  So a group is ONE launch: the single-pass form of the fusion (fuse_kernel<4>: a tile keeps its vertices
//     in registers, publishes its count, finds its offset by look-back -- over the tiles of the earlier groups too), no count
//     kernel, no scan; the triangle passes run once, behind the last group, over the whole tick;
//   * so kernel stores are the form for ONE small group (a single-sensor call: one launch, no count round trip).  Calls of several
//     groups, and calls that start with the radial correction (~100 us of latency-bound closing rounds per group), build the mesh
//     in HBM and let ASYNCHRONOUS COPIES (hipMemcpyAsync into the pinned block, own stream) take it home group by group (fuse_host_grouped): they run beside the kernels (in the trace they are __amd_rocclr_copyBuffer blit kernels; the closing rounds beside them stretch from 72 to ~150 us), and the
//     length a DMA needs comes from a pinned word the group's last tile writes, read behind the launch's event;
//   * registering the caller's arrays (hipHostRegister) was measured in rounds 2 and 4: 1.4 ms to register 8.7 MB, copies from
//     registered memory 43 GB/s, kernel loads from it 40 GB/s -- slower than the pageable copy; the opt-in cache of round 2 is gone.
// $LSN_HOST_PATH=direct / grouped forces one flow for every call (A/B runs); $LSN_HOST_GROUP=n fixes the sensors per group.
#include "host_ctx.hpp"

namespace lsn {
namespace host {

// the lane of the calling thread's own last mesh call (lsnLastMesh* read that lane's mesh; include/NativeUtils.h)
thread_local Lane *t_last_lane = nullptr;

Ctx &ctx()
{
    static Ctx *c = new Ctx();   // never destroyed: its HIP objects must not be released from a static destructor after the runtime is gone
    return *c;
}

// "0,1,2" -> devices; every entry must name a visible device, an entry may repeat (two shards on one device: the rehearsal a one-GPU box allows)
int parse_device_list(const char *text, int n_visible, std::vector<int> &out)
{
    out.clear();
    const char *p = text;
    while (*p) {
        char *end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p || v < 0 || v >= n_visible || (int)out.size() >= kMaxShards) {
            lsn::set_error("NativeUtils: LSN_HOST_DEVICES=%s: expected up to %d comma-separated device numbers below %d", text, kMaxShards, n_visible);
            return -1;
        }
        out.push_back((int)v);
        p = end;
        while (*p == ',' || *p == ' ') p++;
    }
    return 0;
}

// Takes c.init_mu itself; callers may hold c.mu or c.icp_mu.
int ensure_ready(Ctx &c)
{
    std::lock_guard<std::mutex> g(c.init_mu);
    if (c.ready) return hipSetDevice(c.device) == hipSuccess ? 0 : -1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        lsn::set_error("NativeUtils: no HIP device is available -- this library has no CPU path");
        return -1;
    }
    const char *env = getenv("LSN_DEVICE");
    c.device = env ? atoi(env) : 0;
    if (c.device < 0 || c.device >= n) {
        lsn::set_error("NativeUtils: LSN_DEVICE=%d but %d device(s) are visible", c.device, n);
        return -1;
    }
    LSN_HIP(hipSetDevice(c.device));
    LSN_HIP(hipStreamCreateWithFlags(&c.icp_stream, hipStreamNonBlocking));
    std::vector<int> shard_devices;
    if (const char *e = getenv("LSN_HOST_DEVICES")) {
        if (parse_device_list(e, n, shard_devices)) return -1;
        if (shard_devices.size() < 2) shard_devices.clear();   // one device: nothing to shard over
    }
    std::vector<Lane *> lanes = {&c.merge, &c.single};
    c.merge.device = c.single.device = c.device;
    for (int dev : shard_devices) {
        HostShard *sh = new HostShard();
        sh->lane.device = dev;
        c.shards.push_back(sh);
        lanes.push_back(&sh->lane);
    }
    for (Lane *l : lanes) {
        LSN_HIP(hipSetDevice(l->device));
        for (hipStream_t *s : {&l->stream, &l->up, &l->down, &l->back}) LSN_HIP(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
        LSN_HIP(hipEventCreateWithFlags(&l->ev_tri, hipEventDisableTiming));
        for (int i = 0; i < kMaxGroups; i++) LSN_HIP(hipEventCreateWithFlags(&l->ev_group[i], hipEventDisableTiming));
    }
    LSN_HIP(hipSetDevice(c.device));
    if (const char *e = getenv("LSN_HOST_PATH")) c.host_path = !strcmp(e, "direct") ? 1 : (!strcmp(e, "grouped") || !strcmp(e, "copy")) ? 2 : 0;
    if (const char *e = getenv("LSN_HOST_GROUP")) c.group_override = atoi(e);
    c.ready = true;
    return 0;
}

// waits for everything the context has in flight on the merge streams (error paths: no copy may touch the caller's arrays
// or our buffers once the call has returned)
void drain(Lane &l)
{
    (void)hipStreamSynchronize(l.up);
    (void)hipStreamSynchronize(l.stream);
    (void)hipStreamSynchronize(l.down);
    (void)hipStreamSynchronize(l.back);
    (void)hipGetLastError();
}

void *pinned_get(Ctx &c, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    lsn::test_fault_point(1);
    std::lock_guard<std::mutex> tg(c.tab_mu);
    auto it = c.pool.lower_bound(bytes);
    if (it != c.pool.end() && it->first <= bytes * 2 + 4096) {
        void *p = it->second;
        c.live[p] = it->first;
        c.pool.erase(it);
        return p;
    }
    void *p = nullptr;
    size_t cap = (bytes + 4095) & ~(size_t)4095;
    // portable: every device of $LSN_HOST_DEVICES stores into the same block
    if (hipHostMalloc(&p, cap, c.shards.empty() ? hipHostMallocDefault : hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        lsn::set_error("NativeUtils: hipHostMalloc(%zu) failed", cap);
        return nullptr;
    }
    try {
        c.live[p] = cap;
    } catch (...) {   // the table cannot take it (its node allocation failed): the block is freed, not lost (the mirror of pinned_put)
        (void)hipHostFree(p);
        throw;
    }
    return p;
}

void pinned_put(Ctx &c, void *p)
{
    std::lock_guard<std::mutex> tg(c.tab_mu);
    auto it = c.live.find(p);
    if (it == c.live.end()) return;  // not ours: leave it alone
    size_t cap = it->second;
    c.live.erase(it);
    // keep a handful of blocks around, free the rest (and a block the table cannot take -- its node allocation failed -- is freed, not lost)
    if (c.pool.size() >= 8) {
        (void)hipHostFree(p);
        return;
    }
    try {
        c.pool.emplace(cap, p);
    } catch (...) {
        (void)hipHostFree(p);
    }
}

// The reference hands out `new int[0]` when a mesh has no triangles (depthprocessing.cpp:1640): a valid pointer that is never
// dereferenced (KinectServer.cs:344-345).  Here it is one static word: nothing to allocate, nothing to track, and deleteMesh leaves it
// alone like every pointer it does not own.
int g_no_triangles[1] = {0};

void empty_mesh(Mesh *m) noexcept
{
    m->nVertices = 0;
    m->vertices = nullptr;
    m->nTriangles = 0;
    m->triangles = g_no_triangles;
}

// The lane's cached single-tick plan for sensors [first, first + count) of a call.  The lane's lock is held.
LsnFusion *get_plan(Ctx &c, Lane &l, const int *widths, const int *heights, int first, int count)
{
    // `first` is part of the key: generateVerticesFromDepthMap is called for sensor 0, 1, ... in turn (KinectServer.cs:527-554) and every
    // sensor keeps its own plan, so its calibration stays set (no parameter upload, no table rebuild per call) and its count pass
    // can run from the per-pixel depth thresholds from the second round on
    std::vector<int> key;
    key.push_back(count);
    key.push_back(first);
    for (int i = 0; i < count; i++) key.push_back(widths[first + i]);
    for (int i = 0; i < count; i++) key.push_back(heights[first + i]);
    auto it = l.plans.find(key);
    if (it != l.plans.end()) return it->second;
    if (l.plans.size() > 64) {
        // unbounded variety of geometries: start over.  Only this lane's plans, with nothing of this lane in flight (its lock is
        // held, its streams are drained here), and none of the plans the call in progress has already picked
        drain(l);
        for (auto kv = l.plans.begin(); kv != l.plans.end();) {
            bool in_use = kv->second == l.last_plan;
            for (const Group &g : l.groups) in_use |= g.radial_plan == kv->second;
            if (in_use) {
                ++kv;
                continue;
            }
            lsnFusionDestroy(kv->second);
            kv = l.plans.erase(kv);
        }
    }
    LsnFusion *plan = lsnFusionCreate(l.device, 1, count, widths + first, heights + first);
    if (!plan) return nullptr;
    try {
        l.plans[key] = plan;
    } catch (...) {
        lsnFusionDestroy(plan);
        throw;
    }
    return plan;
}

// The pinned offset-table mirrors of a lane hold at least n ints each.
int ensure_tables(Lane &l, int n)
{
    if (l.h_off_cap >= n) return 0;
    if (l.h_off) (void)hipHostFree(l.h_off);
    if (l.h_toff) (void)hipHostFree(l.h_toff);
    l.h_off = l.h_toff = nullptr;
    l.h_off_cap = 0;
    LSN_HIP(hipHostMalloc((void **)&l.h_off, sizeof(int) * (size_t)(n + 64), hipHostMallocPortable));
    LSN_HIP(hipHostMalloc((void **)&l.h_toff, sizeof(int) * (size_t)(n + 64), hipHostMallocPortable));
    l.h_off_cap = n + 64;
    return 0;
}

// The upload schedule of a call.  A pageable copy of >= 1 MiB is pinned in place by the runtime (~52 GB/s, ~10 us of fixed cost),
// a smaller one is staged through a bounce buffer at a quarter of the rate, so frames go up in runs of at least 1 MiB, and a group
// of sensors is launched as soon as its frames are there, i.e. after `ready_after` copies.  How many sensors make a group
// (measured, 8 x 512x424, gpurun_out/r04/host_ab8.txt):
//   * merge calls (kernel stores): every storing launch costs ~8 us of ramp and drain, every group one depth and one colour copy --
//     groups are the >= 1 MiB runs of the DEPTH array (three sensors): 0.417-0.420 / 0.740 ms against 0.429-0.432 / 0.753 for
//     groups of two (the runs of the colour array) and 0.427 / 0.752 for groups of four.  D[0-2] C[0-2] | D[3-7] C[3-5] | C[6-7];
//   * calls that start with the radial correction: a group pays ~100 us of latency-bound closing rounds whatever its size --
//     groups of >= 2.5 MB of colours (four sensors): 1.05 ms against 1.10 (three) and 1.16 (two); the first group three (round 5: 1.03).
//     D[0-2] C[0-2] | D[3-7] C[3-7].
void plan_schedule(std::vector<Group> &groups, std::vector<Copy> &copies, const int *widths, const int *heights, int first, int count, bool radial,
                   int group_override, bool small_first)
{
    groups.clear();
    copies.clear();
    constexpr size_t kPinnedCopy = (size_t)1 << 20;
    constexpr size_t kRadialGroup = 2500000;   // colours per group of a call that starts with the radial correction: four 512x424 sensors
    constexpr size_t kRadialFirstGroup = 1900000;   // ... and of its first group: three
    const int end = first + count;
    auto dsz = [&](int i) { return (size_t)widths[i] * heights[i] * 2; };
    auto csz = [&](int i) { return (size_t)widths[i] * heights[i] * 3; };
    size_t d_src0 = 0, c_src0 = 0;
    for (int i = 0; i < first; i++) {   // sensor `first` starts after the frames before it (depthprocessing.cpp:1646-1650)
        d_src0 += dsz(i);
        c_src0 += csz(i);
    }
    int per = group_override > 0 ? group_override : 0;
    if (per == 0 && count > kMaxGroups) {
        size_t cb = 0;
        for (int i = first; i < end; i++) cb += csz(i);
        if (cb / kMaxGroups >= kRadialGroup) per = (count + kMaxGroups - 1) / kMaxGroups;   // many big sensors: at most kMaxGroups groups
    }
    size_t d_off = 0, c_off = 0;
    for (int i = first; i < end;) {
        Group g;
        g.first = i;
        g.d_off = d_off; g.c_off = c_off;
        g.d_src = d_src0 + d_off; g.c_src = c_src0 + c_off;
        auto weight = [&](const Group &q) { return radial ? q.cbytes : q.dbytes; };   // what a group is sized by
        const size_t full = radial ? kRadialGroup : kPinnedCopy;
        // small_first: the FIRST group of a call that starts with the radial correction AND goes on to the fusion is a little smaller (>= 1.9 MB of colours: three 512x424 sensors):
        // nothing leaves for the host before the first group is up, corrected and fused, and the groups behind it hide their ~100 us of closing
        // rounds behind its download anyway (8 x 512x424, tick as one call: 3 | 5 sensors 1.025-1.034 ms against 4 | 4 1.041-1.050; 2 | 4 | 2
        // 1.08, 1 | 4 | 3 1.14 -- $LSN_HOST_FIRST_GROUP=n forces n sensors for the A/B).  The radial export ALONE keeps equal groups: nothing
        // leaves before a group is corrected there either, but its way home is as long as its way up (0.445 ms with 4 | 4, 0.454 with 3 | 5)
        static const int first_n = getenv("LSN_HOST_FIRST_GROUP") ? atoi(getenv("LSN_HOST_FIRST_GROUP")) : 0;
        const int cut = (radial && small_first && first_n > 0 && groups.empty()) ? first_n : 0;
        const size_t want = (radial && small_first && groups.empty()) ? kRadialFirstGroup : full;
        while (i < end && (cut > 0 ? g.count < cut : per > 0 ? g.count < per : (g.count == 0 || weight(g) < want))) {
            g.dbytes += dsz(i);
            g.cbytes += csz(i);
            g.count++;
            i++;
        }
        d_off += g.dbytes;
        c_off += g.cbytes;
        if (per == 0 && !groups.empty() && i == end && weight(g) * 2 < full) {
            Group &b = groups.back();   // a short tail: one more bounce-buffer copy would cost more than the overlap wins
            b.count += g.count;
            b.dbytes += g.dbytes;
            b.cbytes += g.cbytes;
        } else {
            groups.push_back(g);
        }
    }
    // a group's radial correction works on its slice of the buffers through a plan of its own: the slices must keep the alignment
    // the wide-load kernels ask for (16 B depth, 8 B colours); and there are kMaxGroups events.  A rig that breaks either goes as
    // one group
    bool ok = groups.size() <= (size_t)kMaxGroups;
    for (const Group &g : groups) ok &= (g.d_off % 16) == 0 && (g.c_off % 8) == 0;
    if (!ok && groups.size() > 1) {
        Group all = groups.front();
        for (size_t k = 1; k < groups.size(); k++) {
            all.count += groups[k].count;
            all.dbytes += groups[k].dbytes;
            all.cbytes += groups[k].cbytes;
        }
        groups.assign(1, all);
    }
    // copies: depth runs interleaved with the groups' colour runs
    int depth_to = first;          // sensors [first, depth_to) have their depth scheduled
    size_t depth_off = 0;
    for (Group &g : groups) {
        if (depth_to < g.first + g.count) {
            Copy d;
            d.dev_off = depth_off;
            d.src_off = d_src0 + depth_off;
            int to = depth_to;
            while (to < end && (to < g.first + g.count || d.bytes < kPinnedCopy)) d.bytes += dsz(to++);
            size_t rest = 0;
            for (int i = to; i < end; i++) rest += dsz(i);
            if (rest > 0 && rest < kPinnedCopy) {   // what is left would be a bounce-buffer copy: take it along
                d.bytes += rest;
                to = end;
            }
            copies.push_back(d);
            depth_to = to;
            depth_off += d.bytes;
        }
        Copy cc;
        cc.colours = true;
        cc.dev_off = g.c_off;
        cc.src_off = g.c_src;
        cc.bytes = g.cbytes;
        copies.push_back(cc);
        g.ready_after = (int)copies.size();
    }
}

int make_schedule(Ctx &c, Lane &l, const int *widths, const int *heights, int first, int count, bool radial, bool small_first)
{
    plan_schedule(l.groups, l.copies, widths, heights, first, count, radial, c.group_override, small_first);
    if (radial)
        for (Group &g : l.groups) {
            g.radial_plan = get_plan(c, l, widths, heights, g.first, g.count);
            if (!g.radial_plan) return -1;
        }
    return 0;
}

// $LSN_HOST_TRACE=1: wall-clock marks of the phases of a direct call, printed for three calls once the first ten have gone by
struct PhaseTrace {
    bool on = false;
    int n = 0;
    const char *name[64];
    double t[64];
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void mark(const char *what)
    {
        if (on && n < 64) {
            name[n] = what;
            t[n++] = now();
        }
    }
    void print() const
    {
        if (!on || n == 0) return;
        fprintf(stderr, "[NativeUtils trace]");
        for (int i = 1; i < n; i++) fprintf(stderr, " %s +%.1f", name[i], 1e6 * (t[i] - t[i - 1]));
        fprintf(stderr, " | total %.1f us\n", 1e6 * (t[n - 1] - t[0]));
    }
};

// Everything one merge / single-sensor / correct-and-merge call holds while it runs, and the steps its two flows share: the plan and
// the upload schedule, the lane's device buffers, the pinned blocks that become Mesh::vertices / Mesh::triangles, the upload loop that
// hands every group of sensors to the flow the moment its frames are on the device, the hand-over to the caller.
// The destructor IS the error path: unless commit() has handed the blocks to the caller, it drains the lane -- no kernel may store to,
// and no copy land in, a block that goes back to the pool, and no kernel may still write the lane's pinned tables when the lock is
// released -- and returns the blocks.  Whatever ended the call takes it: a failed HIP call, a refused launch, or an exception
// (std::bad_alloc from a table that grows, a test's fault injection) on its way to the trampoline.  The lane's lock is held throughout.
struct HostCall {
    Ctx &c;
    Lane &l;
    const unsigned char *depth_maps, *depth_colors;
    const int *widths, *heights;
    const float *intr, *wt, *bounds6;
    int first, count;                       // the sensors of the call (generateVerticesFromDepthMap uses one)
    bool with_triangles, radial;            // radial: the call starts with the radial correction of the frames, on the device
    unsigned char *back_d, *back_c;         // optional: the corrected maps are also copied to these host arrays, like the separate export does
    LsnFusion *plan = nullptr;
    size_t G = 0;
    long long cap = 0;
    void *host = nullptr, *host_tri = nullptr;
    bool back = false, committed = false;
    const char *run_d = nullptr, *run_c = nullptr;   // what the fusion and triangle launches read: the raw frames, or the corrected ones
    PhaseTrace tr;

    HostCall(Ctx &c_, Lane &l_, const unsigned char *dm, const unsigned char *dc, const int *w, const int *h, const float *in, const float *wtp,
             const float *b6, int first_, int count_, bool tri, bool rad, unsigned char *bd, unsigned char *bc)
        : c(c_), l(l_), depth_maps(dm), depth_colors(dc), widths(w), heights(h), intr(in), wt(wtp), bounds6(b6), first(first_), count(count_),
          with_triangles(tri), radial(rad), back_d(bd), back_c(bc)
    {
    }
    HostCall(const HostCall &) = delete;
    HostCall &operator=(const HostCall &) = delete;
    ~HostCall()
    {
        if (committed) return;
        drain(l);
        if (host) pinned_put(c, host);
        if (host_tri) pinned_put(c, host_tri);
    }

    // Plan, schedule, buffers, the vertex block.  in_hbm: the mesh is built in the lane's HBM buffers (the copy-engine flow), else the
    // kernels store into the pinned blocks and the triangle block is taken here as well; extra_tab: pinned table words beyond
    // offsets [count + 1] and the give-up flag.
    int begin(bool in_hbm, int extra_tab)
    {
        l.last_nv = -1;
        l.last_plan = nullptr;
        l.last_sharded = false;
        static const bool trace_env = getenv("LSN_HOST_TRACE") && atoi(getenv("LSN_HOST_TRACE")) != 0;
        static std::atomic<int> trace_calls{0};
        if (trace_env) {
            const int k = trace_calls++;
            tr.on = k >= 10 && k < 13;
        }
        tr.mark("enter");
        l.groups.clear();
        plan = get_plan(c, l, widths, heights, first, count);
        if (!plan) return -1;
        l.last_plan = plan;
        if (make_schedule(c, l, widths, heights, first, count, radial, radial)) return -1;
        G = l.groups.size();
        size_t dbytes = 0, cbytes = 0;
        for (const Group &g : l.groups) {
            dbytes += g.dbytes;
            cbytes += g.cbytes;
        }
        cap = lsnFusionTickCapacity(plan);
        if (l.d_depth.reserve(dbytes + 16) || l.d_colors.reserve(cbytes + 16) || l.d_off.reserve(sizeof(int) * (size_t)(count + 1)) ||
            l.d_tri_off.reserve(sizeof(int) * (size_t)(count + 1)) || ensure_tables(l, count + 2 + extra_tab))
            return -1;
        if (radial && (l.d_depth2.reserve(dbytes + 16) || l.d_colors2.reserve(cbytes + 16))) return -1;
        if (in_hbm && (l.d_out.reserve((size_t)cap * 16) || (with_triangles && l.d_tri.reserve((size_t)cap * 2 * 12)))) return -1;
        back = radial && back_d && back_c;
        // the mesh's host blocks, sized for the most the frames can give (recycled through the pool: the same blocks tick after tick);
        // capacity-sized because the first vertices leave before the count is known
        host = pinned_get(c, (size_t)cap * sizeof(VertexC4ubV3f));
        if (!host) return -1;
        if (!in_hbm && with_triangles && !(host_tri = pinned_get(c, (size_t)cap * 2 * 12))) return -1;
        l.h_off[count] = -1;       // the total: stored by the last tile of the last group
        l.h_off[count + 1] = 0;    // the look-back's give-up flag
        l.h_toff[count] = 0;
        if (lsnFusionSetParams(plan, intr + 7 * first, wt + 12 * first, bounds6, l.stream)) return -1;
        run_d = radial ? l.d_depth2.as<char>() : l.d_depth.as<char>();
        run_c = radial ? l.d_colors2.as<char>() : l.d_colors.as<char>();
        tr.mark("setup");
        return 0;
    }

    // The upload schedule, run by run; group k is handed to on_group(k) -- corrected first, if the call asks for it -- as soon as its
    // frames have landed.  One group (a single sensor: ~1 MB up, ~2 MB down, all fixed latency; or frames the scheme cannot cut): its
    // copies go asynchronously on the kernels' stream.  Several groups: every copy blocks on the upload stream until the bytes are
    // there, so the launch behind it needs no event, and runs while the next copy is on its way up.
    template <class F>
    int upload(F &&on_group)
    {
        size_t next_group = 0;
        for (size_t i = 0; i < l.copies.size(); i++) {
            const Copy &cp = l.copies[i];
            char *dst = (cp.colours ? l.d_colors.as<char>() : l.d_depth.as<char>()) + cp.dev_off;
            const unsigned char *src = (cp.colours ? depth_colors : depth_maps) + cp.src_off;
            const hipError_t e = G == 1 ? hipMemcpyAsync(dst, src, cp.bytes, hipMemcpyHostToDevice, l.stream)
                                        : hipMemcpyWithStream(dst, src, cp.bytes, hipMemcpyHostToDevice, l.up);
            if (e != hipSuccess) {
                lsn::set_error("NativeUtils: upload failed: %s", hipGetErrorString(e));
                return -1;
            }
            tr.mark(cp.colours ? "upC" : "upD");
            for (; next_group < G && l.groups[next_group].ready_after == (int)i + 1; next_group++) {
                const Group &g = l.groups[next_group];
                // out of place: raw frames in d_depth / d_colors, corrected ones in the second pair, which the launches read
                if (radial && lsnFusionRadialCorrectTo(g.radial_plan, intr + 7 * g.first, l.d_depth.as<char>() + g.d_off, l.d_colors.as<char>() + g.c_off,
                                                       l.d_depth2.as<char>() + g.d_off, l.d_colors2.as<char>() + g.c_off, l.stream))
                    return -1;
                if (on_group(next_group, g)) return -1;
                tr.mark("launch");
            }
        }
        return 0;
    }

    // One launch, single pass, over group k's sensors, continuing where group k - 1 stopped (fusion.hip: run_frames).
    int fuse_group(size_t k, const Group &g, void *vertices, int *group_end_mirror, bool host_out)
    {
        return lsn::run_frames(plan, run_d, run_c, vertices, l.d_off.as<int>(), g.first - first, g.first - first + g.count, k == 0, with_triangles, l.h_off,
                               group_end_mirror, host_out, l.stream);
    }

    // The mesh is the caller's from here on (deleteMesh returns its blocks to the pool).
    void commit(Mesh *out, int nv, int nt, bool in_hbm)
    {
        if (nt == 0 && host_tri) {
            pinned_put(c, host_tri);
            host_tri = nullptr;
        }
        l.last_nv = nv;
        l.last_nt = nt;
        l.last_in_hbm = in_hbm;
        l.last_radial = radial;
        l.last_tri = with_triangles;
        out->nVertices = nv;
        out->vertices = static_cast<VertexC4ubV3f *>(host);
        out->nTriangles = nt;
        out->triangles = nt > 0 ? static_cast<int *>(host_tri) : g_no_triangles;
        committed = true;
        c.last_lane.store(&l);
        t_last_lane = &l;
        tr.mark("done");
        tr.print();
    }
};

// The corrected maps of group k go home in the runs of the upload schedule (>= 1 MiB each: a smaller pageable copy is staged through
// a bounce buffer): every run whose last group is k.  Pageable destination: these copies keep the thread; on their own stream.
int write_back_runs(Lane &l, size_t k, unsigned char *back_d, unsigned char *back_c)
{
    const size_t G = l.groups.size();
    for (const Copy &cp : l.copies) {
        size_t last = 0;   // the last group this run covers
        for (size_t q = 0; q < G; q++)
            if ((cp.colours ? l.groups[q].c_off : l.groups[q].d_off) < cp.dev_off + cp.bytes) last = q;
        if (last != k) continue;
        unsigned char *dst = (cp.colours ? back_c : back_d) + cp.src_off;
        const char *src = (cp.colours ? l.d_colors2.as<char>() : l.d_depth2.as<char>()) + cp.dev_off;
        LSN_HIP(hipMemcpyWithStream(dst, src, cp.bytes, hipMemcpyDeviceToHost, l.back));
    }
    return 0;
}

// Flow 1: the kernels store the vertices and the triangles straight into the mesh's host blocks (file comment).
int fuse_host_direct(HostCall &h, Mesh *out)
{
    Lane &l = h.l;
    if (h.begin(false, 0)) return -1;
    if (h.upload([&](size_t k, const Group &g) -> int {
            if (h.back && hipEventRecord(l.ev_group[k], l.stream) != hipSuccess) return -1;   // "group k's corrected maps are final"
            return h.fuse_group(k, g, h.host, nullptr, true);
        }))
        return -1;
    if (h.with_triangles && lsn::run_triangles(h.plan, h.run_d, h.host_tri, l.d_tri_off.as<int>(), l.h_toff, true, l.stream)) return -1;
    if (h.back) {
        // the corrected maps go home group by group (copy engine, pageable destination: each copy blocks) while the launches run
        for (size_t k = 0; k < h.G; k++) {
            const Group &b = l.groups[k];
            if (hipStreamWaitEvent(l.down, l.ev_group[k], 0) != hipSuccess ||
                hipMemcpyWithStream(h.back_d + b.d_src, l.d_depth2.as<char>() + b.d_off, b.dbytes, hipMemcpyDeviceToHost, l.down) != hipSuccess ||
                hipMemcpyWithStream(h.back_c + b.c_src, l.d_colors2.as<char>() + b.c_off, b.cbytes, hipMemcpyDeviceToHost, l.down) != hipSuccess) {
                lsn::set_error("NativeUtils: write-back of the corrected maps failed: %s", hipGetErrorString(hipGetLastError()));
                return -1;
            }
            h.tr.mark("back");
        }
    }
    h.tr.mark("queued");
    if (hipStreamSynchronize(l.stream) != hipSuccess) {
        lsn::set_error("NativeUtils: %s", hipGetErrorString(hipGetLastError()));
        return -1;
    }
    h.tr.mark("sync");
    // the kernels left the counts in the pinned tables
    const int nv = l.h_off[h.count], nt = h.with_triangles ? l.h_toff[h.count] : 0;
    if (l.h_off[h.count + 1] != 0) {
        (void)lsnFusionCheck(h.plan, l.stream);   // clears the plan's sticky flag
        lsn::set_error("NativeUtils: the single-pass fusion gave up on a predecessor tile (look-back spin limit)");
        return -1;
    }
    if (nv < 0 || nv > h.cap || nt < 0 || nt > 2 * h.cap) {
        lsn::set_error("NativeUtils: device returned impossible counts (%d vertices, %d triangles)", nv, nt);
        return -1;
    }
    h.commit(out, nv, nt, false);
    return 0;
}

// Flow 2: the mesh is built in HBM and asynchronous copies take it home -- the form for calls that start with the radial correction.
// While a kernel streams to host memory no other kernel completes (file comment), so with several groups the storing launches
// serialise with everything else; a copy on another stream does not have that problem to the same degree: group g's vertices leave
// (pinned destination, asynchronous) while group g+1 uploads, is corrected and fused.  What a DMA needs and a storing kernel does not
// is a LENGTH: the last tile of a group's launch leaves the group's end offset in a pinned word, and the host reads it once the
// launch's event has fired -- by then it has uploaded the next group, so the wait is short or none.
int fuse_host_grouped(HostCall &h, Mesh *out)
{
    Lane &l = h.l;
    if (h.begin(true, (int)kMaxGroups)) return -1;   // table: offsets [count + 1], give-up flag, then the groups' end offsets
    const int count = h.count;
    int *h_end = l.h_off + count + 2;
    int sent = 0;   // vertices already on their way home
    // group k's launches have been enqueued: when its event has fired, its vertices go home
    auto service = [&](size_t k) -> int {
        const Group &g = l.groups[k];
        LSN_HIP(hipEventSynchronize(l.ev_group[k]));
        const int end = h_end[k];
        if (l.h_off[count + 1] != 0 || end < sent || end > h.cap) {
            lsn::set_error("NativeUtils: the fusion of sensors %d..%d failed on the device (end offset %d, flag %d)", g.first, g.first + g.count - 1, end,
                           l.h_off[count + 1]);
            return -1;
        }
        if (end > sent)
            LSN_HIP(hipMemcpyAsync(static_cast<char *>(h.host) + (size_t)sent * 16, l.d_out.as<char>() + (size_t)sent * 16, (size_t)(end - sent) * 16,
                                   hipMemcpyDeviceToHost, l.down));
        sent = end;
        h.tr.mark("down");
        return 0;
    };
    // the corrected maps only start home once every upload and launch of the call has been issued (their copies keep the thread)
    auto write_back = [&](size_t k) -> int {
        const int rc = write_back_runs(l, k, h.back_d, h.back_c);
        h.tr.mark("back");
        return rc;
    };
    if (h.upload([&](size_t k, const Group &g) -> int {
            h_end[k] = -1;
            if (h.fuse_group(k, g, l.d_out.p, h_end + k, false) || hipEventRecord(l.ev_group[k], l.stream) != hipSuccess) return -1;
            return k > 0 ? service(k - 1) : 0;   // the group before: done while this one was uploading
        }))
        return -1;
    // the triangle passes over the whole tick, behind the last group (they only read what the groups left in HBM); ev_tri fires when the
    // triangle COUNTS are in the pinned table -- behind the scan, before the write pass -- so the block for the triangles is there and
    // its download queued while the write pass still runs
    if (h.with_triangles && lsn::run_triangles(h.plan, h.run_d, l.d_tri.p, l.d_tri_off.as<int>(), l.h_toff, false, l.stream, l.ev_tri)) return -1;
    // the corrected maps of all groups but the last (their events fired long ago), the last group's vertices, then the triangles
    // (asynchronous) BEFORE the last group's maps, so that the thread-keeping copies share the link with the triangle download
    if (h.back)
        for (size_t k = 0; k + 1 < h.G; k++)
            if (hipEventSynchronize(l.ev_group[k]) != hipSuccess || write_back(k)) return -1;
    if (service(h.G - 1)) return -1;
    const int nv = sent;
    int nt = 0;
    if (h.with_triangles) {
        if (hipEventSynchronize(l.ev_tri) != hipSuccess) {
            lsn::set_error("NativeUtils: %s", hipGetErrorString(hipGetLastError()));
            return -1;
        }
        nt = l.h_toff[count];
        if (nt < 0 || nt > 2 * h.cap) {
            lsn::set_error("NativeUtils: device returned an impossible triangle count %d", nt);
            return -1;
        }
        if (nt > 0) {
            // on the kernels' stream: the copy starts when the triangle write pass, which may still be running, has ended
            h.host_tri = pinned_get(h.c, (size_t)nt * 12);
            if (!h.host_tri || hipMemcpyAsync(h.host_tri, l.d_tri.p, (size_t)nt * 12, hipMemcpyDeviceToHost, l.stream) != hipSuccess) {
                if (!lsn::has_error()) lsn::set_error("NativeUtils: triangle download failed: %s", hipGetErrorString(hipGetLastError()));
                return -1;
            }
        }
        h.tr.mark("tri");
    }
    if (h.back && write_back(h.G - 1)) return -1;
    if (hipStreamSynchronize(l.down) != hipSuccess || hipStreamSynchronize(l.stream) != hipSuccess) {
        lsn::set_error("NativeUtils: %s", hipGetErrorString(hipGetLastError()));
        return -1;
    }
    h.tr.mark("sync");
    if (nv != l.h_off[count]) {
        lsn::set_error("NativeUtils: the groups' end offsets (%d) and the tick's total (%d) disagree", nv, l.h_off[count]);
        return -1;
    }
    h.commit(out, nv, nt, true);
    return 0;
}

// ---- flow 3: the call sharded over several devices ($LSN_HOST_DEVICES) ----------------------------------------------------------------
//
// One GPU's merge call sits at the floor of ONE PCIe link: 8.7 MB up and 15-43 MB down around ~30 us of kernels.
// The consumer of the merged cloud is the host (Marshal.Copy, KinectServer.cs:376-389), so the form in which more GPUs buy anything is
// more LINKS: device d takes the contiguous sensor block d of the tick (the reference's per-sensor fan-out, depthprocessing.cpp:708-733,
// with a device where it has a thread), uploads it over its own link and stores its vertices and triangles over its own link into
// the SAME pinned Mesh blocks -- behind the blocks before it, which is formMesh's sensor order (:1594-1608) and its triangle rebase
// (:1614-1626).  No GPU talks to another one; nothing is gathered.
// What a device needs from the others is where its vertices START, i.e. the others' counts.  So every device runs the two-pass form
// here: count pass (depth only -- it runs while the colours are still on their way up) + scan -> the block's count lands in a pinned
// word -> the worker threads exchange the counts through atomics (each waits only for devices BEFORE it, and nobody waits before it has
// published: no cycle) -> write pass, storing straight to the host block at the block's base.  The triangles repeat the pattern.
// One thread per device, because a pageable upload keeps the thread that issues it.
struct ShardedCall {
    Ctx &c;
    const unsigned char *depth_maps, *depth_colors;
    const int *widths, *heights;
    const float *intr, *wt, *bounds6;
    int count = 0, D = 0;
    bool with_triangles = false, radial = false, back = false;
    bool only_radial = false;                          // depthMapAndColorSetRadialCorrection: correct and write back, no mesh
    unsigned char *back_d = nullptr, *back_c = nullptr;
    int first[kMaxShards + 1] = {};                    // device d owns sensors [first[d], first[d + 1])
    void *host = nullptr, *host_tri = nullptr;
    std::atomic<int> nv[kMaxShards], nt[kMaxShards];   // -1 = not known yet
    std::atomic<int> failed{0};
    std::mutex err_mu;
    char error[lsn::kErrorLen] = {0};

    explicit ShardedCall(Ctx &c_) : c(c_)
    {
        for (int d = 0; d < kMaxShards; d++) {
            nv[d].store(-1);
            nt[d].store(-1);
        }
    }
    void fail(const char *what)   // first failure wins; its text reaches the caller's error channel (the worker's own is thread-local)
    {
        std::lock_guard<std::mutex> g(err_mu);
        if (!failed.load()) snprintf(error, sizeof(error), "%s", what && *what ? what : "a device's part of the call failed");
        failed.store(1);
    }
    // sum of `counts` of the devices before d, or -1 once somebody has failed
    long long base_of(const std::atomic<int> *counts, int d)
    {
        long long base = 0;
        for (int e = 0; e < d; e++) {
            int v;
            while ((v = counts[e].load(std::memory_order_acquire)) < 0) {
                if (failed.load()) return -1;
                std::this_thread::yield();
            }
            base += v;
        }
        return failed.load() ? -1 : base;
    }
};

// How a call of `count` sensors is cut over D devices: contiguous blocks in sensor order, sizes as even as they come.
void plan_shards(int count, int n_devices, int *first, int &D)
{
    D = n_devices < count ? n_devices : count;
    if (D > kMaxShards) D = kMaxShards;
    if (D < 1) D = 1;
    for (int d = 0; d <= D; d++) first[d] = (int)((long long)count * d / D);
}

// Device d's part of a sharded call; runs on worker d (d = 0: on the calling thread).  Returns 0 or -1 with the thread's error text set.
int shard_part(ShardedCall &sc, int d)
{
    Ctx &c = sc.c;
    Lane &l = c.shards[d]->lane;
    LSN_HIP(hipSetDevice(l.device));
    const int f0 = sc.first[d], n = sc.first[d + 1] - f0;
    size_t d_src = 0, c_src = 0, dbytes = 0, cbytes = 0;
    for (int i = 0; i < f0 + n; i++) {
        const size_t px = (size_t)sc.widths[i] * sc.heights[i];
        (i < f0 ? d_src : dbytes) += px * 2;
        (i < f0 ? c_src : cbytes) += px * 3;
    }
    l.groups.clear();
    LsnFusion *plan = get_plan(c, l, sc.widths, sc.heights, f0, n);
    if (!plan) return -1;
    if (sc.only_radial) {
        // the radial export alone: this block up over this device's link, corrected out of place, home again (both ways pageable copies
        // that keep this thread -- which is why every device has one)
        if (l.d_depth.reserve(dbytes + 16) || l.d_colors.reserve(cbytes + 16) || l.d_depth2.reserve(dbytes + 16) || l.d_colors2.reserve(cbytes + 16)) return -1;
        LSN_HIP(hipMemcpyWithStream(l.d_depth.p, sc.depth_maps + d_src, dbytes, hipMemcpyHostToDevice, l.up));
        LSN_HIP(hipMemcpyWithStream(l.d_colors.p, sc.depth_colors + c_src, cbytes, hipMemcpyHostToDevice, l.up));
        if (lsnFusionRadialCorrectTo(plan, sc.intr + 7 * f0, l.d_depth.p, l.d_colors.p, l.d_depth2.p, l.d_colors2.p, l.stream)) return -1;
        LSN_HIP(hipStreamSynchronize(l.stream));
        LSN_HIP(hipMemcpyWithStream(sc.back_d + d_src, l.d_depth2.p, dbytes, hipMemcpyDeviceToHost, l.back));
        LSN_HIP(hipMemcpyWithStream(sc.back_c + c_src, l.d_colors2.p, cbytes, hipMemcpyDeviceToHost, l.back));
        return 0;
    }
    const long long cap = lsnFusionTickCapacity(plan);
    if (l.d_depth.reserve(dbytes + 16) || l.d_colors.reserve(cbytes + 16) || l.d_off.reserve(sizeof(int) * (size_t)(n + 1)) ||
        l.d_tri_off.reserve(sizeof(int) * (size_t)(n + 1)) || ensure_tables(l, n + 2))
        return -1;
    if (sc.radial && (l.d_depth2.reserve(dbytes + 16) || l.d_colors2.reserve(cbytes + 16))) return -1;
    if (lsnFusionSetParams(plan, sc.intr + 7 * f0, sc.wt + 12 * f0, sc.bounds6, l.stream)) return -1;
    const char *run_d = sc.radial ? l.d_depth2.as<char>() : l.d_depth.as<char>();
    const char *run_c = sc.radial ? l.d_colors2.as<char>() : l.d_colors.as<char>();
    l.h_off[n] = -1;
    l.h_toff[n] = -1;
    hipEvent_t ev_counted = l.ev_group[0], ev_corrected = l.ev_group[1];
    // depth up; the count pass (depth only) runs while the colours follow -- unless the call starts with the correction, which needs both
    LSN_HIP(hipMemcpyWithStream(l.d_depth.p, sc.depth_maps + d_src, dbytes, hipMemcpyHostToDevice, l.up));
    if (!sc.radial && lsn::run_count(plan, run_d, run_c, l.d_off.as<int>(), l.h_off, ev_counted, l.stream)) return -1;
    LSN_HIP(hipMemcpyWithStream(l.d_colors.p, sc.depth_colors + c_src, cbytes, hipMemcpyHostToDevice, l.up));
    if (sc.radial) {
        if (lsnFusionRadialCorrectTo(plan, sc.intr + 7 * f0, l.d_depth.p, l.d_colors.p, l.d_depth2.p, l.d_colors2.p, l.stream)) return -1;
        LSN_HIP(hipEventRecord(ev_corrected, l.stream));
        if (lsn::run_count(plan, run_d, run_c, l.d_off.as<int>(), l.h_off, ev_counted, l.stream)) return -1;
    }
    LSN_HIP(hipEventSynchronize(ev_counted));
    const int nv = l.h_off[n];
    if (nv < 0 || nv > cap) {
        lsn::set_error("NativeUtils: device %d returned an impossible vertex count %d for sensors %d..%d", l.device, nv, f0, f0 + n - 1);
        return -1;
    }
    sc.nv[d].store(nv, std::memory_order_release);
    const long long base = sc.base_of(sc.nv, d);
    if (base < 0) return -1;   // another device failed: its text is the call's
    if (base + nv > 0x7FFFFFFFll) {
        lsn::set_error("NativeUtils: the merged cloud exceeds 2^31-1 vertices");
        return -1;
    }
    if (lsn::run_write(plan, run_d, run_c, static_cast<char *>(sc.host) + (size_t)base * 16, l.d_off.as<int>(), sc.with_triangles, true, l.stream)) return -1;
    if (sc.with_triangles && lsn::run_triangles_count(plan, run_d, l.d_tri_off.as<int>(), l.h_toff, l.ev_tri, l.stream)) return -1;
    if (sc.back) {
        // the corrected maps of this block go home (pageable destination: the copies keep this thread) while the write pass stores and the
        // triangle count pass, already queued, runs behind it
        LSN_HIP(hipStreamWaitEvent(l.back, ev_corrected, 0));
        LSN_HIP(hipMemcpyWithStream(sc.back_d + d_src, l.d_depth2.p, dbytes, hipMemcpyDeviceToHost, l.back));
        LSN_HIP(hipMemcpyWithStream(sc.back_c + c_src, l.d_colors2.p, cbytes, hipMemcpyDeviceToHost, l.back));
    }
    if (sc.with_triangles) {
        LSN_HIP(hipEventSynchronize(l.ev_tri));
        const int nt = l.h_toff[n];
        if (nt < 0 || nt > 2 * cap) {
            lsn::set_error("NativeUtils: device %d returned an impossible triangle count %d", l.device, nt);
            return -1;
        }
        sc.nt[d].store(nt, std::memory_order_release);
        const long long tbase = sc.base_of(sc.nt, d);
        if (tbase < 0) return -1;
        if (tbase + nt > 0x7FFFFFFFll / 3) {
            lsn::set_error("NativeUtils: the merged mesh exceeds the triangle count an int index array can hold");
            return -1;
        }
        if (nt > 0 && lsn::run_triangles_write(plan, run_d, static_cast<char *>(sc.host_tri) + (size_t)tbase * 12, (int)base, true, l.stream)) return -1;
    } else {
        sc.nt[d].store(0, std::memory_order_release);
    }
    LSN_HIP(hipStreamSynchronize(l.stream));
    LSN_HIP(hipStreamSynchronize(l.back));
    return 0;   // (no lsnFusionCheck: the frames both passes read are the lane's own buffers, which nothing else touches under the lock)
}

void shard_job(void *arg, int d)
{
    ShardedCall &sc = *static_cast<ShardedCall *>(arg);
    int rc = -1;
    try {
        lsn::clear_error();
        rc = shard_part(sc, d);
    } catch (const std::exception &e) {
        lsn::set_error("device part %d: %s", d, e.what());
    } catch (...) {
        lsn::set_error("device part %d: unknown exception", d);
    }
    if (rc) {
        sc.fail(lsn::error_buffer());
        // nobody may wait for this device's counts any longer (they see `failed`), and nothing of it may stay in flight
        drain(sc.c.shards[d]->lane);
    }
}

// Runs the D parts of a sharded call -- part 0 on the calling thread, the others on their devices' workers -- and waits for all of them.
// 0, or -1 with the first failure's text in the caller's error channel and nothing of any part left in flight.
// $LSN_HOST_SHARD_SOLO=1 (measurement aid, read once): the parts of a sharded call run ONE AFTER THE OTHER on the calling thread and the
// wall time of each is kept (lsnHostShardPartMicros).  On a box with one GPU the parts of a rehearsal share its one link; alone, a part
// shows what it costs on a link of its own -- upload, count, exchange, stores, triangles -- which is what it would have on its own device.
// (In order 0, 1, ...: every part finds the counts of the parts before it already published.)
std::atomic<long long> g_part_micros[kMaxShards];
std::atomic<int> g_part_count{0};

int run_parts(Ctx &c, ShardedCall &sc)
{
    static const bool solo = getenv("LSN_HOST_SHARD_SOLO") && atoi(getenv("LSN_HOST_SHARD_SOLO")) != 0;
    if (solo) {
        for (int d = 0; d < sc.D; d++) {
            const auto t0 = std::chrono::steady_clock::now();
            shard_job(&sc, d);
            g_part_micros[d].store(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() / 1000);
        }
        g_part_count.store(sc.D);
        (void)hipSetDevice(c.device);
        if (sc.failed.load()) {
            for (int d = 0; d < sc.D; d++) drain(c.shards[d]->lane);
            lsn::set_error("%s", sc.error);
            return -1;
        }
        return 0;
    }
    struct Join {   // the workers hold pointers into the caller's frame: nothing leaves it before they have finished
        Ctx &c;
        ShardedCall &sc;
        int started = 0;
        Join(Ctx &c_, ShardedCall &sc_) : c(c_), sc(sc_) {}
        ~Join()
        {
            // only reached with workers outstanding when something threw on the way (a thread that could not be started): the parts that did
            // start may be waiting for the counts of one that never will -- tell them before waiting for them
            if (started > 0) sc.fail("a device part could not be started");
            for (int d = 1; d <= started; d++) c.shards[d]->worker.wait();
        }
    } join(c, sc);
    for (int d = 1; d < sc.D; d++) {
        c.shards[d]->worker.submit(shard_job, &sc, d);
        join.started = d;
    }
    shard_job(&sc, 0);
    for (int d = 1; d < sc.D; d++) c.shards[d]->worker.wait();
    join.started = 0;
    (void)hipSetDevice(c.device);
    if (sc.failed.load()) {
        for (int d = 0; d < sc.D; d++) drain(c.shards[d]->lane);   // every device's part has ended; nothing may store into a returned block
        lsn::set_error("%s", sc.error);
        return -1;
    }
    return 0;
}

// depthMapAndColorSetRadialCorrection over the devices of $LSN_HOST_DEVICES: every device corrects its block of sensors and writes it back
// into the caller's arrays.  The merge lane's lock is held.
int radial_sharded(Ctx &c, Lane &ml, int n_maps, unsigned char *depth_maps, unsigned char *depth_colors, const int *widths, const int *heights,
                   const float *intr)
{
    ml.last_nv = -1;
    ml.last_plan = nullptr;
    ml.last_sharded = false;
    ShardedCall sc(c);
    sc.depth_maps = depth_maps; sc.depth_colors = depth_colors;
    sc.widths = widths; sc.heights = heights;
    sc.intr = intr;
    sc.count = n_maps;
    sc.only_radial = true;
    sc.back_d = depth_maps; sc.back_c = depth_colors;
    plan_shards(n_maps, (int)c.shards.size(), sc.first, sc.D);
    return run_parts(c, sc);
}

// The merge lane's lock is held (it serialises the calls; the shards' lanes are only ever used under it).
int fuse_host_sharded(Ctx &c, Lane &ml, const unsigned char *depth_maps, const unsigned char *depth_colors, const int *widths, const int *heights,
                      const float *intr, const float *wt, Mesh *out, const float *bounds6, int count, bool with_triangles, bool radial,
                      unsigned char *radial_back_d, unsigned char *radial_back_c)
{
    ml.last_nv = -1;
    ml.last_plan = nullptr;
    ml.last_sharded = false;
    ShardedCall sc(c);
    sc.depth_maps = depth_maps; sc.depth_colors = depth_colors;
    sc.widths = widths; sc.heights = heights;
    sc.intr = intr; sc.wt = wt; sc.bounds6 = bounds6;
    sc.count = count;
    sc.with_triangles = with_triangles;
    sc.radial = radial;
    sc.back = radial && radial_back_d && radial_back_c;
    sc.back_d = radial_back_d; sc.back_c = radial_back_c;
    plan_shards(count, (int)c.shards.size(), sc.first, sc.D);
    long long cap = 0;
    for (int i = 0; i < count; i++) cap += (long long)widths[i] * heights[i];
    if (cap > 0x7FFFFFFFll) {
        lsn::set_error("NativeUtils: a tick may not exceed 2^31-1 pixels (Mesh.nVertices is an int)");
        return -1;
    }
    struct Blocks {   // the error path, like HostCall's: by the time it runs every worker has finished and drained its lane
        Ctx &c;
        void *host = nullptr, *host_tri = nullptr;
        bool committed = false;
        explicit Blocks(Ctx &c_) : c(c_) {}
        ~Blocks()
        {
            if (committed) return;
            if (host) pinned_put(c, host);
            if (host_tri) pinned_put(c, host_tri);
        }
    } blocks(c);
    blocks.host = pinned_get(c, (size_t)cap * sizeof(VertexC4ubV3f));
    if (!blocks.host) return -1;
    if (with_triangles && !(blocks.host_tri = pinned_get(c, (size_t)cap * 2 * 12))) return -1;
    sc.host = blocks.host;
    sc.host_tri = blocks.host_tri;
    if (run_parts(c, sc)) return -1;
    long long nv = 0, nt = 0;
    for (int d = 0; d < sc.D; d++) {
        nv += sc.nv[d].load();
        nt += sc.nt[d].load();
    }
    if (nt == 0 && blocks.host_tri) {
        pinned_put(c, blocks.host_tri);
        blocks.host_tri = nullptr;
    }
    // what lsnLastMesh* would need to rebuild this mesh on one device (materialize)
    ml.last_w.assign(widths, widths + count);
    ml.last_h.assign(heights, heights + count);
    ml.last_intr.assign(intr, intr + 7 * (size_t)count);
    ml.last_wt.assign(wt, wt + 12 * (size_t)count);
    ml.last_bounds.assign(bounds6, bounds6 + 6);
    ml.last_nv = (int)nv;
    ml.last_nt = (int)nt;
    ml.last_in_hbm = false;
    ml.last_sharded = true;
    ml.last_radial = radial;
    ml.last_tri = with_triangles;
    out->nVertices = (int)nv;
    out->vertices = static_cast<VertexC4ubV3f *>(blocks.host);
    out->nTriangles = (int)nt;
    out->triangles = nt > 0 ? static_cast<int *>(blocks.host_tri) : g_no_triangles;
    blocks.committed = true;
    c.last_lane.store(&ml);
    t_last_lane = &ml;
    return 0;
}

// lsnLastMesh* after a sharded call: the frames the devices fused (corrected, if the call started with the correction) are still in their
// lanes' buffers; they are gathered onto the merge lane's device, where the whole rig's plan rebuilds the mesh.  Merge lane's lock held.
int materialize_sharded(Ctx &c, Lane &l)
{
    const int count = (int)l.last_w.size();
    int first[kMaxShards + 1], D = 0;
    plan_shards(count, (int)c.shards.size(), first, D);
    LSN_HIP(hipSetDevice(l.device));
    l.groups.clear();
    LsnFusion *plan = get_plan(c, l, l.last_w.data(), l.last_h.data(), 0, count);
    if (!plan) return -1;
    const long long cap = lsnFusionTickCapacity(plan);
    if (l.d_depth.reserve((size_t)cap * 2 + 16) || l.d_colors.reserve((size_t)cap * 3 + 16) || l.d_out.reserve((size_t)cap * 16) ||
        l.d_off.reserve(sizeof(int) * (size_t)(count + 1)) || l.d_tri_off.reserve(sizeof(int) * (size_t)(count + 1)) ||
        (l.last_tri && l.d_tri.reserve((size_t)cap * 2 * 12)))
        return -1;
    if (lsnFusionSetParams(plan, l.last_intr.data(), l.last_wt.data(), l.last_bounds.data(), l.stream)) return -1;
    size_t d_off = 0, c_off = 0;
    for (int d = 0; d < D; d++) {
        Lane &sl = c.shards[d]->lane;
        size_t px = 0;
        for (int i = first[d]; i < first[d + 1]; i++) px += (size_t)l.last_w[i] * l.last_h[i];
        const void *src_d = l.last_radial ? sl.d_depth2.p : sl.d_depth.p, *src_c = l.last_radial ? sl.d_colors2.p : sl.d_colors.p;
        LSN_HIP(hipMemcpyPeerAsync(l.d_depth.as<char>() + d_off, l.device, src_d, sl.device, px * 2, l.stream));
        LSN_HIP(hipMemcpyPeerAsync(l.d_colors.as<char>() + c_off, l.device, src_c, sl.device, px * 3, l.stream));
        d_off += px * 2;
        c_off += px * 3;
    }
    if (l.last_tri ? lsn::run_mesh(plan, l.d_depth.p, l.d_colors.p, l.d_out.p, l.d_off.as<int>(), l.d_tri.p, l.d_tri_off.as<int>(), l.stream, nullptr)
                   : lsn::run_hooked(plan, l.d_depth.p, l.d_colors.p, l.d_out.p, l.d_off.as<int>(), l.stream, nullptr))
        return -1;
    LSN_HIP(hipStreamSynchronize(l.stream));
    l.last_in_hbm = true;
    return 0;
}

// lsnLastMesh*: the mesh of the lane's last call in d_out / d_tri.  The direct path left it in host memory only; its inputs are
// still resident, so the plan's ordinary launches rebuild it in HBM (bit-identical: same arithmetic, same frames).  Lane lock held.
int materialize(Lane &l)
{
    if (l.last_nv < 0 || l.last_in_hbm) return 0;
    if (l.last_sharded) return materialize_sharded(ctx(), l);
    LsnFusion *plan = l.last_plan;
    if (!plan) return -1;
    const long long cap = lsnFusionTickCapacity(plan);
    if (l.d_out.reserve((size_t)cap * 16) || (l.last_tri && l.d_tri.reserve((size_t)cap * 2 * 12))) return -1;
    const void *run_d = l.last_radial ? l.d_depth2.p : l.d_depth.p, *run_c = l.last_radial ? l.d_colors2.p : l.d_colors.p;
    if (l.last_tri ? lsn::run_mesh(plan, run_d, run_c, l.d_out.p, l.d_off.as<int>(), l.d_tri.p, l.d_tri_off.as<int>(), l.stream, nullptr)
                   : lsn::run_hooked(plan, run_d, run_c, l.d_out.p, l.d_off.as<int>(), l.stream, nullptr))
        return -1;
    LSN_HIP(hipStreamSynchronize(l.stream));
    l.last_in_hbm = true;
    return 0;
}

// Which flow a call takes (measured on one box, 8 x 512x424, gpurun_out/r04/host_ab5.txt):
//   merge / single-sensor calls: the kernels store straight into the mesh's host blocks (0.43 / 0.75 ms against 0.55 / 0.84 ms for the
//       copy-engine form: a pageable upload and an asynchronous download do not run side by side -- the third and fourth upload
//       run of a call take 103 instead of 37 us while the previous group's vertices are on their way down);
//   calls that start with the radial correction: mesh in HBM, asynchronous copies home group by group (1.1 against 1.26 ms: the ~100 us
//       of latency-bound closing rounds per group cannot hide behind a storing kernel, but they do hide behind a DMA).
// $LSN_HOST_PATH=direct / grouped forces one of them for every call (A/B runs).
int fuse_host(Ctx &c, Lane &l, const unsigned char *depth_maps, const unsigned char *depth_colors, const int *widths, const int *heights,
              const float *intr, const float *wt, Mesh *out, const float *bounds6, int first, int count, bool with_triangles, bool radial,
              unsigned char *radial_back_d, unsigned char *radial_back_c)
{
    // a merge call on a context with several devices ($LSN_HOST_DEVICES), more than one sensor: one sensor block per device and link
    if (&l == &c.merge && c.shards.size() >= 2 && count >= 2 && first == 0)
        return fuse_host_sharded(c, l, depth_maps, depth_colors, widths, heights, intr, wt, out, bounds6, count, with_triangles, radial, radial_back_d,
                                 radial_back_c);
    HostCall h(c, l, depth_maps, depth_colors, widths, heights, intr, wt, bounds6, first, count, with_triangles, radial, radial_back_d, radial_back_c);
    const bool direct = c.host_path == 1 || (c.host_path == 0 && !radial);
    return direct ? fuse_host_direct(h, out) : fuse_host_grouped(h, out);
}

// depthMapAndColorSetRadialCorrection on the lane (one device) or over the devices of $LSN_HOST_DEVICES.  The lane's lock is held.
void radial_host(Ctx &c, Lane &l, int n_maps, unsigned char *depth_maps, unsigned char *depth_colors, const int *widths, const int *heights,
                 const float *intr_params)
{
    if (c.shards.size() >= 2 && n_maps >= 2) {
        (void)radial_sharded(c, l, n_maps, depth_maps, depth_colors, widths, heights, intr_params);
        return;
    }
    l.last_nv = -1;   // the lane's buffers are about to be reused
    l.last_plan = nullptr;
    l.groups.clear();
    // the upload schedule of a call that starts with the correction: groups of >= 2.5 MB of colours, each with a plan for its warp tables
    if (make_schedule(c, l, widths, heights, 0, n_maps, true, false)) return;
    const size_t G = l.groups.size();
    size_t dbytes = 0, cbytes = 0;
    for (const Group &q : l.groups) {
        dbytes += q.dbytes;
        cbytes += q.cbytes;
    }
    if (l.d_depth.reserve(dbytes + 16) || l.d_colors.reserve(cbytes + 16) || l.d_depth2.reserve(dbytes + 16) || l.d_colors2.reserve(cbytes + 16)) return;
    // Both directions are pageable copies that keep the thread; what overlaps is the correction itself (~100 us per group, latency
    // bound) with the next group's upload and the previous group's way home.  Out of place on the device (the warped, un-closed
    // maps stay in LDS); a group's slice of the caller's arrays is overwritten once ITS kernels have run -- a call that fails
    // later leaves the earlier groups corrected and the rest untouched.
    size_t next_group = 0, written = 0;
    auto finish = [&](size_t k) -> int {   // group k's corrected maps into the caller's arrays
        if (hipEventSynchronize(l.ev_group[k]) != hipSuccess) {
            lsn::set_error("depthMapAndColorSetRadialCorrection: the correction failed: %s", hipGetErrorString(hipGetLastError()));
            return -1;
        }
        return write_back_runs(l, k, depth_maps, depth_colors);
    };
    bool ok = true;
    for (size_t i = 0; ok && i < l.copies.size(); i++) {
        const Copy &cp = l.copies[i];
        char *dst = (cp.colours ? l.d_colors.as<char>() : l.d_depth.as<char>()) + cp.dev_off;
        const unsigned char *src = (cp.colours ? depth_colors : depth_maps) + cp.src_off;
        if (hipMemcpyWithStream(dst, src, cp.bytes, hipMemcpyHostToDevice, l.up) != hipSuccess) {
            lsn::set_error("depthMapAndColorSetRadialCorrection: upload failed: %s", hipGetErrorString(hipGetLastError()));
            ok = false;
            break;
        }
        for (; ok && next_group < G && l.groups[next_group].ready_after == (int)i + 1; next_group++) {
            const Group &q = l.groups[next_group];
            ok = lsnFusionRadialCorrectTo(q.radial_plan, intr_params + 7 * q.first, l.d_depth.as<char>() + q.d_off, l.d_colors.as<char>() + q.c_off,
                                          l.d_depth2.as<char>() + q.d_off, l.d_colors2.as<char>() + q.c_off, l.stream) == 0 &&
                 hipEventRecord(l.ev_group[next_group], l.stream) == hipSuccess;
            // the group before goes home while this one is being corrected -- but never before every upload run it shares with a later
            // group has been read (a depth run may cover sensors of the next group: write_back_runs only takes runs that END in k)
            if (ok && next_group > 0 && written < next_group) {
                ok = finish(written) == 0;
                written++;
            }
        }
    }
    for (; ok && written < G; written++) ok = finish(written) == 0;
    if (!ok) drain(l);
}

}  // namespace host
}  // namespace lsn

using namespace lsn::host;

// Host-only (no device needed): how a merge call over `n_maps` sensors is cut over `n_devices` devices of $LSN_HOST_DEVICES, as text --
// "0:[0-3] 1:[4-7]": shard:[first-last sensor], blocks contiguous and in sensor order (the order their vertices take in the Mesh) -- and,
// when first_out is given, the block bounds themselves (n_devices + 1 ints are enough).  Returns the number of shards used
// (min(n_devices, n_maps), at most 16), -1 on bad arguments.  n_devices = 0: the devices this process was configured with.
static int lsnHostShardDescribe_impl(int n_maps, int n_devices, int *first_out, char *buf, int len)
{
    lsn::clear_error();
    if (n_devices == 0) {
        std::vector<int> devs;
        const char *e = getenv("LSN_HOST_DEVICES");
        if (e && parse_device_list(e, 1 << 30, devs)) return -1;
        n_devices = devs.size() >= 2 ? (int)devs.size() : 1;
    }
    if (n_maps <= 0 || n_devices < 0) {
        lsn::set_error("lsnHostShardDescribe: bad arguments");
        return -1;
    }
    int first[kMaxShards + 1], D = 0;
    plan_shards(n_maps, n_devices, first, D);
    std::string out;
    for (int d = 0; d < D; d++) {
        char item[64];
        if (first[d + 1] - first[d] == 1) snprintf(item, sizeof(item), "%s%d:[%d]", d ? " " : "", d, first[d]);
        else snprintf(item, sizeof(item), "%s%d:[%d-%d]", d ? " " : "", d, first[d], first[d + 1] - 1);
        out += item;
    }
    if (buf && len > 0) snprintf(buf, (size_t)len, "%s", out.c_str());
    if (first_out)
        for (int d = 0; d <= D; d++) first_out[d] = first[d];
    return D;
}

extern "C" int lsnHostShardDescribe(int n_maps, int n_devices, int *first_out, char *buf, int len)
{
    return lsn::guarded<int>("lsnHostShardDescribe", static_cast<int>(-1), [&]() { return lsnHostShardDescribe_impl(n_maps, n_devices, first_out, buf, len); });
}

// Measurement aid: the wall time (microseconds) every part of the LAST sharded call took when the parts ran one after the other
// ($LSN_HOST_SHARD_SOLO=1); returns the number of parts, 0 when no such call has been made.
extern "C" int lsnHostShardPartMicros(long long *out, int n)
{
    const int D = g_part_count.load();
    for (int d = 0; out && d < D && d < n; d++) out[d] = g_part_micros[d].load();
    return D;
}

// Test hooks (tests/test_abi.py): how many fault points of a kind (0 = guarded entries, 1 = device / pinned allocations) the process has
// passed -- so that a test can aim $LSN_TEST_FAIL_ALLOC at one particular allocation of one particular call -- and what the pool of
// pinned mesh blocks holds: blocks out with callers (or leaked), blocks waiting for reuse, bytes out.
extern "C" long long lsnTestFaultPoints(int kind) { return lsn::test_fault_points(kind); }

extern "C" int lsnHostPoolStats(int *live_blocks, int *pooled_blocks, long long *live_bytes)
{
    return lsn::guarded<int>("lsnHostPoolStats", static_cast<int>(-1), [&]() {
        Ctx &c = ctx();
        std::lock_guard<std::mutex> tg(c.tab_mu);
        long long bytes = 0;
        for (const auto &kv : c.live) bytes += (long long)kv.second;
        if (live_blocks) *live_blocks = (int)c.live.size();
        if (pooled_blocks) *pooled_blocks = (int)c.pool.size();
        if (live_bytes) *live_bytes = bytes;
        return 0;
    });
}

// Host-only (no device needed): the upload schedule a call with these frames would follow, as text -- "D[0-2] C[0-2] | D[3-7] C[3-5] |
// C[6-7]": runs of the depth / colour arrays in upload order, `|` where a group of sensors becomes ready and is launched.  Exists
// so that the scheduling logic can be tested without a GPU (tests/test_abi.py).
static int lsnHostScheduleDescribe_impl(int n_maps, const int *widths, const int *heights, int first, int count, int radial, int sensors_per_group,
                                       char *buf, int len)
{
    lsn::clear_error();
    if (n_maps <= 0 || !widths || !heights || first < 0 || count <= 0 || first + count > n_maps || !buf || len <= 0) {
        lsn::set_error("lsnHostScheduleDescribe: bad arguments");
        return -1;
    }
    std::vector<Group> groups;
    std::vector<Copy> copies;
    plan_schedule(groups, copies, widths, heights, first, count, radial != 0, sensors_per_group, radial == 1);
    auto sensor_at = [&](bool colours, size_t off) {   // which sensor starts at byte `off` of the lane's (packed) buffer
        size_t at = 0;
        for (int i = first; i < first + count; i++) {
            if (at == off) return i;
            at += (size_t)widths[i] * heights[i] * (colours ? 3 : 2);
        }
        return first + count;
    };
    std::string out;
    size_t next_group = 0;
    for (size_t i = 0; i < copies.size(); i++) {
        const Copy &cp = copies[i];
        const int a = sensor_at(cp.colours, cp.dev_off), b = sensor_at(cp.colours, cp.dev_off + cp.bytes) - 1;
        char item[64];
        if (a == b) snprintf(item, sizeof(item), "%c[%d]", cp.colours ? 'C' : 'D', a);
        else snprintf(item, sizeof(item), "%c[%d-%d]", cp.colours ? 'C' : 'D', a, b);
        if (!out.empty()) out += " ";
        out += item;
        bool ready = false;
        for (; next_group < groups.size() && groups[next_group].ready_after == (int)i + 1; next_group++) ready = true;
        if (ready && i + 1 < copies.size()) out += " |";
    }
    snprintf(buf, (size_t)len, "%s", out.c_str());
    return (int)groups.size();
}

extern "C" int lsnHostScheduleDescribe(int n_maps, const int *widths, const int *heights, int first, int count, int radial, int sensors_per_group,
                                       char *buf, int len)
{
    return lsn::guarded<int>("lsnHostScheduleDescribe", static_cast<int>(-1), [&]() { return lsnHostScheduleDescribe_impl(n_maps, widths, heights, first, count, radial, sensors_per_group, buf, len); });
}

