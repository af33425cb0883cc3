// tick.hip -- the reference's tick, device resident, as ONE call: CorrectRadialDistortionsForDepthMaps then GenerateMesh on every tick
// (LiveScanServer/KinectServer.cs:518-525, :354-374) = lsnFusionRadialCorrectTo + lsnFusionRunMesh on a batch of ticks in HBM.
//
// Why it is a call of its own: the stages of the chain are bound by different things -- the radial correction's closing rounds are a
// latency chain that occupies a quarter of the wave slots for 0.23 ms, the count passes are VALU-bound, the write passes store-bound --
// and a batch cut in two halves that run side by side (the caller's stream and an internal one, two plans) lets one half's waits be
// the other half's work.  Round 4 measured that at +1.7 % and did not build it; with the band kernel regrouped into 27 KB workgroups
// (round 6: room beside the closing's two 64 KB workgroups per CU) two FREE-RUNNING streams gain +4.5 to +6.5 % on scene frames (44.1-44.7 ->
// 46.6-47.4 k ticks/s for 64 ticks x 8 x 512x424; four parts: 42.0 k; hash-noise frames, which have nothing to close: -2 %;
// tools/tick_pipelined.py).  This call forks from and joins to the caller's stream every time -- its outputs are complete on the caller's
// stream when its work there is -- which keeps +2 % of that (45.0-45.1 k against 44.1-44.3; profiles/r06_tick_pipelined.txt): the rest is
// overlap across calls.  Results are those of the two calls on one plan, byte for byte: the halves share nothing but the calibration.
#include "fusion_shared.hpp"

#include <mutex>
#include <vector>

struct LsnTick {
    int device = 0, n_ticks = 0, n_maps = 0, parts = 1;
    LsnFusion *plan[2] = {nullptr, nullptr};
    int first[3] = {0, 0, 0};            // part k covers ticks [first[k], first[k + 1])
    long long cap = 0, tri_cap = 0;      // vertices / triangles per tick
    std::vector<float> intr;             // the radial correction's intrinsics (lsnTickSetParams)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_band = nullptr;
    bool stagger = true;                 // $LSN_TICK_STAGGER=0: both halves start together
    std::mutex mu;
};

static void lsnTickDestroy_impl(LsnTick *t)
{
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->side) {
        (void)hipStreamSynchronize(t->side);
        (void)hipStreamDestroy(t->side);
    }
    if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
    if (t->ev_join) (void)hipEventDestroy(t->ev_join);
    if (t->ev_band) (void)hipEventDestroy(t->ev_band);
    for (LsnFusion *p : t->plan)
        if (p) lsnFusionDestroy(p);
    delete t;
}

extern "C" void lsnTickDestroy(LsnTick *t)
{
    lsn::guarded_void("lsnTickDestroy", [&]() { lsnTickDestroy_impl(t); });
}

static LsnTick *lsnTickCreate_impl(int device, int n_ticks, int n_maps, const int *widths, const int *heights)
{
    lsn::clear_error();
    if (n_ticks <= 0 || n_maps <= 0 || !widths || !heights) {
        lsn::set_error("lsnTickCreate: bad arguments");
        return nullptr;
    }
    LsnTick *t = new (std::nothrow) LsnTick();
    if (!t) return nullptr;
    t->device = device;
    t->n_ticks = n_ticks;
    t->n_maps = n_maps;
    // two halves from 8 ticks up ($LSN_TICK_PARTS=1: one plan, one stream -- the two calls as they are)
    int parts = n_ticks >= 8 ? 2 : 1;
    if (const char *e = getenv("LSN_TICK_PARTS")) parts = atoi(e) >= 2 && n_ticks >= 2 ? 2 : 1;
    t->parts = parts;
    t->first[0] = 0;
    t->first[1] = parts == 2 ? (n_ticks + 1) / 2 : n_ticks;
    if (const char *e = getenv("LSN_TICK_FIRST")) {   // tuning: ticks in the first half
        const int v = atoi(e);
        if (parts == 2 && v >= 1 && v < n_ticks) t->first[1] = v;
    }
    t->first[2] = n_ticks;
    bool bad = false;
    for (int k = 0; k < parts && !bad; k++) {
        t->plan[k] = lsnFusionCreate(device, t->first[k + 1] - t->first[k], n_maps, widths, heights);
        bad = !t->plan[k];
    }
    if (!bad) {
        t->cap = lsnFusionTickCapacity(t->plan[0]);
        t->tri_cap = lsnFusionTickTriangleCapacity(t->plan[0]);
        bad = hipSetDevice(device) != hipSuccess;
        if (!bad && parts == 2)
            bad = hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking) != hipSuccess ||
                  hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming) != hipSuccess ||
                  hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming) != hipSuccess ||
                  hipEventCreateWithFlags(&t->ev_band, hipEventDisableTiming) != hipSuccess;
        if (const char *e = getenv("LSN_TICK_STAGGER")) t->stagger = atoi(e) != 0;
        if (bad && !lsn::has_error()) lsn::set_error("lsnTickCreate: %s", hipGetErrorString(hipGetLastError()));
    }
    if (bad) {
        lsnTickDestroy_impl(t);
        return nullptr;
    }
    return t;
}

extern "C" LsnTick *lsnTickCreate(int device, int n_ticks, int n_maps, const int *widths, const int *heights)
{
    return lsn::guarded<LsnTick *>("lsnTickCreate", static_cast<LsnTick *>(nullptr), [&]() { return lsnTickCreate_impl(device, n_ticks, n_maps, widths, heights); });
}

extern "C" long long lsnTickCapacity(const LsnTick *t) { return t ? t->cap : 0; }
extern "C" long long lsnTickTriangleCapacity(const LsnTick *t) { return t ? t->tri_cap : 0; }
extern "C" int lsnTickParts(const LsnTick *t) { return t ? t->parts : 0; }

static int lsnTickSetParams_impl(LsnTick *t, const float *intr, const float *wt, const float *bounds6, void *stream)
{
    lsn::clear_error();
    if (!t || !intr || !wt || !bounds6) {
        lsn::set_error("lsnTickSetParams: null argument");
        return -1;
    }
    std::lock_guard<std::mutex> g(t->mu);
    t->intr.assign(intr, intr + 7 * (size_t)t->n_maps);
    for (int k = 0; k < t->parts; k++)
        if (lsnFusionSetParams(t->plan[k], intr, wt, bounds6, stream)) return -1;
    return 0;
}

extern "C" int lsnTickSetParams(LsnTick *t, const float *intr, const float *wt, const float *bounds6, void *stream)
{
    return lsn::guarded<int>("lsnTickSetParams", static_cast<int>(-1), [&]() { return lsnTickSetParams_impl(t, intr, wt, bounds6, stream); });
}

static int lsnTickRun_impl(LsnTick *t, const void *d_depth_in, const void *d_colors_in, void *d_depth_corr, void *d_colors_corr, void *d_vertices,
                           int *d_offsets, void *d_triangles, int *d_tri_offsets, void *stream)
{
    lsn::clear_error();
    if (!t || !d_depth_in || !d_colors_in || !d_depth_corr || !d_colors_corr || !d_vertices || !d_offsets || !d_triangles || !d_tri_offsets) {
        lsn::set_error("lsnTickRun: null argument");
        return -1;
    }
    if (t->intr.empty()) {
        lsn::set_error("lsnTickRun: lsnTickSetParams has not been called");
        return -1;
    }
    std::lock_guard<std::mutex> g(t->mu);
    LSN_HIP(hipSetDevice(t->device));
    hipStream_t s = lsn::as_stream(stream);
    // part k's slices of the caller's arrays: [tick][pixels], [tick][pixels][3], [tick][capacity] vertices, [tick][n_maps + 1], ...
    auto part = [&](int k, hipStream_t st) -> int {
        const size_t t0 = (size_t)t->first[k];
        const size_t px = (size_t)t->cap, nm = (size_t)t->n_maps + 1;
        const unsigned char *din = static_cast<const unsigned char *>(d_depth_in) + 2 * px * t0, *cin = static_cast<const unsigned char *>(d_colors_in) + 3 * px * t0;
        unsigned char *dco = static_cast<unsigned char *>(d_depth_corr) + 2 * px * t0, *cco = static_cast<unsigned char *>(d_colors_corr) + 3 * px * t0;
        if (lsnFusionRadialCorrectTo(t->plan[k], t->intr.data(), din, cin, dco, cco, st)) return -1;
        return lsnFusionRunMesh(t->plan[k], dco, cco, static_cast<unsigned char *>(d_vertices) + 16 * px * t0, d_offsets + nm * t0,
                                static_cast<unsigned char *>(d_triangles) + 12 * (size_t)t->tri_cap * t0, d_tri_offsets + nm * t0, st);
    };
    if (t->parts == 1) return part(0, s);
    // fork: the side stream starts where the caller's stream stands -- and, staggered, only when the first half's band kernel is through: two
    // halves that start together march in step (band beside band, closing beside closing) and gain nothing; half a stage apart, one half's
    // closing rounds run beside the other half's band kernel, then beside its count / write / triangle passes.  join: the caller's stream
    // continues behind both halves.
    LSN_HIP(hipEventRecord(t->ev_fork, s));
    LSN_HIP(hipStreamWaitEvent(t->side, t->ev_fork, 0));
    t->plan[0]->after_band = t->stagger ? t->ev_band : nullptr;
    const int rc_a = part(0, s);
    t->plan[0]->after_band = nullptr;
    char err_a[lsn::kErrorLen];
    snprintf(err_a, sizeof(err_a), "%s", lsn::error_buffer());   // (every export clears the channel on entry: the second half's calls would wipe the first half's text)
    if (t->stagger) (void)hipStreamWaitEvent(t->side, t->ev_band, 0);   // (a closing route without a band kernel records nothing new: no wait, the halves start together)
    const int rc_b = part(1, t->side);
    if (rc_a) lsn::set_error("%s", err_a);
    // (the join is enqueued whatever happened: nothing of a failed half may still be running unobserved when the caller's stream goes on)
    if (hipEventRecord(t->ev_join, t->side) == hipSuccess) (void)hipStreamWaitEvent(s, t->ev_join, 0);
    else (void)hipGetLastError();
    return rc_a || rc_b ? -1 : 0;
}

extern "C" int lsnTickRun(LsnTick *t, const void *d_depth_in, const void *d_colors_in, void *d_depth_corr, void *d_colors_corr, void *d_vertices,
                          int *d_offsets, void *d_triangles, int *d_tri_offsets, void *stream)
{
    return lsn::guarded<int>("lsnTickRun", static_cast<int>(-1), [&]() {
        return lsnTickRun_impl(t, d_depth_in, d_colors_in, d_depth_corr, d_colors_corr, d_vertices, d_offsets, d_triangles, d_tri_offsets, stream);
    });
}
