// lsn_common.hpp -- shared host-side plumbing of libNativeUtils.so (error channel, HIP checks, small RAII).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <string>

#include "../../include/NativeUtils.h"

namespace lsn {

// Thread-local error text behind lsnGetLastError().  The reference's exports have no error channel at all
// (void / constant returns, src/NativeUtils/depthprocessing.cpp:1631,1715; icp.cpp:176); nothing may throw
// across the C-ABI, so failures end up here.
constexpr int kErrorLen = 1024;
char *error_buffer() noexcept;                         // fixed thread-local storage: setting an error never allocates
void set_error(const char *fmt, ...) noexcept;
inline void clear_error() noexcept { error_buffer()[0] = 0; }
inline bool has_error() noexcept { return error_buffer()[0] != 0; }
long test_fault_points(int kind);                      // how many fault points of that kind the process has passed
void test_fault_point(int kind);                       // 0 = guarded entry, 1 = allocation; throws std::bad_alloc when a test asks for it

// Every extern "C" entry point runs its body through this: no exception may cross the C-ABI into a P/Invoke frame
// (SURVEY 8b "exceptions must not escape"; the reference itself lets nanoflann throw, include/nanoflann.h:904).
template <class R, class F>
inline R guarded(const char *name, R fail, F &&body) noexcept
{
    try {
        test_fault_point(0);
        return body();
    } catch (const std::exception &e) {
        set_error("%s: %s", name, e.what());
    } catch (...) {
        set_error("%s: unknown exception", name);
    }
    return fail;
}
template <class F>
inline void guarded_void(const char *name, F &&body) noexcept
{
    try {
        test_fault_point(0);
        body();
        return;
    } catch (const std::exception &e) {
        set_error("%s: %s", name, e.what());
    } catch (...) {
        set_error("%s: unknown exception", name);
    }
}

#define LSN_HIP(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            lsn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);   \
            return -1;                                                                                   \
        }                                                                                                \
    } while (0)

#define LSN_HIP_NULL(expr)                                                                               \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            lsn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);   \
            return nullptr;                                                                              \
        }                                                                                                \
    } while (0)

// Device buffer that frees itself; not copyable.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    // grow-only
    int reserve(size_t n) {
        if (n <= bytes) return 0;
        test_fault_point(1);
        release();
        LSN_HIP(hipMalloc(&p, n));
        bytes = n;
        return 0;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

inline hipStream_t as_stream(void *s) { return static_cast<hipStream_t>(s); }

// Optional extras of one run, for hosts that overlap copies with the passes (host_flows.hip): all members may be null.
struct RunHooks {
    hipEvent_t colours_ready = nullptr;   // the write pass waits for it (the count pass only reads depth)
    int *h_offsets = nullptr;             // pinned host copy of the offset table, issued right after the scan ...
    hipEvent_t counted = nullptr;         // ... and this event recorded behind it (the vertex count is known before the write pass ends)
    hipEvent_t written = nullptr;         // recorded after the write pass (the vertices may leave while the triangulation runs)
    int *h_tri_offsets = nullptr;         // lsnFusionRunMesh: pinned host copy of the triangle offset table ...
    hipEvent_t tri_counted = nullptr;     // ... and the event behind it
    bool mirror = false;                  // h_tri_offsets is pinned, device-visible memory and the scan kernel stores the table there itself: no copy
    bool host_out = false;                // the triangle write pass's output is pinned host memory (its HOST form)
};
// lsnFusionRun with the plan's mutex already held; with_pixmap also fills the pixel -> vertex map the triangulation reads.
int run_locked(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets, hipStream_t s, bool with_pixmap,
               const RunHooks *hooks);
// lsnFusionRun with hooks (takes the mutex).
int run_hooked(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets, hipStream_t s, const RunHooks *hooks);
// lsnFusionRunMesh with hooks (takes the mutex once for the vertex and the triangle passes).
int run_mesh(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets, void *d_triangles, int *d_tri_offsets,
             hipStream_t s, const RunHooks *hooks);

// One launch, single pass, over frames [f0, f1) of a one-tick plan (fusion.hip); and the triangle passes alone over the whole tick, for
// a pixel -> vertex map that run_frames(with_pixmap) launches have filled (mesh.hip).  tri_mirror: optional pinned copy of the table;
// tri_counted: optional event recorded behind the scan (the counts are in the mirror), before the triangle write pass.
int run_frames(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets, int f0, int f1, bool first_of_tick,
               bool with_pixmap, int *offsets_mirror, int *group_end_mirror, bool host_out, hipStream_t s);
int run_triangles(LsnFusion *p, const void *d_depth, void *d_triangles, int *d_tri_offsets, int *tri_mirror, bool host_out, hipStream_t s,
                  hipEvent_t tri_counted = nullptr);

// The two halves of the two-pass form of a one-tick plan (fusion.hip) and of its triangle passes (mesh.hip), for a caller that places
// the tick's vertices / triangles behind somebody else's: host_flows.hip's calls sharded over devices.  index_base: added to every vertex index a
// triangle names (formMesh's rebase, depthprocessing.cpp:1614-1626, across devices).
int run_count(LsnFusion *p, const void *d_depth, const void *d_colors, int *d_offsets, int *offsets_mirror, hipEvent_t counted, hipStream_t s);
int run_write(LsnFusion *p, const void *d_depth, const void *d_colors, void *vertices, int *d_offsets, bool with_pixmap, bool host_out, hipStream_t s);
int run_triangles_count(LsnFusion *p, const void *d_depth, int *d_tri_offsets, int *tri_mirror, hipEvent_t tri_counted, hipStream_t s);
int run_triangles_write(LsnFusion *p, const void *d_depth, void *d_triangles, int index_base, bool host_out, hipStream_t s);

// The survivor exchange's two ends with the back-to-back stream layout (exchange.hip; see their definitions).
int pack_survivors(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_mask, void *d_depth_c, void *d_rgb_c, int *d_tile_prefix,
                   int *d_offsets, int *d_tick_base, void *stream);
int reconstruct(LsnFusion *all, int n_shards, int maps_per_shard, const void *d_masks, const void *d_depth_c, const void *d_rgb_c, long long slab,
                const int *d_tile_prefix, const int *d_shard_offsets, void *d_merged, int *d_merged_offsets, int *d_tick_base, void *stream,
                int tick0 = 0, int n_chunk_ticks = 0, bool fill_tick_base = true);

// lsnMergeShards, optionally for shards laid out [n_ticks][n_shards][shard_cap] (per-tick all-gathers).
int merge_shards(int device, int n_shards, int n_ticks, int maps_per_shard, const void *d_shards, long long shard_cap, const int *d_shard_offsets,
                 void *d_merged, long long merged_cap, int *d_merged_offsets, bool tick_major, void *stream);

}  // namespace lsn
