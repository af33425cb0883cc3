// host_ctx.hpp -- the process-wide context behind the reference's exports on HOST arrays: lanes (streams, device buffers, plan tables), the
// pool of pinned mesh blocks, the devices of $LSN_HOST_DEVICES with their worker threads -- and the entry points of the three call flows
// that run on it (host_flows.hip; the design notes are at the top of that file).  abi.hip holds the exports themselves.
#pragma once

#include "lsn_common.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <mutex>
#include <new>
#include <thread>
#include <unordered_map>
#include <vector>

namespace lsn {
namespace host {

// One group of consecutive sensors of a call: fused by one launch as soon as its frames are on the device (file comment).
struct Group {
    int first = 0, count = 0;        // sensors [first, first + count) of the caller's arrays
    size_t d_off = 0, c_off = 0;     // where the group's frames start in the lane's device buffers (bytes)
    size_t d_src = 0, c_src = 0;     // ... and in the caller's arrays
    size_t dbytes = 0, cbytes = 0;
    int ready_after = 0;             // how many copies of the call's upload schedule must have landed before its launch
    LsnFusion *radial_plan = nullptr;   // calls that start with the radial correction: the group's own plan for it (its warp tables)
};

// One blocking upload of the schedule: a run of whole frames of the caller's depth or colour array.
struct Copy {
    size_t dev_off = 0, src_off = 0, bytes = 0;
    bool colours = false;
};

constexpr int kMaxGroups = 16;

// What one call in flight needs: streams, events, device buffers, plans.  LiveScanServer runs its merge calls (updateWorker: radial
// correction, generateMeshFromDepthMaps) and its refine calls (refineWorker: generateVerticesFromDepthMap per sensor, then ICP) on two
// threads (MainWindowForm.cs:238,304); each of the three families has its own lane, so they only meet at the pool of pinned blocks.
struct Lane {
    std::mutex mu;
    int device = 0;           // the device the lane's streams, buffers and plans live on
    hipStream_t stream = nullptr, up = nullptr, down = nullptr, back = nullptr;   // kernels; uploads; mesh downloads; write-backs of corrected maps
    hipEvent_t ev_group[kMaxGroups] = {};    // "group g's vertices are in HBM (and its corrected maps final)"
    hipEvent_t ev_tri = nullptr;             // "the triangle counts are known"
    int *h_off = nullptr, *h_toff = nullptr;   // pinned: the offset tables of the call in progress
    int h_off_cap = 0;
    lsn::DevBuf d_depth, d_colors, d_depth2, d_colors2, d_out, d_off, d_tri, d_tri_off;
    // the lane's plans: key = n sensors, first sensor, widths..., heights...  Owned by the lane and only touched under its lock, so a
    // plan never runs on two lanes' streams at once and an eviction cannot pull a plan from under the other lane's call
    std::map<std::vector<int>, LsnFusion *> plans;
    // The mesh of the lane's last call (lsnLastMesh* read it): its vertex / triangle counts, and whether it is in d_out / d_tri.
    // The direct path stores the mesh to host memory only; the inputs stay in d_depth / d_colors (d_depth2 / d_colors2 after the
    // radial correction), so the mesh can be rebuilt in HBM by last_plan when somebody asks for it.
    int last_nv = -1, last_nt = 0;
    bool last_in_hbm = false, last_radial = false, last_tri = false;
    LsnFusion *last_plan = nullptr;
    // ... or it was a call sharded over the devices of $LSN_HOST_DEVICES: the mesh only exists in host memory and its inputs are spread
    // over the shards' lanes; what materialize() needs to rebuild it here
    bool last_sharded = false;
    std::vector<int> last_w, last_h;
    std::vector<float> last_intr, last_wt, last_bounds;
    std::vector<Group> groups;
    std::vector<Copy> copies;
};

// A thread that runs one job at a time for the thread that hands it over.  A pageable upload keeps the thread that issues it until the
// bytes are on the device (file comment), so D links are only busy at once when D threads issue the copies: one worker per device of
// $LSN_HOST_DEVICES beyond the first (the calling thread serves the first).  Started on first use, never joined (the context is never
// destroyed): an idle worker sits in its condition variable and touches nothing.
struct Worker {
    std::mutex mu;
    std::condition_variable cv;
    void (*fn)(void *, int) = nullptr;
    void *arg = nullptr;
    int index = 0;
    bool busy = false, started = false;
    void loop()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return fn != nullptr; });
            void (*f)(void *, int) = fn;
            void *a = arg;
            const int i = index;
            fn = nullptr;
            lk.unlock();
            f(a, i);   // never throws: the job catches everything itself
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
    void submit(void (*f)(void *, int), void *a, int i)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (!started) {
            std::thread(&Worker::loop, this).detach();
            started = true;
        }
        fn = f;
        arg = a;
        index = i;
        busy = true;
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !busy; });
    }
};

// One device of $LSN_HOST_DEVICES: its lane (streams, buffers, plans -- only used under the merge lane's lock) and its worker.
struct HostShard {
    Lane lane;
    Worker worker;
};

constexpr int kMaxShards = 16;

struct Ctx {
    Lane merge, single;       // lane of the merge / radial / last-mesh calls; lane of the single-sensor calls
    std::mutex icp_mu;        // ICP: own buffers, own stream
    std::mutex init_mu;
    std::mutex tab_mu;        // the pool of pinned blocks
    std::atomic<Lane *> last_lane{nullptr};   // the lane whose call finished last: lsnLastMesh* read the mesh it left in HBM
    bool ready = false;
    int device = 0;
    std::vector<HostShard *> shards;   // $LSN_HOST_DEVICES=a,b,...: merge calls are sharded over these devices (>= 2 entries; an entry may repeat)
    int host_path = 0;        // $LSN_HOST_PATH: 0 = by call (default), 1 = "direct" (kernel stores) always, 2 = "grouped" (copy engine) always
    int group_override = 0;   // $LSN_HOST_GROUP: sensors per group (0 = by size)
    hipStream_t icp_stream = nullptr;
    lsn::DevBuf d_v1, d_v2, d_Rt;
    LsnIcp *icp = nullptr;
    int icp_n1 = 0, icp_n2 = 0;
    // pinned host blocks handed out as Mesh::vertices, recycled by deleteMesh
    std::unordered_map<void *, size_t> live;          // ptr -> capacity (bytes)
    std::multimap<size_t, void *> pool;               // capacity -> ptr
    std::mutex wire_mu;       // the packer and its output buffer (lsnLastMesh*)
    LsnTransfer *xfer = nullptr;
    int xfer_v = 0, xfer_t = 0;
    lsn::DevBuf d_wire;
    bool warned_flags = false;
};

// the lane of the calling thread's own last mesh call (lsnLastMesh* read that lane's mesh; include/NativeUtils.h)
extern thread_local Lane *t_last_lane;

Ctx &ctx();
int ensure_ready(Ctx &c);                  // takes c.init_mu itself; callers may hold a lane's lock or c.icp_mu
void drain(Lane &l);                       // waits for everything the lane has in flight
void *pinned_get(Ctx &c, size_t bytes);    // a pinned block of the pool (handed out as Mesh::vertices / Mesh::triangles; ICP's scratch)
void pinned_put(Ctx &c, void *p);          // ... back (a pointer that is not ours is left alone)
extern int g_no_triangles[1];              // what Mesh::triangles points at when a mesh has no triangles
void empty_mesh(Mesh *m) noexcept;

// The upload schedule of a call (pure host logic: lsnHostScheduleDescribe) and the split of a call over devices (lsnHostShardDescribe).
void plan_schedule(std::vector<Group> &groups, std::vector<Copy> &copies, const int *widths, const int *heights, int first, int count, bool radial,
                   int group_override, bool small_first);
void plan_shards(int count, int n_devices, int *first, int &D);
int parse_device_list(const char *text, int n_visible, std::vector<int> &out);

// The calls.  The lane's lock is held by the caller; `out` is left untouched on failure (the export then returns an empty mesh).
int fuse_host(Ctx &c, Lane &l, const unsigned char *depth_maps, const unsigned char *depth_colors, const int *widths, const int *heights,
              const float *intr, const float *wt, Mesh *out, const float *bounds6, int first, int count, bool with_triangles, bool radial = false,
              unsigned char *radial_back_d = nullptr, unsigned char *radial_back_c = nullptr);
void radial_host(Ctx &c, Lane &l, int n_maps, unsigned char *depth_maps, unsigned char *depth_colors, const int *widths, const int *heights,
                 const float *intr_params);
int materialize(Lane &l);                  // lsnLastMesh*: the mesh of the lane's last call in d_out / d_tri

}  // namespace host
}  // namespace lsn
