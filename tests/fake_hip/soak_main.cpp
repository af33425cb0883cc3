// soak_main.cpp -- drives the exports of libNativeUtils' HOST side, built against tests/fake_hip (no GPU) under a sanitizer.
//   1. every export with NULL pointers and zero sizes;
//   2. the call mix of LiveScanServer from four threads at once (MainWindowForm.cs:238,304: updateWorker's merge calls, refineWorker's
//      single-sensor calls + ICP, plus the radial export and the last-mesh formats), every result checked against what the runtime
//      double's kernels "compute" (every non-zero depth pixel survives, one triangle per vertex);
//   3. the pool of pinned mesh blocks must be empty at the end.
// $LSN_HOST_DEVICES (read by the library) switches the merge calls to the sharded flow; $LSN_TEST_FAIL_ALLOC / $LSN_TEST_THROW make
// single calls fail, which must leave empty meshes and still an empty pool.  Exit code 0 = all checks held (the sanitizer adds its own).
#include "../../include/NativeUtils.h"

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

std::atomic<int> g_bad{0};
const bool g_faults = getenv("LSN_TEST_FAIL_ALLOC") || getenv("LSN_TEST_THROW");

#define CHECK(cond, ...)                          \
    do {                                          \
        if (!(cond)) {                            \
            fprintf(stderr, "CHECK failed: %s: ", #cond); \
            fprintf(stderr, __VA_ARGS__);         \
            fprintf(stderr, "\n");                \
            g_bad++;                              \
        }                                         \
    } while (0)

struct Rig {
    int n, w, h;
    std::vector<int> widths, heights;
    std::vector<unsigned char> depth, colours;   // the caller's packed arrays (KinectServer.cs:453-498)
    std::vector<float> intr, wt;
    std::vector<int> valid;                       // non-zero depth pixels per sensor
    float b[6] = {-1.5f, -1.0f, -1.5f, 1.5f, 1.5f, 1.5f};
    Rig(int n_, int w_, int h_, unsigned seed) : n(n_), w(w_), h(h_), widths(n_, w_), heights(n_, h_), depth((size_t)n_ * w_ * h_ * 2), colours((size_t)n_ * w_ * h_ * 3),
                                                 intr(7 * n_), wt(12 * n_), valid(n_, 0)
    {
        uint32_t x = seed * 2654435761u + 1;
        unsigned short *d = reinterpret_cast<unsigned short *>(depth.data());
        for (int s = 0; s < n; s++)
            for (int i = 0; i < w * h; i++) {
                x = x * 1664525u + 1013904223u;
                const unsigned short v = (x >> 28) == 0 ? 0 : (unsigned short)(500 + ((x >> 8) % 4000));
                d[(size_t)s * w * h + i] = v;
                valid[s] += v != 0;
            }
        for (size_t i = 0; i < colours.size(); i++) colours[i] = (unsigned char)(i * 7);
        for (int s = 0; s < n; s++) {
            const float in[7] = {(w - 1) * 0.5f, (h - 1) * 0.5f, 365.0f, 365.0f, 0.09f, -0.27f, 0.09f};
            const float tr[12] = {0, 0, -2, 1, 0, 0, 0, 1, 0, 0, 0, 1};
            memcpy(&intr[7 * s], in, sizeof(in));
            memcpy(&wt[12 * s], tr, sizeof(tr));
        }
    }
    // sensors of different sizes; the packed arrays are EXACTLY as long as the sensors need, so a read or a write-back that runs over
    // the end of the caller's arrays (a copy run of the upload schedule cut wrongly, say) is a sanitizer report
    Rig(const std::vector<int> &ws, const std::vector<int> &hs, unsigned seed) : n((int)ws.size()), w(0), h(0), widths(ws), heights(hs), intr(7 * ws.size()),
                                                                                   wt(12 * ws.size()), valid(ws.size(), 0)
    {
        size_t px = 0;
        for (int s = 0; s < n; s++) px += (size_t)widths[s] * heights[s];
        depth.resize(px * 2);
        colours.resize(px * 3);
        uint32_t x = seed * 2654435761u + 1;
        unsigned short *d = reinterpret_cast<unsigned short *>(depth.data());
        size_t at = 0;
        for (int s = 0; s < n; s++) {
            for (int i = 0; i < widths[s] * heights[s]; i++) {
                x = x * 1664525u + 1013904223u;
                const unsigned short v = (x >> 28) == 0 ? 0 : (unsigned short)(500 + ((x >> 8) % 4000));
                d[at++] = v;
                valid[s] += v != 0;
            }
            const float in[7] = {(widths[s] - 1) * 0.5f, (heights[s] - 1) * 0.5f, 365.0f, 365.0f, 0.09f, -0.27f, 0.09f};
            const float tr[12] = {0, 0, -2, 1, 0, 0, 0, 1, 0, 0, 0, 1};
            memcpy(&intr[7 * s], in, sizeof(in));
            memcpy(&wt[12 * s], tr, sizeof(tr));
        }
        for (size_t i = 0; i < colours.size(); i++) colours[i] = (unsigned char)(i * 7);
    }
    int total() const
    {
        int t = 0;
        for (int v : valid) t += v;
        return t;
    }
};

// deleteMesh as a caller that checks would use it: the call nulls the pointers it has released, so pointers that are still there mean the call
// itself did not run -- which only the fault hooks can make happen ($LSN_TEST_THROW hitting deleteMesh's own entry) -- and it is made again.
void release(Mesh &m)
{
    deleteMesh(&m);
    if (m.vertices) deleteMesh(&m);
}

bool failed_call(const Mesh &m)   // a call that a fault hook hit: empty mesh + a message
{
    char msg[256];
    return g_faults && m.nVertices == 0 && lsnGetLastError(msg, sizeof(msg)) > 0;
}

void check_mesh(const Mesh &m, int want_v, bool triangles, const char *what)
{
    if (failed_call(m)) return;
    CHECK(m.nVertices == want_v, "%s: %d vertices, expected %d", what, m.nVertices, want_v);
    CHECK(m.nTriangles == (triangles ? want_v : 0), "%s: %d triangles, expected %d", what, m.nTriangles, triangles ? want_v : 0);
    CHECK(m.triangles != nullptr, "%s: null triangle pointer", what);
    // every byte the call promises is readable (ASan) and carries what the double's kernels wrote: A = 255, indices inside the cloud
    long long bad = 0;
    for (int i = 0; i < m.nVertices; i++) bad += m.vertices[i].A != 255;
    for (int i = 0; i < 3 * m.nTriangles; i++) bad += m.triangles[i] < 0 || m.triangles[i] >= m.nVertices;
    CHECK(bad == 0, "%s: %lld bad vertices / indices", what, bad);
}

void merge_thread(int iters)
{
    Rig rig(8, 512, 424, 1), small(3, 250, 121, 2);
    for (int it = 0; it < iters; it++) {
        Mesh m;
        memset(&m, 0, sizeof(m));
        generateMeshFromDepthMaps(rig.n, rig.depth.data(), rig.colours.data(), rig.widths.data(), rig.heights.data(), rig.intr.data(), rig.wt.data(), &m, false,
                                  rig.b[0], rig.b[1], rig.b[2], rig.b[3], rig.b[4], rig.b[5], false);
        check_mesh(m, rig.total(), true, "merge 8x512x424");
        release(m);
        generateMeshFromDepthMaps(small.n, small.depth.data(), small.colours.data(), small.widths.data(), small.heights.data(), small.intr.data(),
                                  small.wt.data(), &m, false, small.b[0], small.b[1], small.b[2], small.b[3], small.b[4], small.b[5], false);
        check_mesh(m, small.total(), true, "merge 3x250x121");
        const long long bound = lsnLastMeshTransferFrame(nullptr, 0);   // whichever lane finished last: the TransferServer stream of its mesh
        if (bound > 0) {
            std::vector<unsigned char> frame((size_t)bound);
            (void)lsnLastMeshTransferFrame(frame.data(), bound);
        }
        release(m);
        // the tick as one call: the radial kernels are not emulated, the corrected maps read as zeros -> an empty cloud, through every copy
        // and event of the flow; the caller's arrays are overwritten with the (zero) corrected maps, so they are copies
        std::vector<unsigned char> d2 = rig.depth, c2 = rig.colours;
        lsnCorrectAndGenerateMesh(rig.n, d2.data(), c2.data(), rig.widths.data(), rig.heights.data(), rig.intr.data(), rig.wt.data(), &m, rig.b[0], rig.b[1],
                                  rig.b[2], rig.b[3], rig.b[4], rig.b[5], it & 1);
        check_mesh(m, 0, false, "tick as one call");
        release(m);
    }
}

void single_thread(int iters)
{
    Rig rig(8, 512, 424, 3);
    for (int it = 0; it < iters; it++)
        for (int s = 0; s < rig.n; s++) {
            Mesh m;
            memset(&m, 0, sizeof(m));
            generateVerticesFromDepthMap(rig.depth.data(), rig.colours.data(), rig.widths.data(), rig.heights.data(), rig.intr.data(), rig.wt.data(), &m, rig.b[0],
                                         rig.b[1], rig.b[2], rig.b[3], rig.b[4], rig.b[5], s);
            check_mesh(m, rig.valid[s], false, "single sensor");
            release(m);
        }
}

void radial_thread(int iters)
{
    Rig rig(8, 512, 424, 4);
    for (int it = 0; it < iters; it++) {
        std::vector<unsigned char> d2 = rig.depth, c2 = rig.colours;
        depthMapAndColorSetRadialCorrection(rig.n, d2.data(), c2.data(), rig.widths.data(), rig.heights.data(), rig.intr.data());
    }
}

void icp_thread(int iters)
{
    const int n1 = 5000, n2 = 3000;
    std::vector<Point3f> a(n1), b(n2);
    for (int i = 0; i < n1; i++) a[i] = {0.001f * i, 0.002f * (i % 97), 0.5f};
    for (int i = 0; i < n2; i++) b[i] = {0.0015f * i, 0.002f * (i % 89), 0.51f};
    for (int it = 0; it < iters; it++) {
        float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
        const float r = ICP(a.data(), b.data(), n1, n2, R, t, 3);
        CHECK(r == 1.0f, "ICP returned %f", r);
    }
}

void null_sweep()
{
    Mesh m;
    memset(&m, 0, sizeof(m));
    generateMeshFromDepthMaps(0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &m, false, 0, 0, 0, 0, 0, 0, false);
    generateMeshFromDepthMaps(4, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &m, false, 0, 0, 0, 0, 0, 0, false);
    generateMeshFromDepthMaps(4, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false, 0, 0, 0, 0, 0, 0, false);
    generateVerticesFromDepthMap(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &m, 0, 0, 0, 0, 0, 0, 0);
    generateVerticesFromDepthMap(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, -1);
    depthMapAndColorSetRadialCorrection(0, nullptr, nullptr, nullptr, nullptr, nullptr);
    depthMapAndColorSetRadialCorrection(2, nullptr, nullptr, nullptr, nullptr, nullptr);
    lsnCorrectAndGenerateMesh(0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &m, 0, 0, 0, 0, 0, 0, 1);
    lsnCorrectAndGenerateMesh(3, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 1);
    CHECK(m.nVertices == 0 && m.nTriangles == 0, "a refused call left a mesh behind");
    deleteMesh(nullptr);
    deleteMesh(&m);
    deleteMesh(&m);
    float R[9] = {0}, t[3] = {0};
    CHECK(ICP(nullptr, nullptr, 0, 0, R, t, 10) == 1.0f, "ICP(null)");
    CHECK(ICP(nullptr, nullptr, 5, 5, nullptr, nullptr, 0) == 1.0f, "ICP(null, 5)");
    (void)lsnLastMeshTransferFrame(nullptr, 0);
    (void)lsnLastMeshPly(nullptr, 0);
    char buf[64];
    CHECK(lsnHostScheduleDescribe(0, nullptr, nullptr, 0, 0, 0, 0, buf, sizeof(buf)) == -1, "schedule(null)");
    CHECK(lsnHostShardDescribe(0, 0, nullptr, buf, sizeof(buf)) == -1, "shards(0)");
    CHECK(lsnFusionCreate(0, 0, 0, nullptr, nullptr) == nullptr, "lsnFusionCreate(0)");
    lsnFusionDestroy(nullptr);
    CHECK(lsnFusionRun(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == -1, "lsnFusionRun(null)");
    CHECK(lsnFusionRunMesh(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == -1, "lsnFusionRunMesh(null)");
    Mesh *heap = createMesh();
    CHECK(heap && heap->nVertices == 0 && heap->vertices == nullptr, "createMesh");
    deleteMesh(heap);
    free(heap);
}

}  // namespace

// Random rigs of 1-8 sensors of different sizes (1 x 1 up to ~0.9 MB of colours each: below and above the size at which the upload
// schedule cuts a call into groups and merges short copy runs) through every host export, one after the other on this thread.
void ragged_rigs(int n_rigs)
{
    uint32_t x = 12345;
    auto rnd = [&](int lo, int hi) {
        x = x * 1664525u + 1013904223u;
        return lo + (int)((x >> 8) % (uint32_t)(hi - lo + 1));
    };
    for (int r = 0; r < n_rigs; r++) {
        const int n = rnd(1, 8);
        std::vector<int> ws, hs;
        for (int s = 0; s < n; s++) {
            const int kind = rnd(0, 3);
            ws.push_back(kind == 0 ? rnd(1, 12) : kind == 1 ? 8 * rnd(1, 80) + rnd(-1, 1) : rnd(200, 640));
            hs.push_back(kind == 0 ? rnd(1, 9) : kind == 1 ? rnd(1, 60) : rnd(100, 480));
        }
        Rig rig(ws, hs, 100 + r);
        char what[96];
        snprintf(what, sizeof(what), "ragged rig %d (%d sensors)", r, n);
        Mesh m;
        generateMeshFromDepthMaps(rig.n, rig.depth.data(), rig.colours.data(), rig.widths.data(), rig.heights.data(), rig.intr.data(), rig.wt.data(), &m, false,
                                  rig.b[0], rig.b[1], rig.b[2], rig.b[3], rig.b[4], rig.b[5], false);
        check_mesh(m, rig.total(), true, what);
        release(m);
        const int one = rnd(0, n - 1);
        generateVerticesFromDepthMap(rig.depth.data(), rig.colours.data(), rig.widths.data(), rig.heights.data(), rig.intr.data(), rig.wt.data(), &m, rig.b[0],
                                     rig.b[1], rig.b[2], rig.b[3], rig.b[4], rig.b[5], one);
        check_mesh(m, rig.valid[one], false, what);
        release(m);
        {
            std::vector<unsigned char> d2 = rig.depth, c2 = rig.colours;   // exact-size copies: the export writes the corrected maps back into them
            depthMapAndColorSetRadialCorrection(rig.n, d2.data(), c2.data(), rig.widths.data(), rig.heights.data(), rig.intr.data());
        }
        for (int back = 0; back < 2; back++) {
            std::vector<unsigned char> d2 = rig.depth, c2 = rig.colours;
            lsnCorrectAndGenerateMesh(rig.n, d2.data(), c2.data(), rig.widths.data(), rig.heights.data(), rig.intr.data(), rig.wt.data(), &m, rig.b[0], rig.b[1],
                                      rig.b[2], rig.b[3], rig.b[4], rig.b[5], back);
            if (!failed_call(m)) CHECK(m.nVertices >= 0 && m.triangles != nullptr, "%s: tick as one call", what);
            release(m);
        }
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4;
    null_sweep();
    ragged_rigs(3 * iters);
    {
        std::vector<std::thread> th;
        th.emplace_back(merge_thread, iters);
        th.emplace_back(single_thread, iters);
        th.emplace_back(radial_thread, iters);
        th.emplace_back(icp_thread, iters);
        for (auto &t : th) t.join();
    }
    int live = -1, pooled = -1;
    long long bytes = -1;
    CHECK(lsnHostPoolStats(&live, &pooled, &bytes) == 0, "lsnHostPoolStats");
    CHECK(live == 0 && bytes == 0, "%d pinned block(s) (%lld bytes) never came back to the pool", live, bytes);
    char shards[256] = {0};
    const int D = lsnHostShardDescribe(8, 0, nullptr, shards, sizeof(shards));
    printf("soak: %d iteration(s) per thread, merge calls over %d device part(s) [%s], pool %d live / %d pooled, fault points %lld, %d check(s) failed\n", iters, D,
           shards, live, pooled, lsnTestFaultPoints(1), g_bad.load());
    return g_bad.load() ? 1 : 0;
}
