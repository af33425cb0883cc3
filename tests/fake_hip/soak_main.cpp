// soak_main.cpp -- drives the exports of libNativeUtils' HOST side, built against tests/fake_hip (no GPU) under a sanitizer.
//   1. every export with NULL pointers and zero sizes; random rigs of sensors of different sizes through every host export, the caller's
//      arrays exactly as long as needed; the inbound formats (frame messages, recordings) valid, truncated, with flipped bits and lying headers;
//   2. the call mix of LiveScanServer from four threads at once (MainWindowForm.cs:238,304: updateWorker's merge calls, refineWorker's
//      single-sensor calls + ICP, plus the radial export and the last-mesh formats), every result checked against what the runtime
//      double's kernels "compute" (every non-zero depth pixel survives, one triangle per vertex);
//   3. the pool of pinned mesh blocks must be empty at the end.
// $LSN_HOST_DEVICES (read by the library) switches the merge calls to the sharded flow; $LSN_TEST_FAIL_ALLOC / $LSN_TEST_THROW make
// single calls fail, which must leave empty meshes and still an empty pool.  Exit code 0 = all checks held (the sanitizer adds its own).
#include "../../include/NativeUtils.h"

#include <cstddef>
extern "C" int hipMalloc(void **, size_t);   // the runtime double's (fake_hip.cpp); hipSuccess = 0
extern "C" int hipFree(void *);
static int fakeDevMalloc(void **p, size_t n) { return hipMalloc(p, n); }
static int fakeDevFree(void *p) { return hipFree(p); }

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
    // (device memory for the device-resident call below comes straight from the runtime double: this file includes no HIP header)
    // the chained tick as one call: NULL handles and arguments, then a real two-half pipeline on the double (two plans, side stream, fork / join):
    // what it allocates and enqueues is checked by the sanitizers and by the double's device discipline
    CHECK(lsnTickCreate(0, 0, 0, nullptr, nullptr) == nullptr, "lsnTickCreate(0)");
    lsnTickDestroy(nullptr);
    CHECK(lsnTickRun(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == -1, "lsnTickRun(null)");
    CHECK(lsnTickSetParams(nullptr, nullptr, nullptr, nullptr, nullptr) == -1, "lsnTickSetParams(null)");
    CHECK(lsnTickCapacity(nullptr) == 0 && lsnTickTriangleCapacity(nullptr) == 0 && lsnTickParts(nullptr) == 0, "lsnTick getters(null)");
    {
        const int T = 9, n = 2, w[2] = {64, 24}, h[2] = {48, 16};
        LsnTick *tk = lsnTickCreate(0, T, n, w, h);
        CHECK(g_faults || (tk != nullptr && lsnTickParts(tk) == 2), "lsnTickCreate");   // (a fault hook may hit any of its allocations: then it returns NULL)
        if (tk) {
            const long long cap = lsnTickCapacity(tk), tcap = lsnTickTriangleCapacity(tk);
            CHECK(cap == 64 * 48 + 24 * 16 && tcap == 2 * cap, "lsnTick capacities");
            float intr[14], wt[24], bounds[6] = {-2, -2, -2, 2, 2, 2};
            for (int i = 0; i < n; i++) {
                const float k[7] = {w[i] / 2.0f, h[i] / 2.0f, 365.0f * w[i] / 512.0f, 365.0f * w[i] / 512.0f, 0.09f, -0.05f, 0.01f};
                memcpy(intr + 7 * i, k, sizeof(k));
                const float p[12] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1};
                memcpy(wt + 12 * i, p, sizeof(p));
            }
            void *d_in = nullptr, *c_in = nullptr, *d_co = nullptr, *c_co = nullptr, *v = nullptr, *tr = nullptr;
            int *o = nullptr, *to = nullptr;
            CHECK(lsnTickRun(tk, &d_in, &c_in, &d_co, &c_co, &v, o, &tr, to, nullptr) == -1, "lsnTickRun before lsnTickSetParams / with null tables");
            const bool set = lsnTickSetParams(tk, intr, wt, bounds, nullptr) == 0;
            CHECK(g_faults || set, "lsnTickSetParams");
            bool ok = fakeDevMalloc(&d_in, 2 * cap * T)== 0 && fakeDevMalloc(&c_in, 3 * cap * T)== 0 && fakeDevMalloc(&d_co, 2 * cap * T)== 0 &&
                      fakeDevMalloc(&c_co, 3 * cap * T)== 0 && fakeDevMalloc(&v, 16 * cap * T)== 0 && fakeDevMalloc(&tr, 12 * tcap * T)== 0 &&
                      fakeDevMalloc((void **)&o, sizeof(int) * (n + 1) * T)== 0 && fakeDevMalloc((void **)&to, sizeof(int) * (n + 1) * T)== 0;
            CHECK(g_faults || ok, "device buffers for lsnTickRun");
            if (ok && set)
                for (int rep = 0; rep < 2; rep++) {
                    char why[256] = "";
                    const int rc = lsnTickRun(tk, d_in, c_in, d_co, c_co, v, o, tr, to, nullptr);
                    if (rc) (void)lsnGetLastError(why, sizeof(why));
                    CHECK(g_faults || rc == 0, "lsnTickRun: %s", why);
                }
            for (void *q : {d_in, c_in, d_co, c_co, v, tr, (void *)o, (void *)to}) (void)fakeDevFree(q);
            lsnTickDestroy(tk);
        }
    }
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

// The inbound formats are what arrives from the network and from disk: valid messages, then the same messages truncated, with bytes
// flipped and with headers that lie about sizes, from buffers exactly as long as what is handed over.  Nothing may read or write outside
// them (the sanitizers watch), and every answer is either a sensible length or -1.
void parser_fuzz(int rounds)
{
    uint32_t x = 777;
    auto rnd = [&](uint32_t n) {
        x = x * 1664525u + 1013904223u;
        return (x >> 8) % n;
    };
    for (int r = 0; r < rounds; r++) {
        const int w = 1 + (int)rnd(40), h = 1 + (int)rnd(30), nb = (int)rnd(3);
        // body block (liveScanClient.cpp:233-268): i32 count, then per body one flag byte, i32 joints, 28 bytes per joint
        const int nj = 2;
        std::vector<unsigned char> depth((size_t)w * h * 2), rgb((size_t)w * h * 3), bodies(4 + (size_t)nb * (5 + 28 * nj), 0);
        for (auto &b : depth) b = (unsigned char)rnd(256);
        for (auto &b : rgb) b = (unsigned char)rnd(256);
        memcpy(bodies.data(), &nb, 4);
        for (int k = 0; k < nb; k++) memcpy(bodies.data() + 4 + (size_t)k * (5 + 28 * nj) + 1, &nj, 4);
        const int level = (lsnZstdAvailable() && (r & 1)) ? 3 : 0;
        std::vector<unsigned char> msg(16 + depth.size() + rgb.size() + bodies.size() + 1024);
        const long long len = lsnFrameEncode(depth.data(), rgb.data(), w, h, bodies.data(), (int)bodies.size(), level, msg.data(), (long long)msg.size());
        CHECK(len > 16, "lsnFrameEncode %dx%d level %d: %lld", w, h, level, len);
        if (len <= 16) continue;
        msg.resize((size_t)len);
        for (int variant = 0; variant < 12; variant++) {
            std::vector<unsigned char> m = msg;                       // exact size: the sanitizer sees any byte read past it
            if (variant >= 1 && variant <= 4) m.resize(16 + rnd((uint32_t)(m.size() - 16)));            // truncated payload
            if (variant >= 5 && variant <= 8)
                for (int k = 0; k < 1 + (int)rnd(8); k++) m[rnd((uint32_t)m.size())] ^= (unsigned char)(1u << rnd(8));   // flipped bits, header included
            if (variant >= 9) {                                                                           // a header that lies
                int lie = variant == 9 ? 0x7FFFFFFF : variant == 10 ? -5 : (int)rnd(1u << 20);
                memcpy(m.data() + 4 * rnd(4), &lie, 4);
            }
            LsnFrameInfo info;
            const int rc = lsnFrameParseHeader(m.data(), &info);
            if (rc != 0) continue;
            if (info.width <= 0 || info.height <= 0 || (long long)info.width * info.height > (1 << 22)) continue;   // a caller sizes its buffers from these
            const size_t px = (size_t)info.width * info.height;
            std::vector<unsigned char> d_out(px * 2), c_out(px * 3), b_out(256);
            int n_bodies = -1;
            const int have = (int)m.size() - 16;
            const long long got = lsnFrameDecode(m.data() + 16, std::min(have, info.payload_bytes), info.compressed, info.width, info.height, d_out.data(), c_out.data(),
                                                 b_out.data(), (int)b_out.size(), &n_bodies);
            CHECK(got == -1 || got >= 4, "lsnFrameDecode returned %lld", got);
            if (variant == 0)
                CHECK(got == (long long)bodies.size() && d_out == depth && c_out == rgb, "round trip of a %dx%d frame (level %d): body block %lld of %zu, depth %d, colours %d", w,
                      h, level, got, bodies.size(), (int)(d_out == depth), (int)(c_out == rgb));
        }
        // a recording of three such frames, then cut and damaged
        std::vector<unsigned char> file(3 * (msg.size() + 96));
        long long at = 0;
        for (int k = 0; k < 3; k++) {
            const long long wrote = lsnRecordingAppend(file.data() + at, (long long)file.size() - at, msg.data(), (int)msg.size(), 1000 * k);
            CHECK(wrote > (long long)msg.size(), "lsnRecordingAppend");
            if (wrote > 0) at += wrote;
        }
        file.resize((size_t)at);
        for (int variant = 0; variant < 6; variant++) {
            std::vector<unsigned char> f = file;
            if (variant == 1 || variant == 2) f.resize(rnd((uint32_t)f.size()));
            if (variant >= 3)
                for (int k = 0; k < 1 + (int)rnd(6); k++) f[rnd((uint32_t)f.size())] = (unsigned char)rnd(256);
            long long pos = 0;
            int frames = 0;
            while (pos >= 0 && pos < (long long)f.size() && frames < 16) {
                long long off = 0;
                int flen = 0, ts = 0;
                const long long next = lsnRecordingNext(f.data(), (long long)f.size(), pos, &off, &flen, &ts);
                if (next < 0) break;
                CHECK(next > pos && off >= 0 && flen >= 0 && off + flen <= (long long)f.size(), "lsnRecordingNext: record [%lld, +%d) in a file of %zu", off, flen, f.size());
                pos = next;
                frames++;
            }
            if (variant == 0) CHECK(frames == 3, "a recording of three frames read back %d", frames);
        }
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4;
    null_sweep();
    ragged_rigs(3 * iters);
    parser_fuzz(20 * iters);
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
