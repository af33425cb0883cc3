// abi.hip -- the reference's C-ABI exports (include/NativeUtils.h part 1) on top of the device-resident API.
//
// LiveScanServer hands over host arrays (pinned managed arrays / AllocHGlobal blocks, KinectServer.cs:354-374,
// MainWindowForm.cs:364-370) and reads host memory back (Marshal.Copy, KinectServer.cs:383), so these entry points
// add the H2D / D2H hops around the same kernels bench.py drives directly on HBM-resident data.  How a call runs -- the
// upload schedule, kernels storing into the pinned mesh blocks, the copy-engine flow, the call sharded over devices -- is
// host_flows.hip; the context it runs on (lanes, pinned pool, devices) host_ctx.hpp.  This file: argument checks, locks, the
// exception trampoline, and the exports that need no flow (error channel, device helpers, createMesh / deleteMesh, ICP, the
// outbound formats of the last mesh).
#include "host_ctx.hpp"

using namespace lsn::host;

namespace lsn {

// The error text lives in a fixed thread-local buffer: reporting a failure (an allocation failure, for one) must not allocate.
char *error_buffer() noexcept
{
    static thread_local char buf[kErrorLen] = {0};
    return buf;
}

void set_error(const char *fmt, ...) noexcept
{
    char *buf = error_buffer();
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, kErrorLen, fmt, ap);
    va_end(ap);
    if (getenv("LSN_VERBOSE")) fprintf(stderr, "[NativeUtils] %s\n", buf);
}

// Test hook of the exception trampoline (lsn::guarded): $LSN_TEST_THROW=n makes the n-th guarded entry of the process throw
// std::bad_alloc from inside the guarded region; $LSN_TEST_FAIL_ALLOC=n makes the n-th device / pinned allocation throw it
// (what a std::vector or std::map growing under memory pressure would do).  Read once.
static std::atomic<long> fault_points_seen[2];
long test_fault_points(int kind) { return kind >= 0 && kind < 2 ? fault_points_seen[kind].load() : -1; }

void test_fault_point(int kind)
{
    static const long want[2] = {getenv("LSN_TEST_THROW") ? atol(getenv("LSN_TEST_THROW")) : 0,
                                 getenv("LSN_TEST_FAIL_ALLOC") ? atol(getenv("LSN_TEST_FAIL_ALLOC")) : 0};
    const long n = ++fault_points_seen[kind];
    if (want[kind] > 0 && n == want[kind]) throw std::bad_alloc();
}

}  // namespace lsn

// (not routed through lsn::guarded: nothing in here can throw, and reading the message must not disturb it)
extern "C" int lsnGetLastError(char *buf, int len)
{
    const char *s = lsn::error_buffer();
    if (buf && len > 0) snprintf(buf, (size_t)len, "%s", s);
    return (int)strlen(s);
}

static int lsnDeviceCount_impl(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int lsnDeviceCount(void)
{
    return lsn::guarded<int>("lsnDeviceCount", static_cast<int>(-1), [&]() { return lsnDeviceCount_impl(); });
}

// ---- device memory / streams for hosts without a HIP of their own ------------------------------------------------------------

static void * lsnDeviceMalloc_impl(int device, long long bytes)
{
    lsn::clear_error();
    if (bytes <= 0) {
        lsn::set_error("lsnDeviceMalloc: bad size %lld", bytes);
        return nullptr;
    }
    LSN_HIP_NULL(hipSetDevice(device));
    void *p = nullptr;
    LSN_HIP_NULL(hipMalloc(&p, (size_t)bytes));
    return p;
}

extern "C" void * lsnDeviceMalloc(int device, long long bytes)
{
    return lsn::guarded<void *>("lsnDeviceMalloc", static_cast<void *>(nullptr), [&]() { return lsnDeviceMalloc_impl(device, bytes); });
}

static int lsnDeviceFree_impl(int device, void *d_ptr)
{
    lsn::clear_error();
    if (!d_ptr) return 0;
    LSN_HIP(hipSetDevice(device));
    LSN_HIP(hipFree(d_ptr));
    return 0;
}

extern "C" int lsnDeviceFree(int device, void *d_ptr)
{
    return lsn::guarded<int>("lsnDeviceFree", static_cast<int>(-1), [&]() { return lsnDeviceFree_impl(device, d_ptr); });
}

static int lsnDeviceUpload_impl(int device, void *d_dst, const void *h_src, long long bytes, void *stream)
{
    lsn::clear_error();
    if (!d_dst || !h_src || bytes < 0) {
        lsn::set_error("lsnDeviceUpload: bad arguments");
        return -1;
    }
    LSN_HIP(hipSetDevice(device));
    if (bytes > 0) LSN_HIP(hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, lsn::as_stream(stream)));
    return 0;
}

extern "C" int lsnDeviceUpload(int device, void *d_dst, const void *h_src, long long bytes, void *stream)
{
    return lsn::guarded<int>("lsnDeviceUpload", static_cast<int>(-1), [&]() { return lsnDeviceUpload_impl(device, d_dst, h_src, bytes, stream); });
}

static int lsnDeviceDownload_impl(int device, void *h_dst, const void *d_src, long long bytes, void *stream)
{
    lsn::clear_error();
    if (!h_dst || !d_src || bytes < 0) {
        lsn::set_error("lsnDeviceDownload: bad arguments");
        return -1;
    }
    LSN_HIP(hipSetDevice(device));
    if (bytes > 0) LSN_HIP(hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, lsn::as_stream(stream)));
    return 0;
}

extern "C" int lsnDeviceDownload(int device, void *h_dst, const void *d_src, long long bytes, void *stream)
{
    return lsn::guarded<int>("lsnDeviceDownload", static_cast<int>(-1), [&]() { return lsnDeviceDownload_impl(device, h_dst, d_src, bytes, stream); });
}

static void * lsnStreamCreate_impl(int device)
{
    lsn::clear_error();
    LSN_HIP_NULL(hipSetDevice(device));
    hipStream_t s = nullptr;
    LSN_HIP_NULL(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}

extern "C" void * lsnStreamCreate(int device)
{
    return lsn::guarded<void *>("lsnStreamCreate", static_cast<void *>(nullptr), [&]() { return lsnStreamCreate_impl(device); });
}

static int lsnStreamDestroy_impl(int device, void *stream)
{
    lsn::clear_error();
    if (!stream) return 0;
    LSN_HIP(hipSetDevice(device));
    LSN_HIP(hipStreamDestroy(lsn::as_stream(stream)));
    return 0;
}

extern "C" int lsnStreamDestroy(int device, void *stream)
{
    return lsn::guarded<int>("lsnStreamDestroy", static_cast<int>(-1), [&]() { return lsnStreamDestroy_impl(device, stream); });
}

static int lsnStreamSynchronize_impl(int device, void *stream)
{
    lsn::clear_error();
    LSN_HIP(hipSetDevice(device));
    LSN_HIP(hipStreamSynchronize(lsn::as_stream(stream)));
    return 0;
}

extern "C" int lsnStreamSynchronize(int device, void *stream)
{
    return lsn::guarded<int>("lsnStreamSynchronize", static_cast<int>(-1), [&]() { return lsnStreamSynchronize_impl(device, stream); });
}


static void generateVerticesFromDepthMap_impl(unsigned char *depth_maps, unsigned char *depth_colors, int *widths, int *heights,
                                             float *intr_params, float *wtransform_params, Mesh *out_mesh, float minX, float minY,
                                             float minZ, float maxX, float maxY, float maxZ, int depth_map_index)
{
    lsn::clear_error();
    if (!out_mesh) return;
    Ctx &c = ctx();
    Lane &l = c.single;
    std::lock_guard<std::mutex> g(l.mu);
    if (!depth_maps || !depth_colors || !widths || !heights || !intr_params || !wtransform_params || depth_map_index < 0) {
        lsn::set_error("generateVerticesFromDepthMap: bad arguments");
        empty_mesh(out_mesh);
        return;
    }
    const float b[6] = {minX, minY, minZ, maxX, maxY, maxZ};
    if (ensure_ready(c) || fuse_host(c, l, depth_maps, depth_colors, widths, heights, intr_params, wtransform_params,
                                     out_mesh, b, depth_map_index, 1, false))
        empty_mesh(out_mesh);
}

extern "C" void generateVerticesFromDepthMap(unsigned char *depth_maps, unsigned char *depth_colors, int *widths, int *heights,
                                             float *intr_params, float *wtransform_params, Mesh *out_mesh, float minX, float minY,
                                             float minZ, float maxX, float maxY, float maxZ, int depth_map_index)
{
    const bool done = lsn::guarded<bool>("generateVerticesFromDepthMap", false, [&]() { generateVerticesFromDepthMap_impl(depth_maps, depth_colors, widths, heights, intr_params, wtransform_params, out_mesh, minX, minY, minZ, maxX, maxY, maxZ, depth_map_index); return true; });
    if (!done && out_mesh) empty_mesh(out_mesh);   // nothing reaches the caller but an empty mesh and the message
}

static void generateMeshFromDepthMaps_impl(int n_maps, unsigned char *depth_maps, unsigned char *depth_colors, int *widths, int *heights,
                                          float *intr_params, float *wtransform_params, Mesh *out_mesh, bool bcolor_transfer, float minX,
                                          float minY, float minZ, float maxX, float maxY, float maxZ, bool bgenerate_triangles)
{
    lsn::clear_error();
    if (!out_mesh) return;
    Ctx &c = ctx();
    Lane &l = c.merge;
    std::lock_guard<std::mutex> g(l.mu);
    if (n_maps <= 0 || !depth_maps || !depth_colors || !widths || !heights || !intr_params || !wtransform_params) {
        if (n_maps != 0) lsn::set_error("generateMeshFromDepthMaps: bad arguments");
        empty_mesh(out_mesh);
        return;
    }
    const float b[6] = {minX, minY, minZ, maxX, maxY, maxZ};
    if (ensure_ready(c) ||
        fuse_host(c, l, depth_maps, depth_colors, widths, heights, intr_params, wtransform_params, out_mesh, b, 0, n_maps, true)) {
        empty_mesh(out_mesh);
        return;
    }
    if (bcolor_transfer || bgenerate_triangles) {
        lsn::set_error("generateMeshFromDepthMaps: colour transfer / overlay merge are outside this library's scope; "
                       "returned the cropped vertices of all sensors (flags false,false behaviour)");
        if (!c.warned_flags) {
            // nobody on the C# side reads lsnGetLastError, and bGenerateTriangles = true is LiveScanServer's default
            // (KinectSettings.cs:50): say it once per process where an operator can see it
            c.warned_flags = true;
            fprintf(stderr, "[NativeUtils] generateMeshFromDepthMaps was called with bcolor_transfer=%d bgenerate_triangles=%d: this library "
                            "implements the (false, false) behaviour only (no cross-view overlay merge, no colour transfer); the mesh "
                            "returned is the unmerged one. Set bGenerateTriangles / bColorTransfer to false in LiveScanServer's settings.\n",
                    (int)bcolor_transfer, (int)bgenerate_triangles);
        }
    }
}

extern "C" void generateMeshFromDepthMaps(int n_maps, unsigned char *depth_maps, unsigned char *depth_colors, int *widths, int *heights,
                                          float *intr_params, float *wtransform_params, Mesh *out_mesh, bool bcolor_transfer, float minX,
                                          float minY, float minZ, float maxX, float maxY, float maxZ, bool bgenerate_triangles)
{
    const bool done = lsn::guarded<bool>("generateMeshFromDepthMaps", false, [&]() { generateMeshFromDepthMaps_impl(n_maps, depth_maps, depth_colors, widths, heights, intr_params, wtransform_params, out_mesh, bcolor_transfer, minX, minY, minZ, maxX, maxY, maxZ, bgenerate_triangles); return true; });
    if (!done && out_mesh) empty_mesh(out_mesh);   // nothing reaches the caller but an empty mesh and the message
}

static void lsnCorrectAndGenerateMesh_impl(int n_maps, unsigned char *depth_maps, unsigned char *depth_colors, int *widths, int *heights,
                                          float *intr_params, float *wtransform_params, Mesh *out_mesh, float minX, float minY, float minZ,
                                          float maxX, float maxY, float maxZ, int write_back_corrected)
{
    lsn::clear_error();
    if (!out_mesh) return;
    Ctx &c = ctx();
    Lane &l = c.merge;
    std::lock_guard<std::mutex> g(l.mu);
    if (n_maps <= 0 || !depth_maps || !depth_colors || !widths || !heights || !intr_params || !wtransform_params) {
        if (n_maps != 0) lsn::set_error("lsnCorrectAndGenerateMesh: bad arguments");
        empty_mesh(out_mesh);
        return;
    }
    const float b[6] = {minX, minY, minZ, maxX, maxY, maxZ};
    if (ensure_ready(c) || fuse_host(c, l, depth_maps, depth_colors, widths, heights, intr_params, wtransform_params, out_mesh, b, 0, n_maps,
                                     true, true, write_back_corrected ? depth_maps : nullptr, write_back_corrected ? depth_colors : nullptr))
        empty_mesh(out_mesh);
}

extern "C" void lsnCorrectAndGenerateMesh(int n_maps, unsigned char *depth_maps, unsigned char *depth_colors, int *widths, int *heights,
                                          float *intr_params, float *wtransform_params, Mesh *out_mesh, float minX, float minY, float minZ,
                                          float maxX, float maxY, float maxZ, int write_back_corrected)
{
    const bool done = lsn::guarded<bool>("lsnCorrectAndGenerateMesh", false, [&]() { lsnCorrectAndGenerateMesh_impl(n_maps, depth_maps, depth_colors, widths, heights, intr_params, wtransform_params, out_mesh, minX, minY, minZ, maxX, maxY, maxZ, write_back_corrected); return true; });
    if (!done && out_mesh) empty_mesh(out_mesh);   // nothing reaches the caller but an empty mesh and the message
}

static void depthMapAndColorSetRadialCorrection_impl(int n_maps, unsigned char *depth_maps, unsigned char *depth_colors, int *widths,
                                                    int *heights, float *intr_params)
{
    lsn::clear_error();
    if (n_maps <= 0 || !depth_maps || !depth_colors || !widths || !heights || !intr_params) {
        if (n_maps != 0) lsn::set_error("depthMapAndColorSetRadialCorrection: bad arguments");
        return;
    }
    Ctx &c = ctx();
    Lane &l = c.merge;
    std::lock_guard<std::mutex> g(l.mu);
    if (ensure_ready(c)) return;
    radial_host(c, l, n_maps, depth_maps, depth_colors, widths, heights, intr_params);
}

extern "C" void depthMapAndColorSetRadialCorrection(int n_maps, unsigned char *depth_maps, unsigned char *depth_colors, int *widths,
                                                    int *heights, float *intr_params)
{
    lsn::guarded_void("depthMapAndColorSetRadialCorrection", [&]() { depthMapAndColorSetRadialCorrection_impl(n_maps, depth_maps, depth_colors, widths, heights, intr_params); });
}

static Mesh * createMesh_impl(void)
{
    Mesh *m = (Mesh *)calloc(1, sizeof(Mesh));  // zeroed like depthprocessing.cpp:1820-1825
    return m;
}

extern "C" Mesh * createMesh(void)
{
    return lsn::guarded<Mesh *>("createMesh", static_cast<Mesh *>(nullptr), [&]() { return createMesh_impl(); });
}

static void deleteMesh_impl(Mesh *mesh)
{
    if (!mesh) return;
    Ctx &c = ctx();
    if (mesh->triangles) pinned_put(c, mesh->triangles);   // a pinned block of ours; the static empty array or a foreign pointer is left alone
    if (mesh->vertices) pinned_put(c, mesh->vertices);
    mesh->triangles = nullptr;
    mesh->vertices = nullptr;
}

extern "C" void deleteMesh(Mesh *mesh)
{
    lsn::guarded_void("deleteMesh", [&]() { deleteMesh_impl(mesh); });
}

static float ICP_impl(Point3f *verts1, Point3f *verts2, int nVerts1, int nVerts2, float *R, float *t, int maxIter)
{
    lsn::clear_error();
    const float error = 1.0f;  // icp.cpp:85,176
    if (!verts1 || !verts2 || !R || !t || nVerts1 <= 0 || nVerts2 <= 0 || maxIter <= 0) {
        // the reference would throw out of nanoflann on an empty cloud (include/nanoflann.h:904); callers guard
        if (nVerts1 <= 0 || nVerts2 <= 0) lsn::set_error("ICP: empty cloud (nVerts1=%d nVerts2=%d)", nVerts1, nVerts2);
        return error;
    }
    Ctx &c = ctx();
    std::lock_guard<std::mutex> g(c.icp_mu);   // not c.mu: merge calls go on while a refine call runs
    if (ensure_ready(c)) return error;
    if (!c.icp || nVerts1 > c.icp_n1 || nVerts2 > c.icp_n2) {
        if (c.icp) lsnIcpDestroy(c.icp);
        c.icp_n1 = nVerts1 > c.icp_n1 ? nVerts1 : c.icp_n1;
        c.icp_n2 = nVerts2 > c.icp_n2 ? nVerts2 : c.icp_n2;
        c.icp = lsnIcpCreate(c.device, c.icp_n1, c.icp_n2);
        if (!c.icp) {
            c.icp_n1 = c.icp_n2 = 0;
            return error;
        }
    }
    if (c.d_v1.reserve(sizeof(float) * 3 * (size_t)nVerts1) || c.d_v2.reserve(sizeof(float) * 3 * (size_t)nVerts2) || c.d_Rt.reserve(64))
        return error;
    const char *env = getenv("LSN_NN");
    const int nn_mode = (env && strcmp(env, "brute") == 0) ? 0 : 1;
    auto fail = [&]() { return error; };
    if (hipMemcpyAsync(c.d_v1.p, verts1, sizeof(float) * 3 * (size_t)nVerts1, hipMemcpyHostToDevice, c.icp_stream) != hipSuccess ||
        hipMemcpyAsync(c.d_v2.p, verts2, sizeof(float) * 3 * (size_t)nVerts2, hipMemcpyHostToDevice, c.icp_stream) != hipSuccess ||
        hipMemcpyAsync(c.d_Rt.p, R, sizeof(float) * 9, hipMemcpyHostToDevice, c.icp_stream) != hipSuccess ||
        hipMemcpyAsync(c.d_Rt.as<float>() + 9, t, sizeof(float) * 3, hipMemcpyHostToDevice, c.icp_stream) != hipSuccess) {
        lsn::set_error("ICP: upload failed: %s", hipGetErrorString(hipGetLastError()));
        return fail();
    }
    if (lsnIcpRun(c.icp, c.d_v1.as<float>(), nVerts1, c.d_v2.as<float>(), nVerts2, c.d_Rt.as<float>(), c.d_Rt.as<float>() + 9, maxIter,
                  nn_mode, c.icp_stream))
        return fail();
    // results go to a scratch first so that the caller's buffers stay untouched when anything fails: a pinned block of the pool
    // (recycled call after call; the download runs as DMA into it)
    const size_t v2_bytes = sizeof(float) * 3 * (size_t)nVerts2;
    float *v2 = static_cast<float *>(pinned_get(c, v2_bytes + sizeof(float) * 12));
    if (!v2) return fail();
    float *Rt = v2 + (size_t)nVerts2 * 3;
    if (hipMemcpyAsync(v2, c.d_v2.p, v2_bytes, hipMemcpyDeviceToHost, c.icp_stream) != hipSuccess ||
        hipMemcpyAsync(Rt, c.d_Rt.p, sizeof(float) * 12, hipMemcpyDeviceToHost, c.icp_stream) != hipSuccess ||
        hipStreamSynchronize(c.icp_stream) != hipSuccess) {
        lsn::set_error("ICP: download failed: %s", hipGetErrorString(hipGetLastError()));
        (void)hipStreamSynchronize(c.icp_stream);
        pinned_put(c, v2);
        return fail();
    }
    memcpy(verts2, v2, v2_bytes);
    memcpy(R, Rt, sizeof(float) * 9);
    memcpy(t, Rt + 9, sizeof(float) * 3);
    pinned_put(c, v2);
    return error;
}

extern "C" float ICP(Point3f *verts1, Point3f *verts2, int nVerts1, int nVerts2, float *R, float *t, int maxIter)
{
    return lsn::guarded<float>("ICP", 1.0f, [&]() { return ICP_impl(verts1, verts2, nVerts1, nVerts2, R, t, maxIter); });
}

// ---- the outbound formats of the mesh the last merge call left in HBM (include/NativeUtils.h part 3) ----------------------------

namespace {
// requires the lane's lock; kind 0 = TransferSocket.SendFrame stream, 1 = binary PLY file image
long long last_mesh_bytes(Ctx &c, Lane &l, int kind, unsigned char *out, long long out_cap)
{
    if (ensure_ready(c)) return -1;
    if (l.last_nv < 0) {
        lsn::set_error("lsnLastMesh*: no mesh is resident (call generateMeshFromDepthMaps / generateVerticesFromDepthMap first)");
        return -1;
    }
    const int nv = l.last_nv, nt = l.last_nt;
    const long long bound = kind == 0 ? lsnTransferFrameBound(nv, nt) : lsnPlyBinaryBytes(nv, nt);
    if (!out) return bound;
    if (materialize(l)) return -1;
    std::lock_guard<std::mutex> wg(c.wire_mu);
    if (c.d_wire.reserve((size_t)bound + 16)) return -1;
    long long n = -1;
    if (kind == 0) {
        if (!c.xfer || nv > c.xfer_v || nt > c.xfer_t) {
            if (c.xfer) lsnTransferDestroy(c.xfer);
            c.xfer_v = nv > c.xfer_v ? nv : c.xfer_v;
            c.xfer_t = nt > c.xfer_t ? nt : c.xfer_t;
            c.xfer = lsnTransferCreate(c.device, c.xfer_v, c.xfer_t);
            if (!c.xfer) {
                c.xfer_v = c.xfer_t = 0;
                return -1;
            }
        }
        n = lsnTransferPack(c.xfer, l.d_out.p, nv, nt > 0 ? l.d_tri.as<int>() : nullptr, nt, c.d_wire.p, bound, l.stream);
    } else {
        n = lsnPlyPack(c.device, l.d_out.p, nv, nt > 0 ? l.d_tri.as<int>() : nullptr, nt, c.d_wire.p, bound, l.stream);
    }
    if (n < 0) return -1;
    if (n > out_cap) {
        (void)hipStreamSynchronize(l.stream);
        lsn::set_error("lsnLastMesh*: the result is %lld bytes, the buffer holds %lld", n, out_cap);
        return -1;
    }
    LSN_HIP(hipMemcpyAsync(out, c.d_wire.p, (size_t)n, hipMemcpyDeviceToHost, l.stream));
    LSN_HIP(hipStreamSynchronize(l.stream));
    return n;
}
}  // namespace

static long long lsnLastMeshTransferFrame_impl(unsigned char *out, long long out_cap)
{
    lsn::clear_error();
    Ctx &c = ctx();
    Lane *l = t_last_lane ? t_last_lane : c.last_lane.load();   // this thread's own last mesh call, else the process's
    if (!l) l = &c.merge;
    std::lock_guard<std::mutex> g(l->mu);
    return last_mesh_bytes(c, *l, 0, out, out_cap);
}

extern "C" long long lsnLastMeshTransferFrame(unsigned char *out, long long out_cap)
{
    return lsn::guarded<long long>("lsnLastMeshTransferFrame", static_cast<long long>(-1), [&]() { return lsnLastMeshTransferFrame_impl(out, out_cap); });
}

static long long lsnLastMeshPly_impl(unsigned char *out, long long out_cap)
{
    lsn::clear_error();
    Ctx &c = ctx();
    Lane *l = t_last_lane ? t_last_lane : c.last_lane.load();   // this thread's own last mesh call, else the process's
    if (!l) l = &c.merge;
    std::lock_guard<std::mutex> g(l->mu);
    return last_mesh_bytes(c, *l, 1, out, out_cap);
}

extern "C" long long lsnLastMeshPly(unsigned char *out, long long out_cap)
{
    return lsn::guarded<long long>("lsnLastMeshPly", static_cast<long long>(-1), [&]() { return lsnLastMeshPly_impl(out, out_cap); });
}
