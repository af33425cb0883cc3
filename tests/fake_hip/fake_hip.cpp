// fake_hip.cpp -- TEST DOUBLE of the HIP runtime, not a product path.  It exists so that the HOST side of libNativeUtils (abi.hip and host_flows.hip's call
// flows, lanes, pinned pool, worker threads; the plans' buffer management in fusion.hip / mesh.hip / radial.hip / exchange.hip / icp.hip /
// wire.hip) can be compiled for the host alone (clang -x hip --cuda-host-only) with -fsanitize=address,undefined or -fsanitize=thread and
// run in a container without a GPU: GPU sanitizers are not available on the pool, and the host glue is where the races and the leaks live.
//
// What it is: "device" and "pinned" memory are aligned heap blocks, streams are synchronous (a copy or a launch has completed when the
// call returns), events are tokens, `n` devices are numbers ($LSN_FAKE_HIP_DEVICES, default 2).  Kernel launches arrive here as
// hipLaunchKernel(host stub, grid, block, args): the stub is looked up in what __hipRegisterFunction recorded, and the handful of kernels
// whose RESULTS the host code reads back are emulated just far enough to return plausible, in-bounds values -- every non-zero depth pixel
// "survives", one triangle per pixel with a vertex -- so that counts, offsets, mirrors and every byte the real kernels would write are
// written (ASan checks the extents, TSan the ordering between the threads).  Every other kernel is a no-op on zero-filled memory.
// NOTHING here is numerically meaningful; parity is tested on the GPU (tests/*_gpu.py).
//
// Device discipline (round 6): the devices are DISTINCT.  Every device block, pinned block and stream remembers the device it was
// created under (the calling thread's hipSetDevice) and every use is checked against the device current at the call: a device pointer
// in a copy, a memset or the arguments of an emulated kernel, the stream of a launch or an asynchronous copy, both ends of
// hipMemcpyPeerAsync against the ordinals it names, a pinned block that is not hipHostMallocPortable used under another device than
// the one it was allocated under, a kernel argument that points at pageable memory.  A violation prints "CHECK failed: fake hip: ..."
// (tests/test_host_sanitizers.py fails on that line) -- this is how the sharded host flow's hipSetDevice discipline and its
// portable-pinned-block requirement are checked on a box that has one GPU or none.
#include "../../livescan3d_amd/csrc/fusion_shared.hpp"

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace {

struct CallConfig {
    dim3 grid, block;
    size_t shmem = 0;
    hipStream_t stream = nullptr;
};
thread_local CallConfig g_cfg;
thread_local int g_device = 0;
thread_local hipError_t g_last = hipSuccess;

std::mutex g_reg_mu;
std::map<const void *, std::string> &registry()
{
    static std::map<const void *, std::string> *r = new std::map<const void *, std::string>();
    return *r;
}

std::atomic<long long> g_launches{0}, g_allocs{0}, g_frees{0}, g_violations{0};

// ---- who owns what ----------------------------------------------------------------------------------------------------------------
enum Kind { kDeviceMem = 0, kPinnedMem = 1, kStream = 2 };
struct Block {
    size_t size;
    Kind kind;
    int dev;
    bool portable;
};
std::mutex g_own_mu;
std::map<uintptr_t, Block> &owners()
{
    static std::map<uintptr_t, Block> *m = new std::map<uintptr_t, Block>();
    return *m;
}
void own_add(void *p, size_t n, Kind kind, bool portable)
{
    if (!p) return;
    std::lock_guard<std::mutex> g(g_own_mu);
    owners()[reinterpret_cast<uintptr_t>(p)] = Block{n ? n : 1, kind, g_device, portable};
}
void own_del(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> g(g_own_mu);
    owners().erase(reinterpret_cast<uintptr_t>(p));
}
bool own_find(const void *p, Block &out)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    std::lock_guard<std::mutex> g(g_own_mu);
    auto it = owners().upper_bound(a);
    if (it == owners().begin()) return false;
    --it;
    if (a >= it->first + it->second.size) return false;
    out = it->second;
    return true;
}
extern "C" void __sanitizer_print_stack_trace(void) __attribute__((weak));
void violation(const char *what, const char *where, int have, int want)
{
    if (g_violations++ == 0 && __sanitizer_print_stack_trace) __sanitizer_print_stack_trace();   // where the first one came from
    fprintf(stderr, "CHECK failed: fake hip: %s in %s (belongs to device %d, used under device %d)\n", what, where, have, want);
}
// a pointer the device `dev` is about to touch; must_be_gpu_visible: pageable memory is an error too (kernel arguments)
void check_use(const void *p, const char *where, int dev, bool must_be_gpu_visible)
{
    if (!p) return;
    Block b;
    if (!own_find(p, b)) {
        if (must_be_gpu_visible) violation("a pointer to pageable memory", where, -1, dev);
        return;
    }
    if (b.kind == kDeviceMem && b.dev != dev) violation("a device pointer of another device", where, b.dev, dev);
    if (b.kind == kPinnedMem && !b.portable && b.dev != dev) violation("a pinned block that is not hipHostMallocPortable", where, b.dev, dev);
    if (b.kind == kStream) violation("a stream handle used as memory", where, b.dev, dev);
}
void check_stream(hipStream_t s, const char *where)
{
    if (!s) return;   // the null stream of the current device
    Block b;
    if (!own_find(s, b) || b.kind != kStream) return;
    if (b.dev != g_device) violation("a stream of another device", where, b.dev, g_device);
}
// the pointer members of the emulated kernels' argument blocks (null = not used by the launch; everything else must be memory the
// current device may touch -- a pointer to pageable memory is an error as well)
void check_fuse_args(const FuseArgs &a, const char *where)
{
    const void *ptrs[] = {a.frames, a.tiles, a.params, a.xtab, a.ytab, a.depth, a.rgb, a.out, a.tile_counts, a.run_state, a.ticket, a.offsets, a.pixmap,
                          a.pm_first, a.pm_mask, a.depth_next, a.tile_counts_next, a.error_flag, a.thr, a.group_end_mirror, a.offsets_mirror};
    for (const void *p : ptrs) check_use(p, where, g_device, true);
}
void check_tri_args(const TriArgs &t, const char *where)
{
    const void *ptrs[] = {t.frames, t.tiles, t.depth, t.pixmap, t.pm_first, t.pm_mask, t.tri, t.tile_counts, t.codes};
    for (const void *p : ptrs) check_use(p, where, g_device, true);
}

int device_count()
{
    static const int n = getenv("LSN_FAKE_HIP_DEVICES") ? atoi(getenv("LSN_FAKE_HIP_DEVICES")) : 2;
    return n;
}

void *zalloc(size_t n)
{
    void *p = nullptr;
    if (posix_memalign(&p, 256, n ? n : 1) != 0) return nullptr;
    memset(p, 0, n ? n : 1);
    return p;
}

// ---- the kernels whose results the host reads -----------------------------------------------------------------------------------

int tile_pixels(const FuseArgs &a, int tile, const unsigned short *&dep, int tick)
{
    const TileDesc td = a.tiles[tile];
    const FrameDesc fd = a.frames[td.frame];
    const int px0 = (tile - fd.tile_start) * kTile;
    dep = a.depth + tick * a.tick_depth_stride + fd.depth_off + px0;
    return fd.npix - px0 < kTile ? fd.npix - px0 : kTile;
}

int count_tile(const FuseArgs &a, int tick, int tile)
{
    const unsigned short *dep;
    const int n = tile_pixels(a, tile, dep, tick);
    int c = 0;
    for (int i = 0; i < n; i++) c += dep[i] != 0;
    return c;
}

void emu_count(const FuseArgs &a)   // count_thr_kernel / fuse_kernel<0>
{
    for (int tick = 0; tick < a.n_ticks; tick++)
        for (int tile = 0; tile < a.tiles_per_tick; tile++) a.tile_counts[(long long)tick * a.tiles_per_tick + tile] = count_tile(a, tick, tile);
}

void emu_scan(int *tc, int tiles_per_tick, const FrameDesc *frames, int n_frames, int *offsets, int *mirror, int n_ticks)   // scan_kernel
{
    for (int tick = 0; tick < n_ticks; tick++) {
        int *t = tc + (long long)tick * tiles_per_tick, run = 0;
        for (int i = 0; i < tiles_per_tick; i++) {
            const int v = t[i];
            t[i] = run;
            run += v;
        }
        for (int f = 0; f <= n_frames; f++) {
            const int v = f < n_frames ? t[frames[f].tile_start] : run;
            offsets[(long long)tick * (n_frames + 1) + f] = v;
            if (mirror) mirror[(long long)tick * (n_frames + 1) + f] = v;
        }
    }
}

// one tile's "vertices" at `base` inside the tick's cloud + the pixel -> vertex map; returns the tile's count
int write_tile(const FuseArgs &a, int tick, int tile, int base)
{
    const unsigned short *dep;
    const int n = tile_pixels(a, tile, dep, tick);
    const TileDesc td = a.tiles[tile];
    const FrameDesc fd = a.frames[td.frame];
    const int px0 = (tile - fd.tile_start) * kTile;
    uint4 *out = a.out + tick * a.tick_vert_stride + base;
    int c = 0;
    for (int i0 = 0; i0 < n; i0 += 8) {
        unsigned int mask = 0;
        const int first = base + c;
        for (int k = 0; k < 8 && i0 + k < n; k++) {
            if (dep[i0 + k] == 0) {
                if (a.pixmap) a.pixmap[tick * a.tick_depth_stride + fd.depth_off + px0 + i0 + k] = -1;
                continue;
            }
            if (a.pixmap) a.pixmap[tick * a.tick_depth_stride + fd.depth_off + px0 + i0 + k] = base + c;
            out[c] = make_uint4(0xFF000000u | dep[i0 + k], (unsigned)tile, (unsigned)(i0 + k), (unsigned)tick);
            mask |= 1u << k;
            c++;
        }
        if (a.pm_first && (fd.w % 8) == 0) {
            const long long g = (tick * a.tick_depth_stride + fd.depth_off + px0 + i0) >> 3;
            a.pm_first[g] = first;
            a.pm_mask[g] = (unsigned char)mask;
        }
    }
    return c;
}

void emu_write(const FuseArgs &a)   // fuse_kernel<1>: at the scanned prefixes
{
    for (int tick = 0; tick < a.n_ticks; tick++)
        for (int tile = 0; tile < a.tiles_per_tick; tile++) write_tile(a, tick, tile, a.tile_counts[(long long)tick * a.tiles_per_tick + tile]);
}

void emu_single_pass(const FuseArgs &a, int grid)   // fuse_kernel<4>: tiles [tile0, tile0 + grid) of every tick, continuing the earlier launches of the tick
{
    const int per_tick = a.n_ticks > 1 ? a.tiles_per_tick : grid;
    for (int tick = 0; tick < a.n_ticks; tick++) {
        // (no real kernel ever runs on this memory: the tick's first look-back word holds the running end here)
        unsigned long long *running = a.run_state + (long long)tick * a.tiles_per_tick;
        if (a.tile0 == 0) *running = 0;
        int *off = a.offsets + tick * (a.n_frames + 1);
        int end = (int)*running;
        for (int tile = a.tile0; tile < a.tile0 + per_tick; tile++) {
            const TileDesc td = a.tiles[tile];
            if (tile == a.frames[td.frame].tile_start) {
                off[td.frame] = end;
                if (a.offsets_mirror) a.offsets_mirror[td.frame] = end;
            }
            end += write_tile(a, tick, tile, end);
            if (tile == a.tiles_per_tick - 1) {
                off[a.n_frames] = end;
                if (a.offsets_mirror) a.offsets_mirror[a.n_frames] = end;
            }
        }
        *running = (unsigned long long)end;
        if (a.group_end_mirror) *a.group_end_mirror = end;
    }
}

// pixels of a tile that have a vertex: one "triangle" each
template <class F>
void for_vertices_of_tile(const TriArgs &t, int tick, int tile, bool vec, F &&f)
{
    const TileDesc td = t.tiles[tile];
    const FrameDesc fd = t.frames[td.frame];
    const int px0 = (tile - fd.tile_start) * kTile;
    const int n = fd.npix - px0 < kTile ? fd.npix - px0 : kTile;
    for (int i = 0; i < n; i++) {
        const long long p = tick * t.tick_pix_stride + fd.depth_off + px0 + i;
        int idx = -1;
        if (vec) {
            const unsigned int m = t.pm_mask[p >> 3];
            if ((m >> (p & 7)) & 1u) idx = t.pm_first[p >> 3] + __builtin_popcount(m & ((1u << (p & 7)) - 1u));
        } else {
            idx = t.pixmap[p];
        }
        if (idx >= 0) f(idx);
    }
}

void emu_tri(const TriArgs &t, int grid, bool write, bool vec)   // tri_kernel<0 / 1>
{
    const int n_ticks = grid / t.tiles_per_tick;
    for (int tick = 0; tick < n_ticks; tick++)
        for (int tile = 0; tile < t.tiles_per_tick; tile++) {
            const long long lin = (long long)tick * t.tiles_per_tick + tile;
            if (!write) {
                int c = 0;
                for_vertices_of_tile(t, tick, tile, vec, [&](int) { c++; });
                t.tile_counts[lin] = c;
            } else {
                int *dst = t.tri + 3 * (tick * t.tick_tri_stride + t.tile_counts[lin]);
                for_vertices_of_tile(t, tick, tile, vec, [&](int idx) {
                    dst[0] = dst[1] = dst[2] = idx + t.index_base;
                    dst += 3;
                });
            }
        }
}

bool has(const std::string &name, const char *what) { return name.find(what) != std::string::npos; }

}  // namespace

// ---- what clang's host stubs call -------------------------------------------------------------------------------------------------

extern "C" {

void **__hipRegisterFatBinary(const void *)
{
    static void *handle = nullptr;
    return &handle;
}
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *hostFunction, char *, const char *deviceName, unsigned int, void *, void *, void *, void *, int *)
{
    std::lock_guard<std::mutex> g(g_reg_mu);
    registry()[hostFunction] = deviceName ? deviceName : "";
}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}

hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream)
{
    g_cfg.grid = grid;
    g_cfg.block = block;
    g_cfg.shmem = shmem;
    g_cfg.stream = stream;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream)
{
    *grid = g_cfg.grid;
    *block = g_cfg.block;
    *shmem = g_cfg.shmem;
    *stream = g_cfg.stream;
    return hipSuccess;
}

hipError_t hipLaunchKernel(const void *fn, dim3 grid, dim3, void **args, size_t, hipStream_t stream)
{
    std::string name;
    {
        std::lock_guard<std::mutex> g(g_reg_mu);
        auto it = registry().find(fn);
        if (it != registry().end()) name = it->second;
    }
    g_launches++;
    check_stream(stream, name.c_str());
    if (has(name, "16count_thr_kernel") || has(name, "11fuse_kernelILi")) check_fuse_args(*static_cast<const FuseArgs *>(args[0]), name.c_str());
    else if (has(name, "10tri_kernelILi")) check_tri_args(*static_cast<const TriArgs *>(args[0]), name.c_str());
    else if (has(name, "11scan_kernel"))
        for (int k : {0, 2, 4, 5}) check_use(*static_cast<void **>(args[k]), name.c_str(), g_device, true);
    // Itanium names: <len><identifier>I<template args>E...; Li<N>E an int argument, Lb<0|1>E a bool
    if (has(name, "16count_thr_kernel") || has(name, "11fuse_kernelILi0E")) emu_count(*static_cast<const FuseArgs *>(args[0]));
    else if (has(name, "11scan_kernel"))
        emu_scan(*static_cast<int **>(args[0]), *static_cast<int *>(args[1]), *static_cast<const FrameDesc **>(args[2]), *static_cast<int *>(args[3]),
                 *static_cast<int **>(args[4]), *static_cast<int **>(args[5]), (int)grid.x);
    else if (has(name, "11fuse_kernelILi1E")) emu_write(*static_cast<const FuseArgs *>(args[0]));
    else if (has(name, "11fuse_kernelILi4E")) emu_single_pass(*static_cast<const FuseArgs *>(args[0]), (int)grid.x);
    else if (has(name, "10tri_kernelILi0E")) emu_tri(*static_cast<const TriArgs *>(args[0]), (int)grid.x, false, has(name, "10tri_kernelILi0ELb1E"));
    else if (has(name, "10tri_kernelILi1E")) emu_tri(*static_cast<const TriArgs *>(args[0]), (int)grid.x, true, has(name, "10tri_kernelILi1ELb1E"));
    return hipSuccess;
}

// ---- the runtime API the library uses ------------------------------------------------------------------------------------------

hipError_t hipGetDeviceCount(int *n)
{
    *n = device_count();
    return *n > 0 ? hipSuccess : (g_last = hipErrorNoDevice);
}
hipError_t hipSetDevice(int d)
{
    if (d < 0 || d >= device_count()) return g_last = hipErrorInvalidDevice;
    g_device = d;
    return hipSuccess;
}
hipError_t hipGetLastError(void)
{
    const hipError_t e = g_last;
    g_last = hipSuccess;
    return e;
}
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "fake hip: out of memory" : "fake hip: error"; }

hipError_t hipMalloc(void **p, size_t n)
{
    *p = zalloc(n);
    g_allocs++;
    own_add(*p, n, kDeviceMem, false);
    return *p ? hipSuccess : (g_last = hipErrorOutOfMemory);
}
hipError_t hipFree(void *p)
{
    if (p) g_frees++;
    own_del(p);
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned int flags)
{
    *p = zalloc(n);
    g_allocs++;
    own_add(*p, n, kPinnedMem, (flags & hipHostMallocPortable) != 0);
    return *p ? hipSuccess : (g_last = hipErrorOutOfMemory);
}
hipError_t hipHostFree(void *p)
{
    if (p) g_frees++;
    own_del(p);
    free(p);
    return hipSuccess;
}

hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind)
{
    check_use(dst, "hipMemcpy (destination)", g_device, false);
    check_use(src, "hipMemcpy (source)", g_device, false);
    if (n) memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s)
{
    check_stream(s, "hipMemcpyAsync");
    return hipMemcpy(dst, src, n, k);
}
hipError_t hipMemcpyWithStream(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s)
{
    check_stream(s, "hipMemcpyWithStream");
    return hipMemcpy(dst, src, n, k);
}
hipError_t hipMemcpyPeerAsync(void *dst, int dst_dev, const void *src, int src_dev, size_t n, hipStream_t s)
{
    check_stream(s, "hipMemcpyPeerAsync");
    check_use(dst, "hipMemcpyPeerAsync (destination against the destination ordinal)", dst_dev, true);
    check_use(src, "hipMemcpyPeerAsync (source against the source ordinal)", src_dev, true);
    if (n) memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemset(void *p, int v, size_t n)
{
    check_use(p, "hipMemset", g_device, false);
    if (n) memset(p, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t s)
{
    check_stream(s, "hipMemsetAsync");
    return hipMemset(p, v, n);
}

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int)
{
    *s = reinterpret_cast<hipStream_t>(zalloc(64));
    own_add(*s, 64, kStream, false);
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
    own_del(s);
    free(s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }

hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned int)
{
    *e = reinterpret_cast<hipEvent_t>(zalloc(64));
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e)
{
    free(e);
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t)
{
    *ms = 0.001f;
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

// what the soak reads at the end: launches seen, blocks allocated and freed
long long lsnFakeHipViolations(void) { return g_violations.load(); }

void lsnFakeHipStats(long long *launches, long long *allocs, long long *frees)
{
    if (launches) *launches = g_launches.load();
    if (allocs) *allocs = g_allocs.load();
    if (frees) *frees = g_frees.load();
}

}  // extern "C"
