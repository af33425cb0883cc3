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
#include "../../livescan3d_amd/csrc/fusion_shared.hpp"

#include <atomic>
#include <cstdint>
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

std::atomic<long long> g_launches{0}, g_allocs{0}, g_frees{0};

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

hipError_t hipLaunchKernel(const void *fn, dim3 grid, dim3, void **args, size_t, hipStream_t)
{
    std::string name;
    {
        std::lock_guard<std::mutex> g(g_reg_mu);
        auto it = registry().find(fn);
        if (it != registry().end()) name = it->second;
    }
    g_launches++;
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
    return *p ? hipSuccess : (g_last = hipErrorOutOfMemory);
}
hipError_t hipFree(void *p)
{
    if (p) g_frees++;
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned int)
{
    *p = zalloc(n);
    g_allocs++;
    return *p ? hipSuccess : (g_last = hipErrorOutOfMemory);
}
hipError_t hipHostFree(void *p)
{
    if (p) g_frees++;
    free(p);
    return hipSuccess;
}

hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind)
{
    if (n) memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(dst, src, n, k); }
hipError_t hipMemcpyWithStream(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(dst, src, n, k); }
hipError_t hipMemcpyPeerAsync(void *dst, int, const void *src, int, size_t n, hipStream_t) { return hipMemcpy(dst, src, n, hipMemcpyDefault); }
hipError_t hipMemset(void *p, int v, size_t n)
{
    if (n) memset(p, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { return hipMemset(p, v, n); }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int)
{
    *s = reinterpret_cast<hipStream_t>(zalloc(64));
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
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
void lsnFakeHipStats(long long *launches, long long *allocs, long long *frees)
{
    if (launches) *launches = g_launches.load();
    if (allocs) *allocs = g_allocs.load();
    if (frees) *frees = g_frees.load();
}

}  // extern "C"
