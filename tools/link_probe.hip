// Dev helper: what the PCIe link gives a merge call that overlaps its hops (round 4, DESIGN "host path").
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/link_probe tools/link_probe.hip && /tmp/link_probe
// Measures, at the sizes of one 8 x 512x424 merge call:
//   A  device -> host: hipMemcpyAsync into a pinned block against a KERNEL storing straight into that block (plain / nontemporal 16-B stores)
//   B  pageable host -> device hipMemcpyAsync: how long the call keeps the calling thread, how long until the bytes are there
//   C  both directions at once (kernel stores down, pageable copies up; and copy engine down, pageable copies up)
//   D  host cost of a kernel launch
//   E  a kernel LOADING from a registered caller array (zero-copy upload) and what the registration costs
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));     \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int NT>
__global__ void store_kernel(uint4 *dst, size_t n16, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        v4u v = {(unsigned)i, seed, (unsigned)i ^ seed, 7u};
        if (NT) __builtin_nontemporal_store(v, (v4u *)dst + i);
        else ((v4u *)dst)[i] = v;
    }
}

__global__ void load_kernel(const uint4 *src, size_t n16, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// copies through registers: device-resident source -> (host) destination, like a write pass whose output block is host memory
__global__ void copy_kernel(uint4 *dst, const uint4 *src, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(((const v4u *)src)[i], (v4u *)dst + i);
}

__global__ void empty_kernel() {}

int main()
{
    const size_t down_noise = 15080236, down_scene = 32724624, up_depth = 434176, up_col = 651264;
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    char *d = nullptr, *d2 = nullptr;
    CK(hipMalloc(&d, 64 << 20));
    CK(hipMalloc(&d2, 64 << 20));
    CK(hipMemset(d, 1, 64 << 20));
    char *pin = nullptr;
    CK(hipHostMalloc(&pin, 64 << 20, hipHostMallocDefault));
    memset(pin, 0, 64 << 20);
    unsigned *sink = nullptr;
    CK(hipMalloc(&sink, 4));
    char *pageable = (char *)aligned_alloc(4096, 16 << 20);
    memset(pageable, 3, 16 << 20);

    // ---- A: device -> host ---------------------------------------------------------------------------------------------------
    for (size_t n : {down_noise, down_scene}) {
        const size_t n16 = n / 16;
        double best = 1e9;
        for (int r = 0; r < 10; r++) {
            double t0 = now();
            CK(hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, sa));
            CK(hipStreamSynchronize(sa));
            best = std::min(best, now() - t0);
        }
        printf("A %9zu B  hipMemcpyAsync D2H -> pinned          : %7.1f us  %5.1f GB/s\n", n, best * 1e6, n / best / 1e9);
        for (int nt = 0; nt < 2; nt++)
            for (int wgs : {64, 256, 1024, 4096}) {
                float bestk = 1e9f;
                double bestw = 1e9;
                for (int r = 0; r < 8; r++) {
                    double t0 = now();
                    CK(hipEventRecord(e0, sa));
                    if (nt) store_kernel<1><<<wgs, 256, 0, sa>>>((uint4 *)pin, n16, r);
                    else store_kernel<0><<<wgs, 256, 0, sa>>>((uint4 *)pin, n16, r);
                    CK(hipEventRecord(e1, sa));
                    CK(hipStreamSynchronize(sa));
                    bestw = std::min(bestw, now() - t0);
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    bestk = std::min(bestk, ms);
                }
                // the host must see what the kernel wrote once the stream is idle
                const unsigned *w = (const unsigned *)pin;
                const size_t last = n16 - 1;
                const bool ok = w[0] == 0 && w[1] == 7 && w[last * 4] == (unsigned)last && w[last * 4 + 1] == 7u && w[last * 4 + 3] == 7u;
                printf("A %9zu B  kernel %s 16-B stores, %4d wgs : %7.1f us  %5.1f GB/s (wall %7.1f us)%s\n", n, nt ? "nt   " : "plain", wgs,
                       bestk * 1e3, n / (bestk * 1e-3) / 1e9, bestw * 1e6, ok ? "" : "  HOST DOES NOT SEE THE DATA");
            }
        for (int shift : {1, 2, 3}) {
            // the same stores, every wave's 1 KB starting 16 / 32 / 48 bytes past a 64-byte line (what a compaction's output looks like)
            float bestk = 1e9f;
            for (int r = 0; r < 8; r++) {
                CK(hipEventRecord(e0, sa));
                store_kernel<1><<<1024, 256, 0, sa>>>((uint4 *)pin + shift, n16 - 4, r);
                CK(hipEventRecord(e1, sa));
                CK(hipStreamSynchronize(sa));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                bestk = std::min(bestk, ms);
            }
            printf("A %9zu B  kernel nt    16-B stores, 1024 wgs, runs shifted by %2d B: %7.1f us  %5.1f GB/s\n", n, 16 * shift, bestk * 1e3,
                   n / (bestk * 1e-3) / 1e9);
        }
        float bestk = 1e9f;
        for (int r = 0; r < 8; r++) {
            CK(hipEventRecord(e0, sa));
            copy_kernel<<<1024, 256, 0, sa>>>((uint4 *)pin, (const uint4 *)d, n16);
            CK(hipEventRecord(e1, sa));
            CK(hipStreamSynchronize(sa));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            bestk = std::min(bestk, ms);
        }
        printf("A %9zu B  kernel copy HBM -> pinned, 1024 wgs    : %7.1f us  %5.1f GB/s\n", n, bestk * 1e3, n / (bestk * 1e-3) / 1e9);
    }

    // ---- B: pageable upload: time in the call / time until done -----------------------------------------------------------------
    for (size_t n : {up_depth, up_col, up_depth + up_col, 2 * (up_depth + up_col), 8 * up_depth, 8 * up_col, 8 * (up_depth + up_col)}) {
        double best_call = 1e9, best_done = 1e9;
        for (int r = 0; r < 20; r++) {
            double t0 = now();
            CK(hipMemcpyAsync(d2, pageable, n, hipMemcpyHostToDevice, sa));
            double t1 = now();
            CK(hipStreamSynchronize(sa));
            double t2 = now();
            best_call = std::min(best_call, t1 - t0);
            best_done = std::min(best_done, t2 - t0);
        }
        printf("B %9zu B  pageable H2D: call returns after %7.1f us, done after %7.1f us  %5.1f GB/s\n", n, best_call * 1e6, best_done * 1e6,
               n / best_done / 1e9);
    }
    for (size_t n : {up_depth, up_depth + up_col, 3 * up_depth, 3 * up_col}) {
        double best = 1e9;
        for (int r = 0; r < 20; r++) {
            double t0 = now();
            CK(hipMemcpyWithStream(d2, pageable, n, hipMemcpyHostToDevice, sa));
            best = std::min(best, now() - t0);
        }
        printf("B %9zu B  pageable hipMemcpyWithStream H2D: %7.1f us  %5.1f GB/s\n", n, best * 1e6, n / best / 1e9);
    }
    {
        // the same 8.7 MB as 16 copies back to back on one stream
        double best = 1e9;
        for (int r = 0; r < 10; r++) {
            double t0 = now();
            size_t off = 0;
            for (int s = 0; s < 8; s++) {
                CK(hipMemcpyAsync(d2 + off, pageable + off, up_depth, hipMemcpyHostToDevice, sa));
                off += up_depth;
                CK(hipMemcpyAsync(d2 + off, pageable + off, up_col, hipMemcpyHostToDevice, sa));
                off += up_col;
            }
            CK(hipStreamSynchronize(sa));
            best = std::min(best, now() - t0);
        }
        printf("B 16 pageable copies (8 x depth + colour frame)    : %7.1f us  %5.1f GB/s\n", best * 1e6, 8 * (up_depth + up_col) / best / 1e9);
    }

    // ---- C: both directions at once -------------------------------------------------------------------------------------------
    for (int mode = 0; mode < 3; mode++) {
        // mode 0: kernel nt stores down;  1: kernel copy HBM -> pinned down;  2: copy engine down
        const size_t n = down_noise, n16 = n / 16;
        double best = 1e9, best_up = 1e9;
        for (int r = 0; r < 10; r++) {
            double t0 = now();
            if (mode == 0) store_kernel<1><<<1024, 256, 0, sb>>>((uint4 *)pin, n16, r);
            else if (mode == 1) copy_kernel<<<1024, 256, 0, sb>>>((uint4 *)pin, (const uint4 *)d, n16);
            else CK(hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, sb));
            size_t off = 0;
            for (int s = 0; s < 8; s++) {
                CK(hipMemcpyAsync(d2 + off, pageable + off, up_depth + up_col, hipMemcpyHostToDevice, sa));
                off += up_depth + up_col;
            }
            CK(hipStreamSynchronize(sa));
            double t1 = now();
            CK(hipStreamSynchronize(sb));
            double t2 = now();
            best = std::min(best, t2 - t0);
            best_up = std::min(best_up, t1 - t0);
        }
        printf("C duplex, 15.1 MB down by %-22s + 8.7 MB pageable up: up done %7.1f us, all done %7.1f us\n",
               mode == 0 ? "kernel nt stores" : mode == 1 ? "kernel copy HBM->pinned" : "copy engine", best_up * 1e6, best * 1e6);
    }

    // ---- D: launches ----------------------------------------------------------------------------------------------------------
    {
        double best_host = 1e9, best_all = 1e9;
        for (int r = 0; r < 10; r++) {
            double t0 = now();
            for (int i = 0; i < 100; i++) empty_kernel<<<1, 64, 0, sa>>>();
            double t1 = now();
            CK(hipStreamSynchronize(sa));
            double t2 = now();
            best_host = std::min(best_host, t1 - t0);
            best_all = std::min(best_all, t2 - t0);
        }
        printf("D 100 empty launches on one stream: host %6.2f us each, drained after %6.2f us each\n", best_host * 1e4, best_all * 1e4);
        // launch + event record + cross-stream wait, the unit a pipelined call pays per hand-over
        double best = 1e9;
        for (int r = 0; r < 10; r++) {
            double t0 = now();
            for (int i = 0; i < 50; i++) {
                empty_kernel<<<1, 64, 0, sa>>>();
                CK(hipEventRecord(e0, sa));
                CK(hipStreamWaitEvent(sb, e0, 0));
                empty_kernel<<<1, 64, 0, sb>>>();
            }
            double t1 = now();
            CK(hipStreamSynchronize(sa));
            CK(hipStreamSynchronize(sb));
            best = std::min(best, t1 - t0);
        }
        printf("D launch + record + wait + launch (two streams): host %6.2f us per hand-over\n", best * 1e6 / 50);
    }

    // ---- F: does a small kernel on another stream run WHILE a kernel streams to host memory? -------------------------------------
    {
        hipEvent_t ea, eb, ec, e_nosys;
        CK(hipEventCreate(&ea));
        CK(hipEventCreate(&eb));
        CK(hipEventCreateWithFlags(&ec, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&e_nosys, hipEventDisableTiming | hipEventDisableSystemFence));
        const size_t n16 = down_noise / 16;
        for (int variant = 0; variant < 5; variant++) {
            // 0: tiny kernel alone on sb while sa streams;  1: tiny kernel behind a wait for an (already complete) event of sa;
            // 2: the same with an event created with hipEventDisableSystemFence;  3: tiny kernel that stores 64 B to HOST memory;
            // 4: three tiny kernels back to back on sb
            double best_small = 1e9, best_all = 1e9;
            for (int r = 0; r < 6; r++) {
                CK(hipStreamSynchronize(sa));
                CK(hipStreamSynchronize(sb));
                if (variant == 1) { empty_kernel<<<1, 64, 0, sa>>>(); CK(hipEventRecord(ec, sa)); }
                if (variant == 2) { empty_kernel<<<1, 64, 0, sa>>>(); CK(hipEventRecord(e_nosys, sa)); }
                double t0 = now();
                store_kernel<1><<<1024, 256, 0, sa>>>((uint4 *)pin, n16, r);
                // give the big kernel a head start so that it is streaming when the small one is dispatched
                while (now() - t0 < 40e-6) {}
                if (variant == 1) CK(hipStreamWaitEvent(sb, ec, 0));
                if (variant == 2) CK(hipStreamWaitEvent(sb, e_nosys, 0));
                if (variant == 3) store_kernel<0><<<1, 4, 0, sb>>>((uint4 *)(pin + (48 << 20)), 4, r);
                else load_kernel<<<4, 256, 0, sb>>>((const uint4 *)d, 4096, sink);
                if (variant == 4) { load_kernel<<<4, 256, 0, sb>>>((const uint4 *)d, 4096, sink); load_kernel<<<4, 256, 0, sb>>>((const uint4 *)d, 4096, sink); }
                CK(hipStreamSynchronize(sb));
                double t1 = now();
                CK(hipStreamSynchronize(sa));
                double t2 = now();
                best_small = std::min(best_small, t1 - t0);
                best_all = std::min(best_all, t2 - t0);
            }
            const char *names[] = {"tiny kernel", "tiny kernel behind a stream-wait (default event)", "tiny kernel behind a stream-wait (no-system-fence event)",
                                   "tiny kernel storing 64 B to host", "three tiny kernels"};
            printf("F while 15.1 MB stream to the host on another stream: %-58s done after %7.1f us (big kernel: %7.1f us)\n", names[variant],
                   best_small * 1e6, best_all * 1e6);
        }
        // G: the 15.1 MB as two kernels on two streams at once
        double best = 1e9;
        for (int r = 0; r < 6; r++) {
            CK(hipStreamSynchronize(sa));
            CK(hipStreamSynchronize(sb));
            double t0 = now();
            store_kernel<1><<<512, 256, 0, sa>>>((uint4 *)pin, n16 / 2, r);
            store_kernel<1><<<512, 256, 0, sb>>>((uint4 *)pin + n16 / 2, n16 / 2, r);
            CK(hipStreamSynchronize(sa));
            CK(hipStreamSynchronize(sb));
            best = std::min(best, now() - t0);
        }
        printf("G 15.1 MB as two store kernels on two streams at once: %7.1f us\n", best * 1e6);
        // H: four store kernels of 3.77 MB back to back on ONE stream (the bubbles between dependent launches)
        best = 1e9;
        for (int r = 0; r < 6; r++) {
            CK(hipStreamSynchronize(sa));
            double t0 = now();
            for (int q = 0; q < 4; q++) store_kernel<1><<<512, 256, 0, sa>>>((uint4 *)pin + q * (n16 / 4), n16 / 4, r);
            CK(hipStreamSynchronize(sa));
            best = std::min(best, now() - t0);
        }
        printf("H 15.1 MB as four store kernels back to back on one stream: %7.1f us\n", best * 1e6);
    }

    // ---- E: zero-copy upload from a registered caller array --------------------------------------------------------------------
    {
        const size_t n = 8 * (up_depth + up_col), n16 = n / 16;
        double t0 = now();
        hipError_t e = hipHostRegister(pageable, n, hipHostRegisterDefault);
        double t_reg = now() - t0;
        if (e != hipSuccess) {
            printf("E hipHostRegister failed: %s\n", hipGetErrorString(e));
        } else {
            void *dp = nullptr;
            CK(hipHostGetDevicePointer(&dp, pageable, 0));
            float bestk = 1e9f;
            for (int r = 0; r < 8; r++) {
                CK(hipEventRecord(e0, sa));
                load_kernel<<<1024, 256, 0, sa>>>((const uint4 *)dp, n16, sink);
                CK(hipEventRecord(e1, sa));
                CK(hipStreamSynchronize(sa));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                bestk = std::min(bestk, ms);
            }
            double best = 1e9;
            for (int r = 0; r < 10; r++) {
                double t1 = now();
                CK(hipMemcpyAsync(d2, pageable, n, hipMemcpyHostToDevice, sa));
                CK(hipStreamSynchronize(sa));
                best = std::min(best, now() - t1);
            }
            t0 = now();
            CK(hipHostUnregister(pageable));
            double t_unreg = now() - t0;
            printf("E register 8.7 MB: %7.1f us, unregister %7.1f us; kernel loads from it: %7.1f us  %5.1f GB/s; memcpy from it: %7.1f us  %5.1f GB/s\n",
                   t_reg * 1e6, t_unreg * 1e6, bestk * 1e3, n / (bestk * 1e-3) / 1e9, best * 1e6, n / best / 1e9);
        }
    }
    return 0;
}
