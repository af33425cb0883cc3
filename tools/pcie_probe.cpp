// Dev helper: host<->device copy rates for the sizes of one merge call, from pageable / registered / hipHostMalloc memory.
//   hipcc -O2 -o /tmp/pcie_probe tools/pcie_probe.cpp && /tmp/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t sizes[] = {434176 * 8, 651264 * 8, 15080236, 32724624};
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    void *d;
    hipMalloc(&d, 64 << 20);
    for (size_t n : sizes) {
        void *pageable = aligned_alloc(4096, (n + 4095) & ~4095ull);
        memset(pageable, 1, n);
        void *reg = aligned_alloc(4096, (n + 4095) & ~4095ull);
        memset(reg, 1, n);
        hipError_t e = hipHostRegister(reg, n, hipHostRegisterDefault);
        void *pin;
        hipHostMalloc(&pin, n, hipHostMallocDefault);
        memset(pin, 1, n);
        struct { const char *name; void *p; } src[] = {{"pageable", pageable}, {"registered", reg}, {"hipHostMalloc", pin}};
        for (auto &k : src) {
            for (int dir = 0; dir < 2; dir++) {
                double best = 1e9;
                for (int r = 0; r < 20; r++) {
                    double t0 = now();
                    if (dir == 0) hipMemcpyAsync(d, k.p, n, hipMemcpyHostToDevice, s);
                    else hipMemcpyAsync(k.p, d, n, hipMemcpyDeviceToHost, s);
                    hipStreamSynchronize(s);
                    double t = now() - t0;
                    if (t < best) best = t;
                }
                printf("%9zu B %-14s %s: %7.1f us  %6.1f GB/s%s\n", n, k.name, dir ? "D2H" : "H2D", best * 1e6, n / best / 1e9,
                       (k.p == reg && e != hipSuccess) ? "  (register FAILED)" : "");
            }
        }
        hipHostUnregister(reg);
        hipHostFree(pin);
        free(pageable);
        free(reg);
    }
    return 0;
}
