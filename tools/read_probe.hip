// read_probe.hip -- what a read-only streaming kernel reaches on this box (the ceiling of the count pass, which reads 2 B per pixel and writes
// one int per 2048): every lane sums 16-byte loads, K of them in flight, one int per workgroup out.  hipcc --offload-arch=gfx950 -O3 -o read_probe tools/read_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K>
__global__ __launch_bounds__(256) void read_kernel(const uint4 *src, size_t n16, unsigned int *out)
{
    // a workgroup owns K consecutive chunks of 256 x 16 B; consecutive lanes read consecutive 16-byte words
    size_t base = (size_t)blockIdx.x * 256 * K + threadIdx.x;
    uint4 v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = base + (size_t)k * 256 < n16 ? src[base + (size_t)k * 256] : make_uint4(0, 0, 0, 0);
    unsigned int s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.x], s);
}

template <int K>
static void run(const uint4 *src, size_t bytes, unsigned int *out)
{
    const size_t n16 = bytes / 16;
    const unsigned grid = (unsigned)((n16 + 256 * K - 1) / (256 * K));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(read_kernel<K>, dim3(grid), dim3(256), 0, 0, src, n16, out);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(read_kernel<K>, dim3(grid), dim3(256), 0, 0, src, n16, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("  %4zu MB, %2d x 16 B in flight per lane, %7u workgroups: %7.1f us per pass -> %.2f TB/s\n", bytes >> 20, K, grid, best * 100.0f, bytes / (best * 1e-4) / 1e12);
}

int main()
{
    const size_t max_bytes = (size_t)2 << 30;
    uint4 *src;
    unsigned int *out;
    hipMalloc(&src, max_bytes);
    hipMalloc(&out, 64 << 20);
    hipMemset(src, 1, max_bytes);
    hipMemset(out, 0, 64 << 20);
    // 222 MB = the depth of the headline step (fits the 256 MB Infinity Cache when it is read again and again: see the 2 GB rows for HBM)
    for (size_t bytes : {(size_t)222 << 20, (size_t)1 << 30, (size_t)2 << 30}) {
        run<1>(src, bytes, out);
        run<4>(src, bytes, out);
        run<8>(src, bytes, out);
        run<16>(src, bytes, out);
    }
    return 0;
}
