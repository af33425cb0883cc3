// pk_rate.hip -- issue rate of the packed-f32 VALU ops the brute-force NN kernel is made of (no FMA: contraction is off by
// contract).  Each wave runs a long chain of v_pk_add_f32 / v_pk_mul_f32 (or their scalar forms) on 8 independent
// accumulator pairs; waves per SIMD = 1, 2, 4.  Prints cycles per instruction per wave and the chip-wide flop rate.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pk_rate pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2v __attribute__((ext_vector_type(2)));

template <int PACKED>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, float seed)
{
    f2v a[8];
    float s[16];
    for (int i = 0; i < 8; i++) a[i] = f2v{seed + i, seed - i};
    for (int i = 0; i < 16; i++) s[i] = seed + i;
    const f2v q = {seed * 0.5f, seed * 0.25f};
    for (int it = 0; it < iters; it++) {
        if (PACKED) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                a[i] = q - a[i];
                a[i] = a[i] * a[i];
                a[i] = a[i] + q;
                a[i] = a[i] * q;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                s[i] = q.x - s[i];
                s[i] = s[i] * s[i];
                s[i] = s[i] + q.x;
                s[i] = s[i] * q.y;
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += a[i].x + a[i].y;
    for (int i = 0; i < 16; i++) r += s[i];
    if (r == 12345.678f) out[0] = r;
}

int main()
{
    float *d;
    (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int packed = 0; packed < 2; packed++)
        for (int wps : {1, 2, 4}) {
            const int blocks = 256 * wps;   // 4 waves per block = one per SIMD of a CU
            for (int rep = 0; rep < 2; rep++) {
                (void)hipEventRecord(e0);
                if (packed) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
                else hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
            }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_wave = (double)iters * (packed ? 32 : 64);
            const double flops = (double)blocks * 4 * 64 * iters * 64;
            printf("%s waves/SIMD %d: %.3f ms, %.2f ns per instruction per SIMD, %.1f TFLOP/s\n", packed ? "v_pk_*_f32" : "v_*_f32   ", wps, ms,
                   ms * 1e6 / (instr_per_wave * wps), flops / ms * 1e-9);
        }
    return 0;
}
