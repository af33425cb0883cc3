// int_rate.hip -- issue rate of the integer VALU ops the triangulation count pass is made of: v_add_u32, v_sad_u32, v_mad_u32_u24,
// v_mul_hi_u32 (division by a constant), v_mul_hi_u32_u24, v_mul_lo_u32.  16 independent chains per lane, 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o int_rate int_rate.hip; prints cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned int *out, int iters, unsigned int seed)
{
    unsigned int v[16];
    for (int i = 0; i < 16; i++) v[i] = seed + i * 977u + threadIdx.x;
    const unsigned int c = seed | 1u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 1) asm volatile("v_sad_u32 %0, %0, %1, 0" : "+v"(v[i]) : "v"(c));
            if (OP == 2) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 3) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 4) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 6) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c));
            if (OP == 8) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(v[i]) : "v"(c));
            if (OP == 9) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %1" : : "v"(v[i]), "v"(c) : "s10", "s11");
            if (OP == 10) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(v[i]), "v"(c) : "vcc");
            if (OP == 11) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 12) asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 13) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 14) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(v[i]) : "v"(c) : "s10", "s11");
            if (OP == 15) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 16) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c) : "vcc");
            if (OP == 18 && (i & 3) == 0) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %4\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]) : "v"(c) : "vcc");
            if (OP == 19 && (i & 3) == 0) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %4\n v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]) : "v"(c) : "s10", "s11");
            if (OP == 20 && (i & 3) == 0) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %4\n v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]) : "v"(c) : "vcc");
            if (OP == 21 && (i & 3) == 0) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %4\n v_add_u32 %1, %1, %4\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]) : "v"(c) : "vcc");
            if (OP == 22 && (i & 3) == 0) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %4\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]) : "v"(c) : "vcc");
            if (OP == 17) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c) : "vcc");
        }
    }
    unsigned int r = 0;
    for (int i = 0; i < 16; i++) r ^= v[i];
    if (r == 0x12345678u) out[0] = r;
}

template <int OP>
static void run(const char *name, unsigned int *d)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * 4;
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: 4 waves x iters x 16 instructions
    const double insts = 4.0 * iters * 16;
    printf("%-18s %.3f ms  ->  %.2f ns per wave-instruction per SIMD (4.0 cycles at 2.4 GHz = 1.67 ns)\n", name, best, best * 1e6 / insts);
}

int main()
{
    unsigned int *d;
    (void)hipMalloc(&d, 4);
    run<0>("v_add_u32", d);
    run<1>("v_sad_u32", d);
    run<2>("v_mad_u32_u24", d);
    run<3>("v_mul_hi_u32", d);
    run<4>("v_mul_hi_u32_u24", d);
    run<5>("v_mul_lo_u32", d);
    run<6>("v_min3_u32", d);
    run<7>("v_cndmask_b32 vcc", d);
    run<8>("v_cndmask_b32 sgpr", d);
    run<9>("v_cmp_lt_u32 sgpr", d);
    run<10>("v_cmp_lt_u32 vcc", d);
    run<11>("v_and_b32", d);
    run<12>("v_bfi_b32", d);
    run<13>("v_lshl_add_u32", d);
    run<14>("cmp+cndmask pair", d);
    run<15>("v_sub_u32", d);
    run<16>("cmp+cndmask vcc", d);
    run<17>("cndmask vcc (clob)", d);
    run<18>("cmp+4cnd vcc (x5/4)", d);
    run<19>("cmp+4cnd sgpr(x5/4)", d);
    run<20>("cmp+4cnd_e64 vcc", d);
    run<21>("cmp,add,cnd,add,add", d);
    run<22>("cmp,cnd,add,add,add", d);
    return 0;
}
