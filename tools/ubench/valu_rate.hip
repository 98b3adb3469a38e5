// issue cost of a few VALU instructions on gfx950: cycles per instruction of a long unrolled run of independent instructions,
// one wave per workgroup (so nothing else shares the SIMD), timed with s_memtime.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
template <int OP>
__global__ void k(uint64_t *out, uint32_t seed)
{
    uint32_t a[8]; uint64_t b[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 7 + i; b[i] = ((uint64_t)a[i] << 20) | i; }
    const uint32_t sh = (seed & 7) + 1;
    const uint64_t msk = 0x5555555555555555ull ^ seed;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(sh));
            if (OP == 1) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(b[i]) : "v"(sh));
            if (OP == 2) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(b[i]) : "v"(sh));
            if (OP == 3) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(a[i]) : "v"(sh));
            if (OP == 4) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(a[i]) : "v"(sh));
            if (OP == 5) asm volatile("v_bfe_u32 %0, %0, %1, 8" : "+v"(a[i]) : "v"(sh));
            if (OP == 6) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(sh));
            if (OP == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(sh));
            if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(sh));
            if (OP == 9) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(b[i]));
            if (OP == 10) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            if (OP == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(sh) : "vcc");
            if (OP == 12) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a[i]));
            if (OP == 13) asm volatile("v_bcnt_u32_b32 %0, %0, 0" : "+v"(a[i]));
            if (OP == 14) asm volatile("v_sad_u8 %0, %0, %1, 0" : "+v"(a[i]) : "v"(sh));
            if (OP == 15) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(b[i]) : "v"(sh) : "vcc");
            if (OP == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(sh), "s"(msk));
            if (OP == 17) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
            if (OP == 18) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(sh) : "vcc");
            if (OP == 19) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(sh), "v"(a[(i + 1) & 7]));
            if (OP == 20) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(sh), "v"(a[(i + 1) & 7]));
            if (OP == 21) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(sh), "v"(a[(i + 1) & 7]));
            if (OP == 22) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(sh), "v"(a[(i + 1) & 7]));
            if (OP == 23) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(sh));
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + b[i];
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s; }
}
template <int OP> static void run(const char *name, uint64_t *d)
{
    uint64_t h[2];
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, d, 3u);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, d, 3u);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-18s %6.2f clock ticks per instruction (wave64, 8 independent chains)\n", name, (double)h[0] / (REP * 8));
}
int main()
{
    uint64_t *d; hipMalloc(&d, 64);
    run<0>("v_add_u32", d); run<1>("v_lshlrev_b64", d); run<2>("v_lshrrev_b64", d); run<3>("v_alignbit_b32", d); run<4>("v_perm_b32", d);
    run<5>("v_bfe_u32", d); run<6>("v_pk_add_u16", d); run<7>("v_mul_u32_u24", d); run<8>("v_mul_lo_u32", d); run<9>("v_lshl_add_u64", d);
    run<10>("v_mov_b32_dpp", d); run<11>("v_cndmask_b32", d); run<12>("v_ffbh_u32", d); run<13>("v_bcnt_u32_b32", d); run<14>("v_sad_u8", d);
    run<15>("v_mad_u64_u32", d); run<16>("v_cndmask_b32 sgpr", d); run<17>("v_add_u32 sgpr op", d); run<18>("v_cmp_lt_u32", d);
    run<19>("v_bfi_b32", d); run<20>("v_and_or_b32", d); run<21>("v_lshl_or_b32", d); run<22>("v_add3_u32", d); run<23>("v_min_u32", d);
    return 0;
}
