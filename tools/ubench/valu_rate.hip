// Micro-benchmark: issue rate of the integer VALU ops the DP kernel is made of (gfx950), inline asm so
// nothing is folded.  8 independent dependency chains per lane, 8 waves/SIMD resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a[8], b = seed ^ threadIdx.x, c = 0x05040100u;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i * 77 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 1) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 2) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 4) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 5) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 6) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 7) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 8) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 9) asm volatile("v_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 10) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 11) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 12) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 13) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 14) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 15) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 16) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 17) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 18) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 19) asm volatile("v_add_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 20) asm volatile("v_min_i16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 21) asm volatile("v_min_i16_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 22) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 23) asm volatile("v_add_u16_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 24) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c));
                // mixes (the DP's add -> min pattern); counted as ONE op of the loop below, i.e. halve the printed cycles
                if (OP == 25) asm volatile("v_add_u32 %0, %0, %1\n\tv_pk_min_u16 %0, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 26) asm volatile("v_pk_add_u16 %0, %0, %1\n\tv_pk_min_u16 %0, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 27) asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                // the DP's  min(x + GE, y + gop)  shape: two adds then one packed min (THREE instructions per counted op)
                if (OP == 28) asm volatile("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %3\n\tv_pk_min_u16 %0, %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]) : "v"(b), "v"(c));
                if (OP == 29) asm volatile("v_pk_add_u16 %0, %0, %2\n\tv_pk_add_u16 %1, %1, %3\n\tv_pk_min_u16 %0, %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]) : "v"(b), "v"(c));
                // four adds then two mins (SIX instructions per counted op)
                if (OP == 30) asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %5\n\tv_pk_min_u16 %0, %0, %1\n\tv_pk_min_u16 %2, %2, %3" : "+v"(a[i]), "+v"(a[(i + 1) & 7]), "+v"(a[(i + 2) & 7]), "+v"(a[(i + 3) & 7]) : "v"(b), "v"(c));
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, uint32_t* d, int blocks)
{
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 7);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 256 * iters * 16 * 8 / 64;
    printf("%-22s %8.3f ms  %8.1f G wave-instr/s   %.2f cycles per wave-instr per SIMD (if 2.4 GHz, 1024 SIMDs)\n", name, ms,
           wave_instr / ms / 1e6, 2.4e9 * 1024 / (wave_instr / (ms * 1e-3)));
}
int main()
{
    uint32_t* d; const int blocks = 256 * 8;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    run<0>("v_pk_add_u16", d, blocks);
    run<1>("v_pk_min_i16", d, blocks);
    run<2>("v_pk_min_u16", d, blocks);
    run<3>("v_add_u32", d, blocks);
    run<4>("v_min_i32", d, blocks);
    run<5>("v_alignbit_b32", d, blocks);
    run<6>("v_perm_b32", d, blocks);
    run<7>("v_xor_b32", d, blocks);
    run<8>("v_min3_i32", d, blocks);
    run<9>("v_min_i16", d, blocks);
    run<10>("v_add_u16", d, blocks);
    run<11>("v_lshl_or_b32", d, blocks);
    run<12>("v_and_or_b32", d, blocks);
    run<13>("v_pk_add_i16_2indep", d, blocks);
    run<14>("v_pk_mad_u16", d, blocks);
    run<15>("v_pk_sub_u16", d, blocks);
    run<16>("v_pk_max_i16", d, blocks);
    run<17>("v_mov_b32_dpp_shr1", d, blocks);
    run<18>("v_add_u32_dpp", d, blocks);
    run<19>("v_add_u16_sdwa_w1", d, blocks);
    run<20>("v_min_i16_sdwa_w1", d, blocks);
    run<21>("v_min_i16_sdwa_w0", d, blocks);
    run<22>("v_min_u16", d, blocks);
    run<23>("v_add_u16_sdwa_pad", d, blocks);
    run<24>("v_cndmask_b32", d, blocks);
    for (int occ = 8; occ >= 4; occ >>= 1) {                   // mixes at 8 / 4 / 2 waves per SIMD (2 instructions per counted op)
        printf("# %d waves/SIMD\n", occ);
        run<25>("add_u32+pk_min_u16", d, 256 * occ);
        run<26>("pk_add_u16+pk_min_u16", d, 256 * occ);
        run<27>("add_u32+add_u32", d, 256 * occ);
        run<28>("add32,add32,pkmin (x3)", d, 256 * occ);
        run<29>("pkadd,pkadd,pkmin (x3)", d, 256 * occ);
        run<30>("4xadd32,2xpkmin (x6)", d, 256 * occ);
    }
    return 0;
}
