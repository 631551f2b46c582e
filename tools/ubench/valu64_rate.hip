// Micro-benchmark: issue rate of the bit-manipulation ops the seeding kernel is made of (gfx950): 64-bit shifts vs their
// 32-bit replacements, population count, find-first-bit.  8 independent chains per lane, 8 waves/SIMD, inline asm.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed)
{
    uint64_t a[8];
    uint32_t b = (seed ^ threadIdx.x) & 63u, c = 0x05040100u;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = ((uint64_t)(seed + i * 77 + threadIdx.x) << 32) | (seed * 31 + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint32_t& lo = ((uint32_t*)&a[i])[0];
                uint32_t& hi = ((uint32_t*)&a[i])[1];
                if (OP == 0) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(a[i]));
                if (OP == 3) asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(lo) : "v"(hi), "v"(b));
                if (OP == 4) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(lo) : "v"(c));
                if (OP == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo) : "v"(c));
                if (OP == 6) asm volatile("v_or_b32 %0, %0, %1" : "+v"(lo) : "v"(c));
                if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(c));
                if (OP == 8) asm volatile("v_ffbl_b32 %0, %0" : "+v"(lo));
                if (OP == 9) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(lo) : "v"(b));
                if (OP == 10) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(lo) : "v"(b));
                if (OP == 11) asm volatile("v_not_b32 %0, %0" : "+v"(lo));
                if (OP == 12) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(c), "v"(b));
                if (OP == 13) asm volatile("v_bfe_u32 %0, %0, %1, 7" : "+v"(lo) : "v"(b));
                if (OP == 14) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(lo) : "v"(c));
                if (OP == 15) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(c), "v"(b));
                if (OP == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(lo) : "v"(c) : "s20", "s21");
                if (OP == 17) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(c) : "vcc");
                if (OP == 18) asm volatile("v_max_u32 %0, %0, %1" : "+v"(lo) : "v"(c));
                if (OP == 19) asm volatile("v_min_u32 %0, %0, %1" : "+v"(lo) : "v"(c));
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <int OP> void run(const char* name, uint32_t* d, int blocks)
{
    const int iters = 1000;
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
    run<0>("v_lshlrev_b64 (vgpr)", d, blocks);
    run<1>("v_lshrrev_b64 (vgpr)", d, blocks);
    run<2>("v_lshlrev_b64 (const)", d, blocks);
    run<3>("v_alignbit_b32 (vgpr)", d, blocks);
    run<4>("v_bcnt_u32_b32", d, blocks);
    run<5>("v_and_b32", d, blocks);
    run<6>("v_or_b32", d, blocks);
    run<7>("v_cndmask_b32", d, blocks);
    run<8>("v_ffbl_b32", d, blocks);
    run<9>("v_lshlrev_b32", d, blocks);
    run<10>("v_lshrrev_b32", d, blocks);
    run<11>("v_not_b32", d, blocks);
    run<12>("v_and_or_b32", d, blocks);
    run<13>("v_bfe_u32", d, blocks);
    run<14>("v_xor_b32", d, blocks);
    run<15>("v_or3_b32", d, blocks);
    run<16>("v_cndmask_b32 (sgpr pair)", d, blocks);
    run<17>("v_cmp + v_cndmask (2 instr)", d, blocks);
    run<18>("v_max_u32", d, blocks);
    run<19>("v_min_u32", d, blocks);
    return 0;
}
