// Micro-benchmark (not product code): issue cost of the instructions the decoder's serial roles lean on.
// One workgroup per launch; W wavefronts per SIMD run the same dependent / independent chains.
//   build: hipcc --offload-arch=gfx950 -O3 -o valu_micro valu_micro.hip ; run: ./valu_micro
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// KIND: 0 v_add_u32 dependent, 1 v_mul_lo_u32 dependent, 2 v_mad_u64_u32 dependent, 3 v_lshrrev_b64 dependent,
//       4 v_mul_lo_u32 independent x4, 5 v_add_u32 independent x4, 6 s_mul_i32 dependent, 7 s_lshr_b64 dependent,
//       8 ds_read_b64 dependent (pointer chase), 9 v_mul_u32_u24 dependent, 10 v_alignbit dependent,
//       11 v_readfirstlane -> s_add -> v_mov round trip, 12 ds_bpermute dependent
template <int KIND>
__global__ __launch_bounds__(1024) void k(uint32_t seed, uint64_t* out, uint64_t* cyc, int iters) {
    __shared__ uint64_t tab[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) tab[i] = ((uint64_t)((i * 8 + 8) & 4095)) | ((uint64_t)i << 32);
    __syncthreads();
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 0x55, d = seed + 77;
    uint64_t q = ((uint64_t)seed << 32) | threadIdx.x;
    uint32_t sa = seed | 1;
    uint64_t sq = ((uint64_t)seed << 20) | 12345;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (KIND == 1) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (KIND == 2) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(a), "v"(b) : "vcc");) }
        if (KIND == 3) { REP64(asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q) : "v"(b));) }
        if (KIND == 4) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (KIND == 5) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (KIND == 6) { REP64(asm volatile("s_mul_i32 %0, %0, %0" : "+s"(sa));) }
        if (KIND == 7) { REP64(asm volatile("s_lshr_b64 %0, %0, 1\n s_or_b64 %0, %0, 0x40000000" : "+s"(sq) : : "scc");) }
        if (KIND == 8) { REP64({ uint64_t e; __builtin_memcpy(&e, (const uint8_t*)tab + a, 8); a = (uint32_t)e; asm volatile("" : "+v"(a)); }) }
        if (KIND == 9) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (KIND == 10) { REP64(asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a) : "v"(b));) }
        if (KIND == 11) { REP64({ uint32_t s = __builtin_amdgcn_readfirstlane(a); asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc"); a = s; asm volatile("" : "+v"(a)); }) }
        if (KIND == 12) { REP64({ a = __builtin_amdgcn_ds_bpermute((int)(a & 0xFC), (int)a); asm volatile("" : "+v"(a)); }) }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (KIND == 8) a &= 4095;
    out[blockIdx.x * 1024 + threadIdx.x] = a + b + c + d + q + sa + sq;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_iter, uint64_t* d_out, uint64_t* d_cyc) {
    for (int waves : {1, 4, 8, 16}) { // waves in the workgroup: 1 => one SIMD, 4 => one per SIMD, 8 => two per SIMD, 16 => four per SIMD
        const int iters = 200;
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64 * waves), 0, 0, 8u, d_out, d_cyc, iters);
        hipDeviceSynchronize();
        uint64_t c = 0;
        hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
        printf("%-34s waves/WG %2d : %7.2f cycles per instruction (wave 0's clock)\n", name, waves, (double)c / ((double)iters * 64 * per_iter));
    }
}

int main() {
    uint64_t *d_out, *d_cyc;
    hipMalloc(&d_out, 1024 * 8 * 4); hipMalloc(&d_cyc, 64);
    run<0>("v_add_u32 dependent", 1, d_out, d_cyc);
    run<5>("v_add_u32 4 independent", 4, d_out, d_cyc);
    run<1>("v_mul_lo_u32 dependent", 1, d_out, d_cyc);
    run<4>("v_mul_lo_u32 4 independent", 4, d_out, d_cyc);
    run<2>("v_mad_u64_u32 dependent", 1, d_out, d_cyc);
    run<9>("v_mul_u32_u24 dependent", 1, d_out, d_cyc);
    run<3>("v_lshrrev_b64 dependent", 1, d_out, d_cyc);
    run<10>("v_alignbit_b32 dependent", 1, d_out, d_cyc);
    run<6>("s_mul_i32 dependent", 1, d_out, d_cyc);
    run<7>("s_lshr_b64+s_add_u32 dependent", 2, d_out, d_cyc);
    run<11>("readfirstlane+s_add+v_mov trip", 1, d_out, d_cyc);
    run<8>("ds_read_b64 pointer chase", 1, d_out, d_cyc);
    run<12>("ds_bpermute dependent", 1, d_out, d_cyc);
    return 0;
}
