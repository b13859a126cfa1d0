// Micro-benchmark (not product code): does a VALU instruction of a wavefront with few active lanes
// occupy the SIMD for fewer cycles?  16 wavefronts (4 per SIMD) run independent v_add chains with
// 64 / 16 / 1 active lanes; prints the cycles the whole workgroup needs per instruction per wavefront.
//   build: hipcc --offload-arch=gfx950 -O3 -o exec_micro exec_micro.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
__global__ __launch_bounds__(1024) void k(uint32_t seed, int active, uint64_t* out, uint64_t* cyc, int iters, uint64_t pattern) {
    __shared__ unsigned long long tmin, tmax;
    if (threadIdx.x == 0) { tmin = ~0ull; tmax = 0; }
    __syncthreads();
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 0x55, d = seed + 77;
    uint64_t t0 = __builtin_readcyclecounter();
    if (pattern ? ((pattern >> (threadIdx.x & 63)) & 1) != 0 : (int)(threadIdx.x & 63) < active) {
        for (int it = 0; it < iters; it++) {
            REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));)
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    atomicMin(&tmin, (unsigned long long)t0); atomicMax(&tmax, (unsigned long long)t1);
    __syncthreads();
    out[threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) cyc[0] = tmax - tmin;
}
int main() {
    uint64_t *d_out, *d_cyc;
    if (hipMalloc(&d_out, 1024 * 8) != hipSuccess || hipMalloc(&d_cyc, 64) != hipSuccess) return 1;
    const int iters = 100;
    for (int waves : {4, 16})
        for (int active : {64, 32, 16, 8, 1}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, 8u, active, d_out, d_cyc, iters, (uint64_t)0);
            if (hipDeviceSynchronize() != hipSuccess) return 1;
            uint64_t c = 0;
            if (hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
            printf("waves/WG %2d (=%d per SIMD), %2d active lanes: %6.2f cycles per VALU instruction per wavefront, %5.2f SIMD cycles per instruction\n",
                   waves, waves / 4, active, (double)c / (iters * 64.0 * 4), (double)c / (iters * 64.0 * 4) / (waves / 4));
        }
    // the same with the active lanes spread over the four 16-lane passes of the wavefront
    struct { const char* what; uint64_t m; } pats[] = {{"lanes 0,16,32,48", 0x0001000100010001ull}, {"lanes 0-3 of every 16", 0x000F000F000F000Full}, {"lanes 0-7", 0xFFull}, {"lanes 0-3", 0xFull}, {"lanes 0-15", 0xFFFFull}, {"lanes 0-7 and 32-39", 0x000000FF000000FFull}, {"one lane in every 8", 0x0101010101010101ull}, {"lanes 0-1 of every 8", 0x0303030303030303ull}, {"lanes 0-3 of every 8", 0x0F0F0F0F0F0F0F0Full}, {"one lane in every 4", 0x1111111111111111ull}, {"one lane in every 32", 0x0000000100000001ull}, {"lanes 0-3 of every 32", 0x0000000F0000000Full}, {"lane 0 alone", 0x1ull}};
    for (int waves : {4, 8, 16})
        for (auto& p : pats) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, 8u, 0, d_out, d_cyc, iters, p.m);
            if (hipDeviceSynchronize() != hipSuccess) return 1;
            uint64_t c = 0;
            if (hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
            printf("waves/WG %2d (=%d per SIMD), %-22s: %6.2f cycles per VALU instruction per wavefront\n", waves, waves / 4, p.what, (double)c / (iters * 64.0 * 4));
        }
    return 0;
}
