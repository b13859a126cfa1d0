// Micro-benchmark: cost of per-lane LDS accesses of different widths / alignments for ONE wavefront (the lane-per-sequence copies
// of mzd_lds.hip).  Prints cycles per wave-instruction.   hipcc --offload-arch=gfx950 -O3 -o lds_unaligned_micro lds_unaligned_micro.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
template <int MODE>
__global__ void k(uint64_t* out, const uint32_t* offs, int iters) {
    const int lane = threadIdx.x;
    for (int i = lane; i < 16384 / 4; i += 64) ((uint32_t*)lds)[i] = i;
    __syncthreads();
    uint32_t a = offs[lane];
    uint64_t acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t ad = (a + 64 * u) & 16383 & ~(MODE == 0 || MODE == 3 || MODE == 6 ? 7u : 0u);
            if (MODE == 0 || MODE == 1) { uint64_t v; asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory"); acc += v; }
            if (MODE == 2) { uint32_t v; asm volatile("ds_read_u8 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory"); acc += v; }
            if (MODE == 3 || MODE == 4) { asm volatile("ds_write_b64 %0, %1" :: "v"(ad), "v"(acc) : "memory"); }
            if (MODE == 5) { asm volatile("ds_write_b8 %0, %1" :: "v"(ad), "v"((uint32_t)acc) : "memory"); }
            if (MODE == 6 || MODE == 7) { uint32_t v; asm volatile("ds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad & ~(MODE == 6 ? 3u : 0u)) : "memory"); acc += v; }
            if (MODE == 8) { asm volatile("ds_write_b32 %0, %1" :: "v"(ad), "v"((uint32_t)acc) : "memory"); }
            if (MODE == 9) { asm volatile("ds_write_b16 %0, %1" :: "v"(ad), "v"((uint32_t)acc) : "memory"); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        a += 1;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[0] = t1 - t0; }
    out[1 + lane] = acc;
}
int main() {
    uint64_t* d; uint32_t* o; hipMalloc(&d, 8 * 80); hipMalloc(&o, 4 * 64);
    uint32_t h[64];
    const char* names[] = {"ds_read_b64 aligned (dependent: wait each)", "ds_read_b64 any alignment (wait each)", "ds_read_u8 (wait each)", "ds_write_b64 aligned", "ds_write_b64 any alignment",
                           "ds_write_b8", "ds_read_b32 aligned (wait each)", "ds_read_b32 any alignment (wait each)", "ds_write_b32 any alignment", "ds_write_b16 any alignment"};
    for (int pat = 0; pat < 2; pat++) {
        for (int l = 0; l < 64; l++) h[l] = pat == 0 ? (uint32_t)(l * 264 + (l * 37) % 8) : (uint32_t)((l * 2654435761u) >> 18);
        hipMemcpy(o, h, sizeof(h), hipMemcpyHostToDevice);
        printf("address pattern %d (%s)\n", pat, pat == 0 ? "stride 264 + odd byte offsets" : "pseudo-random");
        for (int m = 0; m < 10; m++) {
            const int iters = 2000;
#define RUN(M) case M: hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 16384, 0, d, o, iters); break;
            switch (m) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) }
            hipDeviceSynchronize();
            uint64_t t; hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
            printf("  %-48s %7.1f cycles per wave-instruction\n", names[m], (double)t / (iters * 8.0));
        }
    }
    return 0;
}
