// Micro-benchmark: the LDS access shapes of the small-file kernel's match copies (mzd_lds.hip), ONE wavefront: four "files" of 16 lanes, every
// file's lanes at consecutive addresses behind the file's own base.  Cycles per wave-instruction (read + wait, or write).
//   hipcc --offload-arch=gfx950 -O3 -o match_micro match_micro.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
// MODE 0: ds_read_u16 at base + 2 sub, all lanes   1: the same, 5 lanes a file   2: ds_read_u8 at base + sub, all lanes   3: u8, 9 lanes a file
//      4: ds_read_u16, all lanes, bases even       5: ds_write_b16 5 lanes a file 6: ds_write_b8 9 lanes a file           7: ds_read_b32 at base + 4 sub (any alignment)
//      8: ds_read_u16 all lanes, lanes 5.. of a file at an address out of range
template <int MODE>
__global__ void k(uint64_t* out, const uint32_t* bases, int iters) {
    const int lane = threadIdx.x, f = lane >> 4, sub = lane & 15;
    for (int i = lane; i < 32768 / 4; i += 64) ((uint32_t*)lds)[i] = i;
    __syncthreads();
    uint32_t b = bases[f];
    if (MODE == 4) b &= ~1u;
    uint64_t acc = 0;
    const uint64_t full = __builtin_amdgcn_read_exec();
    const uint64_t m5 = 0x001F001F001F001Full, m9 = 0x01FF01FF01FF01FFull;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            uint32_t ad = (b + 97 * u) & 16383;
            uint32_t v = 0;
            if (MODE == 0 || MODE == 4) { ad += 2 * sub; asm volatile("ds_read_u16 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory"); }
            if (MODE == 8) { ad = sub < 5 ? ad + 2 * sub : 0xFFFF0000u + 2 * sub; asm volatile("ds_read_u16 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory"); }
            if (MODE == 1) { ad += 2 * sub; asm volatile("s_mov_b64 exec, %2\nds_read_u16 %0, %1\ns_waitcnt lgkmcnt(0)\ns_mov_b64 exec, %3" : "=v"(v) : "v"(ad), "s"(m5), "s"(full) : "memory"); }
            if (MODE == 2) { ad += sub; asm volatile("ds_read_u8 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory"); }
            if (MODE == 3) { ad += sub; asm volatile("s_mov_b64 exec, %2\nds_read_u8 %0, %1\ns_waitcnt lgkmcnt(0)\ns_mov_b64 exec, %3" : "=v"(v) : "v"(ad), "s"(m9), "s"(full) : "memory"); }
            if (MODE == 5) { ad += 2 * sub; asm volatile("s_mov_b64 exec, %2\nds_write_b16 %0, %1\ns_mov_b64 exec, %3" :: "v"(ad), "v"((uint32_t)acc), "s"(m5), "s"(full) : "memory"); }
            if (MODE == 6) { ad += sub; asm volatile("s_mov_b64 exec, %2\nds_write_b8 %0, %1\ns_mov_b64 exec, %3" :: "v"(ad), "v"((uint32_t)acc), "s"(m9), "s"(full) : "memory"); }
            if (MODE == 7) { ad += 4 * sub; asm volatile("ds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory"); }
            acc += v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        b += 1;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) out[0] = t1 - t0;
    out[1 + lane] = acc;
}
int main() {
    uint64_t* d; uint32_t* o; hipMalloc(&d, 8 * 80); hipMalloc(&o, 16);
    const uint32_t h[4] = {1001, 5308, 9611, 13918};
    hipMemcpy(o, h, sizeof(h), hipMemcpyHostToDevice);
    const char* names[] = {"ds_read_u16 base + 2 sub, all lanes", "ds_read_u16, 5 lanes a file", "ds_read_u8 base + sub, all lanes", "ds_read_u8, 9 lanes a file",
                           "ds_read_u16 all lanes, even bases", "ds_write_b16, 5 lanes a file", "ds_write_b8, 9 lanes a file", "ds_read_b32 base + 4 sub, all lanes", "ds_read_u16, lanes 5.. out of range"};
    for (int m = 0; m < 9; m++) {
        const int iters = 2000;
#define RUN(M) case M: hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 32768, 0, d, o, iters); break;
        switch (m) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) }
        hipDeviceSynchronize();
        uint64_t t; hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
        printf("  %-44s %7.1f cycles per wave-instruction\n", names[m], (double)t / (iters * 8.0));
    }
    return 0;
}
