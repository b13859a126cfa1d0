// Micro-benchmark: how long ONE LDS instruction of the small-file kernel's copies occupies the CU's LDS pipeline, by width, alignment and number
// of active lanes: NW wavefronts of one workgroup issue the same instruction back to back (reads: eight in flight, then one wait), so with
// NW = 4 the pipeline, not a wavefront's issue, is the bound.  Per-lane addresses are pseudo-random inside four 4 KiB windows (a file's lanes in
// its window), with the alignment asked for.  Prints cycles of the CU per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -o lds_occ_micro lds_occ_micro.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
enum { R128, R64, R32, W64, W32, W16, W8 };
template <int OP>
__global__ void k(uint64_t* out, int iters, uint32_t align, uint32_t active_of_16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, f = lane >> 4, sub = lane & 15;
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((uint32_t*)lds)[i] = i;
    __syncthreads();
    uint32_t h = (uint32_t)(lane * 2654435761u + wave * 40503u);
    uint64_t acc = 0;
    const bool on = (uint32_t)sub < active_of_16;
    uint32_t ad[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { h = h * 1664525u + 1013904223u; ad[u] = wave * 16384 / 4 * 0 + f * 4096 + (((h >> 12) & 4095u & ~(align - 1)) | (align == 1 ? 0 : 0)); if (ad[u] > f * 4096 + 4080) ad[u] -= 64; ad[u] &= ~(align - 1); }
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (on) {
            if (OP == R128) { uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(ad[u]) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 8; u++) acc += v[u].x + v[u].w; }
            if (OP == R64) { uint64_t v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) asm volatile("ds_read_b64 %0, %1" : "=v"(v[u]) : "v"(ad[u]) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 8; u++) acc += v[u]; }
            if (OP == R32) { uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) asm volatile("ds_read_b32 %0, %1" : "=v"(v[u]) : "v"(ad[u]) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 8; u++) acc += v[u]; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (OP == W64) asm volatile("ds_write_b64 %0, %1" :: "v"(ad[u]), "v"(acc) : "memory");
                if (OP == W32) asm volatile("ds_write_b32 %0, %1" :: "v"(ad[u]), "v"((uint32_t)acc) : "memory");
                if (OP == W16) asm volatile("ds_write_b16 %0, %1" :: "v"(ad[u]), "v"((uint32_t)acc) : "memory");
                if (OP == W8) asm volatile("ds_write_b8 %0, %1" :: "v"(ad[u]), "v"((uint32_t)acc) : "memory");
            }
            if (OP >= W64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    out[1 + threadIdx.x] = acc;
}
int main() {
    uint64_t* d; (void)hipMalloc(&d, 8 * 300);
    const char* names[] = {"ds_read_b128", "ds_read_b64", "ds_read_b32", "ds_write_b64", "ds_write_b32", "ds_write_b16", "ds_write_b8"};
    const int iters = 1000;
    for (int nw = 1; nw <= 4; nw += 3)
        for (int op = 0; op < 7; op++)
            for (uint32_t act = 16; act >= 5; act = act == 16 ? 5 : 0) {
                printf("%d wavefronts, %-13s %2u lanes of 16:", nw, names[op], act);
                for (uint32_t al = 1; al <= 16; al *= 2) {
                    if ((op == R128 && 0) || (op == R64 && al > 8) || (op == R32 && al > 4) || (op == W64 && al > 8) || (op == W32 && al > 4) || (op == W16 && al > 2) || (op == W8 && al > 1)) continue;
#define RUN(O) case O: hipLaunchKernelGGL(k<O>, dim3(1), dim3(64 * nw), 32768, 0, d, iters, al, act); break;
                    switch (op) { RUN(R128) RUN(R64) RUN(R32) RUN(W64) RUN(W32) RUN(W16) RUN(W8) }
                    (void)hipDeviceSynchronize();
                    uint64_t t; (void)hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
                    printf("  align %2u: %6.1f", al, (double)t / (iters * 8.0 * nw));
                }
                printf("\n");
                if (act == 5) break;
            }
    return 0;
}
