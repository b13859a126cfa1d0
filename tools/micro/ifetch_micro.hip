// Micro-benchmark (not product code): what does a taken branch to an instruction-cache line nobody has fetched yet cost one
// wavefront, while the other three wavefronts of its workgroup (a) have ended, (b) stream LDS reads, (c) store bytes all over
// a large buffer, (d) load from it?  Wavefront 0 jumps over 8 pads of 6 KiB of s_nop and times each jump (s_memtime either side).
//   build: hipcc --offload-arch=gfx950 -O3 -o ifetch_micro ifetch_micro.hip      run: ./ifetch_micro [workgroups]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#ifndef PAD
#define PAD 1536
#endif
#define STR_(x) #x
#define STR(x) STR_(x)
#define JUMP(k) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const uint64_t a_ = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)\n s_branch 1f\n .rept " STR(PAD) "\n s_nop 0\n .endr\n1:" ::: "memory"); \
    const uint64_t b_ = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) res[(size_t)blockIdx.x * 8 + k] = b_ - a_; } while (0)
__global__ __launch_bounds__(256) void k(int mode, uint8_t* buf, size_t nbuf, uint64_t* res, uint32_t* sink, int iters) {
    __shared__ uint32_t sh[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = (uint32_t)(i * 2654435761u) & 4095u;
    __syncthreads();
    if (wave == 0) {
        for (int i = 0; i < 40; i++) __builtin_amdgcn_s_sleep(100); // (the others are well under way)
        JUMP(0); JUMP(1); JUMP(2); JUMP(3); JUMP(4); JUMP(5); JUMP(6); JUMP(7);
        return;
    }
    if (mode == 0) return;
    uint32_t x = (uint32_t)lane + 64u * (uint32_t)wave, acc = 0;
    size_t p = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4099u % nbuf;
    for (int it = 0; it < iters; it++) {
        if (mode == 1) { x = sh[x & 4095u]; acc += x; }
        else if (mode == 2) { buf[p] = (uint8_t)it; p += 577u * 64u; if (p >= nbuf) p -= nbuf; }
        else { acc += buf[p]; p += 577u * 64u; if (p >= nbuf) p -= nbuf; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// runs straight through 96 KiB of s_nop on every CU: whatever the instruction caches held is gone
__global__ __launch_bounds__(64) void flush(uint32_t* sink) {
    asm volatile(".rept 24576\n s_nop 0\n .endr" ::: "memory");
    if (threadIdx.x == 12345) sink[1] = 1;
}
int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 256;
    const size_t nbuf = (size_t)256 << 20;
    uint8_t* buf; uint64_t* res; uint32_t* sink;
    if (hipMalloc(&buf, nbuf) != hipSuccess || hipMalloc(&res, (size_t)nwg * 64) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    const char* names[] = {"alone (the other three wavefronts have ended)", "beside three wavefronts streaming LDS reads", "beside three wavefronts storing bytes all over 256 MiB", "beside three wavefronts loading bytes all over 256 MiB"};
    std::vector<uint64_t> h((size_t)nwg * 8);
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(flush, dim3(2048), dim3(64), 0, 0, sink);
            hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 0, 0, mode, buf, nbuf, res, sink, mode == 1 ? 40000 : 20000);
            if (hipDeviceSynchronize() != hipSuccess) return 1;
        }
        if (hipMemcpy(h.data(), res, (size_t)nwg * 64, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        printf("%s, %d workgroups: cycles of a jump to a line not in the instruction cache (in L2: third launch, a 96 KiB kernel in between)\n", names[mode], nwg);
        for (int j = 0; j < 8; j++) {
            std::vector<uint64_t> v(nwg);
            for (int w = 0; w < nwg; w++) v[w] = h[(size_t)w * 8 + j];
            std::sort(v.begin(), v.end());
            printf("  jump %d: p10 %6llu  median %6llu  p90 %7llu  max %8llu\n", j, (unsigned long long)v[nwg / 10], (unsigned long long)v[nwg / 2], (unsigned long long)v[nwg * 9 / 10], (unsigned long long)v[nwg - 1]);
        }
    }
    return 0;
}
