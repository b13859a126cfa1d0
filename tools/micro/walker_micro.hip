// Micro-benchmark of the serial state walk (not product code): cycles per sequence of loop variants.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int kRing = 8192;
struct Sh { uint64_t ll[512], ml[512], of[256]; uint8_t ring[kRing + 16]; };

template <int VARIANT>
__global__ __launch_bounds__(256) void k(const uint64_t* tabs, const uint8_t* ringsrc, uint32_t nseq, uint64_t* out, uint64_t* cyc) {
    __shared__ Sh S;
    for (int i = threadIdx.x; i < 512; i += 256) { S.ll[i] = tabs[i]; S.ml[i] = tabs[512 + i]; }
    for (int i = threadIdx.x; i < 256; i += 256) S.of[i] = tabs[1024 + i];
    for (int i = threadIdx.x; i < kRing + 16; i += 256) S.ring[i] = ringsrc[i];
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const uint8_t* tL = (const uint8_t*)S.ll; const uint8_t* tM = (const uint8_t*)S.ml; const uint8_t* tO = (const uint8_t*)S.of;
    uint32_t vL = 8, vM = 16, vO = 24, G = 60000 * 8;
    __attribute__((address_space(1))) uint64_t* gw = (__attribute__((address_space(1))) uint64_t*)(out + (size_t)blockIdx.x * 65536);
    uint64_t t0 = __builtin_readcyclecounter();
    uint64_t Wlo = 0, Whi = 0; uint32_t wbase = 0; // variant 3: register window
    for (uint32_t i = 0; i < nseq; i++) {
        uint64_t eL, eM, eO;
        __builtin_memcpy(&eL, tL + vL, 8); __builtin_memcpy(&eM, tM + vM, 8); __builtin_memcpy(&eO, tO + vO, 8);
        uint32_t t = G - 57;
        uint64_t X;
        if (VARIANT == 0) __builtin_memcpy(&X, &S.ring[(t >> 3) & (kRing - 1)], 8);           // unaligned 8-byte window
        else if (VARIANT == 1) __builtin_memcpy(&X, &S.ring[(t >> 3) & (kRing - 8)], 8);      // aligned (wrong bits; timing only)
        else if (VARIANT == 2) X = 0x0123456789abcdefull ^ t;                                    // no window read at all
        else if (VARIANT == 5) __builtin_memcpy(&X, &S.ring[(t >> 3) & (kRing - 4)], 8);      // 4-byte aligned 8-byte read
        else if (VARIANT == 6) { uint32_t a = (t >> 3) & (kRing - 4); uint32_t lo, hi; __builtin_memcpy(&lo, &S.ring[a], 4); __builtin_memcpy(&hi, &S.ring[a + 4], 4); X = ((uint64_t)hi << 32) | lo; } // two aligned dwords
        else { X = Wlo; }
        if (VARIANT != 4) gw[i] = (uint64_t)(vL | (vM << 12)) | ((uint64_t)(G | (vO << 21)) << 32);
        asm volatile("" : "+v"(X));
        uint32_t hL = (uint32_t)(eL >> 32), hM = (uint32_t)(eM >> 32), hO = (uint32_t)(eO >> 32);
        uint32_t total = ((hL + hM + hO) >> 8) & 0xFF;
        uint32_t r = t & 7;
        uint32_t oO = r + 57 - total, oM = oO + hO, oL = oM + hM;
        uint32_t bO = __builtin_amdgcn_ubfe((uint32_t)(X >> (oO & 63)), 0, hO);
        uint32_t bM = __builtin_amdgcn_ubfe((uint32_t)(X >> (oM & 63)), 0, hM);
        uint32_t bL = __builtin_amdgcn_ubfe((uint32_t)(X >> (oL & 63)), 0, hL);
        vO = (uint32_t)eO + (bO << 3); vM = (uint32_t)eM + (bM << 3); vL = (uint32_t)eL + (bL << 3);
        G -= total;
        if (G < 4096) G += 50000 * 8;
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; out[(size_t)blockIdx.x * 65536 + 65535] = vL + vM + vO + G + Wlo + Whi + wbase; }
}


__global__ __launch_bounds__(256) void k2(const uint64_t* tabs, const uint8_t* ringsrc, uint32_t nseq, uint64_t* out, uint64_t* cyc) {
    __shared__ Sh S;
    for (int i = threadIdx.x; i < 512; i += 256) { S.ll[i] = tabs[i]; S.ml[i] = tabs[512 + i]; }
    for (int i = threadIdx.x; i < 256; i += 256) S.of[i] = tabs[1024 + i];
    for (int i = threadIdx.x; i < kRing + 16; i += 256) S.ring[i] = ringsrc[i];
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const uint8_t* tL = (const uint8_t*)S.ll; const uint8_t* tM = (const uint8_t*)S.ml; const uint8_t* tO = (const uint8_t*)S.of;
    uint32_t vL = 8, vM = 16, vO = 24, G = 60000 * 8;
    __attribute__((address_space(1))) uint8_t* gw = (__attribute__((address_space(1))) uint8_t*)(out + (size_t)blockIdx.x * 65536);
    uint32_t woff = 0;
    asm volatile("" : "+v"(woff));
    uint64_t t0 = __builtin_readcyclecounter();
    uint32_t nslow = 0;
#pragma unroll 2
    for (uint32_t i = 0; i < nseq; i++) {
        uint64_t eL, eM, eO;
        __builtin_memcpy(&eL, tL + vL, 8); __builtin_memcpy(&eM, tM + vM, 8); __builtin_memcpy(&eO, tO + vO, 8);
        uint32_t u = G - 33;
        uint32_t a = (u >> 3) & (kRing - 4);
        uint64_t X;
        __builtin_memcpy(&X, &S.ring[a], 8);
        *(__attribute__((address_space(1))) uint64_t*)(gw + woff) = (uint64_t)(vL | (vM << 12)) | ((uint64_t)(G | (vO << 21)) << 32);
        woff += 8;
        asm volatile("" : "+v"(X));
        uint32_t hL = (uint32_t)(eL >> 32), hM = (uint32_t)(eM >> 32), hO = (uint32_t)(eO >> 32);
        uint32_t total = ((hL + hM + hO) >> 8) & 0xFF;
        uint32_t av = (u & 31) + 33;
        uint32_t oO = av - total;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(total > av) != 0, 0)) { // window one dword lower
            __builtin_memcpy(&X, &S.ring[(a - 4) & (kRing - 4)], 8);
            oO += 32; nslow++;
        }
        uint32_t oM = oO + hO, oL = oM + hM;
        uint32_t bO = __builtin_amdgcn_ubfe((uint32_t)(X >> (oO & 63)), 0, hO);
        uint32_t bM = __builtin_amdgcn_ubfe((uint32_t)(X >> (oM & 63)), 0, hM);
        uint32_t bL = __builtin_amdgcn_ubfe((uint32_t)(X >> (oL & 63)), 0, hL);
        vO = (uint32_t)eO + (bO << 3); vM = (uint32_t)eM + (bM << 3); vL = (uint32_t)eL + (bL << 3);
        G -= total;
        if (G < 4096) G += 50000 * 8;
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; out[(size_t)blockIdx.x * 65536 + 65535] = vL + vM + vO + G + nslow; }
}


template <int RD>
__global__ __launch_bounds__(256) void k3(const uint64_t* tabs, const uint8_t* ringsrc, uint32_t nseq, uint64_t* out, uint64_t* cyc) {
    __shared__ Sh S;
    for (int i = threadIdx.x; i < 512; i += 256) { S.ll[i] = tabs[i]; S.ml[i] = tabs[512 + i]; }
    for (int i = threadIdx.x; i < 256; i += 256) S.of[i] = tabs[1024 + i];
    for (int i = threadIdx.x; i < kRing + 16; i += 256) S.ring[i] = ringsrc[i];
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const uint8_t* tL = (const uint8_t*)S.ll; const uint8_t* tM = (const uint8_t*)S.ml; const uint8_t* tO = (const uint8_t*)S.of;
    uint32_t vL = 8, vM = 16, vO = 24, G = 60000 * 8;
    __attribute__((address_space(1))) uint8_t* gw = (__attribute__((address_space(1))) uint8_t*)(out + (size_t)blockIdx.x * 65536);
    uint32_t woff = 0;
    asm volatile("" : "+v"(woff));
    uint64_t t0 = __builtin_readcyclecounter();
    uint64_t bad = 0;
    const uint32_t ringbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)S.ring;
    for (uint32_t i = 0; i < nseq; i++) {
        uint64_t eL, eM, eO;
        __builtin_memcpy(&eL, tL + vL, 8); __builtin_memcpy(&eM, tM + vM, 8); __builtin_memcpy(&eO, tO + vO, 8);
        uint32_t u = G - 33;
        uint64_t X;
        if (RD == 0) { uint32_t a = (u >> 3) & (kRing - 4); __builtin_memcpy(&X, &S.ring[a], 8); }
        else {
            uint32_t a = ringbase + ((u >> 3) & (kRing - 4));
            asm volatile("ds_read_b64 %0, %1" : "=v"(X) : "v"(a) : "memory");
        }
        *(__attribute__((address_space(1))) uint64_t*)(gw + woff) = (uint64_t)(vL | (vM << 12)) | ((uint64_t)(G | (vO << 21)) << 32);
        woff += 8;
        if (RD == 0) asm volatile("" : "+v"(X)); else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X) :: "memory");
        uint32_t hL = (uint32_t)(eL >> 32), hM = (uint32_t)(eM >> 32), hO = (uint32_t)(eO >> 32);
        uint32_t total = ((hL + hM + hO) >> 8) & 0xFF;
        uint32_t av = (u & 31) + 33;
        uint32_t oO = av - total;
        bad |= __builtin_amdgcn_ballot_w64(total > av);
        uint32_t oM = oO + hO, oL = oM + hM;
        uint32_t bO = __builtin_amdgcn_ubfe((uint32_t)(X >> (oO & 63)), 0, hO);
        uint32_t bM = __builtin_amdgcn_ubfe((uint32_t)(X >> (oM & 63)), 0, hM);
        uint32_t bL = __builtin_amdgcn_ubfe((uint32_t)(X >> (oL & 63)), 0, hL);
        vO = (uint32_t)eO + (bO << 3); vM = (uint32_t)eM + (bM << 3); vL = (uint32_t)eL + (bL << 3);
        G -= total;
        if (G < 4096) G += 50000 * 8;
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; out[(size_t)blockIdx.x * 65536 + 65535] = vL + vM + vO + G + bad; }
}

int main(int argc, char** argv) {
    uint32_t nseq = 8000;
    uint64_t* tabs; uint8_t* ring; uint64_t *out, *cyc;
    int grid = argc > 1 ? atoi(argv[1]) : 1;
    hipMallocManaged(&tabs, 1280 * 8); hipMallocManaged(&ring, kRing + 16); hipMalloc(&out, (size_t)grid * 65536 * 8); hipMallocManaged(&cyc, grid * 8);
    srand(1);
    for (int t = 0; t < 3; t++) {
        int size = t == 2 ? 256 : 512; uint64_t* tb = tabs + (t == 0 ? 0 : (t == 1 ? 512 : 1024));
        for (int i = 0; i < size; i++) {
            uint32_t nb = 1 + rand() % 5, extra = rand() % 4;
            uint32_t nbase = (rand() % (size >> nb)) << nb; // nbase + bits < size
            uint32_t hi = nb | ((extra + nb) << 8) | (3 << 16) | (extra << 24);
            tb[i] = (uint64_t)(nbase * 8) | ((uint64_t)hi << 32);
        }
    }
    for (int i = 0; i < kRing + 16; i++) ring[i] = rand();
#define RUN(V) { hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, tabs, ring, nseq, out, cyc); hipDeviceSynchronize(); \
                 hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, tabs, ring, nseq, out, cyc); hipDeviceSynchronize(); \
                 double s = 0; for (int b = 0; b < grid; b++) s += cyc[b]; printf("variant %d grid %d: %.1f cycles/seq\n", V, grid, s / grid / nseq); }
    RUN(0) RUN(5)
    { hipLaunchKernelGGL(k2, dim3(grid), dim3(256), 0, 0, tabs, ring, nseq, out, cyc); hipDeviceSynchronize();
      hipLaunchKernelGGL(k2, dim3(grid), dim3(256), 0, 0, tabs, ring, nseq, out, cyc); hipDeviceSynchronize();
      double sm = 0; for (int b = 0; b < grid; b++) sm += cyc[b]; printf("k2 grid %d: %.1f cycles/seq\n", grid, sm / grid / nseq); }
    { hipLaunchKernelGGL(k3<0>, dim3(grid), dim3(256), 0, 0, tabs, ring, nseq, out, cyc); hipDeviceSynchronize();
      hipLaunchKernelGGL(k3<0>, dim3(grid), dim3(256), 0, 0, tabs, ring, nseq, out, cyc); hipDeviceSynchronize();
      double sm = 0; for (int b = 0; b < grid; b++) sm += cyc[b]; printf("k3<read2_b32> deferred check grid %d: %.1f cycles/seq\n", grid, sm / grid / nseq); }
    { hipLaunchKernelGGL(k3<1>, dim3(grid), dim3(256), 0, 0, tabs, ring, nseq, out, cyc); hipDeviceSynchronize();
      hipLaunchKernelGGL(k3<1>, dim3(grid), dim3(256), 0, 0, tabs, ring, nseq, out, cyc); hipDeviceSynchronize();
      double sm = 0; for (int b = 0; b < grid; b++) sm += cyc[b]; printf("k3<ds_read_b64@4> deferred check grid %d: %.1f cycles/seq\n", grid, sm / grid / nseq); }
    return 0;
}
