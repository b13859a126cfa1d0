// Micro-benchmark: how many one-wavefront workgroups with B bytes of dynamic LDS a CU holds -- the occupancy API's answer and a
// census (every workgroup adds itself to a per-CU counter, waits, and reports the maximum it saw).  Fixes the LDS allocation
// granule the small-file kernel's residency arithmetic assumes (mzd_host.cpp).   hipcc --offload-arch=gfx950 -O2 -o lds_granule_micro lds_granule_micro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ unsigned char dyn[];
__global__ __launch_bounds__(64) void census(unsigned* per_cu, unsigned* maxseen, unsigned spin) {
    unsigned cu = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); // HW_REG_XCC_ID bits 0..3
    const unsigned id = (xcc & 15) * 1024 + ((cu >> 8) & 15) + 16 * ((cu >> 12) & 3) + 64 * ((cu >> 13) & 7); // cu_id, sh_id, se_id
    if (threadIdx.x == 0) {
        dyn[0] = 1;
        const unsigned now = atomicAdd(&per_cu[id], 1u) + 1;
        atomicMax(&maxseen[id], now);
        for (unsigned i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(64);
        atomicMax(&maxseen[id], per_cu[id]);
        atomicSub(&per_cu[id], 1u);
    }
}
int main() {
    unsigned *per_cu, *maxseen;
    hipMalloc(&per_cu, 16384 * 4); hipMalloc(&maxseen, 16384 * 4);
    hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned sizes[] = {16384, 17344, 17408, 17920, 18204, 18205, 20480, 27306, 27307, 32000, 32640, 32768, 32769, 33280, 40960, 54613, 54614, 81920};
    for (unsigned b : sizes) {
        int api = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, census, 64, b);
        hipMemset(per_cu, 0, 16384 * 4); hipMemset(maxseen, 0, 16384 * 4);
        hipLaunchKernelGGL(census, dim3(256 * 16), dim3(64), b, 0, per_cu, maxseen, 2000u);
        hipDeviceSynchronize();
        std::vector<unsigned> h(16384);
        hipMemcpy(h.data(), maxseen, 16384 * 4, hipMemcpyDeviceToHost);
        unsigned mx = 0, mn = ~0u, cus = 0;
        for (unsigned v : h) if (v) { cus++; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
        printf("LDS %6u B: API %2d per CU; census over %u CUs: min %u max %u\n", b, api, cus, mn, mx);
    }
    return 0;
}
