// Micro-benchmark (not product code): where do the four wavefronts of a 256-thread workgroup land?
// Launches the decoder's shape (1024 workgroups x 256 threads, ~40 KB LDS each => 4 workgroups per CU)
// and records HW_ID (SIMD, CU) per wavefront.  Prints how often wave i of a workgroup sits on SIMD s,
// and how many "wave 0"s share one SIMD.
//   build: hipcc --offload-arch=gfx950 -O3 -o placement_micro placement_micro.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 4) void k(uint32_t* out, int spin) {
    __shared__ uint8_t pad[40000];
    pad[threadIdx.x] = (uint8_t)threadIdx.x;
    __syncthreads();
    uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, all 32 bits
    uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID, low 4 bits
    uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (uint64_t)spin) __builtin_amdgcn_s_sleep(8); // keep the grid co-resident
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc + pad[threadIdx.x] * 0;
    }
}

int main() {
    const int nwg = 1024;
    uint32_t* d;
    if (hipMalloc(&d, nwg * 4 * 2 * 4) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 0, 0, d, 2000000);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    std::vector<uint32_t> h(nwg * 8);
    if (hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int hist[4][4] = {};
    std::map<uint64_t, int> wave0_per_simd; // key: xcc, se, sh, cu, simd
    std::map<uint64_t, int> wg_per_cu;
    for (int g = 0; g < nwg; g++)
        for (int w = 0; w < 4; w++) {
            uint32_t hw = h[(g * 4 + w) * 2], xcc = h[(g * 4 + w) * 2 + 1] & 15;
            uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            hist[w][simd]++;
            uint64_t cukey = ((uint64_t)xcc << 20) | (se << 12) | (sh << 8) | cu;
            if (w == 0) { wave0_per_simd[(cukey << 2) | simd]++; wg_per_cu[cukey]++; }
        }
    for (int w = 0; w < 4; w++) printf("wave %d of a workgroup: SIMD0 %d  SIMD1 %d  SIMD2 %d  SIMD3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    int share[8] = {};
    for (auto& kv : wave0_per_simd) share[kv.second < 7 ? kv.second : 7]++;
    printf("SIMDs holding n wave-0s: n=1 %d  n=2 %d  n=3 %d  n=4 %d  more %d\n", share[1], share[2], share[3], share[4], share[5] + share[6] + share[7]);
    int cus[8] = {};
    for (auto& kv : wg_per_cu) cus[kv.second < 7 ? kv.second : 7]++;
    printf("CUs holding n workgroups: n=1 %d n=2 %d n=3 %d n=4 %d more %d (distinct CUs %zu)\n", cus[1], cus[2], cus[3], cus[4], cus[5] + cus[6] + cus[7], wg_per_cu.size());
    printf("first workgroups (hw_id, xcc): ");
    for (int g = 0; g < 6; g++) printf("[%08x %x] ", h[g * 8], h[g * 8 + 1] & 15);
    printf("\n");
    return 0;
}
