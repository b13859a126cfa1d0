// Micro-benchmark (not product code): how fast can shader stores write pinned HOST memory over PCIe (the "mirror the output
// to the caller's buffer while decoding" idea of DESIGN.md), against the SDMA copy the host path uses today?
//   build: hipcc --offload-arch=gfx950 -O3 -o hostwrite_micro hostwrite_micro.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int W> __global__ void copy_k(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n) { // W bytes per lane per store
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * W, stride = (size_t)gridDim.x * blockDim.x * W;
    for (; i + W <= n; i += stride) {
        if (W == 16) { uint4 v = *reinterpret_cast<const uint4*>(src + i); *reinterpret_cast<uint4*>(dst + i) = v; }
        else { uint64_t v = *reinterpret_cast<const uint64_t*>(src + i); *reinterpret_cast<uint64_t*>(dst + i) = v; }
    }
}
// the mirror's pattern: wavefront w owns region w (region bytes), all wavefronts advance through their regions together,
// `piece` bytes per visit (16 B per lane per store)
__global__ void regions_k(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t region, size_t piece) {
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint8_t* s = src + w * region; uint8_t* d = dst + w * region;
    for (size_t o = 0; o < region; o += piece)
        for (size_t i = lane * 16; i < piece; i += 1024) { uint4 v = *reinterpret_cast<const uint4*>(s + o + i); *reinterpret_cast<uint4*>(d + o + i) = v; }
}
int main() {
    const size_t n = 128u << 20;
    uint8_t *d_src, *h_dst;
    if (hipMalloc(&d_src, n) != hipSuccess || hipHostMalloc(&h_dst, n, hipHostMallocPortable) != hipSuccess) return 1;
    hipMemset(d_src, 7, n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {64, 256, 1024}) {
        for (int w : {8, 16}) {
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                if (w == 16) hipLaunchKernelGGL(copy_k<16>, dim3(grid), dim3(256), 0, 0, d_src, h_dst, n);
                else hipLaunchKernelGGL(copy_k<8>, dim3(grid), dim3(256), 0, 0, d_src, h_dst, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("shader stores to host, %2d B per lane, %4d workgroups: %.3f ms = %.1f GB/s\n", w, grid, best, n / best / 1e6);
        }
    }
    for (size_t piece : {1024, 2048, 8192, 32768, 131072}) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(regions_k, dim3(1024), dim3(64), 0, 0, d_src, h_dst, (size_t)131072, piece);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("1024 wavefronts, a 128 KiB region each, %6zu B per visit: %.3f ms = %.1f GB/s\n", piece, best, n / best / 1e6);
    }
    { // the same with a host -> device copy running beside it on another stream (the host path's inputs)
        uint8_t *h_src, *d_in; hipStream_t s2;
        hipHostMalloc(&h_src, 64u << 20, hipHostMallocPortable); hipMalloc(&d_in, 64u << 20); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            for (int c = 0; c < 4; c++) hipMemcpyAsync(d_in, h_src, 64u << 20, hipMemcpyHostToDevice, s2);
            hipEventRecord(e0);
            hipLaunchKernelGGL(regions_k, dim3(1024), dim3(64), 0, 0, d_src, h_dst, (size_t)131072, (size_t)4096);
            hipEventRecord(e1); hipEventSynchronize(e1); hipStreamSynchronize(s2);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("1024 wavefronts, 128 KiB regions, 4 KiB per visit, WITH 256 MB of host -> device copies in flight: %.3f ms = %.1f GB/s\n", best, n / best / 1e6);
    }
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0); hipMemcpyAsync(h_dst, d_src, n, hipMemcpyDeviceToHost, 0); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("hipMemcpyAsync device -> host: %.3f ms = %.1f GB/s\n", best, n / best / 1e6);
    return 0;
}
