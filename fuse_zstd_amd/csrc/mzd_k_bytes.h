// mzd_k_bytes.h -- part of the block pipeline of mzd_kernels.hip (see the map at the top of that file).  Included there, inside
// namespace mzd, in dependency order; not a translation unit of its own.
#pragma once
// ------------------------------------------------------------------------------------ copies
// 64 lanes copy n bytes; regions do not overlap.
__device__ __noinline__ void wave_copy(uint8_t* d, const uint8_t* s, uint32_t n, int lane) {
    // head: bring d to 16-B alignment
    uint32_t head = (uint32_t)((16 - ((uintptr_t)d & 15)) & 15);
    if (head > n) head = n;
    if ((uint32_t)lane < head) d[lane] = s[lane];
    d += head; s += head; n -= head;
    uint32_t nv = n >> 4;
    uint32_t i = (uint32_t)lane;
    for (; i + 192 < nv; i += 256) { // four 16-byte loads in flight per lane (a lone wavefront is latency-bound)
        uint4 v0, v1, v2, v3;
        __builtin_memcpy(&v0, s + (size_t)i * 16, 16);
        __builtin_memcpy(&v1, s + (size_t)(i + 64) * 16, 16);
        __builtin_memcpy(&v2, s + (size_t)(i + 128) * 16, 16);
        __builtin_memcpy(&v3, s + (size_t)(i + 192) * 16, 16);
        *reinterpret_cast<uint4*>(d + (size_t)i * 16) = v0;
        *reinterpret_cast<uint4*>(d + (size_t)(i + 64) * 16) = v1;
        *reinterpret_cast<uint4*>(d + (size_t)(i + 128) * 16) = v2;
        *reinterpret_cast<uint4*>(d + (size_t)(i + 192) * 16) = v3;
    }
    for (; i < nv; i += 64) {
        uint4 v;
        __builtin_memcpy(&v, s + (size_t)i * 16, 16);
        *reinterpret_cast<uint4*>(d + (size_t)i * 16) = v;
    }
    uint32_t tail = n & 15;
    if ((uint32_t)lane < tail) d[(size_t)nv * 16 + lane] = s[(size_t)nv * 16 + lane];
}

// n threads-of-a-workgroup version (raw blocks, RLE fills)
__device__ __noinline__ void wg_copy(uint8_t* d, const uint8_t* s, uint32_t n, int tid) {
    uint32_t head = (uint32_t)((16 - ((uintptr_t)d & 15)) & 15);
    if (head > n) head = n;
    if ((uint32_t)tid < head) d[tid] = s[tid];
    d += head; s += head; n -= head;
    uint32_t nv = n >> 4;
    for (uint32_t i = tid; i < nv; i += kWG) {
        uint4 v;
        __builtin_memcpy(&v, s + (size_t)i * 16, 16);
        *reinterpret_cast<uint4*>(d + (size_t)i * 16) = v;
    }
    uint32_t tail = n & 15;
    if ((uint32_t)tid < tail) d[(size_t)nv * 16 + tid] = s[(size_t)nv * 16 + tid];
}

__device__ __noinline__ void wg_fill(uint8_t* d, uint32_t byte, uint32_t n, int tid) {
    uint32_t head = (uint32_t)((16 - ((uintptr_t)d & 15)) & 15);
    if (head > n) head = n;
    if ((uint32_t)tid < head) d[tid] = (uint8_t)byte;
    d += head; n -= head;
    uint32_t w = byte * 0x01010101u;
    uint4 v = make_uint4(w, w, w, w);
    uint32_t nv = n >> 4;
    for (uint32_t i = tid; i < nv; i += kWG) *reinterpret_cast<uint4*>(d + (size_t)i * 16) = v;
    uint32_t tail = n & 15;
    if ((uint32_t)tid < tail) d[(size_t)nv * 16 + tid] = (uint8_t)byte;
}

// 64 lanes replicate the `off` bytes before d over d[0..n)  (a match whose source overlaps its
// destination: byte k = pattern[k mod off]).  Long ones (zero pages, sparse files: one match can be a
// whole block) are not done 64 bytes at a time: once at least 4 KiB of the pattern exist, byte k equals byte
// k - P for any multiple P of off, so the rest is plain 16-byte-per-lane copying from one period back, a period at a
// time (each period is complete -- and its stores have landed -- before the next one reads it).
__device__ __noinline__ void wave_pattern(uint8_t* d, uint32_t off, uint32_t n, int lane) {
    const uint8_t* pat = d - off;
    uint32_t period = off, done = 0;
    if (off < 4096) {
        period = ((4096 + off - 1) / off) * off;
        const uint32_t head = n < period ? n : period;
        uint32_t idx = (uint32_t)lane % off;
        const uint32_t step = 64u % off;
        for (uint32_t k = lane; k < head; k += 64) {
            d[k] = pat[idx];
            idx += step;
            if (idx >= off) idx -= off;
        }
        done = head;
        wg_fence();
    }
    while (done < n) {
        const uint32_t chunk = n - done < period ? n - done : period;
        wave_copy(d + done, d + done - period, chunk, lane);
        done += chunk;
        wg_fence();
    }
}


// 64 lanes copy bytes [lo, hi) of a file's output to its mirror in the caller's pinned host memory (DevJob::dst2): 16 bytes
// per lane and store, i.e. 1 KiB per instruction pair over PCIe (tools/micro/hostwrite_micro.hip: shader stores reach the
// link's rate, 55 GB/s, from as few as 64 workgroups).  The bytes are final when this is called and only the host reads
// the mirror, after the launch: no ordering beyond the end of the kernel is needed.
__device__ __noinline__ void mirror_wave(const uint8_t* from, uint8_t* to, uint64_t lo, uint64_t hi, int lane) {
    typedef __attribute__((address_space(1))) const uint8_t* GSrc;
    typedef __attribute__((address_space(1))) uint8_t* GDst;
    GSrc f = (GSrc)from; GDst t = (GDst)to;
    uint64_t b = lo; // wave-uniform
    for (; b + 4096 <= hi; b += 4096) { // 4 KiB in flight per wavefront: four loads, then four stores
        const uint64_t o = b + (uint64_t)lane * 16;
        uint4 v0, v1, v2, v3;
        __builtin_memcpy(&v0, f + o, 16); __builtin_memcpy(&v1, f + o + 1024, 16);
        __builtin_memcpy(&v2, f + o + 2048, 16); __builtin_memcpy(&v3, f + o + 3072, 16);
        __builtin_memcpy(t + o, &v0, 16); __builtin_memcpy(t + o + 1024, &v1, 16);
        __builtin_memcpy(t + o + 2048, &v2, 16); __builtin_memcpy(t + o + 3072, &v3, 16);
    }
    for (uint64_t o = b + (uint64_t)lane * 16; o + 16 <= hi; o += 1024) {
        uint4 v;
        __builtin_memcpy(&v, f + o, 16);
        __builtin_memcpy(t + o, &v, 16);
    }
    const uint64_t tail = lo + ((hi - lo) & ~15ull); // the last partial 16 bytes: a byte per lane
    if (tail + (uint64_t)lane < hi) t[tail + lane] = f[tail + lane];
}
