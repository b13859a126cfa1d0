// mzd_k_xxh64.h -- part of the block pipeline of mzd_kernels.hip (see the map at the top of that file).  Included there, inside
// namespace mzd, in dependency order; not a translation unit of its own.
#pragma once
// ------------------------------------------------------------------------------------ K7
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xround(uint64_t acc, uint64_t in) { acc += in * XP2; acc = rotl64(acc, 31); return acc * XP1; }
__device__ __forceinline__ uint64_t xmerge(uint64_t h, uint64_t v) { v = xround(0, v); h ^= v; return h * XP1 + XP4; }

// XXH64(seed 0) by one wavefront, incrementally: lanes 0..3 carry the four accumulators `v`;
// `stripes` counts the 32-byte stripes already absorbed.  The hashing wavefront follows the
// executing one through the frame (xxh_advance up to the published output position) and closes
// the digest at the frame end (xxh_finish).
__device__ __forceinline__ uint64_t xxh_init(int lane) {
    const int l = lane & 3; // every group of four lanes carries the same four accumulators
    return l == 0 ? XP1 + XP2 : (l == 1 ? XP2 : (l == 2 ? 0ull : 0ull - XP1));
}
// rotl by 31 as two funnel shifts ({lo,hi} >> 1 and {hi,lo} >> 1)
__device__ __forceinline__ uint64_t rotl64_31(uint64_t x) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    return (uint64_t)__builtin_amdgcn_alignbit(lo, hi, 1) | ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, 1) << 32);
}
__device__ __forceinline__ uint64_t xround31(uint64_t acc, uint64_t in) { acc += in * XP2; acc = rotl64_31(acc); return acc * XP1; }
// acc + (`t` of the lane 4*J further up in the row of 16): DPP row_shl on the addend, folded into the two halves of the
// 64-bit add (lanes past the row's end add 0; `t` was written several instructions earlier -- the chain step in between
// -- which covers the two wait states a DPP read needs after a VALU write)
template <int J>
__device__ __forceinline__ uint64_t add_row_up(uint64_t acc, uint64_t t) {
    if (J == 0) return acc + t;
    uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
    const uint32_t tlo = (uint32_t)t, thi = (uint32_t)(t >> 32);
    if (J == 1) asm("v_add_co_u32_dpp %0, vcc, %2, %0 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32_dpp %1, vcc, %3, %1, vcc row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(lo), "+v"(hi) : "v"(tlo), "v"(thi) : "vcc");
    if (J == 2) asm("v_add_co_u32_dpp %0, vcc, %2, %0 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32_dpp %1, vcc, %3, %1, vcc row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(lo), "+v"(hi) : "v"(tlo), "v"(thi) : "vcc");
    if (J == 3) asm("v_add_co_u32_dpp %0, vcc, %2, %0 row_shl:12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32_dpp %1, vcc, %3, %1, vcc row_shl:12 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(lo), "+v"(hi) : "v"(tlo), "v"(thi) : "vcc");
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
template <int J>
__device__ __forceinline__ uint64_t xchain31(uint64_t acc, uint64_t t) { acc = add_row_up<J>(acc, t); acc = rotl64_31(acc); return acc * XP1; }
__device__ __noinline__ void xxh_advance(uint64_t& v, uint64_t& stripes, uint64_t upto, const uint8_t* p, int lane) {
    if (upto <= stripes) return;
#ifdef MZD_EXP_NOHASH
    stripes = upto; return;
#endif
    { // all 64 lanes run (four copies of a row of 16): no divergent region around the loop
        // Per stripe and accumulator: acc = rotl31(acc + in * P2) * P1.  The product in * P2 is not part of the serial
        // chain, so a row of 16 lanes computes it for FOUR stripes at once (lane 4 s + a: stripe s, accumulator a --
        // 128 contiguous bytes); the chain itself runs in the row's lanes 0..3, which pick the products of stripes
        // 1..3 out of the lanes above them (DPP row shifts folded into the adds).  That is one 64-bit multiply per
        // stripe on the chain instead of two (integer multiplies are quarter rate, and this wavefront shares its SIMD
        // with another file's walker).  Groups of 32 stripes with no per-stripe bounds checks, the next group's loads
        // in flight while the current one is absorbed, two register sets used alternately (no hand-over copies).
        gcptr q = (gcptr)(p + (lane & 15) * 8 + stripes * 32);
        uint64_t n = upto - stripes;
        uint64_t acc = v;
        uint64_t A[8], B[8];
        auto load8 = [&](uint64_t (&r)[8]) {
#pragma unroll
            for (int k = 0; k < 8; k++) __builtin_memcpy(&r[k], q + k * 128, 8);
            q += 1024; n -= 32;
        };
        auto absorb4 = [&](uint64_t in) { // four stripes
            const uint64_t t = in * XP2;
            acc = xchain31<0>(acc, t);
            acc = xchain31<1>(acc, t);
            acc = xchain31<2>(acc, t);
            acc = xchain31<3>(acc, t);
        };
        auto absorb8 = [&](const uint64_t (&r)[8]) {
#pragma unroll
            for (int k = 0; k < 8; k++) absorb4(r[k]);
        };
        if (n >= 32) {
            load8(A);
            for (;;) {
                if (n < 32) { absorb8(A); break; }
                load8(B);
                absorb8(A);
                if (n < 32) { absorb8(B); break; }
                load8(A);
                absorb8(B);
            }
        }
        // fewer than 32 stripes left: whole groups of four, then (only at the end of a frame: the follower advances in
        // groups of 8) the last one to three stripes.
        const uint32_t g4 = (uint32_t)(n >> 2), rest = (uint32_t)(n & 3); // (wave-uniform)
#pragma unroll
        for (int k = 0; k < 8; k++) if ((uint32_t)k < g4 || ((uint32_t)k == g4 && (uint32_t)((lane & 15) >> 2) < rest)) __builtin_memcpy(&A[k], q + k * 128, 8); // (never past stripe `upto`)
#pragma unroll
        for (int k = 0; k < 8; k++) if ((uint32_t)k < g4) absorb4(A[k]);
        if (rest) {
            uint64_t in = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) if ((uint32_t)k == g4) in = A[k];
            const uint64_t t = in * XP2;
            acc = xchain31<0>(acc, t);
            if (rest > 1) acc = xchain31<1>(acc, t);
            if (rest > 2) acc = xchain31<2>(acc, t);
        }
        v = acc;
    }
    stripes = upto;
}
__device__ __noinline__ uint64_t xxh_finish(uint64_t v, const uint8_t* p, uint64_t n, int lane) {
    uint64_t h;
    if (n >= 32) {
        uint64_t v1 = __shfl(v, 0), v2 = __shfl(v, 1), v3 = __shfl(v, 2), v4 = __shfl(v, 3);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = XP5;
    }
    h += n;
    const uint8_t* q = p + (n / 32) * 32;
    const uint8_t* end = p + n;
    while (q + 8 <= end) { h ^= xround(0, ld64(q)); h = rotl64(h, 27) * XP1 + XP4; q += 8; }
    if (q + 4 <= end) { h ^= (uint64_t)ld32(q) * XP1; h = rotl64(h, 23) * XP2 + XP3; q += 4; }
    while (q < end) { h ^= (uint64_t)(*q) * XP5; h = rotl64(h, 11) * XP1; q++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

