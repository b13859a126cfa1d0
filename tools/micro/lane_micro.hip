// Micro-benchmark (not product code): the serial state walk with the three FSE states in three lanes of a quad
// (one table read per step instead of three), checked against the one-lane form on the same synthetic tables.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <cstddef>
constexpr int kRing = 8192;
struct Sh { uint8_t ring[kRing + 16]; uint64_t ll[512], ml[512], of[256]; uint64_t dummy; };
__shared__ Sh S;
#define SDWA1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define DPPALL " row_mask:0xf bank_mask:0xf"

// one-lane reference (the product's order)
#define STEP_REF \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_add3_u32 v64, v49, v51, v53\n" \
    "v_add_u32_e32 v65, v49, v51\n" \
    "v_sub_u32_sdwa v68, %[av], v64" SDWA1 \
    "v_sub_u32_sdwa %[Gm], %[Gm], v64" SDWA1 \
    "v_lshrrev_b64 v[66:67], v68, v[54:55]\n" \
    "v_lshrrev_b32_e32 v71, 3, %[Gm]\n" \
    "v_bfe_u32 v64, v66, 0, v49\n" \
    "v_bfe_u32 v69, v66, v49, v51\n" \
    "v_bfe_u32 v70, v66, v65, v53\n" \
    "v_lshl_add_u32 %[vO], v64, 3, v48\n" \
    "v_lshl_add_u32 %[vM], v69, 3, v50\n" \
    "v_lshl_add_u32 %[vL], v70, 3, v52\n" \
    "ds_read_b64 v[48:49], %[vO] offset:%[oO]\n" \
    "ds_read_b64 v[50:51], %[vM] offset:%[oM]\n" \
    "ds_read_b64 v[52:53], %[vL] offset:%[oL]\n" \
    "v_and_b32_e32 v71, 0x1ffc, v71\n" \
    "ds_read2_b32 v[54:55], v71 offset1:1\n" \
    "v_and_or_b32 %[av], %[Gm], 31, 32\n"

// three lanes: v[48:49] own entry (lo = absolute LDS address of the next state's base, hi as before), %[A] own state address
#define STEP_LANES(SHADOW) \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_add_u32_dpp v64, v49, v49 quad_perm:[1,0,3,2]" DPPALL "\n"          /* pair sums */ \
    "v_mov_b32_dpp v65, v49 row_shr:1" DPPALL " bound_ctrl:0\n"            /* hi of the lane below (0 for O) */ \
    "s_nop 0\n" \
    "v_add_u32_dpp v64, v64, v64 quad_perm:[2,3,0,1]" DPPALL "\n"          /* all three: nb sums | total << 8 */ \
    "v_sub_u32_sdwa v68, %[av], v64" SDWA1 \
    "v_add_u32_dpp v65, v65, v65 row_shr:1" DPPALL " bound_ctrl:0\n"       /* bit offset of the own field: 0, nbO, nbO + nbM */ \
    "v_sub_u32_sdwa %[Gm], %[Gm], v64" SDWA1 \
    "v_lshrrev_b64 v[66:67], v68, v[54:55]\n" \
    "v_lshrrev_b32_e32 v71, 3, %[Gm]\n" \
    "v_bfe_u32 v69, v66, v65, v49\n" \
    "v_lshl_add_u32 %[A], v69, 3, v48\n" \
    "ds_read_b64 v[48:49], %[A]\n" \
    "v_and_b32_e32 v71, 0x1ffc, v71\n" \
    "ds_read2_b32 v[54:55], v71 offset1:1\n" \
    "v_and_or_b32 %[av], %[Gm], 31, 32\n" \
    SHADOW

template <int V>
__global__ __launch_bounds__(256) void k(const uint64_t* tabs, const uint8_t* ringsrc, uint32_t nseq, uint64_t* out, uint64_t* cyc, uint64_t* chk) {
    const uint32_t bL = (uint32_t)offsetof(Sh, ll), bM = (uint32_t)offsetof(Sh, ml), bO = (uint32_t)offsetof(Sh, of), bD = (uint32_t)offsetof(Sh, dummy);
    for (int i = threadIdx.x; i < 512; i += 256) { S.ll[i] = tabs[i] + (V ? bL : 0); S.ml[i] = tabs[512 + i] + (V ? bM : 0); }
    for (int i = threadIdx.x; i < 256; i += 256) S.of[i] = tabs[1024 + i] + (V ? bO : 0);
    for (int i = threadIdx.x; i < kRing + 16; i += 256) S.ring[i] = ringsrc[i];
    if (threadIdx.x == 0) S.dummy = bD; // hi = 0: no bits, next state = itself
    __syncthreads();
    if (threadIdx.x >= 64) return;
    uint32_t vL = 8, vM = 16, vO = 24, Gm = 30000 * 8, av = 0, n = nseq;
    __attribute__((address_space(1))) uint8_t* gw = (__attribute__((address_space(1))) uint8_t*)(out + (size_t)blockIdx.x * 65536);
    uint64_t t0 = __builtin_readcyclecounter();
    uint64_t sum = 0;
    if (V == 0) {
        asm volatile(
            "v_lshrrev_b32_e32 v71, 3, %[Gm]\n ds_read_b64 v[48:49], %[vO] offset:%[oO]\n ds_read_b64 v[50:51], %[vM] offset:%[oM]\n ds_read_b64 v[52:53], %[vL] offset:%[oL]\n"
            "v_and_b32_e32 v71, 0x1ffc, v71\n ds_read2_b32 v[54:55], v71 offset1:1\n v_and_or_b32 %[av], %[Gm], 31, 32\n"
            "1:\n" STEP_REF STEP_REF STEP_REF STEP_REF
            "s_sub_u32 %[n], %[n], 4\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n s_waitcnt lgkmcnt(0)\n"
            : [vL] "+v"(vL), [vM] "+v"(vM), [vO] "+v"(vO), [Gm] "+v"(Gm), [av] "+v"(av), [n] "+s"(n)
            : [base] "s"(gw), [oL] "n"(offsetof(Sh, ll)), [oM] "n"(offsetof(Sh, ml)), [oO] "n"(offsetof(Sh, of))
            : "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "scc", "memory");
        sum = (uint64_t)vL + vM + vO + Gm;
    } else {
        const uint32_t q = threadIdx.x & 3;
        uint32_t A = q == 0 ? bO + 24 : (q == 1 ? bM + 16 : (q == 2 ? bL + 8 : bD));
        uint32_t woff = q * 4;
#define RUNL(SH) asm volatile( \
            "v_lshrrev_b32_e32 v71, 3, %[Gm]\n ds_read_b64 v[48:49], %[A]\n v_and_b32_e32 v71, 0x1ffc, v71\n ds_read2_b32 v[54:55], v71 offset1:1\n v_and_or_b32 %[av], %[Gm], 31, 32\n" \
            "1:\n" STEP_LANES(SH) STEP_LANES(SH) STEP_LANES(SH) STEP_LANES(SH) \
            "s_sub_u32 %[n], %[n], 4\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n s_waitcnt lgkmcnt(0)\n" \
            : [A] "+v"(A), [Gm] "+v"(Gm), [av] "+v"(av), [n] "+s"(n), [woff] "+v"(woff) \
            : [base] "s"(gw), [l3] "s"(0x8888888888888888ull) \
            : "v48", "v49", "v54", "v55", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "scc", "memory")
        if (V == 1) RUNL("");
        if (V == 2) RUNL("v_cndmask_b32_e64 v70, %[A], %[Gm], %[l3]\n global_store_dword %[woff], v70, %[base]\n v_add_u32_e32 %[woff], 16, %[woff]\n v_min_i32_e32 v64, v64, v68\n");
        uint32_t rel = A - (q == 0 ? bO : (q == 1 ? bM : (q == 2 ? bL : A)));
        rel += __shfl_xor(rel, 1); rel += __shfl_xor(rel, 2);
        sum = (uint64_t)rel + Gm;
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; chk[blockIdx.x] = sum; }
}

int main(int argc, char** argv) {
    uint32_t nseq = 8000;
    uint64_t* tabs; uint8_t* ring; uint64_t *out, *cyc, *chk;
    int grid = argc > 1 ? atoi(argv[1]) : 1;
    hipMallocManaged(&tabs, 1280 * 8); hipMallocManaged(&ring, kRing + 16); hipMalloc(&out, (size_t)grid * 65536 * 8); hipMallocManaged(&cyc, grid * 8); hipMallocManaged(&chk, grid * 8);
    srand(1);
    for (int t = 0; t < 3; t++) {
        int size = t == 2 ? 256 : 512; uint64_t* tb = tabs + (t == 0 ? 0 : (t == 1 ? 512 : 1024));
        for (int i = 0; i < size; i++) {
            uint32_t nb = 1 + rand() % 5, extra = rand() % 4;
            uint32_t nbase = (rand() % (size >> nb)) << nb;
            uint32_t hi = nb | ((extra + nb) << 8) | (3 << 16) | (extra << 24);
            tb[i] = (uint64_t)(nbase * 8) | ((uint64_t)hi << 32);
        }
    }
    for (int i = 0; i < kRing + 16; i++) ring[i] = rand();
#define RUN(V, what) { for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, tabs, ring, nseq, out, cyc, chk); hipDeviceSynchronize(); } \
                 double s = 0; for (int b = 0; b < grid; b++) s += cyc[b]; printf("%-50s %.1f cycles/step   check %llu\n", what, s / grid / nseq, (unsigned long long)chk[0]); }
    RUN(0, "one lane (reference), chain only") RUN(1, "three lanes, chain only") RUN(2, "three lanes + record store + slack")
    return 0;
}
