// Micro-benchmark (not product code): where do the cycles of one hand-scheduled walker step go?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <cstddef>
constexpr int kRing = 8192;
struct Sh { uint8_t ring[kRing + 16]; uint64_t ll[512], ml[512], of[256]; };
__shared__ Sh S;

#define CRIT \
    "v_add3_u32 v64, v49, v51, v53\n" \
    "v_add_u32_e32 v65, v49, v51\n" \
    "v_sub_u32_sdwa v68, %[av], v64 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n" \
    "v_sub_u32_sdwa %[Gm], %[Gm], v64 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n" \
    "v_lshrrev_b64 v[66:67], v68, v[54:55]\n" \
    "v_lshrrev_b32_e32 v71, 3, %[Gm]\n" \
    "v_bfe_u32 v64, v66, 0, v49\n" \
    "v_bfe_u32 v69, v66, v49, v51\n" \
    "v_bfe_u32 v70, v66, v65, v53\n" \
    "v_lshl_add_u32 %[vO], v64, 3, v48\n" \
    "v_lshl_add_u32 %[vM], v69, 3, v50\n" \
    "v_lshl_add_u32 %[vL], v70, 3, v52\n"
#define READS \
    "ds_read_b64 v[48:49], %[vO] offset:%[oO]\n" \
    "ds_read_b64 v[50:51], %[vM] offset:%[oM]\n" \
    "ds_read_b64 v[52:53], %[vL] offset:%[oL]\n" \
    "v_and_b32_e32 v71, 0x1ffc, v71\n" \
    "ds_read2_b32 v[54:55], v71 offset1:1\n"
#define STORE "global_store_dwordx2 %[woff], v[80:81], %[base]\n"
#define REST \
    "v_min_i32_e32 %[slack], %[slack], v68\n" \
    "v_lshl_or_b32 v80, %[vM], 12, %[vL]\n" \
    "v_lshl_or_b32 v81, %[vO], 21, %[Gm]\n" \
    "v_add_u32_e32 %[woff], 8, %[woff]\n" \
    "v_and_or_b32 %[av], %[Gm], 31, 32\n"
#define AVONLY "v_and_or_b32 %[av], %[Gm], 31, 32\n"
#define WAIT "s_waitcnt lgkmcnt(0)\n"


#define SDWA1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
// new order: entries first (lgkmcnt(1): everything but the window, which was issued last), table reads interleaved, window read last
#define STEP2(SHADOW) \
    "s_waitcnt lgkmcnt(1)\n" \
    "v_add3_u32 v64, v49, v51, v53\n" \
    "v_add_u32_e32 v65, v49, v51\n" \
    "v_sub_u32_sdwa v68, %[av], v64" SDWA1 \
    "v_sub_u32_sdwa %[Gm], %[Gm], v64" SDWA1 \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_lshrrev_b64 v[66:67], v68, v[54:55]\n" \
    "v_lshrrev_b32_e32 v71, 3, %[Gm]\n" \
    "v_bfe_u32 v64, v66, 0, v49\n" \
    "v_bfe_u32 v69, v66, v49, v51\n" \
    "v_bfe_u32 v70, v66, v65, v53\n" \
    "v_lshl_add_u32 %[vO], v64, 3, v48\n" \
    "ds_read_b64 v[48:49], %[vO] offset:%[oO]\n" \
    "v_lshl_add_u32 %[vM], v69, 3, v50\n" \
    "ds_read_b64 v[50:51], %[vM] offset:%[oM]\n" \
    "v_lshl_add_u32 %[vL], v70, 3, v52\n" \
    "ds_read_b64 v[52:53], %[vL] offset:%[oL]\n" \
    "v_and_b32_e32 v71, 0x1ffc, v71\n" \
    "ds_read2_b32 v[54:55], v71 offset1:1\n" \
    SHADOW
// same, window address before the table reads (window read issued first of the four; plain lgkmcnt(0) at the top)
#define STEP3(SHADOW) \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_add3_u32 v64, v49, v51, v53\n" \
    "v_add_u32_e32 v65, v49, v51\n" \
    "v_sub_u32_sdwa v68, %[av], v64" SDWA1 \
    "v_sub_u32_sdwa %[Gm], %[Gm], v64" SDWA1 \
    "v_lshrrev_b64 v[66:67], v68, v[54:55]\n" \
    "v_lshrrev_b32_e32 v71, 3, %[Gm]\n" \
    "v_and_b32_e32 v71, 0x1ffc, v71\n" \
    "ds_read2_b32 v[54:55], v71 offset1:1\n" \
    "v_bfe_u32 v64, v66, 0, v49\n" \
    "v_bfe_u32 v69, v66, v49, v51\n" \
    "v_bfe_u32 v70, v66, v65, v53\n" \
    "v_lshl_add_u32 %[vO], v64, 3, v48\n" \
    "ds_read_b64 v[48:49], %[vO] offset:%[oO]\n" \
    "v_lshl_add_u32 %[vM], v69, 3, v50\n" \
    "ds_read_b64 v[50:51], %[vM] offset:%[oM]\n" \
    "v_lshl_add_u32 %[vL], v70, 3, v52\n" \
    "ds_read_b64 v[52:53], %[vL] offset:%[oL]\n" \
    SHADOW

template <int V>
__global__ __launch_bounds__(256) void k(const uint64_t* tabs, const uint8_t* ringsrc, uint32_t nseq, uint64_t* out, uint64_t* cyc) {
    for (int i = threadIdx.x; i < 512; i += 256) { S.ll[i] = tabs[i]; S.ml[i] = tabs[512 + i]; }
    for (int i = threadIdx.x; i < 256; i += 256) S.of[i] = tabs[1024 + i];
    for (int i = threadIdx.x; i < kRing + 16; i += 256) S.ring[i] = ringsrc[i];
    __syncthreads();
    if (threadIdx.x >= 64) return;
    uint32_t vL = 8, vM = 16, vO = 24, Gm = 30000 * 8, woff = 0, av = 40, n = nseq;
    int32_t slack = 64;
    __attribute__((address_space(1))) uint8_t* gw = (__attribute__((address_space(1))) uint8_t*)(out + (size_t)blockIdx.x * 65536);
    uint64_t t0 = __builtin_readcyclecounter();
#define BODY(STEP) asm volatile( \
        "v_lshrrev_b32_e32 v71, 3, %[Gm]\n" READS "v_lshl_or_b32 v80, %[vM], 12, %[vL]\n v_lshl_or_b32 v81, %[vO], 21, %[Gm]\n" AVONLY \
        "1:\n" STEP STEP STEP STEP \
        "s_sub_u32 %[n], %[n], 4\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n s_waitcnt lgkmcnt(0)\n" \
        : [vL] "+v"(vL), [vM] "+v"(vM), [vO] "+v"(vO), [Gm] "+v"(Gm), [woff] "+v"(woff), [slack] "+v"(slack), [av] "+v"(av), [n] "+s"(n) \
        : [base] "s"(gw), [oL] "n"(offsetof(Sh, ll)), [oM] "n"(offsetof(Sh, ml)), [oO] "n"(offsetof(Sh, of)) \
        : "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v80", "v81", "scc", "memory", "s20", "s21", "s22", "s23")
    if (V == 0) BODY(WAIT CRIT READS STORE REST);          // the product's step
    if (V == 1) BODY(WAIT CRIT READS REST);                // no record store
    if (V == 2) BODY(WAIT CRIT READS AVONLY);              // chain only
    if (V == 3) BODY(WAIT READS);                          // LDS round trip of the four reads alone (addresses fixed)
    if (V == 4) BODY(WAIT "ds_read_b64 v[48:49], %[vO] offset:%[oO]\n"); // one read, waited
    if (V == 5) BODY(CRIT AVONLY);                         // the ALU part alone (no LDS)
    if (V == 6) BODY(WAIT "v_lshl_add_u32 %[vO], v49, 3, v48\n v_and_b32_e32 %[vO], 0x7f8, %[vO]\n ds_read_b64 v[48:49], %[vO] offset:%[oO]\n"); // dependent chase: 2 VALU + read
    if (V == 7) BODY(WAIT CRIT READS STORE REST "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n");   // + 4 issue slots in the shadow
    if (V == 8) BODY(WAIT CRIT "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n" READS STORE REST);   // + 4 issue slots on the chain
    if (V == 12) BODY(WAIT "ds_read_b32 v48, %[vO] offset:%[oO]\n ds_read_b32 v50, %[vM] offset:%[oM]\n ds_read_b32 v52, %[vL] offset:%[oL]\n ds_read2_b32 v[54:55], v71 offset1:1\n"); // four reads, tables as b32
    if (V == 13) BODY(WAIT "ds_read_b32 v48, %[vO] offset:%[oO]\n ds_read_b32 v50, %[vM] offset:%[oM]\n ds_read_b32 v52, %[vL] offset:%[oL]\n ds_read_b32 v54, v71\n"); // all b32
    if (V == 14) BODY(WAIT "ds_read_b32 v48, %[vO] offset:%[oO]\n"); // one b32
    if (V == 15) BODY(WAIT "ds_read_b64 v[48:49], %[vO] offset:%[oO]\n ds_read_b64 v[50:51], %[vM] offset:%[oM]\n"); // two b64
    if (V == 16) BODY(WAIT "ds_read_b128 v[48:51], %[vO] offset:%[oO]\n"); // one b128
    if (V == 17) BODY(WAIT "ds_read2_b64 v[48:51], %[vO] offset0:2 offset1:9\n"); // one read2_b64
    if (V == 20) { asm volatile("s_mov_b64 exec, 0xffff"); BODY(WAIT CRIT READS STORE REST); }
    if (V == 21) { asm volatile("s_mov_b64 exec, 1"); BODY(WAIT CRIT READS STORE REST); }
    if (V == 22) { asm volatile("s_mov_b64 exec, 0xffff"); BODY(WAIT READS); }
    if (V == 23) { asm volatile("s_mov_b64 exec, 1"); BODY(WAIT READS); }
    if (V == 24) { asm volatile("s_mov_b64 exec, 0xffff"); BODY(CRIT AVONLY); }
    if (V == 25) { asm volatile("s_mov_b64 exec, 1"); BODY(CRIT AVONLY); }
    if (V == 26) { asm volatile("s_mov_b32 exec_hi, 0"); BODY(WAIT CRIT READS STORE REST); }
    if (V == 30) BODY("v_lshrrev_b64 v[66:67], v68, v[54:55]\n v_lshrrev_b64 v[66:67], v66, v[54:55]\n v_lshrrev_b64 v[66:67], v66, v[54:55]\n v_lshrrev_b64 v[66:67], v66, v[54:55]\n"); // 4 dependent 64-bit shifts
    if (V == 31) BODY("v_alignbit_b32 v66, v55, v54, v68\n v_alignbit_b32 v66, v55, v54, v66\n v_alignbit_b32 v66, v55, v54, v66\n v_alignbit_b32 v66, v55, v54, v66\n"); // 4 dependent alignbits
    if (V == 32) BODY("v_lshrrev_b64 v[66:67], v68, v[54:55]\n v_lshrrev_b64 v[64:65], v68, v[54:55]\n v_lshrrev_b64 v[70:71], v68, v[54:55]\n v_lshrrev_b64 v[48:49], v68, v[54:55]\n"); // 4 independent 64-bit shifts
    if (V == 33) BODY("v_alignbit_b32 v66, v55, v54, v68\n v_alignbit_b32 v64, v55, v54, v68\n v_alignbit_b32 v70, v55, v54, v68\n v_alignbit_b32 v48, v55, v54, v68\n");
    if (V == 34) BODY("v_add3_u32 v66, v55, v54, v68\n v_add3_u32 v64, v55, v54, v68\n v_add3_u32 v70, v55, v54, v68\n v_add3_u32 v48, v55, v54, v68\n");
    if (V == 35) BODY("v_add_u32_e32 v66, v55, v54\n v_add_u32_e32 v64, v55, v54\n v_add_u32_e32 v70, v55, v54\n v_add_u32_e32 v48, v55, v54\n");
    if (V == 36) BODY("v_add_u32_e32 v66, v66, v54\n v_add_u32_e32 v66, v66, v54\n v_add_u32_e32 v66, v66, v54\n v_add_u32_e32 v66, v66, v54\n");
    if (V == 37) BODY("v_add_u32_e64 v66, v55, v54\n v_add_u32_e64 v64, v55, v54\n v_add_u32_e64 v70, v55, v54\n v_add_u32_e64 v48, v55, v54\n");
    if (V == 38) BODY("v_and_b32_e32 v66, 0x1ffc, v54\n v_and_b32_e32 v64, 0x1ffc, v54\n v_and_b32_e32 v70, 0x1ffc, v54\n v_and_b32_e32 v48, 0x1ffc, v54\n"); // e32 + literal
    if (V == 39) BODY("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n");
    if (V == 40) BODY("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n");
    if (V == 41) BODY("s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n");
    if (V == 42) BODY("v_add3_u32 v66, v66, v54, v68\n v_add3_u32 v66, v66, v54, v68\n v_add3_u32 v66, v66, v54, v68\n v_add3_u32 v66, v66, v54, v68\n");
    if (V == 9) BODY(STEP2(STORE REST));
    if (V == 10) BODY(STEP3(STORE REST));
    if (V == 11) BODY(STEP2(AVONLY));
    uint64_t t1 = __builtin_readcyclecounter();
    asm volatile("s_mov_b64 exec, -1");
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; out[(size_t)blockIdx.x * 65536 + 65535] = vL + vM + vO + Gm + slack + av; }
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
            uint32_t nbase = (rand() % (size >> nb)) << nb;
            uint32_t hi = nb | ((extra + nb) << 8) | (3 << 16) | (extra << 24);
            tb[i] = (uint64_t)(nbase * 8) | ((uint64_t)hi << 32);
        }
    }
    for (int i = 0; i < kRing + 16; i++) ring[i] = rand();
#define RUN(V, what) { for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, tabs, ring, nseq, out, cyc); hipDeviceSynchronize(); } \
                 double s = 0; for (int b = 0; b < grid; b++) s += cyc[b]; printf("%-60s %.1f cycles/step\n", what, s / grid / nseq); }
    RUN(0, "full step") RUN(1, "no record store") RUN(2, "chain only") RUN(3, "four reads + wait") RUN(4, "one b64 read + wait")
    RUN(12,"four reads, tables b32") RUN(13,"four b32") RUN(14,"one b32") RUN(15,"two b64") RUN(16,"one b128") RUN(17,"one read2_b64") RUN(20,"full, exec 16 lanes") RUN(21,"full, exec 1 lane") RUN(22,"four reads, 16 lanes") RUN(23,"four reads, 1 lane") RUN(24,"ALU, 16 lanes") RUN(25,"ALU, 1 lane") RUN(26,"full, 32 lanes") RUN(30,"4 dep lshr64 (x4 per iter)") RUN(31,"4 dep alignbit") RUN(32,"4 indep lshr64") RUN(33,"4 indep alignbit") RUN(34,"4 indep add3") RUN(35,"4 indep add e32") RUN(36,"4 dep add e32") RUN(37,"4 indep add e64") RUN(38,"4 indep and e32+literal") RUN(39,"4 s_nop") RUN(40,"4 indep s_add") RUN(41,"4 dep s_add") RUN(42,"4 dep add3") RUN(9, "new order, window last + lgkmcnt(1)") RUN(10, "new order, window first") RUN(11, "new order, window last, chain only") RUN(5, "ALU part alone") RUN(6, "2 VALU + read chase") RUN(7, "full + 4 s_nop in the shadow") RUN(8, "full + 4 s_nop on the chain")
    return 0;
}
