/* Experiment (test infrastructure, not the product): how fast does a zstd sequence decoder that starts in the MIDDLE of the
 * backward bitstream, with arbitrary FSE states, fall into step with the true decoder?  (Self-synchronisation of tANS, as used for
 * parallel decoding by Weissenberger & Schmidt, ICPP 2019 -- here for zstd's three interleaved states plus raw extra bits.)
 * Built by tests/native/fse_sync_exp.py: gcc -O2 -shared, includes the oracle with a hook behind its table build. */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
struct fse_tab_s;
static void sync_hook(const void* ll, const void* of, const void* ml, const uint8_t* p, size_t n, uint32_t nseq);
#define OZS_SEQ_HOOK(ll, of, ml, p, n, nseq) sync_hook(ll, of, ml, p, n, nseq)
#include "../../oracle/zstd_oracle.c"

#define MAXLANES 256
static int g_lanes = 64;
static uint64_t g_hist[4096];   /* sync length in sequences (speculative decoder's own count), clipped */
static uint64_t g_never, g_starts, g_blocks, g_seqs, g_bits;
static uint64_t g_rounds_hist[MAXLANES + 2];
static uint64_t g_steps_total, g_steps_serial;  /* lane-steps of the iterated scheme (max over lanes per round, summed) vs nseq */

typedef struct { int64_t pos; uint32_t sll, sof, sml; } st4;

static inline int step(const fse_tab* LL, const fse_tab* OF, const fse_tab* ML, const uint8_t* p, size_t n, st4* s) {
    fse_ent el = LL->e[s->sll], eo = OF->e[s->sof], em = ML->e[s->sml];
    if (el.sym > 35 || em.sym > 52 || eo.sym > 31) return -1;
    int64_t pos = s->pos - eo.sym - ML_BITS[em.sym] - LL_BITS[el.sym];
    if (pos < 0) return -1;
    pos -= el.nb; uint32_t a = (uint32_t)bits_at(p, n, pos, el.nb);
    pos -= em.nb; uint32_t b = (uint32_t)bits_at(p, n, pos, em.nb);
    pos -= eo.nb; uint32_t c = (uint32_t)bits_at(p, n, pos, eo.nb);
    if (pos < 0) return -1;
    s->sll = el.base + a; s->sml = em.base + b; s->sof = eo.base + c; s->pos = pos;
    return 0;
}

static void sync_hook(const void* llv, const void* ofv, const void* mlv, const uint8_t* p, size_t n, uint32_t nseq) {
    const fse_tab* LL = (const fse_tab*)llv; const fse_tab* OF = (const fse_tab*)ofv; const fse_tab* ML = (const fse_tab*)mlv;
    bbr b; if (bbr_init(&b, p, n)) return;
    if (nseq < 256) return;
    st4 s; s.sll = bbr_read(&b, LL->log); s.sof = bbr_read(&b, OF->log); s.sml = bbr_read(&b, ML->log); s.pos = b.pos;
    if (s.pos < 0) return;
    int64_t top = s.pos;
    /* the true chain: sequence index at each start position */
    int32_t* at = (int32_t*)malloc(sizeof(int32_t) * (size_t)(top + 1));
    st4* tr = (st4*)malloc(sizeof(st4) * nseq);
    for (int64_t i = 0; i <= top; i++) at[i] = -1;
    uint32_t i;
    for (i = 0; i < nseq; i++) { tr[i] = s; at[s.pos] = (int32_t)i; if (i + 1 < nseq && step(LL, OF, ML, p, n, &s)) break; }
    if (i < nseq) { free(at); free(tr); return; }
    g_blocks++; g_seqs += nseq; g_bits += (uint64_t)top;
    /* (1) sync length from the slice starts of a g_lanes-way split, states 0 */
    int L = g_lanes;
    int64_t bound[MAXLANES + 1];
    for (int k = 0; k <= L; k++) bound[k] = top - (top * k) / L;   /* bound[0] = top ... bound[L] = 0 */
    for (int k = 1; k < L; k++) {
        st4 q; q.pos = bound[k]; q.sll = 0; q.sof = 0; q.sml = 0;
        uint32_t cnt = 0; int synced = 0;
        while (cnt < 4095) {
            int32_t j = at[q.pos];
            if (j >= 0 && tr[j].sll == q.sll && tr[j].sof == q.sof && tr[j].sml == q.sml) { synced = 1; break; }
            if (step(LL, OF, ML, p, n, &q)) break;
            cnt++;
        }
        g_starts++;
        if (synced) g_hist[cnt]++; else g_never++;
    }
    /* (2) the iterated scheme: lane k decodes from its entry until pos <= bound[k+1]; entry of lane k+1 = exit of lane k */
    st4 entry[MAXLANES], exitst[MAXLANES]; int dead[MAXLANES];
    for (int k = 0; k < L; k++) { entry[k].pos = bound[k]; entry[k].sll = entry[k].sof = entry[k].sml = 0; }
    entry[0] = tr[0];
    int changed[MAXLANES]; for (int k = 0; k < L; k++) changed[k] = 1;
    int rounds = 0;
    for (;;) {
        int any = 0; uint64_t maxsteps = 0;
        for (int k = 0; k < L; k++) if (changed[k]) {
            any = 1;
            st4 q = entry[k]; uint64_t steps = 0; dead[k] = 0;
            while (q.pos > bound[k + 1]) { if (step(LL, OF, ML, p, n, &q)) { dead[k] = 1; break; } steps++; }
            exitst[k] = q; if (steps > maxsteps) maxsteps = steps;
        }
        if (!any) break;
        rounds++; g_steps_total += maxsteps;
        int nchanged[MAXLANES]; memset(nchanged, 0, sizeof nchanged);
        for (int k = 0; k + 1 < L; k++) if (changed[k]) {
            st4 e = exitst[k];
            if (dead[k]) continue;
            if (memcmp(&e, &entry[k + 1], sizeof e) != 0) { entry[k + 1] = e; nchanged[k + 1] = 1; }
        }
        memcpy(changed, nchanged, sizeof changed);
    }
    g_steps_serial += nseq;
    g_rounds_hist[rounds > MAXLANES ? MAXLANES + 1 : rounds]++;
    free(at); free(tr);
}

void exp_reset(int lanes) { g_lanes = lanes; memset(g_hist, 0, sizeof g_hist); g_never = g_starts = g_blocks = g_seqs = g_bits = 0; memset(g_rounds_hist, 0, sizeof g_rounds_hist); g_steps_total = g_steps_serial = 0; }
void exp_get(uint64_t* hist4096, uint64_t* misc8, uint64_t* rounds) {
    memcpy(hist4096, g_hist, sizeof g_hist);
    misc8[0] = g_never; misc8[1] = g_starts; misc8[2] = g_blocks; misc8[3] = g_seqs; misc8[4] = g_bits; misc8[5] = g_steps_total; misc8[6] = g_steps_serial;
    memcpy(rounds, g_rounds_hist, sizeof g_rounds_hist);
}
