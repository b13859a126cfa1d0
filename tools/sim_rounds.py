"""Diagnostic (CPU only): the execution rounds of the small-file kernel simulated on the cfg4 corpus from the oracle's sequence dumps -- rounds per step under the
readiness rule the kernel uses, ready lanes per round, how many rounds can skip the upper 16 bytes, and the cost of the two-form policy (wide rounds until no file has more
than T sequences waiting, then the file's lanes together) for several T.  The numbers quoted in DESIGN.md 3b.   python tools/sim_rounds.py"""
import sys, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, corpus, oracle
n = 400; LPF = 16; XG = 4
kind, cfg, km, _ = bench.WORKLOADS["cfg4"]
cp = corpus.build_corpus(kind, cfg, bench.file_sizes("cfg4", n, 0, 1), kind_mod=km)
files = []
mlh = collections.Counter(); llh = collections.Counter()
for i in range(n):
    src = cp.comp[int(cp.comp_offs[i]):int(cp.comp_offs[i]) + int(cp.comp_sizes[i])].tobytes()
    rc, out, blocks, ex = oracle.decode(src, want_trace=True, dump=True)
    pos = 0; S = []
    for (ll, ml, off) in ex["seq"]:
        S.append((pos, pos + ll, ml, off, ll)); pos += ll + ml
        mlh[min(ml, 40)] += 1; llh[min(ll, 40)] += 1
    files.append(S)
tot = sum(mlh.values())
print("P(ml>=16) %.3f P(ml>=8) %.3f P(ll>=16) %.3f P(ll>=8) %.3f P(ll==0) %.3f" % (sum(v for k, v in mlh.items() if k >= 16) / tot, sum(v for k, v in mlh.items() if k >= 8) / tot, sum(v for k, v in llh.items() if k >= 16) / tot, sum(v for k, v in llh.items() if k >= 8) / tot, llh[0] / tot))
# lockstep simulation: groups of XG files, steps of LPF; per round: does any ready lane have m >= 16 / >= 8 ; ready lanes count
rounds = 0; r_any16 = 0; r_any8 = 0; ready_hist = collections.Counter(); steps = 0; lit_any16 = 0
for g0 in range(0, n, XG):
    grp = files[g0:g0 + XG]
    ns = max(len(S) for S in grp)
    for c0 in range(0, ns, LPF):
        steps += 1
        sts = [S[c0:c0 + LPF] for S in grp]
        if any(s[4] >= 16 and s[4] < 32 for st in sts for s in st): lit_any16 += 1
        pend = [[s[2] != 0 for s in st] for st in sts]
        while any(any(p) for p in pend):
            rounds += 1; a16 = a8 = False; nready = 0
            for fi, st in enumerate(sts):
                p = pend[fi]
                if not any(p): continue
                first = p.index(True); F = st[first][1]
                newp = list(p)
                for k, (op, mp, ml, off, ll) in enumerate(st):
                    if not p[k]: continue
                    simple = ml < 32 and off >= ml and off <= mp
                    if simple and mp - off + ml <= F:
                        newp[k] = False; nready += 1
                        a16 |= ml >= 16; a8 |= (ml & 8) != 0
                    elif k == first and not simple: newp[k] = False
                pend[fi] = newp
            r_any16 += a16; r_any8 += a8; ready_hist[min(nready, 20)] += 1
print("steps %d rounds %d (%.2f/step); rounds with a ready lane m>=16: %.2f, with an 8-piece: %.2f; literal steps with 16<=ll<32: %.2f" % (steps, rounds, rounds / steps, r_any16 / rounds, r_any8 / rounds, lit_any16 / steps))
print("ready lanes per round:", sorted(ready_hist.items()))
# hybrid policy: wide rounds until max pending per file <= T, then cooperative iterations (one match per file per iteration)
for T in (0, 1, 2, 3, 4, 6):
    cost = 0; wide = 0; coop = 0
    for g0 in range(0, n, XG):
        grp = files[g0:g0 + XG]
        ns = max(len(S) for S in grp)
        for c0 in range(0, ns, LPF):
            sts = [S[c0:c0 + LPF] for S in grp]
            pend = [[s[2] != 0 for s in st] for st in sts]
            first_round = True
            while any(any(p) for p in pend):
                mx = max(sum(p) for p in pend)
                if not first_round and mx <= T:
                    coop += mx; break
                first_round = False
                wide += 1
                for fi, st in enumerate(sts):
                    p = pend[fi]
                    if not any(p): continue
                    first = p.index(True); F = st[first][1]
                    newp = list(p)
                    for k, (op, mp, ml, off, ll) in enumerate(st):
                        if not p[k]: continue
                        simple = ml < 32 and off >= ml and off <= mp
                        if (simple and mp - off + ml <= F) or k == first: newp[k] = False
                    pend[fi] = newp
    print("T=%d: wide rounds/step %.2f, cooperative iterations/step %.2f -> cost/step at 70/25 instrs: %.0f (LDS ops 9/3: %.1f)" % (T, wide / steps, coop / steps, (70 * wide + 25 * coop) / steps, (9 * wide + 3 * coop) / steps))
