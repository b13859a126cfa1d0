"""Diagnostic: GPU-side timeline of the last host-path call in a rocprofv3 rocpd database (kernel + memory-copy trace).
  python tools/trace_span.py <results.db> [gap_ms]   -- groups activity separated by more than gap_ms, prints the last group"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2e6
ev = [(s, e, "K " + n[:40] + " grid %d" % g) for n, s, e, g in c.execute("select name,start,end,grid_x from kernels")]
ev += [(s, e, "C %s %d B" % (n, z)) for n, s, e, z in c.execute("select name,start,end,size from memory_copies")]
ev.sort()
groups, cur = [], []
for x in ev:
    if cur and x[0] - max(y[1] for y in cur) > gap: groups.append(cur); cur = []
    cur.append(x)
groups.append(cur)
g = groups[-1]
t0 = g[0][0]
print("%d groups; last: %d events, span %.3f ms" % (len(groups), len(g), (max(y[1] for y in g) - t0) / 1e6))
for s, e, n in g: print("  %8.3f .. %8.3f  (%7.3f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
