"""Union-busy time / gaps of a rocprofv3 kernel trace (steady-state part). usage: python scratch/r3_timeline.py <db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name, stream_id from kernels order by start").fetchall()
n = len(rows); sub = rows[int(n * 0.35):int(n * 0.9)]
span = max(r[1] for r in sub) - sub[0][0]
# union of intervals
iv = sorted((r[0], r[1]) for r in sub)
busy = 0; cs, ce = iv[0]
gaps = []
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; gaps.append((s - ce, ce)); cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
tot = sum(r[1] - r[0] for r in sub)
print("kernels %d span %.2f ms union-busy %.2f ms (idle %.2f%%) sum-of-durations %.2f ms (overlap %.2f ms)" % (
    len(sub), span / 1e6, busy / 1e6, 100 * (1 - busy / span), tot / 1e6, (tot - busy) / 1e6))
streams = {}
for r in sub:
    streams.setdefault(r[3], [0, 0]); streams[r[3]][0] += 1; streams[r[3]][1] += r[1] - r[0]
print({k: (v[0], round(v[1] / 1e6, 2)) for k, v in streams.items()})
big = sorted(gaps, reverse=True)[:8]
print("largest gaps (us):", [round(g[0] / 1e3, 1) for g in big], " total gap ms %.2f, gaps>10us: %d = %.2f ms" % (
    sum(g[0] for g in gaps) / 1e6, sum(1 for g in gaps if g[0] > 10000), sum(g[0] for g in gaps if g[0] > 10000) / 1e6))
