import sys, collections
rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
by_g = collections.defaultdict(dict)
for w, g, a, b, c, d, e, m0, m1, m2 in rows:
    if a: by_g[g][w] = (a, b, c, d, e, m0, m1, m2)
t00 = min(v[0] for v in by_g[0].values())
print('clock ticks are s_memtime units (100 MHz?)')
for g in sorted(by_g)[2:9]:
    ws = by_g[g]
    line = []
    for w in sorted(ws):
        a, b, c, d, e, m0, m1, m2 = ws[w]
        line.append(f'w{w}: q {m0} {m1-m0} {m2-m1} {b-a-m2} wait {c-b} xf {d-c} bar {e-d}')
    print(g, 'start', min(v[0] for v in ws.values()) - t00, '|', ' | '.join(line))
