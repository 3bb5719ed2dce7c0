#!/usr/bin/env python3
"""Summarise rocprofv3 output (copied from gpurun_out/) into the small CSV/JSON kept under profiles/.
usage: summarize.py <kernel_stats.csv> <pmc_fetch/counter_collection.csv> <pmc_write/counter_collection.csv> <tag>"""
import csv, json, re, sys, collections

def short(name):
    m = re.search(r'conv3x3_mfma<(\d+), (\d+), (\d), (true|false)', name)
    if m:
        return f'conv3x3_mfma<{m.group(1)}->{m.group(2)},s{m.group(3)}{",pool" if m.group(4) == "true" else ""}>'
    m = re.search(r'(\w+)(<|\()', name.replace('(anonymous namespace)::', '').replace('void ', ''))
    return m.group(1) if m else name[:60]

stats, fetch, write, tag = sys.argv[1:5]
rows = []
for r in csv.DictReader(open(stats)):
    rows.append(dict(kernel=short(r['Name']), calls=int(r['Calls']), total_us=int(r['TotalDurationNs']) / 1e3,
                     avg_us=float(r['AverageNs']) / 1e3, pct=float(r['Percentage'])))
def pmc(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            a = acc[short(r['Kernel_Name'])]
            a[0] += float(r['Counter_Value']); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}
f, w = pmc(fetch, 'FETCH_SIZE'), pmc(write, 'WRITE_SIZE')
for r in rows:
    k = r['kernel']
    r['FETCH_SIZE_KB_per_launch'] = round(f.get(k, float('nan')), 1)
    r['WRITE_SIZE_KB_per_launch'] = round(w.get(k, float('nan')), 1)
with open(f'profiles/{tag}_kernels.csv', 'w') as out:
    wr = csv.DictWriter(out, fieldnames=list(rows[0].keys()))
    wr.writeheader()
    for r in rows:
        r['total_us'] = round(r['total_us'], 1); r['avg_us'] = round(r['avg_us'], 2)
        wr.writerow(r)
print(open(f'profiles/{tag}_kernels.csv').read())
