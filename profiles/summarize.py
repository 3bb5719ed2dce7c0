#!/usr/bin/env python3
"""Summarise rocprofv3 output (copied from gpurun_out/) into the small CSV/JSON kept under profiles/.
usage: summarize.py <kernel_stats.csv> <pmc_fetch/counter_collection.csv> <pmc_write/counter_collection.csv> <tag>"""
import csv, json, re, sys, collections

def short(name):
    m = re.search(r'conv3x3_mfma<(\d+), (\d+), (true|false)', name)
    if m:
        return f'conv3x3_mfma<{m.group(1)}->{m.group(2)},s1{",pool" if m.group(3) == "true" else ""}>'
    m = re.search(r'conv3x3_bf16x3<(\d+), (true|false)', name)
    if m:
        return f'conv3x3_bf16x3<{m.group(1)}->80,s1{",pool" if m.group(2) == "true" else ""}>'
    m = re.search(r'conv3x3_wino<(\d+), (true|false)', name)
    if m:
        return f'conv3x3_wino<{m.group(1)}->80,s1{",pool" if m.group(2) == "true" else ""}>'
    m = re.search(r'conv3x3_s2_k1<(\d+), (\d+)', name)
    if m:
        return f'conv3x3_s2_k1<{m.group(1)}->{m.group(2)},s2>'
    m = re.search(r'hungarian_pair_kernel<(\d)(?:, (\d))?>', name)
    if m:
        return f'hungarian_pair_kernel<gap{m.group(1)}>'
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
# per-launch HBM traffic of the dominant kernel for bench.py's roofline.traffic (gfx950: FETCH_SIZE counts 64 B per
# 128-B request for wide coalesced reads -> doubled per MI355X_MICROARCH.md; WRITE_SIZE is exact; units KB)
dom = max((r for r in rows if r['kernel'].startswith('conv3x3')), key=lambda r: r['total_us'])
traffic = {'kernel': dom['kernel'], 'fetch_kb_raw': dom['FETCH_SIZE_KB_per_launch'], 'write_kb': dom['WRITE_SIZE_KB_per_launch'],
           'hbm_bytes_per_launch': int((2 * dom['FETCH_SIZE_KB_per_launch'] + dom['WRITE_SIZE_KB_per_launch']) * 1024),
           'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py; FETCH doubled (gfx950 correction '
                   'for wide coalesced reads; this kernel mixes 4-B and 16-B loads, so the doubled figure is an upper bound)'}
json.dump(traffic, open(f'profiles/{tag}_traffic.json', 'w'), indent=1)
print(traffic)
