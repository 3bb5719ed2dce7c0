#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 counter_collection.csv. usage: pmc_summary.py <csv> [filter]"""
import csv, re, sys, collections
def short(name):
    m = re.search(r'conv3x3_mfma<(\d+), (\d+), (true|false)', name)
    if m:
        return f'conv<{m.group(1)}->{m.group(2)},s1{",pool" if m.group(3) == "true" else ""}>'
    m = re.search(r'conv3x3_bf16x3<(\d+), (true|false)', name)
    if m:
        return f'conv3x3_bf16x3<{m.group(1)}->80,s1{",pool" if m.group(2) == "true" else ""}>'
    m = re.search(r'conv3x3_wino<(\d+), (true|false)', name)
    if m:
        return f'conv3x3_wino<{m.group(1)}->80,s1{",pool" if m.group(2) == "true" else ""}>'
    m = re.search(r'conv3x3_s2_k1<(\d+), (\d+)', name)
    if m:
        return f'conv<{m.group(1)}->{m.group(2)},s2>'
    m = re.search(r'(\w+)(<|\()', name.replace('(anonymous namespace)::', '').replace('void ', ''))
    return m.group(1) if m else name[:40]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r['Kernel_Name'])
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    a = acc[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
names = sorted({c for v in acc.values() for c in v})
w = csv.writer(sys.stdout)                   # (kernel names hold commas: quoted)
w.writerow(['kernel'] + names)
for k, v in acc.items():
    w.writerow([k] + [f'{v[c][0] / max(v[c][1], 1):.0f}' for c in names])
