#!/usr/bin/env python3
"""Condensed view of the instruction order around the MFMAs of one kernel (runs of MFMAs collapsed).
usage: isa_mfma_phase.py file.s kernel-name-substring [first_line_count]"""
import sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 120
i = s.index(name); i = s.index(':\n', i); j = s.index('s_endpgm', i)
body = [l.strip() for l in s[i + 1:j].split('\n')]
body = [l for l in body if l and not l.startswith(';') and not l.startswith('.L__') and not l.startswith('.p2')]
mf = [k for k, l in enumerate(body) if l.startswith('v_mfma')]
out, run = [], 0
for l in body[mf[0] - 14:]:
    if l.startswith('v_mfma'):
        run += 1
        continue
    if run:
        out.append(f'   <{run} mfma>'); run = 0
    out.append(l.split(';')[0][:64])
    if len(out) >= n:
        break
print('\n'.join(out))
