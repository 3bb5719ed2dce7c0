#!/usr/bin/env python3
"""Control-flow skeleton of one kernel in a hipcc -save-temps .s file: labels / branches with the instruction
mix between them.  usage: isa_blocks.py file.s kernel-name-substring"""
import sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name); i = s.index(':\n', i); j = s.index('s_endpgm', i)
body = [l.strip() for l in s[i + 1:j].split('\n')]
body = [l for l in body if l and not l.startswith(';') and not l.startswith('.')]
cnt, kinds = 0, {}
def flush():
    global cnt, kinds
    if cnt:
        print(f'   [{cnt}: ' + ' '.join(f'{k}={v}' for k, v in sorted(kinds.items())) + ']')
    cnt, kinds = 0, {}
for l in body:
    op = l.split()[0]
    if l.endswith(':') or op.startswith('s_cbranch') or op.startswith('s_branch'):
        flush(); print(l)
        continue
    cnt += 1
    if op.startswith('v_mfma'): k = 'mfma'
    elif op.startswith('v_'): k = 'valu'
    elif op.startswith('s_waitcnt'): k = 'wait'
    elif op.startswith('s_'): k = 'salu'
    else: k = op
    kinds[k] = kinds.get(k, 0) + 1
flush()
