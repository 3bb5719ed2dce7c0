#!/usr/bin/env python3
"""Register / spill summary of every kernel in a hipcc -save-temps .s file.  usage: isa_regs.py file.s [name-filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'\.amdhsa_kernel (\S+)', s):
    name = m.group(1)
    if flt not in name:
        continue
    blk = s[m.start():s.index('.end_amdhsa_kernel', m.start())]
    g = lambda k: (re.search(r'\.amdhsa_' + k + r'\s+(\S+)', blk) or [None, '?'])[1]
    i = s.index(name + ':'); j = s.index('s_endpgm', i)
    body = s[i:j]
    print(f"{name[:64]:64s} vgpr {g('next_free_vgpr'):>4} sgpr {g('next_free_sgpr'):>4} scratch {g('private_segment_fixed_size'):>5} "
          f"spill_instrs {body.count('scratch_')}")
