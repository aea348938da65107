#!/usr/bin/env python3
"""Static instruction counts (VALU / SALU / memory) per basic block of one kernel in a .s file.
usage: tools/isa_blocks.py build/isa/kl_kernels.s k_intersect_bvhILb0E [min_valu]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 8
on = False; blk = 'entry'; stats = {}; order = []
for l in open(path):
    if not on:
        if re.match(r'^_Z\w*' + re.escape(key) + r'\w*:', l): on = True
        continue
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: blk = m.group(1)
    t = l.strip().split()
    if not t or t[0].startswith(';') or t[0].startswith('.') or t[0].endswith(':'): continue
    op = t[0]
    k = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'mem'
    if blk not in stats: stats[blk] = {'valu': 0, 'salu': 0, 'mem': 0}; order.append(blk)
    stats[blk][k] += 1
    if op == 's_endpgm': break
tot = {'valu': 0, 'salu': 0, 'mem': 0}
for b in order:
    s = stats[b]
    for k in tot: tot[k] += s[k]
    if s['valu'] >= minv: print(b, s)
print('total', tot)
