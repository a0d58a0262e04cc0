#!/usr/bin/env python3
"""Per-basic-block instruction / scratch / accvgpr / LDS counts of one kernel in a hipcc .s file.
usage: asm_blocks.py file.s kernel_substring [min_instr]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 100
start = [i for i, l in enumerate(lines) if key in l and re.match(r'^_Z\S+:', l)][0]
end = [i for i, l in enumerate(lines) if i > start and '.end_amdhsa_kernel' in l][0]
blocks = []
cur = ['entry', 0, 0, 0, 0, start]
for n, l in enumerate(lines[start:end]):
    t = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        blocks.append(cur)
        cur = [m.group(1), 0, 0, 0, 0, start + n]
        continue
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    cur[1] += 1
    if 'scratch_' in t: cur[2] += 1
    if 'v_accvgpr' in t: cur[3] += 1
    if t.startswith('ds_'): cur[4] += 1
blocks.append(cur)
print('total instr', sum(b[1] for b in blocks), 'scratch', sum(b[2] for b in blocks), 'accvgpr', sum(b[3] for b in blocks), 'ds', sum(b[4] for b in blocks))
print('label instr scratch accvgpr ds line')
for b in blocks:
    if b[1] >= mn:
        print(*b)
# back edges
labels = {b[0]: b[5] for b in blocks}
for n, l in enumerate(lines[start:end]):
    t = l.strip()
    m = re.match(r'^s_c?branch\S*\s+(\.LBB\d+_\d+)', t)
    if m and m.group(1) in labels and labels[m.group(1)] < start + n:
        print('backedge at line', start + n, '->', m.group(1), 'line', labels[m.group(1)], 'span', start + n - labels[m.group(1)])
