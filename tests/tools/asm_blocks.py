#!/usr/bin/env python3
"""Basic-block census of one kernel in hipcc's gfx950 assembly (-S --cuda-device-only):
per block the VALU / SALU / LDS / scratch / lane-spill (v_readlane, v_writelane) instruction counts
and the branch targets, to locate spill traffic relative to the tile loop.
usage: asm_blocks.py file.s kernel_mangled_name_prefix"""
import re, sys
src, name = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.rstrip().split(':')[0].startswith(name) and ':' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks = []
cur = {'label': 'entry', 'v': 0, 's': 0, 'ds': 0, 'scr': 0, 'lane': 0, 'f64': 0, 'br': [], 'line': start, 'vm': 0, 'trans': 0}
for i in range(start + 1, end):
    l = lines[i].strip()
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur)
        cur = {'label': m.group(1), 'v': 0, 's': 0, 'ds': 0, 'scr': 0, 'lane': 0, 'f64': 0, 'br': [], 'line': i, 'vm': 0, 'trans': 0}
        continue
    if not l or l.startswith(';') or l.startswith('.'):
        continue
    op = l.split()[0]
    if op.startswith('v_'):
        cur['v'] += 1
        if 'readlane' in op or 'writelane' in op: cur['lane'] += 1
        if '_f64' in op: cur['f64'] += 1
        if re.match(r'v_(rcp|rsq|sqrt|exp|log|sin|cos)_', op): cur['trans'] += 1
    elif op.startswith('s_'):
        cur['s'] += 1
        if op.startswith('s_cbranch') or op == 's_branch':
            cur['br'].append(l.split()[-1])
    elif op.startswith('ds_'): cur['ds'] += 1
    elif op.startswith('scratch_'): cur['scr'] += 1
    elif op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_'): cur['vm'] += 1
blocks.append(cur)
idx = {b['label']: k for k, b in enumerate(blocks)}
tot = {k: sum(b[k] for b in blocks) for k in ('v', 's', 'ds', 'scr', 'lane', 'f64', 'vm')}
print('total', tot)
for k, b in enumerate(blocks):
    back = [t for t in b['br'] if t in idx and idx[t] <= k]
    print('%4d %-12s v=%4d f64=%4d s=%4d ds=%3d vm=%2d scr=%2d lane=%3d %s%s' % (
        k, b['label'], b['v'], b['f64'], b['s'], b['ds'], b['vm'], b['scr'], b['lane'],
        ' '.join('%s(%d)' % (t, idx.get(t, -1)) for t in b['br']), '  <== BACK' if back else ''))
