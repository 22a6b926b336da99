"""Per-basic-block summary of a kernel in a .s file: instruction count, scratch ops, DPP ops, barriers, LDS ops, source-line range.
usage: isa_blocks.py file.s kernel_mangled_name [min_scratch]"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]; mins = int(sys.argv[3]) if len(sys.argv) > 3 else 0
b = next(i for i, l in enumerate(s) if l.startswith(name + ':'))
e = next(i for i in range(b, len(s)) if s[i].startswith('.Lfunc_end'))
files = {}
for l in s:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
blocks = []; cur = dict(label='entry', n=0, sc=0, dpp=0, bar=0, lds=0, vm=0, lines=set(), br='')
for l in s[b + 1:e]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur); cur = dict(label=m.group(1), n=0, sc=0, dpp=0, bar=0, lds=0, vm=0, lines=set(), br=''); continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur['lines'].add((files.get(int(m.group(1)), '?'), int(m.group(2)))); continue
    if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
        cur['n'] += 1
        if 'scratch_' in l: cur['sc'] += 1
        if 'dpp' in l or 'quad_perm' in l: cur['dpp'] += 1
        if 's_barrier' in l: cur['bar'] += 1
        if '\tds_' in l: cur['lds'] += 1
        if 'global_' in l or 'buffer_' in l: cur['vm'] += 1
        m = re.match(r'\s*s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m: cur['br'] += m.group(1) + ' '
blocks.append(cur)
for k in blocks:
    if k['sc'] >= mins:
        ls = sorted(k['lines']); f = {}
        for fn, ln in ls: f.setdefault(fn, []).append(ln)
        rng = ' '.join(f"{fn}:{min(v)}-{max(v)}" for fn, v in f.items() if fn.startswith('rp_islands') or fn.startswith('rp_lane'))
        print(f"{k['label']:12s} n={k['n']:5d} scratch={k['sc']:3d} dpp={k['dpp']:3d} bar={k['bar']} lds={k['lds']:3d} vmem={k['vm']:3d} -> {k['br']:24s} {rng}")
