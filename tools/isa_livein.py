"""Approximate VGPR live-in set of a basic block (registers read before being written in the block).
usage: isa_livein.py file.s kernel label"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
name, label = sys.argv[2], sys.argv[3]
b = next(i for i, l in enumerate(s) if l.startswith(name + ':'))
e = next(i for i in range(b, len(s)) if s[i].startswith('.Lfunc_end'))
bb = next(i for i in range(b, e) if s[i].startswith(label + ':'))
be = next(i for i in range(bb + 1, e) if re.match(r'^\.LBB\d+_\d+:', s[i]))
def regs(tok):
    out = []
    for m in re.finditer(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]', tok):
        if m.group(1): out.append(int(m.group(1)))
        else: out += list(range(int(m.group(2)), int(m.group(3)) + 1))
    return out
written, livein = set(), set()
for l in s[bb + 1:be]:
    if not l.startswith('\t') or l.startswith('\t.') or l.startswith('\t;'): continue
    l = l.split(';')[0]
    parts = l.strip().split(None, 1)
    if len(parts) < 2: continue
    op, args = parts
    ops = [a.strip() for a in args.split(',')]
    stores = op.startswith(('ds_write', 'ds_store', 'global_store', 'scratch_store', 'buffer_store', 's_', 'v_cmp', 'v_writelane')) and not op.startswith('v_cmpx')
    dst = [] if stores else regs(ops[0])
    src = regs(','.join(ops if stores else ops[1:]))
    if op.startswith('v_writelane'): src = regs(ops[0]); dst = regs(ops[0])
    for r in src:
        if r not in written: livein.add(r)
    written.update(dst)
print(label, 'live-in VGPRs:', len(livein), 'written:', len(written), 'max reg', max(livein | written))
