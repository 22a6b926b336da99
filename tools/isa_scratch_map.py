"""Map scratch (spill) instructions of a kernel in a -gline-tables-only .s file to source lines.
usage: isa_scratch_map.py file.s kernel_mangled_name"""
import re, sys, collections
s = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
b = next(i for i, l in enumerate(s) if l.startswith(name + ':'))
e = next(i for i in range(b, len(s)) if s[i].startswith('.Lfunc_end'))
files = {}
for l in s:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
cur = None
hist = collections.Counter(); tot = collections.Counter()
for l in s[b:e]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
    if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
        tot[cur] += 1
        if 'scratch_' in l: hist[cur] += 1
for k, v in sorted(hist.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    print(k, v)
print('total scratch', sum(hist.values()), 'instrs', sum(tot.values()))
