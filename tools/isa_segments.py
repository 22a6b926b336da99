"""Static instruction mix of a kernel between s_barrier's (latency-bound kernels: count what issues)."""
import sys
from collections import Counter
s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ':')
j = s.index('.end_amdhsa_kernel', i)
body = [l.strip() for l in s[i:j].split('\n')]
ins = [l for l in body if l and not l.startswith(('.', ';', '//')) and not l.endswith(':')]
print('total', len(ins))
seg, cur = [], []
for l in ins:
    cur.append(l)
    if l.startswith('s_barrier'):
        seg.append(cur); cur = []
seg.append(cur)
for k, sg in enumerate(seg):
    c = Counter(x.split()[0] for x in sg)
    acc = c['v_accvgpr_read_b32'] + c['v_accvgpr_write_b32']
    mov = c['v_mov_b32_e32'] + c['v_pk_mov_b32'] + c['v_mov_b64_e32']
    rl = c['v_readlane_b32'] + c['v_writelane_b32']
    ds = sum(v for kk, v in c.items() if kk.startswith('ds_'))
    gl = sum(v for kk, v in c.items() if kk.startswith('global_'))
    sc = sum(v for kk, v in c.items() if kk.startswith('scratch_'))
    print(k, len(sg), 'acc', acc, 'mov', mov, 'lane', rl, 'ds', ds, 'global', gl, 'scratch', sc, 'pk', c['v_pk_mul_f32'] + c['v_pk_add_f32'] + c['v_pk_fma_f32'])
