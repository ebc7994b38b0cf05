"""Static instruction counts of a kernel's largest loop from hipcc's device assembly: `hipcc ... --cuda-device-only -S -o x.s file.hip; python tools/count_loop.py x.s <mangled-prefix>`.
Used for tools/ubench_transpose.hip (VALU / DPP / select / LDS / barrier instructions per iteration of each mode)."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
for m in re.finditer(r'^(%s\S*):[^\n]*\n(.*?)\.Lfunc_end' % re.escape(sys.argv[2]), s, re.S | re.M):
    lines = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith((';', '.p2align', '.long', '.section'))]
    labels = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'\.?\w+:', l)}
    best = None
    for i, l in enumerate(lines):
        mm = re.match(r's_cbranch_\w+ (\S+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i and (best is None or i - labels[mm.group(1)] > best[1] - best[0]):
            best = (labels[mm.group(1)], i)
    c = Counter()
    for l in (lines[best[0]:best[1] + 1] if best else []):
        op = l.split()[0]
        if op.startswith('v_'):
            c['valu'] += 1
            c['dpp'] += ('row_' in l or 'quad_perm' in l)
            c['cndmask'] += op.startswith('v_cndmask')
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith('s_barrier'):
            c['barrier'] += 1
        elif op.startswith('s_waitcnt'):
            c['waitcnt'] += 1
    print(m.group(1)[:12], dict(c))
