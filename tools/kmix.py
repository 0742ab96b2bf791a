#!/usr/bin/env python3
"""usage: tools/kmix.py <file.s> <mangled-name-substring>  -> static instruction mix of one kernel"""
import collections
import sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [l.split(':')[0] for l in s.splitlines() if l.startswith('_Z') and key in l and ':' in l]
for name in names[:int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
    i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
    cnt = collections.Counter()
    for l in s[i:j].splitlines():
        l = l.strip()
        if not l or l.startswith(('.', ';', '_Z')) or l.endswith(':'):
            continue
        cnt[l.split()[0]] += 1
    groups = collections.Counter()
    for op, c in cnt.items():
        g = 'mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_')) else 'other'
        groups[g] += c
    print(name, sum(cnt.values()), dict(groups))
    print('   ', ', '.join(f"{op}:{c}" for op, c in cnt.most_common(24)))
