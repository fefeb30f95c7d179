#!/usr/bin/env python3
"""Instruction mix of one kernel from a device-only -S dump (hipcc --cuda-device-only -S).  Usage: isa_mix.py file.s substring"""
import sys
from collections import Counter
s = open(sys.argv[1]).read()
sub = sys.argv[2]
names = [l.split(':')[0] for l in s.splitlines() if sub in l and l.startswith('_Z') and ':' in l]
for name in names[:int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
    i = s.index('\n' + name + ':'); j = s.index('s_endpgm', i)
    ins = []
    for l in s[i:j].splitlines():
        t = l.strip()
        if not l.startswith('\t') or not t or t.startswith(('.', ';')):
            continue
        ins.append(t.split()[0])
    c = Counter(ins)
    print(name, len(ins))
    print(' '.join(f"{k}:{v}" for k, v in c.most_common(60)))
