#!/usr/bin/env python3
"""Instruction histogram per kernel from a hipcc -S device assembly file."""
import collections
import re
import sys

def main(path, patterns):
    s = open(path).read()
    for m in re.finditer(r'^(_Z\w+):', s, re.M):
        name = m.group(1)
        if patterns and not any(p in name for p in patterns):
            continue
        j = s.find('s_endpgm', m.end())
        if j < 0:
            continue
        body = s[m.end():j]
        ops = collections.Counter()
        for l in body.split('\n'):
            t = l.strip()
            if not l.startswith('\t') or not t or t[0] in '.;':
                continue
            ops[t.split()[0]] += 1
        tot = sum(ops.values())
        valu = sum(v for k, v in ops.items() if k.startswith('v_'))
        print(f"{name[:90]}\n  total {tot}  valu {valu}  mul {sum(v for k,v in ops.items() if 'mul' in k or 'mad_u64' in k)}")
        print('  ' + ', '.join(f'{k}:{v}' for k, v in ops.most_common(32)))

if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
