#!/usr/bin/env python3
"""Compressed instruction timeline of one kernel from a device .s file: runs of VALU (v), SALU (s),
loads (L), stores (S), LDS (D), waits (W[...]) and barriers (B)."""
import re
import sys

def main(path, pat):
    s = open(path).read()
    for m in re.finditer(r'^(_Z\w+):', s, re.M):
        name = m.group(1)
        if pat not in name:
            continue
        j = s.find('s_endpgm', m.end())
        cat = []
        for l in s[m.end():j].split('\n'):
            t = l.strip()
            if not l.startswith('\t') or not t or t[0] in '.;':
                continue
            op = t.split()[0]
            if op.startswith(('global_load', 'buffer_load', 'flat_load')): c = 'L'
            elif op.startswith(('global_store', 'buffer_store', 'flat_store')): c = 'S'
            elif op.startswith('s_waitcnt'): c = 'W[' + ' '.join(t.split()[1:]) + ']'
            elif op.startswith('s_barrier'): c = 'B'
            elif op.startswith('ds_'): c = 'D'
            elif op.startswith('v_'): c = 'v'
            elif op.startswith('s_'): c = 's'
            else: c = '?'
            cat.append(c)
        out, prev, cnt = [], None, 0
        for c in cat + [None]:
            if c == prev and c is not None and len(c) == 1:
                cnt += 1
            else:
                if prev is not None:
                    out.append(prev + (str(cnt) if cnt > 1 else ''))
                prev, cnt = c, 1
        print(name[:100]); print(' '.join(out)); print()

if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
