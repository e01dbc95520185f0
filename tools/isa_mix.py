#!/usr/bin/env python3
"""Instruction mix of the hot loop of each strip kernel in a hipcc -save-temps .s file.

usage: tools/isa_mix.py file.s [substring-of-kernel-name ...]
"""
import collections
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2:]
    starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_ZN8ssim_hip\S*):', s, re.M)]
    for idx, (pos, name) in enumerate(starts):
        if 'strip' not in name or 'kernel' not in name:
            continue
        if want and not any(w in name for w in want):
            continue
        end = starts[idx + 1][0] if idx + 1 < len(starts) else len(s)
        lines = s[pos:end].split('\n')
        lab = {}
        for i, l in enumerate(lines):
            m = re.match(r'^(\.LBB\d+_\d+):', l)
            if m:
                lab[m.group(1)] = i
        # hot loop = the shortest backward-branch span holding >= 90% of the function's FMAs
        isfma = [1 if re.match(r'\s*v_(pk_)?fma(c)?_f(32|64)', l) else 0 for l in lines]
        total = sum(isfma)
        best = None
        for i, l in enumerate(lines):
            m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in lab and lab[m.group(1)] < i:
                span = (lab[m.group(1)], i)
                if sum(isfma[span[0]:span[1]]) >= 0.9 * total and (best is None or span[1] - span[0] < best[1] - best[0]):
                    best = span
        body = lines[best[0]:best[1] + 1] if best else lines
        c = collections.Counter()
        for l in body:
            l = l.strip()
            if not l or l[0] in ';.' or l.endswith(':'):
                continue
            c[l.split()[0]] += 1
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        print('%s\n  loop instrs %d  VALU %d  (pk %d)  DS %d  VMEM %d  SALU %d  waitcnt %d' % (
            name, sum(c.values()), valu, sum(v for k, v in c.items() if k.startswith('v_pk_')),
            sum(v for k, v in c.items() if k.startswith('ds_')),
            sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_'))),
            sum(v for k, v in c.items() if k.startswith('s_') and not k.startswith('s_waitcnt')), c['s_waitcnt']))
        print('   ' + '  '.join('%s:%d' % kv for kv in c.most_common(45)))


if __name__ == '__main__':
    main()
