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
        # every loop (backward-branch span) with a substantial share of the function's FMAs, innermost first
        isfma = [1 if re.match(r'\s*v_(pk_)?fma(c)?_f(32|64)', l) else 0 for l in lines]
        total = max(sum(isfma), 1)
        spans = []
        for i, l in enumerate(lines):
            m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in lab and lab[m.group(1)] < i:
                span = (lab[m.group(1)], i)
                if sum(isfma[span[0]:span[1]]) >= 0.1 * total:
                    spans.append(span)
        spans = [sp for sp in spans if not any(o != sp and sp[0] <= o[0] and o[1] <= sp[1] for o in spans)] or [(0, len(lines) - 1)]
        print(name)
        for span in spans:
            body = lines[span[0]:span[1] + 1]
            c = collections.Counter()
            for l in body:
                l = l.strip()
                if not l or l[0] in ';.' or l.endswith(':'):
                    continue
                c[l.split()[0]] += 1
            valu = sum(v for k, v in c.items() if k.startswith('v_'))
            print('  loop @%d: instrs %d  VALU %d  (pk %d)  DS %d  VMEM %d  SALU %d  waitcnt %d' % (
                span[0], sum(c.values()), valu, sum(v for k, v in c.items() if k.startswith('v_pk_')),
                sum(v for k, v in c.items() if k.startswith('ds_')),
                sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_'))),
                sum(v for k, v in c.items() if k.startswith('s_') and not k.startswith('s_waitcnt')), c['s_waitcnt']))
            print('     ' + '  '.join('%s:%d' % kv for kv in c.most_common(45)))


if __name__ == '__main__':
    main()
