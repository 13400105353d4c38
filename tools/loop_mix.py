#!/usr/bin/env python3
"""usage: tools/loop_mix.py FILE.s KERNEL_SUBSTRING -- instruction mix of the biggest loop (by matrix instructions) of a kernel in
an assembly listing (hipcc -save-temps): counts per class and the most frequent vector opcodes."""
import re, sys, collections
s = open(sys.argv[1]).read()
want = sys.argv[2]
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n', s):
    name = m.group(1)
    if want not in name: continue
    end = s.find('.Lfunc_end', m.end())
    lines = s[m.end():end].split('\n')
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r'(\.LBB\d+_\d+):', l)
        if mm: labels[mm.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        mm = re.match(r'\s+s_cbranch\w*\s+(\.LBB\d+_\d+)', l) or re.match(r'\s+s_branch\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            body = lines[labels[mm.group(1)]:i + 1]
            n = sum('v_mfma' in b for b in body)
            if best is None or n > best[0]: best = (n, body)
    if not best: continue
    cnt, ops = collections.Counter(), collections.Counter()
    for line in best[1]:
        line = line.strip()
        if not line or line[0] in ';.': continue
        op = line.split()[0]
        ops[op] += 1
        if op.startswith('v_mfma'): cnt['mfma'] += 1
        elif op.startswith('v_'): cnt['valu'] += 1
        elif op.startswith('ds_'): cnt['lds'] += 1
        elif op.startswith(('buffer_', 'global_')): cnt['vmem'] += 1
        elif op.startswith('s_'): cnt['salu'] += 1
        else: cnt['other'] += 1
    print(name, dict(cnt))
    print('  vector:', sorted([(v, k) for k, v in ops.items() if k.startswith('v_') and not k.startswith('v_mfma')], reverse=True)[:30])
    if '--seq' in sys.argv:  # the loop as one character per instruction: M matrix, v vector, r / w LDS read / write, L / S global load / store, | s_waitcnt, n s_nop, . scalar
        def ch(op):
            if op.startswith('v_mfma'): return 'M'
            if op.startswith('v_'): return 'v'
            if op.startswith('ds_read'): return 'r'
            if op.startswith('ds_'): return 'w'
            if op.startswith(('buffer_load', 'global_load')): return 'L'
            if op.startswith(('buffer_', 'global_')): return 'S'
            if op == 's_waitcnt': return '|'
            if op == 's_nop': return 'n'
            return '.' if op.startswith('s_') else '?'
        seq = ''.join(ch(l.split()[0]) for l in (x.strip() for x in best[1]) if l and l[0] not in ';.')
        for i in range(0, len(seq), 120): print('  ' + seq[i:i + 120])
    print('  other :', sorted([(v, k) for k, v in ops.items() if not k.startswith('v_')], reverse=True)[:14])
