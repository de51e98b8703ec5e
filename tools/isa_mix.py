#!/usr/bin/env python3
"""Count instructions in the hottest loop (or whole body) of each kernel in a hipcc -save-temps .s file."""
import re, sys, collections
s = open(sys.argv[1]).read()
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
filt = sys.argv[3] if len(sys.argv) > 3 else ''
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if filt and filt not in name: continue
    loops = re.findall(r'(\.LBB\d+_\d+):.*?\n(.*?)s_cbranch_\w+ \1\n', body, re.S)
    text = max(loops, key=lambda x: len(x[1]))[1] if loops else body
    ins = [l.split()[0] for l in text.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    c = collections.Counter(ins)
    v = sum(n for k, n in c.items() if k.startswith('v_'))
    mul = sum(n for k, n in c.items() if 'mul' in k or 'mad' in k)
    print('%-28s %s VALU=%.1f mul/mad=%.1f total=%.1f  ' % (name, 'loop' if loops else 'body', v / div, mul / div, len(ins) / div),
          {k: round(n / div, 2) for k, n in c.most_common(12)})
