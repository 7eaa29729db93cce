#!/usr/bin/env python3
"""Static VALU instruction mix per kernel of a gfx950 assembly listing (hipcc -save-temps): totals, the 4-cycle class, LDS, scratch.
    python tools/isa_count.py listing.s [name-substring]"""
import sys, re, collections
FOUR = re.compile(r'v_(mad_u64|mad_i64|add_co|addc|sub_co|subb|subrev_co|subbrev|alignbit|mul_lo|mul_hi|lshl_add_u64)')
def main():
    lines = open(sys.argv[1]).read().split('\n')
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    cur = None; res = {}
    for line in lines:
        s = line.strip()
        m = re.match(r'^(_Z\w+):', line)
        if m and want in m.group(1):
            cur = m.group(1); res[cur] = collections.Counter(); continue
        if cur is None: continue
        if s.startswith('s_endpgm'): cur = None; continue
        if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'): continue
        res[cur][s.split()[0]] += 1
    for k, ops in res.items():
        valu = sum(v for o, v in ops.items() if o.startswith('v_'))
        four = sum(v for o, v in ops.items() if FOUR.match(o))
        print(k[:70], 'total', sum(ops.values()), 'valu', valu, 'four-cycle-class', four, 'two-cycle-class', valu - four,
              'ds', sum(v for o, v in ops.items() if o.startswith('ds_')), 'scratch', sum(v for o, v in ops.items() if o.startswith('scratch_')))
        print('    ', ' '.join('%s:%d' % (o, v) for v, o in sorted(((v, o) for o, v in ops.items() if o.startswith('v_')), reverse=True)[:18]))
main()
