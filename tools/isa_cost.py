#!/usr/bin/env python3
"""VALU issue-cycle estimate of the hot loop of each kernel in a hipcc -S file, from the measured steady-state costs
of tools/ubench_issue.hip (profiles/r02_ubench_issue_costs.txt): sum over the loop's instructions of cycles per
wave-instruction with the SIMD saturated.   usage: isa_cost.py file.s [name-filter]"""
import collections, re, sys
COST = {  # shader cycles per wave-instruction per SIMD at 4 waves/SIMD
    'v_mul_lo_u32': 4.04, 'v_mul_hi_u32': 4.04, 'v_mad_u64_u32': 4.10, 'v_lshl_add_u64': 4.29, 'v_add3_u32': 4.05,
    'v_add_u32': 2.16, 'v_sub_u32': 2.16, 'v_subrev_u32': 2.16, 'v_add_co_u32': 4.06, 'v_addc_co_u32': 4.15, 'v_sub_co_u32': 4.06,
    'v_subrev_co_u32': 4.06, 'v_subb_co_u32': 4.15, 'v_subbrev_co_u32': 4.15, 'v_lshlrev_b32': 4.04, 'v_lshrrev_b32': 2.18, 'v_and_b32': 2.16,
    'v_or_b32': 2.16, 'v_xor_b32': 2.16, 'v_bfe_u32': 4.04, 'v_and_or_b32': 4.05, 'v_lshl_add_u32': 4.04, 'v_lshl_or_b32': 4.05,
    'v_cndmask_b32': 4.1, 'v_cmp': 4.06, 'v_mov_b32': 2.09, 'v_mov_b64': 4.2, 'v_mad_u32_u24': 4.04, 'v_mul_u32_u24': 4.04,
    'v_alignbit_b32': 4.04, 'v_lshlrev_b64': 4.04, 'v_lshrrev_b64': 4.04, 'v_readfirstlane_b32': 4.0, 'v_accvgpr': 2.1, 'v_or3_b32': 4.05,
    'v_not_b32': 2.16, 'v_ashrrev_i32': 2.18, 'v_add_lshl_u32': 4.04, 'v_xad_u32': 4.05, 'v_min_u32': 2.16, 'v_max_u32': 2.16,
}
def cost(op):
    op = re.sub(r'_(e32|e64|sdwa|dpp)$', '', op)
    if op.startswith('v_cmp'): return COST['v_cmp']
    return COST.get(op)
def main():
    lines = open(sys.argv[1]).read().split('\n')
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    kern, cur = {}, None
    for l in lines:
        m = re.match(r'^(_Z\w+):', l)
        if m: cur = m.group(1); kern[cur] = []; continue
        if cur is not None:
            kern[cur].append(l)
            if 's_endpgm' in l: cur = None
    for name, body in kern.items():
        if filt and filt not in name: continue
        labels, best = {}, None
        for i, l in enumerate(body):
            m = re.match(r'^(\.LBB\d+_\d+):', l)
            if m: labels[m.group(1)] = i
            m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels:
                seg = [x.split()[0] for x in body[labels[m.group(1)]:i] if x.strip() and not x.strip().startswith((';', '.'))]
                if best is None or len(seg) > len(best): best = seg
        if not best: continue
        c = collections.Counter(best)
        tot, unknown, nv = 0.0, collections.Counter(), 0
        for op, n in c.items():
            if op.startswith('v_'):
                nv += n
                k = cost(op)
                if k is None: unknown[op] += n; k = 4.1
                tot += k * n
        other = {k: n for k, n in c.items() if not k.startswith('v_')}
        print('%s\n   loop: %d instr, %d VALU, VALU issue estimate %.0f cycles/wave  (x4 waves/SIMD = %.0f per polynomial)' % (name[:60], len(best), nv, tot, 4 * tot))
        print('   VALU mix:', {k: n for k, n in c.most_common(40) if k.startswith('v_')})
        print('   other   :', dict(sorted(other.items(), key=lambda x: -x[1])[:14]))
        if unknown: print('   uncosted (taken as 4.1):', dict(unknown))


if __name__ == '__main__':
    main()
