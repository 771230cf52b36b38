#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a `hipcc -S` listing.

usage: isa_blocks.py file.s [kernel-substring] [--dump LABEL]
Classes: mov (v_mov/v_accvgpr/v_readlane/v_writelane/v_readfirstlane), sel (v_cndmask), cmp (v_cmp*),
f64 (v_*_f64), f32 (v_*_f32 / v_fma_mix), int (other VALU), lds (ds_*), vmem (global_/scratch_/buffer_/flat_),
salu (s_* except waitcnt/branch/nop), wait (s_waitcnt), br (s_cbranch/s_branch), other.
"""
import re, sys, collections

def classify(op):
    if op.startswith('v_'):
        if op.startswith(('v_mov', 'v_accvgpr', 'v_readlane', 'v_writelane', 'v_readfirstlane', 'v_swap')): return 'mov'
        if op.startswith('v_cndmask'): return 'sel'
        if op.startswith('v_cmp'): return 'cmp'
        if '_f64' in op: return 'f64'
        if '_f32' in op or 'fma_mix' in op or '_f16' in op: return 'f32'
        return 'int'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith(('s_cbranch', 's_branch')): return 'br'
    if op.startswith('s_nop') or op.startswith('s_endpgm'): return 'other'
    if op.startswith('s_'): return 'salu'
    return 'other'

CLASSES = ['mov', 'sel', 'cmp', 'f64', 'f32', 'int', 'lds', 'vmem', 'salu', 'wait', 'br', 'other']

def main():
    path = sys.argv[1]
    kern = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else 'rtow_trace_fastILi3ELb1ELb0ELi1E'
    dump = sys.argv[sys.argv.index('--dump') + 1] if '--dump' in sys.argv else None
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if kern in l and l.rstrip().endswith(':') or (kern in l and l.startswith('_Z') and ': ' in l):
            start = i
            break
    if start is None:
        sys.exit('kernel not found')
    blocks = []  # (label, depth, counter, [lines])
    cur = ['entry', 0, collections.Counter(), []]
    for l in lines[start + 1:]:
        t = l.strip()
        if l.startswith('.LBB') or l.startswith('; %bb'):
            m = re.search(r'Depth=(\d+)', l)
            blocks.append(cur)
            lab = l.split(':')[0].lstrip('; ')
            cur = [lab, int(m.group(1)) if m else 0, collections.Counter(), []]
            continue
        if not t or t.startswith(';') or t.startswith('.'):
            if t.startswith(';'):
                cur[3].append(l)
            continue
        op = t.split()[0]
        cur[2][classify(op)] += 1
        cur[3].append(l)
        if op == 's_endpgm':
            break
    blocks.append(cur)
    if dump:
        for b in blocks:
            if b[0] == dump:
                print('\n'.join(b[3]))
        return
    tot = collections.Counter()
    bydepth = collections.defaultdict(collections.Counter)
    print('%-14s %2s %5s | ' % ('block', 'd', 'n') + ' '.join('%4s' % c for c in CLASSES))
    for lab, d, c, _ in blocks:
        n = sum(c.values())
        if n == 0:
            continue
        tot.update(c)
        bydepth[d].update(c)
        if n >= 12:
            print('%-14s %2d %5d | ' % (lab, d, n) + ' '.join('%4d' % c[k] for k in CLASSES))
    for d in sorted(bydepth):
        c = bydepth[d]
        print('%-14s %2d %5d | ' % ('DEPTH', d, sum(c.values())) + ' '.join('%4d' % c[k] for k in CLASSES))
    print('%-14s %2s %5d | ' % ('TOTAL', '', sum(tot.values())) + ' '.join('%4d' % tot[k] for k in CLASSES))

main()
