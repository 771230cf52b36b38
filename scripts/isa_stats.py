import re,sys,collections
def stats(path, kern='rtow_trace_fastILi3ELb1ELb0'):
    lines=open(path).read().split('\n')
    start=None
    for i,l in enumerate(lines):
        if l.startswith('_ZN4rtow12_GLOBAL__N_115'+kern+'EEEvNS_11TraceParamsE:'):
            start=i;break
    depth=0; cnt=collections.Counter(); valu=collections.Counter(); scr=collections.Counter(); mov=collections.Counter()
    for l in lines[start:]:
        if 's_endpgm' in l: break
        m=re.search(r'Depth=(\d+)',l)
        if l.startswith('.LBB') or l.startswith('; %bb'):
            depth=int(m.group(1)) if m else 0
            continue
        t=l.strip()
        if not t or t.startswith(';') or t.startswith('.'): continue
        op=t.split()[0]
        cnt[depth]+=1
        if op.startswith('v_'): valu[depth]+=1
        if op.startswith('scratch_'): scr[depth]+=1
        if op.startswith('v_mov') or op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_accvgpr'): mov[depth]+=1
    return cnt,valu,scr,mov
for p in sys.argv[1:]:
    c,v,s,m=stats(p)
    print(p, 'total',sum(c.values()), {d:(c[d],v[d],s[d],m[d]) for d in sorted(c)})
