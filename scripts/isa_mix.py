import re,sys,collections
f=sys.argv[1]
L=open(f).read().split('\n')
# basic blocks by label
labels={}
for i,l in enumerate(L):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
# find backward branches => loops
loops=[]
for i,l in enumerate(L):
    m=re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)',l) or re.match(r'\s+s_branch\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        loops.append((labels[m.group(1)],i))
def classify(op):
    if op.startswith('v_'):
        if re.search(r'_f64|_dpp|f64_f32|f32_f64',op) or 'dpp' in op: return 'V4'
        if op.startswith('v_pk_'): return 'V4'
        return 'V2'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith('buffer_') or op.startswith('global_') or op.startswith('scratch_') or op.startswith('flat_'): return 'VMEM'
    if op.startswith('s_'): return 'S'
    return 'other'
for a,b in loops:
    cnt=collections.Counter(); ops=collections.Counter()
    for l in L[a:b+1]:
        m=re.match(r'\s+([a-z_0-9]+)\s*(.*)',l)
        if not m or l.strip().startswith(('.',';')): continue
        op=m.group(1); rest=m.group(2)
        c=classify(op)
        if c=='V2' and ('row_' in rest or 'wave_' in rest or 'quad_perm' in rest): c='V4'; op+='_dpp'
        cnt[c]+=1; ops[op]+=1
    print(f"loop lines {a}-{b} ({b-a}):",dict(cnt))
    if len(sys.argv)>2: print('   ',ops.most_common(40))
