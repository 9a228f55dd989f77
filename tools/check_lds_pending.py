import re,sys
def regs(tok):
    m=re.match(r'v\[(\d+):(\d+)\]',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r'v(\d+)$',tok)
    if m: return {int(m.group(1))}
    return set()
pending=[]  # list of (regset, line)
flags=0
for ln,line in enumerate(open(sys.argv[1]),1):
    s=line.split(';')[0].strip()
    if not s or s.startswith('.') or s.endswith(':'): continue
    parts=s.replace(',',' ').split()
    op=parts[0]; ops=parts[1:]
    if op=='s_waitcnt':
        m=re.search(r'lgkmcnt\((\d+)\)',s)
        if m:
            n=int(m.group(1)); pending=pending[len(pending)-n:] if n>0 else []
        continue
    if op.startswith('s_') : continue
    if op.startswith('ds_read'):
        d=regs(ops[0]); src=set().union(*[regs(o) for o in ops[1:]]) if len(ops)>1 else set()
        for p,l in pending:
            if p&src: print("READ-BEFORE-WAIT",ln,s,"pending from",l); flags+=1
            if p&d: print("WAW on pending",ln,s,"pending from",l); flags+=1
        pending.append((d,ln)); continue
    # other instructions: sources = all operands except first (dest) for most; stores: all
    isstore = 'store' in op or op.startswith('ds_write') or op.startswith('global_load_lds')
    srcs=ops if isstore else ops[1:]
    dst=set() if isstore else (regs(ops[0]) if ops else set())
    if op.startswith('v_mfma'): srcs=ops[1:]
    src=set().union(*[regs(o) for o in srcs]) if srcs else set()
    for p,l in pending:
        if p&src: print("READ-BEFORE-WAIT",ln,s,"| pending ds_read at",l); flags+=1
        if p&dst: print("WRITE-OVER-PENDING",ln,s,"| pending ds_read at",l); flags+=1
print("flags:",flags)
