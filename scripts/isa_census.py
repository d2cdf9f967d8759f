"""Static census of one kernel's ISA by basic block: matrix (v_mfma 16x16 / 4x4), other vector, LDS, scalar, waits, vector-memory
instruction counts.  Input: the kernel's part of a `hipcc -save-temps` .s file (from its label to s_endpgm):
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Igaunegf_amd/csrc -c gaunegf_amd/csrc/k_chain1d_rs.hip -o /tmp/rs.o -save-temps=obj
    awk '/^_ZN.*chain1d_rs_kernelILi51ELi3ELb1ELb0E.*:/,/s_endpgm/' /tmp/k_chain1d_rs-hip-amdgcn-amd-amdhsa-gfx950.s > /tmp/k51.s
    python scripts/isa_census.py /tmp/k51.s
Blocks with >= 3 matrix instructions or >= 60 vector instructions are listed (profiles/r05_chain_census.txt)."""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
blocks=[]; cur={'name':'entry','ins':[]}
for l in lines:
    t=l.strip()
    m=re.match(r'^(\.LBB[0-9_]+):',t)
    if not m and (not t or t.startswith(';') or t.startswith('.')): continue
    if m:
        blocks.append(cur); cur={'name':m.group(1),'ins':[]}; continue
    if t.endswith(':'): continue
    op=t.split()[0]
    cur['ins'].append(op)
blocks.append(cur)
def cls(op):
    if op.startswith('v_mfma'): return 'mfma16' if '16x16' in op else 'mfma4'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_'): return 'salu'
    if op.startswith('global_') or op.startswith('scratch_') or op.startswith('buffer_'): return 'vmem'
    return 'other'
tot={}
for b in blocks:
    c={}
    for op in b['ins']: c[cls(op)]=c.get(cls(op),0)+1
    b['c']=c
    for k,v in c.items(): tot[k]=tot.get(k,0)+v
print('static totals',tot)
for i,b in enumerate(blocks):
    c=b['c']; m=c.get('mfma16',0)+c.get('mfma4',0)
    if m>=3 or c.get('valu',0)>=60:
        print(f"{i:4d} {b['name']:12s} n={len(b['ins']):4d} mfma16={c.get('mfma16',0):3d} mfma4={c.get('mfma4',0):3d} valu={c.get('valu',0):4d} lds={c.get('lds',0):3d} salu={c.get('salu',0):3d} wait={c.get('wait',0):3d} nop={c.get('nop',0):3d} vmem={c.get('vmem',0):3d}  valu/mfma={c.get('valu',0)/max(m,1):.1f}")
