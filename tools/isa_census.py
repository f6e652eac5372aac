#!/usr/bin/env python3
"""Static instruction census of the loops of one kernel in hipcc's assembly output:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S --cuda-device-only -o k.s file.hip
   python tools/isa_census.py k.s <substring of the mangled kernel name>
For every loop (a backward branch to a label): FP64 / other vector / scalar / branch / LDS / memory instructions
between the loop's header and its last backward branch.  Static counts: a block under a branch that is rarely
taken counts like any other."""
import re,collections,sys
src=open(sys.argv[1]).read().split('\n')
pat=sys.argv[2]
# find kernel by mangled substring
start=None
for i,l in enumerate(src):
    if re.match(r'^_Z\S*'+pat+r'\S*:',l): start=i;break
end=start
while not src[end].strip().startswith('s_endpgm'): end+=1
lines=src[start:end+1]
blocks=[];cur=['entry',0,[]];blocks.append(cur)
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m:
        cur=[m.group(1),i,[]];blocks.append(cur);continue
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur[2].append(t.split()[0])
lab={b[0]:k for k,b in enumerate(blocks)}
def cls(op):
    if op.startswith('v_') and '_f64' in op: return 'f64'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_','buffer_','scratch_','flat_')): return 'vmem'
    return 'other'
# loops: back edges
loops=[]
for i,l in enumerate(lines):
    m=re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in lab:
        tl=blocks[lab[m.group(1)]][1]
        if tl<i: loops.append((tl,i))
# merge by header
hdr={}
for tl,i in loops: hdr[tl]=max(hdr.get(tl,0),i)
for tl,i in sorted(hdr.items()):
    c=collections.Counter()
    for b in blocks:
        if tl<=b[1]<=i:
            for op in b[2]: c[cls(op)]+=1
    print('loop lines %d-%d (%d lines):'%(tl,i,i-tl),dict(c))
m=re.search(r'\.vgpr_count:\s+(\d+)','\n'.join(src[end:end+400]))
