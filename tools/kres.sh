#!/bin/bash
# register / spill table of every kernel in a HIP source: tools/kres.sh <file.hip> [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $f -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
rows=[];cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur={'name':m.group(1)};rows.append(cur);continue
    for k,pat in (('v',r' VGPRs: (\d+)'),('a',r'AGPRs: (\d+)'),('sp',r'VGPR Spill: (\d+)'),('scr',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)')):
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=m.group(1)
    if 'error' in l: print(l,end='')
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.split('\n') if rows else []
for r,n in zip(rows,names):
    n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'\(.*','',n)
    print(f\"{n:60s} v={r.get('v','?'):>3} a={r.get('a','?'):>3} spill={r.get('sp','?'):>4} scratch={r.get('scr','?'):>5} occ={r.get('occ','?')} lds={r.get('lds','?')}\")
"
rm -f /tmp/kres_$$.o
