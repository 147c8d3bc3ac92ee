#!/bin/bash
# usage: scratch/resusage.sh <file.hip> [grep pattern]  -> one line per kernel: name vgprs spills scratch lds occupancy
f=$1; pat=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I /root/repo/include -I /root/repo/ladder_latent_data_distribution_modelling_amd/csrc -c $f -o /tmp/resusage.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur={'name':m.group(1)}; rows.append(cur); continue
    for k,pat in (('vgpr',r' VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('spill',r'VGPR Spill: (\d+)'),('sspill',r'SGPRs Spill: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)')):
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=m.group(1)
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()[:90]
    print(n, {k:v for k,v in r.items() if k!='name'})
" | grep -E "$pat"
