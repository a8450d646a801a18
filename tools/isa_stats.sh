#!/bin/bash
# usage: tools/isa_stats.sh <file.hip> <kernel-name-substring>  -> register/scratch usage and MFMA spacing of one kernel
set -e
SRC=$(readlink -f "$1"); K="$2"; D=$(mktemp -d); cd "$D"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/root/repo/include -I/root/repo/cognn_amd/csrc -c "$SRC" -o x.o -save-temps -Rpass-analysis=kernel-resource-usage 2> remarks.txt || { cat remarks.txt | tail; exit 1; }
grep -A12 "Function Name.*$K" remarks.txt | grep -E "Function Name|VGPRs:|AGPRs:|Scratch|Occupancy|LDS" | sed 's/.*remark: *//' | head -8
python3 - "$K" <<'PY'
import glob,sys
K=sys.argv[1]
S=[f for f in glob.glob('*gfx950*.s')][0]
txt=open(S).read().split('\n')
c=[i for i,l in enumerate(txt) if l.startswith('_Z') and ':' in l and K in l.split(':')[0]]
i0=c[0]; end=[i for i in range(i0,len(txt)) if 's_endpgm' in txt[i]][0]
body=txt[i0:end+1]
open('/tmp/kernel.s','w').write('\n'.join(body))
mf=[i for i,l in enumerate(body) if 'v_mfma' in l]
print("lines",len(body),"mfma",len(mf),"scratch ops",sum('scratch_' in l for l in body),"barriers",sum('s_barrier' in l for l in body),
      "vmcnt waits",sum('vmcnt' in l for l in body))
if mf:
    gaps=[mf[i+1]-mf[i]-1 for i in range(len(mf)-1)]
    print("instructions between consecutive MFMAs:",gaps)
PY
