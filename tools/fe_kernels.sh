#!/bin/bash
# fe_k.sh <lib suffix or "">: bench --lean, print front-end + per-kernel times
L=$1
if [ -n "$L" ]; then export SSX_LIB=$PWD/ssvio_amd/libssx.so.$L; fi
python bench.py --lean --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$L', 'value', d['value'], 'fe', d['frontend']['value'], ' '.join(f\"{n}={v['ms_per_step']:.3f}\" for n,v in k.items() if v['part']=='frontend'))"
