#!/bin/bash
run() {
  echo "== $*"
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('it/s',d['value'],'ms',d['ms_per_step'],'GB/s',d['iteration_GBps'],'frac',d['iteration_frac_of_hbm_peak'],'valid',d['valid'])
print('   '+'  '.join('%s %.0f'%(k,v['GBps']) for k,v in d['kernels'].items()))
"
}
run
run --ell 1
run --workload const
run --workload const --ell 1
run --grid 256
run --grid 256 --ell 1
run --grid 256 --workload const
run --grid 256 --workload const --ell 1
