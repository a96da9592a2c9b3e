#!/bin/bash
run() {
  echo "== $*"
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
s=sum(v['avg_ms'] for v in d['kernels'].values())
print('it/s',d['value'],'ms',d['ms_per_step'],'sumk %.4f'%s,'GB/s',d['iteration_GBps'],'frac',d['iteration_frac_of_hbm_peak'],'valid',d['valid'])
print('   '+'  '.join('%s %.0f'%(k,v['GBps']) for k,v in d['kernels'].items()))
"
}
timeout -k 10 300 python -u -m pytest tests/test_gpu_bitwise.py tests/test_gpu_slabs.py -m gpu -q -x --timeout 300 --timeout-method thread 2>&1 | tail -2
run --grid 256
run --grid 256 --ell 1
run --grid 256 --workload const --ell 1
run --grid 128
run
