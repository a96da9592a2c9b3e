run() { KRY_PROFILE=0 timeout -k 10 150 python bench.py --no-cpu --steps 30 --warmup 4 "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:70], d["value"], d["valid"])'; }
run --workload var
run --workload const
run --workload mixed
run --workload var --ell 1
run --workload const --ell 1
run --workload var --grid 256
run --workload const --grid 256 --ell 1 --steps 200
run --workload mixed --grid 256
run --workload const --grid 256 --steps 100
