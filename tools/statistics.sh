#!/bin/bash
# Size sweep in the protocol of the reference's BiCGStab3D_CL/statistics.sh: every test 1..5 on
# a list of grid sizes, RUNS times each, one tab-separated --stats row per run appended to FILE.
#   tools/statistics.sh FILE [RUNS] [SIZES...]
DRIVER="$(dirname "$0")/../krylov_amd/driver/bicgstab"
[ $# -ge 1 ] || { echo "Usage: $0 FILE [RUNS] [SIZES...]"; exit 1; }
FILE=$1; RUNS=${2:-5}; shift; shift
SIZES=${*:-"$(seq 80 4 276)"}
for t in 1 2 3 4 5; do
	for n in $SIZES; do
		for _ in $(seq 1 "$RUNS"); do
			"$DRIVER" --gpu --stats -t "$t" -n "$n" >> "$FILE" || { echo "solver failed: test $t, n = $n"; exit 1; }
		done
	done
done
