# scratch driver for gpurun calls: put the commands of the next GPU run here
set -e
tools/r3_final_profiles.sh
