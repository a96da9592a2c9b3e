set -e
tools/r3_profile_const256.sh r3
tools/collect_profiles.sh r3_var var
tools/collect_profiles.sh r3_var_l1 var --ell 1
echo ok
