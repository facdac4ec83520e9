#!/bin/bash
# usage: tools/variant_sweep.sh <out.txt> "<variant names>" <inc_slices_sweep.py args...>
# Same-box A/B of variant libraries (tools/build_variant.py -> variants/libdisenlink_hip_<name>.so): the incidence-slicing
# sweep of the one-pass training scorer once per variant ("product" = the library as built).
out=$1; names=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
: > "$out"
for n in $names; do
  echo "=== variant $n" >> "$out"
  if [ "$n" = product ]; then
    python3 "$root/tools/inc_slices_sweep.py" "$@" >> "$out" 2>&1 || exit 1
  else
    DL_LIB_PATH="$root/variants/libdisenlink_hip_$n.so" python3 "$root/tools/inc_slices_sweep.py" "$@" >> "$out" 2>&1 || exit 1
  fi
done
