#!/bin/bash
# A/B of the experiment builds (separate processes; effects < ~3% need an in-process A/B instead)
# usage: tools/variants.sh "<variant names, '-' = default build>" <sweep.py args...>
cd $GRAFT_REPO_ROOT
VARS=$1; shift
for v in $VARS; do
  lib=""; [ "$v" != "-" ] && lib="--lib $GRAFT_REPO_ROOT/mono_dataset_code_amd/variants/libmdc_hip_$v.so"
  echo "== variant: $v"
  python tools/sweep.py "$@" $lib 2>&1 | grep "^tiled\|^gather\|^auto\|Error\|error"
done
