#!/bin/bash
# In-process A/B of experiment builds (mono_dataset_code_amd/variants/libmdc_hip_<name>.so, see build.py:build_variant)
# usage: tools/variants.sh "<variant names, '-' = default build>" <sweep.py args...>
cd $GRAFT_REPO_ROOT
VARS=$1; shift
LIBS=""
for v in $VARS; do
  if [ "$v" = "-" ]; then l=default; else l=mono_dataset_code_amd/variants/libmdc_hip_$v.so; fi
  LIBS="$LIBS${LIBS:+,}$l"
done
python tools/sweep.py --libs $LIBS "$@" 2>&1 | grep -v "amdgpu.ids"
