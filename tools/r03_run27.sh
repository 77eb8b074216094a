#!/bin/bash
# round 3, run 27: gradient kernel v2 (128 x 8 pixels per workgroup, all loads up front) -- parity, DSO rate; store / load cache policies re-tested
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_27; mkdir -p $O
( time timeout 900 python -m pytest tests -x -q -m gpu -k "grad or dso" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -5
timeout 300 python tools/dso_rate.py 2>&1 | grep -v amdgpu.ids > $O/dso_rate.txt; tail -9 $O/dso_rate.txt
L=default,mono_dataset_code_amd/variants/libmdc_hip_aux0.so,mono_dataset_code_amd/variants/libmdc_hip_aux1.so,mono_dataset_code_amd/variants/libmdc_hip_aux3.so,mono_dataset_code_amd/variants/libmdc_hip_aux16.so,mono_dataset_code_amd/variants/libmdc_hip_aux17.so,mono_dataset_code_amd/variants/libmdc_hip_aux18.so,mono_dataset_code_amd/variants/libmdc_hip_loadnt.so
timeout 600 python tools/exp.py --out 640x480 --frames 4096 --pyramid 0 --libs $L --rounds 4 --iters 3 2>&1 | grep -v amdgpu.ids > $O/aux_headline.txt; cat $O/aux_headline.txt
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --pyramid 1 --libs $L --rounds 4 --iters 3 2>&1 | grep -v amdgpu.ids > $O/aux_pyramid.txt; cat $O/aux_pyramid.txt
