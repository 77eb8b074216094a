#!/bin/bash
# round 3, run 5: strip kernel -- parity, soak, A/B against the direct kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_05; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "strip or two_stage or pyramid" ) > $O/pytest.txt 2>&1; grep -E "passed|failed|Error|error" $O/pytest.txt | tail -3
timeout 600 python tools/soak.py 36 > $O/soak.txt 2>&1; tail -3 $O/soak.txt
V=mono_dataset_code_amd/variants
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --two-stage 2,1 --nbuf 0,1 --pyramid 0,1 2>&1 | grep -v amdgpu.ids | tee $O/exp_strip.txt
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --two-stage 1 --fpb 8,16,32,64,128 --pyramid 1 2>&1 | grep -v amdgpu.ids | tee $O/exp_strip_fpb.txt
timeout 600 python tools/exp.py --out 1280x1024 --frames 128 --two-stage 2,1 --pyramid 0,1 --iters 20 2>&1 | grep -v amdgpu.ids | tee $O/exp_strip_128f.txt
for a in "--two-stage 2" "" "--nbuf 1"; do
timeout 400 python bench.py --workload pyramid --no-cpu-baseline --steps 20 --warmup 5 $a 2>/dev/null > $O/bench.json
python -c "
import json;d=json.loads(open('$O/bench.json').readline());r=d['roofline'];print('bench pyramid $a', r['frames_per_launch'], r['frac'], r['kernel_ms'], r['same_box_mix_ceiling']['ms_median'], r['frac_of_same_box_mix_ceiling'], r['kernel'], d['parity'])"
done
