#!/bin/bash
# round-3 closing run, third take (after the slab image pool and the growing lookahead): GPU tests, smoke, default bench,
# reader rates (tools/reader_rate.py: 3 passes, first-call costs included; tools/reader_trace.py: steady state over 10-20 passes)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c_final; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v "amdgpu.ids\|^Input\|^Out\|resolution\|Reading\|Success" | tail -3 > $O/smoke.txt
timeout 900 python bench.py 2>/dev/null > $O/bench_default.json
timeout 900 python tools/reader_rate.py 256 2>&1 | grep -av amdgpu.ids > $O/reader_rate.txt
bash tools/r03_run71.sh > $O/reader_steady.txt 2>&1
timeout 300 python tools/zero_copy_rate.py 2>&1 | grep -av "amdgpu.ids\|^Input\|^Out\|resolution\|Reading\|Success" > $O/zero_copy_rate.txt
grep -a "passed\|failed" $O/pytest.log | tail -2; cat $O/rc.txt; cat $O/smoke.txt | tail -2
python -c "
import json;d=json.loads(open('$O/bench_default.json').readline());r=d['roofline'];print(d['value'], d['ms_per_step'], r['frac'], r['kernel_ms'], r['frac_of_same_box_mix_ceiling'], d['cpu_baseline']['value'], d.get('parity'))"
grep -a "READER_RATE reader\|^--\|^==" $O/reader_rate.txt | tail -22; cat $O/reader_steady.txt
