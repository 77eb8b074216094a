#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_61; mkdir -p $O
( time timeout 900 python -m pytest tests/test_reader.py -x -q -m gpu -k "huffman or jpeg" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -8
for q in 90 98 75; do Q=$q timeout 300 python tools/huffman_rate.py 2>&1 | grep -v amdgpu.ids | grep "KB per\|n   1\|n  64\|n 256"; done
