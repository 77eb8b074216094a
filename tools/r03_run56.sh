#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_56; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_reader.py tests/test_gpu_parity.py tests/test_dropin.py -x -q -m gpu -k "reader or zero_copy or frames_host or dropin or jpeg or huffman or get_image" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -8
MDC_RATE_KINDS=zip_jpg,zip_png timeout 900 python tools/reader_rate.py 256 2>&1 | grep -v amdgpu.ids | grep "batch\|stage\|==" | tail -12
timeout 300 python tools/zero_copy_rate.py 2>&1 | grep "zero copy o" 
