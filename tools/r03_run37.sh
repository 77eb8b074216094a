#!/bin/bash
# round 3, run 37: reader with the Huffman decoding on the device -- tests, rates
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_37; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_reader.py tests/test_dropin.py -x -q -m gpu ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -12
MDC_RATE_KINDS=zip_jpg timeout 900 python tools/reader_rate.py 512 2>&1 | grep -v amdgpu.ids > $O/reader_rate_jpg.txt; cat $O/reader_rate_jpg.txt
MDC_RATE_KINDS=zip_jpg MDC_READER_TRACE=1 timeout 900 python tools/reader_rate.py 512 2>&1 | grep -i "getImages:" | tail -6
