#!/bin/bash
# round 3, run 15: GPU JPEG stage -- parity, reader rates (JPEG only, host vs GPU inverse DCT)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_15; mkdir -p $O
( time timeout 900 python -m pytest tests/test_reader.py -x -q -m gpu ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -5
MDC_RATE_KINDS=zip_jpg timeout 600 python tools/reader_rate.py 512 > $O/reader_rate.txt 2>&1; grep "READER_RATE\|==\|^--" $O/reader_rate.txt | tail -8
for t in 8 32; do echo "decode threads $t"; MDC_RATE_KINDS=zip_jpg MDC_READER_THREADS=$t timeout 300 python tools/reader_rate.py 512 2>&1 | grep "batch\|^--"; done
