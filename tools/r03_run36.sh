#!/bin/bash
# round 3, run 36: Huffman decoding on the device -- parity with the host decoder's records
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_36; mkdir -p $O
( time timeout 900 python -m pytest tests/test_reader.py -x -q -m gpu -k "huffman" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -12
