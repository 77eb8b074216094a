#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_38; mkdir -p $O
timeout 600 python tools/huffman_rate.py 2>&1 | grep -v amdgpu.ids > $O/huffman_rate.txt; cat $O/huffman_rate.txt
