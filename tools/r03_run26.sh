#!/bin/bash
# round 3, run 26: write patterns with compile-time tile shapes (is the 5.6 TB/s of tiles the memory system or wpat's division?)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_26; mkdir -p $O
timeout 300 tools/bin/wpat2 640 480 4096 > $O/wpat2_640.txt 2>&1; cat $O/wpat2_640.txt
timeout 300 tools/bin/wpat2 1280 1024 1024 > $O/wpat2_1280.txt 2>&1
