#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp5; mkdir -p $O
timeout 600 tools/variants.sh "- v1 skipstore skipload skipboth" --frames 1024 --rounds 7 --iters 10 > $O/ab.txt 2>&1
cat $O/ab.txt
