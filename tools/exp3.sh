#!/bin/bash
# GPU experiment batch 3: v2 tiled kernel (buffer addressing + LDS-DMA): parity, tile shapes, diagnosis variants
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp3; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
timeout 600 python tools/sweep.py --frames 1024 --rounds 5 --iters 10 --rows 32,60,64,16 --order 0,1 --fpb 0,16,32,64 > $O/sweep_shapes.txt 2>&1
timeout 300 tools/variants.sh "- skipstore skipload skipboth plainst ntload" --frames 1024 --rounds 5 --iters 20 > $O/variants.txt 2>&1
cat $O/pytest.txt $O/sweep_shapes.txt $O/variants.txt
