#!/bin/bash
# GPU experiment batch 1 (run via gpurun): tests, ceilings, diagnosis variants, placement/fpb sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp1; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
timeout 300 tools/bin/hbm_mix > $O/hbm_mix.txt 2>&1
timeout 300 tools/variants.sh "- skipstore skipload skipboth plainst" --frames 1024 --rounds 4 --iters 5 > $O/variants.txt 2>&1
timeout 600 python tools/sweep.py --frames 1024 --rounds 4 --iters 5 --order 0,1,2 --fpb 0,16,22,43,64,103,205 > $O/sweep_order_fpb.txt 2>&1
timeout 300 python tools/sweep.py --frames 1024 --rounds 4 --iters 5 --rows 16 --order 0,1,2 --fpb 0,16,64 > $O/sweep_rows16.txt 2>&1
# fabric-level read bytes per placement (FETCH_SIZE x2 = bytes)
cd /tmp && export TMPDIR=/tmp
for ord in 0 1 2; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/fetch_ord$ord -- python $GRAFT_REPO_ROOT/tools/sweep.py --frames 1024 --rounds 1 --iters 2 --order $ord > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $O/fetch_by_order.txt
import csv, glob
for o in (0,1,2):
    v=[]
    for f in glob.glob("gpurun_out/exp1/fetch_ord%d/**/*counter_collection.csv"%o, recursive=True):
        for r in csv.DictReader(open(f)):
            if "remap_tiled" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": v.append(float(r["Counter_Value"]))
    if v: print("order",o,"FETCH_SIZE KiB mean",sum(v)/len(v),"-> read MB/frame", 2*1024*sum(v)/len(v)/1024/1e6, "n",len(v))
PY
rm -rf $O/fetch_ord*
cat $O/pytest.txt $O/hbm_mix.txt $O/variants.txt $O/sweep_order_fpb.txt $O/sweep_rows16.txt $O/fetch_by_order.txt
