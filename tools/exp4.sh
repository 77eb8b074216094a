#!/bin/bash
# GPU experiment batch 4: exact per-row windows: parity, timing, fabric read bytes
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
timeout 600 python tools/sweep.py --frames 1024 --rounds 5 --iters 10 --rows 32,60,64 --order 0,1 --fpb 0,32,64 > $O/sweep_shapes.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for cfg in "32 0 0" "32 0 64" "60 0 0" "60 1 64"; do
  set -- $cfg
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/fetch_$1_$2_$3 -- python $GRAFT_REPO_ROOT/tools/sweep.py --frames 1024 --rounds 1 --iters 2 --rows $1 --order $2 --fpb $3 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $O/fetch.txt
import csv, glob
for d in sorted(glob.glob("gpurun_out/exp4/fetch_*")):
    v=[]
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "remap_tiled" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": v.append(float(r["Counter_Value"]))
    if v: print(d.split("/")[-1],"(rows_order_fpb) FETCH_SIZE KiB mean",sum(v)/len(v),"-> read MB/frame", 2*1024*sum(v)/len(v)/1024/1e6, "n",len(v))
PY
rm -rf $O/fetch_*/
cat $O/pytest.txt $O/sweep_shapes.txt $O/fetch.txt
