#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp7; mkdir -p $O
for r in fov affine affine128; do
  echo "== remap $r" >> $O/remap.txt
  timeout 600 python tools/sweep.py --libs default --frames 1024 --rounds 5 --iters 10 --rows 32,60 --fpb 0,64 --remap $r 2>&1 | grep -v amdgpu.ids >> $O/remap.txt
done
timeout 300 tools/bin/hbm_mix > $O/hbm_mix.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/fetch_mix -- $GRAFT_REPO_ROOT/tools/bin/hbm_mix > /dev/null 2>&1
for r in affine affine128; do
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/fetch_$r -- python $GRAFT_REPO_ROOT/tools/sweep.py --frames 1024 --rounds 1 --iters 2 --remap $r > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $O/fetch.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/exp7/fetch_*")):
    agg=collections.OrderedDict()
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]=="FETCH_SIZE":
                k=r["Kernel_Name"].split("(")[0][-70:]
                agg.setdefault(k,[]).append(float(r["Counter_Value"]))
    print(d)
    for k,v in agg.items(): print("  %-72s n=%d  read MB/launch %.1f" % (k, len(v), 2*1024*sum(v)/len(v)/1e6))
PY
rm -rf $O/fetch_*/
cat $O/remap.txt $O/hbm_mix.txt $O/fetch.txt
