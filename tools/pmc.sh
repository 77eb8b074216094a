#!/bin/bash
# PMC passes over one kernel configuration (run on the GPU box through gpurun).
# Each pass is its own rocprofv3 run with --kernel-trace only, as the guide prescribes.
# usage: tools/pmc.sh <outdir-under-gpurun_out> -- <sweep.py args>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT" \
         "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- python $GRAFT_REPO_ROOT/tools/sweep.py --rounds 1 --iters 2 "$@" > $OUT/pass$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "synth" in k or "rocclr" in k: continue
        agg[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as w:
    for k, cs in agg.items():
        w.write(k + "\n")
        for c, v in sorted(cs.items()):
            w.write("  %-32s mean/dispatch %.6g  (n=%d)\n" % (c, sum(v) / len(v), len(v)))
print(open(out + "/summary.txt").read())
PY
