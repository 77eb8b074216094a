#!/bin/bash
# PC sampling of the fused kernel (stochastic: stall reasons per instruction)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r39; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 -d $O/stoch --output-format csv -- python $GRAFT_REPO_ROOT/tools/pc_sample_target.py fused 40 > $O/stoch.log 2>&1
echo "stochastic rc=$?" | tee -a $O/rc.txt
tail -5 $O/stoch.log
find $O/stoch -type f | head; 
if ! find $O/stoch -name "*pc_sampling*" | grep -q .; then
timeout 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 10 -d $O/trap --output-format csv -- python $GRAFT_REPO_ROOT/tools/pc_sample_target.py fused 40 > $O/trap.log 2>&1
echo "host_trap rc=$?" | tee -a $O/rc.txt
tail -5 $O/trap.log; find $O/trap -type f | head
fi
du -sh $O
