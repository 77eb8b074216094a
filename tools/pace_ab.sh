#!/bin/bash
# pacing of the tiled kernel (MDC_TILE_SYNC: a check every N frames, 0 = off; MDC_TILE_SYNC_SLEEP; MDC_TILE_SYNC_MAXLEAD): the headline launch
# per setting, one process each (the settings are read once per process), on buffers from the product's allocator
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pace}
mkdir -p $OUT
for cfg in ${PACE_CFGS:-0:2:12 2:2:12 4:2:12 8:2:12 0:2:12 2:1:12 2:4:12 4:4:12 1:2:12 4:2:4}; do
  IFS=: read -r every sleep maxlead <<< "$cfg"
  MDC_TILE_SYNC=$every MDC_TILE_SYNC_SLEEP=$sleep MDC_TILE_SYNC_MAXLEAD=$maxlead timeout 300 python tools/placed_probe.py auto 1 ${PACE_FRAMES:-4096} 2>&1 | grep -a "PLACED\|rror\|fault" | sed "s/^PLACED/PACE every $every sleep $sleep maxlead $maxlead:/" | cut -c1-250 >> $OUT/pace_ab.txt
done
cat $OUT/pace_ab.txt
