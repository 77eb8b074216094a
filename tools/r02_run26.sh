#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run26; mkdir -p $O
python - > $O/mk.txt 2>&1 <<'PY'
import sys; sys.argv=['x','512']
exec(open('tools/reader_rate.py').read().split("for kind in")[0])
for kind in ("folder_png","zip_jpg"):
    d,avg=make(kind); print("SEQ",kind,d)
PY
for kind in folder_png zip_jpg; do
  D=$(grep "SEQ $kind" $O/mk.txt | awk '{print $3}')
  for T in 0 12 16 20; do
    MDC_READER_TRACE=1 MDC_READER_THREADS=$T oracle/_ref/reader_rate_fast $D 1111 3 batch 2>&1 | grep "READER_RATE reader\|getImages" | tail -2 | tr '\n' ' '; echo " [$kind T=$T]"
  done
done
timeout 300 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps(d['cpu_baseline'], indent=0))"
