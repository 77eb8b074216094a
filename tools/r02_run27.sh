#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['cpu_baseline']; print(c['value'], c['cores'], c['sample'], c.get('other_thread_count'), c['one_thread_as_shipped'], c['unmap_only'])"
D=$(python - 2>/dev/null <<'PY' | tail -1
import sys; sys.argv=['x','512']
exec(open('tools/reader_rate.py').read().split("for kind in")[0])
d,avg=make("folder_png"); print(d)
PY
)
for T in 0 16; do MDC_READER_TRACE=1 MDC_READER_THREADS=$T oracle/_ref/reader_rate_fast $D 1111 3 batch 2>&1 | grep "READER_RATE reader\|getImages" | tail -2 | tr '\n' ' '; echo " [T=$T]"; done
oracle/_ref/reader_rate_fast $D 1111 3 2>&1 | grep "READER_RATE"
