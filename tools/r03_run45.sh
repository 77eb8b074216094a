#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
sed -i 's/MDC_READER_TRACE="1")/MDC_READER_TRACE="1", MDC_PIPE_TRACE="1")/' tools/r03_run39.sh
sed -i 's/if "getImages" in l or "READER_RATE" in l/if "getImages" in l or "READER_RATE" in l or "streams drained" in l/' tools/r03_run39.sh
bash tools/r03_run39.sh 2>&1 | grep -A12 "stage 2" | tail -8
