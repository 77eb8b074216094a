#!/bin/bash
# One runner for the GPU sessions of a round (replaces the per-call scripts of rounds 2 and 3).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_session.sh <tag> <stage> [<stage> ...]'
# Every stage writes under gpurun_out/<tag>/ and prints a short tail; what is judged is copied into profiles/ by hand
# (tools/collect_profiles.py for the rocprof summaries).  Stages:
#   tests            the GPU suite, driver style (pytest -x -q -m gpu) + smoke()
#   tests_new        only the files touched this round (MDC_TESTS="tests/a.py tests/b.py")
#   shapes           tools/sweep.py over tile shapes / frames per workgroup on the headline batch (SHAPES=..., FPB=..., FRAMES=...)
#   bracket          tools/mall_bracket.py: the same launch with its reads served by HBM / Infinity Cache / L2
#   ea               rocprofv3 --pmc passes (fabric read requests: total, DRAM-bound, 32 B / 64 B / 128 B, L2 hit / miss)
#                    over tools/mall_bracket.py (one launch per variant)
#   pmc_sq           SQ / TCC / TCP counter passes over tools/sweep.py (PMC_ARGS="--shapes 128x16 --fpb 64")
#   launch_size      bench.py --frames 4096 / 8192 / 16384 (headline only)
#   bench            plain `python bench.py` (the driver's command) -> bench.json
#   pyramid_alone    bench.py --workload pyramid in its own process (against the secondary entry of `bench`)
#   bench2           `python bench.py --gpus 2` without torchrun (gloo, shared GPU)
#   profile          tools/profile_round.py: rocprofv3 --kernel-trace --stats + FETCH_SIZE / WRITE_SIZE passes of bench.py's workloads, cut to the timed region
#   pipe_trace       per-chunk time stamps of the JPEG-stream pipeline (upload / Huffman done / decoded / output) on a 256-frame getImages
#   soak             reader soak (one lane, two lanes), tiled-kernel soak, 8-thread soak
#   reader2          the reader's rates with two lanes on the one GPU (MDC_DEVICES=0,0)
#   reader / dso / huffman / vcal / distort   the secondary rate tools
#   huffman_ab       the one-component Huffman decoder per library build (LIBS="product <variant> ...", mono_dataset_code_amd/variants/)
#   huffman_trace    rocprofv3 --kernel-trace --stats of tools/huffman_rate.py (COUNTS=64: streams per call)
#   reader_jpeg_trace  the same of the reader's getImages on a zip of N JPEGs (oracle/_ref/reader_rate_fast ... batch)
set -u
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
for stage in "$@"; do
  echo "=== stage $stage ($(date +%T))"
  case $stage in
    tests)
      timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/tests.txt" 2>&1; echo "rc=$?" >> "$OUT/tests.txt"
      grep -aE "passed|failed|error|rc=" "$OUT/tests.txt" | tail -5
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; grep -a "smoke" "$OUT/smoke.txt" | tail -2 ;;
    tests_new)
      timeout 1200 python -m pytest ${MDC_TESTS:-tests/test_gpu_parity.py} -x -q -m gpu > "$OUT/tests_new.txt" 2>&1; echo "rc=$?" >> "$OUT/tests_new.txt"
      grep -aE "passed|failed|error|rc=|assert" "$OUT/tests_new.txt" | tail -12 ;;
    shapes)
      timeout 600 python tools/sweep.py --frames ${FRAMES:-4096} --rounds ${ROUNDS:-4} --iters 4 --shapes ${SHAPES:-128x16,320x16,640x8,128x32} \
        --fpb ${FPB:-32,64,96} > "$OUT/shapes.txt" 2>&1; cat "$OUT/shapes.txt" | grep -av amdgpu.ids ;;
    bracket)
      timeout 600 python tools/mall_bracket.py --shapes ${SHAPES:-128x16,320x16} --rounds ${ROUNDS:-4} > "$OUT/bracket.txt" 2>&1; grep -av amdgpu.ids "$OUT/bracket.txt" ;;
    ea)
      ( cd /tmp && export TMPDIR=/tmp; i=0
        for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum" \
                 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_READ_sum TCC_REQ_sum"; do
          i=$((i+1))
          timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/ea_pass$i" -- python "$GRAFT_REPO_ROOT/tools/mall_bracket.py" \
            --shapes ${SHAPES:-128x16,320x16} --rounds 1 --iters 1 > "$OUT/ea_pass$i.log" 2>&1
        done )
      python3 tools/pmc_table.py "$OUT" ea_pass > "$OUT/ea_summary.txt" 2>&1; cat "$OUT/ea_summary.txt" ;;
    pmc_sq)
      # SQ / TCC / TCP counter passes of one kernel configuration (each its own rocprofv3 run, --kernel-trace only): PMC_ARGS = sweep.py arguments
      ( cd /tmp && export TMPDIR=/tmp; i=0
        for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
                 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU" \
                 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT" \
                 "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
          i=$((i+1))
          timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/sq_pass$i" -- python "$GRAFT_REPO_ROOT/tools/sweep.py" --rounds 1 --iters 2 ${PMC_ARGS:-} > "$OUT/sq_pass$i.log" 2>&1
        done )
      python3 tools/pmc_table.py "$OUT" sq_pass > "$OUT/sq_summary.txt" 2>&1; tail -8 "$OUT/sq_summary.txt" ;;
    launch_size)
      for n in 4096 8192 16384; do
        timeout 300 python bench.py --frames $n --steps 60 --warmup 10 --no-cpu-baseline > "$OUT/bench_frames_$n.json" 2> "$OUT/bench_frames_$n.err"
        python3 -c "import json,sys; d=json.loads([l for l in open('$OUT/bench_frames_$n.json') if l.startswith('{')][-1]); r=d['roofline']; print($n, 'frac', r['frac'], 'kernel_ms', r['kernel_ms'], 'of_ceiling', r.get('frac_of_same_box_mix_ceiling'), r['kernel'], d['config']['plan'])"
      done ;;
    ab)      # same-box A/B of library builds / knobs on the headline batch: tools/sweep.py with AB_ARGS (interleaved rounds, outputs compared bit for bit)
      timeout 900 python tools/sweep.py --frames ${FRAMES:-4096} --rounds ${ROUNDS:-5} --iters 4 ${AB_ARGS:-} > "$OUT/ab_${AB_NAME:-sweep}.txt" 2>&1; grep -av amdgpu.ids "$OUT/ab_${AB_NAME:-sweep}.txt" | tail -40 ;;
    pmc_rd)  # fabric read requests and L2 hits per variant of the same sweep (one launch each): separate --pmc passes, --kernel-trace only
      ( cd /tmp && export TMPDIR=/tmp; i=0
        for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_READ_sum"; do
          i=$((i+1))
          timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/rd_pass$i" -- python "$GRAFT_REPO_ROOT/tools/sweep.py" --frames ${FRAMES:-4096} --rounds 1 --iters 1 ${AB_ARGS:-} > "$OUT/rd_pass$i.log" 2>&1
        done )
      python3 tools/pmc_table.py "$OUT" rd_pass > "$OUT/rd_summary.txt" 2>&1; tail -${RD_TAIL:-30} "$OUT/rd_summary.txt" | cut -c1-260 ;;
    placement)  # the headline launch on buffers at different places: fresh pairs, offsets inside one arena, a sequence-sized pair
      timeout 600 python tools/placement_probe.py ${ROUNDS:-4} > "$OUT/placement_probe.txt" 2>&1; grep -av amdgpu.ids "$OUT/placement_probe.txt" | tail -40
      # the same with PyTorch's allocator on expandable segments (its own use of hipMemCreate / hipMemMap)
      PYTORCH_HIP_ALLOC_CONF=expandable_segments:True PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True timeout 300 python tools/alloc_probe.py 2 5 > "$OUT/alloc_probe_expandable.txt" 2>&1
      grep -a "torch.empty\|Error\|error" "$OUT/alloc_probe_expandable.txt" | tail -5 ;;
    alloc)   # linear write / read rate per allocation (does the rate depend on where a buffer lies?)
      timeout 600 python tools/alloc_probe.py ${ALLOC_ARGS:-32 5} > "$OUT/alloc_probe.txt" 2>&1; grep -av amdgpu.ids "$OUT/alloc_probe.txt" | tail -70 ;;
    classmap)  # which pieces of the device's memory are slow with which (tools/class_map.py)
      timeout 600 python tools/class_map.py ${CLASS_ARGS:-200 1} > "$OUT/class_map.txt" 2>&1; grep -av amdgpu.ids "$OUT/class_map.txt" | tail -60 ;;
    probe)   # why the same launch ran 8 % apart within one process: launch time against clocks / idle gaps / placement
      timeout 300 python tools/clock_probe.py ${PROBE_ARGS:-12 4 3} > "$OUT/clock_probe.txt" 2>&1; grep -a "===\|^A \|^B \|plan:\|idle snapshot" -A0 "$OUT/clock_probe.txt" | tail -8 ;;
    bench_each)  # every workload in its own process (which one faults?)
      for wl in fused unmap pyramid dso seq50k; do
        timeout 300 python3 bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_each_$wl.json" 2> "$OUT/bench_each_$wl.err"; echo "$wl rc=$?"
        tail -c 300 "$OUT/bench_each_$wl.err" | grep -a "fault\|Error\|error" ; python3 -c "
import json,sys
try:
    d=json.loads([l for l in open('$OUT/bench_each_$wl.json') if l.startswith('{')][-1]); print('  frac', d['roofline']['frac'], d['config']['placement'], d['parity']['mismatching_pixels'])
except Exception as e: print('  no line', e)"
      done ;;
    bench_driver)  # exactly what the driver runs at round end
      ( time timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 ${BENCH_ARGS:-} > "$OUT/bench_driver${BENCH_TAG:-}.json" 2> "$OUT/bench_driver.err" ) 2>&1 | grep real; tail -c 400 "$OUT/bench_driver.err"
      python3 - "$OUT/bench_driver${BENCH_TAG:-}.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("headline", d["value"], "Mpix/s  frac", r["frac"], "again", r.get("frac_again"), "kernel_ms", r["kernel_ms"], "of ceiling", r.get("frac_of_same_box_mix_ceiling"), r["kernel"], "parity", d["parity"])
print("preroll", d["config"]["preroll"], "again", d.get("again"))
print("placement", d["config"].get("placement"), r.get("placement"))
print("clocks idle", d.get("clocks_idle_at_start"), "after", r.get("clocks_in_timed_region"), "traffic", r["traffic"], r["traffic_source"])
for k, v in (d.get("secondary") or {}).items():
    print("secondary", k, "frac", v["frac"], "kernel_ms", v["kernel_ms"], "of ceiling", v["frac_of_same_box_mix_ceiling"], v["kernel"], "preroll s", v["preroll"]["seconds"], "parity", v["parity"] if isinstance(v["parity"], str) else v["parity"]["mismatching_pixels"])
PY
      ;;
    bench)
      ( time timeout 1200 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" ) 2>&1 | grep real; tail -c 600 "$OUT/bench.err"
      python3 - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("headline", d["value"], "Mpix/s  frac", r["frac"], "kernel_ms", r["kernel_ms"], "of ceiling", r.get("frac_of_same_box_mix_ceiling"), r["kernel"], "parity", d["parity"])
for k, v in (d.get("secondary") or {}).items():
    print("secondary", k, "frac", v["frac"], "kernel_ms", v["kernel_ms"], "of ceiling", v["frac_of_same_box_mix_ceiling"], v["kernel"], "parity", v["parity"])
print("cpu_baseline", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores", "kind")}, "build_flags", repr(d.get("build_flags")), "ranks", d.get("ranks"))
PY
      ;;
    pyramid_alone)
      for i in 1 2; do timeout 300 python bench.py --workload pyramid --no-cpu-baseline > "$OUT/bench_pyramid_alone_$i.json" 2> /dev/null
        python3 -c "import json; d=json.loads([l for l in open('$OUT/bench_pyramid_alone_$i.json') if l.startswith('{')][-1]); r=d['roofline']; print('pyramid alone: frac', r['frac'], 'kernel_ms', r['kernel_ms'], 'of ceiling', r.get('frac_of_same_box_mix_ceiling'), r.get('launches_per_step'))"; done ;;
    bench2)
      MDC_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 > "$OUT/bench_gpus2_gloo.json" 2> "$OUT/bench_gpus2_gloo.err"; echo "rc=$?"
      tail -c 1500 "$OUT/bench_gpus2_gloo.json"; echo
      timeout 120 python bench.py --gpus 8 > "$OUT/bench_gpus8_refused.txt" 2>&1; echo "--gpus 8 on this box: rc=$?"; tail -2 "$OUT/bench_gpus8_refused.txt" ;;
    profile)  # the round's rocprofv3 evidence, one session of one build: tools/profile_round.py (PROFILE_WLS="fused unmap ...")
      timeout ${PROFILE_TIMEOUT:-2400} python3 tools/profile_round.py ${PROFILE_TAG:-$TAG} ${PROFILE_WLS:-} > "$OUT/profile_round.txt" 2>&1
      grep -av amdgpu.ids "$OUT/profile_round.txt" | tail -12 | cut -c1-400 ;;
    pipe_trace)
      MDC_TRACE_ENV=MDC_PIPE_TRACE=1 timeout 600 python tools/reader_trace.py 256 3 batch > "$OUT/pipe_trace.txt" 2>&1
      grep -a "chunks, ms since" "$OUT/pipe_trace.txt" | tail -3; grep -a "READER_RATE /\|READER_RATE r" "$OUT/pipe_trace.txt" | tail -2 ;;
    soak)
      timeout 200 python tools/reader_soak.py ${SOAK_S:-40} 1 > "$OUT/reader_soak.txt" 2>&1; grep -av amdgpu.ids "$OUT/reader_soak.txt" | tail -3
      MDC_DEVICES=0,0 timeout 200 python tools/reader_soak.py ${SOAK_S:-40} 2 > "$OUT/reader_soak_two_lanes.txt" 2>&1; grep -av amdgpu.ids "$OUT/reader_soak_two_lanes.txt" | tail -3
      timeout 300 python tools/soak.py 60 > "$OUT/soak.txt" 2>&1; tail -2 "$OUT/soak.txt"
      timeout 400 bash tools/soak_threads.sh 800 > "$OUT/thread_soak.txt" 2>&1; tail -4 "$OUT/thread_soak.txt" ;;
    soak_device)  # the reader soak with getImagesDevice as most of the mix, seeds 1..3, default lanes and MDC_DEVICES=0,0
      for seed in 1 2 3; do
        SOAK_DEVICE_SHARE=0.8 timeout 200 python tools/reader_soak.py ${SOAK_S:-30} $seed > "$OUT/reader_soak_device_$seed.txt" 2>&1; grep -a "MISMATCH\|position\|READER_SOAK\|Error" "$OUT/reader_soak_device_$seed.txt" | cut -c1-400
      done
      MDC_DEVICES=0,0 SOAK_DEVICE_SHARE=0.8 timeout 200 python tools/reader_soak.py ${SOAK_S:-30} 4 > "$OUT/reader_soak_device_two_lanes.txt" 2>&1; grep -a "MISMATCH\|position\|READER_SOAK\|Error" "$OUT/reader_soak_device_two_lanes.txt" | cut -c1-400 ;;
    reader_device)  # getImagesDevice on a zipped JPEG sequence: lanes per device / frames per pipeline chunk (RATE_ENVS="A=1 B=2;A=3")
      MDC_RATE_ONLY=device MDC_RATE_KINDS=${KINDS:-zip_jpg} MDC_RATE_ENVS="${RATE_ENVS:-MDC_DEVICES=0,0}" timeout 900 python tools/reader_rate.py ${N:-1024} > "$OUT/reader_device_rates.txt" 2>&1
      grep -a "^==\|^--\|READER_RATE reader\|READER_RATE device [0-9]" "$OUT/reader_device_rates.txt" | cut -c1-200 ;;
    reader2) MDC_DEVICES=0,0 MDC_RATE_KINDS=zip_jpg timeout 900 python tools/reader_rate.py ${N:-512} > "$OUT/reader_rates_two_lanes.txt" 2>&1; grep -av amdgpu.ids "$OUT/reader_rates_two_lanes.txt" | tail -30 ;;
    reader)  timeout 900 python tools/reader_rate.py ${N:-512} > "$OUT/reader_rates.txt" 2>&1; grep -av amdgpu.ids "$OUT/reader_rates.txt" | tail -30 ;;
    dso)     timeout 600 python tools/dso_rate.py > "$OUT/dso_rate.txt" 2>&1; grep -av amdgpu.ids "$OUT/dso_rate.txt" | tail -20 ;;
    dso_ab)  # the DSO hand-off per library build (LIBS="product <variant> ...", chunk sizes swept inside tools/dso_rate.py), two rounds each
      for r in 1 2; do for l in ${LIBS:-product}; do
        lib=""; [ $l != product ] && lib="$GRAFT_REPO_ROOT/mono_dataset_code_amd/variants/libmdc_hip_$l.so"
        echo "--- $l (round $r)" >> "$OUT/dso_ab.txt"
        MDC_LIB_HIP=$lib timeout 300 python tools/dso_rate.py 2>&1 | grep -av amdgpu.ids | grep -a "launches\|chunks\|bit for bit" >> "$OUT/dso_ab.txt"
      done; done; cat "$OUT/dso_ab.txt" ;;
    huffman) timeout 600 python tools/huffman_rate.py > "$OUT/huffman_rate.txt" 2>&1; grep -av amdgpu.ids "$OUT/huffman_rate.txt" | tail -20 ;;
    huffman_ab)  # the one-component decoder per library build (LIBS="product huffseg4 ...": mono_dataset_code_amd/variants/libmdc_hip_<name>.so), two rounds
      for r in 1 2; do for l in ${LIBS:-product}; do
        lib=""; [ $l != product ] && lib="$GRAFT_REPO_ROOT/mono_dataset_code_amd/variants/libmdc_hip_$l.so"
        echo "--- $l (round $r)" >> "$OUT/huffman_ab.txt"
        MDC_LIB_HIP=$lib HUFF_KINDS=gray REPS=20 timeout 300 python tools/huffman_rate.py 2>&1 | grep -a "^n " >> "$OUT/huffman_ab.txt"
      done; done; cat "$OUT/huffman_ab.txt" ;;
    huffman_trace)  # per-kernel times of the one-component decoder at COUNTS streams per call
      ( cd /tmp && export TMPDIR=/tmp
        HUFF_KINDS=gray REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/huff_trace" -- python "$GRAFT_REPO_ROOT/tools/huffman_rate.py" > "$OUT/huffman_trace.log" 2>&1 )
      f=$(find "$OUT/huff_trace" -name "*kernel_stats.csv" | head -1); cut -d, -f1-7 "$f" | grep -a "jpeg\|Name" | cut -c1-150 ;;
    reader_jpeg_trace)  # rocprofv3 --kernel-trace --stats of the reader's JPEG path: getImages on a zip of N JPEGs, 6 passes
      python3 - "$OUT" ${N:-1024} <<'PY' > "$OUT/reader_jpeg_make.txt" 2>&1
import os, sys
sys.argv = ["reader_rate.py", sys.argv[2]]
os.environ["MDC_RATE_KINDS"] = ""
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
src = open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools", "reader_rate.py")).read().split("\nfor kind in os.environ.get")[0]
exec(compile(src, "reader_rate_head", "exec"))
d, avg = make("zip_jpg")
print("DATASET", d)
PY
      D=$(grep -a "^DATASET" "$OUT/reader_jpeg_make.txt" | cut -d" " -f2)
      ( cd /tmp && export TMPDIR=/tmp
        timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/reader_jpeg_trace" -- "$GRAFT_REPO_ROOT/oracle/_ref/reader_rate_fast" "$D" 1111 6 batch > "$OUT/reader_jpeg_trace.log" 2>&1 )
      grep -a "READER_RATE reader" "$OUT/reader_jpeg_trace.log" | tail -1
      f=$(find "$OUT/reader_jpeg_trace" -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Name"])
    print("%-62s calls %5s  avg %8.1f us  total %8.2f ms  %5.1f %%" % (m.group(0) if m else r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
      ;;
    placed)  # the headline launch on buffers from the product's allocator, one strategy / setting per process (PLACED_CFGS="strategy:piece MiB:compose:rounds[:frames] ...")
      for cfg in ${PLACED_CFGS:-first:0:0:1 malloc:0:0:1 vmm:1024:0:1 vmm:1024:1:1 vmm:1024:2:1 vmm:256:0:1 vmm:256:2:1 vmm:64:2:1}; do
        IFS=: read -r strat piece comp rounds frames <<< "$cfg"
        MDC_PLACE_PIECE_MIB=${piece:-1024} MDC_PLACE_COMPOSE=${comp:-0} timeout ${PLACED_TIMEOUT:-400} python tools/placed_probe.py $strat ${rounds:-1} ${frames:-4096} >> "$OUT/placed_probe.txt" 2>&1
        echo "rc=$? ($cfg)" >> "$OUT/placed_probe.txt"
      done
      grep -a "PLACED\|rc=\|fault\|Error\|^   " "$OUT/placed_probe.txt" | cut -c1-420 ;;
    power)   # power / clocks / launch time per library build (tools/power_ab.py; LIBS="lutrep16 fake1 ...")
      libs=default; for l in ${LIBS:-lutrep16 lutrep8 fake1 fake2 skipstore skipload padvalu64}; do libs="$libs,mono_dataset_code_amd/variants/libmdc_hip_$l.so"; done
      timeout 600 python tools/power_ab.py --libs "$libs" --rounds ${ROUNDS:-3} ${POWER_ARGS:-} > "$OUT/power_ab.txt" 2>&1; grep -av amdgpu.ids "$OUT/power_ab.txt" | tail -14 | cut -c1-330 ;;
    pyramid_sweep)  # config 5: frames per prefetched chunk x streams (PYR_CHUNKS, PYR_STREAMS, PYR_PLACED=1: buffers from the product's allocator)
      timeout 900 python tools/pyramid_sweep.py ${PYR_N:-1024} ${ROUNDS:-3} > "$OUT/pyramid_sweep${PYR_TAG:-}.txt" 2>&1; grep -av amdgpu.ids "$OUT/pyramid_sweep${PYR_TAG:-}.txt" | tail -24 ;;
    vmm_check)  # hipMemMap with an offset; address ranges given back and re-used / kept / re-used after an ordinary hipMalloc + hipFree (tools/vmm_offset_check.hip)
      [ -x tools/bin/vmm_offset_check ] || { mkdir -p tools/bin; hipcc --offload-arch=gfx950 -O2 -o tools/bin/vmm_offset_check tools/vmm_offset_check.hip; }
      ( echo "--- stripes of 64 MiB (an offset into the handle)"; tools/bin/vmm_offset_check 3 512 64 1
        echo "--- whole pieces, address ranges given back and re-used"; tools/bin/vmm_offset_check 3 512 512 4
        echo "--- whole pieces, address ranges kept (VMM_KEEP_VA=1)"; VMM_KEEP_VA=1 tools/bin/vmm_offset_check 3 512 512 5
        echo "--- address ranges re-used, a hipMalloc + hipFree of 1 GiB between the rounds (VMM_CHURN=1)"; VMM_CHURN=1 tools/bin/vmm_offset_check 3 512 512 4
        echo "--- 12 pieces, address ranges kept"; VMM_KEEP_VA=1 tools/bin/vmm_offset_check 12 512 512 4
        echo "--- the allocator: ranges of the process after earlier ones were freed (tools/striped_alloc_check.py)"; python tools/striped_alloc_check.py 2>&1 | grep -a "frame\|^set\|ptr" | cut -c1-200 ) > "$OUT/vmm_offset_check.txt" 2>&1
      grep -a "^---\|^round\|MISMATCH\|all pages\|refused\|BAD\|False" "$OUT/vmm_offset_check.txt" | cut -c1-200 | tail -30 ;;
    distort) timeout 300 python tools/distort_rate.py > "$OUT/distort_rate.txt" 2>&1; grep -av amdgpu.ids "$OUT/distort_rate.txt" | tail -5 ;;
    vcal)    timeout 600 python tools/vcal_rate.py > "$OUT/vcal_rate.txt" 2>&1; grep -av amdgpu.ids "$OUT/vcal_rate.txt" | tail -20 ;;
    *) echo "unknown stage $stage" ;;
  esac
done
echo "=== done ($(date +%T))"
