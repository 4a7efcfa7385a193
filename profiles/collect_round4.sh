#!/bin/bash
# Collects the rocprofv3 evidence of round 4 on the GPU box (run through gpurun from the repo root):
#   1. kernel-trace statistics of a bench workload shape on a 10 Mb prefix (per-kernel average durations)
#   2. HBM traffic counters, one pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), on a 3 Mb prefix
#      (rocprofv3 7.2 crashed in the FETCH_SIZE pass of the 18 370-launch 10 Mb run of round 3)
# usage: profiles/collect_round4.sh <tag> '<shape json>' [extra bench.py flags]; outputs under gpurun_out/r4_<tag>_*
set -e
TAG=${1:-C3}; SHAPE=${2:-'{"nsam": 4, "np": 10000, "epochs": 32, "pops": 1}'}; shift; shift || true
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --length 1e7 --steps 1 --warmup 0 --no-cpu $*"
PLEN=${PMC_LENGTH:-3e6}
PARGS="$REPO/bench.py --length $PLEN --steps 1 --warmup 0 --no-cpu $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG}_stats -o stats -- python3 $ARGS > $REPO/gpurun_out/r4_${TAG}_bench_under_rocprof.json
# (a counter pass that crashes in the profiler's finalisation has usually written its csv already: go on and summarise what is there)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof_${TAG}_fetch -o fetch -- python3 $PARGS > /dev/null || echo "FETCH_SIZE pass ended with an error"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof_${TAG}_write -o write -- python3 $PARGS > /dev/null || echo "WRITE_SIZE pass ended with an error"
cd $REPO && python3 profiles/summarize.py gpurun_out/prof_${TAG} gpurun_out/r4_${TAG} "$SHAPE" "counter passes on a prefix of $PLEN bp (--length $PLEN), kernel-trace statistics on a 10 Mb prefix (--length 1e7)"
