#!/bin/bash
# Collects the rocprofv3 evidence behind DESIGN.md section 7 (round 3) on the GPU box (run through gpurun from the repo root):
#   1. kernel-trace statistics of the bench workload shape on a 10 Mb prefix (per-kernel average durations)
#   2. HBM traffic counters, one pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950)
# usage: profiles/collect_round3.sh <tag> [extra bench.py flags]; outputs under gpurun_out/prof_<tag>_*
set -e
TAG=${1:-c3}; shift || true
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --length 1e7 --steps 1 --warmup 0 --no-cpu $*"
# the counter passes on a shorter prefix when PMC_LENGTH is set (rocprofv3 7.2 crashed in the FETCH_SIZE pass of the 18 370-launch C3 run)
PARGS="$REPO/bench.py --length ${PMC_LENGTH:-1e7} --steps 1 --warmup 0 --no-cpu $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG}_stats -o stats -- python3 $ARGS > $REPO/gpurun_out/prof_${TAG}_stats.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof_${TAG}_fetch -o fetch -- python3 $PARGS > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof_${TAG}_write -o write -- python3 $PARGS > /dev/null
cd $REPO && python3 profiles/summarize.py gpurun_out/prof_${TAG} gpurun_out/sum_${TAG}
