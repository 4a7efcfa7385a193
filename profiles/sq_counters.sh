#!/bin/bash
# Wavefront-level counters of the row kernel (one rocprofv3 pass, SQ block only): is a step of several chunks short of workgroup slots
# (wavefronts parked at s_waitcnt: SQ_WAIT_ANY) or short of issue cycles (SQ_ACTIVE_INST_*)?
# usage (from the repo root, through gpurun): profiles/sq_counters.sh <tag> [bench.py flags]; output gpurun_out/sq_<tag>.json
set -e
TAG=$1; shift
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --output-format csv -d $REPO/gpurun_out/prof_sq_${TAG} -o sq -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu "$@" > /dev/null || echo "counter pass ended with an error"
cd $REPO && python3 profiles/sq_summary.py gpurun_out/prof_sq_${TAG} gpurun_out/sq_${TAG}.json "$*"
