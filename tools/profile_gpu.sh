#!/bin/bash
# Collect the rocprofv3 evidence for one bench workload on the GPU box (run through gpurun):
#   tools/profile_gpu.sh <tag> <bench args...>
# Pass 1: kernel trace + stats.  Passes 2-4: PMC counters, each in its own run (TCC slots: FETCH_SIZE
# needs 3, WRITE_SIZE 2, so they cannot share a pass).  The program itself follows `--` (no wrappers).
set -e
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --no-cpu-baseline "$@" > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -40
