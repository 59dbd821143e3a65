#!/bin/bash
# The round's measurement deliverables on the current build (GPU box, repo root):
#   tools/final_measure.sh <tag>      e.g. r03_final
# 1. default bench line  2. kernel trace + stats  3. FETCH_SIZE / WRITE_SIZE passes (k_rb_spmv and k_rb_fc_first)
set -e
tag=$1
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench464_default.json 2> gpurun_out/${tag}_bench464_default.err
tail -c 600 gpurun_out/${tag}_bench464_default.json; echo
tools/trace_run.sh ${tag}_rocprof_bench464 --cpu-iters 0 --verbose
tools/pmc_run.sh ${tag}_pmc464_fetch "FETCH_SIZE" --cpu-iters 0 --steps 10 --warmup 2
tools/pmc_run.sh ${tag}_pmc464_write "WRITE_SIZE" --cpu-iters 0 --steps 10 --warmup 2
