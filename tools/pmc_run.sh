#!/bin/bash
# One rocprofv3 --pmc pass of bench.py, summarised per kernel into gpurun_out/pmc_<tag>_summary.csv.
# usage (on the GPU box, from the repo root):  tools/pmc_run.sh <tag> "<counter> <counter> ..." <bench.py args ...>
#   PMC_SCRIPT=tools/pmc_calibrate.py selects another program of this repo instead of bench.py
# The program itself follows `--` (no env / bash -c hop: the profiler's preloaded library initialises the GPU).
set -e
tag=$1; counters=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/prof_$tag"
rocprofv3 --pmc $counters -d "/tmp/prof_$tag" -o "$tag" -- python3 -u "$REPO/${PMC_SCRIPT:-bench.py}" "$@" > "$REPO/gpurun_out/pmc_$tag.log" 2>&1
db=$(find "/tmp/prof_$tag" -name "*.db" | head -1)
python3 "$REPO/tools/pmc_summary.py" "$db" k_rb > "$REPO/gpurun_out/pmc_${tag}_summary.csv"
cp "$db" "/tmp/pmc_$tag.db"
echo "pmc pass $tag done: $(wc -l < "$REPO/gpurun_out/pmc_${tag}_summary.csv") rows"
