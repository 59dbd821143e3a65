#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command, summarised per kernel into gpurun_out/<tag>_kernel_stats.csv
# usage (GPU box, repo root): tools/trace_run.sh <tag> <bench.py args ...>
set -e
tag=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/prof_$tag"
rocprofv3 --kernel-trace --stats -d "/tmp/prof_$tag" -o "$tag" -- python3 -u "$REPO/bench.py" "$@" > "$REPO/gpurun_out/${tag}_run.log" 2>&1
db=$(find "/tmp/prof_$tag" -name "*.db" | head -1)
python3 "$REPO/tools/kernel_stats.py" "$db" > "$REPO/gpurun_out/${tag}_kernel_stats.csv"
echo "trace $tag done: $(wc -l < "$REPO/gpurun_out/${tag}_kernel_stats.csv") kernels"
