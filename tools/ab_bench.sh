#!/bin/bash
# Same-box A/B of library builds: tools/ab_bench.sh <tag> <size> <lib1> <lib2> ... (libs relative to pflare_amd/)
# each library runs bench.py twice, alternating, logs under gpurun_out/<tag>_<lib>_<rep>.log
tag=$1; size=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$REPO/gpurun_out"
for rep in 1 2; do
  for lib in "$@"; do
    name=$(basename "$lib" .so)
    AIRHIP_LIB="$REPO/pflare_amd/$lib" python "$REPO/bench.py" --size "$size" --steps 10 --warmup 3 --cpu-iters 0 --verbose > "$REPO/gpurun_out/${tag}_${name}_${rep}.log" 2>&1
    python - "$REPO/gpurun_out/${tag}_${name}_${rep}.log" "$name" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); r=d["roofline"]
        print(f"{sys.argv[2]:24s} {d['value']:8.2f} it/s  k_rb_spmv {r['achieved']:.0f} GB/s frac {r['frac']:.4f}  avg launch {r['avg_launch_ms']:.4f} ms", flush=True)
PY
  done
done
