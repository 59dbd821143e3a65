REPO=$(pwd)
timeout -k 10 300 python -m pytest tests/test_gpu_setup.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sgprof
rocprofv3 --kernel-trace --stats -d /tmp/sgprof -o sg -- python3 -u $REPO/tools/spgemm_lab.py 112 6 1 1 > $REPO/gpurun_out/sgprof.log 2>&1
db=$(find /tmp/sgprof -name "*.db" | head -1)
python3 - <<PY
import sqlite3, re
db = sqlite3.connect("$db")
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = list(db.execute("select * from kernels"))
ni = cols.index("name"); si = cols.index("start"); ei = cols.index("end")
rows.sort(key=lambda r: r[si])
for r in rows[-40:]:
    m = re.search(r"(k_[a-z0-9_]+(<\d+>)?)", r[ni])
    if m and "spgemm" in m.group(1): print(m.group(1), (r[ei]-r[si])/1e3, "us", r[cols.index("grid_x")])
PY
grep "^\[lab\]\|spgemm\]" $REPO/gpurun_out/sgprof.log | tail -4 | cut -c1-250
