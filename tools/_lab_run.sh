REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
pass() {
  tag=$1; shift
  rm -rf /tmp/sgpmc_$tag
  rocprofv3 --pmc $@ -d /tmp/sgpmc_$tag -o sg -- python3 -u $REPO/tools/spgemm_lab.py 112 6 1 1 > $REPO/gpurun_out/sgpmc_$tag.log 2>&1
  db=$(find /tmp/sgpmc_$tag -name "*.db" | head -1)
  python3 - <<PY
import sqlite3, re
db = sqlite3.connect("$db")
rows = list(db.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection where kernel_name like '%k_spgemm_lds%'"))
last = {}
for d, k, c, v in rows:
    m = re.search(r"(k_[a-z0-9_]+(<\d+>)?)", k); k = m.group(1)
    last[k] = max(last.get(k, 0), d)
for d, k, c, v in sorted(rows):
    m = re.search(r"(k_[a-z0-9_]+(<\d+>)?)", k); k = m.group(1)
    if last[k] == d and "4096" in k: print(f"{k:22s} {c:28s} {v:.6g}")
PY
}
pass a SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
