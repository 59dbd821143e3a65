#!/usr/bin/env python3
"""Known-byte-count workload for calibrating FETCH_SIZE / WRITE_SIZE on the row-block SpMV's own access widths
(MI355X_MICROARCH.md, "HBM": widths other than 16 B per lane are uncalibrated).  A banded operator with rows of 64
consecutive columns, 3 GB of matrix stream (well past the 256-MiB Infinity Cache): every x line is used by neighbouring
blocks of the same XCD at the same time, so the HBM traffic of one product is the algorithmic traffic
S = 12 nnz + 4 (m+1) + 8 m + 8 n within a percent.  Prints S; run under `tools/pmc_run.sh` with PMC_SCRIPT set."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import pflare_amd as P
from spmv_lab import csr_banded

L = 64
m = int(256e6 / L)
A = csr_banded(m, L)
mm, n = A[3]
nnz = int(A[0][-1])
dm = P.DeviceMat(A)
x = torch.randn(n, dtype=torch.float64, device="cuda")
y = torch.empty(mm, dtype=torch.float64, device="cuda")
for _ in range(10):
    dm.mult(x, y)
torch.cuda.synchronize()
S = 12.0 * nnz + 4.0 * (mm + 1) + 8.0 * mm + 8.0 * n
print(f"calibration: banded L={L} m={mm} nnz={nnz}: algorithmic bytes per launch S = {S:.0f} (stream {12.0 * nnz:.0f}, "
      f"written {8.0 * mm:.0f}); 10 launches of k_rb_spmv")
