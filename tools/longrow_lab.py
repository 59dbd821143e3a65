#!/usr/bin/env python3
"""Row-length lab (GPU box): achieved GB/s of airhip_mat_mult on banded operators of 3.2e8 non-zeros whose rows all have
the same length L - below, at and above the 512 entries of a row block (a longer row is one block whose tail is read in a
loop).  Shows what a row longer than a block costs, apart from the gather locality of the real coarse operator.
usage: python tools/longrow_lab.py [nnz_millions] [spread]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmv_lab import csr_banded, run

if __name__ == "__main__":
    nnz_m = float(sys.argv[1]) if len(sys.argv) > 1 else 320.0
    spread = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    for L in (128, 256, 400, 512, 513, 576, 640, 768, 1024, 1536, 2048, 4096):
        m = int(nnz_m * 1e6 / L)
        run(f"banded L={L} spread={spread}", csr_banded(m + 2 * spread * (L + 2), L, spread=spread))
