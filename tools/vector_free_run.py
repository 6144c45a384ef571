#!/usr/bin/env python3
"""A short vector-free (Gram) L-BFGS run at n = 1e8, m = 10, for profiling its kernels under rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/vector_free_run.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lbfgs_amd as R  # noqa: E402
from rust_lbfgs_amd import objectives  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
with R.Context(n) as ctx:
    st = R.lbfgs().with_m(10).with_epsilon(0.0).with_vector_free(True).build(np.zeros(n), objectives.Quadratic(), ctx=ctx)
    for _ in range(24):
        st.propagate()
    st.close()
print("done", flush=True)
