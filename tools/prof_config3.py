import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import rust_lbfgs_amd as R
from rust_lbfgs_amd import objectives
n = 10_000_000
ctx = R.Context(n)
st = R.lbfgs().with_orthantwise(0.5, 0, None).with_epsilon(0.0).build(np.zeros(n), objectives.Logistic(), ctx=ctx)
for _ in range(22):
    if st.is_converged(): break
    st.propagate()
st.close(); ctx.close()
