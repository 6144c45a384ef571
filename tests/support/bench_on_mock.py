"""TEST INFRASTRUCTURE: bench.py's rank process on the CPU test double of the C-ABI.

The `-m "not gpu"` suite uses this as LBFGS_BENCH_WORKER to exercise bench.py's N > 1 supervisor (both launch
forms, leg timeouts, the one-JSON-line contract) with two gloo ranks and no GPU.  Nothing here is reachable from
the product or from bench.py's default path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

torch.cuda.set_device = lambda *_a, **_k: None
torch.cuda.synchronize = lambda *_a, **_k: None

from tests.support import mock  # noqa: E402

mock.install()

import bench  # noqa: E402

sys.exit(bench.main())
