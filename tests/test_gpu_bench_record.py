"""`python bench.py` (N = 1) end to end on the GPU at a small size: ONE JSON line with the contract's keys, `roofline` priced on
the persistent kernel's own bytes, and -- round 5 -- the counter passes and rocprofv3's kernel time taken by the run itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_with_its_own_counter_passes():
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    n, m = 4_000_003, 6
    env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCPROF", "ROCP_")) and k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, "bench.py", "--dim", str(n), "--hist", str(m), "--steps", "20", "--repeats", "2", "--no-cpu-baseline",
           "--no-vector-free"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in j, key
    r = j["roofline"]
    assert j["n_gpus"] == 1 and j["dtype"] == "f64" and j["value"] > 0 and r["bound"] == "hbm" and r["peak"] == 8000.0
    assert r["kernel"].startswith("two_loop_resident_kernel") and r["resident_elements"] == n
    algo = 8 * (4 * m + 1) * n
    assert r["bytes_per_launch"] == algo and r["frac"] == pytest.approx(algo / (r["avg_ms"] * 1e-3) / 1e9 / 8000.0)
    # the run's own passes: either they were taken -- then the counters must say what the kernel is designed to move, and
    # rocprofv3's duration must agree with the HIP events -- or the record says why not and carries no invented figure
    if r.get("traffic_live") is not None:
        assert r["traffic"] == r["traffic_live"] and r["traffic_is_current"] is True and "this run" in r["traffic_source"]
        assert 0.93 * algo <= r["traffic"] * 1e9 <= 1.08 * algo, (r["traffic"], algo / 1e9)
        assert r["traffic_live_kernel"].startswith("two_loop_resident_kernel<")
    else:
        assert r.get("traffic_live_error"), r
        assert r["traffic"] is None          # (no committed passes exist at this size)
    if r.get("rocprofv3_avg_ms") is not None:
        assert 0.85 <= r["rocprofv3_over_hip_events"] <= 1.05, r["rocprofv3_over_hip_events"]
        assert r["rocprofv3_launches"] >= 20
    else:
        assert r.get("rocprofv3_error"), r
    print(f"bench n={n}: {j['value']:.0f} it/s, kernel {r['avg_ms'] * 1e3:.1f} us by HIP events, "
          f"{(r.get('rocprofv3_avg_ms') or 0) * 1e3:.1f} us by rocprofv3; traffic {r.get('traffic')} GB of {algo / 1e9:.4f} algorithmic"
          f"{'' if r.get('traffic_live') is not None else ' (' + str(r.get('traffic_live_error')) + ')'}")
