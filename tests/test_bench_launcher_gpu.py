"""bench.py through the launcher branch the driver uses for N > 1 (`python bench.py --gpus 2` spawns
`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`): two ranks on this box's single GPU with gloo in
place of RCCL (SODT_BENCH_ONE_DEVICE / SODT_BENCH_BACKEND rehearsal hooks of bench.py), small shapes, as a CHILD process;
the one JSON line of rank 0 must carry the contract's keys with n_gpus == 2."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_launcher_child_process():
    env = dict(os.environ, SODT_BENCH_ONE_DEVICE="1", SODT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "128", "--batch", "2", "--steps", "2",
           "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 4
    assert out["scaling"] == "weak" and out["steps"] == 2 and out["value"] > 0 and out["unit"] == "images/sec"
    assert out["roofline"]["bound"] in ("mfma", "hbm") and out["roofline"]["frac"] > 0
    assert "cpu_baseline" not in out          # rank 0 at N = 1 only
