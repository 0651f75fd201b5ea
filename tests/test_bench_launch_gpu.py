"""The driver's N > 1 launch of bench.py (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) rehearsed with
two ranks on the one GPU a test box has: SPS_BENCH_REHEARSAL=one-gpu puts every rank on cuda:0 and swaps RCCL for gloo (RCCL
refuses two ranks on one device); the barriers, the max-over-ranks timing, the validation of the timed work on every rank and
the one packed all-gather of the sampled indices are the code the real launch runs."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_rank_launch_prints_one_valid_line():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SPS_BENCH_REHEARSAL="one-gpu")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    done = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, done.stdout[-3000:]                       # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 2 * 8
    assert line["validated"] == {"progress_wait_timeouts": False, "split_fp16_overflow": False,
                                 "last_step_bit_identical_to_sequential_pass": True}
    assert "REHEARSAL" in line["data"]
    # BASELINE configs[2]: the one packed all-gather of the sampled indices is INSIDE the timed step, and the same steps
    # were timed without it as well
    assert line["config"]["workload"].startswith("BASELINE configs[2]") and "all-gather" in line["config"]["parallelism"]
    assert line["exchange"]["per_step"] == 1 and line["exchange"]["bytes_per_rank"] == 4 * 8 * (4096 + 1024 + 512)
    assert line["ms_per_step_no_exchange"] > 0
    assert line["dtype"] == "f32" and "value_fp16x2" in line
    # whole-job aggregate: both ranks' points over the slowest rank's time
    assert abs(line["value"] - 2 * 8 * 16384 * 3 / (line["ms_per_step"] * 3e-3)) <= 1e-6 * line["value"]
    assert "cpu_baseline" not in line and "training_step" not in line    # N = 1 only
