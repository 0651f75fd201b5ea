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
    # a lost overlap is loud: every rank's helper-stream placement is on the line (AND over the ranks), and so is what was
    # not placed.  (Two ranks SHARE the one GPU here, so the verdict itself is not asserted -- only that it is reported, and
    # that a lost overlap carries the warning.)
    assert isinstance(line["overlap_verified"], bool) and isinstance(line["helper_streams"]["unplaced"], list)
    assert line["overlap_verified"] == (not line["helper_streams"]["unplaced"])
    if not line["overlap_verified"]:
        assert "overlap_warning" in line


@pytest.mark.gpu
def test_one_rank_rccl_exchange_inside_the_step():
    """--force-exchange: RCCL itself (backend "nccl", a group of one rank) carries the packed all-gather at the end of every
    timed step -- the collective the N > 1 launch issues, on the library the driver's multi-GPU run will use."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--force-exchange",
           "--no-fp16x2-leg", "--no-cpu-baseline", "--no-training-leg"]
    done = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, done.stdout[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["exchange"]["ranks"] == 1 and "RCCL" in line["exchange"]["collective"]
    assert line["exchange"]["bytes_per_rank"] == 4 * 8 * (4096 + 1024 + 512)
    assert line["ms_per_step_no_exchange"] > 0 and line["ms_per_step"] > 0
    assert line["config"]["workload"].startswith("BASELINE configs[1]") and "ONE rank" in line["config"]["workload"]
    assert line["validated"]["last_step_bit_identical_to_sequential_pass"] is True
    # RCCL's own streams must not push the FPS producer and its consumers onto one hardware queue (the pass would take
    # FPS + everything else, 3.2 ms, instead of their maximum, 2.3 ms): the helper streams are probed at set-up
    assert line["helper_streams"]["probes"] > 0
    assert line["overlap_verified"] is True and line["helper_streams"]["unplaced"] == [] and "overlap_warning" not in line
    assert line["ms_per_step_no_exchange"] < line["roofline"]["launch_ms"] + 1.0, (line["ms_per_step_no_exchange"], line["roofline"])


@pytest.mark.gpu
def test_one_command_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's natural command, and the reference's one command per
    node: tools/scripts/dist_train.sh) starts two ranks itself and reports them -- not one GPU under the label it was asked for."""
    env = dict(os.environ, SPS_BENCH_REHEARSAL="one-gpu")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-fp16x2-leg"]
    done = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, done.stdout[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["exchange"]["ranks"] == 2 and line["config"]["global_batch"] == 16
    assert line["config"]["workload"].startswith("BASELINE configs[2]")
    assert line["validated"]["last_step_bit_identical_to_sequential_pass"] is True


def test_world_size_that_differs_from_gpus_is_an_error():
    """A launcher that set WORLD_SIZE=1 under `--gpus 2` (or the reverse) must not produce a line: bench.py exits non-zero before
    it touches a GPU (runs on CPU)."""
    for world, gpus in (("1", "2"), ("2", "1")):
        env = dict(os.environ, WORLD_SIZE=world, RANK="0", LOCAL_RANK="0")
        done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus, "--steps", "1", "--warmup", "0"],
                              cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert done.returncode != 0 and "WORLD_SIZE" in done.stderr, (done.returncode, done.stderr[-500:])
        assert not [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
