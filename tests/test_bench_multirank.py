"""The N > 1 path of bench.py under test: the driver runs `bench.py --gpus N` on an 8-GPU node at round end, and no 8-GPU
node has been available to it yet - so the path (self-launch before any GPU call, one rank per process, barriers, the per-48-step all-gather inside
the timed region, max-over-ranks timing, rank 0's single JSON line) is exercised here every round with two gloo ranks sharing the one GPU of the
test box.  A rehearsal, labelled so in the line it prints; the measured configuration is nccl (= RCCL), one GPU per rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_over_gloo():
    env = dict(os.environ, LM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):      # the child must self-launch its ranks
        env.pop(k, None)
    # a fresh child process (never an exec of this one, which may have touched the GPU); it starts torch.distributed.run before any HIP call
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "96", "--warmup", "4", "--no-cpu-baseline", "--timed-only"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                   # ONE JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 96 and r["warmup"] == 4 and r["scaling"] == "weak" and r["higher_is_better"] is True
    c = r["config"]
    assert c["global_envs"] == 8192 and c["envs_per_gpu"] == 4096
    assert c["all_gather_blocks_timed"] == 2 and c["all_gather_ms_per_block_rank0"] > 0          # steps 47 and 95 of the 100 close a 48-step rollout
    assert "REHEARSAL" in r["data"]                             # gloo on a shared GPU is never reported as a measurement
    assert r["value"] > 0 and abs(r["value"] - 8192 * 96 / (r["ms_per_step"] * 96e-3)) / r["value"] < 1e-6
    assert r["roofline"]["kernel_ms"] > 0 and r["roofline"]["step_period_ms"] >= r["roofline"]["kernel_ms"]
    assert "cpu_baseline" not in r                              # N > 1: rank 0 does not time the CPU


@pytest.mark.gpu
def test_rccl_collectives_run_on_this_box():
    """backend "nccl" IS RCCL on ROCm.  Two RCCL ranks cannot share the one GPU of the test box, so the library itself is exercised with a
    one-rank group in a fresh child process: process-group init on the device, the (2, 48, N) all-gather of bench.py / distributed.all_gather_rollout
    and the all-reduce of the PPO harness on device tensors.  Catches an image or environment in which RCCL cannot start (e.g. the IPC mode
    variable the pool needs) before the driver's 8-GPU run does."""
    code = (
        "import os, torch, torch.distributed as dist\n"
        "os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group(backend='nccl', rank=0, world_size=1)\n"
        "x = torch.arange(2 * 48 * 4096, device='cuda', dtype=torch.float32).reshape(2, 48, 4096); out = torch.empty(2, 48, 4096, device='cuda')\n"
        "dist.all_gather_into_tensor(out, x); g = torch.ones(58649, device='cuda'); dist.all_reduce(g); dist.barrier(device_ids=[0]); torch.cuda.synchronize()\n"
        "assert torch.equal(out, x) and float(g.sum()) == 58649.0\n"
        "dist.destroy_process_group(); print('rccl ok')\n")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_PORT=str(port)); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "rccl ok" in p.stdout, p.stderr[-2000:]
