"""The RCCL leg of the multi-GPU plumbing on the one GPU a test box has: a fresh child process under torch.distributed.run
(world size 1, backend nccl = RCCL) runs every collective bench.py uses on DEVICE tensors - barrier, all_reduce(MAX),
all_gather, scatter - and one bench.py step goes through that communicator.  It proves the leg imports, initialises and
moves device memory before an 8-GPU run ever happens (no scaling is measured here)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, os.path.join(%r, "video-stab_amd"))
    import numpy as np
    from vsamd import dist
    c = dist.Comm()                       # VS_DIST_FORCE=1: nccl at world size 1
    assert c.dist is not None and c.device == "cuda" and c.dist.get_backend() == "nccl"
    c.barrier()
    mx = c.max_over_ranks(3.5)
    allc = c.gather_counters([c.rank, 7.0, 9.0])
    nbytes = 4 * 64 * 48 * 3
    payload = (np.arange(nbytes) %% 251).astype(np.uint8)
    got, secs = c.fan_out([payload], nbytes, src=0)
    ok = bool(got.is_cuda and np.array_equal(got.cpu().numpy(), payload))
    c.device_sync()
    print(json.dumps({"world": c.world, "max": mx, "gathered": allc, "fanout_ok": ok, "fanout_s": secs}))
    c.close()
""") % ROOT


def _run(args, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", VS_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", "29531"] + args,
                          capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_rccl_collectives_on_device_tensors(gpu, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    r = _run([str(script)])
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["world"] == 1 and out["max"] == 3.5 and out["gathered"] == [[0.0, 7.0, 9.0]] and out["fanout_ok"] is True


def test_bench_step_through_the_rccl_communicator(gpu):
    r = _run([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--regions", "2", "--clip-frames", "16",
              "--no-extras", "--no-cpu-baseline"])
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["timed_frames_per_region_per_rank"] == [128]
