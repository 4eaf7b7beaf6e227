"""`python bench.py --gpus N` must really be N ranks (VERDICT r3 "missing" 1): the flag re-launches the script under
torch.distributed.run, refuses when the node has fewer GPUs, and refuses a rank count that differs from the flag."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _run(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, env=env, capture_output=True, text=True, timeout=600)


def test_gpus_flag_spawns_that_many_ranks_gloo():
    r = _run(["--gpus", "2", "--backend", "gloo", "--rendezvous_only"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rendezvous_only"] is True


def test_single_rank_stays_in_process():
    r = _run(["--gpus", "1", "--backend", "gloo", "--rendezvous_only"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_more_gpus_than_visible_is_refused():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    want = have + 1 if have else 2
    r = _run(["--gpus", str(max(want, 2))])
    assert r.returncode != 0
    assert f"{max(want, 2)} GPUs requested, {have} visible" in r.stderr


def test_rank_count_that_differs_from_the_flag_is_refused():
    r = _run(["--gpus", "4", "--backend", "gloo", "--rendezvous_only"],
             env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0
    assert "--gpus 4" in r.stderr and "1 rank" in r.stderr
