"""`bench.py --gpus N` launches its own N ranks: the launcher and the one-line contract on two gloo ranks with a
stub body, and the loud refusal when fewer than N devices are visible. The reference fans its workers out from inside its CLI as well
(generate_vessel_graph.py:112-129)."""
import json
import os
import subprocess
import sys

import pytest

from octa_autosegmentation_amd.utils import launch, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def test_gpus_flag_launches_that_many_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "128", "--launcher-selftest"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["self_launched"] is True
    assert len(line["per_rank_value"]) == 2 and line["steps"] == 3 and line["warmup"] == 1
    # whole-job value = all ranks' units / MAX time over ranks: not above the sum of the per-rank rates, not below twice the slowest
    assert 2 * min(line["per_rank_value"]) * 0.999 <= line["value"] <= sum(line["per_rank_value"]) * 1.001
    (a0, a1), (b0, b1) = line["seed_ranges"]
    assert a1 < b0 and b0 - a0 == sharding.SEEDS_PER_RANK            # disjoint seed ranges, no data-path collective


def test_one_rank_is_not_relaunched():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--launcher-selftest"], capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["self_launched"] is False


def test_refuses_when_fewer_devices_are_visible():
    # this container has no GPU; a GPU box has one: --gpus 2 without the self-test body must fail before anything starts
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=_env(), timeout=300)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the refusal cannot be provoked here")
    assert r.returncode != 0
    assert "--gpus 2: only" in r.stderr and "nothing was started" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_gpus_must_match_the_launchers_world():
    assert launch.needs_self_launch(4, {}) and not launch.needs_self_launch(1, {}) and not launch.needs_self_launch(4, {"WORLD_SIZE": "4"})
    assert launch.check_world(4, {"WORLD_SIZE": "4"}) == 4 and launch.check_world(1, {}) == 1
    with pytest.raises(launch.LaunchError):
        launch.check_world(8, {"WORLD_SIZE": "2"})
    with pytest.raises(launch.LaunchError):
        launch.self_launch(BENCH, [], 8, n_visible=1)
