"""GPU: one block swept by several row-sharded engines gives bit-identical adjacency, level and
separating-set records to a single engine.  The ranks share the box's one MI355X and join through gloo on
host-staged buffers (tools/row_shard_run.py; on a multi-GPU node the same script runs one rank per GPU over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "row_shard_run.py")] + list(extra)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("extra", [
    ["--world", "2"],
    ["--world", "3", "--markers", "1100", "--max-level", "5", "--seed", "4"],
    ["--world", "2", "--option", "rows=0"],                      # level 1 on the pair kernel (work items)
    ["--world", "2", "--option", "fast=0", "--markers", "700"],   # exact arithmetic everywhere
    ["--world", "2", "--option", "queue_capacity=64", "--markers", "900"],  # recheck queue overflows -> local exact redo
])
def test_row_sharded_engines_match_single_engine(extra):
    out = run(*extra)
    assert out["identical_to_single_engine"], out
    assert out["records"] > 0 and out["level"] >= 1
    assert out["tests_all_ranks"] > 0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("extra", [
    ["--world", "2", "--oracle"],                                                   # Skeleton: adjacency + sepsets vs the oracle
    ["--world", "2", "--engine", "hetcor", "--oracle"],                             # cuskss, uniform ESS, time index (row kernel at level 1)
    ["--world", "3", "--engine", "het", "--oracle", "--markers", "1100", "--max-level", "3"],  # per-pair ESS (work-item kernels)
    ["--world", "2", "--engine", "hetcor", "--oracle", "--option", "rows=0", "--markers", "900"],
])
def test_row_sharded_engines_match_the_oracle(extra):
    """C3 is a cuskss config: the hetcor engine shards too (marks joined by the same unsigned MIN), and the sharded
    results are checked against the oracle, not only against a single engine"""
    out = run(*extra)
    assert out["identical_to_oracle"] and out["identical_to_single_engine"], out
    assert out["level"] >= 1 and out["tests_all_ranks"] > 0 and out["edges_final"] > 0


@pytest.mark.timeout(600)
def test_device_buffer_exchange_over_rccl_single_rank():
    """the on-device branch of the exchange (what one-rank-per-GPU runs use): D2D staging, sign-flip MIN over the
    nccl (= RCCL) backend, copy back -- with one rank the buffer must come back unchanged, none entries included"""
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import cigwas_amd as cg
from cigwas_amd import shard
for dtype in (np.int64, np.int32):
    host = np.array([5, -1, 0, 2 ** 20, -1, 7], dtype)
    d = cg.DeviceArray(host)
    ex = shard.make_min_exchange(device=0)
    assert ex(1, d.ptr, host.size, host.itemsize, True, None) == 0
    back = d.download(dtype, host.shape)
    assert np.array_equal(back, host), (back, host)
dist.destroy_process_group()
print("ok")
''' % ROOT
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=500)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
