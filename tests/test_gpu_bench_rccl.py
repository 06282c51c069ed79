"""GPU: the collectives of bench.py's N > 1 path through RCCL itself - on a communicator of ONE rank (RIR_BENCH_RCCL_SOLO=1), which is what a
one-GPU box can run: process-group set-up with a device id, the identity all-reduce and object gather, barriers, the timing all-reduce, the decoded
and the compressed exchange (all_gather_into_tensor on uint8 views of device tensors), the intact-shard check.  N real ranks are the driver's to run;
the control flow for N > 1 is covered on gloo (tests/test_distributed_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_exchange_runs_on_rccl_with_one_rank():
    env = dict(os.environ, RIR_BENCH_RCCL_SOLO="1")
    for k in ("RIR_BENCH_BACKEND", "RIR_BENCH_SHARE_GPU", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-abi", "--no-cpu-baseline",
                        "--frames", "200"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["bit_exact_roundtrip"] is True and line["value"] > 0
    assert line["ranks_seen"] == 1 and line["single_rank_communicator (rehearsal)"] is True
    assert line["rccl_version"] and not str(line["rccl_version"]).startswith("unknown"), line["rccl_version"]
    assert len(line["ranks"]) == 1 and line["ranks"][0]["device_key"]
    assert "exchange_error" not in line, line.get("exchange_error")
    assert line["backend"] == "rccl"
    assert line["decoded_allgather"]["every_shard_intact_on_every_rank"] is True
    assert line["compressed_allgather"]["every_shard_intact_on_every_rank"] is True
    assert line["value_with_exchange"] > 0 and line["value_with_compressed_exchange"] > 0
