"""RCCL itself, as far as one GPU lets it run: `bench.py` with WIFIRX_BENCH_FORCE_DIST=1 takes the multi-rank code path
with ONE rank -- process group on the nccl backend (= RCCL on ROCm), barriers, the max-over-ranks all-reduce, the PDU
all-gather (`all_gather_into_tensor` issued on the library's own stream, chunk by chunk behind decode_mac) and the
`collective` fields of the JSON line.  What N > 1 adds -- more peers, xGMI -- cannot run on the build pool; the rank
arithmetic for it is covered on the CPU over gloo (tests/test_dist_gloo.py, tests/test_dist_recording.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_multi_rank_path_on_rccl_with_one_rank():
    env = dict(os.environ)
    env["WIFIRX_BENCH_FORCE_DIST"] = "1"
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = "29541"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "WIFIRX_BENCH_BACKEND", "WIFIRX_BENCH_STUB"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "40000", "--steps", "2", "--warmup", "1",
                        "--no-cpu", "--pdu-steps", "2", "--gather-chunks", "4"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    c = d["collective"]
    assert c["backend"] == "nccl" and c["world_size_seen"] == 1 and c["dist_world_size"] == 1
    assert c["rccl_version"] and not c["rccl_version"].startswith("unavailable")
    assert len(c["ranks_hosts"]) == 1 and c["ranks_hosts"][0]["rank"] == 0
    leg = d["pdu_leg"]
    assert leg["gather_chunks"] == 4 and leg["gather_consistent"] is True
    assert leg["frames_crc_ok_all_ranks"] == 40000 == d["frames_crc_ok"]
    assert leg["all_gather_ms"] is not None and leg["all_gather_ms"] > 0
    assert leg["gathered_bytes_per_rank"] == (320 + 32) * 40000
    assert d["n_gpus"] == 1 and d["frames_complete"] == 40000
