#!/usr/bin/env python3
"""Multi-process form of the continuous-recording split (wifirx.dist.gather_recording): one process per rank, each
demodulating its ownership range + pre-roll + halo of BASELINE config 5's recording on the GPU, seam check between
neighbours, ONE all-gather of the PDUs; rank 0 compares with the one-rank run and prints one JSON line.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/recording_sharded.py

On an N-GPU node the backend is "nccl" (RCCL); on the one-GPU build pool WIFIRX_BENCH_BACKEND=gloo rehearses with all
ranks on the one device (labelled so in the output) -- the collective then goes through host memory.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    from test_gpu_recording import make_recording
    from wifirx import capi, dist as wdist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("WIFIRX_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(n_dev, 1)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    x, truth = make_recording()
    eng = wdist.gpu_stream_engine(max_sym=128, bandwidth=20e6, frequency=5.89e9, device=dev)
    t0 = time.perf_counter()
    res = wdist.gather_recording(x, eng, stride=320, device=coll_dev)
    dt = time.perf_counter() - t0
    digest = int(np.frombuffer(res["frames"].tobytes() + res["psdu"].tobytes(), dtype=np.uint8).astype(np.uint64).sum())
    digests = [None] * world
    dist.all_gather_object(digests, digest)
    if rank == 0:
        one_f, one_p = eng(x)
        same = bool(np.array_equal(res["frames"], one_f) and np.array_equal(res["psdu"], one_p[:, :320]))
        ok = (res["frames"]["flags"] & capi.F_CRC_OK) != 0
        print(json.dumps({"what": "config-5 recording (six images, %d samples) cut %d ways: pre-roll %d, halo %d samples"
                                  % (x.size, world, wdist.STREAM_PRE_ROLL, wdist.STREAM_HALO),
                          "backend": backend + ("" if backend == "nccl" else " [REHEARSAL: all ranks on %d GPU(s)]" % n_dev),
                          "world": world, "frames_per_rank": res["counts"], "frames": int(len(res["frames"])),
                          "crc_ok": int(ok.sum()), "equal_to_one_rank_run": same,
                          "every_rank_holds_the_same_stream": len(set(digests)) == 1,
                          "own_range": res["part"]["own"], "read_range": res["part"]["read"], "seconds": dt}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
