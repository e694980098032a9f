"""Worker for test_dist_gloo.test_self_launcher_two_ranks: what bench.py's ranks do around the timed region
(join the group from the launcher's environment, barrier, gather per-rank values, rank 0 prints ONE JSON line),
on the gloo backend."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eabnet_amd import dist  # noqa: E402

if __name__ == "__main__":
    rank, world, local = dist.env_rank()
    assert dist.init("gloo")
    dist.barrier()
    per_rank = dist.gather_over_ranks(10.0 + rank)
    worst = dist.max_over_ranks(10.0 + rank)
    if len(sys.argv) > 1 and sys.argv[1] == "fail" and rank == 1:
        sys.exit(3)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"world": torch.distributed.get_world_size(), "local": local, "per_rank": per_rank, "max": worst,
                          "master": os.environ["MASTER_ADDR"]}))
    else:
        print("noise from a non-zero rank must not reach the job's stdout")
    torch.distributed.destroy_process_group()
