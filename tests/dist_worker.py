"""Worker for tests/test_distributed_cpu.py: one rank of a world_size-2 gloo job.  Exercises bench.py's
distributed plumbing (rendezvous, barrier, max-over-ranks, weak-scaling shard/seed rule) with the CPU
oracle standing in for the GPU engine, and checks that the union of the shards equals the unsharded run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--expect-world", type=int, default=2)
    ap.add_argument("--fail-rank", type=int, default=-1)
    args = ap.parse_args()
    if args.fail_rank == int(os.environ.get("RANK", "0")):
        sys.exit(7)  # a rank that dies before the rendezvous: the launcher must end the job, not hang
    D = bench.Dist(backend="gloo")
    assert D.world == args.expect_world == 2
    B, T = 64, 40
    lo, hi, seeds = D.shard(B)
    assert (lo, hi) == (D.rank * B, (D.rank + 1) * B) and seeds[0] == 1 + lo
    o = po.OracleBatch(B, env_offset=lo)
    o.reset(seeds)
    acts_all = np.random.default_rng(0).integers(0, 8, (T, 2 * B))
    D.barrier()
    for t in range(T):
        o.step(acts_all[t, lo:hi], autoreset=po.AUTORESET_SAME_STEP)
    elapsed = 0.5 + D.rank  # rank 1 is "slower": the max over ranks must win
    (mx,) = D.max(elapsed)
    assert mx == 1.5
    assert D.gather(elapsed) == [0.5, 1.5]
    # shard checksums travel as full 64-bit integers (top bit included) and equal those of the unsharded run's halves
    assert D.gather_u64((0xF000000000000001 if D.rank else 0x7FFFFFFFFFFFFFFF)) == [0x7FFFFFFFFFFFFFFF, 0xF000000000000001]
    sums = D.gather_u64(bench.shard_checksum(*o.get_state()))
    assert bench.throughput(D.world, B, T, mx) == 2 * B * T / 1.5
    mine = torch.from_numpy(o.get_state()[0].copy())
    parts = [torch.empty_like(mine) for _ in range(2)]
    dist.all_gather(parts, mine)  # test-only gather; the product path has no collective
    if D.rank == 0:
        full = po.OracleBatch(2 * B)
        full.reset(np.arange(1, 2 * B + 1, dtype=np.uint64))
        for t in range(T):
            full.step(acts_all[t], autoreset=po.AUTORESET_SAME_STEP)
        assert np.array_equal(torch.cat(parts).numpy(), full.get_state()[0]), "sharded != unsharded"
        fb, fr, fg = full.get_state()
        assert sums == [bench.shard_checksum(fb[k * B:(k + 1) * B], fr[k * B:(k + 1) * B], fg[k * B:(k + 1) * B]) for k in range(2)]
        print("DIST_OK")
    D.close()


if __name__ == "__main__":
    main()
