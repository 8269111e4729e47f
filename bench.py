#!/usr/bin/env python3
"""bench.py — env steps/sec of the batched Tetris step on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (one step kernel) over one batch of 65 536 envs per GPU (10x20 board,
NumPy rule-set, uniform random actions resident in HBM, autoreset so the batch never drains).  One
process per GPU; envs are sharded by contiguous ranges with NO data-path collective (weak scaling);
torch.distributed (gloo) is used only for the barrier and the max-over-ranks of the timing.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tetris_gymnasium_amd import Engine, _native as N  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def alg_bytes_per_step(W, H, obs):
    """board read + observation write + board write-back (p_lock fixed at 1.0) + 76 B of
    packed scalar state / action / outputs.  Dict surface: padded board + mask + holder + queue written."""
    if obs == "dict":
        return H * W + H * W + (2 * (H + 4) * (W + 8) + 16 + 64) + 76
    return 3 * H * W + 76


def cpu_baseline(W, H, Q, budget_s=12.0):
    """The oracle's C restatement (kind "port") on this host's cores, bounded sample of the same workload."""
    from oracle import pyoracle as po

    try:
        avail = len(os.sched_getaffinity(0))  # cores this process may actually use (cgroup/affinity), not the box total
    except AttributeError:
        avail = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota of the container (e.g. "1600000 100000" = 16 cores on a 256-CPU host)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            avail = min(avail, max(1, int(q) // int(per)))
    except Exception:
        pass
    cores = max(1, min(avail, 64))  # OpenMP over envs
    B = 16384
    o = po.OracleBatch(B, W, H, po.RULESET_NUMPY, True, Q, nthreads=cores)
    o.reset(np.arange(1, B + 1, dtype=np.uint64))
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 8, (64, B)).astype(np.int32)
    for t in range(8):
        o.step(acts[t], autoreset=po.AUTORESET_SAME_STEP)
    n = 0
    t0 = time.perf_counter()
    while True:
        o.step(acts[n % 64], autoreset=po.AUTORESET_SAME_STEP)
        o.observe(po.OBS_BOARD)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": B * n / dt, "unit": "env steps/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{B} envs x {n} steps of the same workload (C oracle, OpenMP over envs, incl. observation pack)"}


class Dist:
    """torch.distributed plumbing for N > 1 (one process per GPU): rendezvous, barrier, max-over-ranks.
    No data-path collective exists — envs are independent shards."""

    def __init__(self, backend="gloo"):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.pg = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.pg = dist

    def barrier(self):
        if self.pg is not None:
            self.pg.barrier()

    def max(self, *values):
        if self.pg is None:
            return values
        import torch

        t = torch.tensor(values, dtype=torch.float64)
        self.pg.all_reduce(t, op=self.pg.ReduceOp.MAX)
        return tuple(float(v) for v in t)

    def shard(self, envs_per_rank):
        """Weak scaling: rank r owns global envs [r*B, (r+1)*B), seeds 1 + global index."""
        lo = self.rank * envs_per_rank
        return lo, lo + envs_per_rank, np.arange(1 + lo, 1 + lo + envs_per_rank, dtype=np.uint64)

    def close(self):
        if self.pg is not None:
            self.pg.barrier()
            self.pg.destroy_process_group()


def throughput(world, envs_per_rank, steps, max_elapsed_s):
    """Whole-job env steps/s: units all ranks processed / max-over-ranks time."""
    return world * envs_per_rank * steps / max_elapsed_s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--width", type=int, default=10)
    ap.add_argument("--height", type=int, default=20)
    ap.add_argument("--obs", choices=["board", "dict", "fn", "none", "rgb"], default="board")
    ap.add_argument("--ruleset", choices=["numpy", "functional"], default="numpy")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fuse", type=int, default=1,
                    help="transitions per launch of the timed rollout (TG_FUSE). Default 1: `value` is the "
                         "one-launch-per-transition number; the fused rollout is reported beside it")
    ap.add_argument("--no-fused-rollout", action="store_true")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE config 5: 20x40 board, queue_size 7, grouped hard-drop action set, legal actions sampled on device")
    args = ap.parse_args()

    D = Dist()
    rank, world, local = D.rank, D.world, D.local
    barrier = D.barrier

    W, H, B = args.width, args.height, args.batch
    Q = 4
    if args.config5:
        W, H, Q = 20, 40, 7
    ruleset = N.RULESET_NUMPY if args.ruleset == "numpy" else N.RULESET_FUNCTIONAL
    obs_mode = {"board": N.OBS_BOARD, "dict": N.OBS_DICT, "fn": N.OBS_FN, "none": N.OBS_NONE, "rgb": N.OBS_RGB}[args.obs]
    n_act = 8 if ruleset == N.RULESET_NUMPY else 7
    lo, hi, seeds = D.shard(B)
    ndev = N.load().tg_device_count()
    if ndev > 0:
        local = local % ndev  # (more ranks than devices only happens when testing the N > 1 path on a 1-GPU box)
    os.environ["TG_FUSE"] = str(args.fuse)  # read by tg_create
    e = Engine(B, W, H, ruleset=ruleset, queue_size=(Q if ruleset == N.RULESET_NUMPY else 7), obs_mode=obs_mode,
               autoreset=N.AUTORESET_SAME_STEP, device=local, env_offset=lo, grouped=args.config5)
    e.reset(seeds)

    # pre-generated uniform random actions, uploaded once: [T][B] int8 resident in HBM
    T = min(args.warmup + args.steps, 512)
    rng = np.random.default_rng(rank)
    acts = rng.integers(0, n_act, (T, B)).astype(np.int8)
    d_acts = e.dev_alloc(acts.nbytes)
    e.dev_upload(d_acts, acts)

    def run(k0, k):
        if args.config5:  # legal placements drawn on the device inside the step kernel (tg_grouped_rollout)
            e.grouped_rollout(k, seed=rank + 1, counter0=k0, fetch=False)
            return
        done = 0
        while done < k:  # cycle through the resident action rows
            row = (k0 + done) % T
            n = min(k - done, T - row)
            e.step_k_device(d_acts + row * B, n)
            done += n

    run(0, args.warmup)
    e.sync()
    barrier()
    e.sync()
    t0 = time.perf_counter()
    e.timer_start()
    run(args.warmup, args.steps)
    gpu_ms = e.timer_stop()  # HIP events on the engine's own stream
    e.sync()
    t1 = time.perf_counter()
    barrier()
    elapsed, gpu_ms = D.max(t1 - t0, gpu_ms)

    # The same rollout with the engine's default grouping (8 transitions per launch, state resident on chip in
    # between): reported beside `value`, never instead of it.
    fused = None
    if not args.config5 and not args.no_fused_rollout and args.fuse == 1 and args.obs in ("board", "fn", "none"):
        G = 8
        os.environ["TG_FUSE"] = str(G)
        e2 = Engine(B, W, H, ruleset=ruleset, queue_size=(Q if ruleset == N.RULESET_NUMPY else 7), obs_mode=obs_mode,
                    autoreset=N.AUTORESET_SAME_STEP, device=local, env_offset=lo)
        e2.reset(seeds)
        d2 = e2.dev_alloc(acts.nbytes)
        e2.dev_upload(d2, acts)
        ksteps = max(G, (min(args.steps, T) // G) * G)
        e2.step_k_device(d2, min(T, 64)); e2.sync()
        barrier(); e2.sync()
        f0 = time.perf_counter(); e2.timer_start()
        e2.step_k_device(d2, ksteps)
        f_gpu_ms = e2.timer_stop(); e2.sync()
        f1 = time.perf_counter()
        barrier()
        f_elapsed, f_gpu_ms = D.max(f1 - f0, f_gpu_ms)
        fb = H * W + 1 + (2 * H * W + 64 + 8) / G  # obs write + action; board read/write-back, record, outputs once per launch
        fused = {"transitions_per_launch": G, "steps": ksteps, "value": throughput(world, B, ksteps, f_elapsed),
                 "unit": "env steps/s", "ms_per_step": f_elapsed / ksteps * 1e3, "launch_ms": f_gpu_ms / (ksteps // G),
                 "alg_bytes_per_env_step": fb, "achieved_gbps": fb * B / (f_gpu_ms / ksteps * 1e-3) / 1e9,
                 "frac_of_hbm_peak": fb * B / (f_gpu_ms / ksteps * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                 "note": "tg_step_k with TG_FUSE=8: records, column masks and uint8 slabs stay in registers/LDS across the "
                         "8 transitions of a launch; the observation of every transition is still written"}
        e2.dev_free(d2)
        e2.close()

    if rank == 0:
        bytes_alg = alg_bytes_per_step(W, H, args.obs)
        launch_ms = gpu_ms / args.steps
        achieved = bytes_alg * B / (launch_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env steps/sec (whole node), 10x20 board, batch=65536 per GPU",
            "value": throughput(world, B, args.steps, elapsed),
            "unit": "env steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": (f"Batch {B} envs/GPU, {W}x{H} board, queue_size 7, grouped action set (column*4+rotation "
                                    f"→ teleport + hard drop), legal actions sampled on device, obs={args.obs}, same-step "
                                    f"autoreset") if args.config5 else
                                   f"Batch {B} envs/GPU, {W}x{H} board, {args.ruleset} rule-set, uniform random "
                                   f"actions (int8, resident in HBM), obs={args.obs}, same-step autoreset, "
                                   f"{args.fuse} transition(s) per launch",
                       "envs_per_gpu": B, "total_envs": world * B, "parallelism": f"env-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "step_kernel", "kernel_ms": launch_ms, "alg_bytes_per_env_step": bytes_alg},
        }
        if fused is not None:
            out["fused_rollout"] = fused
        try:
            out["hbm_copy_gbps"] = e.stream_copy_gbps(1 << 30, 10)
        except Exception as ex:  # pragma: no cover
            out["hbm_copy_gbps"] = None
            out["hbm_copy_error"] = str(ex)
        if world == 1 and not args.no_cpu_baseline:
            if not args.config5:
                out["cpu_baseline"] = cpu_baseline(W, H, Q)
                out["cpu_baseline"]["reference_python_single_env"] = \
                    "~3.4k steps/s on one 2.6 GHz Xeon core, measured in the survey container, not on this host"
        print(json.dumps(out), flush=True)
    e.dev_free(d_acts)
    e.close()
    D.close()


if __name__ == "__main__":
    main()
