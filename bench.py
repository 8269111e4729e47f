#!/usr/bin/env python3
"""bench.py — env steps/sec of the batched Tetris step on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W                       (self-launching: N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (ONE launch of the step kernel) over one batch of 65 536 envs per GPU (10x20
board, NumPy rule-set, uniform random int8 actions resident in HBM, same-step autoreset so the batch never drains).
One process per GPU; envs are sharded by contiguous ranges (global env i = rank * 65536 + local i, seed 1 + i) with
NO data-path collective (weak scaling).  torch.distributed (gloo) carries only the barrier, the max-over-ranks of the
timing and the per-rank timing list.

`--gpus N` without a torchrun environment makes THIS process a launcher: it spawns N fresh rank processes
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and waits for them.  The launcher never loads the HIP
library and makes no HIP call — a process that has touched the GPU must not fork/exec ranks.  With more ranks than
devices (`--gpus 2` on a 1-GPU box) ranks share devices round-robin: that exercises the N > 1 path, it is not a
scaling number.

Output (rank 0).  The LAST line of stdout is ONE COMPACT JSON object (< 4 KB: metric, value, unit, n_gpus, steps, warmup,
ms_per_step, dtype, config, roofline, cpu_baseline, hbm_copy_gbps, per-rank timings and shard checksums, `blocks_file`, and a
three-number summary per further block).  At N = 1 the further BASELINE configs are measured as BLOCKS (config 2 = 4 096 envs,
one and sixteen transitions per launch; config 5 = 20x40 / queue 7 / grouped action set; the reference's dict observation; the
RGB wrapper's image; the fused rollout WITH its trajectory; the functional rule-set; the gymnasium.vector adapter; the
single-env drop-in class): each block is printed in full on its own EARLIER line, prefixed `[bench-block] <name> `, and the
whole set (with the notes, said once) is written to `bench_blocks.json` beside this file (TG_BENCH_BLOCKS_FILE overrides the
path).  Round 4 put everything into the last line; at 22 KB the driver no longer parsed it.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PROFILED = os.path.join(ROOT, "profiles", "latest.json")  # rocprofv3 figures of the committed profiles, per block


# ------------------------------------------------------------------------------------------------------------------
# launcher (parent process: no HIP, no engine import)
# ------------------------------------------------------------------------------------------------------------------
def free_port() -> int:
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def parse_cpulist(text: str):
    """'0-3,8,10-11' (sysfs cpulist) -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def cpus_near_pci_device(bdf: str, sysfs: str = "/sys"):
    """(numa_node, cpus) of the PCI device `bdf` ('0000:c1:00.0'): /sys/bus/pci/devices/<bdf>/numa_node and that node's cpulist;
    (-1, []) when the platform does not say (one node, a VM, no such file)."""
    try:
        node = int(open(os.path.join(sysfs, "bus", "pci", "devices", bdf.lower(), "numa_node")).read().strip())
        if node < 0:
            return -1, []
        return node, parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read())
    except (OSError, ValueError):
        return -1, []


def bind_near_device(lib, device: int, sysfs: str = "/sys") -> str:
    """Bind this rank's process (its launch thread, and the CPU baseline's threads on rank 0) to the cores of the NUMA node its GPU hangs
    off — on a two-socket host a launch thread on the far socket pays the cross-socket hop on every doorbell write and completion
    read.  Returns what was done, for the bench line (`per_rank_binding`).  TG_BENCH_BIND=0 switches it off."""
    import ctypes

    if os.environ.get("TG_BENCH_BIND", "1") == "0":
        return "off (TG_BENCH_BIND=0)"
    buf = ctypes.create_string_buffer(64)
    if not hasattr(lib, "tg_device_pci_bus_id") or lib.tg_device_pci_bus_id(int(device), buf, 64) != 0:
        return "unbound (no PCI bus id)"
    bdf = buf.value.decode()
    node, cpus = cpus_near_pci_device(bdf, sysfs)
    if not cpus or not hasattr(os, "sched_setaffinity"):
        return f"unbound ({bdf}: numa_node {node})"
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))  # (a cgroup / taskset of the job stays in force)
    if not allowed:
        return f"unbound ({bdf}: node {node} has no allowed cpu)"
    os.sched_setaffinity(0, allowed)
    return f"{bdf} node {node} cpus {allowed[0]}-{allowed[-1]} ({len(allowed)})"


def device_for_rank(local: int, world: int, ndev: int, share: bool):
    """(device ordinal, error text or None): one process per GPU.  More ranks than visible devices is an error — a mis-provisioned
    8-GPU run would otherwise produce a plausible-looking line from ranks that silently share devices — unless --share-devices
    says it is meant (exercising the N > 1 path on a smaller box; the line then carries a note)."""
    if ndev <= 0:
        return -1, "no HIP device visible (this engine has no CPU fallback)"
    if world > ndev and not share:
        return -1, (f"--gpus {world} but only {ndev} HIP device(s) visible: one process per GPU needs {world}; pass --share-devices to let "
                    f"ranks share devices round-robin (tests the N > 1 path, not a scaling measurement)")
    return local % ndev, None


def launch_ranks(n: int, argv, script: str | None = None, timeout_s: float = 3600.0) -> int:
    """Spawn `n` fresh rank processes of `script` (default: this file) with the torchrun environment contract and wait.
    Returns the first non-zero exit code (the other ranks are then terminated by PID), else 0.  Makes no HIP call."""
    script = script or os.path.abspath(__file__)
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TG_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=env))
    deadline = time.monotonic() + timeout_s
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
        if rc != 0 or time.monotonic() > deadline:
            for p in live:  # exactly the processes started above
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            if rc == 0:
                rc = 124
            break
        time.sleep(0.02)
    return rc


# ------------------------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------------------------
class Dist:
    """torch.distributed plumbing for N > 1 (one process per GPU): rendezvous, barrier, max-over-ranks, timing gather.
    No data-path collective exists — envs are independent shards."""

    def __init__(self, backend="gloo"):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.pg = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:  # (no fixed default: two jobs on one host would meet on it)
                raise SystemExit("bench.py: WORLD_SIZE > 1 without MASTER_PORT — launch with `python bench.py --gpus N` (picks a free "
                                 "port) or torchrun / torch.distributed.run")
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

    def gather(self, value: float):
        """The value of every rank, in rank order (timing metadata only)."""
        if self.pg is None:
            return [float(value)]
        import torch

        mine = torch.tensor([value], dtype=torch.float64)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.pg.all_gather(parts, mine)
        return [float(p[0]) for p in parts]

    def gather_obj(self, value):
        """A small picklable value of every rank, in rank order (metadata: device ordinals, core bindings)."""
        if self.pg is None:
            return [value]
        parts = [None] * self.world
        self.pg.all_gather_object(parts, value)
        return parts

    def gather_u64(self, value: int):
        """A 64-bit integer of every rank, in rank order (the shard checksums: metadata, not data path)."""
        if self.pg is None:
            return [int(value)]
        import torch

        mine = torch.tensor([int(value) - (1 << 64) if value >= (1 << 63) else int(value)], dtype=torch.int64)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.pg.all_gather(parts, mine)
        return [int(p[0]) & 0xFFFFFFFFFFFFFFFF for p in parts]

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


def global_actions(T, lo, hi, n_act, salt=0):
    """Uniform pseudo-random actions int8 [T][hi - lo] that are a function of (transition t, GLOBAL env index) only — a 64-bit
    mix of both — so every sharding of the batch over ranks plays the same game in env i ("1-GPU vs 8-GPU boards
    identical"; the shard checksums below are what shows it)."""
    t = np.arange(T, dtype=np.uint64)[:, None]
    g = np.arange(lo, hi, dtype=np.uint64)[None, :]
    x = (t << np.uint64(32)) ^ g ^ (np.uint64(salt) << np.uint64(56))
    x = x + np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return ((x >> np.uint64(33)) % np.uint64(n_act)).astype(np.int8)


def shard_checksum(board, rec, rng) -> int:
    """64 bits over a shard's final state as tg_get_state returns it (boards, env records, generator states)."""
    import hashlib

    h = hashlib.blake2b(digest_size=8)
    for a in (board, rec, rng):
        h.update(np.ascontiguousarray(a).tobytes())
    return int.from_bytes(h.digest(), "little")


NOTES = {
    "roofline": ("achieved = algorithmic bytes per env-step (SURVEY 8(d); DESIGN.md §3a) x envs / kernel time per launch; `clock` names "
                 "the time used: the committed rocprofv3 kernel-trace mean (profiles/latest.json) when it is <= this run's wall-clock "
                 "ms_per_step, else this run's HIP-event median over `reps` repetitions of the timed region (both are always listed: "
                 "kernel_us_rocprof / kernel_us_hip_events).  Handles that keep the board as nibbles (kFeatPacked) are priced by "
                 "the bytes of THAT format; the uint8-formula figure is `frac_uint8_formula`, never `frac`."),
    "resident": ("blocks whose working set fits the 256 MB Infinity Cache (`resident: Infinity Cache`) are launch-floor + latency "
                 "bound; their HBM-peak fraction is nominal"),
    "fused": ("fused blocks run tg_rollout with per-transition trajectory slots (kFeatTraj): every observation and output record "
              "written is readable by a consumer; `last_obs_only_us_per_step` is the same launch through tg_step_k, where each "
              "transition overwrites TG_BUF_OBS (round 4's figure, dead stores)"),
    "cpu_baseline": ("the oracle's C restatement (kind `port`), the same workload as the GPU block (same B, seeds 1 + i, actions of "
                     "the same generator, same-step autoreset, observation packed every step), OpenMP over envs; the reference's own "
                     "NumPy Tetris.step runs ~3.4k steps/s on one core (survey container, BASELINE.md §2)"),
    "vector_env": ("TetrisVectorEnv.step at 65 536 envs on one GPU, median of three regions: `host` returns NumPy observations "
                   "(gymnasium contract; `host_pinned`: copy=False, views of page-locked buffers), `device` returns DeviceArray views "
                   "(__cuda_array_interface__ / DLPack) and copies nothing back, `device_actions` additionally takes the actions from a "
                   "device pointer; Python overhead included"),
    "single_env": ("BASELINE config 1 through the drop-in class Tetris(): one env per launch is all launch boundary + PCIe latency; "
                   "the reference's NumPy Tetris.step: ~3.4k steps/s on one 2.6 GHz Xeon core (survey container)"),
    "vector_env_shards": ("8 192 envs per shard, step(actions, to_host=False) with per-shard device action pointers; host_us_per_step = "
                          "time to enqueue (Python + one tg_step_fetch_many call for all shards); k shards on ONE GPU serialise their "
                          "kernels"),
    "functional": "functional rule-set: the jax threefry piece sequence is parity-unpinned (no jax in the image, DESIGN §5); engine == oracle is tested",
}


def alg_bytes_per_step(W, H, obs, Q=4):
    """board read + observation write + board write-back (p_lock fixed at 1.0) + 76 B of packed scalar
    state / action / outputs.  Dict surface: padded board + mask + holder + queue strips written instead of H*W."""
    if obs == "dict":
        return H * W + H * W + (2 * (H + 4) * (W + 8) + 16 + 16 * Q) + 76
    if obs == "rgb":  # RgbObservation (wrappers/observation.py:38-74): (H+4) x (W+8+4Q) pixels x 3 bytes written
        return H * W + H * W + 3 * (H + 4) * (W + 8 + 4 * max(Q, 1)) + 76
    return 3 * H * W + 76


def host_threads() -> int:
    """Cores this process may actually use: affinity mask and cgroup v2 quota, not the box total."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            avail = min(avail, max(1, int(q) // int(per)))
    except Exception:
        pass
    return max(1, min(avail, 64))


def cpu_baseline(W, H, Q, B, *, grouped=False, obs="board", budget_s=6.0, ruleset="numpy"):
    """SURVEY §8d(i) / BASELINE.md §3: the oracle's C restatement (kind "port") on this host, the SAME workload as the GPU
    block (same B, seeds 1 + i, actions of the same generator, same-step autoreset, observation packed every step),
    timed at 1 thread and at all usable cores on a bounded number of steps (`budget_s` seconds per thread count).
    `value` is the all-cores rate."""
    from oracle import pyoracle as po

    mode = {"board": po.OBS_BOARD, "dict": po.OBS_DICT, "rgb": po.OBS_RGB, "fn": po.OBS_FN}[obs]
    fn = ruleset == "functional"
    full_B = B
    if grouped:  # the wrapper's step re-derives the legal mask from all 4W placements (wrappers/grouped.py:124-207,
        B = min(B, 8192)  # as the reference does): ~40 us per env-step on one core — sample the first 8 192 envs
    seeds = np.arange(1, B + 1, dtype=np.uint64)
    rng = np.random.default_rng(0)
    T = 64
    acts = global_actions(T, 0, B, 7 if fn else 8).astype(np.int32)
    res = {}
    cores = host_threads()
    for nt in ([1] if cores == 1 else [1, cores]):
        o = po.OracleBatch(B, W, H, po.RULESET_FUNCTIONAL if fn else po.RULESET_NUMPY, True, 7 if fn else Q, nthreads=nt)
        o.reset(seeds)

        def one(t):
            if grouped:  # uniform over the legal placements (drawn outside the timed sections)
                mask = o.grouped_enumerate()
                u = rng.random(B)
                cnt = mask.sum(1)
                k = np.minimum((u * cnt).astype(np.int64), cnt - 1)
                a = (np.cumsum(mask, 1) > k[:, None]).argmax(1).astype(np.int32)
                t0 = time.perf_counter()
                o.grouped_step(a, True, po.AUTORESET_SAME_STEP)
            else:
                t0 = time.perf_counter()
                o.step(acts[t % T], autoreset=po.AUTORESET_SAME_STEP)
            o.observe(mode)
            return time.perf_counter() - t0

        for t in range(4):
            one(t)
        n, spent, wall0 = 0, 0.0, time.perf_counter()
        while spent < budget_s and time.perf_counter() - wall0 < 3 * budget_s:
            spent += one(n)
            n += 1
        res[nt] = (B * n / spent, n)
    all_rate, all_n = res[cores]
    one_rate, one_n = res[1]
    return {"value": all_rate, "unit": "env steps/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
            "single_thread": {"value": one_rate, "cores": 1, "steps": one_n},
            "sample": f"{'first ' if B != full_B else ''}{B} envs x {all_n} steps at {cores} threads, x {one_n} at 1 thread"
                      f"{' (legal-action draw not timed)' if grouped else ''}"}


def profiled(block: str) -> dict:
    """rocprofv3 figures of the committed profiles for this block (profiles/latest.json), or {}."""
    try:
        return json.load(open(PROFILED)).get(block, {})
    except Exception:
        return {}


def run_block(D: Dist, name: str, *, W, H, Q, B, obs, ruleset, grouped, steps, warmup, reps, fuse=1, local=0,
              want_cpu=True, cpu_budget=6.0, copy_first=False, traj=False, checksum=False, secondary=True):
    """One benchmark block on every rank: W warm-up launches, EXACTLY `steps` timed launches bracketed by barrier + device
    sync on both sides (wall clock → value), then `reps` further repetitions of the same region timed with HIP events on
    the engine's stream (median → roofline).  `traj` (fused blocks): the launches go through tg_rollout with a [fuse][B]
    trajectory of observations and output records — every transition's stores are readable by a consumer."""
    from tetris_gymnasium_amd import Engine, _native as N

    rs = N.RULESET_NUMPY if ruleset == "numpy" else N.RULESET_FUNCTIONAL
    obs_mode = {"board": N.OBS_BOARD, "dict": N.OBS_DICT, "fn": N.OBS_FN, "none": N.OBS_NONE, "rgb": N.OBS_RGB}[obs]
    n_act = 8 if rs == N.RULESET_NUMPY else 7
    lo, hi, seeds = D.shard(B)
    e = Engine(B, W, H, ruleset=rs, queue_size=(Q if rs == N.RULESET_NUMPY else 7), obs_mode=obs_mode,
               autoreset=N.AUTORESET_SAME_STEP, device=local, env_offset=lo, grouped=grouped, fuse=fuse)  # (tg_config.fuse)
    e.reset(seeds)
    # pre-generated uniform random actions, uploaded once: [T][B] int8 resident in HBM, a function of the GLOBAL env index
    T = min(max(warmup + steps, fuse), 512)
    T = (T + fuse - 1) // fuse * fuse
    acts = global_actions(T, lo, hi, n_act)
    d_acts = e.dev_alloc(acts.nbytes)
    e.dev_upload(d_acts, acts)
    traj = bool(traj and fuse > 1 and not grouped)
    obs_bytes = e.lib.tg_obs_bytes_per_env(e.cfg) if traj else 0
    d_obs_traj = e.dev_alloc(fuse * B * obs_bytes) if traj and obs_bytes else 0
    d_out_traj = e.dev_alloc(fuse * B * 8) if traj else 0
    copy_gbps = None
    if copy_first:  # the practical ceiling, measured right before the warm-up (it also brings the clocks up: with the host-side
        try:        # action generation behind it the GPU idled for milliseconds before a short timed region)
            copy_gbps = e.stream_copy_gbps(1 << 30, 20)
        except Exception:  # pragma: no cover
            copy_gbps = None

    def run(k0, k, last_obs_only=False):
        if grouped:  # legal placements drawn on the device inside the step kernel (tg_grouped_rollout)
            e.grouped_rollout(k, seed=D.rank + 1, counter0=k0, fetch=False)
            return
        done = 0
        while done < k:  # cycle through the resident action rows
            row = (k0 + done) % T
            n = min(k - done, T - row)
            if traj and not last_obs_only:  # one launch per call: its trajectory slots are the consumer's to read
                n = min(n, fuse)
                e.rollout_device(d_acts + row * B, n, d_obs_traj, d_out_traj)
            else:
                e.step_k_device(d_acts + row * B, n)
            done += n

    run(0, warmup)
    e.sync()
    D.barrier()
    e.sync()
    t0 = time.perf_counter()
    run(warmup, steps)
    e.sync()
    t1 = time.perf_counter()
    D.barrier()
    (elapsed,) = D.max(t1 - t0)
    per_rank_ms = D.gather((t1 - t0) / steps * 1e3)
    # repetitions of the same region timed with HIP events on the engine's own stream (kept out of the wall-clock region
    # above: two event records and their host round trips are a measurable share of a 20-launch region)
    rep_us = []
    k0 = warmup + steps
    for _ in range(max(1, reps)):
        e.timer_start()
        run(k0, steps)
        rep_us.append(e.timer_stop() / steps * 1e3)
        k0 += steps
    last_only_us = None
    if traj and secondary:  # the same launches through tg_step_k (every transition overwrites TG_BUF_OBS): round 4's figure, kept as a secondary
        lo_us = []
        for _ in range(3):
            e.timer_start()
            run(k0, steps, last_obs_only=True)
            lo_us.append(e.timer_stop() / steps * 1e3)
            k0 += steps
        last_only_us = statistics.median(lo_us)
    hip_us = statistics.median(rep_us)
    (hip_us,) = D.max(hip_us)
    rank_devices = [int(x) for x in D.gather(float(local))]  # HIP device ordinal of every rank
    sums = None
    if checksum:  # what this shard's envs look like after warmup + (1 + reps) x steps transitions (deterministic)
        sums = D.gather_u64(shard_checksum(*e.get_state()))
    out = None
    if D.rank == 0:
        sym = e.step_kernel_name(fuse, grouped, traj)
        try:  # packed handles (kFeatPacked, FEAT bit 256): the board crosses HBM as nibbles
            feat_bits = int(sym.rstrip(">").split(",")[-1])
        except ValueError:
            feat_bits = 0
        packed = bool(feat_bits & 256)
        board_io = (H * W) // 2 if packed else H * W  # bytes of one board read / one write-back
        if fuse == 1:
            bytes_alg = alg_bytes_per_step(W, H, obs, Q) - 2 * (H * W - board_io)
            bytes_u8 = alg_bytes_per_step(W, H, obs, Q)
        else:  # per transition: observation + action (+ the 8-byte output record with a trajectory); per launch: board read /
            #    write-back, env record, last outputs
            bytes_alg = H * W + 1 + (8 if traj else 0) + (2 * board_io + 64 + 8) / fuse
            bytes_u8 = H * W + 1 + (8 if traj else 0) + (2 * H * W + 64 + 8) / fuse
        prof = profiled(name)
        wall_us = elapsed / steps * 1e6
        prof_us = prof.get("kernel_us")
        prof_us_step = None if prof_us is None else prof_us / fuse
        use_prof = prof_us_step is not None and prof_us_step <= wall_us
        kernel_us = prof_us_step if use_prof else hip_us
        achieved = bytes_alg * B / (kernel_us * 1e-6) / 1e9
        # the same time against the bytes the counters saw (committed profile of this block; per launch): the algorithmic
        # figure charges a board write-back per transition (p_lock = 1.0) that the kernel only does on a lock
        traffic = prof.get("hbm_bytes_per_launch")
        out = {
            "value": throughput(D.world, B, steps, elapsed), "unit": "env steps/s", "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "per_rank_ms_per_step": per_rank_ms,
            "per_rank_device": rank_devices,
            "config": {"workload": (f"Batch {B} envs/GPU, {W}x{H} board, queue_size {Q}, "
                                    + ("grouped action set (column*4+rotation -> teleport + hard drop), legal actions drawn "
                                       "on device inside the step kernel, " if grouped else
                                       f"{ruleset} rule-set, uniform random actions (int8, resident in HBM), ")
                                    + f"obs={obs}, same-step autoreset, {fuse} transition(s) per launch"
                                    + (" with per-transition trajectory slots (tg_rollout)" if traj else "")),
                       "envs_per_gpu": B, "total_envs": D.world * B,
                       "parallelism": f"env-sharded x{D.world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": prof.get("traffic_source"),
                         "kernel": sym, "kernel_us": kernel_us * fuse,
                         "clock": (f"rocprofv3 kernel trace ({prof.get('kernel_source')})" if use_prof
                                   else "HIP events on the engine's stream, this run"),
                         "kernel_us_hip_events": hip_us * fuse, "kernel_us_rocprof": prof_us,
                         "frac_hip_events": bytes_alg * B / (hip_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                         "kernel_us_per_step_reps": [round(x, 3) for x in rep_us],
                         "alg_bytes_per_env_step": bytes_alg,
                         "resident": "Infinity Cache" if bytes_alg * B < 200e6 else "HBM"},
        }
        if traffic:
            out["roofline"]["frac_measured_bytes"] = traffic / fuse / (kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBPS
        if packed:
            out["roofline"]["board_format"] = "nibbles in HBM (kFeatPacked): half the board bytes of SURVEY 8(d)'s uint8 formula"
            out["roofline"]["frac_uint8_formula"] = bytes_u8 * B / (kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBPS
        if last_only_us is not None:
            out["roofline"]["last_obs_only_us_per_step"] = last_only_us
        if copy_gbps is not None:
            out["hbm_copy_gbps"] = copy_gbps
        if sums is not None:
            out["shard_checksums"] = [f"{v:016x}" for v in sums]
            out["shard_transitions"] = warmup + (1 + max(1, reps)) * steps
        if want_cpu and obs in ("board", "dict", "rgb", "fn"):  # (rank 0 only, whatever the number of ranks)
            out["cpu_baseline"] = cpu_baseline(W, H, Q, B, grouped=grouped, obs=obs, budget_s=cpu_budget, ruleset=ruleset)
    for d in (d_obs_traj, d_out_traj, d_acts):
        if d:
            e.dev_free(d)
    e.close()
    return out


def vector_env_block(local, *, B=65536, steps=64, warmup=8):
    """env steps/s through the gymnasium.vector adapter (TetrisVectorEnv.step, what examples/train_ppo.py:240-245 calls):
    with the observation copied to host NumPy (the SyncVectorEnv contract), with the results left on the device as
    zero-copy views (`step(actions, to_host=False)`, actions still a host int32 array), and with the actions on the device
    too (a device pointer per step: what a policy running on the GPU hands over).  Wall clock of Python + C ABI + kernels
    + PCIe where there is any.  Never `value`: the rate of the drop-in surface."""
    from tetris_gymnasium_amd.vector import TetrisVectorEnv

    out = {}
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 8, (warmup + steps, B)).astype(np.int32)
    for kind in ("board", "dict"):
        res = {}
        for form in ("host", "host_pinned", "device", "device_actions"):
            if form in ("host", "host_pinned"):  # (host_pinned = gymnasium's copy=False: views of rotating page-locked slabs)
                if form == "host_pinned":
                    e0.dev_free(d_acts)
                    v.close()
                v = TetrisVectorEnv(B, observation=kind, autoreset_mode="same_step", devices=(local,), copy=(form == "host"))
                v.reset(seed=1)
                e0 = v.engines[0]
                d_acts = e0.dev_alloc(acts.nbytes)  # the last form: the policy's actions are on the device already (int32 [T][B])
                e0.dev_upload(d_acts, acts)
            act = (lambda t: d_acts + t * B * 4) if form == "device_actions" else (lambda t: acts[t])
            for t in range(warmup):
                v.step(act(t), to_host=form.startswith("host"))
            e0.sync()
            dts = []
            for _ in range(3):  # (median of three regions: a 64-step region is 0.5-20 ms, one host hiccup shows)
                t0 = time.perf_counter()
                for t in range(warmup, warmup + steps):
                    v.step(act(t), to_host=form.startswith("host"))
                e0.sync()
                dts.append(time.perf_counter() - t0)
            dt = statistics.median(dts)
            res[form] = {"value": B * steps / dt, "unit": "env steps/s", "us_per_step": dt / steps * 1e6,
                         "us_per_step_regions": [round(x / steps * 1e6, 2) for x in dts]}
        e0.dev_free(d_acts)
        obs_bytes = 200 if kind == "board" else 944
        res["host"]["pcie_bytes_per_env_step"] = res["host_pinned"]["pcie_bytes_per_env_step"] = obs_bytes + 25 + 4  # (+ TG_BUF_HOST_PACK + actions)
        res["device"]["pcie_bytes_per_env_step"] = 4  # the host actions
        res["device_actions"]["pcie_bytes_per_env_step"] = 0
        out[kind] = res
        v.close()
    return out


def single_env_block(local, *, steps=4000, warmup=200):
    """BASELINE config 1 through the drop-in class: `Tetris()` (reference envs/tetris.py:203-307) — one 10x20 env, random
    policy, `reset()` on done — steps/s of a plain `step` loop (one call into the C ABI per step, results copied to host
    NumPy), beside the oracle's single-env rate on this host (the reference's own ~3.4k steps/s was measured in the survey
    container: the Python reference cannot travel).  Also: the same loop when every step reads an attribute (`env.board`:
    one tg_get_state per step, and an upload before the next one since the array is mutable)."""
    from oracle import pyoracle as po
    from tetris_gymnasium_amd.envs.tetris import Tetris

    acts = np.random.default_rng(0).integers(0, 8, warmup + steps)
    out = {}
    for form in ("plain_loop", "reads_board_every_step"):
        env = Tetris(device=local)
        env.reset(seed=1)
        n = steps if form == "plain_loop" else steps // 8
        t0 = None
        for t in range(warmup + n):
            if t == warmup:
                t0 = time.perf_counter()
            _, _, term, _, _ = env.step(int(acts[t]))
            if form != "plain_loop":
                env.board  # noqa: B018 - the read is the point
            if term:
                env.reset()
        dt = time.perf_counter() - t0
        out[form] = {"value": n / dt, "unit": "env steps/s", "us_per_step": dt / n * 1e6, "steps": n}
        env.close()
    o = po.OracleBatch(1, 10, 20, po.RULESET_NUMPY, True, 4, nthreads=1)
    o.reset(np.array([1], np.uint64))
    t0 = time.perf_counter()
    for t in range(steps):
        _, term, _ = o.step(acts[t:t + 1].astype(np.int32), autoreset=po.AUTORESET_SAME_STEP)
        o.observe(po.OBS_DICT)
    dt = time.perf_counter() - t0
    out["cpu_port_single_env"] = {"value": steps / dt, "unit": "env steps/s", "cores": 1, "kind": "port",
                                  "sample": f"{steps} steps of ONE env through the C oracle's Python binding (step + dict observation)"}
    return out


def shards_block(local, *, per_shard=8192, steps=256, warmup=32):
    """TetrisVectorEnv(devices=(d,) * k): microseconds per step against the number of shards driven from ONE Python process
    (VERDICT r03 weak #7), device-resident results, actions a device pointer per shard.  All k shards share one GPU here
    (a 1-GPU box): what is measured is the host-side cost per shard — one call into the C ABI for all of them."""
    from tetris_gymnasium_amd.vector import TetrisVectorEnv

    out = {}
    for k in (1, 2, 4, 8):
        B = per_shard * k
        v = TetrisVectorEnv(B, observation="board", autoreset_mode="same_step", devices=(local,) * k)
        v.reset(seed=1)
        acts = np.random.default_rng(k).integers(0, 8, (warmup + steps, per_shard)).astype(np.int32)
        ptrs = []
        for e in v.engines:
            d = e.dev_alloc(acts.nbytes)
            e.dev_upload(d, acts)
            ptrs.append(d)
        for t in range(warmup):
            v.step([p + t * per_shard * 4 for p in ptrs], to_host=False)
        for e in v.engines:
            e.sync()
        regions = []
        for _ in range(3):  # (median region by total time)
            t0 = time.perf_counter()
            for t in range(warmup, warmup + steps):
                v.step([p + t * per_shard * 4 for p in ptrs], to_host=False)
            t_enq = time.perf_counter() - t0
            for e in v.engines:
                e.sync()
            regions.append((time.perf_counter() - t0, t_enq))
        dt, t_enq = sorted(regions)[1]
        out[f"k{k}"] = {"shards": k, "envs": B, "us_per_step": dt / steps * 1e6, "host_us_per_step": t_enq / steps * 1e6,
                        "value": B * steps / dt, "unit": "env steps/s",
                        "us_per_step_regions": [round(r[0] / steps * 1e6, 2) for r in regions]}
        # the same shards driven through tg_rollout in chunks of 16 transitions (TetrisVectorEnv.rollout_device): one enqueue per shard
        # and chunk instead of one per shard and transition — the form for one process in front of several devices
        K = steps // 16 * 16
        a8 = np.ascontiguousarray(acts[warmup:warmup + K].astype(np.int8))
        p8, po, pr = [], [], []
        for e in v.engines:
            d = e.dev_alloc(a8.nbytes); e.dev_upload(d, a8); p8.append(d)
            po.append(e.dev_alloc(K * per_shard * 200)); pr.append(e.dev_alloc(K * per_shard * 8))
        v.rollout_device(p8, 32, po, pr, chunk=16)
        for e in v.engines:
            e.sync()
        regions = []
        for _ in range(3):
            t0 = time.perf_counter()
            v.rollout_device(p8, K, po, pr, chunk=16)
            t_enq = time.perf_counter() - t0
            for e in v.engines:
                e.sync()
            regions.append((time.perf_counter() - t0, t_enq))
        dt, t_enq = sorted(regions)[1]
        out[f"k{k}"]["fused16"] = {"us_per_step": dt / K * 1e6, "host_us_per_step": t_enq / K * 1e6, "value": B * K / dt,
                                   "unit": "env steps/s", "note": "tg_rollout, 16 transitions per launch, trajectory slots on the device"}
        for e, d, d8, do, dr in zip(v.engines, ptrs, p8, po, pr):
            for x in (d, d8, do, dr):
                e.dev_free(x)
        v.close()
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--width", type=int, default=10)
    ap.add_argument("--height", type=int, default=20)
    ap.add_argument("--obs", choices=["board", "dict", "fn", "none", "rgb"], default="board")
    ap.add_argument("--ruleset", choices=["numpy", "functional"], default="numpy")
    ap.add_argument("--reps", type=int, default=7, help="repetitions of the timed region for the roofline median (>= 5)")
    ap.add_argument("--share-devices", action="store_true", help="--gpus N with fewer than N visible devices: let ranks share devices "
                                                                "round-robin (exercises the N > 1 path; not a scaling measurement)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=6.0, help="seconds of CPU work per thread count of the headline's cpu_baseline")
    ap.add_argument("--fuse", type=int, default=1,
                    help="transitions per launch of the timed rollout (TG_FUSE). Default 1: `value` is the "
                         "one-launch-per-transition number; the fused rollout is reported in blocks")
    ap.add_argument("--traj", action="store_true", help="with --fuse > 1: tg_rollout with per-transition trajectory slots")
    ap.add_argument("--no-secondary", action="store_true", help="profiling runs of the fused blocks: skip the tg_step_k (last observation "
                                                                "only) secondary, so that the trace holds ONE instantiation")
    ap.add_argument("--no-blocks", action="store_true", help="only the headline block (profiling runs)")
    ap.add_argument("--only-block", default=None, help="profiling runs: make this block of the default run the headline "
                                                       "(config2, config5, dict, rgb, fused_rollout, config2_fused, config2_dict, "
                                                       "config2_rgb, functional, functional_fused)")
    ap.add_argument("--config5", action="store_true",
                    help="make BASELINE config 5 the headline block: 20x40 board, queue_size 7, grouped hard-drop action "
                         "set, legal actions sampled on device")
    return ap.parse_args(argv)


def block_specs(ks, kw):
    """The further blocks of a default N = 1 run: name → run_block keyword arguments (profile name = key of profiles/latest.json)."""
    f16 = (ks + 15) // 16 * 16
    num = dict(W=10, H=20, Q=4, ruleset="numpy", grouped=False)
    fn = dict(W=10, H=20, Q=7, B=65536, obs="fn", grouped=False, ruleset="functional")
    return {
        "config2": dict(prof="config2", B=4096, obs="board", steps=ks, warmup=kw, cpu=True, **num),
        "config5": dict(prof="config5", W=20, H=40, Q=7, B=65536, obs="board", ruleset="numpy", grouped=True, steps=ks, warmup=kw, cpu=True),
        "dict": dict(prof="dict", B=65536, obs="dict", steps=ks, warmup=kw, cpu=True, **num),
        "rgb": dict(prof="rgb", B=65536, obs="rgb", steps=ks, warmup=kw, cpu=True, **num),
        # tg_rollout, 16 transitions per launch, records / column masks / uint8 slabs resident on chip in between, every
        # transition's observation and output record in its own trajectory slot
        "fused_rollout": dict(prof="fused16_traj", B=65536, obs="board", fuse=16, traj=True, steps=f16, warmup=64, cpu=False, **num),
        # config 2 again with 16 transitions per launch: at 4 096 envs (64 workgroups on 256 CUs) a one-transition launch is
        # all launch boundary; the fused launch is what a rollout at that batch size should use
        "config2_fused": dict(prof="config2_fused16_traj", B=4096, obs="board", fuse=16, traj=True, steps=f16, warmup=64, cpu=False, **num),
        # the two packed observation surfaces at config 2's batch (the regime RL runs with pixel / dict observations live in)
        "config2_dict": dict(prof="config2_dict", B=4096, obs="dict", steps=ks, warmup=kw, cpu=False, **num),
        "config2_rgb": dict(prof="config2_rgb", B=4096, obs="rgb", steps=ks, warmup=kw, cpu=False, **num),
        # the functional rule-set (envs/tetris_fn.py step / batched_step: the subsystem north_star names as replaced)
        "functional": dict(prof="functional", steps=ks, warmup=kw, cpu=True, **fn),
        "functional_fused": dict(prof="functional_fused16_traj", fuse=16, traj=True, steps=f16, warmup=64, cpu=False, **fn),
    }


def compact_line(out, blocks):
    """The LAST stdout line: the contract's keys, `roofline` and `cpu_baseline` without prose, one [value, frac, us] triple per block."""
    r = out["roofline"]
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data", "config", "per_rank_ms_per_step", "per_rank_device",
                                "n_devices_visible", "per_rank_binding") if k in out}
    line["per_rank_ms_per_step"] = [round(x, 6) for x in line["per_rank_ms_per_step"]]
    line["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_us", "clock",
                                              "kernel_us_hip_events", "kernel_us_rocprof", "frac_hip_events", "frac_measured_bytes",
                                              "alg_bytes_per_env_step", "resident") if r.get(k) is not None or k == "traffic"}
    if "cpu_baseline" in out:
        c = out["cpu_baseline"]
        line["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "sample": c["sample"],
                                "single_thread_value": c["single_thread"]["value"]}
    for k in ("hbm_copy_gbps", "shard_checksums", "shard_transitions", "note", "blocks_file"):
        if k in out:
            line[k] = out[k]
    if blocks:
        mini = {}
        for name, b in blocks.items():
            if isinstance(b, dict) and "roofline" in b:
                mini[name] = [float(f"{b['value']:.4g}"), round(b["roofline"]["frac"], 3), round(b["roofline"]["kernel_us"], 2)]
            elif isinstance(b, dict) and "error" in b:
                mini[name] = "error"
            elif name == "vector_env":  # env steps/s: NumPy observations back to the host | everything device-resident
                mini[name] = [float(f"{b['board']['host']['value']:.4g}"), float(f"{b['board']['device_actions']['value']:.4g}")]
            elif name == "single_env":  # Tetris().step loop, steps/s
                mini[name] = [float(f"{b['plain_loop']['value']:.4g}")]
            elif name == "vector_env_shards":  # host microseconds per step driving 1 / 8 shards from one Python process; ... 8 shards through tg_rollout
                mini[name] = [round(b["k1"]["host_us_per_step"], 2), round(b["k8"]["host_us_per_step"], 2),
                              round(b["k8"].get("fused16", {}).get("host_us_per_step", 0.0), 2)]
        line["blocks"] = mini
        line["blocks_key"] = "[env steps/s, roofline.frac, kernel us per launch]; full blocks: blocks_file and the '[bench-block]' lines above"
    return line


def rank_main(args) -> int:
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:  # checked before the rendezvous (a mismatched job would hang in it)
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: launch with `python bench.py --gpus N` (self-"
              f"launching) or torchrun --nproc-per-node N with the same N", file=sys.stderr)
        return 2
    D = Dist()
    from tetris_gymnasium_amd import _native as N

    lib = N.load()
    ndev = lib.tg_device_count()  # (first HIP call of this process: the launcher has long since forked)
    local, err = device_for_rank(D.local, D.world, ndev, args.share_devices)
    if err:
        print("bench.py: " + err, file=sys.stderr)
        D.close()
        return 3 if ndev <= 0 else 4
    binding = D.gather_obj(bind_near_device(lib, local))
    reps = max(5, args.reps)
    want_cpu = not args.no_cpu_baseline
    blocks = {}
    if args.only_block:  # a block of the default run as the headline of a profiling run (same arguments, same instantiation)
        spec = dict(block_specs(max(args.steps, 16), args.warmup)[args.only_block])
        spec.update(steps=(args.steps + spec.get("fuse", 1) - 1) // spec.get("fuse", 1) * spec.get("fuse", 1), warmup=args.warmup)
        prof, cpu = spec.pop("prof"), spec.pop("cpu")
        head = run_block(D, prof, reps=reps, local=local, want_cpu=want_cpu and cpu, cpu_budget=args.cpu_budget, copy_first=True,
                         secondary=not args.no_secondary, **spec)
        std_head = False
    else:
        W, H, B, Q = args.width, args.height, args.batch, 4
        grouped = False
        if args.config5:
            W, H, Q, grouped = 20, 40, 7, True
        pname = "config5" if args.config5 else (args.obs if args.obs in ("dict", "rgb") else
                                                ("functional" if args.ruleset == "functional" else "config3"))
        if args.fuse > 1:
            pname = {"config3": "fused", "functional": "functional_fused"}.get(pname, pname) + f"{args.fuse}" + ("_traj" if args.traj else "")
        head = run_block(D, pname, W=W, H=H, Q=Q, B=B, obs=args.obs, ruleset=args.ruleset, grouped=grouped, steps=args.steps,
                         warmup=args.warmup, reps=reps, fuse=args.fuse, local=local, want_cpu=want_cpu,
                         cpu_budget=args.cpu_budget, copy_first=True, traj=args.traj, checksum=not grouped, secondary=not args.no_secondary)
        std_head = not args.config5 and args.obs == "board" and args.ruleset == "numpy" and args.fuse == 1 \
            and (W, H, B) == (10, 20, 65536)
    if D.world == 1 and not args.no_blocks and std_head:
        for name, spec in block_specs(max(args.steps, 256), max(args.warmup, 64)).items():
            spec = dict(spec)
            prof, cpu = spec.pop("prof"), spec.pop("cpu")
            blocks[name] = run_block(D, prof, reps=reps, local=local, want_cpu=want_cpu and cpu, cpu_budget=min(3.0, args.cpu_budget), **spec)
            print(f"[bench-block] {name} " + json.dumps(blocks[name]), flush=True)
        for name, fn in (("vector_env", vector_env_block), ("single_env", single_env_block), ("vector_env_shards", shards_block)):
            try:
                blocks[name] = fn(local)
            except Exception as ex:  # pragma: no cover - adapter problems must not lose the headline line
                blocks[name] = {"error": repr(ex)}
            print(f"[bench-block] {name} " + json.dumps(blocks[name]), flush=True)
    # (N > 1: the further configs are single-GPU cases (BASELINE.json configs[1], [4]); config 4 IS this run at N = 8)
    if D.rank == 0:
        out = {"metric": "env steps/sec (whole node), 10x20 board, batch=65536 per GPU",
               "value": head["value"], "unit": "env steps/s", "n_gpus": D.world, "steps": head["steps"],
               "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u8", "data": "synthetic", "config": head["config"],
               "per_rank_ms_per_step": head["per_rank_ms_per_step"], "per_rank_device": head["per_rank_device"],
               "n_devices_visible": ndev, "per_rank_binding": binding, "roofline": head["roofline"]}
        if D.world > ndev:
            out["note"] = (f"{D.world} ranks on {ndev} visible device(s): ranks share devices round-robin — this exercises the "
                           f"N > 1 path, it is not a scaling measurement")
        for k in ("hbm_copy_gbps", "cpu_baseline", "shard_checksums", "shard_transitions"):
            if k in head:
                out[k] = head[k]
        full = dict(out, notes=NOTES)
        if blocks:
            full["blocks"] = blocks
            path = os.environ.get("TG_BENCH_BLOCKS_FILE", os.path.join(ROOT, "bench_blocks.json"))
            try:
                with open(path, "w") as f:
                    json.dump(full, f, indent=1)
                out["blocks_file"] = os.path.relpath(path, ROOT)
            except OSError as ex:  # (a read-only checkout: the '[bench-block]' lines above carry the same content)
                out["blocks_file"] = f"not written ({ex.__class__.__name__})"
        print(json.dumps(compact_line(out, blocks), separators=(",", ":")), flush=True)
    D.close()
    return 0


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, argv)  # this process stays HIP-free
    return rank_main(args)


if __name__ == "__main__":
    sys.exit(main())
