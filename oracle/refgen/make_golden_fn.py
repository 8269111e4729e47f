"""One-command pin of the FUNCTIONAL rule-set's piece sequence (jax threefry) — runs wherever `import jax` works.

TEST INFRASTRUCTURE.  The build container has no jax / chex, so the functional surface is labelled
"parity unpinned" for everything RNG-dependent until this script has been run ONCE on a machine with the reference's
pinned jax 0.5.3 + chex (pyproject.toml:14-15) and its output committed:

    PYTHONPATH=/root/reference:. python -m oracle.refgen.make_golden_fn        → tests/golden/fn_streams.npz

Only inputs and the reference's outputs are stored (keys, actions, boards; queues, keys, rewards, boards) — no source.
tests/test_oracle_fn_golden.py (CPU, oracle) and tests/test_gpu_fn_golden.py (GPU, engine) replay the file and `skip`
while it is absent.

What it pins (reference file:line):
  * jax.random.PRNGKey / split / permutation(key, arange(7))        functional/queue.py:35, 60; envs/tetris_fn.py:328
  * reset(): first bag, active piece, returned key, state.rng_key     envs/tetris_fn.py:318-367
  * 50 consecutive refills of bag_queue_get_next_element            functional/queue.py:38-67
  * a 64-env x 400-step random trace of step() incl. locks/refills   envs/tetris_fn.py:276-315, 370-413
  * the jnp.take(-H) wrap-to-row-0 corner of clear_filled_rows       functional/core.py:185-227
"""
from __future__ import annotations

import os
import sys
import zlib

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "fn_streams.npz")


def main() -> int:
    try:
        import jax
        import jax.numpy as jnp
    except ImportError:
        print("make_golden_fn: `import jax` failed — run this on a machine with jax 0.5.3 + chex installed; the functional "
              "surface stays 'parity unpinned' until tests/golden/fn_streams.npz exists", file=sys.stderr)
        return 3
    ref_root = os.environ.get("TG_REFERENCE", "/root/reference")
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    from tetris_gymnasium.envs.tetris_fn import reset, step
    from tetris_gymnasium.functional.core import EnvConfig, clear_filled_rows
    from tetris_gymnasium.functional.queue import bag_queue_get_next_element
    from tetris_gymnasium.functional.tetrominoes import TETROMINOES

    cfg = EnvConfig(width=10, height=20, padding=4, queue_size=7)
    out = {"jax_version": np.array(jax.__version__),
           "threefry_partitionable": np.array(int(bool(jax.config.jax_threefry_partitionable)))}

    # --- primitives -------------------------------------------------------------------------------------------
    seeds = np.array([0, 1, 2, 3, 42, 123, 65536, 2**31 - 1], np.uint64)
    keys = [jax.random.PRNGKey(int(s)) for s in seeds]
    out["seeds"] = seeds
    out["keys"] = np.stack([np.asarray(k, np.uint32) for k in keys])
    out["split"] = np.stack([np.asarray(jax.random.split(k), np.uint32) for k in keys])
    out["perm"] = np.stack([np.asarray(jax.random.permutation(k, jnp.arange(7)), np.uint8) for k in keys])
    # permutation(key, arange(n)) for every bag length EnvConfig.queue_size = n the engine accepts (functional/queue.py:35), padded with 255
    pn = np.full((len(keys), 7, 7), 255, np.uint8)
    for i, k in enumerate(keys):
        for n in range(1, 8):
            pn[i, n - 1, :n] = np.asarray(jax.random.permutation(k, jnp.arange(n)), np.uint8)
    out["perm_n"] = pn

    # --- reset + refill chains -----------------------------------------------------------------------------------
    rq, ra, rk, rs = [], [], [], []
    cp, ck, cq = [], [], []
    for k in keys:
        key, state, _ = reset(TETROMINOES, k, cfg)
        rq.append(np.asarray(state.queue, np.uint8)); ra.append(int(state.active_tetromino))
        rk.append(np.asarray(key, np.uint32)); rs.append(np.asarray(state.rng_key, np.uint32))
        q, kk = state.queue, state.rng_key
        pieces, kys, qs = [], [], []
        for _ in range(50):  # queue_index forced to queue_size: every call refills (functional/queue.py:59-67)
            piece, q, _, kk = bag_queue_get_next_element(cfg, q, 7, kk)
            pieces.append(int(piece)); kys.append(np.asarray(kk, np.uint32)); qs.append(np.asarray(q, np.uint8))
        cp.append(pieces); ck.append(np.stack(kys)); cq.append(np.stack(qs))
    out.update(reset_queue=np.stack(rq), reset_active=np.array(ra, np.uint8), reset_key=np.stack(rk),
               reset_state_key=np.stack(rs), chain_pieces=np.array(cp, np.uint8), chain_keys=np.stack(ck),
               chain_queues=np.stack(cq))

    # --- random trace (no autoreset: a finished env stays frozen, envs/tetris_fn.py:284-289) ---------------------
    B, T = 64, 400
    tseeds = np.arange(1000, 1000 + B, dtype=np.uint64)
    acts = np.random.default_rng(7).integers(0, 7, (T, B)).astype(np.int8)
    states = [reset(TETROMINOES, jax.random.PRNGKey(int(s)), cfg)[1] for s in tseeds]
    jstep = jax.jit(lambda st, a: step(TETROMINOES, st, a, cfg))
    rew = np.zeros((T, B), np.float32); term = np.zeros((T, B), np.uint8); lines = np.zeros((T, B), np.uint8)
    crc = np.zeros((T, B), np.uint32)
    for t in range(T):
        for i in range(B):
            states[i], obs, r, d, info = jstep(states[i], int(acts[t, i]))
            rew[t, i], term[t, i], lines[t, i] = float(r), int(bool(d)), int(info["lines_cleared"])
            crc[t, i] = zlib.crc32(np.asarray(obs, np.int8).tobytes())
    out.update(trace_seeds=tseeds, trace_actions=acts, trace_reward=rew, trace_terminated=term, trace_lines=lines,
               trace_obs_crc=crc,
               final_board=np.stack([np.asarray(s.board, np.int8) for s in states]),
               final_active=np.array([int(s.active_tetromino) for s in states], np.uint8),
               final_rot=np.array([int(s.rotation) for s in states], np.uint8),
               final_x=np.array([int(s.x) for s in states], np.int32),
               final_y=np.array([int(s.y) for s in states], np.int32),
               final_queue=np.stack([np.asarray(s.queue, np.uint8) for s in states]),
               final_queue_index=np.array([int(s.queue_index) for s in states], np.uint8),
               final_score=np.array([float(s.score) for s in states], np.float32),
               final_key=np.stack([np.asarray(s.rng_key, np.uint32) for s in states]))

    # --- jnp.take corner: cleared rows take gather index -H, which wraps to row 0 of the OLD board -----------------
    rng = np.random.default_rng(3)
    boards_in, boards_out, ns = [], [], []
    for case in range(12):
        pf = np.zeros((20, 10), np.int8)
        pf[0] = rng.integers(0, 2, 10) * rng.integers(2, 9)          # occupied cells in row 0
        if case % 3 == 0:
            pf[0] = 0
        full = rng.choice(np.arange(4, 20), size=1 + case % 4, replace=False)
        pf[full] = rng.integers(2, 9, (len(full), 10))
        part = rng.choice(np.setdiff1d(np.arange(1, 20), full), size=5, replace=False)
        pf[part] = (rng.random((5, 10)) < 0.6) * rng.integers(2, 9, (5, 10))
        pf[part, rng.integers(0, 10, 5)] = 0
        board = np.pad(pf, ((0, 4), (4, 4)), constant_values=1).astype(np.int8)
        nb, n = clear_filled_rows(cfg, jnp.asarray(board))
        boards_in.append(board); boards_out.append(np.asarray(nb, np.int8)); ns.append(int(n))
    out.update(clear_in=np.stack(boards_in), clear_out=np.stack(boards_out), clear_n=np.array(ns, np.uint8))

    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), {k: getattr(v, "shape", None) for k, v in out.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
