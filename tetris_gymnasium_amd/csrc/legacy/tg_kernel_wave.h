// legacy/tg_kernel_wave.h — FIRST formulation of the step kernel: one env per wavefront (the design BASELINE.json's
// north_star names; 41.6 us per 65 536-env step, SALU-bound: DESIGN.md §4).  Not part of the product library: built only
// into libtetris_gym_hip_legacy.so (`make legacy`, -DTG_LEGACY_KERNELS) as an independent implementation for the GPU
// test-suite (tools/gpu_check.sh runs every parity test against it).
#pragma once
#include "../tg_common.h"

namespace tg {

// prefetchable per-env inputs: the slab's dwords (one or more per lane) and one word per lane holding the
// 32-byte record (lanes 0..7) and the action word (lanes >= 8).  All loads are unmasked: buffers carry
// tail padding (tg_engine.hip) so that lanes past the slab read harmless bytes.
template <int W, int H>
__device__ __forceinline__ void load_env(const KParams& p, const uint32_t env, const int lane,
                                         uint32_t (&bv)[Geo<W, H>::DPL], uint32_t& recv) {
  using G = Geo<W, H>;
  const uint32_t* gb = reinterpret_cast<const uint32_t*>(p.board + env * (uint32_t)G::CELLS);
#pragma unroll
  for (int k = 0; k < G::DPL; ++k) bv[k] = gb[lane + 64 * k];
  const uint32_t* rw = p.rec + env * 8u + (uint32_t)(lane & 7);
  const uint32_t* aw = reinterpret_cast<const uint32_t*>(p.actions) + (p.act_i8 ? (env >> 2) : env);
  recv = *(lane < 8 ? rw : aw);
}

template <int W, int H, int RULESET>
__device__ __forceinline__ void process_env(const KParams& p, const uint32_t env, const int lane, uint8_t* slab,
                                            const uint32_t (&bv)[Geo<W, H>::DPL], const uint32_t recv) {
  using G = Geo<W, H>;
  uint32_t* slab_dw = reinterpret_cast<uint32_t*>(slab);
  const uint32_t boff = env * (uint32_t)G::CELLS;  // byte offset of this env's slab (B*H*W < 2^31 checked host-side)
  uint32_t* gboard = reinterpret_cast<uint32_t*>(p.board + boff);
  // stage the playfield slab in LDS (the per-wave LDS region is 256*DPL bytes: unmasked writes are safe)
#pragma unroll
  for (int k = 0; k < G::DPL; ++k) slab_dw[lane + 64 * k] = bv[k];
  uint32_t d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = __builtin_amdgcn_readlane(recv, i);
  const uint32_t aword = __builtin_amdgcn_readlane(recv, 8);
  const int action_in = p.act_i8 ? (int)(int8_t)(aword >> (8 * (env & 3))) : (int)aword;
  const tg_rng_rec* rp = p.rng + env;
  const int Q = p.queue_size;
  wave_sync();

  // ---------------- fast path: plain move / rotate / no-op + gravity, nothing locks -----------------
  // (≈ 80 % of random-policy steps).  Straight-line, branch-free action handling with integer masks
  // (keeps the scalar unit on s_cmp/s_cselect instead of lane-mask juggling); anything rarer (lock, hard
  // drop, effective swap, game-over / pending-reset states) falls through to the general path below,
  // which recomputes everything from the raw record.
  {
    const uint32_t fl = d[1] >> 8;  // flags in bits 0..7
    uint32_t e;
    if (RULESET == TG_RULESET_NUMPY) {
      e = (uint32_t)(kActLutNumpy >> (8 * (action_in & 7)));
    } else {
      const int br = action_in < 0 ? 0 : (action_in > 6 ? 6 : action_in);
      e = (uint32_t)(kActLutFn >> (8 * br));
    }
    const uint32_t swap_eff = (e >> 7) & ~(fl >> 1) & 1u;  // swap requested and not yet swapped (LUT bit 7: NUMPY only)
    const uint32_t slow0 = (fl & (TG_F_GAME_OVER | TG_F_PENDING_RESET)) | (e & 0x40u) | swap_eff | (uint32_t)p.action_set;
    if (slow0 == 0) {
      const uint32_t piece = d[0] & 0xFFu, rot = (d[0] >> 8) & 0xFFu, x = (d[0] >> 16) & 0xFFu, y = d[0] >> 24;
      const uint64_t row = p.shapes[piece];
      const uint32_t tx = x + (e & 3u) - 1u;
      const uint32_t ty = y + ((e >> 2) & 1u);
      const uint32_t tr = (rot + (e >> 3)) & 3u;
      const uint32_t shapeC = (uint32_t)(row >> (16 * rot));  // bits above 15 are never tested (cell < 16)
      const uint32_t shapeT = (uint32_t)(row >> (16 * tr));
      const uint64_t m = test4<W, H>(slab, lane, shapeT, (int)tx, (int)ty, shapeC, (int)x, (int)y);
      const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
      const uint32_t accept = ((lo & 0xFFFFu) == 0 ? 1u : 0u) & (e >> 5);
      const uint32_t w = accept ? lo : hi;
      const uint32_t grav = (uint32_t)p.gravity;
      const uint32_t blocked = (w >> 16) != 0 ? grav : 0u;
      if (blocked == 0) {
        const uint32_t nx = accept ? tx : x;
        const uint32_t nr = accept ? tr : rot;
        const uint32_t ny = (accept ? ty : y) + grav;
        // NUMPY projection rule: the piece is drawn unless it collides where it stands (only possible
        // without gravity; after a gravity move the piece is free)
        const uint32_t hide = RULESET == TG_RULESET_NUMPY ? ((w & 0xFFFFu) != 0 ? (grav ^ 1u) : 0u) : 0u;
        emit_obs<W, H>(slab, lane, p.obs + boff, p.obs_mode, hide == 0, accept ? shapeT : shapeC, (int)nx, (int)ny,
                       (int)(d[1] & 0xFFu));
        const uint32_t nd0 = piece | (nr << 8) | (nx << 16) | (ny << 24);
        float reward = 0.0f;
        uint32_t nd6 = d[6];
        if (RULESET == TG_RULESET_FUNCTIONAL) {
          if (accept & (e >> 2) & 1u) {  // soft drop +1 (tetris_fn.py:183-190)
            const float old = __uint_as_float(d[6]);
            const float t = (old + 1.0f) + 0.0f;
            nd6 = __float_as_uint(t);
            reward = t - old;
          }
        }
        if (lane == 0) {
          uint32_t* rec_out = p.rec + env * 8u;
          rec_out[0] = nd0;
          if (RULESET == TG_RULESET_FUNCTIONAL && nd6 != d[6]) rec_out[6] = nd6;
          p.out[env] = make_uint2(__float_as_uint(reward), 0u);
          if (p.episode) {
            tg_episode_rec ep = p.episode[env];
            ep.ret += reward;
            ep.len += 1;
            p.episode[env] = ep;
          }
        }
        return;
      }
    }
  }

  // hot fields only
  int piece = d[0] & 0xFF, rot = (d[0] >> 8) & 0xFF, x = (d[0] >> 16) & 0xFF, y = d[0] >> 24;
  int val = d[1] & 0xFF;
  uint32_t flags = (d[1] >> 8) & 0xFF;

  float reward = 0.0f;
  int lines = 0;
  bool board_dirty = false;  // slab changed → write back
  bool full_dirty = false;   // more than d[0]/d[1] changed → store the whole record
  bool final_coll = false;   // NUMPY: the active piece collides at its final position (projection rule)
  Rng r;
  r.loaded = false;
  r.dirty = false;

  const bool pending = p.autoreset == TG_AUTORESET_NEXT_STEP && (flags & TG_F_PENDING_RESET);
  bool frozen = RULESET == TG_RULESET_FUNCTIONAL && (flags & TG_F_GAME_OVER);  // tetris_fn.py:284-289
  // grouped action set (wrappers/grouped.py:209-269): action = column * 4 + rotations → teleport + hard drop.
  // The wrapper's legal mask only looks at the frame (collision_with_frame, :104-122): a placement is illegal iff
  // one of the rotated piece's cells lies outside the playfield columns — a function of piece/rotation/column.
  int action = action_in;
  bool g_illegal = false;
  if (RULESET == TG_RULESET_NUMPY && p.action_set == TG_ACTIONS_GROUPED && !pending) {
    const int col = action_in >> 2, nrot = (rot + (action_in & 3)) & 3;
    const int nx = col + 4 - (piece == 0 ? 2 : 1);  // x + padding - matrix.shape[1] // 2
    const uint32_t sh = (uint32_t)(p.shapes[piece] >> (16 * nrot)) & 0xFFFFu;
    bool legal = (unsigned)action_in < (unsigned)(4 * W);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (sh & (0x1111u << c)) legal = legal && (unsigned)(nx + c - 4) < (unsigned)W;
    if (legal) { x = nx; rot = nrot; action = 5; }
    else {
      g_illegal = true;
      action = 7;
      if (p.grouped_term) {
        frozen = true;
        final_coll = test1<W, H>(slab, lane, (uint32_t)(p.shapes[piece] >> (16 * rot)) & 0xFFFFu, x, y);
      }
    }
  }
  if (pending) {
    EnvS s;
    raw_to_env(s, d);
    env_reset<W, H, RULESET>(s, r, rp, slab_dw, lane, Q);
    env_to_raw(s, d);
    piece = s.piece; rot = 0; x = s.x; y = 0; val = s.val; flags = 0;
    board_dirty = true;
    full_dirty = true;
  } else if (!frozen) {
    const uint64_t row = p.shapes[piece];
    const uint32_t shapeC = (uint32_t)(row >> (16 * rot)) & 0xFFFFu;
    uint32_t e;
    if (RULESET == TG_RULESET_NUMPY) {
      e = (uint32_t)(kActLutNumpy >> (8 * (action & 7))) & 0xFFu;
    } else {  // lax.switch clamps its index (envs/tetris_fn.py:214-241)
      const int br = action < 0 ? 0 : (action > 6 ? 6 : action);
      e = (uint32_t)(kActLutFn >> (8 * br)) & 0xFFu;
    }
    int tp = piece, tv = val;
    int tx = x + (int)(e & 3u) - 1;
    int ty = y + (int)((e >> 2) & 1u);
    int tr = (rot + (int)((e >> 3) & 3u)) & 3;
    const bool check = (e & 0x20u) != 0;
    const bool hard = (e & 0x40u) != 0;
    bool forced = false;
    uint64_t rowT = row;
    if (RULESET == TG_RULESET_NUMPY && (e & 0x80u) && !(flags & TG_F_HAS_SWAPPED)) {
      // swap (envs/tetris.py:242-252, components/tetromino_holder.py:31-49)
      EnvS s;
      raw_to_env(s, d);
      if (s.hp == TG_NONE) {
        tp = np_queue_pop(s, r, rp, Q);  // spawn_tetromino; its collision result is ignored (:248-250)
        tr = 0;
        tv = tp + 2;
      } else {
        tp = s.hp; tr = s.hr; tv = s.hv;
      }
      s.hp = piece; s.hr = rot; s.hv = val;
      env_to_raw(s, d);
      flags |= TG_F_HAS_SWAPPED;
      tx = np_spawn_x<W>(tp);
      ty = 0;
      forced = true;
      full_dirty = true;
      rowT = p.shapes[tp];
    }
    const uint32_t shapeT = (uint32_t)(rowT >> (16 * tr)) & 0xFFFFu;

    if (RULESET == TG_RULESET_FUNCTIONAL && hard) {
      // functional/core.py:230-251 hard_drop, then lock (tetris_fn.py:263-269)
      const uint64_t C = column_scan<W, H>(slab, lane, shapeC, x);
      const int yd = drop_row(C, y);
      const int drop_reward = 2 * (yd - y);
      y = yd;
      int lock_reward = 0;
      if (p.gravity || action == 6) {
        EnvS s;
        raw_to_env(s, d);
        s.piece = piece; s.rot = rot; s.x = x; s.y = y; s.flags = (int)flags;
        // (shared lock code below, functional flavour)
        stamp<W, H>(slab, lane, shapeC, x, y, (uint32_t)(piece + 2), true);
        wave_sync();
        lines = clear_rows<W, H, RULESET>(slab, lane);
        board_dirty = true;
        lock_reward = lines == 4 ? 800 : (lines > 0 ? lines * 200 - 100 : 0);
        rng_load(r, rp);
        uint32_t a0, a1;
        threefry2x32(r.k0, r.k1, 0u, 0u, a0, a1);
        if (s.bidx >= 7) {
          uint32_t b0, b1;
          threefry2x32(r.k0, r.k1, 0u, 1u, b0, b1);
          s.bag = jax_permutation7(b0, b1);
          s.bidx = 0;
        }
        s.piece = (int)((s.bag >> (8 * s.bidx)) & 0xFF);
        s.bidx += 1;
        r.k0 = a0; r.k1 = a1;
        r.dirty = true;
        s.rot = 0; s.val = s.piece + 2; s.x = (W + 8) / 2 - 2; s.y = 0;
        if (test1<W, H>(slab, lane, (uint32_t)p.shapes[s.piece] & 0xFFFFu, s.x, 0)) s.flags |= TG_F_GAME_OVER;
        piece = s.piece; rot = 0; x = s.x; y = 0; val = s.val; flags = (uint32_t)s.flags;
        const float old = s.score;
        float t = old + (float)drop_reward;
        t = t + (float)lock_reward;
        s.score = t;
        reward = t - old;
        env_to_raw(s, d);
        full_dirty = true;
      } else {
        float old = __uint_as_float(d[6]);
        float t = old + (float)drop_reward;
        t = t + 0.0f;
        d[6] = __float_as_uint(t);
        reward = t - old;
        full_dirty = true;
      }
    } else {
      const uint64_t m = test4<W, H>(slab, lane, shapeT, tx, ty, shapeC, x, y);
      const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
      const bool accept = forced || (check && (lo & 0xFFFFu) == 0);
      uint32_t shape = shapeC;
      bool coll_here, coll_below;
      if (accept) {
        piece = tp; rot = tr; x = tx; y = ty; val = tv; shape = shapeT;
        coll_here = (lo & 0xFFFFu) != 0;
        coll_below = (lo >> 16) != 0;
      } else {
        coll_here = (hi & 0xFFFFu) != 0;
        coll_below = (hi >> 16) != 0;
      }
      bool lock = false;
      if (RULESET == TG_RULESET_NUMPY) {
        final_coll = coll_here;
        lock = hard;
        if (p.gravity && !hard) {  // envs/tetris.py:259-264
          if (!coll_below) { y += 1; final_coll = false; }
          else lock = true;
        }
      } else {
        if (p.gravity) {  // functional/core.py:150-182 + tetris_fn.py:242-247
          if (!coll_below) y += 1;
          else lock = true;
        }
      }
      int drop_reward = (RULESET == TG_RULESET_FUNCTIONAL && accept && (e & 4u)) ? 1 : 0;
      int lock_reward = 0;
      if (lock) {
        EnvS s;
        raw_to_env(s, d);
        s.flags = (int)flags;
        if (RULESET == TG_RULESET_NUMPY) {  // envs/tetris.py:450-479 commit_active_tetromino
          if (coll_here) {
            reward = 0.0f;  // rewards.game_over
            flags |= TG_F_GAME_OVER;
          } else {
            const uint64_t C = column_scan<W, H>(slab, lane, shape, x);
            const int yd = drop_row(C, y);  // :445-448 step-wise drop
            stamp<W, H>(slab, lane, shape, x, yd, (uint32_t)val, false);
            wave_sync();
            lines = clear_rows<W, H, RULESET>(slab, lane);
            board_dirty = true;
            const int np_ = np_queue_pop(s, r, rp, Q);  // :393-401 spawn_tetromino
            piece = np_; rot = 0; val = np_ + 2; x = np_spawn_x<W>(np_); y = 0;
            const bool sc = test1<W, H>(slab, lane, (uint32_t)p.shapes[np_] & 0xFFFFu, x, 0);
            reward = (float)(lines * lines * W + 1);  // score + rewards.alife
            if (sc) { flags |= TG_F_GAME_OVER; reward = 0.0f; }
            else flags &= ~(uint32_t)TG_F_GAME_OVER;
            flags &= ~(uint32_t)(TG_F_HAS_SWAPPED | TG_F_FULLROW);
            final_coll = sc;
            s.piece = piece; s.rot = 0; s.x = x; s.y = 0; s.val = val; s.flags = (int)flags;
            env_to_raw(s, d);
            full_dirty = true;
          }
        } else {  // envs/tetris_fn.py:370-413 place_active_tetromino
          stamp<W, H>(slab, lane, shape, x, y, (uint32_t)(piece + 2), true);
          wave_sync();
          lines = clear_rows<W, H, RULESET>(slab, lane);
          board_dirty = true;
          lock_reward = lines == 4 ? 800 : (lines > 0 ? lines * 200 - 100 : 0);  // core.py:124-146
          rng_load(r, rp);
          uint32_t a0, a1;
          threefry2x32(r.k0, r.k1, 0u, 0u, a0, a1);  // split(rng_key)[0] → new rng_key (:399)
          if (s.bidx >= 7) {                         // functional/queue.py:59-67 refill from split(rng_key)[1]
            uint32_t b0, b1;
            threefry2x32(r.k0, r.k1, 0u, 1u, b0, b1);
            s.bag = jax_permutation7(b0, b1);
            s.bidx = 0;
          }
          piece = (int)((s.bag >> (8 * s.bidx)) & 0xFF);
          s.bidx += 1;
          r.k0 = a0; r.k1 = a1;
          r.dirty = true;
          rot = 0; val = piece + 2; x = (W + 8) / 2 - 2; y = 0;
          if (test1<W, H>(slab, lane, (uint32_t)p.shapes[piece] & 0xFFFFu, x, 0)) flags |= TG_F_GAME_OVER;
          s.piece = piece; s.rot = 0; s.x = x; s.y = 0; s.val = val; s.flags = (int)flags;
          env_to_raw(s, d);
          full_dirty = true;
        }
      }
      if (RULESET == TG_RULESET_FUNCTIONAL) {
        const float old = __uint_as_float(d[6]);
        float t = old + (float)drop_reward;
        t = t + (float)lock_reward;
        if (t != old) { d[6] = __float_as_uint(t); full_dirty = true; }
        reward = t - old;  // tetris_fn.py:312
      }
    }
  }

  if (g_illegal) reward = -0.1f;  // rewards.invalid_action
  const int terminated = ((flags & TG_F_GAME_OVER) || (g_illegal && p.grouped_term)) ? 1 : 0;
  if (terminated && p.autoreset != TG_AUTORESET_NONE) {
    if (p.autoreset == TG_AUTORESET_SAME_STEP) {
      if (p.final_obs) {
        const bool show = RULESET == TG_RULESET_NUMPY ? !final_coll : false;
        if (board_dirty) wave_sync();
        emit_obs<W, H>(slab, lane, p.final_obs + boff, p.obs_mode, show,
                       (uint32_t)(p.shapes[piece] >> (16 * rot)) & 0xFFFFu, x, y, val);
      }
      EnvS s;
      d[0] = (uint32_t)piece | ((uint32_t)rot << 8) | ((uint32_t)x << 16) | ((uint32_t)y << 24);
      d[1] = (d[1] & 0xFFFF0000u) | (uint32_t)val | (flags << 8);
      raw_to_env(s, d);
      env_reset<W, H, RULESET>(s, r, rp, slab_dw, lane, Q);
      env_to_raw(s, d);
      piece = s.piece; rot = 0; x = s.x; y = 0; val = s.val; flags = 0;
      board_dirty = true;
      full_dirty = true;
      final_coll = false;
    } else {
      flags |= TG_F_PENDING_RESET;
    }
  }

  if (board_dirty) {  // write the slab back (only on lock / reset: ~18 % of random-policy steps)
    wave_sync();
#pragma unroll
    for (int k = 0; k < G::DPL; ++k) {
      const int idx = lane + 64 * k;
      if (idx < G::NDW) gboard[idx] = slab_dw[idx];
    }
  }
  {
    const bool show = RULESET == TG_RULESET_NUMPY ? !final_coll : !(flags & TG_F_GAME_OVER);
    emit_obs<W, H>(slab, lane, p.obs + boff, p.obs_mode, show, (uint32_t)(p.shapes[piece] >> (16 * rot)) & 0xFFFFu, x,
                   y, val);
  }
  d[0] = (uint32_t)piece | ((uint32_t)rot << 8) | ((uint32_t)x << 16) | ((uint32_t)y << 24);
  d[1] = (d[1] & 0xFFFF0000u) | (uint32_t)val | (flags << 8);
  rng_store(r, p.rng + env, lane);
  if (lane == 0) {
    uint32_t* rec_out = p.rec + env * 8u;
    if (full_dirty) {
      *reinterpret_cast<uint4*>(rec_out) = make_uint4(d[0], d[1], d[2], d[3]);
      *reinterpret_cast<uint4*>(rec_out + 4) = make_uint4(d[4], d[5], d[6], d[7]);
    } else {
      *reinterpret_cast<uint2*>(rec_out) = make_uint2(d[0], d[1]);
    }
    p.out[env] = make_uint2(__float_as_uint(reward), (uint32_t)terminated | ((uint32_t)lines << 8));
    if (p.episode) {
      tg_episode_rec ep = p.episode[env];
      if (pending) {  // the NEXT_STEP reset step belongs to no episode (gymnasium RecordEpisodeStatistics)
        ep.ret = 0.0f; ep.len = 0;
      } else {
        ep.ret += reward;
        ep.len += 1;
        if (terminated) { ep.last_ret = ep.ret; ep.last_len = ep.len; ep.ret = 0.0f; ep.len = 0; }
      }
      p.episode[env] = ep;
    }
  }
}

// Persistent launch: a fixed grid of waves, each walking envs  wave_id, wave_id + n_waves, ...  with the
// next env's slab / record / action already in flight while the current one is processed (the loads of
// env i+1 are issued before env i is touched).  Measured reason (profiles/r01_*): with one short-lived
// wave per env the kernel was bound by wave launch rate and by the serial load→use chains of each
// wave, with only half of the wave slots occupied on average.
template <int W, int H, int RULESET>
__global__ __launch_bounds__(kBlock) void step_kernel(const KParams p) {
  using G = Geo<W, H>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * G::LDS_SLAB];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = rfl((uint32_t)threadIdx.x) >> 6;
  const uint32_t n_waves = gridDim.x * kWavesPerBlock;
  const uint32_t B = (uint32_t)p.batch;
  uint32_t env = blockIdx.x * kWavesPerBlock + wave;
  if (env >= B) return;
  uint8_t* slab = lds + wave * G::LDS_SLAB;
  uint32_t bv[G::DPL];
  uint32_t recv;
  load_env<W, H>(p, env, lane, bv, recv);
  for (;;) {
    const uint32_t nenv = env + n_waves;
    const bool last = nenv >= B;
    uint32_t nbv[G::DPL] = {};
    uint32_t nrecv = 0;
    if (!last) load_env<W, H>(p, nenv, lane, nbv, nrecv);  // prefetch the next env of this wave
    process_env<W, H, RULESET>(p, env, lane, slab, bv, recv);
    if (last) break;
#pragma unroll
    for (int k = 0; k < G::DPL; ++k) bv[k] = nbv[k];
    recv = nrecv;
    env = nenv;
    wave_sync();
  }
}

}  // namespace tg
