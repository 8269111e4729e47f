// legacy/tg_kernel_lanes.h — second formulation of the step kernel: 64 (or 32 / 16) envs per wavefront, byte-probing rule
// logic on the uint8 slabs in LDS (17.6 / 13.9 us per 65 536-env step).  Not part of the product library: built only into
// libtetris_gym_hip_legacy.so (`make legacy`) as an independent implementation for the GPU test-suite.
#pragma once
#include "../tg_lane.h"

namespace tg {

// Generator.shuffle of the 7-byte bag for the lanes in `need` (others untouched).
// numpy: for i = 6..1: j = masked-rejection draw in [0, i] from 32-bit outputs (low half of a 64-bit output first,
// the high half is buffered); swap(bag[i], bag[j]).  Every lane runs its OWN position i: one wave-level trip =
// one PCG64 step (the expensive part: 10 quarter-rate multiplies) + two draw attempts (low half, high half), so the
// trip count is the maximum over lanes of ceil(total attempts / 2) (≈ 5-7) instead of the sum over the six draws of
// the per-draw maximum.  (Profiled: the previous per-draw loop was 6 of the 16.7 µs of a 65 536-env step.)
__device__ __forceinline__ void lane_shuffle7(LaneRng& r, uint64_t& bag, const bool need) {
  if (!__any(need)) return;
#ifdef TG_EXPERIMENT_NO_SHUFFLE
  return;  // measurement only: breaks parity
#endif
  lane_rng_ensure(r);
  uint64_t s_lo = r.s_lo, s_hi = r.s_hi;
  uint32_t uinteger = r.uinteger, has32 = r.has32;
  // the bag as seven nibbles of one 32-bit word: every swap is full-rate 32-bit shift/mask work
  uint32_t nb = 0;
#pragma unroll
  for (int k = 0; k < 7; ++k) nb |= (uint32_t)((bag >> (8 * k)) & 0xF) << (4 * k);
  int i = 6;
  bool run = need;
  auto attempt = [&](const uint32_t val, const bool on) {
    const uint32_t mask = (0x7773310u >> (4 * i)) & 7u;  // i >= 4 ? 7 : (i >= 2 ? 3 : 1)
    const uint32_t j = val & mask;
    if (on && j <= (uint32_t)i) {
      const uint32_t bi = (nb >> (4 * i)) & 0xFu, bj = (nb >> (4 * j)) & 0xFu;
      nb &= ~((0xFu << (4 * i)) | (0xFu << (4 * j)));
      nb |= (bj << (4 * i)) | (bi << (4 * j));
      i -= 1;
    }
  };
  // lanes that hold a buffered high half consume it first; afterwards every running lane is in phase
  {
    const bool on = run && has32 != 0;
    attempt(uinteger, on);
    if (on) has32 = 0;
    run = run && i >= 1;
  }
  while (__any(run)) {
    uint64_t n_lo, n_hi, out;
    lane_pcg_step(s_lo, s_hi, r.i_lo, r.i_hi, n_lo, n_hi, out);
    if (run) { s_lo = n_lo; s_hi = n_hi; uinteger = (uint32_t)(out >> 32); has32 = 1; }  // low half taken, high half buffered
    attempt((uint32_t)out, run);
    run = run && i >= 1;
    attempt(uinteger, run);
    if (run) has32 = 0;
    run = run && i >= 1;
  }
  if (need) {
    r.s_lo = s_lo; r.s_hi = s_hi;
    r.uinteger = uinteger; r.has32 = has32;
    r.dirty = true;
    uint64_t b = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) b |= (uint64_t)((nb >> (4 * k)) & 0xFu) << (8 * k);
    bag = b;
  }
}

// BagRandomizer (components/tetromino_randomizer.py:65-85).  The reference reshuffles EAGERLY when the 7th piece
// is drawn; the shuffled bag is first needed by the env's NEXT draw, so inside a step the reshuffle is kept
// pending (bidx == 7) and resolved (a) before a later draw of the same step, (b) together with the reset
// shuffles, or (c) at the end of the step — one PCG64 shuffle invocation per wave-step in the common case
// instead of one per call site.  Per env the generator is advanced in exactly the reference's order.
__device__ __forceinline__ void lane_resolve_shuffle(LaneS& s, LaneRng& r, const bool need) {
  lane_shuffle7(r, s.bag, need);
  if (need) s.bidx = 0;
}

__device__ __forceinline__ int lane_randomizer_next(LaneS& s, LaneRng& r, const bool on) {
  lane_resolve_shuffle(s, r, on && s.bidx >= 7);  // rare: a second draw in the same step after a wrap
  const int t = (int)((s.bag >> (8 * (s.bidx & 7))) & 0xFF);
  if (on) s.bidx += 1;
  return t;
}

// TetrominoQueue.get_next_tetromino (tetromino_queue.py:35-42)
__device__ __forceinline__ int lane_queue_pop(LaneS& s, LaneRng& r, const bool on, const int Q) {
  const int t = (int)(s.queue & 0xFF);
  const int nxt = lane_randomizer_next(s, r, on);
  if (on) s.queue = (s.queue >> 8) | ((uint64_t)nxt << (8 * (Q - 1)));
  return t;
}

// inputs of one group (= the 64/L envs a wave handles at a time), fetched ahead of their use
template <int W, int H, int L>
struct GroupIn {
  static constexpr int NCH = ((64 / L) * W * H + 15) / 16;  // 16-byte chunks of a full group's block
  static constexpr int NIT = (NCH + 63) / 64;               // chunk passes
  uint4 stage[NIT];
  uint4 ra, rb;
  int action;
};

template <int W, int H, int L>
__device__ __forceinline__ void lanes_load(const KParams& p, const uint32_t group, const int lane, GroupIn<W, H, L>& in) {
  constexpr int S = W * H, EPW = 64 / L;
  using GI = GroupIn<W, H, L>;
  const uint32_t B = (uint32_t)p.batch;
  const uint32_t base_env = group * (uint32_t)EPW;
  const uint32_t nlive = B - base_env < (uint32_t)EPW ? B - base_env : (uint32_t)EPW;
  const uint32_t eidx = (uint32_t)lane / L;
  const uint32_t env = base_env + (eidx < nlive ? eidx : nlive - 1u);
  // small per-env loads first (needed first): record, action
  const uint4* rec4 = reinterpret_cast<const uint4*>(p.rec) + (size_t)env * 2;
  in.ra = rec4[0];
  in.rb = rec4[1];
  in.action = p.act_i8 ? (int)reinterpret_cast<const int8_t*>(p.actions)[env]
                       : reinterpret_cast<const int32_t*>(p.actions)[env];
  // the group's slabs: one contiguous block, coalesced 16 B per lane; all loads in flight before the first LDS
  // write (one latency, not NIT of them; clamped + unconditional so the staging array stays in registers)
  const uint32_t nchunks = (nlive * (uint32_t)S + 15u) / 16u;
  const uint4* gsrc = reinterpret_cast<const uint4*>(p.board + (size_t)base_env * S);
#pragma unroll
  for (int k = 0; k < GI::NIT; ++k) {
    const uint32_t c = (uint32_t)lane + 64u * k;
    in.stage[k] = gsrc[c < nchunks ? c : nchunks - 1u];
  }
}

// what one env's transition produced (besides its updated state `s`, generator `r` and LDS slab)
struct LaneOut {
  float reward;
  int lines, terminated;
  bool board_dirty;  // the slab changed (lock / clear / reset)
  bool final_coll;   // NUMPY projection rule: the active piece collides where it stands → not drawn
  bool pending;      // this step was a NEXT_STEP-autoreset step
};

// The rule logic of one transition for the env served by this lane group (L lanes), on its slab in LDS.  Callable
// with any assignment of envs to lane groups (`slab_off` = byte offset of the env's slab inside `slabs`); the rare
// heavy paths are entered by wave-uniform any() branches and the line-clear compaction is cooperative.
template <int W, int H, int RULESET, int L>
__device__ __forceinline__ void lanes_logic(const KParams& p, LaneS& s, LaneRng& r, const LaneBoard<W, H, L>& bd,
                                            uint8_t* slabs, const uint64_t* cell_tab, const int action_in,
                                            const bool live, const int slab_off, const int lane, LaneOut& out) {
  using G = Geo<W, H>;
  constexpr int S = G::CELLS;
  const int sub = lane % L;
  const bool leader = sub == 0;
  const int Q = p.queue_size;
  const int grav = p.gravity;
  float reward = 0.0f;
  int lines = 0;
  bool board_dirty = false;
  bool final_coll = false;

  const bool pending = p.autoreset == TG_AUTORESET_NEXT_STEP && (s.flags & TG_F_PENDING_RESET);
  bool frozen = RULESET == TG_RULESET_FUNCTIONAL && (s.flags & TG_F_GAME_OVER);
  // grouped action set (wrappers/grouped.py:209-269): action = column * 4 + rotations → teleport + hard drop; illegal
  // iff a cell of the rotated piece lies outside the playfield columns (collision_with_frame, :104-122)
  int action = action_in;
  bool g_illegal = false;
  if (RULESET == TG_RULESET_NUMPY && p.action_set == TG_ACTIONS_GROUPED) {
    const int col = action_in >> 2, nrot = (s.rot + (action_in & 3)) & 3;
    const int nx = col + 4 - (s.piece == 0 ? 2 : 1);  // x + padding - matrix.shape[1] // 2
    const uint32_t cg = lane_cells(cell_tab, s.piece, nrot);
    bool legal = (unsigned)action_in < (unsigned)(4 * W);
#pragma unroll
    for (int j = 0; j < 4; ++j) legal = legal && (unsigned)(nx + (int)((cg >> (4 * j)) & 3) - 4) < (unsigned)W;
    if (!pending) {
      if (legal) { s.x = nx; s.rot = nrot; action = 5; }
      else { g_illegal = true; action = 7; if (p.grouped_term) frozen = true; }
    }
  }
  const bool stepped = live && !pending && !frozen && p.stagger != -1;  // (stagger == -1: measurement skeleton, no env steps)

  // ---- C: action → tentative placement (branch-free LUT, per lane) ----
  uint32_t e;
  if (RULESET == TG_RULESET_NUMPY) {
    e = (uint32_t)(kActLutNumpy >> (8 * (action & 7))) & 0xFFu;
  } else {
    const int br = action < 0 ? 0 : (action > 6 ? 6 : action);
    e = (uint32_t)(kActLutFn >> (8 * br)) & 0xFFu;
  }
  int tp = s.piece, tv = s.val;
  int tx = s.x + (int)(e & 3u) - 1;
  int ty = s.y + (int)((e >> 2) & 1u);
  int tr = (s.rot + (int)((e >> 3) & 3u)) & 3;
  const bool check = (e & 0x20u) != 0;
  const bool hard = stepped && (e & 0x40u) != 0;
  // swap (envs/tetris.py:242-252, components/tetromino_holder.py:31-49)
  const bool swap_eff = RULESET == TG_RULESET_NUMPY && stepped && (e & 0x80u) && !(s.flags & TG_F_HAS_SWAPPED);
  if (RULESET == TG_RULESET_NUMPY && __any(swap_eff)) {
    const bool empty = swap_eff && s.hp == TG_NONE;
    const int popped = lane_queue_pop(s, r, empty, Q);  // spawn_tetromino; its collision result is ignored
    if (swap_eff) {
      if (empty) { tp = popped; tr = 0; tv = popped + 2; }
      else { tp = s.hp; tr = s.hr; tv = s.hv; }
      s.hp = s.piece; s.hr = s.rot; s.hv = s.val;
      s.flags |= TG_F_HAS_SWAPPED;
      tx = np_spawn_x<W>(tp);
      ty = 0;
    }
  }
  TG_STAMP(2);
  const uint32_t cellsC = lane_cells(cell_tab, s.piece, s.rot);
  const uint32_t cellsT = lane_cells(cell_tab, tp, tr);

  bool lock = false;
  int drop_reward = 0;
  uint32_t cells = cellsC;
  if (RULESET == TG_RULESET_FUNCTIONAL && __any(hard)) {
    // functional/core.py:230-251 hard_drop: step-wise, stops at the first blocked row below y
    const int yd = bd.drop(cellsC, s.x, s.y, hard);
    if (hard) {
      drop_reward = 2 * (yd - s.y);
      s.y = yd;
      lock = grav || action == 6;  // blocked below by construction → should_lock iff gravity (tetris_fn.py:247-264)
    }
  }
  {
    bool cT, cT1, cC, cC1;
    {
      // tentative {here, below} and current {here, below}: all reads in flight before the first use
      const int xa[2] = {tx, tx}, ya[2] = {ty, ty + 1}, xb[2] = {s.x, s.x}, yb[2] = {s.y, s.y + 1};
      const uint32_t ha = bd.template collide_n<2>(cellsT, xa, ya);
      const uint32_t hb = bd.template collide_n<2>(cellsC, xb, yb);
      cT = ha & 1u; cT1 = (ha >> 1) & 1u; cC = hb & 1u; cC1 = (hb >> 1) & 1u;
    }
    const bool moving = stepped && !(RULESET == TG_RULESET_FUNCTIONAL && hard);
    if (RULESET == TG_RULESET_NUMPY && !stepped) final_coll = cC;  // projection rule for an untouched env
    const bool accept = moving && (swap_eff || (check && !cT));
    if (accept) { s.piece = tp; s.rot = tr; s.x = tx; s.y = ty; s.val = tv; cells = cellsT; }
    const bool coll_here = accept ? cT : cC;
    const bool coll_below = accept ? cT1 : cC1;
    if (RULESET == TG_RULESET_FUNCTIONAL && accept && (e & 4u)) drop_reward = 1;  // soft drop (tetris_fn.py:183-190)
    if (moving) {
      if (RULESET == TG_RULESET_NUMPY) {
        final_coll = coll_here;
        lock = hard;
        if (grav && !hard) {  // envs/tetris.py:259-264
          if (!coll_below) { s.y += 1; final_coll = false; }
          else lock = true;
        }
      } else if (grav) {  // functional/core.py:150-182 + tetris_fn.py:242-247
        if (!coll_below) s.y += 1;
        else lock = true;
      }
    }
    TG_STAMP(3);
    // ---- D: lock path ----
    if (__any(lock)) {
      bool place = lock;
      if (RULESET == TG_RULESET_NUMPY) {  // envs/tetris.py:450-479 commit_active_tetromino
        const bool dead = lock && coll_here;  // already colliding: game over, nothing placed (:461-463)
        if (dead) { s.flags |= TG_F_GAME_OVER; reward = 0.0f; }
        place = lock && !coll_here;
        const int yd = bd.drop(cells, s.x, s.y, place && hard);  // :445-448 (only hard drops can still move)
        if (place) s.y = yd;
        bd.stamp(cells, s.x, s.y, (uint32_t)s.val, false, place);
      } else {
        bd.stamp(cells, s.x, s.y, (uint32_t)(s.piece + 2), true, place);  // functional/core.py:103-121
      }
      wave_sync();
      TG_STAMP(4);
      // full rows can only appear among the rows the piece touched (boards never hold a stale full row:
      // see TG_F_FULLROW in tg_set_state); envs that need it are compacted cooperatively, one at a time
      const bool any_full = bd.any_row_full4(s.y);
      const bool scan = place && (any_full || (s.flags & TG_F_FULLROW));
      uint64_t todo = __ballot(scan && leader);
      while (todo) {
        const int l = (int)__builtin_ctzll(todo) / L;  // env within the wave
        todo &= todo - 1;
        const int n = clear_rows<W, H, RULESET>(slabs + __shfl(slab_off, l * L), lane);
        if (lane / L == l) lines = n;
      }
      if (place) {
        s.flags &= ~TG_F_FULLROW;
        board_dirty = true;
      }
      if (RULESET == TG_RULESET_FUNCTIONAL) {
        // the jnp.take wrap copies old row 0 into the vacated rows: if that row was full, full rows remain
        const bool stale = place && lines > 0 && bd.row_full(0);
        if (stale) s.flags |= TG_F_FULLROW;
      }
      TG_STAMP(5);
      // spawn the next piece
      if (RULESET == TG_RULESET_NUMPY) {
        const int np_ = lane_queue_pop(s, r, place, Q);  // envs/tetris.py:393-401
        if (place) {
          s.piece = np_; s.rot = 0; s.val = np_ + 2; s.x = np_spawn_x<W>(np_); s.y = 0;
        }
        const bool sc = bd.collide(lane_cells(cell_tab, s.piece, 0), s.x, 0);
        if (place) {
          reward = (float)(lines * lines * W + 1);  // score + rewards.alife
          if (sc) { s.flags |= TG_F_GAME_OVER; reward = 0.0f; }
          else s.flags &= ~TG_F_GAME_OVER;
          s.flags &= ~TG_F_HAS_SWAPPED;
          final_coll = sc;
        }
      } else {  // envs/tetris_fn.py:370-413 place_active_tetromino
        lane_rng_ensure(r);
        uint32_t a0, a1, b0, b1;
        threefry2x32(r.k0, r.k1, 0u, 0u, a0, a1);  // split(rng_key)[0]
        const bool refill = place && s.bidx >= 7;
        if (__any(refill)) {
          threefry2x32(r.k0, r.k1, 0u, 1u, b0, b1);  // split(rng_key)[1] (functional/queue.py:59-67)
          const uint64_t nb = jax_permutation7(b0, b1);
          if (refill) { s.bag = nb; s.bidx = 0; }
        }
        if (place) {
          s.piece = (int)((s.bag >> (8 * (s.bidx & 7))) & 0xFF);
          s.bidx += 1;
          r.k0 = a0; r.k1 = a1;
          r.dirty = true;
          s.rot = 0; s.val = s.piece + 2; s.x = (W + 8) / 2 - 2; s.y = 0;
        }
        const bool sc = bd.collide(lane_cells(cell_tab, s.piece, 0), s.x, 0);
        if (place && sc) s.flags |= TG_F_GAME_OVER;
      }
    }
    if (RULESET == TG_RULESET_FUNCTIONAL && stepped) {
      const int lock_reward = lock ? (lines == 4 ? 800 : (lines > 0 ? lines * 200 - 100 : 0)) : 0;  // core.py:124-146
      const float old = s.score;
      float t = old + (float)drop_reward;
      t = t + (float)lock_reward;
      s.score = t;
      reward = t - old;  // tetris_fn.py:312
    }
  }

  TG_STAMP(6);
  // ---- E: termination / autoreset ----
  if (g_illegal) reward = -0.1f;  // rewards.invalid_action
  const int terminated = (!pending && ((s.flags & TG_F_GAME_OVER) || (g_illegal && p.grouped_term))) ? 1 : 0;
  const bool do_reset = live && (pending || (terminated && p.autoreset == TG_AUTORESET_SAME_STEP));
  if (terminated && p.autoreset == TG_AUTORESET_NEXT_STEP && !pending) s.flags |= TG_F_PENDING_RESET;
  TG_STAMP(7);
  if (__any(do_reset)) {
    // (final observation of the terminal state is not produced by this kernel; TG_FLAG_FINAL_OBS selects the other one)
    uint32_t* myw = reinterpret_cast<uint32_t*>(bd.my);
    if (do_reset)
      for (int i = sub; i < S / 4; i += L) myw[i] = 0u;
    if (RULESET == TG_RULESET_NUMPY) {
      lane_resolve_shuffle(s, r, do_reset && s.bidx >= 7);  // rare: this step's spawn wrapped the bag, then game over
      if (do_reset) { s.bag = 0x06050403020100ULL; s.queue = 0; s.bidx = 7; }
      TG_STAMP(8);
      lane_resolve_shuffle(s, r, s.bidx >= 7);  // reset shuffles + pending spawn reshuffles of the other envs, at once
      TG_STAMP(9);
      for (int i = 0; i < Q; ++i) {
        const int t = lane_randomizer_next(s, r, do_reset);
        if (do_reset) s.queue |= (uint64_t)t << (8 * i);
      }
      const int first = lane_queue_pop(s, r, do_reset, Q);
      if (do_reset) { s.piece = first; s.x = np_spawn_x<W>(first); }
    } else {
      lane_rng_ensure(r);
      uint32_t a0, a1, b0, b1;
      threefry2x32(r.k0, r.k1, 0u, 0u, a0, a1);
      threefry2x32(r.k0, r.k1, 0u, 1u, b0, b1);
      const uint64_t nb = jax_permutation7(a0, a1);
      if (do_reset) {
        s.bag = nb; s.queue = 0; s.piece = (int)(nb & 0xFF); s.bidx = 1;
        r.k0 = b0; r.k1 = b1;
        r.dirty = true;
        s.x = (W + 8) / 2 - 2;
      }
    }
    if (do_reset) {
      s.rot = 0; s.y = 0; s.val = s.piece + 2;
      s.hp = TG_NONE; s.hr = 0; s.hv = 0;
      s.flags = 0;
      s.score = 0.0f;
      board_dirty = true;
      final_coll = false;
    }
  }
  wave_sync();
  out.reward = reward;
  out.lines = lines;
  out.terminated = terminated;
  out.board_dirty = board_dirty;
  out.final_coll = final_coll;
  out.pending = pending;
}

// ------------------------------------------------------------------------------------------------
// L = lanes per env (1: 64 envs per wave; 4: 16 envs per wave, the four lanes of an env split the piece's
// four cells and OR their results with DPP quad permutes; everything else is computed redundantly by the quad).
// Measured (profiles/r01_*): with L = 1 there is exactly one wave per SIMD at batch 65 536, so nothing hides
// LDS/VALU dependency latency and the load / compute / store phases of all waves run in lockstep; L = 4 gives
// four waves per SIMD with a quarter of the per-wave work.
template <int W, int H, int RULESET, int L>
__device__ __forceinline__ void lanes_process(const KParams& p, const uint32_t group, const int lane, uint8_t* lds,
                                              const GroupIn<W, H, L>& in) {
  using G = Geo<W, H>;
  constexpr int S = G::CELLS;  // per-env stride in LDS (bytes), same as in HBM: the wave's slabs are one block
  constexpr int EPW = 64 / L;  // envs per wave
  constexpr int NCH = (EPW * S + 15) / 16;  // 16-byte chunks of a full group's block
  constexpr int NIT = (NCH + 63) / 64;      // chunk passes
  uint64_t* cell_tab = reinterpret_cast<uint64_t*>(lds + ((EPW * S + 15) / 16) * 16);  // 7 x 8 bytes behind the slabs
  const int eidx = lane / L, sub = lane % L;
  const uint32_t B = (uint32_t)p.batch;
  const uint32_t base_env = group * (uint32_t)EPW;
  const uint32_t nlive = B - base_env < (uint32_t)EPW ? B - base_env : (uint32_t)EPW;
  const bool live = (uint32_t)eidx < nlive;
  const uint32_t env = base_env + (live ? (uint32_t)eidx : nlive - 1u);  // dead lanes shadow the last live env
  const bool leader = sub == 0;

#ifdef TG_PHASE_TIMING
  if (p.dbg && lane == 0)
    for (int i = 1; i < 16; ++i) p.dbg[(size_t)blockIdx.x * 16 + i] = 0;
#endif
  TG_STAMP(0);
  const uint4 ra = in.ra, rb = in.rb;
  const int action_in = in.action;
  // ---- A: stage the group's slabs in LDS ----
  const uint32_t nbytes = nlive * (uint32_t)S;
  const uint32_t nchunks = (nbytes + 15u) / 16u;
  uint4* lds4 = reinterpret_cast<uint4*>(lds);
  const bool full_wave = nlive == (uint32_t)EPW;
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const uint32_t c = (uint32_t)lane + 64u * k;
    if (c < nchunks) lds4[c] = in.stage[k];
  }
  if (lane < 7) cell_tab[lane] = c_cells.cells[lane];

  LaneS s;
  lane_unpack(s, ra, rb);
  LaneRng r;
  r.s_lo = r.s_hi = r.i_lo = r.i_hi = 0;
  r.uinteger = r.has32 = r.k0 = r.k1 = 0;
  r.src = p.rng + env;
  r.loaded = false;
  r.dirty = false;
  LaneBoard<W, H, L> bd;
  bd.my = lds + eidx * S;
  bd.sub = sub;
  wave_sync();
  TG_STAMP(1);

  LaneOut lo;
  lanes_logic<W, H, RULESET, L>(p, s, r, bd, lds, cell_tab, action_in, live, eidx * S, lane, lo);
  const float reward = lo.reward;
  const int lines = lo.lines, terminated = lo.terminated;
  const bool board_dirty = lo.board_dirty, final_coll = lo.final_coll, pending = lo.pending;

  TG_STAMP(7);
  // ---- F: write back dirty slabs, pack observations, store records ----
  {
    const uint64_t dirty = __ballot(board_dirty && live && leader);  // bit L*e = env e of this wave
    if (dirty) {  // chunk c covers bytes [16c, 16c+16): at most two envs
      uint4* gdst = reinterpret_cast<uint4*>(p.board + (size_t)base_env * S);
      if (full_wave) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int c = lane + 64 * k;
          const uint32_t e0 = ((uint32_t)c * 16u) / (uint32_t)S, e1 = ((uint32_t)c * 16u + 15u) / (uint32_t)S;
          if (c < NCH && (((dirty >> ((e0 * L) & 63u)) | (dirty >> ((e1 < (uint32_t)EPW ? e1 : (uint32_t)EPW - 1u) * L))) & 1ULL))
            gdst[c] = lds4[c];
        }
      } else {
        for (uint32_t c = lane; c < nchunks; c += 64) {
          const uint32_t e0 = (c * 16u) / (uint32_t)S, e1 = (c * 16u + 15u) / (uint32_t)S;
          if (((dirty >> ((e0 * L) & 63u)) | (dirty >> ((e1 < (uint32_t)EPW ? e1 : (uint32_t)EPW - 1u) * L))) & 1ULL)
            gdst[c] = lds4[c];
        }
      }
    }
  }
  if (p.obs_mode == TG_OBS_BOARD || p.obs_mode == TG_OBS_FN) {
    const bool show = live && (RULESET == TG_RULESET_NUMPY ? !final_coll : !(s.flags & TG_F_GAME_OVER));
    bd.stamp(lane_cells(cell_tab, s.piece, s.rot), s.x, s.y, p.obs_mode == TG_OBS_FN ? 0xFFu : (uint32_t)s.val, false,
             show);
    wave_sync();
    uint4* odst = reinterpret_cast<uint4*>(p.obs + (size_t)base_env * S);
    const bool fn = p.obs_mode == TG_OBS_FN;
    auto bin1 = [](uint32_t x) -> uint32_t {  // bytes: 0xFF (active) stays -1, other non-zero → 1  (tetris_fn.py:137-158)
      const uint32_t nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
      const uint32_t ff = ((x & 0x7F7F7F7Fu) + 0x01010101u) & x & 0x80808080u;
      return (nz >> 7) | ((ff >> 7) * 0xFFu);
    };
    auto binarize = [&](uint4 v) -> uint4 { return make_uint4(bin1(v.x), bin1(v.y), bin1(v.z), bin1(v.w)); };
    if (full_wave) {
      uint4 stage[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int c = lane + 64 * k;
        stage[k] = lds4[c < NCH ? c : NCH - 1];
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int c = lane + 64 * k;
        if (c < NCH) {
          odst[c] = fn ? binarize(stage[k]) : stage[k];
        }
      }
    } else {
      for (uint32_t c = lane; c < nchunks; c += 64) {
        uint4 v = lds4[c];
        if (fn) v = binarize(v);
        if (c * 16u + 16u <= nbytes) {
          odst[c] = v;
        } else {  // ragged tail of a partial wave
          const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
          for (uint32_t i = 0; c * 16u + i < nbytes; ++i)
            p.obs[(size_t)base_env * S + c * 16u + i] = (uint8_t)(w4[i >> 2] >> (8 * (i & 3)));
        }
      }
    }
  }
  if (RULESET == TG_RULESET_NUMPY) lane_resolve_shuffle(s, r, s.bidx >= 7);  // pending reshuffle (no reset in this wave)
  TG_STAMP(8);
  if (live && leader) {
    uint4 oa, ob;
    lane_pack(s, oa, ob);
    uint4* rec_out = reinterpret_cast<uint4*>(p.rec) + (size_t)env * 2;
    rec_out[0] = oa;
    rec_out[1] = ob;
    p.out[env] = make_uint2(__float_as_uint(reward), (uint32_t)terminated | ((uint32_t)lines << 8));
    if (r.dirty) {
      uint4* rng_out = reinterpret_cast<uint4*>(p.rng + env);
      rng_out[0] = make_uint4((uint32_t)r.s_lo, (uint32_t)(r.s_lo >> 32), (uint32_t)r.s_hi, (uint32_t)(r.s_hi >> 32));
      rng_out[2] = make_uint4(r.uinteger, r.has32, r.k0, r.k1);
    }
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
  TG_STAMP(9);
}

// The kernel.  PIPE = false: one group per wave.  PIPE = true: two groups per wave (g and g + gridDim.x); the
// second group's inputs are requested before the first group is processed, so that half of the loads and half
// of the stores overlap compute instead of every wave loading, computing and storing in lockstep.
template <int W, int H, int RULESET, int L, bool PIPE>
__global__ __launch_bounds__(64, 4) void step_kernel_lanes(const KParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int EPW = 64 / L;
  const int lane = threadIdx.x;
  const uint32_t n_groups = ((uint32_t)p.batch + EPW - 1) / EPW;
  const uint32_t g0 = blockIdx.x;
  if (g0 >= n_groups) return;
  if (p.stagger > 0 && (blockIdx.x & 1)) {
    // All waves of a launch would otherwise load, compute and store in lockstep (three serial bursts).  Starting
    // every other wave late lets its loads run under the early waves' compute and its compute under their stores.
    for (int i = 0; i < p.stagger; i += 64) __builtin_amdgcn_s_sleep(64);
  }
  GroupIn<W, H, L> a;
  lanes_load<W, H, L>(p, g0, lane, a);
  if constexpr (PIPE) {
    const uint32_t g1 = g0 + gridDim.x;
    const bool two = g1 < n_groups;
    GroupIn<W, H, L> b;
    lanes_load<W, H, L>(p, two ? g1 : g0, lane, b);  // (a harmless re-read when this wave has a single group)
    lanes_process<W, H, RULESET, L>(p, g0, lane, lds, a);
    if (two) {
      wave_sync();
      lanes_process<W, H, RULESET, L>(p, g1, lane, lds, b);
    }
  } else {
    lanes_process<W, H, RULESET, L>(p, g0, lane, lds, a);
  }
}

}  // namespace tg
