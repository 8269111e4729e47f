// tg_common.h — device code shared by every kernel of the MI355X Tetris step engine (gfx950, wave64): launch parameter
// blocks, the piece tables, numpy PCG64 / jax threefry primitives, the wave-level slab routines (collision ballots, row
// compaction by ballot/popcount, observation emit) and the kernels that are not the hot step: reset_kernel,
// observe_kernel (DICT / RGB re-pack from state), grouped_kernel (legal mask + device-side sampling), the stream-copy
// ceiling kernel.  The step kernel itself is tg_kernel_bits.h; its earlier formulations live in legacy/ (test-only build).
// Integer / byte work only: no MFMA.  Bound: HBM (see DESIGN.md for bytes per env-step).
#pragma once
#ifndef __HIPCC_RTC__  // (hiprtc, which compiles board sizes that are not built in at tg_create, brings its own runtime declarations)
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
using __hip_internal::int8_t; using __hip_internal::uint8_t; using __hip_internal::int16_t; using __hip_internal::uint16_t;
using __hip_internal::int32_t; using __hip_internal::uint32_t; using __hip_internal::int64_t; using __hip_internal::uint64_t;
#endif

#include "../../include/tetris_gym_hip.h"
#include "tg_tables.h"

namespace tg {

__constant__ ShapeTable c_shapes = make_shape_table();

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;

struct KParams {
  uint8_t* board;           // [B][H][W]
  uint32_t* rec;            // [B][8]
  tg_rng_rec* rng;          // [B]
  const void* actions;      // i32[B] or i8[B] (act_i8)
  uint8_t* obs;             // per obs_mode
  uint8_t* final_obs;       // optional (SAME_STEP autoreset)
  uint2* out;               // [B] tg_step_out {f32 reward; u8 terminated; u8 lines; u16 0}
  tg_episode_rec* episode;  // optional
  uint64_t shapes[7];       // per piece: 4 rotations x 16-bit cell masks (tg_tables.h)
  int batch;
  int gravity;
  int autoreset;
  int obs_mode;
  int queue_size;
  int act_i8;
  int action_set;    // TG_ACTIONS_*
  int grouped_term;  // wrappers/grouped.py:49 terminate_on_illegal_action
  int stagger;       // lane kernels: odd waves start this many x64 cycles late (de-phases load/compute/store bursts)
  unsigned long long* dbg;  // optional phase-timing buffer (TG_PHASE_TIMING builds only)
  void* cols;               // bits kernel only: column masks [B][W] (tg_kernel_bits.h); null otherwise
  uint4* ahead;             // bits kernel only: generator states of the permutation ring [B][2][2]
  uint64_t* ring;           // bits kernel only: permutation rings [2][B]
  uint32_t* hbuf;           // bits kernel only: consumed counters + state location [2][B]
  uint4* inl;               // bits kernel only: generator states after in-line shuffles [2][B][2]
  int ring_cur;             // bits kernel only: ring parity (flips with every launch that carries the RNG role)
  int hbuf_cur;             // bits kernel only: consumed-counter parity (flips with every launch)
  int rng_role;             // bits kernel only: this launch carries the RNG-role workgroups
  uint32_t ring_stride;     // bits kernel only: elements between the parities of ring / hbuf
  uint32_t cols_bytes;      // bits kernel only: bytes per env of `cols` (host-side shard offsets)
  int nsteps;               // bits kernel only: transitions per launch (tg_step_k), int8 action rows act_stride apart
  uint32_t act_stride;
  uint8_t* obs_traj;        // optional (tg_rollout): observation of transition k of this launch at obs_traj + k * obs_traj_stride
  uint2* out_traj;          // optional: outputs of transition k at out_traj + k * act_stride
  size_t obs_traj_stride;   // bytes between transitions (the handle's batch * observation bytes per env)
  // bits kernel only, grouped action set: draw the action of every transition on the device, uniformly from the env's
  // legal placements (the rule of grouped_kernel / tg_grouped_sample: same hash, same order) instead of reading it
  int sample_mode;
  uint32_t sample_seed, sample_counter;  // transition k of the launch uses counter + k
  uint32_t env_index_base;  // handle-local index of this launch's env 0 (stream shards)
  int8_t* act_traj;         // optional: the drawn actions, rows act_stride apart
  int ring_fill;            // bits kernel only: ring entries the RNG role maintains (TG_RING_FILL, default all)
  int dict_in_kernel;       // bits kernel only: TG_OBS_DICT is packed by the step kernel's two waves (sizes with (H+4)(W+8) % 16 == 0)
  uint8_t* pboard;          // bits kernel, packed handles (kFeatPacked): the board as nibbles [B][H*W/2] — what the step kernel streams
                            // instead of `board`; null otherwise
  int fn_lazy;              // bits kernel, FUNCTIONAL (TG_FN_LAZY, tests): bits 0-1: 0 = a lock sorts the reset bag ahead only when the stack is
                            // within reach of the spawn rows, 1 = never (every episode end takes the in-line sort), 2 = always (round 3);
                            // bit 2: the look-ahead never catches up behind a reset (the chain stays one key deep)
  uint32_t* nib_over;       // bits kernel, packed FUNCTIONAL handles: host-visible flag word, set when a lock's `board += id` leaves the
                            // nibble range (only a piece uploaded on top of filled cells does: functional/core.py:103-121 adds)
  // bits kernel, NUMPY: the handle's ActionsMapping as the per-action table (kActLutNumpy for the stock mapping; tg_engine.hip
  // act_lut_from_ids) and its RewardsMapping (mappings/rewards.py:12-15; stock 1, 0, -0.1) — data, not constants, since ABI 4
  uint64_t act_lut;
  float r_alife, r_game_over, r_invalid;
};

struct ResetParams {
  uint8_t* board;
  uint32_t* rec;
  tg_rng_rec* rng;
  const tg_rng_rec* new_rng;  // [B] or null: staged generator states / keys
  const uint8_t* reseed;      // [B] or null: 1 = take new_rng[i]
  const uint8_t* mask;        // [B] or null (all)
  uint8_t* obs;
  tg_episode_rec* episode;
  int batch;
  int obs_mode;
  int queue_size;
};

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS traffic between lanes of ONE wave: DS ops of a wave execute in order, so only the compiler has
// to be kept from reordering them.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int W, int H>
struct Geo {
  static constexpr int CELLS = W * H;
  static constexpr int NDW = (CELLS + 3) / 4;             // dwords of an env's slab (the last one partial when CELLS % 4 != 0)
  static constexpr int DPL = (NDW + 63) / 64;             // dwords per lane
  static constexpr int SLAB = ((CELLS + 15) / 16) * 16;   // slab bytes
  static constexpr int LDS_SLAB = 256 * DPL;              // LDS bytes per wave (whole dword-per-lane passes)
  static constexpr int WP = W + 8;
  static constexpr int HP = H + 4;
  static_assert(H + 4 <= 64, "one row per lane: H + padding must fit a wave");
  static_assert(W <= 32, "row masks are 32-bit");
};

// Dword idx of an env's uint8 slab in HBM.  With W * H not a multiple of 4 the slabs start at 2- or 1-byte alignment (global
// memory takes unaligned dword accesses) and the last dword is partial: a load reads into the next env's slab (the buffers
// carry tail padding; the bytes land behind the slab in LDS, where nobody looks), a store writes the tail byte by byte.
typedef uint32_t u32_a2 __attribute__((aligned(2)));
typedef uint32_t u32_a1 __attribute__((aligned(1)));
template <int CELLS>
__device__ __forceinline__ uint32_t slab_ld(const uint8_t* base, const int idx) {
  if constexpr (CELLS % 4 == 0) return reinterpret_cast<const uint32_t*>(base)[idx];
  else if constexpr (CELLS % 2 == 0) return reinterpret_cast<const u32_a2*>(base)[idx];
  else return reinterpret_cast<const u32_a1*>(base)[idx];
}
template <int CELLS>
__device__ __forceinline__ void slab_st(uint8_t* base, const int idx, const uint32_t v) {
  if constexpr (CELLS % 4 == 0) {
    reinterpret_cast<uint32_t*>(base)[idx] = v;
  } else {
    if (4 * idx + 4 <= CELLS) {
      if constexpr (CELLS % 2 == 0) reinterpret_cast<u32_a2*>(base)[idx] = v;
      else reinterpret_cast<u32_a1*>(base)[idx] = v;
    } else {
      for (int b = 0; 4 * idx + b < CELLS; ++b) base[4 * idx + b] = (uint8_t)(v >> (8 * b));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// numpy PCG64 (restated from numpy/random/src/pcg64/pcg64.h; pinned by tests through the oracle)
// ------------------------------------------------------------------------------------------------
struct Rng {
  uint64_t s_lo, s_hi, i_lo, i_hi;
  uint32_t uinteger, has32;
  uint32_t k0, k1;
  bool loaded, dirty;
};

__device__ __forceinline__ void rng_load(Rng& r, const tg_rng_rec* p) {
  if (r.loaded) return;
  r.s_lo = p->state_lo; r.s_hi = p->state_hi;
  r.i_lo = p->inc_lo;   r.i_hi = p->inc_hi;
  r.uinteger = p->uinteger; r.has32 = p->has_uint32;
  r.k0 = p->key[0]; r.k1 = p->key[1];
  r.loaded = true;
}

__device__ __forceinline__ void rng_store(const Rng& r, tg_rng_rec* p, int lane) {
  if (!r.dirty || lane != 0) return;
  p->state_lo = r.s_lo; p->state_hi = r.s_hi;
  p->inc_lo = r.i_lo;   p->inc_hi = r.i_hi;
  p->uinteger = r.uinteger; p->has_uint32 = r.has32;
  p->key[0] = r.k0; p->key[1] = r.k1;
}

__device__ __forceinline__ uint32_t pcg_next32(Rng& r) {
  r.dirty = true;
  if (r.has32) {
    r.has32 = 0;
    return r.uinteger;
  }
  constexpr uint64_t MLO = 0x4385DF649FCCF645ULL, MHI = 0x2360ED051FC65DA4ULL;
  // state = state * mult + inc  (128-bit, from 64-bit pieces)
  uint64_t lo = r.s_lo * MLO;
  uint64_t hi = __umul64hi(r.s_lo, MLO) + r.s_lo * MHI + r.s_hi * MLO;
  uint64_t nlo = lo + r.i_lo;
  hi = hi + r.i_hi + (nlo < lo ? 1ULL : 0ULL);
  r.s_lo = nlo;
  r.s_hi = hi;
  uint64_t x = hi ^ nlo;           // XSL-RR of the NEW state
  unsigned rot = (unsigned)(hi >> 58);
  uint64_t out = (x >> rot) | (x << ((64u - rot) & 63u));
  r.has32 = 1;
  r.uinteger = (uint32_t)(out >> 32);
  return (uint32_t)out;
}

// Generator.shuffle on the 7-entry bag (bytes of a 56-bit word): Fisher-Yates from the top with numpy's
// masked-rejection bounded draw (random_interval, 32-bit path).
__device__ __forceinline__ uint64_t pcg_shuffle7(Rng& r, uint64_t bag) {
  for (int i = 6; i >= 1; --i) {
    const uint32_t mask = i >= 4 ? 7u : (i >= 2 ? 3u : 1u);
    uint32_t j;
    do {
      j = pcg_next32(r) & mask;
    } while (j > (uint32_t)i);
    const uint64_t bi = (bag >> (8 * i)) & 0xFF, bj = (bag >> (8 * j)) & 0xFF;
    bag &= ~((0xFFULL << (8 * i)) | (0xFFULL << (8 * j)));
    bag |= (bj << (8 * i)) | (bi << (8 * j));
  }
  return bag;
}

// ------------------------------------------------------------------------------------------------
// jax threefry2x32 / split / permutation(7)   (restated from jax 0.5.3 _src/prng.py, partitionable
// scheme = the jax >= 0.5.0 default; PARITY UNPINNED)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int d) { return (x << d) | (x >> (32 - d)); }

__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0,
                                             uint32_t& o1) {
  const uint32_t ks0 = k0, ks1 = k1, ks2 = k0 ^ k1 ^ 0x1BD11BDAu;
  uint32_t x0 = c0 + ks0, x1 = c1 + ks1;
#define TG_MIX(r) x0 += x1; x1 = rotl32(x1, r); x1 ^= x0;
  TG_MIX(13) TG_MIX(15) TG_MIX(26) TG_MIX(6)   x0 += ks1; x1 += ks2 + 1u;
  TG_MIX(17) TG_MIX(29) TG_MIX(16) TG_MIX(24)  x0 += ks2; x1 += ks0 + 2u;
  TG_MIX(13) TG_MIX(15) TG_MIX(26) TG_MIX(6)   x0 += ks0; x1 += ks1 + 3u;
  TG_MIX(17) TG_MIX(29) TG_MIX(16) TG_MIX(24)  x0 += ks1; x1 += ks2 + 4u;
  TG_MIX(13) TG_MIX(15) TG_MIX(26) TG_MIX(6)   x0 += ks2; x1 += ks0 + 5u;
#undef TG_MIX
  o0 = x0;
  o1 = x1;
}

// permutation(key, arange(n)), n = EnvConfig.queue_size in 1..7 (functional/queue.py:35): key', sub = split(key); bits_i = hi^lo of
// threefry(sub, (0,i)); stable sort (jax _shuffle: one round for 2 <= n <= 7, none for n = 1 — a single element sorts to itself here too).
// Entries n..6 of the result are 0.
__device__ __forceinline__ uint64_t jax_permutation(uint32_t k0, uint32_t k1, const int n) {
  uint32_t s0, s1;
  threefry2x32(k0, k1, 0u, 1u, s0, s1);  // split(key)[1]
  uint32_t bits[7];
#pragma unroll
  for (uint32_t i = 0; i < 7; ++i) {
    uint32_t a, b;
    threefry2x32(s0, s1, 0u, i, a, b);
    bits[i] = a ^ b;
  }
  // rank of element i = #{j < n : bits[j] < bits[i] or (== and j < i)}  (stable)
  uint64_t bag = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) rank += (j < n && (bits[j] < bits[i] || (bits[j] == bits[i] && j < i))) ? 1 : 0;
    bag |= i < n ? (uint64_t)i << (8 * rank) : 0ull;
  }
  return bag;
}
__device__ __forceinline__ uint64_t jax_permutation7(uint32_t k0, uint32_t k1) { return jax_permutation(k0, k1, 7); }

// ------------------------------------------------------------------------------------------------
// wave-level board primitives (slab = this wave's LDS copy of the playfield, H*W bytes)
// ------------------------------------------------------------------------------------------------
// 4 candidate placements tested at once.  Lane group q = lane>>4 tests candidate q, lane&15 = matrix
// cell.  q0 = (shapeT, xT, yT)   q1 = (shapeT, xT, yT+1)   q2 = (shapeC, xC, yC)   q3 = (shapeC, xC, yC+1)
// → low dword of the ballot = tentative {here | below << 16}, high dword = current {here | below << 16}.
// Cells outside the playfield are bedrock (envs/tetris.py:632-641 pads with 1) → hit.
template <int W, int H>
__device__ __forceinline__ uint64_t test4(const uint8_t* slab, int lane, uint32_t shapeT, int xT, int yT,
                                          uint32_t shapeC, int xC, int yC) {
  const bool useT = lane < 32;
  const uint32_t shape = useT ? shapeT : shapeC;
  const int x = useT ? xT : xC;
  const int y = (useT ? yT : yC) + ((lane >> 4) & 1);
  const int cell = lane & 15;
  const int row = y + (cell >> 2), col = x + (cell & 3) - 4;
  const bool filled = (shape >> cell) & 1u;
  const bool inb = (unsigned)col < (unsigned)W && (unsigned)row < (unsigned)H;
  const uint8_t occ = slab[inb ? row * W + col : 0];
  return __ballot(filled && (!inb || occ != 0));
}

template <int W, int H>
__device__ __forceinline__ bool test1(const uint8_t* slab, int lane, uint32_t shape, int x, int y) {
  const int cell = lane & 15;
  const int row = y + (cell >> 2), col = x + (cell & 3) - 4;
  const bool filled = lane < 16 && ((shape >> cell) & 1u);
  const bool inb = (unsigned)col < (unsigned)W && (unsigned)row < (unsigned)H;
  const uint8_t occ = slab[inb ? row * W + col : 0];
  return __ballot(filled && (!inb || occ != 0)) != 0;
}

// bit r of the result = "shape at (x, y' = r) collides".  Rows >= H are bedrock, so high bits are set.
template <int W, int H>
__device__ __forceinline__ uint64_t column_scan(const uint8_t* slab, int lane, uint32_t shape, int x) {
  bool hit = false;
  uint32_t s = shape;
  while (s) {
    const int cell = __builtin_ctz(s);
    s &= s - 1;
    const int row = lane + (cell >> 2), col = x + (cell & 3) - 4;
    const bool inb = (unsigned)col < (unsigned)W && row < H;
    hit |= !inb || slab[inb ? row * W + col : 0] != 0;
  }
  return __ballot(hit);
}

// The reference's drop loop `while not collision(y + 1): y += 1` (envs/tetris.py:445-448,
// functional/core.py:243-249) stops at the FIRST blocked row below y: smallest y' >= y with C[y'+1] set.
__device__ __forceinline__ int drop_row(uint64_t C, int y) {
  const int sh = y + 1 < 63 ? y + 1 : 63;
  return y + (int)__builtin_ctzll((C | (1ULL << 63)) >> sh);
}

// write the piece's cells into the slab (lanes 0..15, one cell each)
template <int W, int H>
__device__ __forceinline__ void stamp(uint8_t* slab, int lane, uint32_t shape, int x, int y, uint32_t val,
                                      bool add) {
  const int cell = lane & 15;
  const int row = y + (cell >> 2), col = x + (cell & 3) - 4;
  const bool filled = lane < 16 && ((shape >> cell) & 1u);
  const bool inb = (unsigned)col < (unsigned)W && (unsigned)row < (unsigned)H;
  if (filled && inb) {
    const int a = row * W + col;
    slab[a] = (uint8_t)(add ? slab[a] + val : val);
  }
}

// Full-row detect + compaction.  Returns the number of cleared rows.
//   NUMPY (envs/tetris.py:481-512): vacated top rows are empty.
//   FUNCTIONAL (functional/core.py:185-227): cleared rows get gather index -H which jnp.take wraps to
//   row 0 of the pre-compaction board → vacated rows are copies of old row 0.
template <int W, int H, int RULESET>
__device__ __forceinline__ int clear_rows(uint8_t* slab, int lane) {
  uint8_t rowb[W];
  const int r = lane < H ? lane : H - 1;
  bool full = lane < H;
  bool ones = RULESET == TG_RULESET_NUMPY;  // NUMPY: a row of bedrock cells only is never taken (envs/tetris.py:493)
#pragma unroll
  for (int c = 0; c < W; ++c) {
    rowb[c] = slab[r * W + c];
    full &= rowb[c] != 0;
    ones &= rowb[c] == 1;
  }
  full &= !ones;
  const uint64_t F = __ballot(full);
  const int n = __popcll(F);
  if (n > 0) {
    const uint64_t below_mask = lane < 63 ? (F >> (lane + 1)) : 0ULL;
    const int dest = lane + __popcll(below_mask);
    const bool keep = lane < H && !full;
    wave_sync();
    if (keep) {
#pragma unroll
      for (int c = 0; c < W; ++c) slab[dest * W + c] = rowb[c];
    }
    uint8_t top[W];
#pragma unroll
    for (int c = 0; c < W; ++c)
      top[c] = RULESET == TG_RULESET_FUNCTIONAL ? (uint8_t)__shfl((int)rowb[c], 0) : (uint8_t)0;
    if (lane < n) {
#pragma unroll
      for (int c = 0; c < W; ++c) slab[lane * W + c] = top[c];
    }
    wave_sync();
  }
  return n;
}

// NUMPY spawn column: width_padded // 2 - matrix.shape[0] // 2  (envs/tetris.py:536-541)
template <int W>
__device__ __forceinline__ int np_spawn_x(int piece) {
  return (W + 8) / 2 - (piece == 0 ? 2 : 1);
}

// ------------------------------------------------------------------------------------------------
// per-env uniform state (unpacked tg_env_rec)
// ------------------------------------------------------------------------------------------------
struct EnvS {
  int piece, rot, x, y, val, flags, hp, hr, hv, bidx;
  uint64_t bag, queue;  // 7 bytes each
  float score;
  uint32_t reserved;
};

__device__ __forceinline__ void rec_unpack(EnvS& s, const uint32_t* p) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  const uint4 b = *reinterpret_cast<const uint4*>(p + 4);
  const uint32_t d0 = rfl(a.x), d1 = rfl(a.y), d2 = rfl(a.z), d3 = rfl(a.w);
  const uint32_t d4 = rfl(b.x), d5 = rfl(b.y), d6 = rfl(b.z), d7 = rfl(b.w);
  s.piece = d0 & 0xFF; s.rot = (d0 >> 8) & 0xFF; s.x = (d0 >> 16) & 0xFF; s.y = d0 >> 24;
  s.val = d1 & 0xFF; s.flags = (d1 >> 8) & 0xFF; s.hp = (d1 >> 16) & 0xFF; s.hr = d1 >> 24;
  s.hv = d2 & 0xFF; s.bidx = (d2 >> 8) & 0xFF;
  s.bag = (uint64_t)(d2 >> 16) | ((uint64_t)d3 << 16) | ((uint64_t)(d4 & 0xFF) << 48);
  s.queue = (uint64_t)(d4 >> 8) | ((uint64_t)d5 << 24);
  s.score = __uint_as_float(d6);
  s.reserved = d7;
}

__device__ __forceinline__ void rec_pack(const EnvS& s, uint32_t* p, int lane) {
  uint4 a, b;
  a.x = (uint32_t)s.piece | ((uint32_t)s.rot << 8) | ((uint32_t)s.x << 16) | ((uint32_t)s.y << 24);
  a.y = (uint32_t)s.val | ((uint32_t)s.flags << 8) | ((uint32_t)s.hp << 16) | ((uint32_t)s.hr << 24);
  a.z = (uint32_t)s.hv | ((uint32_t)s.bidx << 8) | ((uint32_t)(s.bag & 0xFFFF) << 16);
  a.w = (uint32_t)(s.bag >> 16);
  b.x = (uint32_t)((s.bag >> 48) & 0xFF) | ((uint32_t)(s.queue & 0xFFFFFF) << 8);
  b.y = (uint32_t)(s.queue >> 24);
  b.z = __float_as_uint(s.score);
  b.w = s.reserved;
  if (lane == 0) {
    *reinterpret_cast<uint4*>(p) = a;
    *reinterpret_cast<uint4*>(p + 4) = b;
  }
}

// BagRandomizer.get_next_tetromino (components/tetromino_randomizer.py:65-85): eager reshuffle
__device__ __forceinline__ int np_randomizer_next(EnvS& s, Rng& r, const tg_rng_rec* rp) {
  const int t = (int)((s.bag >> (8 * s.bidx)) & 0xFF);
  s.bidx += 1;
  if (s.bidx >= 7) {
    rng_load(r, rp);
    s.bag = pcg_shuffle7(r, s.bag);
    s.bidx = 0;
  }
  return t;
}

// TetrominoQueue.get_next_tetromino (components/tetromino_queue.py:35-42)
__device__ __forceinline__ int np_queue_pop(EnvS& s, Rng& r, const tg_rng_rec* rp, int Q) {
  const int t = (int)(s.queue & 0xFF);
  const int nxt = np_randomizer_next(s, r, rp);
  s.queue = (s.queue >> 8) | ((uint64_t)nxt << (8 * (Q - 1)));
  return t;
}

// ------------------------------------------------------------------------------------------------
// reset of one env (uniform).  The generator state / key to use is already in `r` (loaded from rp).
//   NUMPY: envs/tetris.py:274-307 + components/tetromino_queue.py:23-33 + tetromino_randomizer.py:87-91
//   FUNCTIONAL: envs/tetris_fn.py:318-367
// ------------------------------------------------------------------------------------------------
template <int W, int H, int RULESET>
__device__ __forceinline__ void env_reset(EnvS& s, Rng& r, const tg_rng_rec* rp, uint32_t* slab_dw, int lane,
                                          int Q) {
  using G = Geo<W, H>;
#pragma unroll
  for (int k = 0; k < G::DPL; ++k) {
    const int idx = lane + 64 * k;
    if (idx < G::NDW) slab_dw[idx] = 0u;
  }
  rng_load(r, rp);
  s.rot = 0; s.y = 0;
  s.hp = TG_NONE; s.hr = 0; s.hv = 0;
  s.flags = 0;
  s.score = 0.0f;
  if (RULESET == TG_RULESET_NUMPY) {
    s.bag = pcg_shuffle7(r, 0x06050403020100ULL);
    s.bidx = 0;
    s.queue = 0;
    for (int i = 0; i < Q; ++i) {
      const int t = np_randomizer_next(s, r, rp);
      s.queue |= (uint64_t)t << (8 * i);
    }
    s.piece = np_queue_pop(s, r, rp, Q);
    s.x = np_spawn_x<W>(s.piece);
  } else {
    uint32_t a0, a1, b0, b1;
    threefry2x32(r.k0, r.k1, 0u, 0u, a0, a1);  // key   = split(key_in)[0]
    threefry2x32(r.k0, r.k1, 0u, 1u, b0, b1);  // subkey = split(key_in)[1] → State.rng_key
    s.bag = jax_permutation(a0, a1, Q);  // (FUNCTIONAL: Q = EnvConfig.queue_size, the bag length)
    s.queue = 0;
    s.piece = (int)(s.bag & 0xFF);
    s.bidx = 1;
    r.k0 = b0; r.k1 = b1;
    r.dirty = true;
    s.x = (W + 8) / 2 - 2;  // functional/core.py:79-83
  }
  s.val = s.piece + 2;
  wave_sync();
}

// observation surfaces written from the LDS slab (TG_OBS_BOARD / TG_OBS_FN); dirties the slab
template <int W, int H>
__device__ __forceinline__ void emit_obs(uint8_t* slab, int lane, uint8_t* dst_env, int mode, bool show,
                                         uint32_t shape, int x, int y, int val) {
  using G = Geo<W, H>;
  if (mode != TG_OBS_BOARD && mode != TG_OBS_FN) return;
  // the piece is only ever drawn onto free cells (NUMPY: hidden when it collides, envs/tetris.py:558-560;
  // FUNCTIONAL: hidden when game over), so a plain byte write equals the reference's `+=`
  if (show && mode == TG_OBS_FN) {
    // (board > 0) + (-1) * piece (envs/tetris_fn.py:148-155): -1 on a free cell, 1 - 1 = 0 on a filled one (a piece uploaded on top of
    // filled cells)
    const int cell = lane & 15;
    const int row = y + (cell >> 2), col = x + (cell & 3) - 4;
    if (lane < 16 && ((shape >> cell) & 1u) && (unsigned)col < (unsigned)W && (unsigned)row < (unsigned)H)
      slab[row * W + col] = slab[row * W + col] ? (uint8_t)0 : (uint8_t)0xFF;
  } else if (show) {
    stamp<W, H>(slab, lane, shape, x, y, (uint32_t)val, false);
  }
  wave_sync();
  const uint32_t* slab_dw = reinterpret_cast<const uint32_t*>(slab);
#pragma unroll
  for (int k = 0; k < G::DPL; ++k) {
    const int idx = lane + 64 * k;
    if (idx < G::NDW) {
      uint32_t v = slab_dw[idx];
      if (mode == TG_OBS_FN) {  // bytes: 0xFF (active) stays -1, other non-zero → 1  (tetris_fn.py:137-158)
        const uint32_t nz = (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
        const uint32_t ff = ((v & 0x7F7F7F7Fu) + 0x01010101u) & v & 0x80808080u;
        v = (nz >> 7) | ((ff >> 7) * 0xFFu);
      }
      slab_st<G::CELLS>(dst_env, idx, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// THE step kernel: one env per wavefront
//
// Measured on MI355X (profiles/r01_*): with one env per wave the rule logic is scalar and the kernel is
// bound by the CU's single scalar ALU (1 SALU instruction / cycle / CU), not by HBM.  The hot path
// (no lock: ~82 % of random-policy steps) is therefore written to minimise scalar instructions:
//   * the 32-byte record arrives with one s_load_dwordx8 and only d0/d1 are unpacked;
//   * the action switch is a shift of a 64-bit LUT constant (branch-free: dx, dy, drot, flags);
//   * both shapes come from one 8-byte row of the shape table carried in the kernel arguments;
//   * bag / queue / holder / generator state are unpacked only inside the rare lock, swap and reset
//     paths, and only the first 8 record bytes are stored back when nothing else changed;
//   * reward / terminated / lines leave as ONE 8-byte record.
// ------------------------------------------------------------------------------------------------
// per-action LUT byte: [1:0] dx+1, [2] dy, [4:3] drot, [5] needs collision check, [6] hard drop, [7] swap
constexpr uint64_t kActLutNumpy =  // mappings/actions.py:12-19
    0x20ULL | (0x22ULL << 8) | (0x25ULL << 16) | (0x29ULL << 24) | (0x39ULL << 32) | (0x41ULL << 40) |
    (0x81ULL << 48) | (0x01ULL << 56);
constexpr uint64_t kActLutFn =  // envs/tetris_fn.py:470-478 (3 = rotation-1, 4 = rotation+1, 5 = no-op, 6 = hard drop)
    0x20ULL | (0x22ULL << 8) | (0x25ULL << 16) | (0x39ULL << 24) | (0x29ULL << 32) | (0x01ULL << 40) |
    (0x41ULL << 48) | (0x41ULL << 56);

__device__ __forceinline__ void raw_to_env(EnvS& s, const uint32_t d[8]) {
  s.piece = d[0] & 0xFF; s.rot = (d[0] >> 8) & 0xFF; s.x = (d[0] >> 16) & 0xFF; s.y = d[0] >> 24;
  s.val = d[1] & 0xFF; s.flags = (d[1] >> 8) & 0xFF; s.hp = (d[1] >> 16) & 0xFF; s.hr = d[1] >> 24;
  s.hv = d[2] & 0xFF; s.bidx = (d[2] >> 8) & 0xFF;
  s.bag = (uint64_t)(d[2] >> 16) | ((uint64_t)d[3] << 16) | ((uint64_t)(d[4] & 0xFF) << 48);
  s.queue = (uint64_t)(d[4] >> 8) | ((uint64_t)d[5] << 24);
  s.score = __uint_as_float(d[6]);
  s.reserved = d[7];
}

__device__ __forceinline__ void env_to_raw(const EnvS& s, uint32_t d[8]) {
  d[0] = (uint32_t)s.piece | ((uint32_t)s.rot << 8) | ((uint32_t)s.x << 16) | ((uint32_t)s.y << 24);
  d[1] = (uint32_t)s.val | ((uint32_t)s.flags << 8) | ((uint32_t)s.hp << 16) | ((uint32_t)s.hr << 24);
  d[2] = (uint32_t)s.hv | ((uint32_t)s.bidx << 8) | ((uint32_t)(s.bag & 0xFFFF) << 16);
  d[3] = (uint32_t)(s.bag >> 16);
  d[4] = (uint32_t)((s.bag >> 48) & 0xFF) | ((uint32_t)(s.queue & 0xFFFFFF) << 8);
  d[5] = (uint32_t)(s.queue >> 24);
  d[6] = __float_as_uint(s.score);
  d[7] = s.reserved;
}

// ------------------------------------------------------------------------------------------------
// reset kernel (tg_reset): optional mask, optional staged generator states
// ------------------------------------------------------------------------------------------------
template <int W, int H, int RULESET>
__global__ __launch_bounds__(kBlock) void reset_kernel(const ResetParams p) {
  using G = Geo<W, H>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * G::LDS_SLAB];
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int)(threadIdx.x >> 6));
  const int env = blockIdx.x * kWavesPerBlock + wave;
  if (env >= p.batch) return;
  if (p.mask && rfl((int)p.mask[env]) == 0) return;
  uint8_t* slab = lds + wave * G::LDS_SLAB;
  uint32_t* slab_dw = reinterpret_cast<uint32_t*>(slab);
  EnvS s;
  rec_unpack(s, p.rec + (size_t)env * 8);
  Rng r;
  r.loaded = false;
  r.dirty = false;
  const bool reseed = p.reseed && rfl((int)p.reseed[env]) != 0;
  rng_load(r, reseed ? p.new_rng + env : p.rng + env);
  r.dirty = true;
  env_reset<W, H, RULESET>(s, r, p.rng + env, slab_dw, lane, p.queue_size);
  uint8_t* gboard = p.board + (size_t)env * G::CELLS;
#pragma unroll
  for (int k = 0; k < G::DPL; ++k) {
    const int idx = lane + 64 * k;
    if (idx < G::NDW) slab_st<G::CELLS>(gboard, idx, 0u);
  }
  emit_obs<W, H>(slab, lane, p.obs + (size_t)env * G::CELLS, p.obs_mode, true, c_shapes.mask[s.piece][0], s.x, 0,
                 s.val);
  rec_pack(s, p.rec + (size_t)env * 8, lane);
  rng_store(r, p.rng + env, lane);
  if (p.episode && lane == 0) {
    tg_episode_rec e = p.episode[env];
    e.ret = 0.0f;
    e.len = 0;
    p.episode[env] = e;
  }
}

// ------------------------------------------------------------------------------------------------
// observe kernel (tg_observe; also the TG_OBS_DICT surface after every step)
//   dict layout: board[B][Hp][Wp] | mask[B][Hp][Wp] | holder[B][4][4] | queue[B][4][4*Q]
//   (envs/tetris.py:566-615 _get_obs)
// ------------------------------------------------------------------------------------------------
struct ObserveParams {
  const uint8_t* board;
  const uint32_t* rec;
  uint8_t* obs;
  int batch;
  int obs_mode;
  int queue_size;
  int ruleset;
};

// pixel colour table of the reference (envs/tetris.py:45-75: BASE_PIXELS + tetromino colours), r | g << 8 | b << 16
__constant__ uint32_t c_rgb_lut[9] = {0x000000u, 0x808080u, 0xF0F000u, 0x00F0F0u, 0xF000A0u,
                                      0x00F000u, 0x0000F0u, 0xF00000u, 0x00A0F0u};

constexpr int kObserveEnvsPerWave = 4;  // envs a wave handles back to back, all their loads issued up front

template <int W, int H>
struct ObserveGeo {
  using G = Geo<W, H>;
  static constexpr int IDS = ((G::HP * (G::WP + 28) + 15) / 16) * 16;  // image ids: padded board (+ RGB: queue/holder strip)
  static constexpr int MSK = ((G::HP * G::WP + 15) / 16) * 16;         // DICT: active_tetromino_mask
  static constexpr int RGBB = 3 * IDS;                                 // RGB: the packed image, staged for 16-byte stores
  static constexpr int PER_WAVE = G::LDS_SLAB + IDS + MSK + 64 + RGBB;
};

// one env's observation from its slab (already staged in LDS) and record
template <int W, int H>
__device__ __forceinline__ void observe_env(const ObserveParams& p, const int env, const int lane, uint8_t* slab,
                                            const EnvS& s) {
  using G = Geo<W, H>;
  constexpr int IDS = ObserveGeo<W, H>::IDS, MSK = ObserveGeo<W, H>::MSK;
  const uint32_t shape = c_shapes.mask[s.piece][s.rot];
  const bool coll = test1<W, H>(slab, lane, shape, s.x, s.y);
  if (p.obs_mode == TG_OBS_BOARD || p.obs_mode == TG_OBS_FN) {
    const bool show = p.ruleset == TG_RULESET_NUMPY ? !coll : !(s.flags & TG_F_GAME_OVER);
    emit_obs<W, H>(slab, lane, p.obs + (size_t)env * G::CELLS, p.obs_mode, show, shape, s.x, s.y, s.val);
    return;
  }
  if (p.obs_mode != TG_OBS_RGB && p.obs_mode != TG_OBS_DICT) return;
  // Both image surfaces start from the padded board with the projected piece (envs/tetris.py:543-564), staged in LDS
  // by regions (no per-byte index arithmetic): bedrock fill, playfield cells, piece cells — with row stride RS = the
  // image width (RGB: board | queue/holder strip; DICT: the padded board itself).
  constexpr int WP = G::WP, HP = G::HP, PC = WP * HP;
  const int Q = p.queue_size;
  const bool rgb = p.obs_mode == TG_OBS_RGB;
  const int ML = 4 * (Q > 1 ? Q : 1);
  const int RS = rgb ? WP + ML : WP;
  const int NPX = HP * RS;
  uint8_t* img = slab + G::LDS_SLAB;
  uint8_t* msk = img + IDS;
  uint32_t* lut = reinterpret_cast<uint32_t*>(msk + MSK);
  if (lane < 9) lut[lane] = c_rgb_lut[lane];
  const int fill = rgb ? IDS : MSK;  // bytes of the image area in use (rounded up to 16)
  for (int i = lane * 16; i < fill; i += 1024) *reinterpret_cast<uint4*>(img + i) = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
  if (!rgb)
    for (int i = lane * 16; i < MSK; i += 1024) *reinterpret_cast<uint4*>(msk + i) = make_uint4(0u, 0u, 0u, 0u);
  wave_sync();
  for (int i = lane; i < G::CELLS; i += 64) {
    const int R = i / W, c = i - R * W;
    img[R * RS + 4 + c] = slab[i];
  }
  if (rgb) {
    // wrappers/observation.py:38-74: queue strip (rotation 0, values id + 2) on top, holder at the bottom (ones when
    // empty, else the held matrix with its rotation and values), ones in between (already filled)
    for (int i = lane; i < 16 * Q; i += 64) {
      const int q = i >> 4, R = (i >> 2) & 3, c = i & 3;
      const int t = (int)((s.queue >> (8 * q)) & 0xFF);
      img[R * RS + WP + 4 * q + c] = ((c_shapes.mask[t < 7 ? t : 0][0] >> (4 * R + c)) & 1u) ? (uint8_t)(t + 2) : (uint8_t)0;
    }
    if (lane < 16 && s.hp != TG_NONE)
      img[(HP - 4 + (lane >> 2)) * RS + WP + (lane & 3)] = ((c_shapes.mask[s.hp][s.hr] >> lane) & 1u) ? (uint8_t)s.hv : (uint8_t)0;
  } else {
    // active_tetromino_mask: the n x n bounding box of the native matrix (envs/tetris.py:572-576)
    const int n = c_shapes.n[s.piece];
    const int k = lane >> 2, c = lane & 3;
    if (lane < 16 && k < n && c < n && s.y + k < HP && s.x + c < WP) msk[(s.y + k) * WP + s.x + c] = 1;
  }
  wave_sync();
  if (!coll && lane < 16 && ((shape >> lane) & 1u)) {  // free cells only: board + matrix == matrix value
    const int a = (s.y + (lane >> 2)) * RS + s.x + (lane & 3);
    img[a] = (uint8_t)(img[a] + s.val);
  }
  wave_sync();
  if (rgb) {
    // four pixels → three dwords; the image leaves as one contiguous, coalesced stream
    uint32_t* stg = reinterpret_cast<uint32_t*>(msk + MSK + 64);  // staged, then streamed out 16 bytes per lane
    uint8_t* gdst = p.obs + (size_t)env * (size_t)NPX * 3;
    const bool wide = (NPX * 3) % 16 == 0;
    uint32_t* dst = wide ? stg : reinterpret_cast<uint32_t*>(gdst);
    for (int q = lane; q < NPX / 4; q += 64) {
      const uint32_t ids = *reinterpret_cast<const uint32_t*>(img + 4 * q);
      uint32_t c4[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t id = (ids >> (8 * t)) & 0xFFu;
        c4[t] = lut[id <= 8 ? id : 8];
      }
      dst[3 * q] = c4[0] | (c4[1] << 24);
      dst[3 * q + 1] = (c4[1] >> 8) | (c4[2] << 16);
      dst[3 * q + 2] = (c4[2] >> 16) | (c4[3] << 8);
    }
    if (NPX % 4 != 0 && lane < NPX % 4) {  // (an image that is no whole number of pixel quads: the last pixels byte by byte)
      const int px = NPX / 4 * 4 + lane;
      const uint32_t id = img[px], c = lut[id <= 8 ? id : 8];
      uint8_t* d8 = reinterpret_cast<uint8_t*>(dst) + 3 * px;
      d8[0] = (uint8_t)c; d8[1] = (uint8_t)(c >> 8); d8[2] = (uint8_t)(c >> 16);
    }
    if (wide) {
      wave_sync();
      for (int i = lane * 16; i < NPX * 3; i += 1024)
        *reinterpret_cast<uint4*>(gdst + i) = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(stg) + i);
    }
    return;
  }
  // DICT layout: board[B][Hp][Wp] | mask[B][Hp][Wp] | holder[B][4][4] | queue[B][4][4*Q]  (envs/tetris.py:566-615)
  const size_t B = (size_t)p.batch;
  uint8_t* o_board = p.obs + (size_t)env * PC;
  uint8_t* o_mask = p.obs + B * PC + (size_t)env * PC;
  uint8_t* o_holder = p.obs + 2 * B * PC + (size_t)env * 16;
  uint8_t* o_queue = p.obs + 2 * B * PC + B * 16 + (size_t)env * 16 * Q;
  if constexpr (PC % 16 == 0) {
    for (int i = lane * 16; i < PC; i += 1024) {
      *reinterpret_cast<uint4*>(o_board + i) = *reinterpret_cast<const uint4*>(img + i);
      *reinterpret_cast<uint4*>(o_mask + i) = *reinterpret_cast<const uint4*>(msk + i);
    }
  } else if constexpr (PC % 4 == 0) {
    for (int i = lane * 4; i < PC; i += 256) {
      *reinterpret_cast<uint32_t*>(o_board + i) = *reinterpret_cast<const uint32_t*>(img + i);
      *reinterpret_cast<uint32_t*>(o_mask + i) = *reinterpret_cast<const uint32_t*>(msk + i);
    }
  } else {  // (padded boards that are no whole dwords, e.g. 15 x 18: an env's arrays start at 2- or 1-byte alignment)
    for (int i = lane; i < PC; i += 64) {
      o_board[i] = img[i];
      o_mask[i] = msk[i];
    }
  }
  if (lane < 16) {  // holder: ones when empty, else the held matrix (keeps its rotation and values)
    uint8_t v = 1;
    if (s.hp != TG_NONE) v = ((c_shapes.mask[s.hp][s.hr] >> lane) & 1u) ? (uint8_t)s.hv : (uint8_t)0;
    o_holder[lane] = v;
  }
  for (int i = lane; i < 16 * Q; i += 64) {  // queue strip [4][4*Q]: rotation 0, values id+2 (i = piece, row, column)
    const int q = i >> 4, row = (i >> 2) & 3, c = i & 3;
    const int t = (int)((s.queue >> (8 * q)) & 0xFF);
    const uint32_t m = c_shapes.mask[t < 7 ? t : 0][0];
    o_queue[row * 4 * Q + 4 * q + c] = ((m >> (4 * row + c)) & 1u) ? (uint8_t)(t + 2) : (uint8_t)0;
  }
}


// One-env-per-wave kernels of 65 536 waves are bound by waves x load latency (measured: 30 µs for the 62 MB DICT
// surface); here a wave owns kObserveEnvsPerWave consecutive envs and has all their slab / record loads in flight
// before the first one is packed.
template <int W, int H>
__global__ __launch_bounds__(kBlock) void observe_kernel(const ObserveParams p) {
  using G = Geo<W, H>;
  constexpr int NE = kObserveEnvsPerWave;
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * ObserveGeo<W, H>::PER_WAVE];
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int)(threadIdx.x >> 6));
  const int env0 = (blockIdx.x * kWavesPerBlock + wave) * NE;
  if (env0 >= p.batch) return;
  uint8_t* slab = lds + wave * ObserveGeo<W, H>::PER_WAVE;
  uint32_t* slab_dw = reinterpret_cast<uint32_t*>(slab);
  uint32_t bv[NE][G::DPL];
  uint4 ra[NE], rb[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const int env = env0 + j < p.batch ? env0 + j : p.batch - 1;
    const uint8_t* gboard = p.board + (size_t)env * G::CELLS;
#pragma unroll
    for (int k = 0; k < G::DPL; ++k) bv[j][k] = slab_ld<G::CELLS>(gboard, lane + 64 * k);  // (buffers carry tail padding: unmasked)
    const uint4* r4 = reinterpret_cast<const uint4*>(p.rec + (size_t)env * 8);
    ra[j] = r4[0];
    rb[j] = r4[1];
  }
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    if (env0 + j >= p.batch) break;
#pragma unroll
    for (int k = 0; k < G::DPL; ++k) slab_dw[lane + 64 * k] = bv[j][k];
    const uint32_t d[8] = {rfl(ra[j].x), rfl(ra[j].y), rfl(ra[j].z), rfl(ra[j].w),
                           rfl(rb[j].x), rfl(rb[j].y), rfl(rb[j].z), rfl(rb[j].w)};
    EnvS s;
    raw_to_env(s, d);
    wave_sync();
    observe_env<W, H>(p, env0 + j, lane, slab, s);
    wave_sync();
  }
}

// ---- grouped action surface (wrappers/grouped.py:124-207): the legal mask depends only on the active piece,
// its rotation and the column (frame collisions only) → one thread per env ----
struct GroupedParams {
  const uint32_t* rec;
  uint8_t* mask;      // [B][4W] or null
  int32_t* actions;   // [B] or null: sample one legal action per env (uniform, counter-based hash)
  uint64_t shapes[7];
  uint32_t seed, counter;
  int batch, width;
};

__device__ __forceinline__ bool grouped_legal(const uint64_t* shapes, int piece, int rot, int a, int W) {
  const int col = a >> 2, nrot = (rot + (a & 3)) & 3;
  const int nx = col + 4 - (piece == 0 ? 2 : 1);
  const uint32_t sh = (uint32_t)(shapes[piece] >> (16 * nrot)) & 0xFFFFu;
  bool legal = true;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (sh & (0x1111u << c)) legal = legal && (unsigned)(nx + c - 4) < (unsigned)W;
  return legal;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {  // murmur3 finaliser
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}

// Legal columns of one (piece, rotation) form a range: the placement x = col + padding - n/2 is legal iff every
// occupied matrix column c satisfies 0 <= col - n/2 + c < W  ⇔  col in [n/2 - cmin, W - 1 + n/2 - cmax].
__global__ __launch_bounds__(256) void grouped_kernel(const GroupedParams p) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= p.batch) return;
  const uint32_t d0 = p.rec[(size_t)env * 8];
  const int piece = d0 & 0xFF, rot = (d0 >> 8) & 3;
  const int W = p.width, NA = 4 * W;
  const int n_half = piece == 0 ? 2 : 1;
  int lo[4], hi[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t sh = (uint32_t)(p.shapes[piece] >> (16 * ((rot + r) & 3))) & 0xFFFFu;
    const uint32_t m = (sh | (sh >> 4) | (sh >> 8) | (sh >> 12)) & 0xFu;  // occupied matrix columns
    const int cmin = __builtin_ctz(m), cmax = 31 - __builtin_clz(m);
    lo[r] = n_half - cmin > 0 ? n_half - cmin : 0;
    hi[r] = W - 1 + n_half - cmax < W - 1 ? W - 1 + n_half - cmax : W - 1;
  }
  if (p.mask) {
    uint32_t* mrow = reinterpret_cast<uint32_t*>(p.mask + (size_t)env * NA);  // 4 flags of a column = one dword
    for (int c = 0; c < W; ++c) {
      uint32_t v = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) v |= (c >= lo[r] && c <= hi[r] ? 1u : 0u) << (8 * r);
      mrow[c] = v;
    }
  }
  if (p.actions) {
    int n_legal = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) n_legal += hi[r] >= lo[r] ? hi[r] - lo[r] + 1 : 0;
    int pick = 0;
    if (n_legal > 0) {
      // the k-th legal action in increasing action id (= column-major, rotation-minor) order
      int k = (int)(mix32(mix32(p.seed ^ (uint32_t)env * 0x9E3779B9u) + p.counter) % (uint32_t)n_legal);
      for (int c = 0; c < W; ++c) {
        int nc = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) nc += (c >= lo[r] && c <= hi[r]) ? 1 : 0;
        if (k < nc) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (c >= lo[r] && c <= hi[r]) {
              if (k == 0) pick = c * 4 + r;
              --k;
            }
          break;
        }
        k -= nc;
      }
    }
    p.actions[env] = pick;
  }
}

// packed board (kFeatPacked, tg_kernel_bits.h) <-> uint8 board: cell 2j / 2j + 1 of an env = low / high nibble of its packed byte j
__global__ __launch_bounds__(256) void pack_board_kernel(const uint8_t* __restrict__ board, uint8_t* __restrict__ packed, size_t n_bytes) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // packed dword index: 8 cells
  if (i * 4 >= n_bytes) return;
  const uint2 v = reinterpret_cast<const uint2*>(board)[i];
  const uint32_t even = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u), odd = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);
  reinterpret_cast<uint32_t*>(packed)[i] = (even & 0x0F0F0F0Fu) | ((odd & 0x0F0F0F0Fu) << 4);
}
__global__ __launch_bounds__(256) void unpack_board_kernel(const uint8_t* __restrict__ packed, uint8_t* __restrict__ board, size_t n_bytes) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i * 4 >= n_bytes) return;
  const uint32_t d = reinterpret_cast<const uint32_t*>(packed)[i];
  const uint32_t lo = d & 0x0F0F0F0Fu, hi = (d >> 4) & 0x0F0F0F0Fu;
  reinterpret_cast<uint2*>(board)[i] = make_uint2(__builtin_amdgcn_perm(hi, lo, 0x05010400u), __builtin_amdgcn_perm(hi, lo, 0x07030602u));
}

// stream copy: the practical HBM ceiling reported beside the roofline.  U 16-byte vectors in flight per lane, all loads
// of a tile issued before its stores; NT = nontemporal stores, NTL = nontemporal loads too (tools/ubench/copy_ceiling*.hip
// sweep the variants: at 1 GiB nontemporal both ways with 16k-64k workgroups reaches 6.3-6.4 TB/s)
typedef uint32_t copy_vec __attribute__((ext_vector_type(4)));
template <int U, bool NT, bool NTL = false>
__global__ __launch_bounds__(256) void stream_copy_kernel(const copy_vec* __restrict__ src, copy_vec* __restrict__ dst,
                                                          size_t n) {
  const size_t tile = (size_t)256 * U;
  for (size_t base = (size_t)blockIdx.x * tile; base < n; base += (size_t)gridDim.x * tile) {
    copy_vec v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const size_t i = base + (size_t)k * 256 + threadIdx.x;
      v[k] = NTL ? __builtin_nontemporal_load(src + (i < n ? i : n - 1)) : src[i < n ? i : n - 1];
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const size_t i = base + (size_t)k * 256 + threadIdx.x;
      if (i < n) {
        if (NT) __builtin_nontemporal_store(v[k], dst + i);
        else dst[i] = v[k];
      }
    }
  }
}

}  // namespace tg
