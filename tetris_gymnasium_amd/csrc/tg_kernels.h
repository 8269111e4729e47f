// tg_kernels.h — device code of the MI355X Tetris step engine (gfx950, wave64).
//
// Execution model: ONE ENVIRONMENT PER WAVEFRONT.  All game state of an env (piece, position, queue,
// bag, flags) is wave-uniform, so the whole rule logic is uniform control flow; the 64 lanes are
// used for the data-parallel parts:
//   * the env's uint8 playfield slab (H*W bytes, contiguous in HBM) is fetched with one coalesced
//     dword-per-lane load and staged in LDS;
//   * collision tests: 4 candidate placements x 16 matrix cells = 64 lanes, one LDS byte read each,
//     one wave ballot (v_cmp → SGPR pair) answers move / gravity / commit checks at once;
//   * hard drop: lane r tests "piece at row r" → one ballot gives the collision column, the landing
//     row is a scalar ctz (exactly the reference's step-wise loop: first blocked row below y);
//   * line clear: lane r owns row r, ballot = full-row mask, popcount of the mask below a row is
//     its shift; compaction is a scatter through the LDS slab;
//   * observation pack: piece overlay in LDS, dword-per-lane coalesced store.
// Integer / byte work only: no MFMA.  Bound: HBM (see DESIGN.md for bytes per env-step).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tetris_gym_hip.h"
#include "tg_tables.h"

namespace tg {

__constant__ ShapeTable c_shapes = make_shape_table();

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;

struct KParams {
  uint8_t* board;        // [B][H][W]
  uint32_t* rec;         // [B][8]
  tg_rng_rec* rng;       // [B]
  const int32_t* act32;  // [B] or null
  const int8_t* act8;    // [B] or null
  uint8_t* obs;          // per obs_mode
  uint8_t* final_obs;    // optional (SAME_STEP autoreset)
  float* reward;         // [B]
  uint8_t* terminated;   // [B]
  uint8_t* lines;        // [B]
  tg_episode_rec* episode;  // optional
  int batch;
  int gravity;
  int autoreset;
  int obs_mode;
  int queue_size;
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
  static constexpr int NDW = CELLS / 4;
  static constexpr int DPL = (NDW + 63) / 64;             // dwords per lane
  static constexpr int SLAB = ((CELLS + 15) / 16) * 16;   // LDS bytes per wave
  static constexpr int WP = W + 8;
  static constexpr int HP = H + 4;
  static_assert(CELLS % 4 == 0, "W*H must be a multiple of 4 (dword-per-lane slab transfers)");
  static_assert(H + 4 <= 64, "one row per lane: H + padding must fit a wave");
  static_assert(W <= 32, "row masks are 32-bit");
};

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

// permutation(key, arange(7)): key', sub = split(key); bits_i = hi^lo of threefry(sub, (0,i)); stable sort
__device__ __forceinline__ uint64_t jax_permutation7(uint32_t k0, uint32_t k1) {
  uint32_t s0, s1;
  threefry2x32(k0, k1, 0u, 1u, s0, s1);  // split(key)[1]
  uint32_t bits[7];
#pragma unroll
  for (uint32_t i = 0; i < 7; ++i) {
    uint32_t a, b;
    threefry2x32(s0, s1, 0u, i, a, b);
    bits[i] = a ^ b;
  }
  // rank of element i = #{j : bits[j] < bits[i] or (== and j < i)}  (stable)
  uint64_t bag = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) rank += (bits[j] < bits[i] || (bits[j] == bits[i] && j < i)) ? 1 : 0;
    bag |= (uint64_t)i << (8 * rank);
  }
  return bag;
}

// ------------------------------------------------------------------------------------------------
// wave-level board primitives (slab = this wave's LDS copy of the playfield, H*W bytes)
// ------------------------------------------------------------------------------------------------
// 4 candidate placements tested at once.  Lane group q = lane>>4 tests candidate q, lane&15 = matrix
// cell.  q0 = (shapeT, xT, yT)   q1 = (shapeT, xT, yT+1)   q2 = (shapeC, xC, yC+1)   q3 = (shapeC, xC, yC)
// Cells outside the playfield are bedrock (envs/tetris.py:632-641 pads with 1) → hit.
template <int W, int H>
__device__ __forceinline__ uint64_t test4(const uint8_t* slab, int lane, uint32_t shapeT, int xT, int yT,
                                          uint32_t shapeC, int xC, int yC) {
  const bool useT = lane < 32;
  const uint32_t shape = useT ? shapeT : shapeC;
  const int x = useT ? xT : xC;
  const int y = (useT ? yT : yC) + (((lane + 16) >> 5) & 1);
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
#pragma unroll
  for (int c = 0; c < W; ++c) {
    rowb[c] = slab[r * W + c];
    full &= rowb[c] != 0;
  }
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
    s.bag = jax_permutation7(a0, a1);
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
  if (show) stamp<W, H>(slab, lane, shape, x, y, mode == TG_OBS_FN ? 0xFFu : (uint32_t)val, mode == TG_OBS_BOARD);
  wave_sync();
  const uint32_t* slab_dw = reinterpret_cast<const uint32_t*>(slab);
  uint32_t* dst = reinterpret_cast<uint32_t*>(dst_env);
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
      dst[idx] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// THE step kernel: one env per wavefront
// ------------------------------------------------------------------------------------------------
template <int W, int H, int RULESET>
__global__ __launch_bounds__(kBlock) void step_kernel(const KParams p) {
  using G = Geo<W, H>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * G::SLAB];
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int)(threadIdx.x >> 6));
  const int env = blockIdx.x * kWavesPerBlock + wave;
  if (env >= p.batch) return;
  uint8_t* slab = lds + wave * G::SLAB;
  uint32_t* slab_dw = reinterpret_cast<uint32_t*>(slab);
  uint32_t* gboard = reinterpret_cast<uint32_t*>(p.board + (size_t)env * G::CELLS);

  // stage the playfield slab: coalesced dword-per-lane load → LDS
#pragma unroll
  for (int k = 0; k < G::DPL; ++k) {
    const int idx = lane + 64 * k;
    if (idx < G::NDW) slab_dw[idx] = gboard[idx];
  }
  EnvS s;
  rec_unpack(s, p.rec + (size_t)env * 8);
  const int action = rfl(p.act32 ? p.act32[env] : (int)p.act8[env]);
  Rng r;
  r.loaded = false;
  r.dirty = false;
  const tg_rng_rec* rp = p.rng + env;
  const int Q = p.queue_size;
  wave_sync();

  float reward = 0.0f;
  int lines = 0;
  bool board_dirty = false;
  bool final_coll = false;  // does the active piece collide at its final position (NUMPY projection rule)
  bool stepped = true;

  if (p.autoreset == TG_AUTORESET_NEXT_STEP && (s.flags & TG_F_PENDING_RESET)) {
    env_reset<W, H, RULESET>(s, r, rp, slab_dw, lane, Q);
    board_dirty = true;
    stepped = false;
  } else if (RULESET == TG_RULESET_FUNCTIONAL && (s.flags & TG_F_GAME_OVER)) {
    stepped = false;  // envs/tetris_fn.py:284-289: identity
  }

  if (stepped) {
    const uint32_t shapeC = c_shapes.mask[s.piece][s.rot];
    int tp = s.piece, tr = s.rot, tx = s.x, ty = s.y, tv = s.val;
    bool check = false, forced = false, hard = false;
    if (RULESET == TG_RULESET_NUMPY) {
      // mappings/actions.py:12-19; envs/tetris.py:223-256
      switch (action) {
        case 0: tx -= 1; check = true; break;
        case 1: tx += 1; check = true; break;
        case 2: ty += 1; check = true; break;
        case 3: tr = (s.rot + 1) & 3; check = true; break;  // np.rot90(k=+1)
        case 4: tr = (s.rot + 3) & 3; check = true; break;  // np.rot90(k=-1)
        case 5: hard = true; break;
        case 6:
          if (!(s.flags & TG_F_HAS_SWAPPED)) {  // components/tetromino_holder.py:31-49
            const int op = s.piece, orot = s.rot, ov = s.val;
            if (s.hp == TG_NONE) {
              tp = np_queue_pop(s, r, rp, Q);  // spawn_tetromino (collision result ignored, :248-250)
              tr = 0;
              tv = tp + 2;
            } else {
              tp = s.hp; tr = s.hr; tv = s.hv;
            }
            s.hp = op; s.hr = orot; s.hv = ov;
            s.flags |= TG_F_HAS_SWAPPED;
            tx = np_spawn_x<W>(tp);
            ty = 0;
            forced = true;
          }
          break;
        default: break;  // 7 = no_op (ids outside 0..7 are rejected host-side; on device they are no-ops)
      }
      const uint32_t shapeT = c_shapes.mask[tp][tr];
      const uint64_t m = test4<W, H>(slab, lane, shapeT, tx, ty, shapeC, s.x, s.y);
      const bool cT = (m & 0xFFFFULL) != 0, cT1 = ((m >> 16) & 0xFFFFULL) != 0;
      const bool cC1 = ((m >> 32) & 0xFFFFULL) != 0, cC = (m >> 48) != 0;
      const bool accept = forced || (check && !cT);
      if (accept) { s.piece = tp; s.rot = tr; s.x = tx; s.y = ty; s.val = tv; }
      const uint32_t shape = accept ? shapeT : shapeC;
      const bool coll_here = accept ? cT : cC;
      const bool coll_below = accept ? cT1 : cC1;
      bool commit = hard;
      final_coll = coll_here;
      if (p.gravity && !hard) {  // envs/tetris.py:259-264
        if (!coll_below) { s.y += 1; final_coll = false; }
        else commit = true;
      }
      if (commit) {  // envs/tetris.py:450-479 commit_active_tetromino
        if (coll_here) {
          reward = 0.0f;  // rewards.game_over
          s.flags |= TG_F_GAME_OVER;
        } else {
          const uint64_t C = column_scan<W, H>(slab, lane, shape, s.x);
          const int yd = drop_row(C, s.y);  // :445-448 step-wise drop
          stamp<W, H>(slab, lane, shape, s.x, yd, (uint32_t)s.val, false);
          wave_sync();
          lines = clear_rows<W, H, RULESET>(slab, lane);
          board_dirty = true;
          const int np_ = np_queue_pop(s, r, rp, Q);  // :393-401 spawn_tetromino
          s.piece = np_; s.rot = 0; s.val = np_ + 2;
          s.x = np_spawn_x<W>(np_);
          s.y = 0;
          const bool sc = test1<W, H>(slab, lane, c_shapes.mask[np_][0], s.x, 0);
          reward = (float)(lines * lines * W + 1);  // score + rewards.alife
          if (sc) { s.flags |= TG_F_GAME_OVER; reward = 0.0f; }
          else s.flags &= ~TG_F_GAME_OVER;
          s.flags &= ~TG_F_HAS_SWAPPED;
          final_coll = sc;
        }
      }
    } else {
      // envs/tetris_fn.py:161-273 update_state; lax.switch clamps its index, `action == 6` does not
      const int branch = action < 0 ? 0 : (action > 6 ? 6 : action);
      int drop_reward = 0;
      bool lock = false;
      uint32_t shape = shapeC;
      if (branch == 6) {  // functional/core.py:230-251 hard_drop
        const uint64_t C = column_scan<W, H>(slab, lane, shapeC, s.x);
        const int yd = drop_row(C, s.y);
        drop_reward = 2 * (yd - s.y);
        s.y = yd;
        lock = p.gravity || action == 6;  // blocked below by construction → should_lock iff gravity
      } else {
        switch (branch) {
          case 0: tx -= 1; check = true; break;
          case 1: tx += 1; check = true; break;
          case 2: ty += 1; check = true; break;
          case 3: tr = (s.rot + 3) & 3; check = true; break;  // rotation - 1
          case 4: tr = (s.rot + 1) & 3; check = true; break;  // rotation + 1
          default: break;                                       // 5 = do_nothing
        }
        const uint32_t shapeT = c_shapes.mask[s.piece][tr];
        const uint64_t m = test4<W, H>(slab, lane, shapeT, tx, ty, shapeC, s.x, s.y);
        const bool cT = (m & 0xFFFFULL) != 0, cT1 = ((m >> 16) & 0xFFFFULL) != 0;
        const bool cC1 = ((m >> 32) & 0xFFFFULL) != 0;
        const bool accept = check && !cT;
        if (accept) { s.rot = tr; s.x = tx; s.y = ty; shape = shapeT; }
        if (branch == 2 && accept) drop_reward = 1;
        const bool coll_below = accept ? cT1 : cC1;
        if (p.gravity) {  // functional/core.py:150-182 + tetris_fn.py:242-247
          if (!coll_below) s.y += 1;
          else lock = true;
        }
      }
      int lock_reward = 0;
      if (lock) {  // envs/tetris_fn.py:370-413 place_active_tetromino
        stamp<W, H>(slab, lane, shape, s.x, s.y, (uint32_t)(s.piece + 2), true);
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
        s.piece = (int)((s.bag >> (8 * s.bidx)) & 0xFF);
        s.bidx += 1;
        r.k0 = a0; r.k1 = a1;
        r.dirty = true;
        s.rot = 0; s.val = s.piece + 2;
        s.x = (W + 8) / 2 - 2;
        s.y = 0;
        if (test1<W, H>(slab, lane, c_shapes.mask[s.piece][0], s.x, 0)) s.flags |= TG_F_GAME_OVER;
      }
      const float old = s.score;
      float t = old + (float)drop_reward;
      t = t + (float)lock_reward;
      s.score = t;
      reward = t - old;  // tetris_fn.py:312
    }
  }

  const int terminated = (s.flags & TG_F_GAME_OVER) ? 1 : 0;
  const size_t obs_stride = (size_t)G::CELLS;
  if (terminated && p.autoreset == TG_AUTORESET_SAME_STEP) {
    if (p.final_obs) {
      const bool show = RULESET == TG_RULESET_NUMPY ? !final_coll : false;
      if (board_dirty) wave_sync();
      emit_obs<W, H>(slab, lane, p.final_obs + (size_t)env * obs_stride, p.obs_mode, show,
                     c_shapes.mask[s.piece][s.rot], s.x, s.y, s.val);
    }
    env_reset<W, H, RULESET>(s, r, rp, slab_dw, lane, Q);
    board_dirty = true;
    final_coll = false;
  } else if (terminated && p.autoreset == TG_AUTORESET_NEXT_STEP) {
    s.flags |= TG_F_PENDING_RESET;
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
    const bool show = RULESET == TG_RULESET_NUMPY ? !final_coll : !(s.flags & TG_F_GAME_OVER);
    emit_obs<W, H>(slab, lane, p.obs + (size_t)env * obs_stride, p.obs_mode, show, c_shapes.mask[s.piece][s.rot],
                   s.x, s.y, s.val);
  }
  rec_pack(s, p.rec + (size_t)env * 8, lane);
  rng_store(r, p.rng + env, lane);
  if (lane == 0) {
    p.reward[env] = reward;
    p.terminated[env] = (uint8_t)terminated;
    p.lines[env] = (uint8_t)lines;
    if (p.episode) {
      tg_episode_rec e = p.episode[env];
      e.ret += reward;
      e.len += 1;
      if (terminated) { e.last_ret = e.ret; e.last_len = e.len; e.ret = 0.0f; e.len = 0; }
      p.episode[env] = e;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// reset kernel (tg_reset): optional mask, optional staged generator states
// ------------------------------------------------------------------------------------------------
template <int W, int H, int RULESET>
__global__ __launch_bounds__(kBlock) void reset_kernel(const ResetParams p) {
  using G = Geo<W, H>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * G::SLAB];
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int)(threadIdx.x >> 6));
  const int env = blockIdx.x * kWavesPerBlock + wave;
  if (env >= p.batch) return;
  if (p.mask && rfl((int)p.mask[env]) == 0) return;
  uint8_t* slab = lds + wave * G::SLAB;
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
  uint32_t* gboard = reinterpret_cast<uint32_t*>(p.board + (size_t)env * G::CELLS);
#pragma unroll
  for (int k = 0; k < G::DPL; ++k) {
    const int idx = lane + 64 * k;
    if (idx < G::NDW) gboard[idx] = 0u;
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

template <int W, int H>
__global__ __launch_bounds__(kBlock) void observe_kernel(const ObserveParams p) {
  using G = Geo<W, H>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * G::SLAB];
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int)(threadIdx.x >> 6));
  const int env = blockIdx.x * kWavesPerBlock + wave;
  if (env >= p.batch) return;
  uint8_t* slab = lds + wave * G::SLAB;
  uint32_t* slab_dw = reinterpret_cast<uint32_t*>(slab);
  const uint32_t* gboard = reinterpret_cast<const uint32_t*>(p.board + (size_t)env * G::CELLS);
#pragma unroll
  for (int k = 0; k < G::DPL; ++k) {
    const int idx = lane + 64 * k;
    if (idx < G::NDW) slab_dw[idx] = gboard[idx];
  }
  EnvS s;
  rec_unpack(s, p.rec + (size_t)env * 8);
  wave_sync();
  const uint32_t shape = c_shapes.mask[s.piece][s.rot];
  const bool coll = test1<W, H>(slab, lane, shape, s.x, s.y);
  if (p.obs_mode == TG_OBS_BOARD || p.obs_mode == TG_OBS_FN) {
    const bool show = p.ruleset == TG_RULESET_NUMPY ? !coll : !(s.flags & TG_F_GAME_OVER);
    emit_obs<W, H>(slab, lane, p.obs + (size_t)env * G::CELLS, p.obs_mode, show, shape, s.x, s.y, s.val);
    return;
  }
  if (p.obs_mode != TG_OBS_DICT) return;
  constexpr int WP = G::WP, HP = G::HP, PC = WP * HP;
  const size_t B = (size_t)p.batch;
  uint8_t* o_board = p.obs + (size_t)env * PC;
  uint8_t* o_mask = p.obs + B * PC + (size_t)env * PC;
  uint8_t* o_holder = p.obs + 2 * B * PC + (size_t)env * 16;
  const int Q = p.queue_size;
  uint8_t* o_queue = p.obs + 2 * B * PC + B * 16 + (size_t)env * 16 * Q;
  const int n = c_shapes.n[s.piece];
  // padded board with the projected piece, and the bounding-box mask; 4 bytes per lane per pass
  for (int base = 0; base < PC; base += 256) {
    const int i0 = base + lane * 4;
    uint32_t vb = 0, vm = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j;
      const int R = i / WP, Cc = i - R * WP;
      const int col = Cc - 4;
      const bool inb = (unsigned)col < (unsigned)W && R < H;
      uint32_t b = inb ? (uint32_t)slab[inb ? R * W + col : 0] : 1u;
      const int k = R - s.y, c = Cc - s.x;
      const bool inbox = (unsigned)k < 4u && (unsigned)c < 4u;
      if (!coll && inbox && ((shape >> (4 * k + c)) & 1u)) b = (b + (uint32_t)s.val) & 0xFFu;
      const uint32_t mk = ((unsigned)k < (unsigned)n && (unsigned)c < (unsigned)n) ? 1u : 0u;
      vb |= b << (8 * j);
      vm |= mk << (8 * j);
    }
    if (i0 + 3 < PC) {
      *reinterpret_cast<uint32_t*>(o_board + i0) = vb;
      *reinterpret_cast<uint32_t*>(o_mask + i0) = vm;
    } else {
      for (int j = 0; j < 4; ++j)
        if (i0 + j < PC) { o_board[i0 + j] = (uint8_t)(vb >> (8 * j)); o_mask[i0 + j] = (uint8_t)(vm >> (8 * j)); }
    }
  }
  if (lane < 16) {  // holder: ones when empty, else the held matrix (keeps its rotation and values)
    uint8_t v = 1;
    if (s.hp != TG_NONE) v = ((c_shapes.mask[s.hp][s.hr] >> lane) & 1u) ? (uint8_t)s.hv : (uint8_t)0;
    o_holder[lane] = v;
  }
  for (int i = lane; i < 16 * Q; i += 64) {  // queue strip [4][4*Q]: rotation 0, values id+2
    const int row = i / (4 * Q), cc = i - row * 4 * Q;
    const int q = cc >> 2, c = cc & 3;
    const int t = (int)((s.queue >> (8 * q)) & 0xFF);
    const uint32_t m = c_shapes.mask[t < 7 ? t : 0][0];
    o_queue[i] = ((m >> (4 * row + c)) & 1u) ? (uint8_t)(t + 2) : (uint8_t)0;
  }
}

// float4 grid-stride stream copy: the practical HBM ceiling reported beside the roofline
__global__ __launch_bounds__(256) void stream_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                          size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

}  // namespace tg
