/*
 * tg_oracle.c — CPU restatement of the reference's Tetris step path.  TEST INFRASTRUCTURE ONLY
 * (see tg_oracle.h for who may call it and for the parity status of each part).
 *
 * Plain C, scalar, one env at a time, deliberately written the way the reference is written
 * (padded uint8 boards, n×n matrices rotated with rot90, a step-wise drop loop, whole-board row
 * scans) so that it is an independent check of the bitboard/ballot formulation in the HIP engine.
 * Every function cites the reference file:line it follows (paths relative to /root/reference/).
 */
#include "tg_oracle.h"

#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define NPIECES 7
#define MAXCELLS (68 * 40) /* (H+4) x (W+8) upper bound: H <= 64, W <= 32 */

/* ------------------------------------------------------------------------------------------------
 * numpy PCG64 + SeedSequence + Generator.shuffle   (third-party: numpy 2.2.4 pinned by the reference's
 * poetry.lock:1400-1401; call sites components/tetromino_randomizer.py:42-46, 84)
 * ---------------------------------------------------------------------------------------------- */
typedef unsigned __int128 u128;

#define PCG_MULT_HI 0x2360ED051FC65DA4ULL
#define PCG_MULT_LO 0x4385DF649FCCF645ULL

static u128 rec_state(const tg_rng_rec* r) { return ((u128)r->state_hi << 64) | r->state_lo; }
static u128 rec_inc(const tg_rng_rec* r) { return ((u128)r->inc_hi << 64) | r->inc_lo; }
static void rec_set_state(tg_rng_rec* r, u128 s) {
  r->state_hi = (uint64_t)(s >> 64);
  r->state_lo = (uint64_t)s;
}

/* pcg_setseq_128_step_r + pcg_output_xsl_rr_128_64 (numpy/random/src/pcg64/pcg64.h) */
static uint64_t pcg64_next64(tg_rng_rec* r) {
  const u128 mult = ((u128)PCG_MULT_HI << 64) | PCG_MULT_LO;
  u128 s = rec_state(r) * mult + rec_inc(r);
  rec_set_state(r, s);
  uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s;
  uint64_t x = hi ^ lo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((-rot) & 63));
}

/* pcg64_next32: low half first, high half buffered */
uint32_t tgo_pcg64_next32(tg_rng_rec* r) {
  if (r->has_uint32) {
    r->has_uint32 = 0;
    return r->uinteger;
  }
  uint64_t next = pcg64_next64(r);
  r->has_uint32 = 1;
  r->uinteger = (uint32_t)(next >> 32);
  return (uint32_t)next;
}

/* random_interval (numpy/random/src/distributions/distributions.c): masked rejection, 32-bit path */
static uint64_t random_interval(tg_rng_rec* r, uint64_t max) {
  if (max == 0) return 0;
  uint64_t mask = max, value;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
  mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
  while ((value = (tgo_pcg64_next32(r) & mask)) > max) {}
  return value;
}

/* Generator.shuffle → _shuffle_raw: for i = n-1 .. 1: j = random_interval(i); swap(x[i], x[j]) */
void tgo_pcg64_shuffle7(tg_rng_rec* r, uint8_t* bag) {
  for (int i = NPIECES - 1; i >= 1; --i) {
    int j = (int)random_interval(r, (uint64_t)i);
    uint8_t t = bag[i]; bag[i] = bag[j]; bag[j] = t;
  }
}

/* SeedSequence (numpy/random/bit_generator.pyx): pool of 4 u32, hashmix / mix, generate_state */
#define SS_INIT_A 0x43b0d7e5u
#define SS_MULT_A 0x931e8875u
#define SS_INIT_B 0x8b51f9ddu
#define SS_MULT_B 0x58f38dedu
#define SS_MIX_L 0xca01f9ddu
#define SS_MIX_R 0x4973f715u

static uint32_t ss_hashmix(uint32_t value, uint32_t* hash_const) {
  value ^= *hash_const;
  *hash_const *= SS_MULT_A;
  value *= *hash_const;
  value ^= value >> 16;
  return value;
}
static uint32_t ss_mix(uint32_t x, uint32_t y) {
  uint32_t r = SS_MIX_L * x - SS_MIX_R * y;
  r ^= r >> 16;
  return r;
}

void tgo_seed_pcg64(uint64_t seed, tg_rng_rec* out) {
  /* _coerce_to_uint32_array(int): little-endian u32 words, at least one */
  uint32_t entropy[2];
  int n_ent = 1;
  entropy[0] = (uint32_t)seed;
  entropy[1] = (uint32_t)(seed >> 32);
  if (entropy[1]) n_ent = 2;
  uint32_t pool[4];
  uint32_t hc = SS_INIT_A;
  for (int i = 0; i < 4; ++i) pool[i] = ss_hashmix(i < n_ent ? entropy[i] : 0u, &hc);
  for (int s = 0; s < 4; ++s)
    for (int d = 0; d < 4; ++d)
      if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], &hc));
  /* (entropy longer than the pool: not reachable for a 64-bit seed) */
  /* generate_state(4, uint64) = 8 u32 words, pairs little-endian */
  uint32_t w[8];
  uint32_t hb = SS_INIT_B;
  for (int i = 0; i < 8; ++i) {
    uint32_t v = pool[i & 3];
    v ^= hb;
    hb *= SS_MULT_B;
    v *= hb;
    v ^= v >> 16;
    w[i] = v;
  }
  uint64_t s64[4];
  for (int i = 0; i < 4; ++i) s64[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
  /* pcg64_set_seed: initstate = (s64[0] << 64) | s64[1]; initseq = (s64[2] << 64) | s64[3]
   * pcg_setseq_128_srandom_r: state = 0; inc = (initseq << 1) | 1; step; state += initstate; step */
  const u128 mult = ((u128)PCG_MULT_HI << 64) | PCG_MULT_LO;
  u128 initstate = ((u128)s64[0] << 64) | s64[1];
  u128 initseq = ((u128)s64[2] << 64) | s64[3];
  u128 inc = (initseq << 1) | 1;
  u128 st = 0;
  st = st * mult + inc;
  st += initstate;
  st = st * mult + inc;
  memset(out, 0, sizeof(*out));
  out->state_hi = (uint64_t)(st >> 64);
  out->state_lo = (uint64_t)st;
  out->inc_hi = (uint64_t)(inc >> 64);
  out->inc_lo = (uint64_t)inc;
  out->has_uint32 = 0;
  out->uinteger = 0;
}

/* ------------------------------------------------------------------------------------------------
 * jax.random (third-party: jax 0.5.3, poetry.lock:883-884): threefry2x32, split, permutation.
 * Call sites functional/queue.py:35, 60; envs/tetris_fn.py:328, 399.   PARITY UNPINNED (no jax here)
 * except threefry2x32 itself (Random123 known-answer vectors, tests/test_oracle_rng.py).
 * ---------------------------------------------------------------------------------------------- */
static int g_partitionable = 1;
void tgo_set_threefry_partitionable(int on) { g_partitionable = on ? 1 : 0; }

static uint32_t rotl32(uint32_t x, int d) { return (x << d) | (x >> (32 - d)); }

void tgo_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]) {
  static const int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  uint32_t ks[3] = {key[0], key[1], key[0] ^ key[1] ^ 0x1BD11BDAu};
  uint32_t x0 = ctr[0] + ks[0], x1 = ctr[1] + ks[1];
  for (int blk = 0; blk < 5; ++blk) {
    const int* r = (blk & 1) ? R + 4 : R;
    for (int i = 0; i < 4; ++i) {
      x0 += x1;
      x1 = rotl32(x1, r[i]);
      x1 ^= x0;
    }
    x0 += ks[(blk + 1) % 3];
    x1 += ks[(blk + 2) % 3] + (uint32_t)(blk + 1);
  }
  out[0] = x0;
  out[1] = x1;
}

/* jax._src.prng: _threefry_split_foldlike (partitionable) / _threefry_split_original */
void tgo_jax_split(const uint32_t key[2], uint32_t out0[2], uint32_t out1[2]) {
  if (g_partitionable) {
    uint32_t c0[2] = {0, 0}, c1[2] = {0, 1};
    tgo_threefry2x32(key, c0, out0);
    tgo_threefry2x32(key, c1, out1);
  } else {
    uint32_t ca[2] = {0, 2}, cb[2] = {1, 3}, a[2], b[2];
    tgo_threefry2x32(key, ca, a);
    tgo_threefry2x32(key, cb, b);
    out0[0] = a[0]; out0[1] = b[0];
    out1[0] = a[1]; out1[1] = b[1];
  }
}

/* jax.random.permutation(key, arange(n)) → _shuffle (jax/_src/random.py): num_rounds = ceil(3 ln n / ln(2^32 - 1)) — one round for
 * 2 <= n <= 7, none for n = 1 (the array comes back as it is); per round: key, subkey = split(key);
 * sort_keys = random_bits(subkey, 32, (n,)); stable sort by sort_keys.  functional/queue.py:35 calls it with n = config.queue_size. */
void tgo_jax_permutation(const uint32_t key[2], int n, uint8_t* out) {
  uint32_t k0[2], sub[2], bits[8];
  for (int i = 0; i < n; ++i) out[i] = (uint8_t)i;
  if (n < 2) return;
  tgo_jax_split(key, k0, sub);
  if (g_partitionable) {
    for (uint32_t i = 0; i < (uint32_t)n; ++i) {
      uint32_t c[2] = {0, i}, o[2];
      tgo_threefry2x32(sub, c, o);
      bits[i] = o[0] ^ o[1];
    }
  } else {
    /* counts = iota(n) padded to an even length, halves [0 .. m) / [m .. 2m) with m = ceil(n / 2) (n = 7: [0..3] / [4..6, 0]) */
    const uint32_t m = ((uint32_t)n + 1u) / 2u;
    for (uint32_t i = 0; i < m; ++i) {
      uint32_t c[2] = {i, i + m < (uint32_t)n ? i + m : 0u}, o[2];
      tgo_threefry2x32(sub, c, o);
      bits[i] = o[0];
      if (m + i < 8) bits[m + i] = o[1];
    }
  }
  for (int i = 1; i < n; ++i) { /* stable insertion sort */
    uint8_t v = out[i];
    int j = i - 1;
    while (j >= 0 && bits[out[j]] > bits[v]) { out[j + 1] = out[j]; --j; }
    out[j + 1] = v;
  }
}
void tgo_jax_permutation7(const uint32_t key[2], uint8_t out[7]) { tgo_jax_permutation(key, 7, out); }

/* ------------------------------------------------------------------------------------------------
 * Tetromino tables
 * ---------------------------------------------------------------------------------------------- */
/* envs/tetris.py:47-75 (binary native matrices, order I O T S Z J L) */
static const int NATIVE_N[NPIECES] = {4, 2, 3, 3, 3, 3, 3};
static const uint8_t NATIVE_M[NPIECES][16] = {
    {0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0}, /* I 4x4 */
    {1, 1, 1, 1},                                     /* O 2x2 */
    {0, 1, 0, 1, 1, 1, 0, 0, 0},                      /* T */
    {0, 1, 1, 1, 1, 0, 0, 0, 0},                      /* S */
    {1, 1, 0, 0, 1, 1, 0, 0, 0},                      /* Z */
    {1, 0, 0, 1, 1, 1, 0, 0, 0},                      /* J */
    {0, 0, 1, 1, 1, 1, 0, 0, 0},                      /* L */
};

typedef struct {
  int index; /* 0..6, -1 = none */
  int n;     /* matrix is n x n */
  int rot;   /* bookkeeping: rot90(k=+1) count mod 4 */
  uint8_t val;
  uint8_t m[16]; /* row-major n*n, cell values */
} tetromino;

/* np.rot90(m, k=1): out[i][j] = m[j][n-1-i];  k=-1: out[i][j] = m[n-1-j][i]   (envs/tetris.py:429-443) */
static tetromino rotate(const tetromino* t, int clockwise_flag) {
  tetromino o = *t;
  int n = t->n;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      o.m[i * n + j] = clockwise_flag ? t->m[j * n + (n - 1 - i)] : t->m[(n - 1 - j) * n + i];
  o.rot = (t->rot + (clockwise_flag ? 1 : 3)) & 3;
  return o;
}

/* envs/tetris.py:658-679 offset_tetromino_id: matrix * (i + 2) */
static tetromino make_tetromino(int index, int rot, int val) {
  tetromino t;
  memset(&t, 0, sizeof(t));
  t.index = index;
  t.n = NATIVE_N[index];
  t.val = (uint8_t)val;
  for (int i = 0; i < t.n * t.n; ++i) t.m[i] = (uint8_t)(NATIVE_M[index][i] * val);
  for (int r = 0; r < (rot & 3); ++r) t = rotate(&t, 1);
  return t;
}

int tgo_np_matrix(int piece, int rot, uint8_t* out) {
  tetromino t = make_tetromino(piece, rot, piece + 2);
  memcpy(out, t.m, (size_t)(t.n * t.n));
  return t.n;
}

/* functional/tetrominoes.py:82-148: rotations by rot90(k=1), zero-padded bottom/right to 4x4, binary */
void tgo_fn_matrix(int piece, int rot, uint8_t out[16]) {
  tetromino t = make_tetromino(piece, rot, 1);
  memset(out, 0, 16);
  for (int i = 0; i < t.n; ++i)
    for (int j = 0; j < t.n; ++j) out[i * 4 + j] = t.m[i * t.n + j];
}

/* ------------------------------------------------------------------------------------------------
 * Env state (both rule-sets share the struct; unused members stay zero)
 * ---------------------------------------------------------------------------------------------- */
/* mappings/actions.py:5-19 ActionsMapping and mappings/rewards.py:5-15 RewardsMapping: per-batch data
 * (envs/tetris.py:83-84, 189-190).  Field order = declaration order of the dataclasses. */
typedef struct {
  int move_left, move_right, move_down, rotate_clockwise, rotate_counterclockwise, hard_drop, swap, no_op;
  double alife, clear_line, game_over, invalid_action;
} mappings_t;
static const mappings_t STOCK_MAPPINGS = {0, 1, 2, 3, 4, 5, 6, 7, 1.0, 1.0, 0.0, -0.1};

typedef struct {
  const mappings_t* map;
  double last_reward; /* the reward of the last step as Python computes it (int + float in double precision) */
  int last_flags;     /* TG_OUT_* of the last step (tg_step_out.flags: commit_active_tetromino ran / the grouped action was masked out) */
  int W, H, P, Wp, Hp, gravity, qsize;
  uint8_t board[MAXCELLS]; /* padded (H+P) x (W+2P), bedrock = 1 */
  tetromino active;
  int x, y;
  int game_over, has_swapped, pending_reset;
  int holder_len;
  tetromino holder;
  uint8_t queue[7]; /* NUMPY: visible queue (qsize entries) */
  uint8_t bag[7];
  int bag_index;
  tg_rng_rec rng;
  int rng_valid;
  /* functional */
  int fn_rot;
  float score;
} env_t;

struct tgo_batch {
  int ruleset, batch, env_offset;
  env_t* e;
  mappings_t map;
};

/* envs/tetris.py:632-641 create_board / functional/core.py:46-63 */
static void create_board(env_t* e) {
  for (int r = 0; r < e->Hp; ++r)
    for (int c = 0; c < e->Wp; ++c)
      e->board[r * e->Wp + c] = (r >= e->H || c < e->P || c >= e->P + e->W) ? 1 : 0;
}

/* envs/tetris.py:408-427 collision: any(board[y:y+n, x:x+n][matrix > 0] > 0).
 * Cells outside the padded array (numpy would raise on the clipped slice) count as collisions. */
static int collision(const env_t* e, const tetromino* t, int x, int y) {
  for (int i = 0; i < t->n; ++i)
    for (int j = 0; j < t->n; ++j)
      if (t->m[i * t->n + j] > 0) {
        int r = y + i, c = x + j;
        if (r < 0 || r >= e->Hp || c < 0 || c >= e->Wp) return 1;
        if (e->board[r * e->Wp + c] > 0) return 1;
      }
  return 0;
}

/* ---- BagRandomizer (components/tetromino_randomizer.py:49-102) ---- */
static void shuffle_bag(env_t* e) { /* :82-85 */
  tgo_pcg64_shuffle7(&e->rng, e->bag);
  e->bag_index = 0;
}
static int randomizer_next(env_t* e) { /* :65-80 */
  int t = e->bag[e->bag_index];
  e->bag_index += 1;
  if (e->bag_index >= NPIECES) shuffle_bag(e);
  return t;
}
static void randomizer_reset(env_t* e, uint64_t seed, uint64_t default_seed) { /* :87-91 + :34-46 */
  if (seed > 0) {
    tgo_seed_pcg64(seed, &e->rng);
    e->rng_valid = 1;
  } else if (!e->rng_valid) {
    /* reference: np.random.default_rng() (OS entropy).  Deterministic stand-in, same in the engine. */
    tgo_seed_pcg64(default_seed, &e->rng);
    e->rng_valid = 1;
  }
  for (int i = 0; i < NPIECES; ++i) e->bag[i] = (uint8_t)i;
  shuffle_bag(e);
}

/* ---- TetrominoQueue (components/tetromino_queue.py:7-56) ---- */
static void queue_reset(env_t* e, uint64_t seed, uint64_t default_seed) { /* :23-33 */
  randomizer_reset(e, seed, default_seed);
  for (int i = 0; i < e->qsize; ++i) e->queue[i] = (uint8_t)randomizer_next(e);
}
static int queue_get_next(env_t* e) { /* :35-42 */
  int t = e->queue[0];
  for (int i = 0; i + 1 < e->qsize; ++i) e->queue[i] = e->queue[i + 1];
  e->queue[e->qsize - 1] = (uint8_t)randomizer_next(e);
  return t;
}

/* envs/tetris.py:536-541 reset_tetromino_position */
static void reset_position(env_t* e) {
  e->x = e->Wp / 2 - e->active.n / 2;
  e->y = 0;
}

/* envs/tetris.py:393-401 spawn_tetromino */
static int spawn_tetromino(env_t* e) {
  int idx = queue_get_next(e);
  e->active = make_tetromino(idx, 0, idx + 2);
  reset_position(e);
  return !collision(e, &e->active, e->x, e->y);
}

/* envs/tetris.py:481-512 clear_filled_rows (on the padded board) */
static int np_clear_filled_rows(const env_t* e, uint8_t* board) {
  int filled[80], n_filled = 0;
  for (int r = 0; r < e->Hp; ++r) {
    int any0 = 0, all1 = 1;
    for (int c = 0; c < e->Wp; ++c) {
      if (board[r * e->Wp + c] == 0) any0 = 1;
      if (board[r * e->Wp + c] != 1) all1 = 0;
    }
    filled[r] = (!any0) && (!all1);
    n_filled += filled[r];
  }
  if (n_filled > 0) {
    uint8_t tmp[MAXCELLS];
    int dst = 0;
    for (int r = 0; r < n_filled; ++r, ++dst) /* free rows on top, bedrock left/right */
      for (int c = 0; c < e->Wp; ++c)
        tmp[dst * e->Wp + c] = (c < e->P || c >= e->P + e->W) ? 1 : 0;
    for (int r = 0; r < e->Hp; ++r)
      if (!filled[r]) {
        memcpy(tmp + dst * e->Wp, board + r * e->Wp, (size_t)e->Wp);
        ++dst;
      }
    memcpy(board, tmp, (size_t)(e->Hp * e->Wp));
  }
  return n_filled;
}

int tgo_np_score(int rows, int width) { return rows * rows * width; } /* envs/tetris.py:621-630 */

/* envs/tetris.py:543-564 project_tetromino(t, x, y) into out (copy of board) */
static void np_project(const env_t* e, const tetromino* t, int x, int y, uint8_t* out) {
  memcpy(out, e->board, (size_t)(e->Hp * e->Wp));
  if (collision(e, t, x, y)) return;
  for (int i = 0; i < t->n; ++i)
    for (int j = 0; j < t->n; ++j) out[(y + i) * e->Wp + x + j] += t->m[i * t->n + j];
}

/* envs/tetris.py:450-479 commit_active_tetromino */
static void commit_active_tetromino(env_t* e, double* reward, int* lines_cleared) {
  e->last_flags |= TG_OUT_COMMITTED;
  *lines_cleared = 0;
  if (collision(e, &e->active, e->x, e->y)) {
    *reward = e->map->game_over; /* :462 */
    e->game_over = 1;
  } else {
    while (!collision(e, &e->active, e->x, e->y + 1)) e->y += 1; /* :445-448 */
    uint8_t proj[MAXCELLS];
    np_project(e, &e->active, e->x, e->y, proj); /* :403-406 */
    memcpy(e->board, proj, (size_t)(e->Hp * e->Wp));
    *lines_cleared = np_clear_filled_rows(e, e->board);
    *reward = tgo_np_score(*lines_cleared, e->W);
    e->game_over = !spawn_tetromino(e);
    *reward += e->map->alife;                          /* :472 */
    if (e->game_over) *reward = e->map->game_over;     /* :473-474 */
    e->has_swapped = 0;
  }
}

/* envs/tetris.py:274-307 reset */
static void np_reset(env_t* e, uint64_t seed, uint64_t default_seed) {
  create_board(e);
  e->game_over = 0;
  queue_reset(e, seed, default_seed);
  int idx = queue_get_next(e);
  e->active = make_tetromino(idx, 0, idx + 2);
  reset_position(e);
  e->holder_len = 0;
  e->has_swapped = 0;
  e->pending_reset = 0;
}

/* envs/tetris.py:203-272 step.  Returns reward; *lines out. */
static double np_step(env_t* e, int action, int* lines) {
  double reward = 0;
  const mappings_t* A = e->map; /* self.actions: the elif chain compares with the mapping's fields, in this order */
  *lines = 0;
  e->last_flags &= TG_OUT_ILLEGAL; /* (grouped_step sets that bit in front of the no_op it forwards) */
  tetromino* t = &e->active;
  if (action == A->move_left) {
    if (!collision(e, t, e->x - 1, e->y)) e->x -= 1;
  } else if (action == A->move_right) {
    if (!collision(e, t, e->x + 1, e->y)) e->x += 1;
  } else if (action == A->move_down) {
    if (!collision(e, t, e->x, e->y + 1)) e->y += 1;
  } else if (action == A->rotate_clockwise) { /* np.rot90(k=1) */
    tetromino r = rotate(t, 1);
    if (!collision(e, &r, e->x, e->y)) e->active = r;
  } else if (action == A->rotate_counterclockwise) { /* np.rot90(k=-1) */
    tetromino r = rotate(t, 0);
    if (!collision(e, &r, e->x, e->y)) e->active = r;
  } else if (action == A->swap) { /* (:242-252, components/tetromino_holder.py:31-49) */
    if (!e->has_swapped) {
      tetromino cur = e->active;
      int had = e->holder_len;
      tetromino prev = e->holder;
      e->holder = cur;
      e->holder_len = 1;
      e->has_swapped = 1;
      if (!had) {
        (void)spawn_tetromino(e); /* collision result ignored (:248-250) */
      } else {
        e->active = prev;
        reset_position(e);
      }
    }
  } else if (action == A->hard_drop) {
    commit_active_tetromino(e, &reward, lines);
  } /* no_op, or an id no field holds: pass */

  if (e->gravity && action != A->hard_drop) { /* :259-264 */
    if (!collision(e, &e->active, e->x, e->y + 1)) {
      e->y += 1;
    } else {
      commit_active_tetromino(e, &reward, lines);
    }
  }
  e->last_reward = reward;
  return reward;
}

/* envs/tetris.py:566-615 _get_obs */
static void np_obs_dict(const env_t* e, uint8_t* board, uint8_t* mask, uint8_t* holder, uint8_t* queue) {
  int cells = e->Hp * e->Wp, P = e->P;
  if (board) np_project(e, &e->active, e->x, e->y, board);
  if (mask) {
    memset(mask, 0, (size_t)cells);
    for (int i = 0; i < e->active.n; ++i)
      for (int j = 0; j < e->active.n; ++j) {
        int r = e->y + i, c = e->x + j;
        if (r >= 0 && r < e->Hp && c >= 0 && c < e->Wp) mask[r * e->Wp + c] = 1;
      }
  }
  if (holder) {
    if (e->holder_len > 0) {
      memset(holder, 0, (size_t)(P * P));
      for (int i = 0; i < e->holder.n; ++i)
        for (int j = 0; j < e->holder.n; ++j) holder[i * P + j] = e->holder.m[i * e->holder.n + j];
    } else {
      memset(holder, 1, (size_t)(P * P));
    }
  }
  if (queue) {
    int QW = P * e->qsize;
    memset(queue, 0, (size_t)(P * QW));
    for (int q = 0; q < e->qsize; ++q) {
      tetromino t = make_tetromino(e->queue[q], 0, e->queue[q] + 2);
      for (int i = 0; i < t.n; ++i)
        for (int j = 0; j < t.n; ++j) queue[i * QW + q * P + j] = t.m[i * t.n + j];
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * Functional rule-set (functional/core.py, functional/queue.py, envs/tetris_fn.py)
 * ---------------------------------------------------------------------------------------------- */
/* functional/core.py:86-100 collision: dynamic_slice clamps the start so the 4x4 window fits */
static int fn_collision(const env_t* e, const uint8_t m[16], int x, int y) {
  if (y < 0) y = 0;
  if (y > e->Hp - 4) y = e->Hp - 4;
  if (x < 0) x = 0;
  if (x > e->Wp - 4) x = e->Wp - 4;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (m[i * 4 + j] > 0 && e->board[(y + i) * e->Wp + x + j] > 0) return 1;
  return 0;
}

int tgo_fn_score(int rows) { /* functional/core.py:124-146 */
  int standard = rows > 0 ? rows * 200 - 100 : 0;
  return rows == 4 ? 800 : standard;
}

/* functional/core.py:185-227 clear_filled_rows incl. the jnp.take negative-index wrap:
 * cleared rows get index -H, jnp.take(mode=None → "fill") first wraps negative indices (+H) → row 0 of
 * the PRE-compaction sub-board, so vacated top rows are copies of old row 0 (zeros when row 0 is empty). */
static int fn_clear_filled_rows(env_t* e) {
  int H = e->H, W = e->W, P = e->P, Wp = e->Wp;
  int filled[64], n = 0;
  for (int r = 0; r < H; ++r) {
    int all = 1;
    for (int c = 0; c < W; ++c)
      if (!(e->board[r * Wp + P + c] > 0)) all = 0;
    filled[r] = all;
    n += all;
  }
  if (n > 0) {
    int idx[64], k = 0;
    for (int r = 0; r < H; ++r)
      if (filled[r]) idx[k++] = -H; /* sorted ascending: the -H entries come first */
    for (int r = 0; r < H; ++r)
      if (!filled[r]) idx[k++] = r;
    uint8_t sub[64 * 32];
    for (int r = 0; r < H; ++r) {
      int src = idx[r] < 0 ? idx[r] + H : idx[r];
      for (int c = 0; c < W; ++c) sub[r * W + c] = e->board[src * Wp + P + c];
    }
    for (int r = 0; r < H; ++r)
      for (int c = 0; c < W; ++c) e->board[r * Wp + P + c] = sub[r * W + c];
  }
  return n;
}

/* test hook: fn_clear_filled_rows on a bare H x W playfield (values as stored: 0 empty, >0 filled), in place */
int tgo_fn_clear_rows(uint8_t* playfield, int W, int H) {
  env_t* e = (env_t*)calloc(1, sizeof(env_t));
  e->W = W; e->H = H; e->P = 4; e->Wp = W + 8; e->Hp = H + 4;
  create_board(e);
  for (int r = 0; r < H; ++r)
    for (int c = 0; c < W; ++c) e->board[r * e->Wp + e->P + c] = playfield[r * W + c];
  int n = fn_clear_filled_rows(e);
  for (int r = 0; r < H; ++r)
    for (int c = 0; c < W; ++c) playfield[r * W + c] = e->board[r * e->Wp + e->P + c];
  free(e);
  return n;
}

/* functional/queue.py:38-67 bag_queue_get_next_element (returned key is discarded by the caller,
 * envs/tetris_fn.py:388-390) */
static int fn_queue_next(env_t* e) {
  if (e->bag_index >= e->qsize) { /* :67 queue_index >= config.queue_size */
    uint32_t nk[2], sub[2];
    tgo_jax_split(e->rng.key, nk, sub);
    memset(e->bag, 0, sizeof e->bag);
    tgo_jax_permutation(sub, e->qsize, e->bag); /* :35 permutation(key, arange(config.queue_size)) */
    e->bag_index = 1;
    return e->bag[0];
  }
  return e->bag[e->bag_index++];
}

/* envs/tetris_fn.py:318-367 reset(key) → state.rng_key = subkey; queue = permutation(key') */
static void fn_reset(env_t* e, const uint32_t key_in[2]) {
  create_board(e);
  uint32_t key[2], subkey[2];
  tgo_jax_split(key_in, key, subkey);
  memset(e->bag, 0, sizeof e->bag);
  tgo_jax_permutation(key, e->qsize, e->bag);
  e->bag_index = 0;
  e->rng.key[0] = subkey[0];
  e->rng.key[1] = subkey[1];
  int idx = fn_queue_next(e);
  e->active = make_tetromino(idx, 0, idx + 2);
  e->fn_rot = 0;
  e->x = (e->W + 2 * e->P) / 2 - 4 / 2; /* functional/core.py:79-83 */
  e->y = 0;
  e->game_over = 0;
  e->score = 0.0f;
  e->pending_reset = 0;
  e->holder_len = 0;
  e->has_swapped = 0;
}

/* envs/tetris_fn.py:370-413 place_active_tetromino */
static void fn_place(env_t* e, int* lock_reward, int* lines) {
  uint8_t m[16];
  tgo_fn_matrix(e->active.index, e->fn_rot, m);
  /* functional/core.py:103-121 project: board + m * id (no collision check) */
  int id = e->active.index + 2;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (m[i * 4 + j]) {
        int r = e->y + i, c = e->x + j;
        if (r >= 0 && r < e->Hp && c >= 0 && c < e->Wp)
          e->board[r * e->Wp + c] = (uint8_t)(e->board[r * e->Wp + c] + id);
      }
  *lines = fn_clear_filled_rows(e);
  *lock_reward = tgo_fn_score(*lines);
  int idx = fn_queue_next(e);
  e->active = make_tetromino(idx, 0, idx + 2);
  e->fn_rot = 0;
  e->x = (e->W + 2 * e->P) / 2 - 2;
  e->y = 0;
  uint8_t nm[16];
  tgo_fn_matrix(idx, 0, nm);
  e->game_over = fn_collision(e, nm, e->x, e->y); /* functional/core.py:284-302 */
  uint32_t k0[2], k1[2];
  tgo_jax_split(e->rng.key, k0, k1); /* :399 rng_key = split(rng_key)[0] */
  e->rng.key[0] = k0[0];
  e->rng.key[1] = k0[1];
}

/* envs/tetris_fn.py:161-273 update_state + :276-315 step.  Returns reward (float32). */
static float fn_step(env_t* e, int action, int* lines) {
  *lines = 0;
  if (e->game_over) return 0.0f; /* :284-289 identity */
  int branch = action < 0 ? 0 : (action > 6 ? 6 : action); /* lax.switch clamps its index */
  uint8_t m[16];
  tgo_fn_matrix(e->active.index, e->fn_rot, m);
  int x = e->x, y = e->y, rot = e->fn_rot;
  int drop_reward = 0;
  if (branch == 0 && !fn_collision(e, m, x - 1, y)) x -= 1;
  if (branch == 1 && !fn_collision(e, m, x + 1, y)) x += 1;
  if (branch == 2) {
    int ny = !fn_collision(e, m, x, y + 1) ? y + 1 : y;
    drop_reward = ny - y;
    y = ny;
  } else if (branch == 6) { /* functional/core.py:230-251 hard_drop */
    int ny = y;
    while (!fn_collision(e, m, x, ny + 1)) ny += 1;
    drop_reward = 2 * (ny - y);
    y = ny;
  }
  if (branch == 3 || branch == 4) {
    int nr = branch == 4 ? (rot + 1) % 4 : (rot + 3) % 4;
    uint8_t nm[16];
    tgo_fn_matrix(e->active.index, nr, nm);
    if (!fn_collision(e, nm, x, y)) {
      rot = nr;
      memcpy(m, nm, 16);
    }
  }
  int y_gravity = y;
  if (e->gravity && !fn_collision(e, m, x, y + 1)) y_gravity = y + 1; /* core.py:150-182 */
  int should_lock = (y_gravity == y) && e->gravity;
  e->x = x;
  e->y = y_gravity;
  e->fn_rot = rot;
  e->active.rot = rot;
  int lock_reward = 0;
  if (should_lock || action == 6) fn_place(e, &lock_reward, lines);
  float old = e->score;
  float s = old + (float)drop_reward;
  s = s + (float)lock_reward;
  e->score = s;
  return s - old;
}

/* envs/tetris_fn.py:137-158 get_observation → int8 [H][W] */
static void fn_obs(const env_t* e, int8_t* out) {
  for (int r = 0; r < e->H; ++r)
    for (int c = 0; c < e->W; ++c) out[r * e->W + c] = e->board[r * e->Wp + e->P + c] > 0 ? 1 : 0;
  if (!e->game_over) {
    uint8_t m[16];
    tgo_fn_matrix(e->active.index, e->fn_rot, m);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j)
        if (m[i * 4 + j]) {
          int r = e->y + i, c = e->x + j - e->P;
          if (r >= 0 && r < e->H && c >= 0 && c < e->W) out[r * e->W + c] = (int8_t)(out[r * e->W + c] - 1);
        }
  }
}

/* ------------------------------------------------------------------------------------------------
 * Grouped actions (wrappers/grouped.py) — NUMPY rule-set
 * ---------------------------------------------------------------------------------------------- */
/* :104-122 collision_with_frame: any(board[slices][matrix > 0] == 1) */
static int collision_with_frame(const env_t* e, const tetromino* t, int x, int y) {
  for (int i = 0; i < t->n; ++i)
    for (int j = 0; j < t->n; ++j)
      if (t->m[i * t->n + j] > 0) {
        int r = y + i, c = x + j;
        if (r < 0 || r >= e->Hp || c < 0 || c >= e->Wp) return 1;
        if (e->board[r * e->Wp + c] == 1) return 1;
      }
  return 0;
}

/* :124-207 observation(): legal mask + (optionally) the 4W placement boards */
static void grouped_enumerate(const env_t* e, uint8_t* legal, uint8_t* boards) {
  int cells = e->Hp * e->Wp;
  tetromino t = e->active;
  for (int xb = 0; xb < e->W; ++xb) {
    for (int r = 0; r < 4; ++r) {
      int y = 0;
      if (r > 0) t = rotate(&t, 1);
      int x = xb + e->P - t.n / 2;
      while (!collision(e, &t, x, y + 1)) y += 1;
      int a = xb * 4 + r;
      uint8_t* out = boards ? boards + (size_t)a * cells : NULL;
      if (collision_with_frame(e, &t, x, y)) {
        legal[a] = 0;
        if (out) memset(out, 1, (size_t)cells); /* np.ones_like(board_obs) */
      } else if (collision(e, &t, x, y)) {
        legal[a] = 1;
        if (out) memset(out, 0, (size_t)cells); /* game-over placement */
      } else {
        legal[a] = 1;
        if (out) {
          np_project(e, &t, x, y, out);
          (void)np_clear_filled_rows(e, out);
        }
      }
    }
    t = rotate(&t, 1);
  }
}

/* :209-269 step */
static float grouped_step(env_t* e, int action, int terminate_on_illegal, int* lines, int* terminated,
                          int* illegal) {
  uint8_t legal[4 * 32];
  grouped_enumerate(e, legal, NULL); /* the mask the wrapper cached from its last observation() */
  *illegal = 0;
  e->last_flags = 0;
  if (legal[action] == 0) {
    *illegal = 1;
    e->last_flags = TG_OUT_ILLEGAL;
    if (terminate_on_illegal) {
      *terminated = 1;
      *lines = 0;
    } else {
      (void)np_step(e, e->map->no_op, lines); /* wrappers/grouped.py:234 */
      *terminated = e->game_over;
    }
    e->last_reward = e->map->invalid_action;  /* :238 */
    return (float)e->map->invalid_action;
  }
  int xb = action / 4, r = action % 4;
  int x = xb + e->P - e->active.n / 2;
  tetromino nt = e->active;
  for (int i = 0; i < r; ++i) nt = rotate(&nt, 1);
  e->x = x;
  e->active = nt;
  double reward = np_step(e, e->map->hard_drop, lines); /* :257-259 */
  *terminated = e->game_over;
  return (float)reward;
}

/* wrappers/observation.py:236-278 FeatureVectorObservation.observation (+ calc_* :177-234) */
void tgo_feature_vector(const uint8_t* board_in, const uint8_t* mask, int W, int H, int P, uint8_t* out) {
  int Wp = W + 2 * P, Hp = H + P;
  uint8_t b[MAXCELLS];
  memcpy(b, board_in, (size_t)(Hp * Wp));
  /* board_obs[active_tetromino_mask] = 0 with a uint8 (integer) index array: zeroes ROW 0 and, if the
   * mask holds any 1, ROW 1 — it does not remove the active piece. */
  int any1 = 0;
  for (int i = 0; i < Hp * Wp; ++i) any1 |= mask[i] == 1;
  memset(b, 0, (size_t)Wp);
  if (any1) memset(b + Wp, 0, (size_t)Wp);
  int heights[32];
  for (int c = 0; c < W; ++c) {
    int h = 0;
    for (int r = 0; r < H; ++r)
      if (b[r * Wp + P + c] != 0) { h = H - r; break; }
    heights[c] = h;
  }
  int maxh = 0, holes = 0, bump = 0;
  for (int c = 0; c < W; ++c) {
    if (heights[c] > maxh) maxh = heights[c];
    int seen = 0;
    for (int r = 0; r < H; ++r) {
      if (b[r * Wp + P + c] != 0) seen = 1;
      else if (seen) holes++;
    }
    if (c > 0) bump += abs(heights[c] - heights[c - 1]);
  }
  for (int c = 0; c < W; ++c) out[c] = (uint8_t)heights[c];
  out[W] = (uint8_t)maxh;
  out[W + 1] = (uint8_t)holes;
  out[W + 2] = (uint8_t)bump;
}

/* ------------------------------------------------------------------------------------------------
 * Batch API
 * ---------------------------------------------------------------------------------------------- */
tgo_batch* tgo_create(int ruleset, int width, int height, int padding, int queue_size, int gravity,
                      int batch, int env_offset) {
  if (width < 4 || width > 32 || height < 4 || height > 60 || padding != 4 || batch <= 0) return NULL;
  if (queue_size < 1 || queue_size > 7) return NULL;
  tgo_batch* b = (tgo_batch*)calloc(1, sizeof(*b));
  if (!b) return NULL;
  b->ruleset = ruleset;
  b->batch = batch;
  b->env_offset = env_offset;
  b->e = (env_t*)calloc((size_t)batch, sizeof(env_t));
  if (!b->e) { free(b); return NULL; }
  b->map = STOCK_MAPPINGS;
  for (int i = 0; i < batch; ++i) {
    env_t* e = &b->e[i];
    e->map = &b->map;
    e->W = width; e->H = height; e->P = padding;
    e->Wp = width + 2 * padding; e->Hp = height + padding;
    e->gravity = gravity; e->qsize = queue_size;
    create_board(e);
    e->active = make_tetromino(0, 0, 2);
  }
  return b;
}

/* Tetris(actions_mapping=..., rewards_mapping=...) (envs/tetris.py:83-84): NULL keeps what is set */
void tgo_set_mappings(tgo_batch* b, const int32_t action_ids[8], const double rewards[4]) {
  if (action_ids) {
    b->map.move_left = action_ids[0]; b->map.move_right = action_ids[1]; b->map.move_down = action_ids[2];
    b->map.rotate_clockwise = action_ids[3]; b->map.rotate_counterclockwise = action_ids[4];
    b->map.hard_drop = action_ids[5]; b->map.swap = action_ids[6]; b->map.no_op = action_ids[7];
  }
  if (rewards) {
    b->map.alife = rewards[0]; b->map.clear_line = rewards[1]; b->map.game_over = rewards[2];
    b->map.invalid_action = rewards[3];
  }
}

void tgo_last_rewards(const tgo_batch* b, double* out) {
  for (int i = 0; i < b->batch; ++i) out[i] = b->e[i].last_reward;
}

void tgo_last_flags(const tgo_batch* b, uint8_t* out) {
  for (int i = 0; i < b->batch; ++i) out[i] = (uint8_t)b->e[i].last_flags;
}

void tgo_destroy(tgo_batch* b) {
  if (!b) return;
  free(b->e);
  free(b);
}

static void reset_one(tgo_batch* b, int i, uint64_t seed, int have_seed) {
  env_t* e = &b->e[i];
  if (b->ruleset == TG_RULESET_NUMPY) {
    np_reset(e, have_seed ? seed : 0, (uint64_t)(1 + b->env_offset + i));
  } else {
    uint32_t key[2];
    if (have_seed) { key[0] = (uint32_t)(seed >> 32); key[1] = (uint32_t)seed; }
    else { key[0] = e->rng.key[0]; key[1] = e->rng.key[1]; }
    fn_reset(e, key);
  }
}

void tgo_reset(tgo_batch* b, const uint64_t* seeds, const uint8_t* mask) {
  for (int i = 0; i < b->batch; ++i)
    if (!mask || mask[i]) reset_one(b, i, seeds ? seeds[i] : 0, seeds != NULL);
}

void tgo_step(tgo_batch* b, const int32_t* actions, float* reward, uint8_t* terminated, uint8_t* lines,
              int autoreset, int nthreads) {
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
  for (int i = 0; i < b->batch; ++i) {
    env_t* e = &b->e[i];
    int ln = 0;
    float rw;
    if (autoreset == TG_AUTORESET_NEXT_STEP && e->pending_reset) {
      reset_one(b, i, 0, 0);
      reward[i] = 0.0f; terminated[i] = 0; lines[i] = 0;
      e->last_reward = 0.0;
      e->last_flags = 0;
      continue;
    }
    if (b->ruleset == TG_RULESET_NUMPY) { e->last_flags = 0; rw = (float)np_step(e, actions[i], &ln); }
    else { rw = fn_step(e, actions[i], &ln); e->last_reward = (double)rw; e->last_flags = 0; }
    reward[i] = rw;
    terminated[i] = (uint8_t)e->game_over;
    lines[i] = (uint8_t)ln;
    if (e->game_over) {
      if (autoreset == TG_AUTORESET_SAME_STEP) reset_one(b, i, 0, 0);
      else if (autoreset == TG_AUTORESET_NEXT_STEP) e->pending_reset = 1;
    }
  }
}

void tgo_grouped_step(tgo_batch* b, const int32_t* actions, float* reward, uint8_t* terminated,
                      uint8_t* lines, uint8_t* illegal, int terminate_on_illegal, int autoreset,
                      int nthreads) {
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
  for (int i = 0; i < b->batch; ++i) {
    env_t* e = &b->e[i];
    int ln = 0, term = 0, ill = 0;
    if (autoreset == TG_AUTORESET_NEXT_STEP && e->pending_reset) {
      reset_one(b, i, 0, 0);
      reward[i] = 0.0f; terminated[i] = 0; lines[i] = 0;
      e->last_reward = 0.0;
      e->last_flags = 0;
      if (illegal) illegal[i] = 0;
      continue;
    }
    reward[i] = grouped_step(e, actions[i], terminate_on_illegal, &ln, &term, &ill);
    terminated[i] = (uint8_t)term;
    lines[i] = (uint8_t)ln;
    if (illegal) illegal[i] = (uint8_t)ill;
    if (term) {
      if (autoreset == TG_AUTORESET_SAME_STEP) reset_one(b, i, 0, 0);
      else if (autoreset == TG_AUTORESET_NEXT_STEP) e->pending_reset = 1;
    }
  }
}

void tgo_grouped_enumerate(tgo_batch* b, uint8_t* legal_mask, uint8_t* boards, int nthreads) {
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
  for (int i = 0; i < b->batch; ++i) {
    const env_t* e = &b->e[i];
    size_t na = (size_t)(4 * e->W);
    grouped_enumerate(e, legal_mask + (size_t)i * na,
                      boards ? boards + (size_t)i * na * (size_t)(e->Hp * e->Wp) : NULL);
  }
}

/* per-placement feature tensors u8 [B][4W][W+3]: GroupedActionsObservations(observation_wrappers=[FeatureVectorObservation])
 * (wrappers/grouped.py:185-205: every placement board is wrapped with an all-zero active_tetromino_mask) */
void tgo_grouped_features(tgo_batch* b, uint8_t* feats, int nthreads) {
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
  for (int i = 0; i < b->batch; ++i) {
    const env_t* e = &b->e[i];
    const int NA = 4 * e->W, cells = e->Hp * e->Wp, NF = e->W + 3;
    uint8_t legal[4 * 32];
    uint8_t* boards = (uint8_t*)malloc((size_t)NA * cells);
    uint8_t* zero = (uint8_t*)calloc((size_t)cells, 1);
    grouped_enumerate(e, legal, boards);
    for (int a = 0; a < NA; ++a)
      tgo_feature_vector(boards + (size_t)a * cells, zero, e->W, e->H, e->P, feats + ((size_t)i * NA + a) * NF);
    free(boards);
    free(zero);
  }
}

/* FeatureVectorObservation of the current observation u8 [B][W+3] (wrappers/observation.py:236-278) */
void tgo_features(tgo_batch* b, uint8_t* feats, int nthreads) {
  (void)nthreads;
  for (int i = 0; i < b->batch; ++i) {
    const env_t* e = &b->e[i];
    uint8_t board[MAXCELLS], mask[MAXCELLS];
    np_obs_dict(e, board, mask, NULL, NULL);
    tgo_feature_vector(board, mask, e->W, e->H, e->P, feats + (size_t)i * (e->W + 3));
  }
}

void tgo_observe(tgo_batch* b, int obs_mode, void* out, int nthreads) {
  (void)nthreads;
  const env_t* e0 = &b->e[0];
  const int W = e0->W, H = e0->H, P = e0->P, Wp = e0->Wp, Hp = e0->Hp, Q = e0->qsize;
  const size_t B = (size_t)b->batch;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
  for (int i = 0; i < b->batch; ++i) {
    const env_t* e = &b->e[i];
    if (obs_mode == TG_OBS_BOARD) {
      uint8_t full[MAXCELLS];
      np_obs_dict(e, full, NULL, NULL, NULL);
      uint8_t* o = (uint8_t*)out + (size_t)i * (size_t)(H * W);
      for (int r = 0; r < H; ++r) memcpy(o + r * W, full + r * Wp + P, (size_t)W); /* :514-523 crop */
    } else if (obs_mode == TG_OBS_DICT) {
      uint8_t* base = (uint8_t*)out;
      size_t cells = (size_t)(Hp * Wp);
      np_obs_dict(e, base + (size_t)i * cells, base + B * cells + (size_t)i * cells,
                  base + 2 * B * cells + (size_t)i * (size_t)(P * P),
                  base + 2 * B * cells + B * (size_t)(P * P) + (size_t)i * (size_t)(P * P * Q));
    } else if (obs_mode == TG_OBS_FN) {
      fn_obs(e, (int8_t*)out + (size_t)i * (size_t)(H * W));
    } else if (obs_mode == TG_OBS_RGB) {
      /* wrappers/observation.py:38-74: stack = hstack(board, vstack(queue, ones, holder)); rgb = colors[stack];
       * colours = BASE_PIXELS + tetromino colours (envs/tetris.py:45-75) */
      static const uint8_t LUT[9][3] = {{0, 0, 0},   {128, 128, 128}, {0, 240, 240}, {240, 240, 0}, {160, 0, 240},
                                        {0, 240, 0}, {240, 0, 0},     {0, 0, 240},   {240, 160, 0}};
      uint8_t board[MAXCELLS], holder[16], queue[16 * 7];
      np_obs_dict(e, board, NULL, holder, queue);
      const int ML = P * (Q > 1 ? Q : 1), WS = Wp + ML;
      uint8_t* o = (uint8_t*)out + (size_t)i * (size_t)(Hp * WS * 3);
      for (int R = 0; R < Hp; ++R)
        for (int C = 0; C < WS; ++C) {
          int id;
          if (C < Wp) id = board[R * Wp + C];
          else {
            const int cc = C - Wp;
            if (R < P) id = cc < P * Q ? queue[R * (P * Q) + cc] : 1;
            else if (R >= Hp - P) id = cc < P ? holder[(R - (Hp - P)) * P + cc] : 1;
            else id = 1;
          }
          memcpy(o + (R * WS + C) * 3, LUT[id <= 8 ? id : 8], 3);
        }
    }
  }
}

void tgo_get_state(tgo_batch* b, uint8_t* board, tg_env_rec* rec, tg_rng_rec* rng) {
  for (int i = 0; i < b->batch; ++i) {
    const env_t* e = &b->e[i];
    if (board)
      for (int r = 0; r < e->H; ++r)
        memcpy(board + ((size_t)i * e->H + r) * e->W, e->board + r * e->Wp + e->P, (size_t)e->W);
    if (rec) {
      tg_env_rec* q = &rec[i];
      memset(q, 0, sizeof(*q));
      q->piece = (uint8_t)e->active.index;
      q->rot = (uint8_t)(b->ruleset == TG_RULESET_NUMPY ? e->active.rot : e->fn_rot);
      q->x = (uint8_t)e->x;
      q->y = (uint8_t)e->y;
      q->val = e->active.val;
      q->flags = (uint8_t)((e->game_over ? TG_F_GAME_OVER : 0) | (e->has_swapped ? TG_F_HAS_SWAPPED : 0) |
                           (e->pending_reset ? TG_F_PENDING_RESET : 0));
      q->holder_piece = e->holder_len ? (uint8_t)e->holder.index : TG_NONE;
      q->holder_rot = e->holder_len ? (uint8_t)e->holder.rot : 0;
      q->holder_val = e->holder_len ? e->holder.val : 0;
      q->bag_index = (uint8_t)e->bag_index;
      memcpy(q->bag, e->bag, 7);
      memcpy(q->queue, e->queue, 7);
      q->score = e->score;
    }
    if (rng) rng[i] = e->rng;
  }
}

void tgo_set_state(tgo_batch* b, const uint8_t* board, const tg_env_rec* rec, const tg_rng_rec* rng) {
  for (int i = 0; i < b->batch; ++i) {
    env_t* e = &b->e[i];
    if (board) {
      create_board(e);
      for (int r = 0; r < e->H; ++r)
        memcpy(e->board + r * e->Wp + e->P, board + ((size_t)i * e->H + r) * e->W, (size_t)e->W);
    }
    if (rec) {
      const tg_env_rec* q = &rec[i];
      e->active = make_tetromino(q->piece, q->rot, q->val);
      e->fn_rot = q->rot;
      e->x = q->x;
      e->y = q->y;
      e->game_over = (q->flags & TG_F_GAME_OVER) != 0;
      e->has_swapped = (q->flags & TG_F_HAS_SWAPPED) != 0;
      e->pending_reset = (q->flags & TG_F_PENDING_RESET) != 0;
      e->holder_len = q->holder_piece != TG_NONE;
      if (e->holder_len) e->holder = make_tetromino(q->holder_piece, q->holder_rot, q->holder_val);
      e->bag_index = q->bag_index;
      memcpy(e->bag, q->bag, 7);
      memcpy(e->queue, q->queue, 7);
      e->score = q->score;
    }
    if (rng) {
      e->rng = rng[i];
      e->rng_valid = 1;
    }
  }
}

void tgo_get_padded_boards(tgo_batch* b, uint8_t* out) {
  for (int i = 0; i < b->batch; ++i) {
    const env_t* e = &b->e[i];
    memcpy(out + (size_t)i * (size_t)(e->Hp * e->Wp), e->board, (size_t)(e->Hp * e->Wp));
  }
}

void tgo_set_padded_boards(tgo_batch* b, const uint8_t* in) {
  for (int i = 0; i < b->batch; ++i) {
    env_t* e = &b->e[i];
    memcpy(e->board, in + (size_t)i * (size_t)(e->Hp * e->Wp), (size_t)(e->Hp * e->Wp));
  }
}
