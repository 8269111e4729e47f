// tg_kernel_bits.h — fifth formulation of the step kernel (the default): rule logic on per-env COLUMN BITMASKS, one
// env per lane, in a workgroup of two specialised waves, with the bag shuffles produced ahead by other workgroups.
//
// Why (measured, profiles/r01/block1_*): at batch 65 536 every workgroup of a launch is resident at once, so the
// launch time is the critical path of ONE workgroup, not a throughput; 78 % of the wave cycles of the byte-probing
// kernels are waits on dependent LDS byte reads (collision probes, the step-wise drop loop, row scans) and on the
// PCG64 bag shuffle (128-bit multiplies in a rejection loop, behind a dependent HBM load of the generator).  Here:
//   * besides the uint8 playfield (the state the reference defines, envs/tetris.py:632-641) every env carries an
//     acceleration structure: one bitmask per playfield column (bit r = cell (r, c) is filled), W words per env in
//     HBM (40 B at 10x20), kept consistent by this kernel and rebuilt by bits_prepare_kernel after reset/set_state;
//   * LOGIC WAVE: ONE LDS round trip fetches the six columns around the piece; every collision test of the step
//     (tentative here/below, current here/below), the hard drop (the reference's step-wise loop == first set bit of
//     the "piece at row r collides" mask below y) and the lock are register ALU on those six words; full rows = AND
//     over the W columns (the reference's whole-board scan, envs/tetris.py:493-494), the spawn collision reads the
//     same registers; line-clear compaction of the masks is bit arithmetic (rare path);
//   * STREAM WAVE: the uint8 slab is only the payload: it streams HBM → LDS while the logic wave decides, receives
//     the locked cells / cleared rows / resets after one s_barrier, goes back to the board where it changed and
//     leaves as the observation (coalesced 16-byte accesses, nontemporal);
//   * RNG ROLE (trailing workgroups of the same launch, every third one-transition launch): numpy's
//     Generator.shuffle draws do not depend on the array being shuffled (Fisher-Yates positions only), so every env
//     owns a ring of four future bag permutations (7 nibbles each) produced ahead of their use; consuming one is a
//     byte permute (v_perm_b32).  The generator state the reference would hold (tg_get_state) is materialised into
//     tg_rng_rec on demand from the state stored with the last consumed ring entry;
//   * tg_step_k / tg_rollout run up to 16 transitions per launch with all of the above resident on chip in between.
// Round 2: publish before bookkeeping, stores as whole cache lines wherever the layout allows (board
// chunks by compare, records exchanged between lanes), the arguments the first loads need as preloaded scalars, and a
// build for instruction-level parallelism (one logic wave per SIMD is a dependent chain: Makefile flags).
// Round 3: where the generator state after an in-line shuffle lives is a field (`loc`) beside the consumed counter, and the
// in-line buffer written is never the one this launch's readers were pointed at (round 2 had a race there); the functional
// rule-set's jax key chain is looked ahead by the env's own logic wave behind its publish (protocol in front of the kernel);
// a wave issues one instruction per ~4.8 cycles whatever its ILP (tools/ubench/valu_issue.hip): instruction COUNT is the lever;
// exec-mask regions and branches cost 15-45 cycles beyond their instructions (tools/ubench/salu_issue.hip): the decision path is
// selects, predicated global stores are raw-buffer stores with the predicate in the offset; the roles end with s_waitcnt 0 +
// s_endpgm (a return joins the function exit and the wait-count pass then sees the other roles' loads pending at the logic
// wave's start); small and medium batches run two stream waves per workgroup (kFeatSplit, picked by the host).
// Bit-identical to the four earlier step kernels (test-only library) and to the oracle (tests/test_gpu_*.py).
#pragma once
#ifndef __HIPCC_RTC__
#include <type_traits>
#endif

#include "tg_lane.h"

// -DTG_PHASE_TIMING (tools/phase_timing_bits.py, tools/phase_timeline_bits.py; never the shipped build): eight time stamps per wave,
// written to KParams::dbg at the end.  s_memtime = shader cycles of THIS CU's counter (phases of one wave); with
// -DTG_STAMP_REALTIME the device-wide 100 MHz counter instead (absolute timeline across workgroups).
#ifdef TG_PHASE_TIMING
#ifdef TG_STAMP_REALTIME
#define TG_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define TG_NOW() __builtin_amdgcn_s_memtime()
#endif
#define TG_BITS_STAMP(i) tstamp[i] = TG_NOW()
#define TG_TIMING(...) __VA_ARGS__
#else
#define TG_BITS_STAMP(i) do {} while (0)
#define TG_TIMING(...)
#endif

namespace tg {

// per (piece, rot): four column nibbles — nibble c holds, as bits k, the filled cells (k, c) of the 4x4 box
// (the transpose of ShapeTable.mask)
struct ColTable {
  uint64_t cols[7];
};

constexpr ColTable make_col_table() {
  ColTable t{};
  const ShapeTable s = make_shape_table();
  for (int p = 0; p < 7; ++p) {
    uint64_t row = 0;
    for (int r = 0; r < 4; ++r) {
      uint64_t v = 0;
      for (int k = 0; k < 4; ++k)
        for (int c = 0; c < 4; ++c)
          if ((s.mask[p][r] >> (4 * k + c)) & 1) v |= 1ULL << (4 * c + k);
      row |= v << (16 * r);
    }
    t.cols[p] = row;
  }
  return t;
}

__constant__ ColTable c_colmasks = make_col_table();

// every column of every piece in every rotation is one vertical run of cells (what the drop below relies on)
constexpr bool cols_are_runs() {
  const ColTable t = make_col_table();
  for (int p = 0; p < 7; ++p)
    for (int i = 0; i < 16; ++i) {
      uint32_t n = (uint32_t)(t.cols[p] >> (4 * i)) & 0xFu;
      if (n == 0) continue;
      while ((n & 1u) == 0) n >>= 1;
      if ((n & (n + 1u)) != 0) return false;
    }
  return true;
}
static_assert(cols_are_runs(), "drop_row_of tests only the lowest cell of a piece column");

// cells of every piece at rotation 0 (16 bits each, c_cells' encoding) as two immediates: what a fresh episode shows
constexpr uint64_t cells_rot0(const int first) {
  const CellTable t = make_cell_table();
  uint64_t v = 0;
  for (int i = 0; i < 4 && first + i < 7; ++i) v |= (t.cells[first + i] & 0xFFFFull) << (16 * i);
  return v;
}
constexpr uint64_t kCells0Lo = cells_rot0(0), kCells0Hi = cells_rot0(4);

template <bool C, typename A, typename B> struct pick_t { typedef A type; };
template <typename A, typename B> struct pick_t<false, A, B> { typedef B type; };
template <int H>
using ColT = typename pick_t<(H + 6 <= 32), uint32_t, uint64_t>::type;

__device__ __forceinline__ int ctz_t(uint32_t v) { return (int)__builtin_ctz(v); }
__device__ __forceinline__ int ctz_t(uint64_t v) { return (int)__builtin_ctzll(v); }
__device__ __forceinline__ int popc_t(uint32_t v) { return (int)__popc(v); }
__device__ __forceinline__ int popc_t(uint64_t v) { return (int)__popcll(v); }

// column-mask words per env in HBM: the masks travel as 8-byte pairs, so an odd number of 32-bit masks is padded by one
template <int W, int H>
constexpr int bits_cols_words() { return sizeof(ColT<H>) == 4 ? W + (W & 1) : W; }
// bytes per env of the column-mask buffer
template <int W, int H>
constexpr size_t bits_cols_bytes() { return (size_t)bits_cols_words<W, H>() * sizeof(ColT<H>); }

// Envs per workgroup.  ONE STREAM WAVE carries at most EPS0 envs — its slab block has to fit the staging registers: 64 envs up
// to 320 cells, 32 up to 640, 16 above — and by default a workgroup has one stream wave, so the LOGIC wave of a large board runs
// with idle lanes.  A logic wave with idle lanes costs the SIMD as many issue slots as a full one: a 10x20 launch at the byte
// volume of config 5 (52 Mi cells) takes 22.4 / 26.0 / 36.1 us with 64 / 32 / 16 envs per workgroup.
// (Twice the envs per workgroup and TWO stream waves for boards above 320 cells — measured and not adopted: config 5
// 21.9 -> 21.2 us only, and the dict / RGB staging buffers no longer fit beside the slabs.  profiles/r04/ab_envs_per_workgroup.log)
constexpr int bits_eps0(const int W, const int H) { return W * H <= 320 ? 64 : (W * H <= 640 ? 32 : 16); }
constexpr int bits_lds_bytes_max(const int W, const int H, const int epw) {  // (BitsGeo::BYTES_FNX, the largest plain layout)
  const int slabs = ((epw * W * H + 15) / 16) * 16, cst = (W + 2) | 1, tsz = (H + 6 <= 32) ? 4 : 8;
  return ((slabs + 320 + 2 * epw * 16 + epw * cst * tsz + 15) / 16) * 16 + 2 * epw * 8 + 128 * 12;
}
constexpr int bits_epw(const int W, const int H) { return bits_eps0(W, H); }
constexpr int bits_nsw0(const int W, const int H) {  // stream waves a workgroup of this board size has at least
  return bits_epw(W, H) / bits_eps0(W, H);
}
// dynamic LDS of one workgroup of step_kernel_bits.  NSEQ > 1 (kFeatSeq, round 6): the workgroup serves NSEQ x bits_epw envs — the logic
// wave decides them all at once, one env per lane — while its stream wave walks them in NSEQ groups through ONE slab block in LDS.
template <int W, int H, int NSEQ = 1>
struct BitsGeo {
  using T = ColT<H>;
  static constexpr int S = W * H;
  static constexpr int EPW = bits_epw(W, H) * NSEQ;
  static constexpr int NSW0 = bits_nsw0(W, H);
  static constexpr int SLABS = ((EPW / NSEQ * S + 15) / 16) * 16;
  static constexpr int NCH = SLABS / 16;
  static constexpr int NIT = (NCH + 63) / 64;
  static constexpr int OFF_TAB = SLABS;             // cell table, column table, cell table (logic wave's copy), row-mask table, colour table: 5 x 64 B
  static constexpr int OFF_POST = OFF_TAB + 320;    // uint4 [2][EPW]: what the stream wave has to do per env (double-buffered by transition parity)
  static constexpr int OFF_COLS = OFF_POST + 2 * EPW * 16;  // T [EPW][CST]: [0] bedrock, [1..W] playfield columns, [W+1] bedrock
  static constexpr int CST = (W + 2) | 1;           // odd stride: conflict-free for same-index accesses of the 64 lanes
  static constexpr int BYTES = ((OFF_COLS + EPW * CST * (int)sizeof(T) + 15) / 16) * 16;
  // TG_OBS_DICT packed in the kernel: two staging buffers of GD padded boards behind BYTES (stream wave: board, logic
  // wave: active_tetromino_mask)
  static constexpr int PC = (W + 8) * (H + 4);
  static constexpr int GD = 16;
  static constexpr int OFF_DICT = BYTES;
  static constexpr int OFF_OLDPW = BYTES + 2 * GD * PC;  // uint32 [EPW]: the box TG_BUF_OBS's mask array shows per env
  static constexpr int BYTES_DICT = OFF_OLDPW + EPW * 4;
  static constexpr int OFF_DICT2 = BYTES_DICT;             // (kFeatSplit: the second stream wave's board staging buffer)
  static constexpr int BYTES_DICT_SPLIT = BYTES_DICT + GD * PC;
  static constexpr bool DICT_OK = NSEQ == 1 && NSW0 == 1 && PC % 16 == 0 && EPW % GD == 0 && H % 2 == 0 && W % 2 == 0;  // (row pairs as whole dwords)
  // TG_OBS_RGB packed in the kernel (queue_size 4): uint4 post2[EPW] {queue bytes | holder} published by the logic wave,
  // then one id-image staging buffer of GR envs per wave.  Image row = padded board row | 4 queue pieces x 4 columns.
  static constexpr int RS = W + 8 + 16;
  static constexpr int NPX = (H + 4) * RS;
  static constexpr int IDB = ((NPX + 15) / 16) * 16;
  static constexpr int GR = 4;
  static constexpr int OFF_RGB = BYTES;
  static constexpr int BYTES_RGB = BYTES + EPW * 16 + 2 * GR * IDB;
  static constexpr int BYTES_RGB_X = BYTES_RGB + 2 * GR * IDB;  // (kFeatMaskWave with kFeatRgb: two helper waves with a staging buffer each)
  static constexpr int BYTES_RGB_X6 = BYTES_RGB + 6 * GR * IDB; // (... | kFeatMany: six)
  static constexpr bool RGB_OK = NSEQ == 1 && NSW0 == 1 && NPX % 4 == 0 && H % 2 == 0 && W % 2 == 0 && (EPW / 2) % GR == 0;
  // TG_FLAG_FINAL_OBS (observation BOARD / FN only, never together with the dict staging): uint2 [2][EPW] behind BYTES —
  // the terminal piece of every env that ends its episode in this transition
  static constexpr int OFF_FIN = BYTES;
  static constexpr int BYTES_FIN = BYTES + 2 * EPW * 8;
  // FUNCTIONAL rule-set (never together with the dict / RGB staging): the logic wave's exchange area for the sorts of the
  // key-chain look-ahead — uint2 keys [128] | uint32 permutations [128] — behind the FINAL_OBS records
  static constexpr int OFF_FNX = BYTES_FIN;
  static constexpr int BYTES_FNX = BYTES_FIN + 128 * 8 + 128 * 4;
  // ... directly behind BYTES in the instantiations without the FINAL_OBS records (round 5): at 10x20 that is 20 032 instead of 21 056
  // bytes per workgroup — EIGHT workgroups per CU instead of seven fit the 160 KB, i.e. a one-transition launch of 131 072 envs (2 048
  // workgroups of two waves, compiled for four waves per SIMD) is resident in one round: 14.4 -> 11.x us (profiles/r05/batch_sweep.log)
  static constexpr int OFF_FNX0 = BYTES;
  static constexpr int BYTES_FNX0 = BYTES + 128 * 8 + 128 * 4;
  static_assert(NSEQ > 1 || BYTES_FNX == bits_lds_bytes_max(W, H, EPW), "bits_lds_bytes_max restates this layout");
  static_assert(NSW0 == 1 || BYTES_FNX <= 65536, "two stream waves' worth of envs only while the workgroup's LDS stays within 64 KiB");
  static constexpr int WC = bits_cols_words<W, H>();  // column-mask words per env in HBM
  // two stream waves per workgroup (each owns half of the envs and their slabs: no exchange between them) when the halves
  // are whole 16-byte chunks
  static constexpr bool SPLIT_OK = NSEQ == 1 && NSW0 == 1 && EPW == 64 && ((EPW / 2) * S) % 16 == 0;
};


// numpy PCG64 generator of one lane (NUMPY rule-set): 128-bit state, stream increment, buffered 32-bit half
struct BitsRng {
  uint64_t s_lo, s_hi, i_lo, i_hi;
  uint32_t uinteger, has32;
  bool loaded;  // RNG role: the frontier state has been fetched (later entries of a launch continue from registers)
};

// Fisher-Yates positions of Generator.shuffle on a 7-array (numpy/random/_generator.pyx shuffle → random_interval,
// masked rejection on 32-bit draws; components/tetromino_randomizer.py:42-46, 84) for the lanes in `need`, run on the
// identity: the result P (7 nibbles) is the permutation with  shuffled[k] = original[P[k]]  for ANY original array.
// Same vector loop as lane_shuffle7 (legacy/tg_kernel_lanes.h): one PCG64 step + two draw attempts per trip.
__device__ __forceinline__ void bits_shuffle_perm(BitsRng& g, uint32_t& sig, const bool need) {
  uint64_t s_lo = g.s_lo, s_hi = g.s_hi;
  uint32_t uinteger = g.uinteger, has32 = g.has32;
  uint32_t nb = 0x6543210u;
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
  {
    const bool on = run && has32 != 0;  // a buffered high half is consumed first
    attempt(uinteger, on);
    if (on) has32 = 0;
    run = run && i >= 1;
  }
  while (__any(run)) {
    uint64_t n_lo, n_hi, out;
    lane_pcg_step(s_lo, s_hi, g.i_lo, g.i_hi, n_lo, n_hi, out);
    if (run) { s_lo = n_lo; s_hi = n_hi; uinteger = (uint32_t)(out >> 32); has32 = 1; }
    attempt((uint32_t)out, run);
    run = run && i >= 1;
    attempt(uinteger, run);
    if (run) has32 = 0;
    run = run && i >= 1;
  }
  if (need) {
    g.s_lo = s_lo; g.s_hi = s_hi;
    g.uinteger = uinteger; g.has32 = has32;
    sig = nb;
  }
}

// shuffled[k] = src[P[k]]  on 7-byte arrays packed in a 64-bit word: the nibbles of P become the byte selectors of
// two v_perm_b32 over the pool {src.hi : src.lo} (selector 0..3 = low dword bytes, 4..7 = high dword bytes, 0x0C = 0)
__device__ __forceinline__ uint64_t bits_apply_perm(const uint32_t sig, const uint64_t src) {
  const uint32_t sel_lo = (sig & 0xFu) | ((sig & 0xF0u) << 4) | ((sig & 0xF00u) << 8) | ((sig & 0xF000u) << 12);
  const uint32_t s2 = sig >> 16;
  const uint32_t sel_hi = (s2 & 0xFu) | ((s2 & 0xF0u) << 4) | ((s2 & 0xF00u) << 8) | 0x0C000000u;
  const uint32_t lo = (uint32_t)src, hi = (uint32_t)(src >> 32);
  return (uint64_t)__builtin_amdgcn_perm(hi, lo, sel_lo) | ((uint64_t)__builtin_amdgcn_perm(hi, lo, sel_hi) << 32);
}

// The bag permutations of an env are produced ahead of their use by separate RNG workgroups of the same launch
// (role split by blockIdx) and consumed by the logic wave: a four-slot ring per env with monotonic counters (mod 128),
// T = permutations produced (owned by the RNG role), H = permutations consumed (owned by the logic wave).  Entry i
// lives in slot i & 3.  Everything one role reads from the other is double-buffered by launch parity (`cur`), so
// both roles work on the snapshot of the previous launch boundary and never on each other's stores of this launch
// (L2 is per XCD; cross-workgroup visibility inside a launch is not guaranteed).
//
// The generator state the reference would hold (tg_get_state, pinned against numpy) is NOT touched by a transition:
// `ahead` keeps the state after every ring entry, the RNG role never overwrites the slot of the last consumed entry
// (it looks ahead kRing - 1 entries), so "the generator now" = ahead[(H - 1) & 3] — read only by
// bits_materialize_rng_kernel (before tg_get_state / tg_reset / tg_set_state / a fetch of TG_BUF_RNG) and by the two
// rare paths below.  No load, no wait and no store of generator state on the transition's path.
//   ring exhausted → the logic wave shuffles in line, starting from that state.  WHERE "the state after entry H - 1" lives
//   is the `loc` field beside H: kLocAhead (the slot above), kLocRec (tg_rng_rec: after bits_prepare_kernel — a step
//   launch never writes tg_rng_rec), kLocBuf0 / kLocBuf1 (one of the two in-line buffers `inl`).  The state after an
//   in-line shuffle goes to the in-line buffer that nobody reads in this launch: both roles read the location named by
//   the snapshot of the previous launch boundary, so the writer takes the other buffer (kLocBuf0 unless the snapshot
//   names it).  The RNG role then resynchronises from there (T := H).  Round 2 kept that state in tg_rng_rec itself,
//   which the RNG role of the SAME launch may read after the logic wave's store once the grid exceeds residency.
//   `ahead` is written by the RNG role only.
struct BitsParams {
  void* cols;      // ColT [B][W]
  uint4* ahead;    // [B][4 slots][2]: generator state AFTER the slot's shuffle {state_lo, state_hi} {uinteger, has32, 0, 0}
  // ring [2 parities][B][2 x u64]: word 0 = permutations of slots 0 | 1 << 28 (7 nibbles each) | T << 56, word 1 = slots
  // 2 | 3 << 28.  The launch reads one parity and its RNG role writes the other (it flips only with launches that carry
  // the RNG role); the host passes both as pointers — the parity arithmetic stood between the logic wave's first loads.
  const uint4* ring_rd;
  uint4* ring_wr;
  // hbuf [2 parities][B]: H (bits 0-7) | loc << 8 (where the state after entry H - 1 is: kLoc*); read parity / the
  // parity the logic wave writes (flips every launch).  FUNCTIONAL: the same storage is fnw [B] u64 (no parities)
  const uint32_t* hbuf_rd;
  uint32_t* hbuf_wr;
  uint4* inl;      // [2 buffers][B][2]: generator state after an in-line shuffle (layout of an `ahead` slot)
  int nrb;         // number of RNG workgroups (kRngEnvs envs each)
  int fill;          // entries the RNG role keeps ahead in a ring (kRing - 1; tests lower it to exercise the consumer's in-line shuffle)
};

// native 16-byte vector for the slab stream: HIP's uint4 is a struct whose copies become memcpy intrinsics, which
// kept the staging array in scratch memory (and made the kernel wait for every slab load right after issuing it)
constexpr int kMaxFuse = 16;  // transitions per launch (tg_config.fuse): the action bytes of a launch travel in one or two 64-bit words
constexpr int kRing = 4;  // slots per permutation ring: kRing - 1 entries ahead + the last consumed one
__device__ __forceinline__ int ring_diff(const int a, const int b) { return ((a - b) << 24) >> 24; }  // counters are mod 256
// where the generator state (FUNCTIONAL: the key) after the last consumed entry lives (`loc` field of hbuf)
constexpr int kLocAhead = 0, kLocRec = 1, kLocBuf0 = 2, kLocBuf1 = 3;  // (code tests `loc >= kLocBuf0`)
constexpr int kRngEnvs = 1024;  // envs scanned by one RNG-role workgroup (128 threads x 8)

__device__ __forceinline__ uint32_t ring_get(const uint64_t w0, const uint64_t w1, const int slot) {
  const uint64_t w = (slot & 2) ? w1 : w0;
  return (uint32_t)(w >> (28 * (slot & 1))) & 0x0FFFFFFFu;
}

// ---- FUNCTIONAL rule-set: look-ahead along the jax key chain (envs/tetris_fn.py:370-413, 318-367) ----
// With f0(k) = split(k)[0] = threefry(k, (0, 0)), f1(k) = split(k)[1] = threefry(k, (0, 1)) and
// sort7(s) = the argsort of the seven sort keys threefry(s, (0, j)).x ^ .y   (permutation(k) == sort7(f1(k))):
//   a lock replaces the key c by f0(c) (:399) and, with the 7-bag used up, deals permutation(f1(c)) (functional/queue.py:59-67);
//   a reset deals permutation(f0(c)) and continues with the key f1(c) (:318-367).
// Every env sits at a position n of a key chain c_n, c_{n+1} = f0(c_n), with R_m = f1(c_m):
//   a lock at n that refills     deals Q0 = sort7(f1(R_n))
//   a reset at n                 deals S1 = sort7(R_{n+1}), new key R_n            (NEXT_STEP autoreset: the step after the end)
//   a lock at n + a reset        deals S2 = sort7(R_{n+2}), new key R_{n+1}        (SAME_STEP autoreset)
// None of this depends on the game, so it is computed AHEAD, by the env's own logic wave, behind its publish (beside the
// stream wave's tail), for the lanes whose env consumed something — one threefry level (three independent blocks per
// lane) plus one sort per lock, three levels per reset; the seven blocks of every sort the wave owes are spread over
// LANES (8 lanes per sort, 16 sorts per trip).  In front of the publish a transition only picks 3-bit pieces out of one
// 64-bit word.  Round 2 ran threefry in line on every lock (10.3 us per launch, 2.6 us of it in front of the publish); a
// look-ahead produced by the RNG role (as the NUMPY bag permutations are) reached 8.3 us: its waves were the launch's tail.
//   fnw  [B] u64: S1 | S2 << 21 | Q0 << 42   (seven 3-bit pieces each; Q0 is meaningful when the bag index is 7)
//   fnla [B] 128 B: c[4] (c_m in slot m & 3 for m = n+1 .. n+3) | r[4] (R_m in slot m & 3 for m = n .. n+2) | unused;
//   n mod 4 lives in tg_env_rec.reserved (device-private), the key c_n in tg_rng_rec.key as the reference holds it.
// bits_prepare_kernel builds all of it from the key after tg_reset / tg_set_state.
// PARITY UNPINNED like every jax-dependent piece of this rule-set; engine == oracle is what tests check.
constexpr uint32_t kFnNone = 0x1FFFFFu;  // a 21-bit bag field that has not been sorted (no permutation of 0..6 holds a 7)
__device__ __forceinline__ uint32_t fn_pack21(const uint64_t bag) {  // 7 bytes → seven 3-bit fields
  uint32_t p = 0;
#pragma unroll
  for (int k = 0; k < 7; ++k) p |= ((uint32_t)(bag >> (8 * k)) & 7u) << (3 * k);
  return p;
}
__device__ __forceinline__ uint64_t fn_expand21(const uint32_t p) {  // seven 3-bit fields → 7 bytes
  auto four = [](const uint32_t x) -> uint32_t {  // fields at bits 0, 3, 6, 9 → bytes 0..3
    const uint32_t t = (x & 0x3Fu) | ((x & 0xFC0u) << 10);
    return (t & 0x00070007u) | ((t & 0x00380038u) << 5);
  };
  return (uint64_t)four(p & 0xFFFu) | ((uint64_t)four((p >> 12) & 0x1FFu) << 32);
}
// sort7 of one lane (bits_prepare_kernel; the step kernel spreads the seven blocks over lanes); n = the bag length (EnvConfig.queue_size,
// 1..7: functional/queue.py:35 permutes arange(queue_size)) — fields n..6 of the result are 0
__device__ __forceinline__ uint32_t fn_sort7(const uint32_t s0, const uint32_t s1, const int n) {
  uint32_t bits[7];
#pragma unroll
  for (uint32_t i = 0; i < 7; ++i) {
    uint32_t a, b;
    threefry2x32(s0, s1, 0u, i, a, b);
    bits[i] = a ^ b;
  }
  uint32_t p = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) rank += (j < n && (bits[j] < bits[i] || (bits[j] == bits[i] && j < i))) ? 1 : 0;
    p |= i < n ? (uint32_t)i << (3 * rank) : 0u;
  }
  return p;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 __attribute__((aligned(8))) u32x4_a8;  // 16-byte access at 8-byte alignment (env slabs: S % 8 == 0)

// streaming stores: L2 is per XCD and written back at the end of every kernel; data no wave of THIS launch reads again
// (observations, outputs) leaves with the nontemporal hint so that it drains while the kernel still computes
template <typename V>
__device__ __forceinline__ void st_stream(V* dst, const V v) {
  __builtin_nontemporal_store(v, dst);
}

// the two padded arrays of the dict observation (62 MB per transition at 65 536 envs)
template <typename V>
__device__ __forceinline__ void st_dict(V* dst, const V v) {
  __builtin_nontemporal_store(v, dst);
}
// ... of which the mask array (28 MB, almost all zeros, the same lines every transition) leaves with PLAIN stores since round 4: with
// the mask wave 15.5 -> 14.3 us per transition at 65 536 envs, 30.2 -> 26.1 at 131 072, 54.8 -> 48.5 at 262 144, 13.9 -> 10.1 per
// transition of a 16-transition launch; the board array plain as well: 14.6 / 27.4 / 60.3 / 9.8 (profiles/r04/ab_dict_mask_wave.log)
template <typename V>
__device__ __forceinline__ void st_dict_mask(V* dst, const V v) {
  *dst = v;
}

// state that the NEXT launch reads again (board, records, rings).  TG_STATE_ST: 0 = plain stores (dirty in the XCD's L2
// until the end-of-kernel write-back), 1 = nontemporal (default: measured -0.45 us per launch with the chunk-major
// write-back, profiles/r02 — the dirty lines leave while the kernel still runs, the boundary to the next launch gets
// shorter), 2 = write-through (sc1; measured +0.3 us).  The column masks stay plain 8-byte stores (nontemporal: no gain).
constexpr int TG_STATE_ST = 1;
template <typename V>
__device__ __forceinline__ void st_state(V* dst, const V v) {
  static_assert(sizeof(V) == 16 || sizeof(V) == 8, "state stores are 8 or 16 bytes");
  if constexpr (TG_STATE_ST == 1) {
    __builtin_nontemporal_store(v, dst);
  } else if constexpr (TG_STATE_ST == 2) {
    if constexpr (sizeof(V) == 16) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
  } else {
    *dst = v;
  }
}
#define TG_ST_STATE(ptr, val) st_state(ptr, val)
// A PREDICATED 16-byte state store without an exec-mask region: a raw-buffer store whose offset lies outside the
// descriptor's num_records is dropped by the hardware.  `if (c) store` costs a wave s_and_saveexec + s_cbranch_execz +
// s_or exec around every store (the compiler keeps the skip branch for memory instructions): 13 of them per lane in the
// board write-back = 1 416 cycles of issue against 865 this way (tools/ubench/buffer_store_oob.hip, profiles/r03/).
// `base` = a wave-uniform address, `bytes` what may be written behind it (< 2 GiB), `off` the lane's byte offset.
// (The same idea for the piece stamps in LDS — cells that are not drawn go to a dump dword of the lane — is SLOWER, 6.39 ->
// 6.54 us: few lanes draw, and an LDS store of 64 active lanes costs more than the exec-mask region around one of 10.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t st_window(void* base, const uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);  // (gfx9 raw buffer: 32-bit data format, no swizzle)
}
__device__ __forceinline__ void st_state_if(const __amdgpu_buffer_rsrc_t win, const uint32_t off, const u32x4 v, const bool on) {
  __builtin_amdgcn_raw_buffer_store_b128(v, win, on ? off : 0x80000000u, 0, TG_STATE_ST == 1 ? 2 : 0);  // (aux 2 = nontemporal)
}
// ... and the 8-byte form (plain): the sparse per-lane stores of the functional look-ahead
__device__ __forceinline__ void st_b64_if(const __amdgpu_buffer_rsrc_t win, const uint32_t off, const u32x2 v, const bool on) {
  __builtin_amdgcn_raw_buffer_store_b64(v, win, on ? off : 0x80000000u, 0, 0);
}
#define TG_ST_COLS(ptr, val) (*(ptr) = (val))

// FEAT selects what an instantiation carries (the optional features cost registers and instructions in the hot loops
// even when switched off at run time: measured +4 % on the one-transition launch, +18 % on fused rollouts):
//   kFeatMulti  p.nsteps transitions per launch (tg_step_k / tg_rollout); without it the loop collapses (tg_step)
//   kFeatTraj   trajectory outputs (obs_traj / out_traj / act_traj)
//   kFeatDict   TG_OBS_DICT packed by the two waves
//   kFeatSample actions drawn on the device (grouped action set)
//   kFeatFinal  TG_FLAG_FINAL_OBS: the terminal observation of a same-step-autoreset env goes to TG_BUF_FINAL_OBS
//   kFeatRgb    TG_OBS_RGB (RgbObservation, wrappers/observation.py:38-74) packed by the two waves
// Profiles list the kinds of launches separately.
constexpr int kFeatMulti = 1, kFeatTraj = 2, kFeatDict = 4, kFeatSample = 8, kFeatFinal = 16, kFeatRgb = 32;
// kFeatWide: the same code compiled for fewer waves per SIMD (TG_WIDE_WAVES: more VGPRs) — picked by the host when the whole launch is
// resident at that occupancy anyway (FUNCTIONAL rule-set, plain launches: tg_engine.hip bits_feat_rt)
constexpr int kFeatWide = 64;
// kFeatSplit: TWO stream waves per step workgroup (192 threads).  The uint8 payload work is proportional to bytes, not to
// envs: split over two waves it is the same number of instructions in half the time per wave.  Picked by the host while
// every workgroup of the launch is still resident with three waves each (up to ~40 000 envs): 4.7 -> 4.3 us at 8 192 envs,
// 5.7 -> 5.3 at 32 768; at 65 536 the 3-wave workgroups no longer fit beside the RNG role's and the launch is slower
// (profiles/r03/ab_stream_waves.log).  FUNCTIONAL: no gain in round 3, when its logic wave was the long pole (800 instructions
// of look-ahead behind the publish); with that halved twice (round 4) the stream wave's tail — the slab block through the
// {-1, 0, 1} byte map — is what the launch waits for, and two stream waves take 7.53 -> 6.92 us per launch at 65 536 envs,
// 6.33 -> 5.75 at 32 768, 5.25 -> 4.64 at 4 096, 2.95 -> 2.88 per fused transition (profiles/r04/ab_fn_split.log): picked while
// the launch is resident at three waves per SIMD, instead of the kFeatWide build.  Not for the instantiations whose stream
// wave shares staging buffers with the logic wave.
constexpr int kFeatSplit = 128;
// kFeatPacked (round 4): the board travels between HBM and the chip as NIBBLES — cells are 0 or a tetromino value 2..8
// (envs/tetris.py:632-641), two per byte — half the bytes of the board read and of its write-back.  `hot_board` is then the
// handle's packed board (device-private, like the column masks; the uint8 TG_BUF_BOARD is materialised from it on demand,
// like tg_rng_rec); the stream wave expands a packed 16-byte chunk into two chunks of the uint8 slab block in LDS
// (cell 2j / 2j + 1 of packed byte j: with an even slab size the map packed byte -> LDS bytes is linear over the whole
// block) and packs on the way back; everything in between works on the uint8 slabs as before.  Picked by the host for the
// working sets that do not fit the Infinity Cache (config 5: 20x40; batches beyond ~100k envs at 10x20), where the launch
// is bound by bytes: ~250 more instructions in the stream wave buy 17-19 % fewer bytes.  Instantiations without dict / RGB /
// final-observation / split.  FUNCTIONAL: a lock is `board += id` (functional/core.py:103-121), which stays a nibble in play
// (free cells + ids 2..8) but not for a piece uploaded on top of filled cells: the stamp notices the sum leaving the range and
// raises KParams::nib_over, which the engine reports at its next wait.
constexpr int kFeatPacked = 256;
template <int W, int H>
constexpr bool bits_packed(const int feat) {
  return (feat & kFeatPacked) != 0 && (BitsGeo<W, H>::EPW * W * H) % 32 == 0 && (W * H) % 2 == 0 &&
         !(feat & (kFeatDict | kFeatRgb | kFeatFinal | kFeatSplit | kFeatWide));
}
// kFeatSeq (round 6; boards above 640 cells — 16 envs per stream wave —, packed, one transition per launch: config 5).  Those launches
// were residency-bound: a workgroup of 16 envs holds two waves of 128 VGPRs, so EIGHT workgroups = 128 envs fit a CU, 65 536 envs need
// two rounds, and the second round pays the whole latency chain (first loads -> logic -> publish -> slab work) again: +3.7 us at the
// ninth workgroup per CU (profiles/r05/config5_batch_sweep.log) — while the logic wave ran with 16 of its 64 lanes.  Here a workgroup
// serves 32 envs: the logic wave decides them in ONE pass (32 lanes live, half the logic instructions per env), the stream wave
// requests BOTH 16-env slab groups up front (packed: 2 x 28 registers), then expands / applies / writes back / streams out one group
// after the other through the same 12.8 KB of LDS — no second barrier, the second group's data has arrived long before its turn.  LDS
// per workgroup 20 032 B -> eight per CU = 256 envs: the 2 048 workgroups of 65 536 envs are resident in one round.
constexpr int kFeatSeq = 2048;
template <int W, int H>
constexpr int bits_nseq(const int feat) {
  return ((feat & kFeatSeq) && bits_packed<W, H>(feat) && !(feat & kFeatMulti) && bits_eps0(W, H) == 16) ? 2 : 1;
}
// packed 16-byte chunk (32 cells) -> the two 16-byte chunks of uint8 cells, and back (values < 16)
__device__ __forceinline__ void nib_expand(const u32x4 pk, u32x4& a, u32x4& b) {
  auto lo4 = [](const uint32_t d) -> uint32_t {  // cells 0..3 of the dword's eight: lo.b0 hi.b0 lo.b1 hi.b1
    return __builtin_amdgcn_perm((d >> 4) & 0x0F0F0F0Fu, d & 0x0F0F0F0Fu, 0x05010400u);
  };
  auto hi4 = [](const uint32_t d) -> uint32_t {  // cells 4..7: lo.b2 hi.b2 lo.b3 hi.b3
    return __builtin_amdgcn_perm((d >> 4) & 0x0F0F0F0Fu, d & 0x0F0F0F0Fu, 0x07030602u);
  };
  a = u32x4{lo4(pk.x), hi4(pk.x), lo4(pk.y), hi4(pk.y)};
  b = u32x4{lo4(pk.z), hi4(pk.z), lo4(pk.w), hi4(pk.w)};
}
__device__ __forceinline__ u32x4 nib_pack(const u32x4 a, const u32x4 b) {
  auto one = [](const uint32_t p, const uint32_t q) -> uint32_t {  // 8 cells (two dwords) -> one dword
    const uint32_t even = __builtin_amdgcn_perm(q, p, 0x06040200u), odd = __builtin_amdgcn_perm(q, p, 0x07050301u);
    return (even & 0x0F0F0F0Fu) | ((odd & 0x0F0F0F0Fu) << 4);
  };
  return u32x4{one(a.x, a.y), one(a.z, a.w), one(b.x, b.y), one(b.z, b.w)};
}
template <int W, int H>
constexpr int bits_stream_waves(const int feat) {
  if (BitsGeo<W, H>::NSW0 > 1) return BitsGeo<W, H>::NSW0;  // (boards above 320 cells: every instantiation)
  const bool split = (feat & kFeatSplit) && BitsGeo<W, H>::SPLIT_OK && !(feat & (kFeatSample | kFeatFinal | kFeatRgb | kFeatWide)) &&
                     (!(feat & kFeatDict) || (BitsGeo<W, H>::DICT_OK && (BitsGeo<W, H>::EPW / 2) % BitsGeo<W, H>::GD == 0));
  // (four stream waves of 16 envs each for launches of one workgroup per CU: measured 2-3 % at 4 096 envs, not kept —
  // profiles/r04/ab_split4.log)
  return split ? 2 : 1;
}
// kFeatMaskWave (round 4, dict instantiations): a THIRD wave per workgroup that does nothing but obs["active_tetromino_mask"]
// (28 of the 62 MB a dict transition stores at 65 536 envs).  The array needs only the post records — no slab, no lock stamp —
// so it runs beside the stream wave's board pack from the publish on; in the logic wave, behind the bookkeeping, it was the
// launch's tail (phase stamps: logic wave 26 000 cycles against the stream wave's 19 300; profiles/r04/ab_dict_mask_wave.log).
constexpr int kFeatMaskWave = 512;
template <int W, int H>
constexpr bool bits_mask_wave(const int feat) {
  return (feat & kFeatMaskWave) && (feat & kFeatDict) && BitsGeo<W, H>::DICT_OK && !(feat & kFeatSplit);
}
// ... and with kFeatRgb the same bit gives the one-transition RGB instantiation TWO helper waves: the logic wave, the stream wave and
// the helpers pack a quarter of the workgroup's images each.  Below ~32 768 envs the RgbObservation launch is the chain of its packing
// waves (18.6 us at 4 096 envs, of which 16 are two waves packing 32 images each): profiles/r04/ab_rgb_helper_waves.log.
template <int W, int H>
constexpr bool bits_rgb_helpers(const int feat) {
  return (feat & kFeatMaskWave) && (feat & kFeatRgb) && BitsGeo<W, H>::RGB_OK && (BitsGeo<W, H>::EPW / 4) % BitsGeo<W, H>::GR == 0 &&
         !(feat & kFeatSplit) && BitsGeo<W, H>::BYTES_RGB_X <= 65536;
}
// (Board helper waves for the DICT observation — the four groups of 16 padded boards packed by four waves instead of one after the
// other by the stream wave, started by a word in LDS — were built and measured at the end of round 4: 6.7 -> 6.5 us at 4 096 envs,
// and factoring the group body out for them cost the default kernel six spilled registers (14.3 -> 14.7 us at 65 536 envs): not kept.)
// kFeatMany (with kFeatMaskWave | kFeatRgb): SIX helper waves — eight waves pack an eighth of the workgroup's images each; for the
// launches with one workgroup per CU at most (up to 16 384 envs at 10x20), where the packing chain is all there is
constexpr int kFeatMany = 1024;
template <int W, int H>
constexpr int bits_rgb_helper_count(const int feat) {
  if (!bits_rgb_helpers<W, H>(feat)) return 0;
  return ((feat & kFeatMany) && (BitsGeo<W, H>::EPW / 8) % BitsGeo<W, H>::GR == 0 && BitsGeo<W, H>::BYTES_RGB_X6 <= 65536) ? 6 : 2;
}
template <int W, int H>
constexpr int bits_wg_waves(const int feat) {
  return 1 + bits_stream_waves<W, H>(feat) + (bits_mask_wave<W, H>(feat) ? 1 : 0) + bits_rgb_helper_count<W, H>(feat);
}

// bytes: 0xFF (active) stays -1, other non-zero → 1  (the functional observation, envs/tetris_fn.py:137-158).  Board cells
// are 0 or a tetromino value 2..8 (tg_set_state rejects bytes >= 0x80 for this rule-set), so the only byte with its top bit
// set is the drawn piece: 7 instructions per dword instead of 11 — at 52 dwords per lane this conversion is a fifth of
// the stream wave's instructions, and the launch is bound by the instructions its two waves issue on their SIMD.
__device__ __forceinline__ uint32_t bits_bin1(const uint32_t x) {
  const uint32_t a = x & 0x80808080u;                                              // the active piece's bytes
  const uint32_t m = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) >> 7) & 0x01010101u;       // 1 where the low seven bits are non-zero
  return m | a | (a - (a >> 7));                                                   // 0x80 → 0xFF
}

// The same map for four dwords in 17 instructions instead of 28: per byte  min(sign-extended byte, 1)  as a signed 16-bit
// SDWA minimum written back into its byte (0 → 0, 2..8 → 1, 0xFF = -1 → -1).  The four dwords are interleaved so that three
// instructions separate two partial writes of the same register (gfx940+: a VALU write with dst_sel needs a wait state
// before the register is read again; inline assembly is outside the compiler's hazard recogniser — back to back the third
// byte came out wrong), and the trailing s_nop covers whatever the compiler places behind the block.
// Checked against bits_bin1 on 65 536 random dwords (profiles/r03/ab_bin4_sdwa.log) and by the FUNCTIONAL parity tests.
__device__ __forceinline__ u32x4 bits_bin4(const u32x4 v) {
  uint32_t a = v.x, b = v.y, c = v.z, d = v.w;
#define TG_R(B) "v_min_i16_sdwa %0, sext(%0), %4 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:WORD_0\n\t" \
                "v_min_i16_sdwa %1, sext(%1), %4 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:WORD_0\n\t" \
                "v_min_i16_sdwa %2, sext(%2), %4 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:WORD_0\n\t" \
                "v_min_i16_sdwa %3, sext(%3), %4 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:WORD_0\n\t"
  asm(TG_R(0) TG_R(1) TG_R(2) TG_R(3) "s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1u));
#undef TG_R
  return u32x4{a, b, c, d};
}

// Two playfield rows (2 W bytes = 2 W / 4 dwords, row pairs start dword-aligned because W is even) → the same two rows of
// the padded board (bedrock | row | bedrock, W + 8 bytes each; envs/tetris.py:632-641).  Both rows map source byte → output
// byte with a shift that is a multiple of 4 (4 and 12), so every output dword is a constant, a source dword, or half of each.
template <int W>
__device__ __forceinline__ void dict_pair(const uint32_t (&src)[2 * W / 4], uint32_t (&out)[(2 * W + 16) / 4]) {
  constexpr int WP = W + 8;
#pragma unroll
  for (int j = 0; j < 2 * WP / 4; ++j) {
    uint32_t v = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int pos = 4 * j + b;
      const int srcb = (pos >= 4 && pos < 4 + W) ? pos - 4 : ((pos >= WP + 4 && pos < WP + 4 + W) ? pos - 12 : -1);
      if (srcb >= 0) v |= src[srcb / 4] & (0xFFu << (8 * b));
      else v |= 0x01u << (8 * b);
    }
    out[j] = v;
  }
}

// 4 cells of a matrix row (bits) → 4 bytes of value `val`
__device__ __forceinline__ uint32_t bits_spread(const uint32_t nib, const uint32_t val) {
  return ((nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21)) * val;
}

// Two rows of the RgbObservation id image (wrappers/observation.py:38-74): padded board row | strip row (16 bytes: the
// queue pieces on the top four rows, the holder on the bottom four, ones in between), from two playfield rows and the
// two strip rows.  Byte-wise at compile time; the compiler folds it into dword moves, masks and byte permutes.
template <int W>
__device__ __forceinline__ void rgb_pair(const uint32_t (&src)[2 * W / 4], const uint32_t (&st0)[4], const uint32_t (&st1)[4],
                                         uint32_t (&out)[2 * (W + 24) / 4]) {
  constexpr int WP = W + 8, RS = WP + 16;
#pragma unroll
  for (int j = 0; j < 2 * RS / 4; ++j) {
    uint32_t v = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int pos = 4 * j + b;
      const int row = pos >= RS ? 1 : 0, q = pos - row * RS;
      uint32_t byte;
      if (q < 4 || (q >= 4 + W && q < WP)) {
        byte = 1u;
      } else if (q < 4 + W) {
        const int sb = q - 4 + row * W;
        byte = (src[sb / 4] >> (8 * (sb % 4))) & 0xFFu;
      } else {
        const int k = q - WP;
        byte = ((row ? st1[k / 4] : st0[k / 4]) >> (8 * (k % 4))) & 0xFFu;
      }
      v |= byte << (8 * b);
    }
    out[j] = v;
  }
}

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 __attribute__((aligned(4))) u32x3_a4;

// RgbObservation of the envs [e_begin, e_end) of the workgroup (slots), GR at a time: id image built in `ids` (this wave's
// staging buffer) from the slabs (locked cells applied, active piece drawn) and the published queue / holder, then every
// four pixels become three dwords through the colour table and leave as 12-byte stores (a wave writes 768 contiguous
// bytes per instruction).  post2[slot] = {queue bytes 0-3, queue bytes 4-6 | holder piece << 24, holder rot | val << 8, -}.
template <int W, int H, int GR>
__device__ __forceinline__ void rgb_pack(const uint8_t* slabs, const uint4* post2, const uint64_t* row_tab, const uint32_t* lut,
                                         uint8_t* ids, const int e_begin, const int e_end, const uint32_t nlive,
                                         const int lane, uint8_t* dst0, uint8_t* dst1) {
  constexpr int S = W * H, WP = W + 8, HP = H + 4, RS = WP + 16, NPX = HP * RS, IDB = ((NPX + 15) / 16) * 16;
  constexpr int NPAIR = HP / 2, NS = 2 * W / 4, ND = 2 * RS / 4;
  constexpr int BT = (GR * NPAIR + 63) / 64, NQ = GR * NPX / 4, QT = (NQ + 63) / 64;
  for (int e0 = e_begin; e0 < e_end; e0 += GR) {
#pragma unroll
    for (int i = 0; i < BT; ++i) {
      const int t = lane + 64 * i;
      const int tt = t < GR * NPAIR ? t : GR * NPAIR - 1;
      const int el = tt / NPAIR, k = tt - el * NPAIR;
      const uint32_t* sp = reinterpret_cast<const uint32_t*>(slabs + (e0 + el) * S + (k < H / 2 ? k : 0) * 2 * W);
      uint32_t src[NS];
#pragma unroll
      for (int j = 0; j < NS; ++j) src[j] = sp[j];
      const uint4 p2 = post2[e0 + el];
      const uint64_t qb = (uint64_t)p2.x | ((uint64_t)(p2.y & 0x00FFFFFFu) << 32);
      const uint32_t hp = p2.y >> 24, hr = p2.z & 0xFFu, hv = (p2.z >> 8) & 0xFFu;
      uint32_t st[2][4];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) st[r][c] = 0x01010101u;
      if (k < 2) {  // rows 0..3: the next four pieces at rotation 0, values id + 2
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t tq = (uint32_t)((qb >> (8 * q)) & 0xFF);
          const uint32_t m = (uint32_t)row_tab[tq < 7 ? tq : 0] & 0xFFFFu;
          st[0][q] = bits_spread((m >> (8 * k)) & 0xFu, tq + 2u);
          st[1][q] = bits_spread((m >> (8 * k + 4)) & 0xFu, tq + 2u);
        }
      } else if (k >= NPAIR - 2 && hp != TG_NONE) {  // rows Hp-4..Hp-1: the held matrix (keeps its rotation and values)
        const uint32_t m = (uint32_t)(row_tab[hp < 7 ? hp : 0] >> (16 * (hr & 3u))) & 0xFFFFu;
        const int kk = k - (NPAIR - 2);
        st[0][0] = bits_spread((m >> (8 * kk)) & 0xFu, hv);
        st[1][0] = bits_spread((m >> (8 * kk + 4)) & 0xFu, hv);
      }
      if (k >= H / 2) {  // bedrock rows below the playfield
#pragma unroll
        for (int j = 0; j < NS; ++j) src[j] = 0x01010101u;
      }
      uint32_t out[ND];
      rgb_pair<W>(src, st[0], st[1], out);
      if (t < GR * NPAIR) {
        uint32_t* dp = reinterpret_cast<uint32_t*>(ids + el * IDB + k * 2 * RS);
#pragma unroll
        for (int j = 0; j < ND; ++j) dp[j] = out[j];
      }
    }
    wave_sync();
    uint32_t idw[QT];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
      const int q = lane + 64 * i;
      const int ql = q < NQ ? q : NQ - 1;
      const int el = ql / (NPX / 4), qq = ql - el * (NPX / 4);
      idw[i] = reinterpret_cast<const uint32_t*>(ids + el * IDB)[qq];
    }
#pragma unroll
    for (int i = 0; i < QT; ++i) {
      const int q = lane + 64 * i;
      const int el = q / (NPX / 4), qq = q - el * (NPX / 4);
      uint32_t c4[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t id = (idw[i] >> (8 * t)) & 0xFFu;
        c4[t] = lut[id <= 8u ? id : 8u];
      }
      if (q < NQ && (uint32_t)(e0 + el) < nlive) {
        const u32x3 px = u32x3{c4[0] | (c4[1] << 24), (c4[1] >> 8) | (c4[2] << 16), (c4[2] >> 16) | (c4[3] << 8)};
        const size_t off = (size_t)(e0 + el) * (size_t)(NPX * 3) + 12u * (size_t)qq;
        *reinterpret_cast<u32x3_a4*>(dst0 + off) = px;
        if (dst1) *reinterpret_cast<u32x3_a4*>(dst1 + off) = px;
      }
    }
    wave_sync();
  }
}

// obs["active_tetromino_mask"] (envs/tetris.py:572-576: the bounding box of the native matrix) of GD envs: stamp the boxes
// into an all-zero staging buffer, let the group leave as one contiguous block, un-stamp.  `pw` = the .w words of the
// envs' post records {x | y << 8 | piece << 16}.
template <int W, int H, int GD>
__device__ __forceinline__ void dict_mask_group(uint8_t* buf, const uint4* post_g, const uint32_t nlive_g, const int lane,
                                                uint8_t* dst0, uint8_t* dst1) {
  constexpr int WP = W + 8, HP = H + 4, PC = WP * HP, NCHG = GD * PC / 16, CT = (NCHG + 63) / 64, ST = GD * 16 / 64;
  int addr[ST];
  uint32_t pws[ST];
#pragma unroll
  for (int i = 0; i < ST; ++i) pws[i] = post_g[(lane + 64 * i) >> 4].w;
#pragma unroll
  for (int i = 0; i < ST; ++i) {
    const int t = lane + 64 * i;
    const int el = t >> 4, k = (t >> 2) & 3, c = t & 3;
    const uint32_t pw = pws[i];
    const int x = (int)(pw & 0xFFu), y = (int)((pw >> 8) & 0xFFu), pc = (int)((pw >> 16) & 0xFFu);
    const int n = pc == 0 ? 4 : (pc == 1 ? 2 : 3);
    const bool ok = (uint32_t)el < nlive_g && k < n && c < n && y + k < HP && x + c < WP;
    addr[i] = ok ? el * PC + (y + k) * WP + x + c : -1;
  }
#pragma unroll
  for (int i = 0; i < ST; ++i)
    if (addr[i] >= 0) buf[addr[i]] = 1;
  wave_sync();
  u32x4 v[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    const int c = lane + 64 * i;
    v[i] = reinterpret_cast<const u32x4*>(buf)[c < NCHG ? c : NCHG - 1];
  }
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    const int c = lane + 64 * i;
    if (c < NCHG && (uint32_t)c / (PC / 16) < nlive_g) {
      st_dict_mask(reinterpret_cast<u32x4*>(dst0 + 16 * (size_t)c), v[i]);
      if (dst1) st_dict_mask(reinterpret_cast<u32x4*>(dst1 + 16 * (size_t)c), v[i]);
    }
  }
  wave_sync();
#pragma unroll
  for (int i = 0; i < ST; ++i)
    if (addr[i] >= 0) buf[addr[i]] = 0;
  wave_sync();
}

// (Storing into TG_BUF_OBS only the 16-byte chunks that the previous / the new bounding box covers was measured in round 3 and
// rejected: ~7 scattered chunks per env cost 18.0 us per transition against 17.1 us for the whole array as one contiguous
// full-line stream — partial-line stores are several times dearer per byte.  profiles/r03/ab_dict_incremental_mask.log)

// one lane copies its env's slab LDS → HBM (8-byte aligned on both sides): batches of 8-byte LDS reads, all reads of a
// batch in flight before its 16-byte stores.  BIN: through the functional observation's byte map.
template <int S, bool BIN>
__device__ __forceinline__ void bits_copy_slab(const uint8_t* src_lds, uint8_t* dst) {
  if constexpr (S % 4 != 0) {  // (slabs at 2- or 1-byte alignment: byte by byte; the terminal observation is rare)
#pragma unroll 4
    for (int u = 0; u < S; ++u) {
      const uint8_t v = src_lds[u];
      dst[u] = BIN ? (uint8_t)(v >= 0x80u ? 0xFFu : (v ? 1u : 0u)) : v;
    }
    return;
  }
  if constexpr (S % 8 != 0) {  // (slabs of such sizes start at 4-byte alignment only: dword by dword; the terminal observation is rare)
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src_lds);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(dst);
#pragma unroll 4
    for (int u = 0; u < S / 4; ++u) d4[u] = BIN ? bits_bin1(s4[u]) : s4[u];
    return;
  }
  const u32x2* src = reinterpret_cast<const u32x2*>(src_lds);
  constexpr int NU = S / 8, UB = NU <= 32 ? NU : 20;
#pragma unroll 1
  for (int u0 = 0; u0 < NU; u0 += UB) {
    u32x2 t2[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) t2[u] = src[u0 + u < NU ? u0 + u : NU - 1];
    if (BIN) {
#pragma unroll
      for (int u = 0; u < UB; ++u) t2[u] = u32x2{bits_bin1(t2[u].x), bits_bin1(t2[u].y)};
    }
#pragma unroll
    for (int u = 0; u + 1 < UB; u += 2)  // 16-byte stores at 8-byte alignment
      if (u0 + u + 1 < NU)
        *reinterpret_cast<u32x4_a8*>(dst + 8 * (u0 + u)) = u32x4{t2[u].x, t2[u].y, t2[u + 1].x, t2[u + 1].y};
      else if (u0 + u < NU)
        *reinterpret_cast<u32x2*>(dst + 8 * (u0 + u)) = t2[u];
    if (UB % 2 == 1 && u0 + UB - 1 < NU) *reinterpret_cast<u32x2*>(dst + 8 * (u0 + UB - 1)) = t2[UB - 1];
  }
}

constexpr int TG_ONE_WAVES = 4;    // waves per SIMD the one-transition kernels are compiled for (4 → 128 VGPRs, 3 → 168)
constexpr int TG_MULTI_WAVES = 3;  // ... the kernels of several transitions per launch (NUMPY: at 4 the plain one spilled 12 registers: 2.90 -> 2.73 us per
                                   // transition; FUNCTIONAL since round 5: at 4 its one-stream-wave builds spill, 98 304 envs 5.77 -> 4.75: profiles/r05/ab_fn_multi_waves_per_simd.log)
constexpr int TG_WIDE_WAVES = 2;   // ... the kFeatWide instantiations (FUNCTIONAL, up to 65 536 envs: one logic + one stream wave per SIMD)
constexpr int bits_waves_per_simd(const int ruleset, const int feat) {
  if (ruleset == TG_RULESET_FUNCTIONAL && (feat & kFeatSplit)) return 3;  // (three waves per workgroup, one workgroup's worth per SIMD at 65 536 envs)
  return (feat & kFeatWide) ? TG_WIDE_WAVES
         : (feat & kFeatMulti) ? TG_MULTI_WAVES
                               : TG_ONE_WAVES;
}
template <int W, int H, int RULESET, int FEAT>
__global__ __attribute__((amdgpu_flat_work_group_size(64 * bits_wg_waves<W, H>(FEAT), 64 * bits_wg_waves<W, H>(FEAT)),
                          amdgpu_waves_per_eu(bits_waves_per_simd(RULESET, FEAT), bits_waves_per_simd(RULESET, FEAT))))
void step_kernel_bits(
    // The arguments every wave needs BEFORE its first global loads, as scalars: the first 14 dwords of the kernel-argument
    // segment are preloaded into SGPRs by the dispatcher (-mllvm -amdgpu-kernarg-preload-count=16, 14 user SGPRs are free;
    // struct arguments are not eligible), so the slab / record / ring loads leave without waiting for a scalar load of
    // the argument block (a memory round trip at the start of every workgroup).  The structs arrive in their shadow.
    uint8_t* const hot_board, uint32_t* const hot_rec, void* const hot_cols, const uint4* const hot_ring_rd,
    const uint32_t* const hot_hbuf_rd, const void* const hot_actions, const int hot_batch,
    const uint32_t hot_misc,  // step workgroups (bits 0-23) | act_i8 << 24 | nsteps << 25
    const KParams p, const BitsParams bp) {
  constexpr bool MULTI = (FEAT & kFeatMulti) != 0, TRAJ = (FEAT & kFeatTraj) != 0, DICTK = (FEAT & kFeatDict) != 0,
                 SAMPLE = (FEAT & kFeatSample) != 0, FINAL = (FEAT & kFeatFinal) != 0, RGBK = (FEAT & kFeatRgb) != 0;
  using BG = BitsGeo<W, H, bits_nseq<W, H>(FEAT)>;
  using T = ColT<H>;
  [[maybe_unused]] constexpr int S = BG::S, EPW = BG::EPW, NCH = BG::NCH, NIT = BG::NIT, CST = BG::CST;
  constexpr T ONES = ~T(0);
  constexpr T FLOOR = (T)(ONES << H);  // rows >= H are bedrock (envs/tetris.py:632-641)
  constexpr T PLAY = (T)~FLOOR;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  u32x4* lds4 = reinterpret_cast<u32x4*>(lds);
  uint64_t* cell_tab = reinterpret_cast<uint64_t*>(lds + BG::OFF_TAB);
  uint64_t* col_tab = cell_tab + 8;
  uint4* post = reinterpret_cast<uint4*>(lds + BG::OFF_POST);
  u32x2* postF = reinterpret_cast<u32x2*>(lds + BG::OFF_FIN);  // (FINAL instantiations only)
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int)(threadIdx.x >> 6));

  const uint32_t B = (uint32_t)hot_batch;
  // (Round 4 built the RNG role on rotating THIRDS of the batch — every one-transition launch serves one third, the ring's double
  // buffer flips per third, every launch at 65 536 envs fits the two-stream-wave build — parity-green, and measured no gain at
  // 65 536 envs, 3-10 % loss at 32 768 / 131 072; even switched off, its scalar arithmetic in front of the first loads cost the
  // headline launch 0.07 us.  Removed again: commit c0175b2 has it, profiles/r04/ab_rng_thirds.log the numbers.)
  // role split by blockIdx: the RNG workgroups lead (placing them last measured the same)
  // the step workgroups are dispatched first (a grid of this size starts first → last within 0.3-0.7 us, and the step
  // workgroups are the longer pole since the RNG role produces one entry per launch)
  const uint32_t n_step = hot_misc & 0xFFFFFFu;
  // (FUNCTIONAL launches have no RNG-role workgroups: compiled out there — the role's code also kept its argument words alive in
  // the scalar loads at the top of the kernel, see TG_LATE_ARGS)
  const bool rng_role = RULESET == TG_RULESET_NUMPY && blockIdx.x >= n_step;
  const uint32_t step_block = blockIdx.x, rng_block = blockIdx.x - n_step;
  const uint32_t base_env = step_block * (uint32_t)EPW;  // (meaningless in the RNG role)
  const uint32_t nlive = B - base_env < (uint32_t)EPW ? B - base_env : (uint32_t)EPW;
  const bool live = (uint32_t)lane < nlive;
  const int slot = live ? lane : (int)nlive - 1;  // dead lanes shadow the last live env (identical, benign LDS writes)
  const uint32_t env = base_env + (uint32_t)slot;
  // What depends on the argument structs (they arrive by scalar loads) is derived in each wave BEHIND its first global
  // loads: the first use of any struct word is where the wave waits for all of them.
  // (Round 4 tried to make this robust by keeping every word of both structs "used" until that point — a 16-byte scalar load
  // of which an instantiation needs one word has dead registers, the allocator reuses them at once, and overwriting a register
  // that a load in flight will write costs an `s_waitcnt lgkmcnt(0)` for the whole argument block at the head of the logic
  // wave.  Pinning removes that wait and loses more than it wins — NUMPY 6.31 -> 6.54 us, functional fused 2.90 -> 3.21 —
  // because every word is then fetched: profiles/r04/ab_pin_kernargs_rejected.log.)
#define TG_LATE_ARGS()                                                                                            \
  const bool need_slab = p.obs_mode == TG_OBS_BOARD || p.obs_mode == TG_OBS_FN;                                   \
  /* (the host picks the dict / RGB instantiations exactly when the handle's observation is packed in the kernel: no run- */ \
  /* time test — the argument words it read were fetched, waited for and copied at the very head of the logic wave)        */ \
  constexpr bool dict = DICTK && BG::DICT_OK;                                                                     \
  constexpr bool rgb = RGBK && BG::RGB_OK;                                                                        \
  uint8_t* const obs_traj = TRAJ ? p.obs_traj : nullptr;                                                          \
  uint2* const out_traj = TRAJ ? p.out_traj : nullptr;                                                            \
  int8_t* const act_traj = TRAJ ? p.act_traj : nullptr;                                                           \
  uint8_t* const final_obs = FINAL ? p.final_obs : nullptr;                                                       \
  /* (p.queue_size shares a 16-byte scalar load with gravity / autoreset / obs_mode; FUNCTIONAL never reads it, the allocator */ \
  /* handed its register to the first scalar value of the prologue, and overwriting a register that a load in flight will    */ \
  /* write costs `s_waitcnt lgkmcnt(0)` — a wait for the argument block at the very top of every wave: 6.93 vs 7.24 us.      */ \
  /* Kept "used" until here, the word's register is not free that early.)                                                     */ \
  asm volatile("" ::"s"(p.queue_size));                                                                           \
  (void)need_slab; (void)dict; (void)rgb; (void)obs_traj; (void)out_traj; (void)act_traj; (void)final_obs
  uint4* post2 = reinterpret_cast<uint4*>(lds + BG::OFF_RGB);                      // (RGB instantiations only)
  const uint32_t* lut_l = reinterpret_cast<const uint32_t*>(lds + BG::OFF_TAB + 256);  // colour table (RGB)
  constexpr int RGBH = bits_rgb_helper_count<W, H>(FEAT);  // RGB helper waves (0 / 2 / 6): quarters / eighths instead of halves
  constexpr int ELOG = RGBH ? EPW / (2 + RGBH) : (EPW / 2);  // RGB: the logic wave packs the envs below, the stream wave those from ELOG on
  constexpr int ESTR = RGBH ? 2 * ELOG : EPW;                  // ... up to here; helper wave w the ELOG envs from ESTR + (w - 2) ELOG on
  constexpr int WP = W + 8, HP = H + 4, PC = BG::PC;
  // p.nsteps consecutive transitions in this launch (tg_step_k): records / masks / slabs stay in registers and LDS in
  // between, only actions come in and observations go out per transition; one s_barrier per transition, and the
  // stream wave packs transition k while the logic wave already decides transition k + 1
  const int hot_nsteps = (int)((hot_misc >> 25) & 31u);
  const bool hot_act_i8 = ((hot_misc >> 24) & 1u) != 0;
  const int nsteps = MULTI ? (hot_nsteps > 1 ? hot_nsteps : 1) : 1;
  const bool sample_mode = SAMPLE && p.sample_mode;  // (device-side action sampling: the kernels that carry it wait here)

  TG_TIMING(unsigned long long tstamp[8];)  // kept in SGPRs and written at the very end: a store per stamp would itself be waited on
  // A role ends with s_endpgm, not with a return: a return joins the function's single exit, and the wait-count pass then
  // treats the RNG role's loads as possibly pending where the logic wave starts — depending on register allocation it put
  // an s_waitcnt vmcnt(0) between the logic wave's table loads and its record loads (6.4 -> 6.9 us per launch).
// (The builtin does not cut the edge either — the code behind it still "falls through" to the common join — so the role
// waits for its own memory operations first: nothing is pending on that edge any more.  The launch cannot complete before
// those stores anyway.)
#define TG_ROLE_END() do { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_endpgm(); } while (0)
  if (rng_role) {
    // =====================================================================================================
    // RNG ROLE (leading workgroups, 128 envs each): refill every ring that holds fewer than kRing permutations.
    // Generator.shuffle on the 7-bag = numpy PCG64 + masked rejection (components/tetromino_randomizer.py:42-46, 84).
    // =====================================================================================================
    TG_TIMING(const unsigned long long rng_t0 = TG_NOW();)
    if (bits_wg_waves<W, H>(FEAT) > 2 && threadIdx.x >= 128) TG_ROLE_END();  // (a third wave of the workgroup has no RNG range)
    const uint32_t e = rng_block * 128u + threadIdx.x;
    const bool on = e < B;
    const uint32_t ee = on ? e : B - 1u;
    const uint4 rw = hot_ring_rd[ee];
    uint64_t w0 = (uint64_t)rw.x | ((uint64_t)rw.y << 32), w1 = (uint64_t)rw.z | ((uint64_t)rw.w << 32);
    const int Hraw = hot_hbuf_rd[ee];
    const int Hn = Hraw & 0xFF;
    const int loc = (Hraw >> 8) & 7;
    __builtin_amdgcn_s_setprio(2);  // between the stream wave before its barrier (1) and the logic wave (3): A/B in profiles/r02
    // One-transition launches: ONE entry per env and launch (a second one waits for the next launch: 13 % of the waves
    // would otherwise run two shuffles back to back and become the tail of the launch — measured 7.0 us vs 5.8 us for the
    // step workgroups).  Launches of several transitions consume more per env and are long anyway: refill completely.
    const int max_reps = nsteps > 1 ? kRing - 1 : 1;
    BitsRng g;
    g.s_lo = g.s_hi = g.i_lo = g.i_hi = 0;
    g.uinteger = g.has32 = 0;
    g.loaded = false;
    uint4* a4 = bp.ahead + (size_t)ee * (2 * kRing);
    const uint4* rng4 = reinterpret_cast<const uint4*>(p.rng + ee);
    for (int rep = 0; rep < max_reps; ++rep) {
      int Tn = (int)(w0 >> 56) & 0xFF;
      const int d = ring_diff(Tn, Hn);
      const bool need = on && d < bp.fill;
      if (!__any(need)) break;
      if (need && !g.loaded) {
        // frontier = state after entry T - 1 (later entries of this visit continue from the registers).  The consumer may
        // have run past T with in-line shuffles (d <= 0): the entries below H are dead — continue from the consumer's
        // state, wherever `loc` says it is (never a location a step launch writes while this one reads: see above)
        const bool resync = d <= 0;
        if (resync) Tn = Hn;
        const int fs = (Tn - 1) & (kRing - 1);
        const uint4* fsrc = a4 + 2 * fs;
        const uint4* fsrc1 = fsrc + 1;
        if (resync && loc == kLocRec) { fsrc = rng4; fsrc1 = rng4 + 2; }
        if (resync && loc >= kLocBuf0) { fsrc = bp.inl + ((size_t)(loc - kLocBuf0) * p.ring_stride + ee) * 2; fsrc1 = fsrc + 1; }
        const uint4 f0 = *fsrc, f1 = *fsrc1;
        const uint4 gi = rng4[1];
        g.s_lo = (uint64_t)f0.x | ((uint64_t)f0.y << 32); g.s_hi = (uint64_t)f0.z | ((uint64_t)f0.w << 32);
        g.i_lo = (uint64_t)gi.x | ((uint64_t)gi.y << 32); g.i_hi = (uint64_t)gi.z | ((uint64_t)gi.w << 32);
        g.uinteger = f1.x; g.has32 = f1.y;
        g.loaded = true;
      }
      uint32_t sig = 0;
      bits_shuffle_perm(g, sig, need);
      if (need) {
        const int sl = Tn & (kRing - 1);
        a4[2 * sl] = make_uint4((uint32_t)g.s_lo, (uint32_t)(g.s_lo >> 32), (uint32_t)g.s_hi, (uint32_t)(g.s_hi >> 32));
        a4[2 * sl + 1] = make_uint4(g.uinteger, g.has32, 0u, 0u);
        const uint64_t m = 0x0FFFFFFFULL << (28 * (sl & 1)), v = (uint64_t)sig << (28 * (sl & 1));
        if (sl & 2) w1 = (w1 & ~m) | v;
        else w0 = (w0 & ~m) | v;
        w0 = (w0 & ~(0xFFULL << 56)) | ((uint64_t)((Tn + 1) & 0xFF) << 56);
      }
    }
    if (on) {
      u32x4* rdst = reinterpret_cast<u32x4*>(bp.ring_wr) + e;
      const u32x4 rv = u32x4{(uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32)};
      st_state(rdst, rv);
    }
    TG_TIMING(if (p.dbg && (threadIdx.x & 63) == 0) {  // lifetime of the RNG waves, behind the step workgroups' stamps
      const size_t slot = (size_t)(gridDim.x - bp.nrb) * 16 + (size_t)rng_block * 4 + (threadIdx.x >> 6) * 2;
      p.dbg[slot] = rng_t0;
      p.dbg[slot + 1] = TG_NOW();
    })
    TG_ROLE_END();
  }
  TG_BITS_STAMP(0);

  if constexpr (bits_mask_wave<W, H>(FEAT)) {
    if (wave == 1 + bits_stream_waves<W, H>(FEAT)) {
      // =====================================================================================================
      // MASK WAVE (kFeatMaskWave): obs["active_tetromino_mask"] of the workgroup's envs, GD at a time, from the post records
      // the logic wave publishes — one barrier per transition, like the stream wave.
      // =====================================================================================================
      __builtin_amdgcn_s_setprio(1);
      TG_LATE_ARGS();
      const size_t Bt = p.act_stride;  // the handle's batch
      constexpr int GD = BG::GD;
      uint8_t* grpm = lds + BG::OFF_DICT + GD * BG::PC;  // (zeroed by the logic wave in front of its first publish)
      for (int step = 0; step < nsteps; ++step) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        uint8_t* d0 = obs_traj ? obs_traj + (size_t)step * p.obs_traj_stride : p.obs;
        uint8_t* d1 = (step == nsteps - 1 && obs_traj) ? p.obs : nullptr;
        for (int g0 = 0; g0 < EPW; g0 += GD) {
          const uint32_t nlg = nlive > (uint32_t)g0 ? nlive - (uint32_t)g0 : 0u;
          dict_mask_group<W, H, GD>(grpm, post + (step & 1) * EPW + g0, nlg, lane, d0 + Bt * BG::PC + ((size_t)base_env + g0) * BG::PC,
                                    d1 ? d1 + Bt * BG::PC + ((size_t)base_env + g0) * BG::PC : nullptr);
        }
      }
      TG_ROLE_END();
    }
  }
  if constexpr (bits_rgb_helpers<W, H>(FEAT)) {
    if (wave >= 2) {
      // =====================================================================================================
      // RGB HELPER WAVES (kFeatMaskWave | kFeatRgb): a quarter of the workgroup's RgbObservation images each, from the slabs the
      // stream wave has finished (second barrier of the transition) and the queue / holder words the logic wave published.
      // =====================================================================================================
      __builtin_amdgcn_s_setprio(1);
      TG_LATE_ARGS();
      constexpr size_t IMG = (size_t)BG::NPX * 3;
      const int e_lo = ESTR + (wave - 2) * ELOG;
      uint8_t* ids = lds + BG::OFF_RGB + EPW * 16 + wave * BG::GR * BG::IDB;  // (buffers 0 / 1: logic / stream wave)
      for (int step = 0; step < nsteps; ++step) {
        const bool lastt = step == nsteps - 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();  // published
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();  // the slabs are final, the pieces drawn
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        uint8_t* dst0 = (obs_traj ? obs_traj + (size_t)step * p.obs_traj_stride : p.obs) + (size_t)base_env * IMG;
        uint8_t* dst1 = (lastt && obs_traj) ? p.obs + (size_t)base_env * IMG : nullptr;
        rgb_pack<W, H, BG::GR>(lds, post2, cell_tab + 24, lut_l, ids, e_lo, e_lo + ELOG, nlive, lane, dst0, dst1);
        if (!lastt) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
          __builtin_amdgcn_s_barrier();  // everybody is done with the slabs
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
      }
      TG_ROLE_END();
    }
  }
  if constexpr (bits_nseq<W, H>(FEAT) > 1) {
    if (__builtin_expect(wave >= 1, 0)) {
      // =====================================================================================================
      // STREAM WAVE of the kFeatSeq instantiations (packed board, one transition per launch, boards above 640 cells): the workgroup's
      // envs in NSEQ groups of EPS, one after the other through ONE slab block in LDS.  Every group's packed slabs are requested up front
      // (they sit in registers until the group's turn: no load latency between the groups, no second barrier — the logic wave has
      // published all groups at once); per group: expand -> lock cells / row compaction / resets -> write-back -> piece -> observation.
      // =====================================================================================================
      constexpr int NSEQ = bits_nseq<W, H>(FEAT), EPS = BG::EPW / NSEQ, NCHP = NCH / 2, NITP = (NCHP + 63) / 64;
      static_assert(!MULTI && bits_packed<W, H>(FEAT) && !DICTK && !RGBK && !FINAL && NCH == EPS * S / 16, "kFeatSeq: packed one-transition builds");
      __builtin_amdgcn_s_setprio(1);
      __builtin_amdgcn_s_sleep(32);  // (the logic waves' records first: see the general stream wave)
      uint64_t tab_cells = 0;
      if (lane < 7) tab_cells = c_cells.cells[lane];
      u32x4 stage[NSEQ][NITP];
#pragma unroll
      for (int g = 0; g < NSEQ; ++g) {
        const uint32_t gb = (uint32_t)(g * EPS);
        const uint32_t nl = nlive > gb ? (nlive - gb < (uint32_t)EPS ? nlive - gb : (uint32_t)EPS) : 0u;
        const uint32_t ncp = nl ? (nl * (uint32_t)S / 2u + 15u) / 16u : 1u;  // (an empty group of a ragged workgroup re-reads chunk 0 of group 0)
        const u32x4* gsrc = reinterpret_cast<const u32x4*>(hot_board + (size_t)(base_env + (nl ? gb : 0u)) * (S / 2));
#pragma unroll
        for (int k = 0; k < NITP; ++k) {
          const uint32_t c = (uint32_t)lane + 64u * k;
          stage[g][k] = gsrc[c < ncp ? c : ncp - 1u];
        }
      }
      if (lane < 7) cell_tab[lane] = tab_cells;
      TG_LATE_ARGS();
      auto differs = [](const u32x4 a, const u32x4 b) -> bool { return ((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) != 0u; };
      const bool fn = p.obs_mode == TG_OBS_FN;
#pragma unroll
      for (int g = 0; g < NSEQ; ++g) {
        const uint32_t gb = (uint32_t)(g * EPS);
        const uint32_t nl = nlive > gb ? (nlive - gb < (uint32_t)EPS ? nlive - gb : (uint32_t)EPS) : 0u;
        if (g > 0 && nl == 0u) break;  // (ragged last workgroup)
        const uint32_t benv = base_env + gb;
        const bool lv = (uint32_t)lane < nl;
        const int sl = lv ? lane : (int)nl - 1;
        const bool full = nl == (uint32_t)EPS;
        const uint32_t nbytes = nl * (uint32_t)S, nchunks = (nbytes + 15u) / 16u, nchunks_p = (nbytes / 2u + 15u) / 16u;
#pragma unroll
        for (int k = 0; k < NITP; ++k) {
          const uint32_t c = (uint32_t)lane + 64u * k;
          u32x4 a, b;
          nib_expand(stage[g][k], a, b);
          if (c < nchunks_p) { lds4[2 * c] = a; lds4[2 * c + 1] = b; }
        }
        if (g == 0) {
          TG_BITS_STAMP(1);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
          __builtin_amdgcn_s_barrier();  // the logic wave has published what happened to every env of the workgroup
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
          __builtin_amdgcn_s_setprio(3);
          TG_BITS_STAMP(2);
        } else {
          wave_sync();
        }
        const uint4 po = post[(int)gb + sl];  // (layout: the general stream wave)
        const bool place = lv && ((po.x >> 24) & 1u), do_clear = lv && ((po.x >> 25) & 1u), do_reset = lv && ((po.x >> 26) & 1u);
        const bool dirty = place || do_reset;
        LaneBoard<W, H, 1> bd;
        bd.my = lds + sl * S;
        bd.sub = 0;
        if (__any(dirty)) {
          const bool nib_over = bd.template stamp<RULESET == TG_RULESET_FUNCTIONAL>(
              po.x & 0xFFFFu, (int)(po.y & 0xFFu), (int)((po.y >> 8) & 0xFFu), (po.x >> 16) & 0xFFu, RULESET == TG_RULESET_FUNCTIONAL, place);
          if constexpr (RULESET == TG_RULESET_FUNCTIONAL) {
            if (nib_over) *p.nib_over = 1u;
          }
          wave_sync();
          uint64_t todo = __ballot(do_clear);
          while (todo) {
            const int l = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            (void)clear_rows<W, H, RULESET>(lds + l * S, lane);
          }
          todo = __ballot(do_reset);
          while (todo) {
            const int l = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            uint32_t* sw = reinterpret_cast<uint32_t*>(lds + l * S);
            for (int i = lane; i < S / 4; i += 64) sw[i] = 0u;
          }
          wave_sync();
          // board write-back, chunk-major (see the general stream wave): the last full group compares against the registers it was
          // loaded into, the others store the packed chunks that hold cells of a changed env
          const uint64_t dm = __ballot(dirty);
          u32x4* pdst = reinterpret_cast<u32x4*>(hot_board + (size_t)benv * (S / 2));
          const __amdgpu_buffer_rsrc_t pwin = st_window(pdst, nchunks_p * 16u);
          // (measured and not adopted, profiles/r06/ab_seq_variants.log: the compare for EVERY group — group 0's registers kept alive to
          // its write-back, 11 MB fewer bytes stored per launch — 18.3 -> 19.2 us; the slab loads without the initial sleep: 19.2)
          const bool keep_b0 = full && g == NSEQ - 1;
#pragma unroll
          for (int k0 = 0; k0 < NITP; k0 += 4) {
            u32x4 pa[4], pb[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const int c = lane + 64 * (k0 + kk);
              const int cc = c < NCHP ? c : NCHP - 1;
              pa[kk] = lds4[2 * cc]; pb[kk] = lds4[2 * cc + 1];
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              if (k0 + kk >= NITP) continue;
              const uint32_t c = (uint32_t)lane + 64u * (uint32_t)(k0 + kk);
              const u32x4 pw = nib_pack(pa[kk], pb[kk]);
              const uint32_t e0 = (32u * c) / (uint32_t)S, e1r = (32u * c + 31u) / (uint32_t)S;
              const uint32_t e1 = e1r < (uint32_t)EPS ? e1r : (uint32_t)EPS - 1u;
              const bool hit = keep_b0 ? differs(pw, stage[g][k0 + kk])
                                       : ((((dm >> (e0 < (uint32_t)EPS ? e0 : (uint32_t)EPS - 1u)) | (dm >> e1)) & 1ull) != 0);
              st_state_if(pwin, 16u * c, pw, c < nchunks_p && hit);
            }
          }
        }
        if (g == NSEQ - 1) TG_BITS_STAMP(3);
        if (need_slab) {
          const bool show = lv && ((po.z >> 24) & 1u);
          bd.stamp(po.z & 0xFFFFu, (int)(po.w & 0xFFu), (int)((po.w >> 8) & 0xFFu), fn ? 0xFFu : ((po.z >> 16) & 0xFFu), false, show);
          wave_sync();
          u32x4* odst = reinterpret_cast<u32x4*>(p.obs + (size_t)benv * S);
          u32x4* odst2 = reinterpret_cast<u32x4*>((obs_traj ? obs_traj : p.obs) + (size_t)benv * S);  // (trajectory slot 0 of a one-transition rollout)
          if (full) {
            u32x4 o[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
              const int c = lane + 64 * k;
              o[k] = lds4[c < NCH ? c : NCH - 1];
            }
            if (fn) {
#pragma unroll
              for (int k = 0; k < NIT; ++k) o[k] = bits_bin4(o[k]);
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
              const int c = lane + 64 * k;
              if (64 * k + 63 >= NCH) {  // the last, partial pass: the window drops the lanes behind the block
                __builtin_amdgcn_raw_buffer_store_b128(o[k], st_window(odst, (uint32_t)NCH * 16u), 16u * (uint32_t)c, 0, 2);
                if (obs_traj) __builtin_amdgcn_raw_buffer_store_b128(o[k], st_window(odst2, (uint32_t)NCH * 16u), 16u * (uint32_t)c, 0, 2);
                continue;
              }
              st_stream(&odst[c], o[k]);
              if (obs_traj) st_stream(&odst2[c], o[k]);
            }
          } else {
            for (uint32_t c = lane; c < nchunks; c += 64) {
              u32x4 vv = lds4[c];
              if (fn) vv = u32x4{bits_bin1(vv.x), bits_bin1(vv.y), bits_bin1(vv.z), bits_bin1(vv.w)};
              if (c * 16u + 16u <= nbytes) {
                odst[c] = vv;
                if (obs_traj) odst2[c] = vv;
              } else {  // ragged tail of a partial group
                const uint32_t w4[4] = {vv.x, vv.y, vv.z, vv.w};
                for (uint32_t i = 0; c * 16u + i < nbytes; ++i) {
                  const uint8_t bv = (uint8_t)(w4[i >> 2] >> (8 * (i & 3)));
                  reinterpret_cast<uint8_t*>(odst)[c * 16u + i] = bv;
                  if (obs_traj) reinterpret_cast<uint8_t*>(odst2)[c * 16u + i] = bv;
                }
              }
            }
          }
          if (g + 1 < NSEQ) wave_sync();  // (the next group's expansion overwrites the slab block)
        }
      }
      TG_BITS_STAMP(4);
      TG_TIMING(if (p.dbg && lane == 0) {
        for (int i = 0; i < 5; ++i) p.dbg[(size_t)step_block * 16 + 8 + i] = tstamp[i];
        p.dbg[(size_t)step_block * 16 + 14] = __builtin_amdgcn_s_getreg(63492);
      })
      TG_ROLE_END();
    }
  }
  // (the stream block out of line: the logic wave's code comes first, straight behind the common prologue)
  if (__builtin_expect(wave >= 1, 0)) {
    // =====================================================================================================
    // STREAM WAVE(S): the uint8 payload.  Slab block HBM → LDS while the logic wave decides; afterwards the locked
    // cells / cleared rows / resets are applied, changed slabs go back to the board, and the observation leaves.
    // With two stream waves each owns EPS = EPW / 2 envs (lane l < EPS serves env sbase + l) and their NCHS chunks; the
    // names below shadow the workgroup's: everything in this block is relative to the wave's own env range.
    // =====================================================================================================
    __builtin_amdgcn_s_setprio(1);
    // One transition per launch: the slab loads leave ~2 000 cycles late.  The launch starts with every wave of the grid
    // loading at once, and the stream waves' 13 KB per workgroup queue in front of the logic waves' records — which are the
    // start of the critical path, while the slabs are not needed before the publish (6.41 -> 6.28 us, FUNCTIONAL 8.38 ->
    // 8.29; 16 x 64 cycles: half of that; 48: nothing.  profiles/r03/ab_stream_delay.log)
    if (!MULTI) __builtin_amdgcn_s_sleep(32);
    constexpr int NSW = bits_stream_waves<W, H>(FEAT);
    constexpr int EPW_WG = BG::EPW, EPS = EPW_WG / NSW, NCHS = NSW == 1 ? BG::NCH : EPS * S / 16;
    const uint32_t sbase = NSW == 1 ? 0u : (uint32_t)(wave - 1) * (uint32_t)EPS;
    const uint32_t nlive_wg = nlive, base_env_wg = base_env;
    uint8_t* const lds_wg = lds;
    u32x4* const lds4_wg = lds4;
    uint4* const post_wg = post;
    constexpr int EPW = EPS, NCH = NCHS, NIT = (NCHS + 63) / 64;
    const uint32_t nlive = nlive_wg > sbase ? (nlive_wg - sbase < (uint32_t)EPS ? nlive_wg - sbase : (uint32_t)EPS) : 0u;
    if (NSW > 1 && nlive == 0u) {  // (ragged last workgroup: no env in this wave's range — only the barriers)
      for (int step = 0; step < nsteps; ++step) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
      }
      TG_ROLE_END();
    }
    const uint32_t base_env = base_env_wg + sbase;
    const bool live = (uint32_t)lane < nlive;
    const int slot = live ? lane : (int)nlive - 1;
    const uint32_t env = base_env + (uint32_t)slot;
    (void)env;
    uint8_t* const lds = lds_wg + (size_t)sbase * S;
    u32x4* const lds4 = lds4_wg + (NSW == 1 ? 0 : (wave - 1) * NCHS);
    uint4* const post = post_wg + sbase;
    const bool full_wave = nlive == (uint32_t)EPW;
    const uint32_t nbytes = nlive * (uint32_t)S;
    const uint32_t nchunks = (nbytes + 15u) / 16u;
    uint64_t tab_cells = 0;
    if (lane < 7) tab_cells = c_cells.cells[lane];
    constexpr bool PACKED = bits_packed<W, H>(FEAT);
    constexpr int NCHP = NCH / 2, NITP = (NCHP + 63) / 64;      // packed chunks of the wave's slab block (PACKED)
    const uint32_t nchunks_p = (nbytes / 2u + 15u) / 16u;        // ... of a partial workgroup
    u32x4 stage[PACKED ? NITP : NIT];
    if constexpr (PACKED) {
      const u32x4* gsrc = reinterpret_cast<const u32x4*>(hot_board + (size_t)base_env * (S / 2));
#pragma unroll
      for (int k = 0; k < NITP; ++k) {
        const uint32_t c = (uint32_t)lane + 64u * k;
        stage[k] = gsrc[c < nchunks_p ? c : nchunks_p - 1u];
      }
    } else {
      const u32x4* gsrc = reinterpret_cast<const u32x4*>(hot_board + (size_t)base_env * S);
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const uint32_t c = (uint32_t)lane + 64u * k;
        stage[k] = gsrc[c < nchunks ? c : nchunks - 1u];
      }
    }
    if (lane < 7) cell_tab[lane] = tab_cells;
    if constexpr (PACKED) {
#pragma unroll
      for (int k = 0; k < NITP; ++k) {
        const uint32_t c = (uint32_t)lane + 64u * k;
        u32x4 a, b;
        nib_expand(stage[k], a, b);
        if (c < nchunks_p) { lds4[2 * c] = a; lds4[2 * c + 1] = b; }  // (a partial last chunk spills into the block's own tail)
      }
    } else if (full_wave) {
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int c = lane + 64 * k;
        if (c < NCH) lds4[c] = stage[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const uint32_t c = (uint32_t)lane + 64u * k;
        if (c < nchunks) lds4[c] = stage[k];
      }
    }
    TG_BITS_STAMP(1);
    TG_LATE_ARGS();
    // One transition per launch, full workgroup: the slab block stays in the staging registers as it was loaded (B0) and
    // the board write-back stores exactly the chunks whose bytes changed (a lock: one or two of an env's 13) — a compare
    // per chunk instead of index arithmetic, and fewer store instructions in the launch's tail.
    // (Writing the observation twice — B0 from the registers while the logic wave decides, then only the chunks that
    // differ — was measured: the early stores are free, but the sparse second pass costs more than the full one, 8.0 vs
    // 7.3 us; plain or nontemporal, 16-byte or whole-line granularity alike.  profiles/r02/ab_early_obs_store.log)
    constexpr bool ONE = !MULTI;
    const bool keep_b0 = ONE && full_wave;
    auto differs = [](const u32x4 a, const u32x4 b) -> bool { return ((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) != 0u; };
    LaneBoard<W, H, 1> bd;
    bd.my = lds + slot * S;
    bd.sub = 0;
    bool dirty_any = false;  // this lane's env changed its slab in some transition of the launch
    for (int step = 0; step < nsteps; ++step) {
    const bool last = step == nsteps - 1;
    TG_TIMING(if (MULTI && last && step > 0) TG_BITS_STAMP(1);)  // launches of several transitions: the stamps describe the LAST one
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();  // the logic wave has published what happened to every env in this transition
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    // from here on this wave is the workgroup's critical path (round 2 left the functional rule-set out: its logic waves
    // ran threefry in line in front of their publish and were the longer pole; that work now sits behind the publish)
    if (RULESET == TG_RULESET_NUMPY || !MULTI)  // (launches of several functional transitions: the logic wave's look-ahead is on the chain)
    __builtin_amdgcn_s_setprio(3);
    TG_BITS_STAMP(2);
    // post record of this lane's env: x = lock {cells | val << 16 | place << 24 | clear << 25 | reset << 26},
    // y = lock {x | y << 8}, z = observation {cells | val << 16 | show << 24}, w = observation {x | y << 8}
    const uint4 po = post[(step & 1) * EPW_WG + slot];
    const bool place = live && ((po.x >> 24) & 1u), do_clear = live && ((po.x >> 25) & 1u), do_reset = live && ((po.x >> 26) & 1u);
    const bool dirty = place || do_reset;
    dirty_any = dirty_any || dirty;
    if (__any(dirty)) {
      const bool nib_over = bd.template stamp<PACKED && RULESET == TG_RULESET_FUNCTIONAL>(
          po.x & 0xFFFFu, (int)(po.y & 0xFFu), (int)((po.y >> 8) & 0xFFu), (po.x >> 16) & 0xFFu, RULESET == TG_RULESET_FUNCTIONAL, place);
      if constexpr (PACKED && RULESET == TG_RULESET_FUNCTIONAL) {  // (never in play: the engine reports it at its next wait)
        if (nib_over) *p.nib_over = 1u;
      }
      wave_sync();
      uint64_t todo = __ballot(do_clear);
      while (todo) {  // cooperative row compaction of the uint8 slab, one env at a time (rare)
        const int l = (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        (void)clear_rows<W, H, RULESET>(lds + l * S, lane);
      }
      if (FINAL && final_obs) {
        // TG_FLAG_FINAL_OBS: the slab still holds the terminal board of an env that ends its episode here — draw the
        // terminal piece (if the reference would project it), let the env's lane copy it out, un-draw
        const u32x2 pf = postF[(step & 1) * EPW_WG + (int)sbase + slot];  // (the workgroup's array: this wave's envs start at sbase)
        const bool fin = do_reset && ((pf.x >> 25) & 1u);
        if (__any(fin)) {
          const bool showf = fin && ((pf.x >> 24) & 1u);
          const bool fnm = p.obs_mode == TG_OBS_FN;
          bd.stamp(pf.x & 0xFFFFu, (int)(pf.y & 0xFFu), (int)((pf.y >> 8) & 0xFFu), fnm ? 0xFFu : ((pf.x >> 16) & 0xFFu), false, showf);
          wave_sync();
          if (fin) {
            if (fnm) bits_copy_slab<S, true>(bd.my, final_obs + (size_t)env * S);
            else bits_copy_slab<S, false>(bd.my, final_obs + (size_t)env * S);
          }
          wave_sync();
          bd.stamp(pf.x & 0xFFFFu, (int)(pf.y & 0xFFu), (int)((pf.y >> 8) & 0xFFu), 0u, false, showf);
          wave_sync();
        }
      }
      todo = __ballot(do_reset);
      while (todo) {  // cooperative zero of a reset env's slab
        const int l = (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        if constexpr (S % 4 == 0) {
          uint32_t* sw = reinterpret_cast<uint32_t*>(lds + l * S);
          for (int i = lane; i < S / 4; i += 64) sw[i] = 0u;
        } else {  // (slabs of such sizes start at 2- or 1-byte alignment)
          uint8_t* sb = lds + l * S;
          for (int i = lane; i < S; i += 64) sb[i] = 0u;
        }
      }
      wave_sync();
    }
    if (last && __any(dirty_any)) {
      // board write-back, chunk-major: lane l of pass k owns the 16-byte chunk l + 64 k of the workgroup's slab block and
      // stores it when it holds bytes of a changed env.  Neighbouring lanes write neighbouring chunks, so a changed
      // slab costs ~4 full cache-line writes instead of 13 partial ones from its own lane (the launch's store burst is
      // bound by line operations, not bytes).  A chunk shared with an unchanged neighbour rewrites that env's bytes as
      // they are.
      const uint64_t dm = __ballot(dirty_any);
      u32x4* bdst = reinterpret_cast<u32x4*>(hot_board + (size_t)base_env * S);
      if constexpr (PACKED) {
        // packed board: chunk c of the packed block = LDS chunks 2c, 2c + 1; stored when it differs from what was loaded
        // (one transition per launch, full workgroup) or holds cells of a changed env (otherwise).  A partial workgroup's
        // last chunk may run a few bytes past its envs: that is the buffer's tail padding.
        u32x4* pdst = reinterpret_cast<u32x4*>(hot_board + (size_t)base_env * (S / 2));
        const __amdgpu_buffer_rsrc_t pwin = st_window(pdst, nchunks_p * 16u);
#pragma unroll  // (fully: `stage` must stay in registers)
        for (int k0 = 0; k0 < NITP; k0 += 4) {
          u32x4 pa[4], pb[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int c = lane + 64 * (k0 + kk);
            const int cc = c < NCHP ? c : NCHP - 1;
            pa[kk] = lds4[2 * cc]; pb[kk] = lds4[2 * cc + 1];
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            if (k0 + kk >= NITP) continue;
            const uint32_t c = (uint32_t)lane + 64u * (uint32_t)(k0 + kk);
            const u32x4 pw = nib_pack(pa[kk], pb[kk]);
            const uint32_t e0 = (32u * c) / (uint32_t)S, e1r = (32u * c + 31u) / (uint32_t)S;
            const uint32_t e1 = e1r < (uint32_t)EPW ? e1r : (uint32_t)EPW - 1u;
            const bool hit = keep_b0 ? differs(pw, stage[k0 + kk])
                                     : ((((dm >> (e0 < (uint32_t)EPW ? e0 : (uint32_t)EPW - 1u)) | (dm >> e1)) & 1ull) != 0);
            st_state_if(pwin, 16u * c, pw, c < nchunks_p && hit);
          }
        }
      } else if (keep_b0) {
        u32x4 wv[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int c = lane + 64 * k;
          wv[k] = lds4[c < NCH ? c : NCH - 1];
        }
        const __amdgpu_buffer_rsrc_t win = st_window(bdst, (uint32_t)NCH * 16u);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int c = lane + 64 * k;
          st_state_if(win, 16u * (uint32_t)c, wv[k], c < NCH && differs(wv[k], stage[k]));
        }
      } else if (full_wave) {
        // the kernels with the transition loop, full workgroup: which envs own bytes of a chunk is plain index arithmetic
        // (the general path below also carries the ragged tail of a partial workgroup inside its loop: ~70 instructions
        // and three branches per chunk, 4.6k cycles of the stream wave at the end of every launch).  Batches of four
        // chunks: the loop has no 52 registers to spare.
#pragma unroll 1
        for (int k0 = 0; k0 < NIT; k0 += 4) {
          u32x4 wv[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int c = lane + 64 * (k0 + kk);
            wv[kk] = lds4[c < NCH ? c : NCH - 1];
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const uint32_t c = (uint32_t)lane + 64u * (uint32_t)(k0 + kk);
            const uint32_t e0 = (16u * c) / (uint32_t)S, e1r = (16u * c + 15u) / (uint32_t)S;
            const uint32_t e1 = e1r < (uint32_t)EPW ? e1r : (uint32_t)EPW - 1u;
            const bool hit = c < (uint32_t)NCH && (((dm >> e0) | (dm >> e1)) & 1ull) != 0;
            st_state_if(st_window(bdst, (uint32_t)NCH * 16u), 16u * c, wv[kk], hit);
          }
        }
      } else {
      // chunks in flight per lane: all of them in the one-transition kernel; in the transition loop of the fused kernels
      // that would cost 52 registers beside the observation pass's 52 (the kernel spilled)
      constexpr int WBB = MULTI ? 4 : NIT;
#pragma unroll 1
      for (int k0 = 0; k0 < NIT; k0 += WBB) {
        u32x4 wv[WBB];
#pragma unroll
        for (int kk = 0; kk < WBB; ++kk) {
          const int c = lane + 64 * (k0 + kk);
          wv[kk] = lds4[c < NCH ? c : NCH - 1];
        }
#pragma unroll
        for (int kk = 0; kk < WBB; ++kk) {
          const uint32_t c = (uint32_t)lane + 64u * (uint32_t)(k0 + kk);
          const uint32_t e0 = (16u * c) / (uint32_t)S, e1r = (16u * c + 15u) / (uint32_t)S;
          const uint32_t e1 = e1r < (uint32_t)EPW ? e1r : (uint32_t)EPW - 1u;
          const bool hit = k0 + kk < NIT && c < nchunks && (((dm >> e0) | (dm >> e1)) & 1ull) != 0;
          if (hit) {
            if (c * 16u + 16u <= nbytes) {
              st_state(&bdst[c], wv[kk]);
            } else {  // ragged tail of a partial wave
              const uint32_t w4[4] = {wv[kk].x, wv[kk].y, wv[kk].z, wv[kk].w};
              for (uint32_t i = 0; c * 16u + i < nbytes; ++i)
                reinterpret_cast<uint8_t*>(bdst)[c * 16u + i] = (uint8_t)(w4[i >> 2] >> (8 * (i & 3)));
            }
          }
        }
      }
      }
    }
    TG_BITS_STAMP(3);
    if (need_slab) {
      const bool show = live && ((po.z >> 24) & 1u);
      bd.stamp(po.z & 0xFFFFu, (int)(po.w & 0xFFu), (int)((po.w >> 8) & 0xFFu),
               p.obs_mode == TG_OBS_FN ? 0xFFu : ((po.z >> 16) & 0xFFu), false, show);
      wave_sync();
      // trajectory output (tg_rollout): every transition has its own slot; the engine's buffer gets the last one
      const bool to_traj = obs_traj != nullptr && !last;
      uint8_t* obase = to_traj ? obs_traj + (size_t)step * p.obs_traj_stride : p.obs;
      u32x4* odst = reinterpret_cast<u32x4*>(obase + (size_t)base_env * S);
      u32x4* odst2 = reinterpret_cast<u32x4*>((obs_traj ? obs_traj + (size_t)step * p.obs_traj_stride : p.obs) + (size_t)base_env * S);
      const bool fn = p.obs_mode == TG_OBS_FN;
      auto bin1 = [](uint32_t x) -> uint32_t { return bits_bin1(x); };
      if (full_wave) {
        u32x4 o[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int c = lane + 64 * k;
          o[k] = lds4[c < NCH ? c : NCH - 1];
        }
        if (fn) {
#pragma unroll
          for (int k = 0; k < NIT; ++k) o[k] = bits_bin4(o[k]);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int c = lane + 64 * k;
          if (64 * k + 63 >= NCH) {  // the last, partial pass: the window drops the lanes behind the block (no exec-mask region)
            {
              __builtin_amdgcn_raw_buffer_store_b128(o[k], st_window(odst, (uint32_t)NCH * 16u), 16u * (uint32_t)c, 0, 2);
              if (last && obs_traj) __builtin_amdgcn_raw_buffer_store_b128(o[k], st_window(odst2, (uint32_t)NCH * 16u), 16u * (uint32_t)c, 0, 2);
            }
            continue;
          }
          if (c < NCH) {
            st_stream(&odst[c], o[k]);
            if (last && obs_traj) st_stream(&odst2[c], o[k]);
          }
        }
      } else {
        for (uint32_t c = lane; c < nchunks; c += 64) {
          u32x4 vv = lds4[c];
          if (fn) vv = u32x4{bin1(vv.x), bin1(vv.y), bin1(vv.z), bin1(vv.w)};
          if (c * 16u + 16u <= nbytes) {
            odst[c] = vv;
            if (last && obs_traj) odst2[c] = vv;
          } else {  // ragged tail of a partial wave
            const uint32_t w4[4] = {vv.x, vv.y, vv.z, vv.w};
            for (uint32_t i = 0; c * 16u + i < nbytes; ++i) {
              const uint8_t bv = (uint8_t)(w4[i >> 2] >> (8 * (i & 3)));
              reinterpret_cast<uint8_t*>(odst)[c * 16u + i] = bv;
              if (last && obs_traj) reinterpret_cast<uint8_t*>(odst2)[c * 16u + i] = bv;
            }
          }
        }
      }
    }
    if (dict) {
      // obs["board"] (envs/tetris.py:566-571): the padded board with the projected piece.  GD envs at a time: a lane
      // turns playfield row pairs into padded row pairs (dword reads, dword writes) in the staging buffer, then the
      // group leaves as one contiguous coalesced block of GD * Hp * Wp bytes.
      bd.stamp(po.z & 0xFFFFu, (int)(po.w & 0xFFu), (int)((po.w >> 8) & 0xFFu), (po.z >> 16) & 0xFFu, false,
               live && ((po.z >> 24) & 1u));
      wave_sync();
      uint8_t* grp = lds_wg + ((NSW == 2 && wave == 2) ? BG::OFF_DICT2 : BG::OFF_DICT);  // (this wave's staging buffer)
      uint8_t* dst0 = obs_traj ? obs_traj + (size_t)step * p.obs_traj_stride : p.obs;
      uint8_t* dst1 = (last && obs_traj) ? p.obs : nullptr;
      constexpr int GD = BG::GD, NPAIR = HP / 2, NS = 2 * W / 4, ND = 2 * WP / 4;
      constexpr int BT = (GD * NPAIR + 63) / 64, NCHG = GD * PC / 16, CT = (NCHG + 63) / 64;
      for (int g0 = 0; g0 < EPW; g0 += GD) {
        // (all source reads of the lane's row pairs first, then all writes, then all staging reads + row ranges, then
        // the stores: the compiler cannot reorder LDS reads over LDS writes it cannot tell apart, and every dependent
        // round trip costs ~100 cycles of an otherwise idle wave)
        uint32_t src[BT][NS];
#pragma unroll
        for (int i = 0; i < BT; ++i) {
          const int t = lane + 64 * i;
          const int tt = t < GD * NPAIR ? t : GD * NPAIR - 1;
          const int el = tt / NPAIR, k = tt - el * NPAIR;
          const uint32_t* sp = reinterpret_cast<const uint32_t*>(lds + (g0 + el) * S + (k < H / 2 ? k : 0) * 2 * W);
#pragma unroll
          for (int j = 0; j < NS; ++j) src[i][j] = sp[j];
        }
#pragma unroll
        for (int i = 0; i < BT; ++i) {
          const int t = lane + 64 * i;
          const int el = t / NPAIR, k = t - el * NPAIR;
          if (t < GD * NPAIR) {
            uint32_t out[ND];
            dict_pair<W>(src[i], out);
            uint32_t* dp = reinterpret_cast<uint32_t*>(grp + el * PC + k * 2 * WP);
#pragma unroll
            for (int j = 0; j < ND; ++j) dp[j] = k < H / 2 ? out[j] : 0x01010101u;  // (rows below the playfield: bedrock)
          }
        }
        wave_sync();
        u32x4 v[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          const int c = lane + 64 * i;
          v[i] = reinterpret_cast<const u32x4*>(grp)[c < NCHG ? c : NCHG - 1];
        }
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          const int c = lane + 64 * i;
          if (c < NCHG && (uint32_t)g0 + (uint32_t)c / (PC / 16) < nlive) {
            st_dict(reinterpret_cast<u32x4*>(dst0 + ((size_t)base_env + g0) * PC + 16 * (size_t)c), v[i]);
            if (dst1) st_dict(reinterpret_cast<u32x4*>(dst1 + ((size_t)base_env + g0) * PC + 16 * (size_t)c), v[i]);
          }
        }
        wave_sync();
      }
    }
    if (rgb) {
      // RgbObservation: draw the piece, let the logic wave know the slabs are final (second barrier), pack the upper half
      bd.stamp(po.z & 0xFFFFu, (int)(po.w & 0xFFu), (int)((po.w >> 8) & 0xFFu), (po.z >> 16) & 0xFFu, false,
               live && ((po.z >> 24) & 1u));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      constexpr size_t IMG = (size_t)BG::NPX * 3;
      uint8_t* dst0 = (obs_traj ? obs_traj + (size_t)step * p.obs_traj_stride : p.obs) + (size_t)base_env * IMG;
      uint8_t* dst1 = (last && obs_traj) ? p.obs + (size_t)base_env * IMG : nullptr;
      rgb_pack<W, H, BG::GR>(lds, post2, cell_tab + 24, lut_l, lds + BG::OFF_RGB + EPW * 16 + BG::GR * BG::IDB, ELOG, ESTR, nlive, lane, dst0, dst1);
      if (!last) {  // both waves are done with the slabs: un-draw below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      }
    }
    if ((need_slab || dict || rgb) && !last) {  // the slab in LDS is the board again: the drawn piece only ever covers free cells
      wave_sync();
      bd.stamp(po.z & 0xFFFFu, (int)(po.w & 0xFFu), (int)((po.w >> 8) & 0xFFu), 0u, false, live && ((po.z >> 24) & 1u));
      wave_sync();
    }
    }  // transitions
    TG_BITS_STAMP(4);
    TG_TIMING(if (p.dbg && lane == 0) {
      for (int i = 0; i < 5; ++i) p.dbg[(size_t)step_block * 16 + 8 + i] = tstamp[i];
      p.dbg[(size_t)step_block * 16 + 14] = __builtin_amdgcn_s_getreg(63492);  // HW_ID: where the stream wave ran
    })
    TG_ROLE_END();
  }

  // =======================================================================================================
  // LOGIC WAVE: one env per lane on column bitmasks
  // =======================================================================================================
  __builtin_amdgcn_s_setprio(3);  // the decision logic is the critical path of the workgroup: it wins VALU arbitration
  T* colsL = reinterpret_cast<T*>(lds + BG::OFF_COLS) + slot * CST;
  uint64_t tab_cells = 0, tab_cols = 0, tab_rows = 0;
  if (lane < 7) { tab_cells = c_cells.cells[lane]; tab_cols = c_colmasks.cols[lane]; }
  uint32_t tab_lut = 0;
  if (RGBK && lane < 9) tab_lut = c_rgb_lut[lane];
  if ((DICTK || RGBK) && lane < 7)
    tab_rows = (uint64_t)c_shapes.mask[lane][0] | ((uint64_t)c_shapes.mask[lane][1] << 16) |
               ((uint64_t)c_shapes.mask[lane][2] << 32) | ((uint64_t)c_shapes.mask[lane][3] << 48);
  const uint4* rec4 = reinterpret_cast<const uint4*>(hot_rec) + (size_t)env * 2;
  const uint4 ra = rec4[0], rb = rec4[1];
  // (FUNCTIONAL: the jax key is not among the first loads any more — what a transition needs of the key chain comes
  // from the look-ahead ring below, as the bag permutations of the NUMPY rule-set do)
  // the actions of all transitions of this launch, requested up front (int8 rows: one byte per transition and env)
  uint64_t apack = 0, apack1 = 0;
  int action0 = 0;
  if (sample_mode) {
    // (actions are drawn per transition below)
  } else if (hot_act_i8) {
    uint8_t ab[8];  // unconditional loads (row index clamped): all in flight together, no wait between them
#pragma unroll
    for (int k = 0; k < 8; ++k)
      ab[k] = reinterpret_cast<const uint8_t*>(hot_actions)[(size_t)(k < nsteps ? k : nsteps - 1) * p.act_stride + env];
#pragma unroll
    for (int k = 0; k < 8; ++k) apack |= (uint64_t)ab[k] << (8 * k);
    if (kMaxFuse > 8 && MULTI && nsteps > 8) {  // (launches of more than eight transitions: a second word)
      uint8_t ac[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        ac[k] = reinterpret_cast<const uint8_t*>(hot_actions)[(size_t)(8 + k < nsteps ? 8 + k : nsteps - 1) * p.act_stride + env];
#pragma unroll
      for (int k = 0; k < 8; ++k) apack1 |= (uint64_t)ac[k] << (8 * k);
    }
  } else {
    action0 = reinterpret_cast<const int32_t*>(hot_actions)[env];
  }
  uint64_t ring0 = 0, ring1 = 0;
  int Hc = 0;
  uint64_t fnw = 0;  // FUNCTIONAL: the bags this transition may deal (S1 | S2 << 21 | Q0 << 42), see the look-ahead above
  if (RULESET == TG_RULESET_NUMPY) {
    const uint4 rw = hot_ring_rd[env];
    ring0 = (uint64_t)rw.x | ((uint64_t)rw.y << 32);
    ring1 = (uint64_t)rw.z | ((uint64_t)rw.w << 32);
    Hc = hot_hbuf_rd[env];  // H | loc << 8, split below
  } else {
    const uint2 fw = reinterpret_cast<const uint2*>(hot_hbuf_rd)[env];
    fnw = (uint64_t)fw.x | ((uint64_t)fw.y << 32);
  }
  T cv[W];
  {
    constexpr int NP = (int)(BG::WC * sizeof(T) / 8);
    const uint2* gc = reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(hot_cols) + (size_t)env * BG::WC);
    uint2 raw[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) raw[k] = gc[k];
#pragma unroll
    for (int c = 0; c < W; ++c) {
      if constexpr (sizeof(T) == 4) cv[c] = (c & 1) ? raw[c / 2].y : raw[c / 2].x;
      else cv[c] = (T)raw[c].x | ((T)raw[c].y << 32);
    }
  }
  // Episode statistics: the LAST of the first loads — its pointer is not among the preloaded arguments, and nothing needs
  // the value before the barrier.  (Unconditional, from the record when there is no statistics buffer: a branch would
  // make the compiler wait for the load right here.)
  __builtin_amdgcn_sched_barrier(0);  // (the scheduler would lift the pointer test — and with it the wait — above the loads)
  tg_episode_rec ep = *(p.episode ? p.episode + env : reinterpret_cast<const tg_episode_rec*>(rec4));
  TG_LATE_ARGS();
  if (dict)  // the mask staging buffer starts all zero (the groups un-stamp what they stamp)
    for (int i = lane * 16; i < BG::GD * BG::PC; i += 1024)
      *reinterpret_cast<uint4*>(lds + BG::OFF_DICT + BG::GD * BG::PC + i) = make_uint4(0u, 0u, 0u, 0u);
  // private copies of the two piece tables for this wave (the stream wave fills cell_tab for itself after its loads)
  uint64_t* cell_tab_l = col_tab + 8;
  if (lane < 7) { cell_tab_l[lane] = tab_cells; col_tab[lane] = tab_cols; }
  if ((DICTK || RGBK) && lane < 7) col_tab[16 + lane] = tab_rows;
  if (RGBK && lane < 9) reinterpret_cast<uint32_t*>(lds + BG::OFF_TAB + 256)[lane] = tab_lut;
  colsL[0] = ONES;
  colsL[W + 1] = ONES;
#pragma unroll
  for (int c = 0; c < W; ++c) colsL[1 + c] = (T)(cv[c] | FLOOR);
  wave_sync();
  TG_BITS_STAMP(1);

  // ---- B: rule logic (per lane, branch-free except wave-uniform any() branches) ----
  LaneS s;
  lane_unpack(s, ra, rb);
  // NUMPY: the playfield rows that consist of bedrock cells (value 1) only — clear_filled_rows never takes such a row
  // (envs/tetris.py:493: `& ~(board == 1).all(axis=1)`), the column masks cannot tell it from a filled one.  Only an uploaded board
  // holds one; bits_prepare_kernel finds them and keeps the row mask in the two words of the record this rule-set does not use
  // (score | reserved: device-private, stripped by tg_get_state).  It rides in registers the record load fills anyway.
  T brow = 0;
  if (RULESET == TG_RULESET_NUMPY) brow = (T)((uint64_t)__float_as_uint(s.score) | ((uint64_t)s.reserved << 32));
  if (DICTK) {  // the box TG_BUF_OBS's mask array shows: the env's piece as the launch starts (tg_reset / tg_set_state re-pack it)
    reinterpret_cast<uint32_t*>(lds + BG::OFF_OLDPW)[slot] = (uint32_t)(s.x & 0xFF) | ((uint32_t)(s.y & 0xFF) << 8) | ((uint32_t)(s.piece & 0xFF) << 16);
  }
  const int Q = p.queue_size;
  const int grav = p.gravity;
  // consumer side of the permutation ring (NUMPY).  Consuming an entry is pure register work: a nibble gather from the
  // ring word loaded at launch start; the generator state is not touched (see the protocol above).  g holds the state
  // only after an in-line shuffle (ring exhausted: rare).
  const int ringT = (int)(ring0 >> 56) & 0xFF;
  const int loc0 = (Hc >> 8) & 7;  // where the state after entry H - 1 is as the launch starts (kLoc*)
  const int loc_wr = loc0 == kLocBuf0 ? kLocBuf1 : kLocBuf0;  // the in-line buffer this launch may write: the one nobody reads now
  int loc = loc0;
  Hc &= 0xFF;
  bool g_valid = false, inc_loaded = false;
  BitsRng g;
  g.s_lo = g.s_hi = g.i_lo = g.i_hi = 0;
  g.uinteger = g.has32 = 0;
  g.loaded = false;
  const uint4* rng4 = reinterpret_cast<const uint4*>(p.rng + env);
  // BagRandomizer reshuffle (components/tetromino_randomizer.py:42-46, 84) / reset shuffle (:87-91) for the lanes in `need`
  auto consume_perm = [&](const bool need, const bool from_identity) {
    if (!__any(need)) return;
    const int av = ring_diff(ringT, Hc);  // entries left for this lane
    const bool from_ring = need && av > 0;
    const bool fb = need && av <= 0;
    uint32_t sig = ring_get(ring0, ring1, Hc & (kRing - 1));
    if (__any(fb)) {  // rare: ring exhausted → shuffle in-line, from the state the generator holds right now
      if (fb && !g_valid) {
        const uint4* fsrc = bp.ahead + (size_t)env * (2 * kRing) + 2 * ((Hc - 1) & (kRing - 1));
        const uint4* fsrc1 = fsrc + 1;
        if (loc == kLocRec) { fsrc = rng4; fsrc1 = rng4 + 2; }
        if (loc >= kLocBuf0) { fsrc = bp.inl + ((size_t)(loc - kLocBuf0) * p.ring_stride + env) * 2; fsrc1 = fsrc + 1; }
        const uint4 f0 = *fsrc, f1 = *fsrc1;
        g.s_lo = (uint64_t)f0.x | ((uint64_t)f0.y << 32); g.s_hi = (uint64_t)f0.z | ((uint64_t)f0.w << 32);
        g.uinteger = f1.x; g.has32 = f1.y;
      }
      if (fb && !inc_loaded) {
        const uint4 gi = rng4[1];
        g.i_lo = (uint64_t)gi.x | ((uint64_t)gi.y << 32); g.i_hi = (uint64_t)gi.z | ((uint64_t)gi.w << 32);
        inc_loaded = true;
      }
      bits_shuffle_perm(g, sig, fb);
      if (fb) { g_valid = true; loc = loc_wr; }
    }
    const uint64_t nb = bits_apply_perm(sig, from_identity ? 0x06050403020100ULL : s.bag);
    if (need) {
      s.bag = nb; s.bidx = 0; Hc = (Hc + 1) & 0xFF;
      if (from_ring) { g_valid = false; loc = kLocAhead; }
    }
  };
  // BagRandomizer.get_next_tetromino (:65-85): eager reshuffle when the 7th piece is drawn
  auto randomizer_next = [&](const bool on) -> int {
    const int t = (int)((s.bag >> (8 * (s.bidx & 7))) & 0xFF);
    if (on) s.bidx += 1;
    consume_perm(on && s.bidx >= 7, false);
    return t;
  };
  // TetrominoQueue.get_next_tetromino (components/tetromino_queue.py:35-42)
  auto queue_pop = [&](const bool on) -> int {
    const int t = (int)(s.queue & 0xFF);
    const int nxt = randomizer_next(on);
    if (on) s.queue = (s.queue >> 8) | ((uint64_t)nxt << (8 * (Q - 1)));
    return t;
  };
  // FUNCTIONAL: what this transition consumed of the key-chain look-ahead (made good behind the publish, fn_book below)
  bool fn_lock = false, fn_reset = false, fn_dirty = false;
  // ... and whether the stack reaches rows 0-5 after the lock: only then can the NEXT lock end the game (every piece spawns
  // in rows 0-1 and a piece that comes to rest on a stack below row 5 has no cell above row 2), i.e. only then is the bag a
  // reset behind that lock deals (S2) worth a sort now.  Otherwise the field is left kFnNone and sorted if ever needed.
  bool fn_hi = false;

  bool cols_dirty_any = false;
  for (int step = 0; step < nsteps; ++step) {
  TG_TIMING(if (MULTI && step == nsteps - 1 && step > 0) TG_BITS_STAMP(1);)  // (see the stream wave)
  // FUNCTIONAL: the look-ahead keys a lock of this transition will continue from — requested now (their address depends on
  // the record's position counter), consumed behind the publish (fn_book): as a dependent load there they stood in
  // front of three threefry levels while the memory system was busy with the stream waves' stores (3.3 us)
  // One transition per launch, large batches (round 3, second half): requested by EVERY lane they pull the env's whole
  // 128-byte look-ahead block in with every launch — 8.4 MB of the 46 MB a launch of 65 536 envs moves, for the one env in
  // seven that locks.  Beyond the batch whose working set the Infinity Cache holds that is what the launch waits for: there
  // the keys are requested behind the publish, by the event lanes only, and claimed in front of the first store (the logic
  // wave ends 1.4 us before its stream wave: it can wait).  98 304 envs 11.5 -> 9.9 us, 131 072 18.2 -> 16.6, 524 288
  // 57.9 -> 52.0; at 65 536 the same, at 4 096 +2 % — hence the threshold (profiles/r03/ab_functional_late_prefetch.log).
  const bool fn_prefetch_late = RULESET == TG_RULESET_FUNCTIONAL && !MULTI && hot_batch > 49152;
  u32x2 pf_c1 = u32x2{0u, 0u}, pf_cf = pf_c1, pf_r1 = pf_c1;  // (native vectors: HIP's uint2 struct went to scratch memory)
  // position n on the key chain (mod 4) and how far the look-ahead reaches (1..3 keys; tg_env_rec.reserved holds 3 - depth in
  // bits 2-3, so a zeroed field = position 0 at full depth: what bits_prepare_kernel builds)
  const int fn_n0 = (int)(s.reserved & 3u), fn_d0 = 3 - (int)((s.reserved >> 2) & 3u);
  // (an env that only resets in this transition — NEXT_STEP autoreset, the step after its end — continues with R_n)
  const bool fn_pend0 = p.autoreset == TG_AUTORESET_NEXT_STEP && (s.flags & TG_F_PENDING_RESET);
  if (RULESET == TG_RULESET_FUNCTIONAL && !fn_prefetch_late) {  // c_{n+1}, the frontier key c_{n+depth}, R_{n+1} (R_n)
    const u32x2* cs = reinterpret_cast<const u32x2*>(bp.ahead) + (size_t)env * 16;
    pf_c1 = cs[(fn_n0 + 1) & 3]; pf_cf = cs[(fn_n0 + fn_d0) & 3]; pf_r1 = cs[4 + ((fn_pend0 ? fn_n0 : fn_n0 + 1) & 3)];
  }
  int action_in = hot_act_i8 ? (int)(int8_t)(((kMaxFuse > 8 && step >= 8) ? apack1 : apack) >> (8 * (step & 7))) : action0;
  if (RULESET == TG_RULESET_NUMPY && sample_mode) {
    // uniform over the legal placements of the CURRENT piece (wrappers/grouped.py:104-122: legal = inside the frame).
    // Legal columns of a rotation form a range [lo_r, hi_r]; actions are ordered column-major, rotation-minor; the
    // k-th legal one is found by bisection on the number of legal actions in columns < c.  Same draw as grouped_kernel.
    const int n_half = s.piece == 0 ? 2 : 1;
    int lo[4], cnt[4], n_legal = 0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const uint32_t cm = (uint32_t)(col_tab[s.piece] >> (16 * ((s.rot + rr) & 3))) & 0xFFFFu;
      const uint32_t occ = ((cm & 0xFu) ? 1u : 0u) | ((cm & 0xF0u) ? 2u : 0u) | ((cm & 0xF00u) ? 4u : 0u) | ((cm & 0xF000u) ? 8u : 0u);
      const int cmin = (int)__builtin_ctz(occ), cmax = 31 - (int)__builtin_clz(occ);
      const int l = n_half - cmin > 0 ? n_half - cmin : 0;
      const int h = W - 1 + n_half - cmax < W - 1 ? W - 1 + n_half - cmax : W - 1;
      lo[rr] = l;
      cnt[rr] = h >= l ? h - l + 1 : 0;
      n_legal += cnt[rr];
    }
    int pick = 0;
    if (n_legal > 0) {
      const uint32_t gidx = p.env_index_base + env;
      int k = (int)(mix32(mix32(p.sample_seed ^ gidx * 0x9E3779B9u) + p.sample_counter + (uint32_t)step) % (uint32_t)n_legal);
      auto below = [&](const int c) -> int {  // legal actions in columns < c
        int f = 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { const int d = c - lo[rr]; f += d < 0 ? 0 : (d > cnt[rr] ? cnt[rr] : d); }
        return f;
      };
      int a = 0, b = W - 1;  // the column holding the k-th legal action: largest c with below(c) <= k
#pragma unroll 1
      while (a < b) {
        const int m = (a + b + 1) >> 1;
        if (below(m) <= k) a = m; else b = m - 1;
      }
      k -= below(a);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const bool legal_here = a >= lo[rr] && a < lo[rr] + cnt[rr];
        if (legal_here) {
          if (k == 0) pick = a * 4 + rr;
          --k;
        }
      }
    }
    action_in = pick;
    if (act_traj && live) act_traj[(size_t)step * p.act_stride + env] = (int8_t)pick;
  }
  float reward = 0.0f;
  int lines = 0;
  bool final_coll = false;
  const bool pending = p.autoreset == TG_AUTORESET_NEXT_STEP && (s.flags & TG_F_PENDING_RESET);
  bool frozen = RULESET == TG_RULESET_FUNCTIONAL && (s.flags & TG_F_GAME_OVER);
  auto colmask = [&](int piece, int rot) -> uint32_t { return (uint32_t)(col_tab[piece] >> (16 * rot)) & 0xFFFFu; };
  // Table words (all four rotations) of every piece this transition can involve — the active one, the held one, the next
  // two of the stream — fetched in ONE LDS batch up front: a lookup at the point of use is a dependent LDS round trip
  // (~130 cycles of a wave that has nothing else to issue), and there were four of them on the path to the publish.
  const int up0_ = (int)(s.queue & 0xFF);
  const int up1_ = Q >= 2 ? (int)((s.queue >> 8) & 0xFF) : (int)((s.bag >> (8 * (s.bidx & 7))) & 0xFF);
  const int pcH = s.hp < 7 ? s.hp : 0, pc0 = up0_ < 7 ? up0_ : 0, pc1 = up1_ < 7 ? up1_ : 0;
  const int pcC = s.piece;  // (the active piece as the transition starts)
  auto rot16 = [](const uint64_t t, const int rot) -> uint32_t { return (uint32_t)(t >> (16 * rot)) & 0xFFFFu; };
  // grouped action set (wrappers/grouped.py:209-269): action = column * 4 + rotations → teleport + hard drop; illegal
  // iff a cell of the rotated piece lies outside the playfield columns (collision_with_frame, :104-122)
  int action = action_in;
  bool g_illegal = false;
  if (RULESET == TG_RULESET_NUMPY && p.action_set == TG_ACTIONS_GROUPED) {
    const int col = action_in >> 2, nrot = (s.rot + (action_in & 3)) & 3;
    const int nx = col + 4 - (s.piece == 0 ? 2 : 1);  // x + padding - matrix.shape[1] // 2
    const uint32_t cg = colmask(s.piece, nrot);
    // (selects and bitwise bool algebra from here to the publish: a lane-varying `if` or a short-circuit && / || becomes an
    // exec-mask region — s_and_saveexec / s_cbranch_execz / s_or exec — which costs a wave ~45 cycles more than the same
    // work as v_cndmask, profiles/r03/salu_issue.log; there were ~25 of them on the way to the barrier)
    bool legal = (unsigned)action_in < (unsigned)(4 * W);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      legal = legal & ((((cg >> (4 * c)) & 0xFu) == 0) | ((unsigned)(nx + c - 4) < (unsigned)W));
    const bool g_ok = !pending & legal;
    g_illegal = !pending & !legal;
    s.x = g_ok ? nx : s.x;
    s.rot = g_ok ? nrot : s.rot;
    action = g_ok ? 5 : (g_illegal ? 7 : action);
    frozen = frozen | (g_illegal & (p.grouped_term != 0));
  }
  const bool stepped = live & !pending & !frozen;

  // action → tentative placement (branch-free LUT)
  uint32_t e;
  if (RULESET == TG_RULESET_NUMPY) {
    // (the handle's ActionsMapping, mappings/actions.py:12-19; the grouped action set arrives here as the canonical ids 5 / 7 of
    // the wrapper's own hard_drop / no_op: wrappers/grouped.py:234, 258)
    const uint64_t lut = p.action_set == TG_ACTIONS_GROUPED ? kActLutNumpy : p.act_lut;
    e = (uint32_t)(lut >> (8 * (action & 7))) & 0xFFu;
  } else {
    const int br = action < 0 ? 0 : (action > 6 ? 6 : action);
    e = (uint32_t)(kActLutFn >> (8 * br)) & 0xFFu;
  }
  int tp = s.piece, tv = s.val;
  int tx = s.x + (int)(e & 3u) - 1;
  int ty = s.y + (int)((e >> 2) & 1u);
  int tr = (s.rot + (int)((e >> 3) & 3u)) & 3;
  const bool check = (e & 0x20u) != 0;
  // (NUMPY: bit 6 beside a check / swap bit is not a hard drop but "this operation, then no gravity" — the entry tg_create builds for
  // an ActionsMapping whose hard_drop shares its value with a field Tetris.step tests earlier, envs/tetris.py:222-259)
  const bool hard = stepped & (RULESET == TG_RULESET_NUMPY ? (e & 0xE0u) == 0x40u : (e & 0x40u) != 0);
  // swap (envs/tetris.py:242-252, components/tetromino_holder.py:31-49)
  const bool swap_eff = RULESET == TG_RULESET_NUMPY && (stepped & ((e & 0x80u) != 0) & !(s.flags & TG_F_HAS_SWAPPED));
  // NUMPY: the visible queue is consumed lazily.  A transition takes at most two pieces from the env's stream (a hold
  // with an empty holder, then a lock's spawn); WHICH pieces is known without touching the randomizer: the head of the
  // queue, then its second entry (queue_size 1: the bag entry the first pop draws).  The pops themselves — queue shift,
  // bag draw, reshuffle from the permutation ring — are owed (`npop`) and applied after the publish, beside the stream
  // wave's tail instead of in front of it.
  int npop = 0;
  const int up0 = up0_, up1 = up1_;
  const bool swap_empty = swap_eff & (s.hp == TG_NONE);
  if (RULESET == TG_RULESET_NUMPY) {  // (no wave-uniform branch: with 64 envs per wave some lane swaps in practically every transition)
    // swap_empty: spawn_tetromino (its collision result is ignored); otherwise the held piece comes back
    tp = swap_eff ? (swap_empty ? up0 : s.hp) : tp;
    tr = swap_eff ? (swap_empty ? 0 : s.hr) : tr;
    tv = swap_eff ? (swap_empty ? up0 + 2 : s.hv) : tv;
    npop = swap_empty ? 1 : 0;
    s.hp = swap_eff ? s.piece : s.hp;
    s.hr = swap_eff ? s.rot : s.hr;
    s.hv = swap_eff ? s.val : s.hv;
    s.flags |= swap_eff ? TG_F_HAS_SWAPPED : 0;
    tx = swap_eff ? np_spawn_x<W>(tp) : tx;
    ty = swap_eff ? 0 : ty;
  }

  // the six columns around the piece: padded columns wb .. wb+5 (the current piece spans wb+1 .. wb+4, a tentative
  // one wb+dxT .. wb+dxT+3 with dxT in 0..2).  An out-of-range position (only by tg_set_state) collides everywhere.
  const int wb = (swap_eff ? tx : s.x) - 1;
  // (the row test belongs to the position the window serves: a hold swap moves the piece to row 0 — round 6's record-upload fuzz: an
  // O piece uploaded at y = H + 1, the deepest row the reference's slicing tolerates, then `swap`, was locked where it stood)
  const bool oob = ((unsigned)wb > (unsigned)(W + 2)) | (!swap_eff & ((unsigned)s.y > (unsigned)H));
  // (out of range: all six window words are the bedrock column; so are those of a dead lane, which shares its slot with the
  // last live env — the lock's unconditional write-back below must not restore what that env has just changed)
  const int wbo = (oob | !live) ? -16 : wb;
  T w[6];
  int widx[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int j = wbo + i - 3;  // padded column → LDS index: playfield column c sits at 1 + c, bedrock at 0 and W + 1
    widx[i] = j < 0 ? 0 : (j > W + 1 ? W + 1 : j);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) w[i] = colsL[widx[i]];
  // ... and the table words, requested BEHIND the window (LDS answers in order: the collision tests below start on the
  // window words while these are still on their way; requested first, their round trip stood in front of the window's)
  const uint64_t tcolC = col_tab[pcC], tcelC = cell_tab_l[pcC];
  uint64_t tcolH = 0, tcelH = 0, tcol0 = 0, tcel0 = 0, tcol1 = 0, tcel1 = 0;
  if (RULESET == TG_RULESET_NUMPY) {
    tcolH = col_tab[pcH]; tcelH = cell_tab_l[pcH];
    tcol0 = col_tab[pc0]; tcel0 = cell_tab_l[pc0];
    tcol1 = col_tab[pc1]; tcel1 = cell_tab_l[pc1];
  }
  uint64_t tcolT = tcolC, tcelT = tcelC;  // tables of the tentative piece
  if (RULESET == TG_RULESET_NUMPY) {
    tcolT = swap_eff ? (swap_empty ? tcol0 : tcolH) : tcolC;
    tcelT = swap_eff ? (swap_empty ? tcel0 : tcelH) : tcelC;
  }
  TG_BITS_STAMP(2);
  const uint32_t mC = rot16(tcolC, s.rot), mT = rot16(tcolT, tr);
  const int dxT = tx - wb;                 // 0..2 (1 after a swap)
  const uint32_t PT = mT << (4 * dxT);     // nibble i = the tentative piece's cells in window column i
  const uint32_t PC = mC << 4;
  bool cT, cT1, cC, cC1;
  {
    T hT = 0, hT1 = 0, hC = 0, hC1 = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const T a = (T)((PT >> (4 * i)) & 0xFu) << ty;
      hT |= w[i] & a;
      hT1 |= w[i] & (T)(a << 1);
      if (i >= 1 && i <= 4) {
        const T b = (T)((PC >> (4 * i)) & 0xFu) << s.y;
        hC |= w[i] & b;
        hC1 |= w[i] & (T)(b << 1);
      }
    }
    cT = hT != 0; cT1 = hT1 != 0; cC = hC != 0; cC1 = hC1 != 0;
  }
  // "piece at row r collides" for every r at once: OR over the piece's cells (k, i) of (column i >> k); the
  // reference's drop loop `while not collision(y + 1): y += 1` (envs/tetris.py:445-448, functional/core.py:243-249)
  // ends at the first blocked row below y
  // From a position that does not collide only the LOWEST cell of each piece column can be the first to hit something
  // (the cells above it move through rows the piece has just left; a column's cells are one run, static_assert above):
  // 6 terms instead of 24.  `all_cells` = the literal OR over every cell, for a start that already collides (reachable
  // only through tg_set_state).
  auto drop_row_of = [&](const uint32_t P, const int y, const bool all_cells) -> int {
    T bm = 0;
    if (all_cells) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const uint32_t n = (P >> (4 * i)) & 0xFu;
#pragma unroll
        for (int k = 0; k < 4; ++k) bm |= ((n >> k) & 1u) ? (T)(w[i] >> k) : T(0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const uint32_t n = (P >> (4 * i)) & 0xFu;
        const int kb = 31 - (int)__builtin_clz(n | 1u);  // lowest cell of the column = highest bit of the nibble
        bm |= n ? (T)(w[i] >> kb) : T(0);
      }
    }
    const int sh = y + 1 < (int)(8 * sizeof(T)) - 1 ? y + 1 : (int)(8 * sizeof(T)) - 1;
    return y + ctz_t((T)((bm | (T(1) << (8 * sizeof(T) - 1))) >> sh));
  };

  bool lock = false;
  int drop_reward = 0;
  uint32_t PF = PC;                 // window-spread column masks of the piece as it ends the move
  int frot = s.rot;                 // ... and its rotation (for the uint8 stamp)
  if (RULESET == TG_RULESET_FUNCTIONAL) {
    // functional/core.py:230-251 hard_drop: step-wise, stops at the first blocked row below y
    const int yd = drop_row_of(PC, s.y, __any(hard & cC));
    drop_reward = hard ? 2 * (yd - s.y) : 0;
    s.y = hard ? yd : s.y;
    lock = hard & ((grav != 0) | (action == 6));  // blocked below by construction → should_lock iff gravity (tetris_fn.py:247-264)
  }
  const bool moving = stepped & !(RULESET == TG_RULESET_FUNCTIONAL && hard);
  const bool accept = moving & (swap_eff | (check & !cT));
  uint64_t tcelF = tcelC;  // cell table word of the piece as it ends the move (of the spawned one after a lock)
  uint64_t tcolN = tcolC;  // column table word of the piece a lock spawns
  s.piece = accept ? tp : s.piece; s.rot = accept ? tr : s.rot; s.x = accept ? tx : s.x; s.y = accept ? ty : s.y;
  s.val = accept ? tv : s.val; PF = accept ? PT : PF; frot = accept ? tr : frot; tcelF = accept ? tcelT : tcelF;
  const bool coll_here = accept ? cT : cC;
  const bool coll_below = accept ? cT1 : cC1;
  if (RULESET == TG_RULESET_FUNCTIONAL) drop_reward = (accept & ((e & 4u) != 0)) ? 1 : drop_reward;  // soft drop (tetris_fn.py:183-190)
  {
    // envs/tetris.py:259-264 (`action != self.actions.hard_drop`: bit 6 of the entry); functional/core.py:150-182 + tetris_fn.py:242-247
    const bool gq = moving & (grav != 0) & (RULESET == TG_RULESET_NUMPY ? (e & 0x40u) == 0 : !hard);
    const bool fall = gq & !coll_below;
    s.y += fall ? 1 : 0;
    if (RULESET == TG_RULESET_NUMPY) {
      lock = moving & (hard | (gq & coll_below));
      final_coll = moving ? (coll_here & !fall) : cC;  // (!moving: the projection rule for an untouched env)
    } else {
      lock = lock | (gq & coll_below);
    }
  }

  TG_BITS_STAMP(3);
  // ---- lock path ----
  bool place = false;
  uint32_t lock_cells = 0;
  int lock_x = 0, lock_y = 0, lock_val = 0;
  bool do_clear = false;
  if (__any(lock)) {
    place = lock;
    if (RULESET == TG_RULESET_NUMPY) {  // envs/tetris.py:450-479 commit_active_tetromino
      const bool dead = lock & coll_here;  // already colliding: game over, nothing placed (:461-463; the reward is still 0)
      s.flags |= dead ? TG_F_GAME_OVER : 0;
      place = lock & !coll_here;
      {  // (no wave-uniform branch: 1 action in 8 is a hard drop, some lane always takes it)
        const int yd = drop_row_of(PF, s.y, false);  // :445-448 (only hard drops can still move; `place` = not colliding)
        s.y = (place & hard) ? yd : s.y;
      }
    }
    lock_cells = rot16(tcelF, frot);
    lock_x = s.x; lock_y = s.y;
    lock_val = RULESET == TG_RULESET_NUMPY ? s.val : s.piece + 2;  // functional/core.py:103-121 adds piece id
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      // (unconditional: a lane that places nothing writes the word back; a piece that does not collide has no cell on
      // a bedrock word, and an out-of-range one sees six of them)
      const uint32_t n = place ? (PF >> (4 * i)) & 0xFu : 0u;
      colsL[widx[i]] = (T)(w[i] | ((T)n << s.y));
    }
    wave_sync();
    // full rows: AND over the playfield columns (whole-board scan, envs/tetris.py:493-494, functional/core.py:198-201)
    T v[W];
#pragma unroll
    for (int c = 0; c < W; ++c) v[c] = colsL[1 + c];
    // spawn the next piece (needs none of the columns: it runs while the reads above are on their way)
    if (RULESET == TG_RULESET_NUMPY) {
      const int np_ = npop ? up1 : up0;  // envs/tetris.py:393-401 (TetrominoQueue.get_next_tetromino: the pop is owed)
      tcolN = place ? (npop ? tcol1 : tcol0) : tcolN; tcelF = place ? (npop ? tcel1 : tcel0) : tcelF;
      s.piece = place ? np_ : s.piece; s.rot = place ? 0 : s.rot; s.val = place ? np_ + 2 : s.val;
      s.x = place ? np_spawn_x<W>(np_) : s.x; s.y = place ? 0 : s.y;
      npop += place ? 1 : 0;
    }
    if (RULESET != TG_RULESET_NUMPY) {  // envs/tetris_fn.py:370-413 place_active_tetromino
      // the lock moves the env one position along its key chain (rng_key = split(rng_key)[0], :399: owed until after the
      // publish) and, with the bag used up, deals permutation(split(rng_key)[1]) (functional/queue.py:59-67) = Q0
      const bool refill = place & (s.bidx >= Q);  // (functional/queue.py:67: queue_index >= config.queue_size)
      const uint64_t nb = fn_expand21((uint32_t)(fnw >> 42) & 0x1FFFFFu);
      s.bag = refill ? nb : s.bag; s.bidx = refill ? 0 : s.bidx;
      s.piece = place ? (int)((s.bag >> (8 * (s.bidx & 7))) & 0xFF) : s.piece;
      s.bidx += place ? 1 : 0;
      fn_lock = fn_lock | place;
      s.rot = place ? 0 : s.rot; s.val = place ? s.piece + 2 : s.val; s.x = place ? (W + 8) / 2 - 2 : s.x; s.y = place ? 0 : s.y;
    }
    T full = PLAY;
#pragma unroll
    for (int c = 0; c < W; ++c) full &= v[c];
    if (RULESET == TG_RULESET_NUMPY) full &= (T)~brow;
    full = place ? full : T(0);
    lines = popc_t(full);
    do_clear = lines > 0;
    if (__any(do_clear)) {
      // compaction of the masks: delete the full rows top-down (rows above move down by one each time);
      // NUMPY (envs/tetris.py:500-510): vacated top rows are empty; FUNCTIONAL (functional/core.py:203-214): the
      // gather index -H of a cleared row wraps to row 0 of the pre-compaction board → copies of old row 0
#pragma unroll
      for (int c = 0; c < W; ++c) {
        T x = (T)(v[c] & PLAY);
        const T top = (RULESET == TG_RULESET_FUNCTIONAL && (x & 1)) ? (T)((T(1) << lines) - 1) : T(0);
        T f = full;
        while (f) {
          const int rr = ctz_t(f);
          f &= (T)(f - 1);
          const T lowm = (T)((T(1) << rr) - 1);
          x = (T)((x & (T)~((lowm << 1) | 1)) | ((x & lowm) << 1));
        }
        x |= top;
        if (do_clear) { v[c] = (T)(x | FLOOR); colsL[1 + c] = v[c]; }
      }
      if (RULESET == TG_RULESET_NUMPY && __any(brow != 0)) {  // the bedrock rows move down with the rows above a cleared one
        T x = (T)(brow & PLAY);
        T f = full;
        while (f) {
          const int rr = ctz_t(f);
          f &= (T)(f - 1);
          const T lowm = (T)((T(1) << rr) - 1);
          x = (T)((x & (T)~((lowm << 1) | 1)) | ((x & lowm) << 1));
        }
        brow = do_clear ? x : brow;
      }
    }
    s.flags &= place ? ~TG_F_FULLROW : ~0;
    if (RULESET == TG_RULESET_FUNCTIONAL) {  // (see fn_hi; p.fn_lazy & 3: 0 = as described, 1 = never sort ahead, 2 = always: tests)
      T orc = 0;
#pragma unroll
      for (int c = 0; c < W; ++c) orc |= v[c];
      fn_hi = place & ((((orc & T(0x3F)) != 0) & ((p.fn_lazy & 3) != 1)) | ((p.fn_lazy & 3) == 2));
    }
    // collision of the new piece at its spawn position (rows 0..3 of the columns it spans; static columns)
    bool sc;
    {
      constexpr int X0 = (W + 8) / 2 - 2;  // I (and every functional piece) spawns at X0, the others at X0 + 1
      const uint32_t ms = RULESET == TG_RULESET_NUMPY ? rot16(tcolN, 0) : colmask(s.piece, 0);
      const bool at0 = s.x == X0;
      T hs = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const T col = at0 ? v[X0 - 4 + c] : v[X0 - 3 + c];
        hs |= col & (T)((ms >> (4 * c)) & 0xFu);
      }
      sc = hs != 0;
    }
    if (RULESET == TG_RULESET_NUMPY) {
      // score + rewards.alife; rewards.game_over when the piece already collided or the spawn collides (envs/tetris.py:461-474)
      reward = lock ? ((place & !sc) ? (float)(lines * lines * W) + p.r_alife : p.r_game_over) : 0.0f;
      s.flags = place ? ((s.flags & ~(TG_F_GAME_OVER | TG_F_HAS_SWAPPED)) | (sc ? TG_F_GAME_OVER : 0)) : s.flags;
      final_coll = place ? sc : final_coll;
    } else {
      s.flags |= (place & sc) ? TG_F_GAME_OVER : 0;
    }
  }
  if (RULESET == TG_RULESET_FUNCTIONAL) {
    const int lock_reward = lock ? (lines == 4 ? 800 : (lines > 0 ? lines * 200 - 100 : 0)) : 0;  // core.py:124-146
    const float old = s.score;
    float t = old + (float)drop_reward;
    t = t + (float)lock_reward;
    s.score = stepped ? t : old;
    reward = stepped ? t - old : reward;  // tetris_fn.py:312
  }

  TG_BITS_STAMP(4);
  // ---- termination / autoreset ----
  reward = g_illegal ? p.r_invalid : reward;  // rewards.invalid_action (wrappers/grouped.py:238)
  const int terminated = (!pending & (((s.flags & TG_F_GAME_OVER) != 0) | (g_illegal & (p.grouped_term != 0)))) ? 1 : 0;
  const bool do_reset = live & (pending | ((terminated != 0) & (p.autoreset == TG_AUTORESET_SAME_STEP)));
  s.flags |= ((terminated != 0) & (p.autoreset == TG_AUTORESET_NEXT_STEP) & !pending) ? TG_F_PENDING_RESET : 0;
  if (FINAL && final_obs && live) {  // the terminal piece, before the reset replaces it (TG_FLAG_FINAL_OBS)
    const bool fin = do_reset && terminated;
    const bool showf = fin && (RULESET == TG_RULESET_NUMPY ? !final_coll : false);
    postF[(step & 1) * EPW + slot] =
        u32x2{lane_cells(cell_tab_l, s.piece, s.rot) | ((uint32_t)(s.val & 0xFF) << 16) | ((showf ? 1u : 0u) << 24) | ((fin ? 1u : 0u) << 25),
              (uint32_t)(s.x & 0xFF) | ((uint32_t)(s.y & 0xFF) << 8)};
  }
  // The first piece of a new episode, for the publish.  NUMPY (envs/tetris.py:274-307, tetromino_randomizer.py:87-91): the
  // reset shuffles a fresh identity bag and draws its element 0 = nibble 0 of the permutation the reset will consume —
  // the ring entry behind those this transition's owed pops consume (a pop wraps the bag at most once).  Everything else
  // of the reset (queue, bag, holder, flags) is bookkeeping after the publish.  Ring too short for that (rare): the
  // bookkeeping runs first (`early_book`, wave-uniform) and the publish reads its result.
  bool early_book = false;
  if (__any(do_reset)) {
    if (RULESET == TG_RULESET_NUMPY) {
      const int Hr = Hc + ((s.bidx + npop >= 7) ? 1 : 0);
      const bool have = ring_diff(ringT, Hr) > 0;
      early_book = __any(do_reset & !have);
      s.piece = do_reset ? (int)(ring_get(ring0, ring1, Hr & (kRing - 1)) & 0xFu) : s.piece;
      s.x = do_reset ? np_spawn_x<W>(s.piece) : s.x;
      s.rot = do_reset ? 0 : s.rot; s.y = do_reset ? 0 : s.y; s.val = do_reset ? s.piece + 2 : s.val;
      final_coll = final_coll & !do_reset;
    } else {  // envs/tetris_fn.py:318-367
      if (do_reset) {
#pragma unroll
        for (int c = 0; c < W; ++c) colsL[1 + c] = FLOOR;
      }
      // the reset deals permutation(split(key)[0]) and continues with split(key)[1] (owed until after the publish): S1 at
      // the env's position, S2 behind a lock of this transition
      uint32_t f21 = (uint32_t)(fnw >> (fn_lock ? 21 : 0)) & 0x1FFFFFu;
      const bool unsorted = do_reset & (f21 == kFnNone);
      if (__any(unsorted)) {
        // The bag was not sorted ahead (fn_hi: the stack was low when the look-ahead was made good).  By the argument there
        // this does not happen in play; it does after the test knob p.fn_lazy = 1 and would after a rule change — sort now:
        // R_{n+1} (a reset alone) or R_{n+2} (behind a lock), derived from the env's current key (the look-ahead block may
        // not reach that far: its depth grows back over the idle transitions behind a reset).  The key may have been
        // written by an earlier transition of this launch: those stores are made visible first.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        const uint2 ck = reinterpret_cast<const uint2*>(p.rng + env)[5];  // c_n = State.rng_key (fn_book keeps it current)
        uint32_t x0, x1, y0, y1, r0_, r1_;
        threefry2x32(ck.x, ck.y, 0u, 0u, x0, x1);                                   // c_{n+1}
        threefry2x32(x0, x1, 0u, 0u, y0, y1);                                       // c_{n+2}
        threefry2x32(fn_lock ? y0 : x0, fn_lock ? y1 : x1, 0u, 1u, r0_, r1_);       // R_{n+2} | R_{n+1}
        const uint32_t srt = fn_sort7(r0_, r1_, Q);
        f21 = unsorted ? srt : f21;
      }
      const uint64_t nb = fn_expand21(f21);
      s.bag = do_reset ? nb : s.bag; s.queue = do_reset ? 0 : s.queue; s.piece = do_reset ? (int)(nb & 0xFF) : s.piece;
      s.bidx = do_reset ? 1 : s.bidx;
      fn_reset = fn_reset | do_reset;
      s.x = do_reset ? (W + 8) / 2 - 2 : s.x;
      s.rot = do_reset ? 0 : s.rot; s.y = do_reset ? 0 : s.y; s.val = do_reset ? s.piece + 2 : s.val;
      s.hp = do_reset ? TG_NONE : s.hp; s.hr = do_reset ? 0 : s.hr; s.hv = do_reset ? 0 : s.hv;
      s.flags = do_reset ? 0 : s.flags;
      s.score = do_reset ? 0.0f : s.score;
      final_coll = final_coll & !do_reset;
    }
  }
  cols_dirty_any = cols_dirty_any | (live & (place | do_reset));
  if (rgb) early_book = true;  // the image shows the queue and the holder as they are AFTER the transition: bookkeeping first
  // The publish exists twice in the code (it is short), the bookkeeping once: normally publish → barrier → bookkeeping
  // (beside the stream wave's work), with early_book (wave-uniform) the other way round.
  auto publish = [&]() {
    if (rgb && live)
      post2[slot] = make_uint4((uint32_t)s.queue, (uint32_t)((s.queue >> 32) & 0x00FFFFFFu) | ((uint32_t)(s.hp & 0xFF) << 24),
                               (uint32_t)(s.hr & 0xFF) | ((uint32_t)(s.hv & 0xFF) << 8), 0u);
    // the payload work of this lane's env for the stream wave
    const bool show = live & (RULESET == TG_RULESET_NUMPY ? !final_coll : !(s.flags & TG_F_GAME_OVER));
    uint32_t ocells;
    if (RULESET == TG_RULESET_NUMPY) {
      // the piece's table word is already here — unless a reset has just drawn the piece: a new episode starts at
      // rotation 0, whose cells are compile-time constants (no table round trip in front of the barrier)
      ocells = rot16(tcelF, s.rot);
      ocells = do_reset ? (uint32_t)((s.piece < 4 ? kCells0Lo : kCells0Hi) >> (16 * (s.piece & 3))) & 0xFFFFu : ocells;
    } else {
      ocells = lane_cells(cell_tab_l, s.piece, s.rot);
    }
    if (live) post[(step & 1) * EPW + slot] = make_uint4(lock_cells | ((uint32_t)(lock_val & 0xFF) << 16) | ((place ? 1u : 0u) << 24) |
                                ((do_clear ? 1u : 0u) << 25) | ((do_reset ? 1u : 0u) << 26),
                            (uint32_t)(lock_x & 0xFF) | ((uint32_t)(lock_y & 0xFF) << 8),
                            ocells | ((uint32_t)(s.val & 0xFF) << 16) | ((show ? 1u : 0u) << 24),
                            (uint32_t)(s.x & 0xFF) | ((uint32_t)(s.y & 0xFF) << 8) | ((uint32_t)(s.piece & 0xFF) << 16));
    TG_BITS_STAMP(5);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
  };
  auto bookkeeping = [&]() {
    // ---- bookkeeping: the owed pops in order (TetrominoQueue.get_next_tetromino, components/tetromino_queue.py:35-42),
    // then the reset's randomizer / queue / holder state ----
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
      const bool on = k < npop;
      if (!__any(on)) break;
      (void)queue_pop(on);
    }
    if (__any(do_reset)) {
      if (do_reset) {
#pragma unroll
        for (int c = 0; c < W; ++c) colsL[1 + c] = FLOOR;
      }
      consume_perm(do_reset, true);  // tetromino_queue.py:23-33, tetromino_randomizer.py:87-91
      if (Q <= 5) {
        // Q draws fill the queue, the pop takes its head and draws one more: Q + 1 <= 6 draws never wrap the fresh bag
        if (do_reset) {
          s.piece = (int)(s.bag & 0xFF);
          s.queue = (s.bag >> 8) & ((1ULL << (8 * Q)) - 1ULL);
          s.bidx = Q + 1;
        }
      } else {
        if (do_reset) s.queue = 0;
        for (int i = 0; i < Q; ++i) {
          const int t = randomizer_next(do_reset);
          if (do_reset) s.queue |= (uint64_t)t << (8 * i);
        }
        const int first = queue_pop(do_reset);
        if (do_reset) s.piece = first;
      }
      if (do_reset) {
        s.x = np_spawn_x<W>(s.piece);
        s.rot = 0; s.y = 0; s.val = s.piece + 2;
        s.hp = TG_NONE; s.hr = 0; s.hv = 0;
        s.flags = 0;
        s.score = 0.0f;
        brow = 0;  // (a fresh board)
      }
    }
  };
  // FUNCTIONAL: make good what this transition consumed of the key-chain look-ahead (protocol at the top of this file),
  // beside the stream wave's work.  ONE threefry level per lane and transition (round 4; round 3 ran three levels for a reset):
  //   a lock at n        advances the frontier  c_{n+d} -> c_{n+d+1}, R_{n+d}          (position n + 1, depth d as before)
  //   a reset            starts a chain          K -> c_1, R_0                           (position 0, depth 1)
  //   neither, d < 3     catches up              c_{n+d} -> c_{n+d+1}, R_{n+d}           (depth d + 1)
  // — the levels a fresh chain lacks are made in the idle transitions behind the reset, by lanes that would sit out the
  // wave's pass anyway (a pass costs the wave the same however few lanes want it, and some lane locks in practically every
  // transition).  Nothing needs them earlier: the next lock takes c_{n+1}; the bag a refill deals (Q0', six locks away)
  // needs R_{n+1}, which a depth-1 lock makes in this very pass; the bag a reset behind the next lock deals (S2') is
  // sorted ahead only at full depth — a tall stack within two transitions of a reset does not occur, and the in-line sort
  // where the bag is dealt covers whatever does.
  auto fn_book = [&]() {
    const bool ev = fn_lock || fn_reset;
    const bool cu = !ev && fn_d0 < 3 && !(p.fn_lazy & 4);  // (test knob: bit 2 = never catch up — the chain stays one key deep)
    if (!__any(ev | cu)) return;
    const int n = fn_n0, d = fn_d0;
    const bool lk = fn_lock && !fn_reset;           // stays on its chain: position n + 1
    const bool needq = lk && s.bidx >= Q;           // ... whose lock will refill the bag
    u32x2 kb = pf_r1;
    // The key a reset continues with: R_n — or R_{n+1} behind a lock of this transition (SAME_STEP).  The prefetch took R_n for
    // an env that was waiting for its reset (NEXT_STEP) and R_{n+1} otherwise; the two cases it cannot serve are rare and
    // wave-uniform branches: a reset without a lock of an env that was not waiting (a game-over state uploaded under SAME_STEP:
    // R_n is fetched now) and a reset behind a lock at depth 1 (R_{n+1} = f1(c_{n+1}) is not in the block yet).
    {
      const bool want_rn = fn_reset && !fn_lock && !fn_pend0;
      if (__any(want_rn)) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        const u32x2 rn = (reinterpret_cast<const u32x2*>(bp.ahead) + (size_t)env * 16)[4 + (n & 3)];
        kb = want_rn ? rn : kb;
      }
      const bool deep = fn_reset && fn_lock && d < 2;
      if (__any(deep)) {
        uint32_t q0, q1;
        threefry2x32(pf_c1.x, pf_c1.y, 0u, 1u, q0, q1);
        kb = deep ? u32x2{q0, q1} : kb;
      }
    }
    const u32x2 ka = fn_reset ? kb : pf_cf, knew = fn_reset ? kb : pf_c1;
    // (The threefry blocks as a dense task list through LDS — one block for the wave's ~27 tasks instead of two or three — were
    // built in round 3, bit-identical, and not kept: 8.07 -> 7.96 us per launch at 65 536 envs but +2 % per fused transition and
    // at 4 096 envs, where the LDS exchanges are latency on the chain the launch waits for.  profiles/r03/ab_functional_dense.log)
    uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0, z0 = 0, z1 = 0;
    threefry2x32(ka.x, ka.y, 0u, 0u, a0, a1);
    threefry2x32(ka.x, ka.y, 0u, 1u, b0, b1);
    if (__any(needq)) {  // Q0' = sort7(f1(R_{n+1})); at depth 1 R_{n+1} is what this pass has just made
      const bool shallow = d < 2;
      threefry2x32(shallow ? b0 : kb.x, shallow ? b1 : kb.y, 0u, 1u, z0, z1);
    }
    const u32x2 c1 = u32x2{a0, a1}, r0 = u32x2{b0, b1};
    // The sorts, seven lanes each.  Round 4: only the bags the NEXT transition can deal are sorted now — S2' = sort7(R_{n+3}) of a
    // lock that leaves the stack within reach of the spawn rows (fn_hi), Q0' = sort7(f1(R_{n+1})) when the next lock refills;
    // a reset sorts nothing (its board is empty: S1 / S2 of the new chain stay kFnNone until a lock with a tall stack makes
    // them).  Round 3 sorted S2' for every lock and S1, S2 for every reset: ~15 sorts per wave and transition, two trips'
    // worth of threefry streams; now ~6 under the random policy, one stream.
    const bool needs2 = lk && fn_hi && d == 3;  // (b = R_{n+3} only at full depth)
    const uint64_t mA = __ballot(needs2), mB = __ballot(needq);
    const int nA = __popcll(mA), nt = nA + __popcll(mB);
    const int qA = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mA, 0u));
    const int qB = nA + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mB, 0u));
    uint64_t p1 = kFnNone, p2 = 0;
    if (nt > 0) {
    uint2* xk = reinterpret_cast<uint2*>(lds + (FINAL ? BG::OFF_FNX : BG::OFF_FNX0));
    uint32_t* xo = reinterpret_cast<uint32_t*>(lds + (FINAL ? BG::OFF_FNX : BG::OFF_FNX0) + 128 * 8);
    if (needs2) xk[qA] = make_uint2(b0, b1);
    if (needq) xk[qB] = make_uint2(z0, z1);
    wave_sync();
    const int j = lane & 7;
    // stable rank of element j among the 7 of its 8-lane group (jax sort_key_val on 32 random bits; ties by index) → the
    // group's permutation as seven 3-bit fields.  The other lanes' sort keys come by DPP row rotations (a rotation by r
    // within the 16-lane row delivers lane - r; for the lanes whose source falls into the neighbouring group the rotation by
    // r + 8 delivers the same position of their own group): VALU latency instead of six LDS-crossbar round trips
    // (ds_bpermute: 1 200 of the sort's 2 200 cycles).  The idle eighth lane holds the maximum key: it ranks behind everybody.
    auto group_perm = [&](const uint32_t bits_in) -> uint32_t {
      const uint32_t bits = j < Q ? bits_in : 0xFFFFFFFFu;  // (Q = the bag length: elements Q..7 of the group rank behind everybody)
      int rank = 0;
#define TG_ROR(r) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bits, 0x120 + (r), 0xF, 0xF, true)  // row_ror:r
#define TG_OTHER(r)                                                                    \
      {                                                                                \
        const uint32_t near_ = TG_ROR(r), far_ = TG_ROR((r) + 8);                      \
        const bool same = j >= (r);                       /* lane - r stays in the group */ \
        const uint32_t ob = same ? near_ : far_;                                       \
        const int oj = same ? j - (r) : j - (r) + 8;                                   \
        rank += (ob < bits || (ob == bits && oj < j)) ? 1 : 0;                         \
      }
      TG_OTHER(1) TG_OTHER(2) TG_OTHER(3) TG_OTHER(4) TG_OTHER(5) TG_OTHER(6) TG_OTHER(7)
#undef TG_OTHER
#undef TG_ROR
      uint32_t pm = j < Q ? (uint32_t)j << (3 * rank) : 0u;
      pm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pm, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
      pm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pm, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
      pm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pm, 0x141, 0xF, 0xF, true);  // row_half_mirror: the other quad of the 8
      return pm;
    };
    for (int t0 = 0; t0 < nt; t0 += 16) {  // 16 sorts per trip: two independent threefry streams per lane
      const int ta = t0 + (lane >> 3), tb = ta + 8;
      const uint2 xa = xk[ta < nt ? ta : 0], xb = xk[tb < nt ? tb : 0];
      uint32_t u0, u1, v0 = 0, v1 = 0;
      threefry2x32(xa.x, xa.y, 0u, (uint32_t)j, u0, u1);
      const uint32_t pa = group_perm(u0 ^ u1);
      if (j == 0 && ta < nt) xo[ta] = pa;
      if (nt - t0 <= 8) break;  // (eight sorts or fewer left: the second stream would be ~130 instructions for nobody)
      threefry2x32(xb.x, xb.y, 0u, (uint32_t)j, v0, v1);
      const uint32_t pb = group_perm(v0 ^ v1);
      if (j == 0 && tb < nt) xo[tb] = pb;
    }
    wave_sync();
    const uint32_t s1 = xo[needs2 ? qA : 0], s2 = xo[needq ? qB : 0];
    p1 = needs2 ? s1 : kFnNone; p2 = s2;
    wave_sync();
    }
    if (lk) {  // S2 -> S1, the new S2 (or none), the refill bag
      fnw = ((fnw >> 21) & 0x1FFFFFULL) | (p1 << 21) | (needq ? p2 << 42 : 0ULL);
      s.reserved = (s.reserved & ~3u) | (uint32_t)((n + 1) & 3);
    }
    if (fn_reset) {  // a fresh chain, renumbered from position 0, one key deep; nothing sorted
      fnw = (uint64_t)kFnNone | ((uint64_t)kFnNone << 21);
      s.reserved = (s.reserved & ~15u) | (2u << 2);
    }
    if (cu) s.reserved = (s.reserved & ~(3u << 2)) | ((uint32_t)(3 - (d + 1)) << 2);
    // The stores (look-ahead block: c[4] | r[4], 8 bytes each; State.rng_key) as raw-buffer stores with the predicate in the
    // offset: no exec-mask region around any of them (st_state_if).  c_{n+d+1} → c[(n + d + 1) & 3], R_{n+d} → r[(n + d) & 3];
    // a reset's c_1 → c[1], R_0 → r[0]: the same two instructions.
    {
      const __amdgpu_buffer_rsrc_t wla = st_window(bp.ahead, (uint32_t)hot_batch * 128u);
      const uint32_t eo = env * 128u;
      const bool anyl = (ev || cu) && live, evl = ev && live;
      st_b64_if(wla, eo + 8u * (uint32_t)(fn_reset ? 1 : ((n + d + 1) & 3)), c1, anyl);
      st_b64_if(wla, eo + 32u + 8u * (uint32_t)(fn_reset ? 0 : ((n + d) & 3)), r0, anyl);
      st_b64_if(st_window(p.rng, (uint32_t)hot_batch * (uint32_t)sizeof(tg_rng_rec)), env * (uint32_t)sizeof(tg_rng_rec) + 40u,
                knew, evl);  // State.rng_key
      fn_dirty = fn_dirty | evl;
    }
    fn_lock = false; fn_reset = false; fn_hi = false;
  };
  // obs["active_tetromino_mask"] of the workgroup's envs, GD at a time (dict_mask_group), into TG_BUF_OBS / the trajectory slot
  auto dict_masks = [&]() {
    const size_t Bt = p.act_stride;  // the handle's batch
    uint8_t* d0 = obs_traj ? obs_traj + (size_t)step * p.obs_traj_stride : p.obs;
    uint8_t* d1 = (step == nsteps - 1 && obs_traj) ? p.obs : nullptr;
    constexpr int GD = BG::GD;
    uint8_t* grpm = lds + BG::OFF_DICT + GD * BG::PC;  // all zero between groups (zeroed at launch start, un-stamped by every group)
    for (int g0 = 0; g0 < EPW; g0 += GD) {
      const uint32_t nlg = nlive > (uint32_t)g0 ? nlive - (uint32_t)g0 : 0u;
      dict_mask_group<W, H, GD>(grpm, post + (step & 1) * EPW + g0, nlg, lane, d0 + Bt * BG::PC + ((size_t)base_env + g0) * BG::PC,
                                d1 ? d1 + Bt * BG::PC + ((size_t)base_env + g0) * BG::PC : nullptr);
    }
  };
  const bool book_first = RULESET == TG_RULESET_NUMPY && early_book;
  if constexpr (MULTI) {
    // (the kernels with the transition loop are at their register budget: one publish site, two passes)
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      if (RULESET == TG_RULESET_NUMPY && (pass == 0) == book_first) bookkeeping();
      if (pass == 0) publish();
    }
  } else {
    if (!book_first) publish();
    // (the dict observation's mask array depends on nothing behind the publish — but stored HERE, in front of the bookkeeping, the
    // launch takes 23.5 instead of 17.3 us: the bookkeeping's loads then wait behind 28 store instructions per lane, vmcnt counts
    // both.  profiles/r04/ab_dict_mask_early_rejected.log)
    if (RULESET == TG_RULESET_NUMPY) bookkeeping();
    if (book_first) publish();
  }
  // (every load result the rest of the transition uses is claimed HERE, in front of the first store behind the publish: on
  // gfx9 a wait for a load with stores pending is a wait for the stores — vmcnt counts both, and they complete out of order)
  if (fn_prefetch_late && (fn_lock | fn_reset | (fn_d0 < 3))) {
    const u32x2* cs = reinterpret_cast<const u32x2*>(bp.ahead) + (size_t)env * 16;
    pf_c1 = cs[(fn_n0 + 1) & 3]; pf_cf = cs[(fn_n0 + fn_d0) & 3]; pf_r1 = cs[4 + ((fn_pend0 ? fn_n0 : fn_n0 + 1) & 3)];
  }
  if (RULESET == TG_RULESET_FUNCTIONAL) asm volatile("" : "+v"(pf_c1), "+v"(pf_cf), "+v"(pf_r1));
  asm volatile("" : "+v"(ep.ret), "+v"(ep.len), "+v"(ep.last_ret), "+v"(ep.last_len));
  if (pending) {  // the NEXT_STEP reset step belongs to no episode (gymnasium RecordEpisodeStatistics)
    ep.ret = 0.0f; ep.len = 0;
  } else {
    ep.ret += reward;
    ep.len += 1;
    if (terminated) { ep.last_ret = ep.ret; ep.last_len = ep.len; ep.ret = 0.0f; ep.len = 0; }
  }
  // tg_step_out: terminated | lines << 8 | flags << 16 (TG_OUT_COMMITTED: commit_active_tetromino ran; TG_OUT_ILLEGAL)
  const uint32_t out_word = (uint32_t)terminated | ((uint32_t)lines << 8) |
                            (RULESET == TG_RULESET_NUMPY ? ((lock ? (uint32_t)TG_OUT_COMMITTED << 16 : 0u) | (g_illegal ? (uint32_t)TG_OUT_ILLEGAL << 16 : 0u)) : 0u);
  if (step == nsteps - 1 && live)  // the outputs of the LAST transition remain in the buffers
    st_stream(reinterpret_cast<u32x2*>(p.out + env), u32x2{__float_as_uint(reward), out_word});
  if (out_traj && live)
    st_stream(reinterpret_cast<u32x2*>(out_traj + (size_t)step * p.act_stride + env),
              u32x2{__float_as_uint(reward), out_word});
  if (RULESET == TG_RULESET_FUNCTIONAL) {
    // (behind everything that waits for a load of the launch's start — the episode record above: a wait for it behind
    // fn_book's stores is a wait for those stores, a round trip through a memory system busy with the stream waves' tail)
    if (!MULTI) __builtin_amdgcn_s_setprio(0);  // the look-ahead is nobody's critical path: the stream wave's tail goes first
    fn_book();
  }
  if (rgb) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();  // the stream wave has applied the locks / clears / resets and drawn the pieces
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    constexpr size_t IMG = (size_t)BG::NPX * 3;
    const bool lastt = step == nsteps - 1;
    uint8_t* dst0 = (obs_traj ? obs_traj + (size_t)step * p.obs_traj_stride : p.obs) + (size_t)base_env * IMG;
    uint8_t* dst1 = (lastt && obs_traj) ? p.obs + (size_t)base_env * IMG : nullptr;
    rgb_pack<W, H, BG::GR>(lds, post2, cell_tab + 24, lut_l, lds + BG::OFF_RGB + EPW * 16, 0, ELOG, nlive, lane, dst0, dst1);
    if (!lastt) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
  }
  if (dict) {
    // the other three arrays of the dict observation (envs/tetris.py:572-615), while the stream wave packs the board:
    // active_tetromino_mask = bounding box of the native matrix, holder = ones when empty else the held matrix,
    // queue strip = next pieces at rotation 0.  Layout: board[B][Hp][Wp] | mask[B][Hp][Wp] | holder[B][4][4] | queue[B][4][4Q]
    const size_t Bt = p.act_stride;  // the handle's batch
    uint8_t* dsts[2] = {obs_traj ? obs_traj + (size_t)step * p.obs_traj_stride : p.obs,
                        (step == nsteps - 1 && obs_traj) ? p.obs : nullptr};
    // (storing only the chunks the old / the new bounding box covers — dict_mask_group_incr, TG_BUF_OBS keeps its content — was
    // measured in round 3 and rejected: 18.0 vs 17.1 us, profiles/r03/ab_dict_incremental_mask.log)
    if constexpr (!bits_mask_wave<W, H>(FEAT)) dict_masks();
    // holder and queue strips change only with a hold, a spawn or a reset: TG_BUF_OBS keeps the strips of the previous
    // transition (tg_reset / tg_observe / tg_set_state re-pack it completely), so only those envs store them; a
    // trajectory slot is fresh memory and gets every env's strips
    const bool hq = live && (obs_traj != nullptr || npop > 0 || swap_eff || do_reset);
    if (__any(hq)) {
      auto spread = [](const uint32_t nib, const uint32_t val) -> uint32_t {  // 4 cells of a matrix row → 4 bytes
        return ((nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21)) * val;
      };
      const uint64_t* row_tab = col_tab + 16;  // per piece: 4 rotations x 16-bit row-major masks
      uint32_t hm = 0, qm[7];
      if (s.hp != TG_NONE) hm = (uint32_t)(row_tab[s.hp] >> (16 * s.hr)) & 0xFFFFu;
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const int t = (int)((s.queue >> (8 * q)) & 0xFF);
        qm[q] = q < Q ? (uint32_t)row_tab[t < 7 ? t : 0] & 0xFFFFu : 0u;
      }
      u32x4 hv4 = u32x4{0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
      if (s.hp != TG_NONE)
        hv4 = u32x4{spread(hm & 0xFu, (uint32_t)s.hv), spread((hm >> 4) & 0xFu, (uint32_t)s.hv),
                    spread((hm >> 8) & 0xFu, (uint32_t)s.hv), spread((hm >> 12) & 0xFu, (uint32_t)s.hv)};
      if (hq) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          if (!dsts[d]) continue;
          uint8_t* o_holder = dsts[d] + 2 * Bt * BG::PC + (size_t)env * 16;
          uint8_t* o_queue = dsts[d] + 2 * Bt * BG::PC + Bt * 16 + (size_t)env * 16 * Q;
          st_stream(reinterpret_cast<u32x4*>(o_holder), hv4);
          if (Q == 4) {  // one 16-byte store per strip row
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const uint32_t t0 = (uint32_t)(s.queue & 0xFF), t1 = (uint32_t)((s.queue >> 8) & 0xFF),
                             t2 = (uint32_t)((s.queue >> 16) & 0xFF), t3 = (uint32_t)((s.queue >> 24) & 0xFF);
              st_stream(reinterpret_cast<u32x4*>(o_queue + rr * 16),
                        u32x4{spread((qm[0] >> (4 * rr)) & 0xFu, t0 + 2u), spread((qm[1] >> (4 * rr)) & 0xFu, t1 + 2u),
                              spread((qm[2] >> (4 * rr)) & 0xFu, t2 + 2u), spread((qm[3] >> (4 * rr)) & 0xFu, t3 + 2u)});
            }
          } else {
            for (int q = 0; q < Q; ++q) {
              const uint32_t t = (uint32_t)((s.queue >> (8 * q)) & 0xFF);
#pragma unroll
              for (int rr = 0; rr < 4; ++rr)
                *reinterpret_cast<uint32_t*>(o_queue + rr * 4 * Q + 4 * q) = spread((qm[q] >> (4 * rr)) & 0xFu, t + 2u);
            }
          }
        }
      }
    }
  }
  }  // transitions
  if (RULESET == TG_RULESET_NUMPY) __builtin_amdgcn_s_setprio(0);
  // ---- D: per-env stores: first record half, episode statistics, column masks ----
  const bool cols_dirty = cols_dirty_any;
  if (RULESET == TG_RULESET_NUMPY) { s.score = __uint_as_float((uint32_t)brow); s.reserved = (uint32_t)((uint64_t)brow >> 32); }
  uint4 oa, ob;
  lane_pack(s, oa, ob);
  uint4* rec_out = reinterpret_cast<uint4*>(hot_rec) + (size_t)env * 2;
  // Records as whole cache lines (one-transition kernels; -0.15 ... -0.25 us per launch, profiles/r02/ab_rec_lines.log):
  // lane l of store s writes the 16-byte chunk l + 64 s of the workgroup's record block (= half (chunk & 1) of env
  // chunk >> 1), fetched from the owning lane by ds_bpermute — stores that cover full lines instead of two stores that
  // each cover half of every line (and the second half is no longer conditional).
  const bool rec_lines = !MULTI && nlive == (uint32_t)EPW;
  constexpr int REC_CHUNKS = 2 * EPW, REC_ST = (REC_CHUNKS + 63) / 64;
  if (rec_lines) {
    u32x4* rblk = reinterpret_cast<u32x4*>(hot_rec) + (size_t)base_env * 2;
#pragma unroll
    for (int st = 0; st < REC_ST; ++st) {
      const int chunk = lane + 64 * st;
      const int src = ((chunk < REC_CHUNKS ? chunk : REC_CHUNKS - 1) >> 1) << 2;
      const bool hi = (lane & 1) != 0;
      const uint32_t ax = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)oa.x), bx = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)ob.x);
      const uint32_t ay = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)oa.y), by = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)ob.y);
      const uint32_t az = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)oa.z), bz = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)ob.z);
      const uint32_t aw = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)oa.w), bw = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)ob.w);
      if (chunk < REC_CHUNKS) TG_ST_STATE(rblk + chunk, (u32x4{hi ? bx : ax, hi ? by : ay, hi ? bz : az, hi ? bw : aw}));
    }
  }
  if (live && !rec_lines) TG_ST_STATE(reinterpret_cast<u32x4*>(rec_out), (u32x4{oa.x, oa.y, oa.z, oa.w}));
  if (live && p.episode) p.episode[env] = ep;
  if (__any(cols_dirty)) {
    T nv[W];
#pragma unroll
    for (int c = 0; c < W; ++c) nv[c] = (T)(colsL[1 + c] & PLAY);
    if (cols_dirty) {
      uint2* gc = reinterpret_cast<uint2*>(reinterpret_cast<T*>(hot_cols) + (size_t)env * BG::WC);
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int k = 0; k < W / 2; ++k) TG_ST_COLS(reinterpret_cast<u32x2*>(gc + k), (u32x2{(uint32_t)nv[2 * k], (uint32_t)nv[2 * k + 1]}));
        if constexpr (W % 2 == 1) TG_ST_COLS(reinterpret_cast<u32x2*>(gc + W / 2), (u32x2{(uint32_t)nv[W - 1], 0u}));
      } else {
#pragma unroll
        for (int c = 0; c < W; ++c) TG_ST_COLS(reinterpret_cast<u32x2*>(gc + c), (u32x2{(uint32_t)nv[c], (uint32_t)((uint64_t)nv[c] >> 32)}));
      }
    }
  }

  TG_BITS_STAMP(6);
  // ---- E: consumed counter; the generator state only after an in-line shuffle (otherwise `ahead` holds it) ----
  if (RULESET == TG_RULESET_NUMPY) {
    if (live) bp.hbuf_wr[env] = (uint32_t)(Hc | (loc << 8));
    if (live && g_valid) {  // (loc == loc_wr)
      uint4* io = bp.inl + ((size_t)(loc_wr - kLocBuf0) * p.ring_stride + env) * 2;
      io[0] = make_uint4((uint32_t)g.s_lo, (uint32_t)(g.s_lo >> 32), (uint32_t)g.s_hi, (uint32_t)(g.s_hi >> 32));
      io[1] = make_uint4(g.uinteger, g.has32, 0u, 0u);
    }
  } else if (live && fn_dirty) {
    reinterpret_cast<uint2*>(bp.hbuf_wr)[env] = make_uint2((uint32_t)fnw, (uint32_t)(fnw >> 32));
  }
  TG_BITS_STAMP(7);
  TG_TIMING(if (p.dbg && lane == 0) {
    for (int i = 0; i < 8; ++i) p.dbg[(size_t)step_block * 16 + i] = tstamp[i];
    p.dbg[(size_t)step_block * 16 + 13] = __builtin_amdgcn_s_getreg(63492);  // HW_ID (wave / SIMD / CU / SE) of the logic wave
    p.dbg[(size_t)step_block * 16 + 15] = __builtin_amdgcn_s_getreg(63508);  // XCC_ID
  })
  if (live && !rec_lines && (ob.x != rb.x || ob.y != rb.y || ob.z != rb.z || ob.w != rb.w))
    TG_ST_STATE(reinterpret_cast<u32x4*>(rec_out + 1), (u32x4{ob.x, ob.y, ob.z, ob.w}));
}

// ------------------------------------------------------------------------------------------------
// bits_prepare_kernel: (re)build the acceleration state of the bits kernel from the reference-defined state —
// column masks from the uint8 board, the look-ahead permutation from the generator.  Runs after tg_reset /
// tg_set_state (not hot).  One thread per env.
// ------------------------------------------------------------------------------------------------
struct PrepareParams {
  const uint8_t* board;
  uint32_t* rec;
  tg_rng_rec* rng;
  void* cols;
  uint4* ahead;
  uint64_t* ring;       // [2][B]
  uint32_t* hbuf;       // [2][B]
  const uint8_t* mask;  // [B] or null: envs whose generator moved (their ring is refilled)
  int batch;
  int ruleset;
  int force;            // refill every ring
  int cur;              // ring parity the next launch reads
  int hcur;             // consumed-counter parity the next launch reads
  int queue_size;       // FUNCTIONAL: the bag length (EnvConfig.queue_size)
  int* err;             // or null: flag word of the handle (bit 2: a row of bedrock cells inside the four spawn rows — see the kernel)
};

template <int W, int H>
__global__ __launch_bounds__(256) void bits_prepare_kernel(const PrepareParams p) {
  using T = ColT<H>;
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= p.batch) return;
  const uint8_t* b = p.board + (size_t)env * W * H;
  T* gc = reinterpret_cast<T*>(p.cols) + (size_t)env * bits_cols_words<W, H>();
  T ones = (T)~T(0);  // rows whose every cell is 1
  for (int c = 0; c < W; ++c) {
    T v = 0, o = 0;
    for (int rr = 0; rr < H; ++rr) {
      const uint8_t cell = b[rr * W + c];
      v |= (T)(cell != 0 ? 1 : 0) << rr;
      o |= (T)(cell == 1 ? 1 : 0) << rr;
    }
    gc[c] = v;
    ones &= o;
  }
  if (bits_cols_words<W, H>() > W) gc[W] = 0;
  // (NUMPY rule-set: `ones` = the rows the reference's clear_filled_rows never takes, envs/tetris.py:493 — kept in the record below.
  // One inside the four spawn rows is refused — tg_set_state on the host, here for an upload through tg_set_state_dev, reported at the
  // next wait —: the grouped wrapper's legality would then depend on the reference's drop loop tunnelling through it from a colliding
  // start, wrappers/grouped.py:147-166, which the closed-form legal mask does not model.)
  ones &= (T)(((T)1 << (H - 1) << 1) - 1);
  if (p.ruleset == TG_RULESET_NUMPY && p.err && (ones & T(0xF)) != 0) atomicOr(p.err, 4);
  const size_t B = (size_t)p.batch;
  const bool moved = p.force || (p.mask ? p.mask[env] != 0 : true);
  uint4* ring4 = reinterpret_cast<uint4*>(p.ring);
  if (!moved) {  // ring untouched: make both parities agree with the one the next launch reads
    if (p.ruleset != TG_RULESET_NUMPY) return;  // (FUNCTIONAL: the look-ahead has no parities)
    ring4[(size_t)(1 - p.cur) * B + env] = ring4[(size_t)p.cur * B + env];
    p.hbuf[(size_t)(1 - p.hcur) * B + env] = p.hbuf[(size_t)p.hcur * B + env];
    return;
  }
  uint32_t* rec = p.rec + (size_t)env * 8;
  if (p.ruleset != TG_RULESET_NUMPY) {
    // FUNCTIONAL: the key-chain look-ahead of the env's position (see the top of this file), renumbered from 0
    const uint2 key = reinterpret_cast<const uint2*>(p.rng + env)[5];
    uint2* cs = reinterpret_cast<uint2*>(p.ahead) + (size_t)env * 16;
    uint2* rs = cs + 4;
    uint2 c = key, rr[3];
    for (int m = 0; m < 3; ++m) {
      uint32_t a0, a1, b0, b1;
      threefry2x32(c.x, c.y, 0u, 0u, a0, a1);
      threefry2x32(c.x, c.y, 0u, 1u, b0, b1);
      rr[m] = make_uint2(b0, b1);
      rs[m] = rr[m];                   // R_m
      c = make_uint2(a0, a1);
      cs[m + 1] = c;                   // c_{m+1}
    }
    const int nq = p.queue_size;
    uint64_t w = (uint64_t)fn_sort7(rr[1].x, rr[1].y, nq) | ((uint64_t)fn_sort7(rr[2].x, rr[2].y, nq) << 21);
    if (((rec[2] >> 8) & 0xFFu) >= (uint32_t)nq) {  // the next lock refills the bag: Q0 = sort7(f1(R_0))
      uint32_t z0, z1;
      threefry2x32(rr[0].x, rr[0].y, 0u, 1u, z0, z1);
      w |= (uint64_t)fn_sort7(z0, z1, nq) << 42;
    }
    reinterpret_cast<uint64_t*>(p.hbuf)[env] = w;
    rec[7] = 0;
    return;
  }
  Rng r;
  r.loaded = false;
  r.dirty = false;
  rng_load(r, p.rng + env);
  // a record uploaded with bag_index 7 holds a reshuffle the reference has already done (eager, :84)
  const uint32_t d2 = rec[2];
  if (((d2 >> 8) & 0xFF) >= 7) {
    const uint32_t d3 = rec[3], d4 = rec[4];
    uint64_t bag = (uint64_t)(d2 >> 16) | ((uint64_t)d3 << 16) | ((uint64_t)(d4 & 0xFF) << 48);
    bag = pcg_shuffle7(r, bag);
    rec[2] = (d2 & 0x000000FFu) | ((uint32_t)(bag & 0xFFFF) << 16);
    rec[3] = (uint32_t)(bag >> 16);
    rec[4] = (d4 & 0xFFFFFF00u) | (uint32_t)((bag >> 48) & 0xFF);
    rng_store(r, p.rng + env, 0);
  }
  rec[6] = (uint32_t)ones;  // the bedrock rows of the board (step_kernel_bits: `brow`); NUMPY uses neither score nor reserved
  rec[7] = (uint32_t)((uint64_t)ones >> 32);
  // T = kRing - 1, H = 0: entries 0 .. kRing - 2 ahead; "the state after entry -1" = the generator as it stands in
  // tg_rng_rec (loc = kLocRec), which this kernel leaves where the reference's generator is
  uint64_t w[2] = {(uint64_t)(kRing - 1) << 56, 0};
  uint4* a4 = p.ahead + (size_t)env * (2 * kRing);
  for (int sl = 0; sl < kRing - 1; ++sl) {
    const uint64_t perm = pcg_shuffle7(r, 0x06050403020100ULL);
    uint64_t sig = 0;
    for (int k = 0; k < 7; ++k) sig |= ((perm >> (8 * k)) & 0xF) << (4 * k);
    w[sl >> 1] |= sig << (28 * (sl & 1));
    a4[2 * sl] = make_uint4((uint32_t)r.s_lo, (uint32_t)(r.s_lo >> 32), (uint32_t)r.s_hi, (uint32_t)(r.s_hi >> 32));
    a4[2 * sl + 1] = make_uint4(r.uinteger, r.has32, 0u, 0u);
  }
  const uint4 rw = make_uint4((uint32_t)w[0], (uint32_t)(w[0] >> 32), (uint32_t)w[1], (uint32_t)(w[1] >> 32));
  ring4[env] = rw;
  ring4[B + env] = rw;
  p.hbuf[env] = (uint32_t)(kLocRec << 8);
  p.hbuf[B + env] = (uint32_t)(kLocRec << 8);
}

// ------------------------------------------------------------------------------------------------
// bits_materialize_rng_kernel: write "the generator the reference holds now" into tg_rng_rec.  A transition never
// touches tg_rng_rec (see the ring protocol); the state after the last consumed shuffle sits in ahead[(H - 1) & 3]
// unless an in-line shuffle already put it into the record (inl).  Runs before anything reads TG_BUF_RNG: tg_get_state,
// tg_reset (a reset without a seed continues the running generator), tg_set_state, tg_fetch / tg_device_ptr of the buffer.
// ------------------------------------------------------------------------------------------------
struct MaterializeParams {
  tg_rng_rec* rng;
  const uint4* ahead;
  const uint32_t* hbuf;  // the parity the next launch reads
  const uint4* inl;      // in-line buffers [2][B][2]
  int batch;
};

__global__ __launch_bounds__(256) void bits_materialize_rng_kernel(const MaterializeParams p) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= p.batch) return;
  const int Hraw = p.hbuf[env];
  const int loc = (Hraw >> 8) & 7;
  if (loc == kLocRec) return;
  const uint4* a4 = (loc == kLocBuf0 || loc == kLocBuf1) ? p.inl + ((size_t)(loc - kLocBuf0) * p.batch + env) * 2
                                    : p.ahead + (size_t)env * (2 * kRing) + 2 * (((Hraw & 0xFF) - 1) & (kRing - 1));
  const uint4 f0 = a4[0], f1 = a4[1];
  uint4* out = reinterpret_cast<uint4*>(p.rng + env);
  out[0] = f0;
  *reinterpret_cast<uint2*>(out + 2) = make_uint2(f1.x, f1.y);
}

}  // namespace tg
