// tg_lane.h — one-env-per-lane building blocks shared by the step kernel (tg_kernel_bits.h) and its predecessors: the
// piece cell table, the per-lane generator record, one PCG64 step on 32/64-bit pieces, the packed env record
// (lane_unpack / lane_pack) and LaneBoard (an env's uint8 slab in LDS addressed by the lane(s) serving the env).
#pragma once
#include "tg_common.h"

namespace tg {

#ifdef TG_PHASE_TIMING
#define TG_STAMP(i) do { if (p.dbg && lane == 0) p.dbg[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TG_STAMP(i) do {} while (0)
#endif

// piece cells as 4 nibbles (k << 2 | c) per (piece, rot): 4 rotations x 16 bit per piece
struct CellTable {
  uint64_t cells[7];
};

constexpr CellTable make_cell_table() {
  CellTable t{};
  const ShapeTable s = make_shape_table();
  for (int p = 0; p < 7; ++p) {
    uint64_t row = 0;
    for (int r = 0; r < 4; ++r) {
      uint64_t v = 0;
      int n = 0;
      for (int cell = 0; cell < 16; ++cell)
        if ((s.mask[p][r] >> cell) & 1) {
          v |= (uint64_t)cell << (4 * n);  // cell index = 4*k + c  → nibble holds (k << 2 | c)
          ++n;
        }
      row |= v << (16 * r);
    }
    t.cells[p] = row;
  }
  return t;
}

__constant__ CellTable c_cells = make_cell_table();

// per-lane generator state (numpy PCG64 + buffered half, or the jax key)
struct LaneRng {
  uint64_t s_lo, s_hi, i_lo, i_hi;
  uint32_t uinteger, has32, k0, k1;
  const tg_rng_rec* src;  // this env's record in HBM
  bool loaded;            // wave-uniform: the 48-byte records are fetched only by wave-steps that draw random numbers
  bool dirty;             // per env
};

__device__ __forceinline__ void lane_rng_ensure(LaneRng& r) {
  if (r.loaded) return;
  const uint4* q = reinterpret_cast<const uint4*>(r.src);
  const uint4 g0 = q[0], g1 = q[1], g2 = q[2];
  r.s_lo = (uint64_t)g0.x | ((uint64_t)g0.y << 32); r.s_hi = (uint64_t)g0.z | ((uint64_t)g0.w << 32);
  r.i_lo = (uint64_t)g1.x | ((uint64_t)g1.y << 32); r.i_hi = (uint64_t)g1.z | ((uint64_t)g1.w << 32);
  r.uinteger = g2.x; r.has32 = g2.y; r.k0 = g2.z; r.k1 = g2.w;
  r.loaded = true;
}

// one PCG64 step (state = state * M + inc, XSL-RR output of the NEW state) from 32/64-bit pieces: 10 multiplies
__device__ __forceinline__ void lane_pcg_step(const uint64_t s_lo, const uint64_t s_hi, const uint64_t i_lo,
                                              const uint64_t i_hi, uint64_t& n_lo, uint64_t& n_hi, uint64_t& out) {
  constexpr uint64_t MLO = 0x4385DF649FCCF645ULL, MHI = 0x2360ED051FC65DA4ULL;
  constexpr uint32_t b0 = (uint32_t)MLO, b1 = (uint32_t)(MLO >> 32);
  const uint32_t a0 = (uint32_t)s_lo, a1 = (uint32_t)(s_lo >> 32);
  const uint64_t p0 = (uint64_t)a0 * b0, p1 = (uint64_t)a0 * b1, p2 = (uint64_t)a1 * b0, p3 = (uint64_t)a1 * b1;
  const uint64_t mid = (p0 >> 32) + (uint32_t)p1 + (uint32_t)p2;
  const uint64_t lo = (uint64_t)(uint32_t)p0 | (mid << 32);
  uint64_t hi = p3 + (p1 >> 32) + (p2 >> 32) + (mid >> 32);  // high half of s_lo * MLO
  hi += s_lo * MHI + s_hi * MLO;
  n_lo = lo + i_lo;
  n_hi = hi + i_hi + (n_lo < lo ? 1ULL : 0ULL);
  const uint64_t x = n_hi ^ n_lo;
  const unsigned rot = (unsigned)(n_hi >> 58);
  out = (x >> rot) | (x << ((64u - rot) & 63u));
}

// per-lane env state
struct LaneS {
  int piece, rot, x, y, val, flags, hp, hr, hv, bidx;
  uint64_t bag, queue;
  float score;
  uint32_t reserved;
};

__device__ __forceinline__ void lane_unpack(LaneS& s, const uint4 a, const uint4 b) {
  s.piece = a.x & 0xFF; s.rot = (a.x >> 8) & 0xFF; s.x = (a.x >> 16) & 0xFF; s.y = a.x >> 24;
  s.val = a.y & 0xFF; s.flags = (a.y >> 8) & 0xFF; s.hp = (a.y >> 16) & 0xFF; s.hr = a.y >> 24;
  s.hv = a.z & 0xFF; s.bidx = (a.z >> 8) & 0xFF;
  s.bag = (uint64_t)(a.z >> 16) | ((uint64_t)a.w << 16) | ((uint64_t)(b.x & 0xFF) << 48);
  s.queue = (uint64_t)(b.x >> 8) | ((uint64_t)b.y << 24);
  s.score = __uint_as_float(b.z);
  s.reserved = b.w;
}

__device__ __forceinline__ void lane_pack(const LaneS& s, uint4& a, uint4& b) {
  a.x = (uint32_t)s.piece | ((uint32_t)s.rot << 8) | ((uint32_t)s.x << 16) | ((uint32_t)s.y << 24);
  a.y = (uint32_t)s.val | ((uint32_t)s.flags << 8) | ((uint32_t)s.hp << 16) | ((uint32_t)s.hr << 24);
  a.z = (uint32_t)s.hv | ((uint32_t)s.bidx << 8) | ((uint32_t)(s.bag & 0xFFFF) << 16);
  a.w = (uint32_t)(s.bag >> 16);
  b.x = (uint32_t)((s.bag >> 48) & 0xFF) | ((uint32_t)(s.queue & 0xFFFFFF) << 8);
  b.y = (uint32_t)(s.queue >> 24);
  b.z = __float_as_uint(s.score);
  b.w = s.reserved;
}

// OR of a small bit-set over the L lanes that serve one env (L = 4: one DPP quad, two v_or_b32_dpp)
template <int L>
__device__ __forceinline__ uint32_t env_or(uint32_t x) {
  if constexpr (L >= 2) x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  if constexpr (L == 4) x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  return x;
}

// One env's slab in LDS, addressed by the L lanes that serve the env: lane `sub` owns piece cell(s)
// sub, sub+L, ... (L = 4: one cell per lane; L = 1: all four).
template <int W, int H, int L>
struct LaneBoard {
  uint8_t* my;  // this env's slab in LDS
  int sub;      // lane index within the env's lane group
  // LDS byte offset of playfield cell (row, col) or -1 when it is implicit bedrock (envs/tetris.py:632-641)
  static __device__ __forceinline__ int cell_addr(int row, int col) {
    const bool inb = (unsigned)col < (unsigned)W && (unsigned)row < (unsigned)H;
    return inb ? row * W + col : -1;
  }
  // N placements of the same 4 cells tested at once: all byte reads are issued before any is consumed;
  // returns bit i = placement i collides
  template <int N>
  __device__ __forceinline__ uint32_t collide_n(uint32_t cells, const int (&xs)[N], const int (&ys)[N]) const {
    constexpr int CPL = 4 / L;  // cells per lane
    int addr[N][CPL];
#pragma unroll
    for (int jj = 0; jj < CPL; ++jj) {
      const int j = jj * L + sub;
      const int k = (cells >> (4 * j + 2)) & 3, c = (cells >> (4 * j)) & 3;
#pragma unroll
      for (int i = 0; i < N; ++i) addr[i][jj] = cell_addr(ys[i] + k, xs[i] + c - 4);
    }
    uint8_t v[N][CPL];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int jj = 0; jj < CPL; ++jj) v[i][jj] = my[addr[i][jj] < 0 ? 0 : addr[i][jj]];
    uint32_t hits = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      bool h = false;
#pragma unroll
      for (int jj = 0; jj < CPL; ++jj) h |= addr[i][jj] < 0 || v[i][jj] != 0;
      hits |= (h ? 1u : 0u) << i;
    }
    return env_or<L>(hits);
  }
  __device__ __forceinline__ bool collide(uint32_t cells, int x, int y) const {
    const int xs[1] = {x}, ys[1] = {y};
    return collide_n<1>(cells, xs, ys) != 0;
  }
  // The reference's drop loop `while not collision(y + 1): y += 1` (envs/tetris.py:445-448,
  // functional/core.py:243-249) for the envs in `go`, four rows per pass.
  __device__ __forceinline__ int drop(uint32_t cells, int x, int y, bool go) const {
    // per cell: LDS offset of its current position, rows left above the floor, column inside the playfield;
    // the four probe rows of a pass are immediate offsets (W, 2W, 3W, 4W) from that position, so a pass costs
    // 4 byte reads + a handful of compares per cell (reads may run a few rows past the slab: LDS is padded and
    // those probes are masked by `lim`).
    constexpr int CPL = 4 / L;
    int cur[CPL], lim[CPL];
    bool colok[CPL];
#pragma unroll
    for (int jj = 0; jj < CPL; ++jj) {
      const int j = jj * L + sub;
      const int k = (cells >> (4 * j + 2)) & 3, c = (cells >> (4 * j)) & 3;
      const int col = x + c - 4, row = y + k;
      colok[jj] = (unsigned)col < (unsigned)W;
      lim[jj] = H - row;  // probe t is inside the playfield iff t < lim
      const int a = row * W + (colok[jj] ? col : 0);
      cur[jj] = a < H * W ? a : H * W - 1;
    }
    int yd = y;
    while (__any(go)) {
      uint8_t v[CPL][4];
#pragma unroll
      for (int jj = 0; jj < CPL; ++jj)
#pragma unroll
        for (int t = 1; t <= 4; ++t) v[jj][t - 1] = my[cur[jj] + t * W];
      uint32_t hit = 0;
#pragma unroll
      for (int t = 1; t <= 4; ++t) {
        bool h = false;
#pragma unroll
        for (int jj = 0; jj < CPL; ++jj) h |= !colok[jj] || t >= lim[jj] || v[jj][t - 1] != 0;
        hit |= (h ? 1u : 0u) << (t - 1);
      }
      hit = env_or<L>(hit);
      const int steps = hit ? (int)__builtin_ctz(hit) : 4;
      const int adv = go ? steps : 0;
      yd += adv;
#pragma unroll
      for (int jj = 0; jj < CPL; ++jj) { cur[jj] += adv * W; lim[jj] -= adv; }
      go = go && steps == 4;
    }
    return yd;
  }
  // (NIBBLE: returns whether a sum left the nibble range — the packed board of the FUNCTIONAL rule-set, whose lock ADDS)
  template <bool NIBBLE = false>
  __device__ __forceinline__ bool stamp(uint32_t cells, int x, int y, uint32_t val, bool add, bool on) const {
    constexpr int CPL = 4 / L;
    int addr[CPL];
    bool over = false;
#pragma unroll
    for (int jj = 0; jj < CPL; ++jj) {
      const int j = jj * L + sub;
      const int k = (cells >> (4 * j + 2)) & 3, c = (cells >> (4 * j)) & 3;
      addr[jj] = cell_addr(y + k, x + c - 4);
    }
    if (add) {
      uint8_t old[CPL];
#pragma unroll
      for (int jj = 0; jj < CPL; ++jj) old[jj] = my[addr[jj] < 0 ? 0 : addr[jj]];
#pragma unroll
      for (int jj = 0; jj < CPL; ++jj) {
        if (on && addr[jj] >= 0) my[addr[jj]] = (uint8_t)(old[jj] + val);
        if (NIBBLE) over |= on && addr[jj] >= 0 && (uint32_t)old[jj] + val > 15u;
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < CPL; ++jj)
        if (on && addr[jj] >= 0) my[addr[jj]] = (uint8_t)val;
    }
    return over;
  }
  // is playfield row `row` completely filled?  (16-bit reads when W is even, all issued before use)
  __device__ __forceinline__ bool row_full(int row) const {
    const bool ok = (unsigned)row < (unsigned)H;
    const int r = ok ? row : 0;
    bool full = ok;
    if constexpr (W % 2 == 0) {
      uint16_t v[W / 2];
      const uint16_t* rp = reinterpret_cast<const uint16_t*>(my + r * W);
#pragma unroll
      for (int c = 0; c < W / 2; ++c) v[c] = rp[c];
#pragma unroll
      for (int c = 0; c < W / 2; ++c) full &= (v[c] & 0xFF) != 0 && (v[c] >> 8) != 0;
    } else {
      uint8_t v[W];
#pragma unroll
      for (int c = 0; c < W; ++c) v[c] = my[r * W + c];
#pragma unroll
      for (int c = 0; c < W; ++c) full &= v[c] != 0;
    }
    return full;
  }
  // are any of the playfield rows y0 .. y0+3 completely filled?
  __device__ __forceinline__ bool any_row_full4(int y0) const {
    uint32_t any = 0;
#pragma unroll
    for (int kk = 0; kk < 4 / L; ++kk) any |= row_full(y0 + kk * L + sub) ? 1u : 0u;
    return env_or<L>(any) != 0;
  }
};

__device__ __forceinline__ uint32_t lane_cells(const uint64_t* tab, int piece, int rot) {
  return (uint32_t)(tab[piece] >> (16 * rot)) & 0xFFFFu;
}

}  // namespace tg
