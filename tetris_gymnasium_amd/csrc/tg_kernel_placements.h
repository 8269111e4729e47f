// tg_kernel_placements.h — the grouped wrapper's observation on the device: for every env all 4W (column, rotation)
// placement boards, u8 [B][4W][H+4][W+8].
//
// Reference: wrappers/grouped.py:124-207 `GroupedActionsObservations.observation`: for column x_base and r extra
// rotations of the active piece: centre it (x = x_base + padding - matrix.shape[1] // 2), hard drop FROM y = 0, then
//   * a cell outside the playfield columns (collision_with_frame, :104-122)  → np.ones_like(board)   (illegal)
//   * colliding where it landed (only possible at y = 0)                     → np.zeros_like(board)  (game over)
//   * otherwise  clear_filled_rows(project_tetromino(t, x, y))  on the PADDED board (bedrock = 1, envs/tetris.py:481-564)
// This is the [4W][Hp][Wp] tensor examples/train_lin_grouped.py's CNN variant consumes; the reference builds it with
// 4W Python-level board copies per step (≈200 steps/s at 10x20).
//
// One env per wavefront.  The slab is reduced once to column / row bit masks (wave ballots), every lane solves one
// placement on the masks (landing row = min over cells of ctz(column >> ...), completed rows), then the wave streams
// the env's 4W boards out as ONE contiguous block with 16-byte (or 4-byte) coalesced stores: a chunk of a regular
// placement is the chunk of the padded base board with at most four piece bytes patched in; placements that complete
// rows (rare) are built row-per-lane in an LDS scratch board first.  HBM-write bound: 17 280 B per env at 10x20.
#pragma once
#include "tg_kernel_features.h"

namespace tg {

struct PlacementParams {
  const uint8_t* board;
  const uint32_t* rec;
  uint8_t* boards;  // [B][4W][Hp][Wp]
  int batch;
};

template <int W, int H>
__global__ __launch_bounds__(kBlock) void placements_kernel(const PlacementParams p) {
  using G = Geo<W, H>;
  using M = typename ColMask<H>::type;
  constexpr int WP = G::WP, HP = G::HP, PB = WP * HP, NA = 4 * W;
  constexpr int UNIT = (PB % 16 == 0) ? 16 : 4;  // bytes per lane per store
  constexpr int NU = (PB + UNIT - 1) / UNIT;     // units per placement board (the last one partial when PB % 4 != 0: such
                                                 // boards start at 2- or 1-byte alignment in HBM; slab_st writes the tail)
  // per wave: slab | column masks | row masks | padded base board | scratch board | placement descriptors
  constexpr int PBA = ((PB + 15) / 16) * 16;
  constexpr int OFF_CM = G::LDS_SLAB, OFF_RM = OFF_CM + W * 8, OFF_PB = OFF_RM + ((H * 4 + 15) / 16) * 16;
  constexpr int OFF_SB = OFF_PB + PBA, OFF_DS = OFF_SB + PBA, PER_WAVE = OFF_DS + NA * 16;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_dyn[];
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int)(threadIdx.x >> 6));
  const int env = blockIdx.x * kWavesPerBlock + wave;
  if (env >= p.batch) return;
  uint8_t* base = lds_dyn + wave * PER_WAVE;
  uint8_t* slab = base;
  M* cm = reinterpret_cast<M*>(base + OFF_CM);
  uint32_t* rm = reinterpret_cast<uint32_t*>(base + OFF_RM);
  uint8_t* pb = base + OFF_PB;
  uint8_t* sb = base + OFF_SB;
  uint4* desc = reinterpret_cast<uint4*>(base + OFF_DS);

  const uint8_t* gboard = p.board + (size_t)env * G::CELLS;
  uint32_t* slab_dw = reinterpret_cast<uint32_t*>(slab);
#pragma unroll
  for (int k = 0; k < G::DPL; ++k) slab_dw[lane + 64 * k] = slab_ld<G::CELLS>(gboard, lane + 64 * k);
  EnvS s;
  rec_unpack(s, p.rec + (size_t)env * 8);
  wave_sync();

  // board → bit masks (lane r owns row r; one ballot per column), and the padded base board (envs/tetris.py:632-641)
  bool pre_full = false;  // a row that is ALREADY full (only a caller-provided board holds one): clear_filled_rows takes it with every placement
  bool row_ones = true;   // this lane's row consists of bedrock cells (value 1) only: never taken (envs/tetris.py:493)
  {
    uint32_t rowbits = 0;
    const int r = lane < H ? lane : H - 1;
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const uint8_t cell = slab[r * W + c];
      const bool f = lane < H && cell != 0;
      row_ones &= cell == 1;
      rowbits |= (f ? 1u : 0u) << c;
      const uint64_t colm = __ballot(f);
      if (lane == c) cm[c] = (M)colm;
    }
    if (lane < H) rm[lane] = rowbits;
    {
      constexpr uint32_t FULLC = W >= 32 ? 0xFFFFFFFFu : ((1u << W) - 1u);
      pre_full = __ballot(lane < H && (rowbits & FULLC) == FULLC && !row_ones) != 0;
    }
    for (int i = lane; i < PB; i += 64) {
      const int R = i / WP, C = i - R * WP;
      const bool inb = R < H && C >= 4 && C < W + 4;
      pb[i] = inb ? slab[inb ? R * W + C - 4 : 0] : (uint8_t)1;
    }
  }
  wave_sync();
  constexpr uint32_t FULL = W >= 32 ? 0xFFFFFFFFu : ((1u << W) - 1u);
  const M FLOOR = (M)(M(1) << H);
  const int n_half = s.piece == 0 ? 2 : 1;  // matrix.shape[1] // 2

  // ---- every lane solves placements lane, lane + 64, ...: descriptor {kind | y << 8 | val << 16, cell positions, F} ----
  //   kind 0 = regular, 1 = illegal (ones), 2 = game over (zeros);  pos_j = byte offset of cell j in the padded board
  for (int a0 = 0; a0 < NA; a0 += 64) {
    const int a = a0 + lane;
    if (a < NA) {
      const int colbase = a >> 2, nrot = (s.rot + (a & 3)) & 3;
      const int x = colbase + 4 - n_half;
      const uint32_t cells = (uint32_t)(c_cells.cells[s.piece] >> (16 * nrot)) & 0xFFFFu;
      int kk[4], cc[4];
      bool legal = true;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kk[j] = (cells >> (4 * j + 2)) & 3;
        cc[j] = x + (int)((cells >> (4 * j)) & 3) - 4;
        legal = legal && (unsigned)cc[j] < (unsigned)W;
      }
      uint32_t kind = 1, y = 0;
      uint32_t pos01 = 0, pos23 = 0;
      M F = 0;
      if (legal) {
        int yfirst = 1 << 20;
        M colm[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          colm[j] = cm[cc[j]];
          const M m = (M)((colm[j] | FLOOR) >> (1 + kk[j]));
          const int yj = m_ctz(m) + 1;
          yfirst = yj < yfirst ? yj : yfirst;
        }
        y = (uint32_t)(yfirst - 1);
        bool over = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) over = over || ((colm[j] >> (y + kk[j])) & 1);
        kind = over ? 2u : 0u;
        if (!over) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int rj = (int)y + kk[j];
            uint32_t add = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) add |= ((int)y + kk[i] == rj) ? (1u << cc[i]) : 0u;
            if (((rm[rj] | add) & FULL) == FULL) F |= (M)(M(1) << rj);
          }
          const uint32_t q0 = (uint32_t)(((int)y + kk[0]) * WP + cc[0] + 4), q1 = (uint32_t)(((int)y + kk[1]) * WP + cc[1] + 4);
          const uint32_t q2 = (uint32_t)(((int)y + kk[2]) * WP + cc[2] + 4), q3 = (uint32_t)(((int)y + kk[3]) * WP + cc[3] + 4);
          pos01 = q0 | (q1 << 16);
          pos23 = q2 | (q3 << 16);
        }
      }
      // (phase 2 recomputes the full-row mask from the row masks, so tall boards need no 64-bit field here)
      desc[a] = make_uint4(kind | (y << 8) | ((uint32_t)s.val << 16) | (((F != 0 || (pre_full && kind == 0)) ? 1u : 0u) << 24), pos01, pos23, 0u);
    }
  }
  wave_sync();

  // ---- phase 1: the env's boards as one linear array of NA * NU units; placements that clear rows are skipped ----
  uint8_t* dst = p.boards + (size_t)env * NA * PB;
  auto patch = [](uint32_t& v, const uint32_t off, const uint32_t q, const uint32_t val) {  // byte q (board offset) into the dword at `off`
    if ((q >> 2) == (off >> 2)) v = (v & ~(0xFFu << (8 * (q & 3)))) | (val << (8 * (q & 3)));
  };
  for (int u0 = 0; u0 < NA * NU; u0 += 64) {
    const int u = u0 + lane;
    if (u < NA * NU) {
      const int a = u / NU, o = (u - a * NU) * UNIT;  // placement, byte offset inside its board
      const uint4 d = desc[a];
      const uint32_t kind = d.x & 0xFFu, val = (d.x >> 16) & 0xFFu;
      if (!((d.x >> 24) & 1u)) {
        uint32_t v[UNIT / 4];
#pragma unroll
        for (int i = 0; i < UNIT / 4; ++i) {
          uint32_t t = *reinterpret_cast<const uint32_t*>(pb + o + 4 * i);
          if (kind == 0) {
            patch(t, (uint32_t)(o + 4 * i), d.y & 0xFFFFu, val);
            patch(t, (uint32_t)(o + 4 * i), d.y >> 16, val);
            patch(t, (uint32_t)(o + 4 * i), d.z & 0xFFFFu, val);
            patch(t, (uint32_t)(o + 4 * i), d.z >> 16, val);
          } else {
            t = kind == 1 ? 0x01010101u : 0u;
          }
          v[i] = t;
        }
        uint8_t* out = dst + (size_t)a * PB + o;
        if constexpr (UNIT == 16) *reinterpret_cast<uint4*>(out) = make_uint4(v[0], v[1], v[2], v[3]);
        else slab_st<PB>(dst + (size_t)a * PB, o / 4, v[0]);
      }
    }
  }

  // ---- phase 2: placements that complete rows (envs/tetris.py:481-512): row-per-lane into the scratch board ----
  for (int a = 0; a < NA; ++a) {
    const uint4 d = desc[a];  // wave-uniform
    if (!((d.x >> 24) & 1u)) continue;
    const uint32_t val = (d.x >> 16) & 0xFFu;
    // full-row mask of this placement, recomputed on the row masks (covers tall boards)
    const uint32_t q[4] = {d.y & 0xFFFFu, d.y >> 16, d.z & 0xFFFFu, d.z >> 16};
    M F = 0;
    int n = 0;
    {
      const int r = lane < H ? lane : H - 1;
      uint32_t bits = rm[r];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((int)(q[j] / WP) == r) bits |= 1u << (q[j] % WP - 4);
      const uint64_t fm = __ballot(lane < H && (bits & FULL) == FULL && !row_ones);
      F = (M)fm;
      n = (int)__popcll(fm);
    }
    wave_sync();
    if (lane < HP) {
      const int R = lane;
      uint8_t* orow = sb + R * WP;
      if (R >= H) {
        for (int c = 0; c < WP; ++c) orow[c] = 1;
      } else if (R < n) {  // fresh empty rows on top
        for (int c = 0; c < WP; ++c) orow[c] = (c >= 4 && c < W + 4) ? 0 : 1;
      } else {
        // output row R = the (R - n)-th kept row: k_i = i + #{f in F : f <= k_i}
        int src = R - n;
        M f = F;
        while (f) {
          const int fr = m_ctz(f);
          f &= (M)(f - 1);
          if (fr <= src) src += 1;
        }
        const uint8_t* irow = pb + src * WP;
        for (int c = 0; c < WP; ++c) {
          uint8_t vv = irow[c];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (q[j] == (uint32_t)(src * WP + c)) vv = (uint8_t)val;
          orow[c] = vv;
        }
      }
    }
    wave_sync();
    uint8_t* out = dst + (size_t)a * PB;
    for (int o = lane * 4; o < PB; o += 256) slab_st<PB>(out, o / 4, *reinterpret_cast<const uint32_t*>(sb + o));
    wave_sync();
  }
}

template <int W, int H>
constexpr size_t placements_lds_bytes() {
  using G = Geo<W, H>;
  constexpr int PB = G::WP * G::HP, PBA = ((PB + 15) / 16) * 16;
  return (size_t)kWavesPerBlock * (G::LDS_SLAB + W * 8 + ((H * 4 + 15) / 16) * 16 + 2 * PBA + 4 * W * 16);
}

}  // namespace tg
