// tg_kernel_features.h — grouped placement observations without materialising the placement boards.
//
// Reference: wrappers/grouped.py:124-207 (for each of the 4W (column, rotation) placements: rotate, centre, hard
// drop from y = 0, then  clear_filled_rows(project_tetromino(...))  → one board per placement) wrapped by
// wrappers/observation.py:236-278 FeatureVectorObservation (column heights, max height, holes, bumpiness as uint8;
// the wrapper's `board_obs[active_tetromino_mask] = 0` with an all-zero uint8 mask wipes ROW 0 only).  This is the
// [4W][W+3] tensor examples/train_lin_grouped.py trains on; the reference builds 4W full boards per step for it
// (≈200 steps/s).
//
// One env per wavefront, one PLACEMENT per lane.  The board is reduced once to bit masks with wave ballots
// (column c occupancy = ballot over the row-per-lane bytes), kept in LDS; a lane then does its whole placement on
// bit masks: landing row = min over the piece's cells of ctz(column mask >> ...), completed rows = row masks |
// piece bits == full, compaction = bit surgery on the column masks, features = ctz / popcount.
#pragma once
#ifndef __HIPCC_RTC__
#include <type_traits>
#endif

#include "tg_lane.h"

namespace tg {

struct FeatureParams {
  const uint8_t* board;
  const uint32_t* rec;
  uint8_t* placement_feats;  // [B][4W][W+3] or null
  uint8_t* feats;            // [B][W+3] or null: FeatureVectorObservation of the current observation
  int batch;
};

template <int H>
struct ColMask {
  using type = typename pick_t<(H + 1 <= 32), uint32_t, uint64_t>::type;
};

template <typename M>
__device__ __forceinline__ int m_ctz(M m) {
  if constexpr (sizeof(M) == 4) return __builtin_ctz(m);
  else return __builtin_ctzll(m);
}
template <typename M>
__device__ __forceinline__ int m_popc(M m) {
  if constexpr (sizeof(M) == 4) return __popc(m);
  else return __popcll(m);
}

// features of one final board given by its W column masks (bit r = row r filled, row 0 = top), rows < wipe wiped
template <int W, int H, typename M, typename ColFn>
__device__ __forceinline__ void column_features(ColFn col, int wipe_rows, uint8_t* out) {
  int prev = 0, maxh = 0, holes = 0, bump = 0;
  const M wipe = (M)((M(1) << wipe_rows) - 1);
#pragma unroll 1
  for (int c = 0; c < W; ++c) {
    const M m = col(c) & ~wipe;
    int h = 0;
    if (m) {
      const int top = m_ctz(m);
      h = H - top;
      holes += (H - 1 - top) - m_popc((M)(m >> (top + 1)));
    }
    out[c] = (uint8_t)h;
    maxh = h > maxh ? h : maxh;
    if (c > 0) bump += h > prev ? h - prev : prev - h;
    prev = h;
  }
  out[W] = (uint8_t)maxh;
  out[W + 1] = (uint8_t)holes;
  out[W + 2] = (uint8_t)bump;
}

template <int W, int H>
__global__ __launch_bounds__(kBlock) void features_kernel(const FeatureParams p) {
  using G = Geo<W, H>;
  using M = typename ColMask<H>::type;
  constexpr int NF = W + 3, NA = 4 * W;
  constexpr int FEAT_BYTES = ((NA + 1) * NF + 15) / 16 * 16;
  // per wave: slab | column masks M[W] | row masks u32[H] | staged features
  constexpr int OFF_CM = G::LDS_SLAB, OFF_RM = OFF_CM + W * 8, OFF_FT = OFF_RM + ((H * 4 + 15) / 16) * 16;
  constexpr int PER_WAVE = OFF_FT + FEAT_BYTES;
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock * PER_WAVE];
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int)(threadIdx.x >> 6));
  const int env = blockIdx.x * kWavesPerBlock + wave;
  if (env >= p.batch) return;
  uint8_t* base = lds + wave * PER_WAVE;
  uint8_t* slab = base;
  M* cm = reinterpret_cast<M*>(base + OFF_CM);
  uint32_t* rm = reinterpret_cast<uint32_t*>(base + OFF_RM);
  uint8_t* ft = base + OFF_FT;

  const uint8_t* gboard = p.board + (size_t)env * G::CELLS;
  uint32_t* slab_dw = reinterpret_cast<uint32_t*>(slab);
#pragma unroll
  for (int k = 0; k < G::DPL; ++k) slab_dw[lane + 64 * k] = slab_ld<G::CELLS>(gboard, lane + 64 * k);
  EnvS s;
  rec_unpack(s, p.rec + (size_t)env * 8);
  wave_sync();

  // board → bit masks: lane r owns row r; one ballot per column
  uint32_t rowbits = 0;
  bool row_ones = true;  // this lane's row consists of bedrock cells (value 1) only: clear_filled_rows never takes it (envs/tetris.py:493)
  {
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
  }
  constexpr uint32_t FULL = W >= 32 ? 0xFFFFFFFFu : ((1u << W) - 1u);
  // rows that are ALREADY full (only a caller-provided board holds one): clear_filled_rows takes them with every placement
  const M PRE = (M)__ballot(lane < H && (rowbits & FULL) == FULL && !row_ones);
  wave_sync();
  const M FLOOR = (M)(M(1) << H);

  const int n_half = s.piece == 0 ? 2 : 1;  // matrix.shape[1] // 2
  // One pass over the lanes: lane a < 4W does placement a; the CURRENT observation (board + projected piece, rows 0 and 1
  // wiped: the wrapper's uint8 mask with ones acts as integer row indices) rides along as lane 4W of the same instruction
  // stream — as a pass of its own on one lane it cost the wave as much as all the placements together (round 4).
  const bool want_cur = p.feats != nullptr;
  for (int a0 = 0; a0 < NA + 1; a0 += 64) {
    const int a = a0 + lane;
    const bool is_cur = want_cur && a == NA;
    if (a < NA || is_cur) {
      uint8_t* out = ft + a * NF;
      const int nrot = is_cur ? s.rot : (s.rot + (a & 3)) & 3;
      const int x = is_cur ? s.x : (a >> 2) + 4 - n_half;
      const uint32_t cells = (uint32_t)(c_cells.cells[s.piece] >> (16 * nrot)) & 0xFFFFu;
      int kk[4], cc[4];
      bool legal = true;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kk[j] = (cells >> (4 * j + 2)) & 3;
        cc[j] = x + (int)((cells >> (4 * j)) & 3) - 4;
        legal = legal && (unsigned)cc[j] < (unsigned)W;
      }
      if (!is_cur && !legal) {  // np.ones_like(board_obs) → every column full below the wiped row 0
#pragma unroll 1
        for (int c = 0; c < W; ++c) out[c] = (uint8_t)(H - 1);
        out[W] = (uint8_t)(H - 1);
        out[W + 1] = 0;
        out[W + 2] = 0;
      } else {
        M colm[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) colm[j] = cm[(unsigned)cc[j] < (unsigned)W ? cc[j] : 0];
        int y, wipe = 1;
        bool draw = true, zero = false;
        M F = 0;
        if (is_cur) {
          // the piece where it is; not drawn when it collides (out of the frame or on a filled cell)
          y = s.y;
          bool coll = !legal;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = y + kk[j];
            coll = coll || r >= H || ((colm[j] >> (r < H ? r : 0)) & 1);
          }
          draw = !coll;
          // bounding box of the native matrix clipped to the padded board: any cell inside ⇒ the mask holds a 1
          const int n = s.piece == 0 ? 4 : (s.piece == 1 ? 2 : 3);
          wipe = (s.y < H + 4 && s.x < W + 8 && n > 0) ? 2 : 1;
        } else {
          // hard drop from y = 0: first y' >= 1 where a cell is blocked (filled cell or floor), minus one
          int yfirst = 1 << 20;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const M m = (M)((colm[j] | FLOOR) >> (1 + kk[j]));  // rows >= 1 + k_j
            const int yj = m_ctz(m) + 1;                        // y' at which cell j first hits
            yfirst = yj < yfirst ? yj : yfirst;
          }
          y = yfirst - 1;
          // collides where it lands (only possible at y = 0): "game over placement" → zero board
#pragma unroll
          for (int j = 0; j < 4; ++j) zero = zero || ((colm[j] >> (y + kk[j])) & 1);
          if (!zero) {  // completed rows among the rows the piece touches (+ the rows that were full before)
            F = PRE;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int rj = y + kk[j];
              uint32_t add = 0;
#pragma unroll
              for (int i = 0; i < 4; ++i) add |= (y + kk[i] == rj) ? (1u << cc[i]) : 0u;
              if (((rm[rj] | add) & FULL) == FULL) F |= (M)(M(1) << rj);
            }
          }
        }
        if (zero) {
#pragma unroll 1
          for (int c = 0; c < NF; ++c) out[c] = 0;
        } else {
          auto col = [&](int c) -> M {
            M m = cm[c];
            if (draw) {
#pragma unroll
              for (int j = 0; j < 4; ++j) m |= (cc[j] == c) ? (M)(M(1) << (y + kk[j])) : (M)0;
            }
            M f = F;
            while (f) {  // remove cleared rows top to bottom: rows above move down by one (envs/tetris.py:498-510)
              const int r = m_ctz(f);
              f &= (M)(f - 1);
              const M low = (M)((M(1) << r) - 1);
              m = (M)((m & ~((low << 1) | 1)) | ((m & low) << 1));
            }
            return m;
          };
          column_features<W, H, M>(col, wipe, out);
        }
      }
    }
  }
  wave_sync();
  // staged features → HBM, coalesced dwords (4W (W + 3) bytes per env: a whole number of dwords, dword-aligned)
  if (p.placement_feats) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(p.placement_feats + (size_t)env * NA * NF);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(ft);
    for (int i = lane; i < NA * NF / 4; i += 64) dst[i] = src[i];
  }
  if (p.feats && lane < NF) p.feats[(size_t)env * NF + lane] = ft[NA * NF + lane];
}

}  // namespace tg
