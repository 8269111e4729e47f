// tg_tables.h — tetromino shape tables for the HIP engine, generated at compile time.
//
// A shape is a 16-bit mask over the 4x4 matrix box: bit (4*row + col) set = cell filled.
// Base matrices and order (I O T S Z J L) follow the reference (envs/tetris.py:47-75; the functional
// LUT functional/tetrominoes.py:43-79 holds the same native matrices, zero-padded bottom/right to
// 4x4, so the occupied cells relative to (x, y) are identical in both rule-sets).  Rotation r is
// np.rot90(native, k=r) inside the native n x n box (envs/tetris.py:429-443,
// functional/tetrominoes.py:82-106).
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

namespace tg {

struct ShapeTable {
  uint16_t mask[7][4];
  uint8_t n[7];  // native box size: I 4, O 2, others 3
};

constexpr ShapeTable make_shape_table() {
  ShapeTable t{};
  const int N[7] = {4, 2, 3, 3, 3, 3, 3};
  //                     I       O       T       S       Z       J       L
  const uint16_t base[7] = {0x00F0, 0x0033, 0x0072, 0x0036, 0x0063, 0x0071, 0x0074};
  for (int p = 0; p < 7; ++p) {
    const int n = N[p];
    t.n[p] = static_cast<uint8_t>(n);
    uint16_t m = base[p];
    for (int r = 0; r < 4; ++r) {
      t.mask[p][r] = m;
      uint16_t o = 0;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)  // rot90(k=1): out[i][j] = m[j][n-1-i]
          if ((m >> (4 * j + (n - 1 - i))) & 1) o = static_cast<uint16_t>(o | (1u << (4 * i + j)));
      m = o;
    }
  }
  return t;
}

}  // namespace tg
