/*
 * tg_oracle.h — CPU restatement of the reference's step path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (tetris_gymnasium_amd + libtetris_gym_hip.so) never links, imports or calls it.
 *
 * Parity status:
 *   NUMPY rule-set  — pinned: checked against golden vectors generated from the unmodified
 *                     reference envs/tetris.py (tests/golden/ fixtures, oracle/refgen/make_golden.py).
 *   numpy PCG64/SeedSequence/shuffle (third-party, numpy 2.2.x) — pinned against numpy itself.
 *   FUNCTIONAL rule-set game logic — pinned by the reference's own known-answer tests restated in
 *                     tests/test_oracle_functional.py, cross-checked against the NUMPY oracle.
 *   FUNCTIONAL piece sequence (jax 0.5.3 threefry split/permutation) — PARITY UNPINNED: jax is
 *                     not installable here; threefry2x32 itself is pinned by the Random123 KATs.
 */
#ifndef TG_ORACLE_H
#define TG_ORACLE_H

#include <stdint.h>
#include "../include/tetris_gym_hip.h" /* tg_env_rec / tg_rng_rec layouts and TG_* constants */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tgo_batch tgo_batch;

tgo_batch* tgo_create(int ruleset, int width, int height, int padding, int queue_size, int gravity,
                      int batch, int env_offset);
void tgo_destroy(tgo_batch* b);
/* Tetris(actions_mapping=..., rewards_mapping=...) (envs/tetris.py:83-84, mappings/actions.py:12-19, mappings/rewards.py:12-15):
 * field values in declaration order; NULL keeps what is set (default: the stock dataclasses) */
void tgo_set_mappings(tgo_batch* b, const int32_t action_ids[8], const double rewards[4]);
/* the rewards of the last tgo_step / tgo_grouped_step as Python computes them (double precision) */
void tgo_last_rewards(const tgo_batch* b, double* out);
/* tg_step_out.flags of the last tgo_step / tgo_grouped_step (TG_OUT_COMMITTED, TG_OUT_ILLEGAL) */
void tgo_last_flags(const tgo_batch* b, uint8_t* out);

/* same semantics as tg_reset / tg_step (include/tetris_gym_hip.h) */
void tgo_reset(tgo_batch* b, const uint64_t* seeds, const uint8_t* mask);
void tgo_step(tgo_batch* b, const int32_t* actions, float* reward, uint8_t* terminated,
              uint8_t* lines, int autoreset, int nthreads);
/* grouped action set (wrappers/grouped.py:209-269); illegal[i] set when the action was masked out */
void tgo_grouped_step(tgo_batch* b, const int32_t* actions, float* reward, uint8_t* terminated,
                      uint8_t* lines, uint8_t* illegal, int terminate_on_illegal, int autoreset,
                      int nthreads);
/* legal mask u8 [B][4W]; boards (optional) u8 [B][4W][Hp][Wp]  (wrappers/grouped.py:124-207) */
void tgo_grouped_enumerate(tgo_batch* b, uint8_t* legal_mask, uint8_t* boards, int nthreads);

/* per-placement feature tensors u8 [B][4W][W+3] and current-state features u8 [B][W+3]
 * (wrappers/grouped.py:185-205 with observation_wrappers=[FeatureVectorObservation], wrappers/observation.py:236-278) */
void tgo_grouped_features(tgo_batch* b, uint8_t* feats, int nthreads);
void tgo_features(tgo_batch* b, uint8_t* feats, int nthreads);

/* observation pack, layouts as TG_OBS_* */
void tgo_observe(tgo_batch* b, int obs_mode, void* out, int nthreads);

void tgo_get_state(tgo_batch* b, uint8_t* board, tg_env_rec* rec, tg_rng_rec* rng);
void tgo_set_state(tgo_batch* b, const uint8_t* board, const tg_env_rec* rec, const tg_rng_rec* rng);

/* padded boards u8 [B][Hp][Wp] exactly as env.unwrapped.board */
void tgo_get_padded_boards(tgo_batch* b, uint8_t* out);
void tgo_set_padded_boards(tgo_batch* b, const uint8_t* in);

/* ---- third-party arithmetic restated (pinned in tests) ---- */
void tgo_seed_pcg64(uint64_t seed, tg_rng_rec* out);                /* SeedSequence(seed) → PCG64 */
uint32_t tgo_pcg64_next32(tg_rng_rec* r);
void tgo_pcg64_shuffle7(tg_rng_rec* r, uint8_t* bag);               /* Generator.shuffle on 7 items */
void tgo_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]);
void tgo_jax_split(const uint32_t key[2], uint32_t out0[2], uint32_t out1[2]);
void tgo_jax_permutation7(const uint32_t key[2], uint8_t out[7]);
void tgo_jax_permutation(const uint32_t key[2], int n, uint8_t* out);   /* permutation(key, arange(n)), 1 <= n <= 7 */
void tgo_set_threefry_partitionable(int on); /* default 1 (jax >= 0.5.0 default) */

/* single-function probes used by the restated reference unit tests */
int tgo_fn_score(int rows_cleared);                                 /* functional/core.py:124-146 */
int tgo_fn_clear_rows(uint8_t* playfield, int width, int height);   /* functional/core.py:185-227 on a bare H x W field, in place */
int tgo_np_score(int rows_cleared, int width);                      /* envs/tetris.py:621-630 */
/* tetromino matrices: NUMPY native n×n (values piece+2) rotated r times by np.rot90(k=1);
 * returns n, writes n*n bytes. FUNCTIONAL: 4×4 binary LUT (functional/tetrominoes.py:124-148). */
int tgo_np_matrix(int piece, int rot, uint8_t* out);
void tgo_fn_matrix(int piece, int rot, uint8_t out[16]);

/* FeatureVectorObservation (wrappers/observation.py:177-278) over a dict observation:
 * board/mask u8 [Hp][Wp]; out u8 [W+3] = heights, max height, holes, bumpiness */
void tgo_feature_vector(const uint8_t* board, const uint8_t* mask, int width, int height, int padding,
                        uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
