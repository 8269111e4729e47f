/*
 * tetris_gym_hip.h — C ABI of the MI355X-native batched Tetris step engine (libtetris_gym_hip.so).
 *
 * The reference (Max-We/Tetris-Gymnasium 0.3.1) has no FFI boundary: the hot path is Python calling
 * NumPy / JAX.  This header is the boundary a maintainer would bind with ctypes to replace
 *   - tetris_gymnasium/envs/tetris.py:203-272   Tetris.step            (ruleset TG_RULESET_NUMPY)
 *   - tetris_gymnasium/envs/tetris.py:274-307   Tetris.reset
 *   - tetris_gymnasium/envs/tetris_fn.py:276-315 step / :416-435 batched_step  (TG_RULESET_FUNCTIONAL)
 *   - tetris_gymnasium/envs/tetris_fn.py:318-367 reset / :438-467 batched_reset
 *   - tetris_gymnasium/wrappers/grouped.py:124-269 grouped placement mask + step (TG_ACTIONS_GROUPED)
 * Plain pointers and sizes only; no torch / numpy types.  All functions return 0 on success or a
 * negative TG_E* code; the message is available from tg_last_error().  No exceptions cross the ABI.
 *
 * Threading: calls on one handle must be serialised by the caller.  Different handles (one per GPU)
 * are independent.  Each handle owns one HIP stream; tg_step* are asynchronous on it, tg_fetch /
 * tg_get_state / tg_sync synchronise.
 */
#ifndef TETRIS_GYM_HIP_H
#define TETRIS_GYM_HIP_H

#ifndef __HIPCC_RTC__ /* (the engine's own run-time compilation of further board sizes has these types built in) */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define TG_ABI_VERSION 6 /* 2: tg_config launch knobs, tg_dev_copy, tg_get_state_dev / tg_set_state_dev
                          * 3: TG_BUF_HOST_PACK, tg_host_pack_layout, tg_fetch_many, tg_step_fetch_many, tg_shape_mask
                          * 4: tg_config.action_ids / rewards (ActionsMapping / RewardsMapping as per-handle data),
                          *    TG_OUT_COMMITTED / TG_OUT_ILLEGAL in tg_step_out.flags
                          * 5: tg_observe_as, tg_features_as (a wrapper's observation of ANY handle's current state)
                          * 6: tg_device_pci_bus_id; an ActionsMapping whose hard_drop aliases an earlier field is accepted;
                          *    NUMPY records: `score` / `reserved` are device-private (tg_get_state returns 0) */

/* ---- rule-sets (the reference contains two different Tetris rule-sets) ---- */
#define TG_RULESET_NUMPY 0      /* envs/tetris.py (the graded oracle) */
#define TG_RULESET_FUNCTIONAL 1 /* functional/core.py + envs/tetris_fn.py */

/* ---- observation surfaces ---- */
#define TG_OBS_NONE 0  /* nothing written */
#define TG_OBS_BOARD 1 /* u8 [B][H][W]  = crop_padding(obs["board"]): locked cell values + active piece
                          (envs/tetris.py:543-564, 514-523) */
#define TG_OBS_DICT 2  /* NUMPY only. u8 board[B][Hp][Wp] | mask[B][Hp][Wp] | holder[B][p][p] |
                          queue[B][p][p*Q]  (envs/tetris.py:566-615) */
#define TG_OBS_FN 3    /* i8 [B][H][W] in {-1,0,1} (envs/tetris_fn.py:137-158) */
#define TG_OBS_RGB 4   /* NUMPY only. u8 [B][Hp][Wp + p*max(1,Q)][3]: RgbObservation — board left, queue strip top
                          right, holder bottom right, through the pixel colour table
                          (wrappers/observation.py:38-74, envs/tetris.py:45-75) */

/* ---- autoreset ---- */
#define TG_AUTORESET_NONE 0      /* caller resets via tg_reset(mask) — reference behaviour */
#define TG_AUTORESET_SAME_STEP 1 /* env is reset inside the step that terminated it; the observation
                                    returned is the reset observation (gymnasium<=0.29 SyncVectorEnv) */
#define TG_AUTORESET_NEXT_STEP 2 /* the step after termination ignores its action and resets
                                    (gymnasium 1.x default) */

/* ---- action sets ---- */
#define TG_ACTIONS_BASE 0    /* NUMPY: mappings/actions.py:12-19 (8 ids); FUNCTIONAL: tetris_fn.py:470-478 (7 ids) */
#define TG_ACTIONS_GROUPED 1 /* NUMPY only: column*4+rotation → teleport + hard drop (wrappers/grouped.py:209-269) */

/* ---- memory kinds for caller-provided action buffers ---- */
#define TG_MEM_HOST 0
#define TG_MEM_DEVICE 1

/* ---- tg_config.flags ---- */
#define TG_FLAG_EPISODE_STATS 1 /* keep device-side episode return/length (RecordEpisodeStatistics) */
#define TG_FLAG_FINAL_OBS 2     /* SAME_STEP autoreset, TG_OBS_BOARD / TG_OBS_FN: also write the terminal observation to
                                 * TG_BUF_FINAL_OBS (gymnasium's infos["final_observation"]; valid where terminated) */
#define TG_FLAG_FEATURES 4      /* allocate the feature buffers used by tg_features */
#define TG_FLAG_CUSTOM_ACTIONS 8  /* NUMPY: tg_config.action_ids holds a non-stock ActionsMapping (mappings/actions.py:5-19) */
#define TG_FLAG_CUSTOM_REWARDS 16 /* NUMPY: tg_config.rewards holds a non-stock RewardsMapping (mappings/rewards.py:5-15) */

/* ---- error codes ---- */
#define TG_OK 0
#define TG_EINVAL (-1)      /* bad argument / unsupported configuration */
#define TG_EHIP (-2)        /* a HIP runtime call failed */
#define TG_ENOMEM (-3)
#define TG_EUNSUPPORTED (-4) /* e.g. board size not compiled in, custom tetromino tables */

/* ---- environment variables the library reads (all of them; everything else is tg_config) ----
 *   TG_FUSE=1..16      overrides tg_config.fuse        } launch knobs: results never depend on them; the tests set them to
 *   TG_RNG_EVERY=1..3  overrides tg_config.rng_every   } drive the rare paths (in-line bag shuffle, RNG role in every launch,
 *   TG_RING_FILL=0..3  overrides tg_config.ring_fill_p1 - 1 } one transition per launch of tg_step_k)
 *   TG_FN_LAZY=0..7    FUNCTIONAL look-ahead test hook: bits 0-1 when a lock sorts the reset bag ahead (0 tall stack only,
 *                      1 never, 2 always), bit 2 the key chain never catches up behind a reset
 *   TG_SPLIT=0         never pick the instantiations with two stream waves per workgroup (tests: both builds must agree)
 *   TG_SEQ=0           never pick the instantiations whose stream wave walks two env groups (boards above 640 cells, packed, one
 *                      transition per launch; A/B runs and tests: both builds must agree)
 *   TG_PACKED=0|1      forbid / force the nibble-packed board format (default: handles of >= 20 Mi cells)
 *   TG_KERNEL=bits     anything else needs the test-only legacy library (make -C csrc legacy) and is refused by the product
 *   TG_CSRC_DIR, TG_CACHE_DIR   where the kernel headers are / where compiled code objects of non-built-in board sizes are
 *                      cached (default: <library dir>/csrc, ~/.cache/tetris_gym_hip)
 * (Python side: TG_LIB = path of the library to load — the test-only legacy build, or, together with TG_LIB_IS_CPU_CHECKER=1, the CPU
 * checker's implementation of this ABI (oracle/tg_oracle_abi.c: a bisect tool, refused without the second variable) —,
 * TG_HIP_RUNTIME = system | torch | auto, TG_VERBOSE.) */

typedef struct tg_config {
  int32_t struct_size; /* = sizeof(tg_config); ABI guard */
  int32_t width;       /* playfield columns (reference default 10) */
  int32_t height;      /* playfield rows    (reference default 20) */
  int32_t padding;     /* must be 4 (reference: max tetromino extent, envs/tetris.py:130) */
  int32_t queue_size;  /* NUMPY: visible queue length 1..7 (reference 4); FUNCTIONAL: EnvConfig.queue_size = the bag length, 1..7
                          (functional/queue.py:35, 67; reference default 7) */
  int32_t gravity;     /* 0/1 */
  int32_t ruleset;     /* TG_RULESET_* */
  int32_t obs_mode;    /* TG_OBS_* */
  int32_t autoreset;   /* TG_AUTORESET_* */
  int32_t action_set;  /* TG_ACTIONS_* */
  int32_t batch;       /* number of environments owned by this handle */
  int32_t device;      /* HIP device ordinal */
  int32_t env_offset;  /* global index of env 0 of this handle (multi-GPU sharding; informational) */
  int32_t grouped_terminate_on_illegal; /* wrappers/grouped.py:49 (default 1) */
  int32_t flags;       /* TG_FLAG_* */
  int32_t reserved;
  /* launch knobs of the step kernel (0 = default; the environment variables TG_FUSE / TG_RNG_EVERY / TG_RING_FILL override
   * them — test hooks).  Results never depend on them. */
  int32_t fuse;         /* transitions per launch of tg_step_k / tg_rollout / tg_grouped_rollout: 1..16, default 16 */
  int32_t rng_every;    /* NUMPY rules: one-transition launches carry the bag-shuffle look-ahead workgroups every n-th time
                           (1..3, default 3) */
  int32_t ring_fill_p1; /* NUMPY rules: 1 + look-ahead bag permutations kept per env (1..4, default 4 = 3 permutations);
                           smaller values exercise the in-line shuffle of the step kernel */
  int32_t streams;      /* > 1: spread every transition of a large batch over this many HIP streams (contiguous env shards of
                           >= 8 192 envs, one launch each; board / functional observations only).  0 / 1 = one launch (default:
                           measured faster); kept for the tests that pin the shard arithmetic */
  /* ---- ABI 4: the reference's two mapping dataclasses as per-handle DATA (envs/tetris.py:83-84, 189-190) ----
   * action_ids (TG_FLAG_CUSTOM_ACTIONS, else ignored = the stock 0..7): the values of ActionsMapping's fields in declaration
   *   order — move_left, move_right, move_down, rotate_clockwise, rotate_counterclockwise, hard_drop, swap, no_op
   *   (mappings/actions.py:12-19).  Tetris.step compares the incoming action with the fields in the order of its elif chain
   *   (envs/tetris.py:222-256: left, right, down, cw, ccw, swap, hard_drop, no_op; the first match wins, an id that matches
   *   no field only takes gravity) and skips gravity iff action == hard_drop (:259).  The action space stays Discrete(8)
   *   (:192), so only ids 0..7 are reachable; field values outside that range are legal and unreachable.  A mapping in
   *   which hard_drop shares its value with a field earlier in the chain is taken as the reference takes it: that id runs
   *   the other operation and then skips gravity (its own table entry since round 6).
   * rewards (TG_FLAG_CUSTOM_REWARDS, else ignored = the stock 1, 1, 0, -0.1): alife, clear_line, game_over,
   *   invalid_action (mappings/rewards.py:12-15).  A commit returns score(lines) + alife, or game_over when the piece
   *   already collides / the spawn collides (envs/tetris.py:461-474); clear_line is a field the reference never reads
   *   (score is rows^2 * width, :621-630) and is carried for completeness; invalid_action is the reward of a masked-out
   *   action of the grouped action set (wrappers/grouped.py:238).  The step kernel works in float32; the exact Python value
   *   (int + float in double precision) is what TG_BUF_HOST_PACK holds — recomputed from tg_step_out.flags / lines.
   *   Precision of what is NOT recomputed: tg_step_out.reward (what tg_fetch / Engine.step / Engine.rollout return) and the
   *   running episode return tg_episode_rec.ret — hence the "episode returns" array of TG_BUF_HOST_PACK and
   *   infos["episode"]["r"] of the vector adapter — are float32 sums of float32 rewards: exact for the stock mapping and every
   *   dyadic one, float32-accurate for a mapping such as alife = 0.1 (the per-step rewards beside them are exact doubles). */
  int32_t action_ids[8];
  double rewards[4];
} tg_config;

/* ---- per-env state records (tg_get_state / tg_set_state; also the device layout) ----
 * board: u8 [B][H][W], playfield only (bedrock padding is implicit: it is a collision convenience,
 *        envs/tetris.py:632-641), cell = 0 empty or the tetromino value 2..8. */
typedef struct tg_env_rec { /* 32 bytes */
  uint8_t piece;        /* active tetromino index 0..6 (I O T S Z J L) */
  uint8_t rot;          /* number of np.rot90(k=+1) applied, 0..3 */
  uint8_t x;            /* padded-board column of the matrix' left edge */
  uint8_t y;            /* padded-board row of the matrix' top edge */
  uint8_t val;          /* value written into the board on lock (piece + 2 unless poked) */
  uint8_t flags;        /* TG_F_* */
  uint8_t holder_piece; /* 0..6 or TG_NONE (NUMPY only) */
  uint8_t holder_rot;
  uint8_t holder_val;
  uint8_t bag_index;    /* NUMPY: BagRandomizer.index 0..6; FUNCTIONAL: queue_index 0..7 */
  uint8_t bag[7];       /* NUMPY: BagRandomizer.bag; FUNCTIONAL: State.queue */
  uint8_t queue[7];     /* NUMPY: visible TetrominoQueue (first queue_size entries); FUNCTIONAL: unused */
  float score;          /* FUNCTIONAL: State.score (float32). NUMPY: 0 through tg_get_state; on the device these two words are private */
  uint32_t reserved;    /* (NUMPY: the mask of playfield rows that consist of bedrock cells — value 1 — only, which the reference's
                           clear_filled_rows never takes, envs/tetris.py:493; rebuilt from the board behind every upload / reset) */
} tg_env_rec;

#define TG_NONE 0xFF
#define TG_F_GAME_OVER 0x01
#define TG_F_HAS_SWAPPED 0x02
#define TG_F_PENDING_RESET 0x04 /* NEXT_STEP autoreset: reset on the next step */
#define TG_F_FULLROW 0x08       /* device-private: the stored board may hold an uncleared full row (only after
                                   tg_set_state with such a board); stripped by tg_get_state */

typedef struct tg_rng_rec { /* 48 bytes */
  /* NUMPY: numpy PCG64 bit-generator state (state, inc as 128-bit little-endian pairs) + the
   * buffered 32-bit half used by next_uint32 (numpy/random/src/pcg64/pcg64.h pcg64_next32). */
  uint64_t state_lo, state_hi;
  uint64_t inc_lo, inc_hi;
  uint32_t uinteger;
  uint32_t has_uint32;
  /* FUNCTIONAL: State.rng_key (threefry2x32 key data) */
  uint32_t key[2];
} tg_rng_rec;

/* ---- engine-owned device buffers (see tg_device_ptr / tg_fetch) ---- */
#define TG_BUF_OBS 0        /* layout per obs_mode */
#define TG_BUF_STEP_OUT 1   /* tg_step_out [B]: reward, terminated, info["lines_cleared"] of the last step */
#define TG_BUF_ACTIONS 2    /* i32 [B] staging buffer used for TG_MEM_HOST actions */
#define TG_BUF_BOARD 3      /* u8 [B][H][W] state boards.  Handles that keep their board as nibbles (>= 20 Mi cells, or TG_PACKED=1)
                             * bring this buffer up to date only when it is asked for (tg_device_ptr / tg_fetch* / tg_get_state*):
                             * a device pointer obtained earlier shows the boards as of THAT call — valid until the next
                             * transition; ask again (Engine.device_array views do so at every export) */
#define TG_BUF_REC 4        /* tg_env_rec [B] */
#define TG_BUF_RNG 5        /* tg_rng_rec [B].  NUMPY rules: a transition leaves the generator state in the engine's
                             * look-ahead ring; tg_get_state / tg_fetch / tg_device_ptr of this buffer first enqueue the
                             * kernel that writes the generators the reference would hold now (tg_sync before reading the
                             * device pointer from another stream) */
#define TG_BUF_LEGAL_MASK 6 /* u8 [B][4*W]  grouped legal-action mask (TG_ACTIONS_GROUPED) */
#define TG_BUF_FINAL_OBS 7  /* SAME_STEP autoreset: observation of the terminal state (valid where terminated) */
#define TG_BUF_EPISODE 8    /* tg_episode_rec [B] (TG_FLAG_EPISODE_STATS) */
#define TG_BUF_PLACEMENT_FEATS 9 /* u8 [B][4W][W+3] (TG_FLAG_FEATURES): per-placement feature vectors, see tg_features */
#define TG_BUF_FEATS 10     /* u8 [B][W+3] (TG_FLAG_FEATURES): FeatureVectorObservation of the current observation */
#define TG_BUF_HOST_PACK 11 /* what a gymnasium-contract step() returns beside the observation, in the dtypes and the array-
                             * per-field layout the host hands out (tg_host_pack_layout): rewards f64[B] | episode return f64[B]
                             * | lines_cleared i32[B] | episode length i32[B] | terminated u8[B] — the episode fields hold the
                             * finished episode's statistics where `terminated`, zero elsewhere (the RecordEpisodeStatistics
                             * convention of examples/train_ppo.py:246-260), the invalid-action reward of the grouped action set
                             * is the double -0.1.  Packed on demand from TG_BUF_STEP_OUT / TG_BUF_EPISODE by a small kernel
                             * when the buffer is fetched (tg_fetch / tg_fetch_many / tg_step_fetch_many / tg_device_ptr); it sits
                             * directly behind TG_BUF_OBS in device memory so that both leave in ONE device-to-host copy. */
#define TG_BUF_COUNT 12

typedef struct tg_step_out { /* 8 bytes, written by the step kernel with one store per env */
  float reward;        /* NUMPY: integer-valued with the stock RewardsMapping (envs/tetris.py:468-474); grouped illegal = -0.1f */
  uint8_t terminated;
  uint8_t lines;       /* info["lines_cleared"] */
  uint8_t flags;       /* TG_OUT_* (NUMPY rule-set) */
  uint8_t reserved;
} tg_step_out;
#define TG_OUT_COMMITTED 1 /* commit_active_tetromino ran in this step (envs/tetris.py:450-479): reward = terminated ?
                              rewards.game_over : score(lines) + rewards.alife */
#define TG_OUT_ILLEGAL 2   /* grouped action set: the action was masked out, reward = rewards.invalid_action */

typedef struct tg_episode_rec { /* 16 bytes: device-side RecordEpisodeStatistics */
  float ret;           /* running return of the current episode */
  uint32_t len;        /* running length */
  float last_ret;      /* return of the last finished episode */
  uint32_t last_len;   /* length of the last finished episode */
} tg_episode_rec;

typedef struct tg_handle tg_handle;

int tg_abi_version(void);

/* Number of visible HIP devices (does not initialise a context beyond hipGetDeviceCount). */
int tg_device_count(void);

/* PCI bus id ("0000:c1:00.0") of HIP device `device` into buf (NUL-terminated): lets a launcher find the device's NUMA node
 * (/sys/bus/pci/devices/<id>/numa_node) and bind its rank's host thread next to it (bench.py --gpus N).  TG_EINVAL / TG_EHIP. */
int tg_device_pci_bus_id(int device, char* buf, int len);

/* Create an engine instance: allocates all device state (SoA) for cfg->batch envs on cfg->device.
 * Envs are NOT reset by create; call tg_reset. */
int tg_create(const tg_config* cfg, tg_handle** out);
void tg_destroy(tg_handle* h);

/* Last error message for this handle (h == NULL: last tg_create failure on this thread). */
const char* tg_last_error(const tg_handle* h);

/* Reset envs.  seeds: host u64[B] or NULL; mask: host u8[B] or NULL (NULL = all envs).
 * NUMPY: seeds[i] > 0 reseeds env i's generator exactly like Tetris.reset(seed=seeds[i])
 *        (SeedSequence → PCG64); seeds == NULL or seeds[i] == 0 keeps the running generator
 *        (components/tetromino_randomizer.py:40-46).  An env that was never seeded gets
 *        SeedSequence(1 + env_offset + i).
 * FUNCTIONAL: seeds[i] = ((u64)key[0] << 32) | key[1]  is the PRNG key passed to reset
 *        (envs/tetris_fn.py:318-328); seeds == NULL re-uses State.rng_key. */
int tg_reset(tg_handle* h, const uint64_t* seeds, const uint8_t* mask);

/* One transition for every env.  actions: i32[B] in host or device memory (kind = TG_MEM_*).
 * Asynchronous on the handle's stream; results land in TG_BUF_STEP_OUT / TG_BUF_OBS. */
int tg_step(tg_handle* h, const int32_t* actions, int kind);

/* K consecutive transitions with pre-generated actions i8 [K][B] resident in DEVICE memory (rollout helper, no host
 * round trip; the outputs of the last transition remain in the buffers, per-episode sums go to TG_BUF_EPISODE when
 * enabled).  The default kernel runs up to tg_config.fuse (default 16) transitions per launch with the per-env state
 * resident on chip in between; the observation is written after every transition.
 * Results are identical to K calls of tg_step. */
int tg_step_k(tg_handle* h, const int8_t* actions_dev, int K);

/* tg_step_k that also keeps the trajectory: the observation of transition k goes to obs_traj_dev + k * B * obs_bytes
 * (layout of the handle's obs_mode; TG_BUF_OBS still receives the last one) and its outputs to out_traj_dev[k][B].
 * Both are caller-owned DEVICE buffers; either may be NULL.  What a random-policy data collector or a search that
 * rolls out action sequences needs from `for k: obs, r, d, _, info = env.step(a[k])` (envs/tetris.py:203-272). */
int tg_rollout(tg_handle* h, const int8_t* actions_dev, int K, uint8_t* obs_traj_dev, tg_step_out* out_traj_dev);

/* Device pointer and size of an engine-owned buffer. */
int tg_device_ptr(tg_handle* h, int which, void** ptr, size_t* bytes);

/* Synchronous copy of an engine-owned buffer to host memory (waits for the stream). */
int tg_fetch(tg_handle* h, int which, void* host_dst, size_t bytes);

/* Byte offsets of the five arrays of TG_BUF_HOST_PACK for a config, in the order listed there, and the buffer's size in
 * offsets[5] (every array starts 16-byte aligned). */
int tg_host_pack_layout(const tg_config* cfg, size_t offsets[6]);

/* m buffers in one call and ONE wait: all copies are enqueued, then the stream is synchronised once (tg_fetch costs a
 * synchronisation per buffer: three per step of a gymnasium-style step()).  Requests that are neighbours in device memory
 * AND in host memory at the same distance (TG_BUF_OBS followed by TG_BUF_HOST_PACK into one page-locked slab laid out like
 * the device side: host_dst[1] - host_dst[0] == their device distance) are merged into a single copy.  src_off (or NULL):
 * byte offset into each buffer — a range of a buffer, e.g. one array of the dict observation of one shard straight into its
 * slice of the adapter's whole-batch array.  TG_BUF_RNG, TG_BUF_HOST_PACK, TG_BUF_BOARD and (grouped action set) TG_BUF_LEGAL_MASK
 * are brought up to date for the state as it stands before they are copied. */
int tg_fetch_many(tg_handle* h, int m, const int* which, void* const* host_dst, const size_t* bytes, const size_t* src_off);

/* One transition on n handles (one per GPU, or several shards of one GPU) and the fetches that follow it, from ONE host
 * thread and with one wait per handle: phase 1 enqueues, handle by handle, the action upload, the step launch and the m
 * device-to-host copies (nothing in it blocks when the destinations are page-locked); phase 2 waits for every stream.
 * actions[k]: i32[batch of handle k] in host or device memory (kind), or actions == NULL to fetch only.  which[m] is
 * the same for every handle; host_dst, bytes and src_off (or NULL) are [n][m] row-major.  What gymnasium.vector.SyncVectorEnv.step does
 * with a Python loop over envs (examples/train_ppo.py:240-248), for all GPUs of a node in one call. */
int tg_step_fetch_many(int n, tg_handle* const* hs, const void* const* actions, int kind, int m, const int* which,
                       void* const* host_dst, const size_t* bytes, const size_t* src_off);

/* Full state snapshot / restore (host buffers; any pointer may be NULL to skip).
 * board u8[B][H][W], rec tg_env_rec[B], rng tg_rng_rec[B].  (envs/tetris.py:681-708)
 * Handles that keep their board as nibbles (boards above 20 Mi cells, or TG_PACKED=1) refuse cell values > 15.  Functional
 * rule-set on such a handle: a lock is `board += id` (functional/core.py:103-121); should a piece uploaded on top of filled
 * cells produce a sum > 15, the next call that waits for the stream returns TG_EINVAL (never in play; TG_PACKED=0 handles keep
 * bytes and hold such sums like the reference's int8 board). */
int tg_get_state(tg_handle* h, uint8_t* board, tg_env_rec* rec, tg_rng_rec* rng);
int tg_set_state(tg_handle* h, const uint8_t* board, const tg_env_rec* rec, const tg_rng_rec* rng);

/* The same between DEVICE buffers of the caller (layouts as above; any pointer may be NULL), asynchronous on the handle's
 * stream: what a host program that keeps many states resident (a functional State per call, envs/tetris_fn.py:276-315; a
 * search that branches) needs instead of a round trip through host memory.  tg_set_state_dev rebuilds the engine's private
 * acceleration state exactly as tg_set_state does.  The host cannot range-check device-resident state, so a kernel does, right
 * behind the copies: whatever tg_set_state would refuse (piece / rotation / position / bag or queue entries out of range; for the
 * functional rule-set board bytes >= 0x80; for handles that keep the board as nibbles cell values > 15) is CLAMPED — nothing
 * then indexes out of range — and the next call on the handle that waits for the stream (tg_sync, tg_fetch*, tg_get_state,
 * tg_step_fetch_many) returns TG_EINVAL once, with the reason in tg_last_error. */
int tg_get_state_dev(tg_handle* h, uint8_t* board_dev, tg_env_rec* rec_dev, tg_rng_rec* rng_dev);
int tg_set_state_dev(tg_handle* h, const uint8_t* board_dev, const tg_env_rec* rec_dev, const tg_rng_rec* rng_dev);

/* Re-pack the observation buffer from the current state without stepping.  (tg_reset, tg_set_state and tg_set_state_dev re-pack
 * it themselves; between transitions the step kernel stores only what changed of the DICT / RGB surfaces' holder / queue strips.) */
int tg_observe(tg_handle* h);

/* The CURRENT state of any handle as observation surface `obs_mode` (TG_OBS_BOARD / TG_OBS_FN / TG_OBS_DICT / TG_OBS_RGB; the last two:
 * NUMPY rule-set), packed into a caller-owned DEVICE buffer of at least batch x (bytes per env of that surface) — whatever
 * obs_mode the handle was created with; TG_BUF_OBS is not touched.  This is what an ObservationWrapper around an env
 * computes (RgbObservation: wrappers/observation.py:38-74) without a second handle that mirrors the env.  Asynchronous on the
 * handle's stream (tg_dev_download / tg_sync order behind it). */
int tg_observe_as(tg_handle* h, int obs_mode, void* obs_dev, size_t bytes);

/* Grouped action surface: recompute TG_BUF_LEGAL_MASK and, if boards_dev != NULL, write the wrapper's observation —
 * all 4*W placement boards u8 [B][4W][Hp][Wp] (rotate, centre, hard drop from y = 0, project, clear rows; illegal
 * placement = all ones, game-over placement = all zeros) — into that caller-owned DEVICE buffer, for the current
 * state (wrappers/grouped.py:124-207).  HBM-write bound: 4W*Hp*Wp bytes per env (17 280 B at 10x20). */
int tg_grouped_enumerate(tg_handle* h, uint8_t* boards_dev);

/* Grouped action surface: draw one action per env uniformly from its legal mask into TG_BUF_ACTIONS (device-side
 * random policy for benchmarks / epsilon-greedy exploration: examples/train_lin_grouped.py:284-285). Deterministic in
 * (seed, counter, env). Step afterwards with tg_step(h, <TG_BUF_ACTIONS pointer>, TG_MEM_DEVICE). */
int tg_grouped_sample(tg_handle* h, uint32_t seed, uint32_t counter);

/* K transitions of the uniform-random-LEGAL policy of the grouped action set, drawn on the device: transition k uses
 * exactly the action tg_grouped_sample(h, seed, counter0 + k) would draw for the state at that point (so the result
 * equals K x {tg_grouped_sample; tg_step}).  Optional caller-owned DEVICE outputs: the drawn actions i8 [K][B], the
 * observation and output trajectories as in tg_rollout.  The default kernel draws inside the step kernel and runs up
 * to tg_config.fuse transitions per launch (BASELINE config 5: 20x40, queue 7, random legal placements). */
int tg_grouped_rollout(tg_handle* h, uint32_t seed, uint32_t counter0, int K, int8_t* actions_out_dev,
                       uint8_t* obs_traj_dev, tg_step_out* out_traj_dev);

/* Feature observations of the current state (NUMPY rule-set), written to TG_BUF_PLACEMENT_FEATS / TG_BUF_FEATS:
 *  - per placement (column*4 + rotations): heights[W], max height, holes, bumpiness of the board that placement
 *    produces, i.e. GroupedActionsObservations(env, observation_wrappers=[FeatureVectorObservation(env)])
 *    (wrappers/grouped.py:124-207 + wrappers/observation.py:236-278) — without building the 4W boards;
 *  - the same 13-entry vector for the current observation (info["board"] of the wrapper). */
int tg_features(tg_handle* h);

/* The same two feature observations for ANY NUMPY handle (no TG_FLAG_FEATURES needed), written to caller-owned DEVICE buffers:
 * placement_feats_dev u8 [B][4W][W+3] or NULL, feats_dev u8 [B][W+3] or NULL (FeatureVectorObservation(env).observation:
 * wrappers/observation.py:236-278).  Asynchronous on the handle's stream. */
int tg_features_as(tg_handle* h, uint8_t* placement_feats_dev, uint8_t* feats_dev);

int tg_sync(tg_handle* h);

/* Stream timing with HIP events recorded on the handle's own stream (bench.py). */
int tg_timer_start(tg_handle* h);
int tg_timer_stop(tg_handle* h, float* elapsed_ms); /* records stop, synchronises, returns ms */

/* Symbol of the step-kernel instantiation a tg_step / tg_step_k / tg_rollout / tg_grouped_rollout call on this handle
 * launches (the name rocprofv3's kernel trace lists), e.g. "tg::step_kernel_bits<10, 20, 0, 0>".  bench.py reports it. */
int tg_step_kernel_name(tg_handle* h, int transitions_per_launch, int trajectory, int sampled, char* buf, size_t n);

/* Host utility: numpy SeedSequence(seed) → PCG64 initial state, as tg_rng_rec (pinned by tests). */
int tg_seed_pcg64(uint64_t seed, tg_rng_rec* out);

/* Host utility: the 16-bit cell mask (bit 4*row + col of the 4x4 box) of tetromino `piece` (0..6: I O T S Z J L) after `rot`
 * np.rot90(k=+1) rotations — the table every kernel uses (tests compare it with the reference's matrices,
 * envs/tetris.py:47-75).  Negative on bad arguments. */
int tg_shape_mask(int piece, int rot);

/* Sizes derived from a config (bytes per env of the observation buffer, number of actions). */
int tg_obs_bytes_per_env(const tg_config* cfg);
int tg_num_actions(const tg_config* cfg);

/* Device memory helpers so a ctypes-only host program needs no other GPU library. */
int tg_dev_alloc(tg_handle* h, size_t bytes, void** ptr);
int tg_dev_free(tg_handle* h, void* ptr);
/* Page-locked host memory (hipHostMalloc): a tg_fetch / tg_dev_download into it is one DMA at PCIe rate instead of a staged
 * copy through the runtime's bounce buffers — what the host-NumPy observation path of the gymnasium.vector adapter uses
 * (TetrisVectorEnv(copy=False); replaces the per-step `np.stack` of SyncVectorEnv, gymnasium/vector/sync_vector_env.py). */
int tg_host_alloc(tg_handle* h, size_t bytes, void** ptr);
int tg_host_free(tg_handle* h, void* ptr); /* h may be NULL (memory that outlived its engine: nothing to wait for) */
int tg_dev_upload(tg_handle* h, void* dst_dev, const void* src_host, size_t bytes);
int tg_dev_download(tg_handle* h, void* dst_host, const void* src_dev, size_t bytes);
/* device → device, asynchronous on the handle's stream (e.g. keep a copy of TG_BUF_OBS before the next transition overwrites it) */
int tg_dev_copy(tg_handle* h, void* dst_dev, const void* src_dev, size_t bytes);

/* Stream-copy kernel (float4 grid-stride) used by bench.py to report the practical HBM ceiling. */
int tg_stream_copy_gbps(tg_handle* h, size_t bytes, int iters, float* gbps);

#ifdef __cplusplus
}
#endif
#endif /* TETRIS_GYM_HIP_H */
