/*
 * tg_oracle_abi.c — the product's C ABI (include/tetris_gym_hip.h) implemented by the CPU restatement.
 * TEST INFRASTRUCTURE ONLY ("the identical ABI is implemented by the CPU restatement").
 *
 * `make -C oracle libtg_oracle_abi.so` links this shim with tg_oracle.c; "device memory" is host memory.  One binding then
 * drives either library:
 *     TG_LIB=$PWD/oracle/libtg_oracle_abi.so TG_LIB_IS_CPU_CHECKER=1 python -m pytest tests/test_gpu_parity.py -m gpu -k "replays_reference_trace"
 * runs the engine's replay tests on a box without a GPU — a one-line bisect tool when engine and fixtures disagree (is it the
 * kernel, or the Python above the ABI?).  The product never loads it: `_native.load()` refuses a TG_LIB that exports `tgo_create` unless
 * TG_LIB_IS_CPU_CHECKER=1 is set as well, tests/test_native_abi.py asserts that nothing under tetris_gymnasium_amd/ names the oracle,
 * and bench.py cannot time it (tg_stream_copy_gbps is TG_EUNSUPPORTED, tg_step_kernel_name says what this library is).
 *
 * Implemented: create / destroy / reset / step / step_k / rollout / get_state / set_state (+ _dev) / observe / observe_as /
 * features / features_as / grouped_enumerate / fetch / fetch_many / step_fetch_many / device_ptr / host_pack / episode statistics /
 * the memory helpers.  Not implemented (TG_EUNSUPPORTED): tg_grouped_sample, tg_grouped_rollout (the engine's own sampling hash),
 * TG_FLAG_FINAL_OBS, tg_stream_copy_gbps, tg_device_pci_bus_id.
 * Round 6, this container (no GPU): `TG_LIB=... TG_LIB_IS_CPU_CHECKER=1 pytest tests -m gpu` -> 315 of the 426 GPU-marked tests pass on
 * this library in 6 minutes; the other 111 assert engine-only things (step-kernel names and FEAT bits, nibble-packed boards, the final
 * observation, the device-side sampling hash, run-time compilation refusals, host-time budgets).
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tg_oracle.h"

#define NTHREADS 4

struct tg_handle {
  tgo_batch* b;
  tg_config cfg;
  int B, W, H, Q;
  size_t obs_bytes;       /* per env */
  uint8_t* obs;           /* TG_BUF_OBS */
  tg_step_out* out;       /* TG_BUF_STEP_OUT */
  int32_t* actions;       /* TG_BUF_ACTIONS */
  uint8_t* board;         /* TG_BUF_BOARD (refreshed on demand) */
  tg_env_rec* rec;        /* TG_BUF_REC */
  tg_rng_rec* rng;        /* TG_BUF_RNG */
  uint8_t* legal;         /* TG_BUF_LEGAL_MASK */
  tg_episode_rec* ep;     /* TG_BUF_EPISODE or NULL */
  uint8_t* pfeats;        /* TG_BUF_PLACEMENT_FEATS or NULL */
  uint8_t* feats;         /* TG_BUF_FEATS or NULL */
  uint8_t* pack;          /* TG_BUF_HOST_PACK */
  size_t pack_off[6];
  double* rew64;          /* scratch: rewards as Python computes them */
  float* rew; uint8_t* term; uint8_t* lines; uint8_t* flags;
  struct timespec t0;
  char err[320];
};

static char g_create_err[320];

static int failv(tg_handle* h, int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(h ? h->err : g_create_err, 320, fmt, ap);
  va_end(ap);
  return code;
}

static size_t obs_bytes_for(const tg_config* c, int mode) {
  const size_t cells = (size_t)c->width * c->height, pc = (size_t)(c->width + 8) * (c->height + 4);
  switch (mode) {
    case TG_OBS_BOARD:
    case TG_OBS_FN: return cells;
    case TG_OBS_DICT: return 2 * pc + 16 + 16 * (size_t)c->queue_size;
    case TG_OBS_RGB: return 3 * (size_t)(c->height + 4) * (size_t)(c->width + 8 + 4 * (c->queue_size > 1 ? c->queue_size : 1));
    default: return 0;
  }
}

static void pack_layout(size_t B, size_t* off) {
#define UP(v) (((v) + 15) / 16 * 16)
  off[0] = 0;
  off[1] = UP(off[0] + 8 * B);
  off[2] = UP(off[1] + 8 * B);
  off[3] = UP(off[2] + 4 * B);
  off[4] = UP(off[3] + 4 * B);
  off[5] = UP(off[4] + B);
#undef UP
}

int tg_abi_version(void) { return TG_ABI_VERSION; }
int tg_device_count(void) { return 1; } /* (the host: what this library computes on) */
int tg_device_pci_bus_id(int device, char* buf, int len) { (void)device; if (buf && len > 0) buf[0] = 0; return TG_EUNSUPPORTED; }
const char* tg_last_error(const tg_handle* h) { return h ? h->err : g_create_err; }
int tg_obs_bytes_per_env(const tg_config* cfg) { return cfg ? (int)obs_bytes_for(cfg, cfg->obs_mode) : TG_EINVAL; }
int tg_num_actions(const tg_config* cfg) {
  if (!cfg) return TG_EINVAL;
  if (cfg->action_set == TG_ACTIONS_GROUPED) return 4 * cfg->width;
  return cfg->ruleset == TG_RULESET_NUMPY ? 8 : 7;
}
int tg_host_pack_layout(const tg_config* cfg, size_t offsets[6]) {
  if (!cfg || !offsets || cfg->batch <= 0) return TG_EINVAL;
  pack_layout((size_t)cfg->batch, offsets);
  return TG_OK;
}
int tg_seed_pcg64(uint64_t seed, tg_rng_rec* out) { if (!out) return TG_EINVAL; tgo_seed_pcg64(seed, out); return TG_OK; }

int tg_shape_mask(int piece, int rot) {
  if (piece < 0 || piece > 6 || rot < 0 || rot > 3) return TG_EINVAL;
  uint8_t m[16];
  const int n = tgo_np_matrix(piece, rot, m);
  int mask = 0;
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c)
      if (m[r * n + c]) mask |= 1 << (4 * r + c);
  return mask;
}

static void repack_obs(tg_handle* h) {
  if (h->cfg.obs_mode != TG_OBS_NONE) tgo_observe(h->b, h->cfg.obs_mode, h->obs, NTHREADS);
}

void tg_destroy(tg_handle* h) {
  if (!h) return;
  tgo_destroy(h->b);
  free(h->obs); free(h->out); free(h->actions); free(h->board); free(h->rec); free(h->rng); free(h->legal); free(h->ep);
  free(h->pfeats); free(h->feats); free(h->pack); free(h->rew64); free(h->rew); free(h->term); free(h->lines); free(h->flags);
  free(h);
}

int tg_create(const tg_config* cfg, tg_handle** out) {
  if (!cfg || !out) return failv(NULL, TG_EINVAL, "tg_create: null argument");
  if (cfg->struct_size != (int32_t)sizeof(tg_config)) return failv(NULL, TG_EINVAL, "tg_create: tg_config.struct_size mismatch");
  if (cfg->padding != 4) return failv(NULL, TG_EUNSUPPORTED, "tg_create: padding must be 4");
  if (cfg->batch <= 0 || cfg->width < 4 || cfg->width > 32 || cfg->height < 4 || cfg->height > 58)
    return failv(NULL, TG_EINVAL, "tg_create: batch / board size out of range");
  if (cfg->flags & TG_FLAG_FINAL_OBS) return failv(NULL, TG_EUNSUPPORTED, "tg_create (oracle ABI): TG_FLAG_FINAL_OBS is not implemented");
  if ((cfg->flags & (TG_FLAG_CUSTOM_ACTIONS | TG_FLAG_CUSTOM_REWARDS)) && cfg->ruleset != TG_RULESET_NUMPY)
    return failv(NULL, TG_EINVAL, "tg_create: ActionsMapping / RewardsMapping belong to the NUMPY rule-set (envs/tetris.py:83-84)");
  if ((cfg->obs_mode == TG_OBS_DICT || cfg->obs_mode == TG_OBS_RGB) && cfg->ruleset != TG_RULESET_NUMPY)
    return failv(NULL, TG_EINVAL, "tg_create: the dict / RGB observation belongs to the NUMPY rule-set");
  tg_handle* h = (tg_handle*)calloc(1, sizeof(tg_handle));
  if (!h) return failv(NULL, TG_ENOMEM, "tg_create: out of memory");
  h->cfg = *cfg;
  h->B = cfg->batch; h->W = cfg->width; h->H = cfg->height; h->Q = cfg->queue_size;
  h->b = tgo_create(cfg->ruleset, cfg->width, cfg->height, cfg->padding, cfg->queue_size, cfg->gravity, cfg->batch, cfg->env_offset);
  if (!h->b) { free(h); return failv(NULL, TG_ENOMEM, "tg_create: tgo_create failed"); }
  if (cfg->flags & (TG_FLAG_CUSTOM_ACTIONS | TG_FLAG_CUSTOM_REWARDS))
    tgo_set_mappings(h->b, (cfg->flags & TG_FLAG_CUSTOM_ACTIONS) ? cfg->action_ids : NULL,
                     (cfg->flags & TG_FLAG_CUSTOM_REWARDS) ? cfg->rewards : NULL);
  const size_t B = (size_t)h->B, NA = 4 * (size_t)h->W, NF = (size_t)h->W + 3;
  h->obs_bytes = obs_bytes_for(cfg, cfg->obs_mode);
  pack_layout(B, h->pack_off);
  h->obs = (uint8_t*)calloc(B * (h->obs_bytes ? h->obs_bytes : 1) + 64, 1);
  h->out = (tg_step_out*)calloc(B, sizeof(tg_step_out));
  h->actions = (int32_t*)calloc(B, 4);
  h->board = (uint8_t*)calloc(B * h->W * h->H, 1);
  h->rec = (tg_env_rec*)calloc(B, sizeof(tg_env_rec));
  h->rng = (tg_rng_rec*)calloc(B, sizeof(tg_rng_rec));
  h->legal = (uint8_t*)calloc(B * NA, 1);
  h->ep = (cfg->flags & TG_FLAG_EPISODE_STATS) ? (tg_episode_rec*)calloc(B, sizeof(tg_episode_rec)) : NULL;
  h->pfeats = (uint8_t*)calloc(B * NA * NF, 1);
  h->feats = (uint8_t*)calloc(B * NF, 1);
  h->pack = (uint8_t*)calloc(h->pack_off[5] + 64, 1);
  h->rew64 = (double*)calloc(B, 8);
  h->rew = (float*)calloc(B, 4); h->term = (uint8_t*)calloc(B, 1); h->lines = (uint8_t*)calloc(B, 1); h->flags = (uint8_t*)calloc(B, 1);
  *out = h;
  return TG_OK;
}

int tg_reset(tg_handle* h, const uint64_t* seeds, const uint8_t* mask) {
  if (!h) return TG_EINVAL;
  tgo_reset(h->b, seeds, mask);
  if (h->ep)
    for (int i = 0; i < h->B; ++i)
      if (!mask || mask[i]) { h->ep[i].ret = 0.0f; h->ep[i].len = 0; }
  repack_obs(h);
  return TG_OK;
}

/* one transition from int32 actions: outputs, episode statistics (as the step kernel keeps them), observation */
static int step_once(tg_handle* h, const int32_t* a) {
  const int na = tg_num_actions(&h->cfg);
  if (h->cfg.ruleset == TG_RULESET_NUMPY)
    for (int i = 0; i < h->B; ++i)
      if (a[i] < 0 || a[i] >= na) return failv(h, TG_EINVAL, "tg_step: action %d of env %d outside Discrete(%d)", a[i], i, na);
  uint8_t* pend = NULL;
  if (h->ep && h->cfg.autoreset == TG_AUTORESET_NEXT_STEP) {
    tgo_get_state(h->b, NULL, h->rec, NULL);
    pend = (uint8_t*)malloc((size_t)h->B);
    for (int i = 0; i < h->B; ++i) pend[i] = (h->rec[i].flags & TG_F_PENDING_RESET) ? 1 : 0;
  }
  if (h->cfg.action_set == TG_ACTIONS_GROUPED)
    tgo_grouped_step(h->b, a, h->rew, h->term, h->lines, NULL, h->cfg.grouped_terminate_on_illegal, h->cfg.autoreset, NTHREADS);
  else
    tgo_step(h->b, a, h->rew, h->term, h->lines, h->cfg.autoreset, NTHREADS);
  tgo_last_flags(h->b, h->flags);
  tgo_last_rewards(h->b, h->rew64);
  for (int i = 0; i < h->B; ++i) {
    h->out[i].reward = h->rew[i]; h->out[i].terminated = h->term[i]; h->out[i].lines = h->lines[i];
    h->out[i].flags = h->cfg.ruleset == TG_RULESET_NUMPY ? h->flags[i] : 0; h->out[i].reserved = 0;
    if (h->ep) {
      tg_episode_rec* e = &h->ep[i];
      if (pend && pend[i]) { e->ret = 0.0f; e->len = 0; }
      else {
        e->ret += h->rew[i]; e->len += 1;
        if (h->term[i]) { e->last_ret = e->ret; e->last_len = e->len; e->ret = 0.0f; e->len = 0; }
      }
    }
  }
  free(pend);
  repack_obs(h);
  return TG_OK;
}

int tg_step(tg_handle* h, const int32_t* actions, int kind) {
  if (!h || !actions) return h ? failv(h, TG_EINVAL, "tg_step: null actions") : TG_EINVAL;
  if (kind != TG_MEM_HOST && kind != TG_MEM_DEVICE) return failv(h, TG_EINVAL, "tg_step: unknown memory kind %d", kind);
  return step_once(h, actions);
}

int tg_rollout(tg_handle* h, const int8_t* actions_dev, int K, uint8_t* obs_traj_dev, tg_step_out* out_traj_dev) {
  if (!h || !actions_dev || K <= 0) return h ? failv(h, TG_EINVAL, "tg_rollout: bad arguments") : TG_EINVAL;
  if (h->B % 4 != 0) return failv(h, TG_EINVAL, "tg_rollout: batch must be a multiple of 4 (packed int8 action rows)");
  if (obs_traj_dev && h->cfg.obs_mode == TG_OBS_NONE) return failv(h, TG_EINVAL, "tg_rollout: the handle has no observation (TG_OBS_NONE)");
  const size_t B = (size_t)h->B, stride = B * h->obs_bytes;
  for (int k = 0; k < K; ++k) {
    for (size_t i = 0; i < B; ++i) {
      const int a = actions_dev[(size_t)k * B + i];
      /* (device-resident actions are not range-checked by the engine: the kernel takes `action & 7` of the base action set) */
      h->actions[i] = h->cfg.action_set == TG_ACTIONS_GROUPED ? a : (h->cfg.ruleset == TG_RULESET_NUMPY ? (a & 7) : a);
    }
    const int rc = step_once(h, h->actions);
    if (rc != TG_OK) return rc;
    if (obs_traj_dev) memcpy(obs_traj_dev + (size_t)k * stride, h->obs, stride);
    if (out_traj_dev) memcpy(out_traj_dev + (size_t)k * B, h->out, B * sizeof(tg_step_out));
  }
  return TG_OK;
}
int tg_step_k(tg_handle* h, const int8_t* actions_dev, int K) { return tg_rollout(h, actions_dev, K, NULL, NULL); }

int tg_grouped_sample(tg_handle* h, uint32_t seed, uint32_t counter) {
  (void)seed; (void)counter;
  return h ? failv(h, TG_EUNSUPPORTED, "tg_grouped_sample (oracle ABI): the engine's sampling hash is not restated here") : TG_EINVAL;
}
int tg_grouped_rollout(tg_handle* h, uint32_t seed, uint32_t counter0, int K, int8_t* a, uint8_t* o, tg_step_out* r) {
  (void)seed; (void)counter0; (void)a; (void)o; (void)r;
  if (h && K > 0 && h->cfg.action_set != TG_ACTIONS_GROUPED)
    return failv(h, TG_EINVAL, "tg_grouped_rollout: handle was not created with TG_ACTIONS_GROUPED");
  return h ? failv(h, TG_EUNSUPPORTED, "tg_grouped_rollout (oracle ABI): the engine's sampling hash is not restated here") : TG_EINVAL;
}

int tg_get_state(tg_handle* h, uint8_t* board, tg_env_rec* rec, tg_rng_rec* rng) {
  if (!h) return TG_EINVAL;
  tgo_get_state(h->b, board, rec, rng);
  if (rec)
    for (int i = 0; i < h->B; ++i) { rec[i].flags &= (uint8_t)~TG_F_FULLROW; rec[i].reserved = 0; }
  return TG_OK;
}
int tg_set_state(tg_handle* h, const uint8_t* board, const tg_env_rec* rec, const tg_rng_rec* rng) {
  if (!h) return TG_EINVAL;
  if (rec)
    for (int i = 0; i < h->B; ++i) {
      const tg_env_rec* r = &rec[i];
      const int xmax = h->W + 4 + (h->cfg.ruleset == TG_RULESET_NUMPY ? (r->piece == 1 ? 2 : (r->piece == 0 ? 0 : 1)) : 0);
      if (r->piece > 6 || r->rot > 3 || (r->holder_piece != TG_NONE && (r->holder_piece > 6 || r->holder_rot > 3)) ||
          r->bag_index > 7 || r->y > h->H + 3 || r->x > xmax)
        return failv(h, TG_EINVAL, "tg_set_state: record %d out of range (piece %d rot %d holder %d bag_index %d x %d y %d)", i, r->piece,
                     r->rot, r->holder_piece, r->bag_index, r->x, r->y);
    }
  tgo_set_state(h->b, board, rec, rng);
  repack_obs(h);
  return TG_OK;
}
int tg_get_state_dev(tg_handle* h, uint8_t* b, tg_env_rec* r, tg_rng_rec* g) { return tg_get_state(h, b, r, g); }
int tg_set_state_dev(tg_handle* h, const uint8_t* b, const tg_env_rec* r, const tg_rng_rec* g) { return tg_set_state(h, b, r, g); }

int tg_observe(tg_handle* h) { if (!h) return TG_EINVAL; repack_obs(h); return TG_OK; }
int tg_observe_as(tg_handle* h, int obs_mode, void* obs_dev, size_t bytes) {
  if (!h || !obs_dev) return h ? failv(h, TG_EINVAL, "tg_observe_as: null buffer") : TG_EINVAL;
  const size_t need = (size_t)h->B * obs_bytes_for(&h->cfg, obs_mode);
  if (need == 0) return failv(h, TG_EINVAL, "tg_observe_as: unknown obs_mode %d", obs_mode);
  if ((obs_mode == TG_OBS_DICT || obs_mode == TG_OBS_RGB) && h->cfg.ruleset != TG_RULESET_NUMPY)
    return failv(h, TG_EINVAL, "tg_observe_as: TG_OBS_DICT / TG_OBS_RGB are observations of the NUMPY rule-set");
  if (bytes < need) return failv(h, TG_EINVAL, "tg_observe_as: %zu bytes given, this surface takes %zu", bytes, need);
  tgo_observe(h->b, obs_mode, obs_dev, NTHREADS);
  return TG_OK;
}
int tg_grouped_enumerate(tg_handle* h, uint8_t* boards_dev) {
  if (!h) return TG_EINVAL;
  if (h->cfg.ruleset != TG_RULESET_NUMPY) return failv(h, TG_EINVAL, "tg_grouped_enumerate: NUMPY rule-set only");
  tgo_grouped_enumerate(h->b, h->legal, boards_dev, NTHREADS);
  return TG_OK;
}
int tg_features_as(tg_handle* h, uint8_t* placement_feats_dev, uint8_t* feats_dev) {
  if (!h) return TG_EINVAL;
  if (h->cfg.ruleset != TG_RULESET_NUMPY) return failv(h, TG_EINVAL, "tg_features_as: NUMPY rule-set only");
  if (placement_feats_dev) tgo_grouped_features(h->b, placement_feats_dev, NTHREADS);
  if (feats_dev) tgo_features(h->b, feats_dev, NTHREADS);
  return TG_OK;
}
int tg_features(tg_handle* h) {
  if (!h) return TG_EINVAL;
  if (!(h->cfg.flags & TG_FLAG_FEATURES)) return failv(h, TG_EINVAL, "tg_features: handle was not created with TG_FLAG_FEATURES");
  return tg_features_as(h, h->pfeats, h->feats);
}

static void refresh_pack(tg_handle* h) {
  const size_t B = (size_t)h->B;
  double* rw = (double*)(h->pack + h->pack_off[0]);
  double* er = (double*)(h->pack + h->pack_off[1]);
  int32_t* ln = (int32_t*)(h->pack + h->pack_off[2]);
  int32_t* el = (int32_t*)(h->pack + h->pack_off[3]);
  uint8_t* tm = h->pack + h->pack_off[4];
  for (size_t i = 0; i < B; ++i) {
    rw[i] = h->cfg.ruleset == TG_RULESET_NUMPY ? h->rew64[i] : (double)h->out[i].reward;
    ln[i] = h->out[i].lines; tm[i] = h->out[i].terminated;
    er[i] = (h->ep && tm[i]) ? (double)h->ep[i].last_ret : 0.0;
    el[i] = (h->ep && tm[i]) ? (int32_t)h->ep[i].last_len : 0;
  }
}

int tg_device_ptr(tg_handle* h, int which, void** ptr, size_t* bytes) {
  if (!h || which < 0 || which >= TG_BUF_COUNT) return h ? failv(h, TG_EINVAL, "tg_device_ptr: bad buffer id %d", which) : TG_EINVAL;
  const size_t B = (size_t)h->B, NA = 4 * (size_t)h->W, NF = (size_t)h->W + 3;
  void* p = NULL; size_t n = 0;
  switch (which) {
    case TG_BUF_OBS: p = h->obs; n = B * h->obs_bytes; break;
    case TG_BUF_STEP_OUT: p = h->out; n = B * sizeof(tg_step_out); break;
    case TG_BUF_ACTIONS: p = h->actions; n = B * 4; break;
    case TG_BUF_BOARD: tgo_get_state(h->b, h->board, NULL, NULL); p = h->board; n = B * h->W * h->H; break;
    case TG_BUF_REC: (void)tg_get_state(h, NULL, h->rec, NULL); p = h->rec; n = B * sizeof(tg_env_rec); break;
    case TG_BUF_RNG: tgo_get_state(h->b, NULL, NULL, h->rng); p = h->rng; n = B * sizeof(tg_rng_rec); break;
    case TG_BUF_LEGAL_MASK:
      if (h->cfg.action_set == TG_ACTIONS_GROUPED) tgo_grouped_enumerate(h->b, h->legal, NULL, NTHREADS);
      p = h->legal; n = B * NA; break;
    case TG_BUF_EPISODE: p = h->ep; n = h->ep ? B * sizeof(tg_episode_rec) : 0; break;
    case TG_BUF_PLACEMENT_FEATS: p = h->pfeats; n = (h->cfg.flags & TG_FLAG_FEATURES) ? B * NA * NF : 0; break;
    case TG_BUF_FEATS: p = h->feats; n = (h->cfg.flags & TG_FLAG_FEATURES) ? B * NF : 0; break;
    case TG_BUF_HOST_PACK: refresh_pack(h); p = h->pack; n = h->pack_off[5]; break;
    default: p = NULL; n = 0; break; /* TG_BUF_FINAL_OBS */
  }
  if (ptr) *ptr = p;
  if (bytes) *bytes = n;
  return TG_OK;
}

int tg_fetch_many(tg_handle* h, int m, const int* which, void* const* host_dst, const size_t* bytes, const size_t* src_off) {
  if (!h || m < 0 || (m && (!which || !host_dst || !bytes))) return h ? failv(h, TG_EINVAL, "tg_fetch_many: bad arguments") : TG_EINVAL;
  for (int j = 0; j < m; ++j) {
    void* p; size_t n;
    const int rc = tg_device_ptr(h, which[j], &p, &n);
    if (rc != TG_OK) return rc;
    const size_t off = src_off ? src_off[j] : 0;
    if (!p || off + bytes[j] > n) return failv(h, TG_EINVAL, "tg_fetch: buffer %d holds %zu bytes, asked for %zu at offset %zu", which[j], n, bytes[j], off);
    memcpy(host_dst[j], (const uint8_t*)p + off, bytes[j]);
  }
  return TG_OK;
}
int tg_fetch(tg_handle* h, int which, void* host_dst, size_t bytes) {
  void* dst[1] = {host_dst};
  return tg_fetch_many(h, 1, &which, dst, &bytes, NULL);
}
int tg_step_fetch_many(int n, tg_handle* const* hs, const void* const* actions, int kind, int m, const int* which,
                       void* const* host_dst, const size_t* bytes, const size_t* src_off) {
  if (n <= 0 || !hs) return TG_EINVAL;
  for (int k = 0; k < n; ++k) {
    if (actions) {
      const int rc = tg_step(hs[k], (const int32_t*)actions[k], kind);
      if (rc != TG_OK) return rc;
    }
    const int rc = tg_fetch_many(hs[k], m, which, host_dst ? host_dst + (size_t)k * m : NULL, bytes ? bytes + (size_t)k * m : NULL,
                                 src_off ? src_off + (size_t)k * m : NULL);
    if (rc != TG_OK) return rc;
  }
  return TG_OK;
}

int tg_sync(tg_handle* h) { return h ? TG_OK : TG_EINVAL; }
int tg_timer_start(tg_handle* h) { if (!h) return TG_EINVAL; clock_gettime(CLOCK_MONOTONIC, &h->t0); return TG_OK; }
int tg_timer_stop(tg_handle* h, float* elapsed_ms) {
  if (!h || !elapsed_ms) return TG_EINVAL;
  struct timespec t1;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  *elapsed_ms = (float)((t1.tv_sec - h->t0.tv_sec) * 1e3 + (t1.tv_nsec - h->t0.tv_nsec) * 1e-6);
  return TG_OK;
}
int tg_step_kernel_name(tg_handle* h, int tpl, int traj, int sampled, char* buf, size_t n) {
  (void)tpl; (void)traj; (void)sampled;
  if (!h || !buf || n == 0) return TG_EINVAL;
  snprintf(buf, n, "tgo::cpu_oracle<%d, %d, %d, 0>", h->W, h->H, h->cfg.ruleset);
  return TG_OK;
}
int tg_stream_copy_gbps(tg_handle* h, size_t bytes, int iters, float* gbps) {
  (void)bytes; (void)iters; (void)gbps;
  return h ? failv(h, TG_EUNSUPPORTED, "tg_stream_copy_gbps (oracle ABI): this is the CPU restatement, there is nothing to measure") : TG_EINVAL;
}

int tg_dev_alloc(tg_handle* h, size_t bytes, void** ptr) { (void)h; if (!ptr) return TG_EINVAL; *ptr = malloc(bytes ? bytes : 1); return *ptr ? TG_OK : TG_ENOMEM; }
int tg_dev_free(tg_handle* h, void* ptr) { (void)h; free(ptr); return TG_OK; }
int tg_host_alloc(tg_handle* h, size_t bytes, void** ptr) { return tg_dev_alloc(h, bytes, ptr); }
int tg_host_free(tg_handle* h, void* ptr) { (void)h; free(ptr); return TG_OK; }
int tg_dev_upload(tg_handle* h, void* d, const void* s, size_t n) { (void)h; memcpy(d, s, n); return TG_OK; }
int tg_dev_download(tg_handle* h, void* d, const void* s, size_t n) { (void)h; memcpy(d, s, n); return TG_OK; }
int tg_dev_copy(tg_handle* h, void* d, const void* s, size_t n) { (void)h; memmove(d, s, n); return TG_OK; }
