// tg_engine.hip — host side of libtetris_gym_hip.so: the C ABI declared in include/tetris_gym_hip.h.
//
// One handle = one GPU = one HIP stream.  All per-env state lives in device memory as SoA arrays
// (board u8[B][H][W] | tg_env_rec[B] | tg_rng_rec[B]); outputs stay on the device until fetched.
// There is NO CPU fallback: every entry point that computes launches a HIP kernel.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "tg_kernel_bits.h"
#include "tg_kernel_features.h"
#include "tg_kernel_placements.h"
#ifdef TG_LEGACY_KERNELS  // test-only build (`make legacy`): the four earlier formulations of the step kernel, selected by TG_KERNEL
#include "legacy/tg_kernel_block.h"
#include "legacy/tg_kernel_lanes.h"
#include "legacy/tg_kernel_wave.h"
#endif

// ---- board sizes compiled in (W, H).  Extend the list and rebuild to support another size. ----
#ifndef TG_FOR_EACH_SIZE
#define TG_FOR_EACH_SIZE(X) X(10, 20) X(20, 40) X(6, 12) X(8, 16) X(10, 24) X(12, 24)
#endif

namespace {

thread_local std::string g_create_err;

// The kernels of one board size: template instantiations built into the library (TG_FOR_EACH_SIZE) or, for any other
// size, a module compiled by hiprtc at tg_create (rtc_launchers below).  Callables so that both fit one interface.
struct Launchers {
  std::function<void(const tg::FeatureParams&, hipStream_t)> features;
  std::function<void(const tg::KParams&, hipStream_t)> step;
  std::function<void(const tg::KParams&, char*, size_t)> step_name;
  std::function<void(const tg::ResetParams&, hipStream_t)> reset;
  std::function<void(const tg::ObserveParams&, hipStream_t)> observe;
  std::function<void(const tg::PrepareParams&, hipStream_t)> prepare;
  std::function<void(const tg::PlacementParams&, hipStream_t)> placements;
  size_t cols_bytes;  // per env: column masks of the bits kernel
  bool dict_ok, rgb_ok;  // the bits kernel packs TG_OBS_DICT / TG_OBS_RGB itself for this board size
};

inline int grid_for(int batch) { return (batch + tg::kWavesPerBlock - 1) / tg::kWavesPerBlock; }

// step kernel selection (TG_KERNEL): 16 = "bits": column-bitmask logic, one env per lane (default, tg_kernel_bits.h);
// (legacy build only) 8 = "block": two-phase workgroup kernel (legacy/tg_kernel_block.h); 4 = "quad": 16 envs per wave, 4 lanes per env;
// 1 = "lanes": 64 envs per wave, one per lane (both legacy/tg_kernel_lanes.h); 0 = "wave": one env per wavefront (legacy/tg_kernel_wave.h)
int g_kernel_lanes = 16;
#ifdef TG_LEGACY_KERNELS
constexpr int g_block_waves = 4;  // waves per workgroup of the block kernel (4 → 64 envs)
constexpr size_t g_lds_pad = 0;
constexpr int g_groups_per_wave = 1;
#endif
constexpr int g_stagger = 0;

int g_simds = 1024;  // SIMDs of the device (4 per CU)
int g_split = 1;     // TG_SPLIT=0: never pick the instantiations with two stream waves per workgroup (kFeatSplit)
int g_seq = 1;       // TG_SEQ=0: never pick the instantiations whose stream wave walks two env groups (kFeatSeq; A/B runs)

// Process-wide settings, read once (the first tg_create; the devices of a node are alike): the SIMD count the instantiation
// rules above compare grids against, TG_SPLIT, and TG_KERNEL (which the product library only accepts as "bits").
std::once_flag g_process_once;
std::string g_process_err;
void process_init(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) g_simds = 4 * prop.multiProcessorCount;
  if (const char* w = getenv("TG_SPLIT")) g_split = atoi(w) != 0;
  if (const char* w = getenv("TG_SEQ")) g_seq = atoi(w) != 0;
  if (const char* k = getenv("TG_KERNEL")) {
#ifdef TG_LEGACY_KERNELS
    g_kernel_lanes = strcmp(k, "wave") == 0    ? 0
                     : strcmp(k, "lanes") == 0 ? 1
                     : strcmp(k, "pair") == 0  ? 2
                     : strcmp(k, "quad") == 0  ? 4
                     : strcmp(k, "block") == 0 ? 8
                                               : 16;
#else
    if (strcmp(k, "bits") != 0)
      g_process_err = std::string("tg_create: TG_KERNEL=") + k + " needs the test-only library with the earlier step kernels "
                      "(make -C tetris_gymnasium_amd/csrc legacy; TG_LIB=<path>/libtetris_gym_hip_legacy.so)";
#endif
  }
}

// RNG-role workgroups of a launch (128 envs each)
int rng_blocks(const tg::KParams& p, const int RS) { return (RS == TG_RULESET_NUMPY && p.rng_role) ? (p.batch + 127) / 128 : 0; }
// the packed scalar argument of step_kernel_bits (layout: there)
uint32_t bits_hot_misc(const tg::KParams& p, const int RS, const int n_step_wgs) {
  (void)RS;
  return (uint32_t)n_step_wgs | ((p.act_i8 ? 1u : 0u) << 24) | ((uint32_t)(p.nsteps > 0 ? p.nsteps : 0) << 25);
}

// which instantiation of the bits kernel a launch with these parameters uses (FEAT flags, tg_kernel_bits.h)
int bits_feat_base(const tg::KParams& p, const int RS, const bool dict_ok, const bool rgb_ok, const int epw, const bool split_ok);
int bits_feat_rt(const tg::KParams& p, const int RS, const bool dict_ok, const bool rgb_ok, const int epw, const bool split_ok, const int rgb_helpers_max) {
  constexpr int M = tg::kFeatMulti, T = tg::kFeatTraj;
  const int feat = bits_feat_base(p, RS, dict_ok, rgb_ok, epw, split_ok);
  // boards above 640 cells (16 envs per stream wave), packed, one transition per launch: workgroups of 32 envs whose stream wave walks
  // two groups (tg_kernel_bits.h kFeatSeq; config 5: 22.6 -> see profiles/r06/)
  if (g_seq && (feat & tg::kFeatPacked) && !(feat & M) && epw == 16) return feat | tg::kFeatSeq;
  // two stream waves per workgroup (kFeatSplit) while all workgroups of the launch — the RNG role's included — are resident
  // with three waves each at the occupancy the instantiation is compiled for
  const int base = feat & ~tg::kFeatWide;
  // dict instantiations: a third wave for the mask array instead of a second stream wave (tg_kernel_bits.h kFeatMaskWave)
  // (one transition per launch: 4 096 envs 8.50 -> 6.73 us, 32 768 10.2 -> 8.65, 65 536 17.1 -> 15.6, 131 072 30.7 -> 29.9; launches of 16
  // transitions 13.7 -> 13.9 per transition, and two stream waves beside it only help at 4 096 envs: profiles/r04/ab_dict_mask_wave.log)
  if (RS == TG_RULESET_NUMPY && base == tg::kFeatDict) return feat | tg::kFeatMaskWave;
  // RGB, one transition per launch: two helper waves (the images by quarters).  Up to ~32 768 envs the launch is the chain of its packing
  // waves: 18.6 -> 12.6 us at 1 024 / 4 096 envs, 20.0 -> 14.2 at 16 384, 23.4 -> 18.4 at 32 768; 31.2 -> 30.8 at 65 536, 47.4 -> 49.0 at
  // 98 304, 86.6 -> 70.3 at 131 072 (profiles/r04/ab_rgb_helper_waves.log)
  // (six helper waves while there is one workgroup per CU at most)
  if (RS == TG_RULESET_NUMPY && base == tg::kFeatRgb && rgb_helpers_max >= 2)
    return feat | tg::kFeatMaskWave | ((rgb_helpers_max >= 6 && (long)((p.batch + epw - 1) / epw) <= (long)(g_simds / 4)) ? tg::kFeatMany : 0);
  // ... and in the instantiation with the transition loop up to 768 workgroups
  if (RS == TG_RULESET_NUMPY && base == (M | T | tg::kFeatSample | tg::kFeatRgb) && rgb_helpers_max >= 2 &&
      (long)((p.batch + epw - 1) / epw) <= 768L)
    return feat | tg::kFeatMaskWave;
  // ... and in the instantiation with the transition loop while the launch is bound by its chain, not by bytes: per transition of a
  // 16-transition launch 6.0 -> 3.7 us at 4 096 envs, 6.6 -> 4.2 at 32 768, 8.0 -> 7.2 at 49 152, but 10.1 -> 10.6 at 65 536
  if (RS == TG_RULESET_NUMPY && base == (M | T | tg::kFeatDict) && (long)((p.batch + epw - 1) / epw) * 4 <= 3L * g_simds)
    return feat | tg::kFeatMaskWave;
  if (!g_split || !split_ok || RS != TG_RULESET_NUMPY || (feat & tg::kFeatPacked) ||
      !(base == 0 || base == M || base == (M | T) || base == tg::kFeatDict || base == (M | T | tg::kFeatDict))) return feat;
  const int per_simd = tg::bits_waves_per_simd(RS, feat);
  const long cap = (long)(g_simds / 4) * (4 * per_simd / 3);
  const long wgs = (p.batch + epw - 1) / epw + rng_blocks(p, RS);
  return wgs <= cap ? (feat | tg::kFeatSplit) : feat;
}
int bits_feat_base(const tg::KParams& p, const int RS, const bool dict_ok, const bool rgb_ok, const int epw, const bool split_ok) {
  constexpr int M = tg::kFeatMulti, T = tg::kFeatTraj, D = tg::kFeatDict, S = tg::kFeatSample, F = tg::kFeatFinal, R = tg::kFeatRgb;
  // FUNCTIONAL, plain launches: when every workgroup of the launch is resident at 2 waves per SIMD anyway, the instantiation
  // compiled for that occupancy (256 VGPRs instead of 128) is the shorter instruction stream (8.4 -> 8.0 us at 65 536 envs;
  // beyond residency the 4-waves build wins: profiles/r03/ab_select_waves.log)
  // FUNCTIONAL, plain launches: two stream waves per workgroup (kFeatSplit, compiled for 3 waves per SIMD) while every
  // workgroup of the launch is resident that way — up to 65 536 envs on 256 CUs (7.53 -> 6.92 us per launch there)
  if (RS == TG_RULESET_FUNCTIONAL && p.pboard) {  // packed handles (batches beyond residency): the plain builds
    const bool trj = p.obs_traj || p.out_traj || p.act_traj;
    return (trj ? (M | T) : (p.nsteps > 1 ? M : 0)) | tg::kFeatPacked;
  }
  if (RS == TG_RULESET_FUNCTIONAL && g_split && !p.final_obs && split_ok && (long)(p.batch + epw - 1) / epw <= (long)g_simds) {
    const bool trj = p.obs_traj || p.out_traj || p.act_traj;
    return (trj ? (M | T) : (p.nsteps > 1 ? M : 0)) | tg::kFeatSplit;
  }
  if (RS == TG_RULESET_FUNCTIONAL && !p.final_obs && !(p.obs_traj || p.out_traj || p.act_traj) && split_ok &&
      (p.batch + epw - 1) / epw * 2 <= 2 * g_simds)
    return (p.nsteps > 1 ? M : 0) | tg::kFeatWide;
  const bool dict = p.obs_mode == TG_OBS_DICT && p.dict_in_kernel && dict_ok;
  const bool rgb = p.obs_mode == TG_OBS_RGB && p.dict_in_kernel && rgb_ok;
  const bool traj = p.obs_traj || p.out_traj || p.act_traj;
  if (p.final_obs) return (p.nsteps > 1 || traj) ? (M | T | F) : F;  // (TG_OBS_BOARD / TG_OBS_FN handles only)
  const int pk = p.pboard ? tg::kFeatPacked : 0;  // (packed handles: NUMPY, board / no observation, no final observation — tg_create)
  if (RS == TG_RULESET_NUMPY) {
    if (rgb) return (p.nsteps > 1 || traj || p.sample_mode) ? (M | T | S | R) : R;
    if (p.sample_mode) return (p.nsteps > 1 ? (M | T | S) : (T | S)) | pk;  // (one transition per launch: the kernel without the transition loop)
    if (dict) return (p.nsteps > 1 || traj) ? (M | T | D) : D;
  }
  if (traj) return M | T | pk;
  return (p.nsteps > 1 ? M : 0) | pk;
}

template <int W, int H, int RS>
int bits_feat(const tg::KParams& p) {
  return bits_feat_rt(p, RS, tg::BitsGeo<W, H>::DICT_OK, tg::BitsGeo<W, H>::RGB_OK, tg::BitsGeo<W, H>::EPW, tg::BitsGeo<W, H>::SPLIT_OK,
                      tg::bits_rgb_helper_count<W, H>(tg::kFeatRgb | tg::kFeatMaskWave | tg::kFeatMany));
}

template <int W, int H, int RS>
void step_name(const tg::KParams& p, char* buf, size_t n) {
  if (p.cols) snprintf(buf, n, "tg::step_kernel_bits<%d, %d, %d, %d>", W, H, RS, bits_feat<W, H, RS>(p));
  else snprintf(buf, n, "tg::step_kernel%s<%d, %d, %d>", g_kernel_lanes == 8 ? "_block" : (g_kernel_lanes ? "_lanes" : ""), W, H, RS);  // (legacy build)
}

template <int W, int H, int RS>
void launch_step(const tg::KParams& p, hipStream_t s) {
  if (p.cols) {  // "bits": the handle carries the column masks / look-ahead permutations
    using BG = tg::BitsGeo<W, H>;
    const int nrb = rng_blocks(p, RS);  // RNG-role workgroups (tg_kernel_bits.h)
    const uint4* const ring4 = reinterpret_cast<const uint4*>(p.ring);
    // (FUNCTIONAL: hbuf is the look-ahead word array fnw [B] u64 — private to each env's logic wave, no parities)
    const size_t hrd = RS == TG_RULESET_NUMPY ? (size_t)p.hbuf_cur * p.ring_stride : 0, hwr = RS == TG_RULESET_NUMPY ? (size_t)(1 - p.hbuf_cur) * p.ring_stride : 0;
    const tg::BitsParams bp{p.cols, p.ahead,
                            ring4 + (size_t)p.ring_cur * p.ring_stride, const_cast<uint4*>(ring4) + (size_t)(1 - p.ring_cur) * p.ring_stride,
                            p.hbuf + hrd, p.hbuf + hwr, p.inl, nrb, p.ring_fill};
    constexpr int M = tg::kFeatMulti, T = tg::kFeatTraj, D = tg::kFeatDict, S = tg::kFeatSample, F = tg::kFeatFinal, R = tg::kFeatRgb;
    const int feat = bits_feat<W, H, RS>(p);
    constexpr int NSEQ = tg::bits_nseq<W, H>(tg::kFeatPacked | tg::kFeatSeq);  // (2 for the sizes that have kFeatSeq instantiations)
    using BGS = tg::BitsGeo<W, H, NSEQ>;
    const bool seq = NSEQ > 1 && (feat & tg::kFeatSeq);
    const int epw_l = seq ? BGS::EPW : BG::EPW;
    const int n_step_wgs = (p.batch + epw_l - 1) / epw_l;
    const dim3 grid(nrb + n_step_wgs);
    const size_t lds = seq ? (size_t)(RS == TG_RULESET_FUNCTIONAL ? BGS::BYTES_FNX0 : BGS::BYTES)
                       : RS == TG_RULESET_FUNCTIONAL ? ((feat & F) ? BG::BYTES_FNX : BG::BYTES_FNX0)
                       : ((feat & D) ? ((feat & tg::kFeatSplit) ? BG::BYTES_DICT_SPLIT : BG::BYTES_DICT)
                                     : ((feat & R) ? ((feat & tg::kFeatMaskWave) ? ((feat & tg::kFeatMany) ? BG::BYTES_RGB_X6 : BG::BYTES_RGB_X) : BG::BYTES_RGB) : ((feat & F) ? BG::BYTES_FIN : BG::BYTES)));
#define TG_LAUNCH_BITS(FT)                                                                                                  \
  hipLaunchKernelGGL((tg::step_kernel_bits<W, H, RS, (FT)>), grid, dim3(64 * tg::bits_wg_waves<W, H>(FT)), lds, s, (p.pboard ? p.pboard : p.board), p.rec, p.cols, bp.ring_rd, bp.hbuf_rd, \
                     p.actions, p.batch, bits_hot_misc(p, RS, n_step_wgs), \
                     p, bp)
    if (feat == F) { TG_LAUNCH_BITS(F); return; }
    if (feat == (M | T | F)) { TG_LAUNCH_BITS(M | T | F); return; }
    if constexpr (tg::bits_packed<W, H>(tg::kFeatPacked)) {
      constexpr int K = tg::kFeatPacked;
      if constexpr (NSEQ > 1) {
        constexpr int Q = tg::kFeatSeq;
        if (feat == (K | Q)) { TG_LAUNCH_BITS(K | Q); return; }
        if constexpr (RS == TG_RULESET_NUMPY) {
          if (feat == (K | T | S | Q)) { TG_LAUNCH_BITS(K | T | S | Q); return; }
        }
      }
      if (feat == K) { TG_LAUNCH_BITS(K); return; }
      if (feat == (K | M)) { TG_LAUNCH_BITS(K | M); return; }
      if (feat == (K | M | T)) { TG_LAUNCH_BITS(K | M | T); return; }
      if constexpr (RS == TG_RULESET_NUMPY) {
        if (feat == (K | T | S)) { TG_LAUNCH_BITS(K | T | S); return; }
        if (feat == (K | M | T | S)) { TG_LAUNCH_BITS(K | M | T | S); return; }
      }
    }
    if constexpr (RS == TG_RULESET_NUMPY) {
      if (feat == R) { TG_LAUNCH_BITS(R); return; }
      if constexpr (tg::bits_rgb_helpers<W, H>(R | tg::kFeatMaskWave)) {
        if (feat == (R | tg::kFeatMaskWave)) { TG_LAUNCH_BITS(R | tg::kFeatMaskWave); return; }
        if (feat == (R | tg::kFeatMaskWave | tg::kFeatMany)) { TG_LAUNCH_BITS(R | tg::kFeatMaskWave | tg::kFeatMany); return; }
        if (feat == (M | T | S | R | tg::kFeatMaskWave)) { TG_LAUNCH_BITS(M | T | S | R | tg::kFeatMaskWave); return; }
      }
      if (feat == (M | T | S | R)) { TG_LAUNCH_BITS(M | T | S | R); return; }
      if (feat == (M | T | S)) { TG_LAUNCH_BITS(M | T | S); return; }
      if (feat == (T | S)) { TG_LAUNCH_BITS(T | S); return; }
      if (feat == (M | T | D)) { TG_LAUNCH_BITS(M | T | D); return; }
      if (feat == D) { TG_LAUNCH_BITS(D); return; }
      if constexpr (BG::DICT_OK) {
        constexpr int X = tg::kFeatMaskWave;
        if (feat == (D | X)) { TG_LAUNCH_BITS(D | X); return; }
        if (feat == (M | T | D | X)) { TG_LAUNCH_BITS(M | T | D | X); return; }
      }
    }
    if constexpr (RS == TG_RULESET_FUNCTIONAL && BG::SPLIT_OK) {
      if (feat == tg::kFeatSplit) { TG_LAUNCH_BITS(tg::kFeatSplit); return; }
      if (feat == (M | tg::kFeatSplit)) { TG_LAUNCH_BITS(M | tg::kFeatSplit); return; }
      if (feat == (M | T | tg::kFeatSplit)) { TG_LAUNCH_BITS(M | T | tg::kFeatSplit); return; }
    }
    if constexpr (RS == TG_RULESET_FUNCTIONAL) {
      if (feat == tg::kFeatWide) { TG_LAUNCH_BITS(tg::kFeatWide); return; }
      if (feat == (M | tg::kFeatWide)) { TG_LAUNCH_BITS(M | tg::kFeatWide); return; }
    }
    if constexpr (BG::SPLIT_OK && RS == TG_RULESET_NUMPY) {
      constexpr int P = tg::kFeatSplit;
      if (feat == (P | D)) { TG_LAUNCH_BITS(P | D); return; }
      if (feat == (P | M | T | D)) { TG_LAUNCH_BITS(P | M | T | D); return; }
      if (feat == (P | M | T)) { TG_LAUNCH_BITS(P | M | T); return; }
      if (feat == (P | M)) { TG_LAUNCH_BITS(P | M); return; }
      if (feat == P) { TG_LAUNCH_BITS(P); return; }
    }
    if (feat == (M | T)) TG_LAUNCH_BITS(M | T);
    else if (feat == M) TG_LAUNCH_BITS(M);
    else TG_LAUNCH_BITS(0);
#undef TG_LAUNCH_BITS
    return;
  }
#ifdef TG_LEGACY_KERNELS
  if (g_kernel_lanes == 8 && !p.final_obs) {  // "block": two-phase kernel with workgroup-level slow-path compaction
    if (g_block_waves == 8) {
      using BG = tg::BlockGeo<W, H, 8>;
      hipLaunchKernelGGL((tg::step_kernel_block<W, H, RS, 8>), dim3((p.batch + BG::EPB - 1) / BG::EPB), dim3(BG::NT), BG::BYTES, s, p);
    } else {
      using BG = tg::BlockGeo<W, H, 4>;
      hipLaunchKernelGGL((tg::step_kernel_block<W, H, RS, 4>), dim3((p.batch + BG::EPB - 1) / BG::EPB), dim3(BG::NT), BG::BYTES, s, p);
    }
    return;
  }
  if (g_kernel_lanes && !p.final_obs) {
    // two groups per wave (software-pipelined inputs) when the batch is large enough to keep >= 2 waves per SIMD
    auto launch = [&](auto kern1, auto kern2, int epw) {
      const int groups = (p.batch + epw - 1) / epw;
      const size_t lds = (((size_t)epw * W * H + 15) / 16) * 16 + 64 + 4 * W + 16 + g_lds_pad;  // + cell table + drop-probe overrun
      if (g_groups_per_wave > 1 && groups >= 4096)
        hipLaunchKernelGGL(kern2, dim3((groups + 1) / 2), dim3(64), lds, s, p);
      else
        hipLaunchKernelGGL(kern1, dim3(groups), dim3(64), lds, s, p);
    };
    if (g_kernel_lanes == 2)
      launch(tg::step_kernel_lanes<W, H, RS, 2, false>, tg::step_kernel_lanes<W, H, RS, 2, true>, 32);
    else if (g_kernel_lanes == 4)
      launch(tg::step_kernel_lanes<W, H, RS, 4, false>, tg::step_kernel_lanes<W, H, RS, 4, true>, 16);
    else
      launch(tg::step_kernel_lanes<W, H, RS, 1, false>, tg::step_kernel_lanes<W, H, RS, 1, false>, 64);
    return;
  }
  const int grid = grid_for(p.batch);
  hipLaunchKernelGGL((tg::step_kernel<W, H, RS>), dim3(grid), dim3(tg::kBlock), 0, s, p);
#else
  (void)s;  // (unreachable: tg_create refuses handles without the column-mask state in the product build)
#endif
}
template <int W, int H, int RS>
void launch_reset(const tg::ResetParams& p, hipStream_t s) {
  hipLaunchKernelGGL((tg::reset_kernel<W, H, RS>), dim3(grid_for(p.batch)), dim3(tg::kBlock), 0, s, p);
}
template <int W, int H>
void launch_features(const tg::FeatureParams& p, hipStream_t s) {
  hipLaunchKernelGGL((tg::features_kernel<W, H>), dim3(grid_for(p.batch)), dim3(tg::kBlock), 0, s, p);
}
template <int W, int H>
void launch_prepare(const tg::PrepareParams& p, hipStream_t s) {
  hipLaunchKernelGGL((tg::bits_prepare_kernel<W, H>), dim3((p.batch + 255) / 256), dim3(256), 0, s, p);
}
template <int W, int H>
void launch_placements(const tg::PlacementParams& p, hipStream_t s) {
  hipLaunchKernelGGL((tg::placements_kernel<W, H>), dim3(grid_for(p.batch)), dim3(tg::kBlock), (tg::placements_lds_bytes<W, H>()), s, p);
}
template <int W, int H>
void launch_observe(const tg::ObserveParams& p, hipStream_t s) {
  const int per_block = tg::kWavesPerBlock * tg::kObserveEnvsPerWave;
  hipLaunchKernelGGL((tg::observe_kernel<W, H>), dim3((p.batch + per_block - 1) / per_block), dim3(tg::kBlock), 0, s, p);
}

bool find_launchers(int W, int H, int ruleset, Launchers* out) {
#define X(w, h)                                                                                     \
  if (W == w && H == h) {                                                                           \
    if (ruleset == TG_RULESET_NUMPY)                                                                \
      *out = {launch_features<w, h>, launch_step<w, h, TG_RULESET_NUMPY>, step_name<w, h, TG_RULESET_NUMPY>, launch_reset<w, h, TG_RULESET_NUMPY>, launch_observe<w, h>, \
              launch_prepare<w, h>, launch_placements<w, h>, tg::bits_cols_bytes<w, h>(), tg::BitsGeo<w, h>::DICT_OK, tg::BitsGeo<w, h>::RGB_OK};          \
    else                                                                                            \
      *out = {launch_features<w, h>, launch_step<w, h, TG_RULESET_FUNCTIONAL>, step_name<w, h, TG_RULESET_FUNCTIONAL>, launch_reset<w, h, TG_RULESET_FUNCTIONAL>,  \
              launch_observe<w, h>, launch_prepare<w, h>, launch_placements<w, h>, tg::bits_cols_bytes<w, h>(), tg::BitsGeo<w, h>::DICT_OK, tg::BitsGeo<w, h>::RGB_OK}; \
    return true;                                                                                    \
  }
  TG_FOR_EACH_SIZE(X)
#undef X
  return false;
}

// ---------------------------------------------------------------------------------------------------------------------
// Board sizes that are not built in: the same kernel templates, instantiated for (W, H) by hiprtc when the first handle of
// that size is created (reference envs/tetris.py:77-91 / functional/core.py:11-25 accept any width / height).  In-process
// compilation — nothing is exec'ed (a process that has touched the GPU must not start another program on this pool) —
// from the headers that ship next to the library (csrc/; TG_CSRC_DIR overrides), with the flags of the Makefile.  The
// code object is cached on disk (TG_CACHE_DIR, default ~/.cache/tetris_gym_hip) under a hash of the headers, so only the
// first process pays the compile (tens of seconds for the ~16 instantiations of a size).
// ---------------------------------------------------------------------------------------------------------------------
struct RtcSize {
  hipModule_t mod = nullptr;
  std::map<int, hipFunction_t> step;  // by FEAT
  hipFunction_t reset = nullptr, observe = nullptr, prepare = nullptr, features = nullptr, placements = nullptr;
  int W = 0, H = 0, RS = 0;
  int geo[14] = {};  // EPW, BYTES, BYTES_DICT, BYTES_RGB, BYTES_FIN, BYTES_FNX, DICT_OK, RGB_OK, cols_bytes, placements lds, NSW0, SPLIT_OK,
                     // kFeatSeq: EPW (0: the size has no such instantiations), BYTES
};
std::mutex g_rtc_mu;
std::map<std::string, RtcSize*> g_rtc;  // "dev:W:H:RS" → module (lives as long as the process)

std::string csrc_dir() {
  if (const char* e = getenv("TG_CSRC_DIR")) return e;
  Dl_info info;
  if (dladdr((const void*)&csrc_dir, &info) && info.dli_fname) {
    std::string lib = info.dli_fname;
    const size_t k = lib.rfind('/');
    return (k == std::string::npos ? std::string(".") : lib.substr(0, k)) + "/csrc";
  }
  return "csrc";
}

bool packed_geometry(int W, int H) {  // (tg::bits_packed for a size known at run time)
  const int S = W * H, epw = tg::bits_epw(W, H);
  return S % 2 == 0 && (epw * S) % 32 == 0;
}
std::vector<int> rtc_feats(int RS, int W = 0, int H = 0) {
  constexpr int M = tg::kFeatMulti, T = tg::kFeatTraj, D = tg::kFeatDict, S = tg::kFeatSample, F = tg::kFeatFinal, R = tg::kFeatRgb;
  constexpr int P = tg::kFeatSplit, Wd = tg::kFeatWide, K = tg::kFeatPacked;
  if (RS == TG_RULESET_NUMPY) {
    std::vector<int> v = {0, M, M | T, F, M | T | F, D, M | T | D, T | S, M | T | S, R, M | T | S | R, P, P | M, P | M | T, P | D, P | M | T | D,
                          D | tg::kFeatMaskWave, M | T | D | tg::kFeatMaskWave, R | tg::kFeatMaskWave, R | tg::kFeatMaskWave | tg::kFeatMany, M | T | S | R | tg::kFeatMaskWave};
    if (W && packed_geometry(W, H))
      for (int f : {K, K | M, K | M | T, K | T | S, K | M | T | S}) v.push_back(f);
    if (W && packed_geometry(W, H) && tg::bits_eps0(W, H) == 16)
      for (int f : {K | tg::kFeatSeq, K | T | S | tg::kFeatSeq}) v.push_back(f);
    return v;
  }
  std::vector<int> v = {0, M, M | T, F, M | T | F, Wd, M | Wd, P, P | M, P | M | T};
  if (W && packed_geometry(W, H))
    for (int f : {K, K | M, K | M | T}) v.push_back(f);
  if (W && packed_geometry(W, H) && tg::bits_eps0(W, H) == 16) v.push_back(K | tg::kFeatSeq);
  return v;
}

uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

// compile (or load from the cache) the kernels of one size; returns an error text or "" on success
std::string rtc_build(int device, int W, int H, int RS, RtcSize** out) {
  char keyb[64];
  snprintf(keyb, sizeof keyb, "%d:%d:%d:%d", device, W, H, RS);
  std::lock_guard<std::mutex> lock(g_rtc_mu);
  auto it = g_rtc.find(keyb);
  if (it != g_rtc.end()) { *out = it->second; return ""; }
  const std::string dir = csrc_dir();
  std::string all;  // the headers, for the cache key
  for (const char* f : {"tg_common.h", "tg_lane.h", "tg_tables.h", "tg_kernel_bits.h", "tg_kernel_features.h", "tg_kernel_placements.h",
                        "../../include/tetris_gym_hip.h"}) {
    std::ifstream in(dir + "/" + f, std::ios::binary);
    if (!in) return "kernel source " + dir + "/" + f + " not found (set TG_CSRC_DIR to tetris_gymnasium_amd/csrc)";
    std::stringstream ss; ss << in.rdbuf(); all += ss.str();
  }
  std::vector<std::string> names;
  char nb[128];
  for (int ft : rtc_feats(RS, W, H)) { snprintf(nb, sizeof nb, "tg::step_kernel_bits<%d, %d, %d, %d>", W, H, RS, ft); names.push_back(nb); }
  snprintf(nb, sizeof nb, "tg::reset_kernel<%d, %d, %d>", W, H, RS); names.push_back(nb);
  snprintf(nb, sizeof nb, "tg::observe_kernel<%d, %d>", W, H); names.push_back(nb);
  snprintf(nb, sizeof nb, "tg::bits_prepare_kernel<%d, %d>", W, H); names.push_back(nb);
  snprintf(nb, sizeof nb, "tg::features_kernel<%d, %d>", W, H); names.push_back(nb);
  snprintf(nb, sizeof nb, "tg::placements_kernel<%d, %d>", W, H); names.push_back(nb);
  char srcb[2048];
  snprintf(srcb, sizeof srcb,
           "#include \"tg_kernel_bits.h\"\n#include \"tg_kernel_features.h\"\n#include \"tg_kernel_placements.h\"\n"
           "typedef tg::BitsGeo<%d, %d> TgBG;\n"
           "typedef tg::BitsGeo<%d, %d, tg::bits_nseq<%d, %d>(tg::kFeatPacked | tg::kFeatSeq)> TgBGS;\n"
           "extern \"C\" __constant__ int tg_rtc_geo[14] = {TgBG::EPW, TgBG::BYTES, TgBG::BYTES_DICT, TgBG::BYTES_RGB, TgBG::BYTES_FIN, "
           "TgBG::BYTES_FNX, TgBG::DICT_OK ? 1 : 0, TgBG::RGB_OK ? 1 : 0, (int)tg::bits_cols_bytes<%d, %d>(), "
           "(int)tg::placements_lds_bytes<%d, %d>(), TgBG::NSW0, TgBG::SPLIT_OK ? 1 : 0, "
           "TgBGS::EPW != TgBG::EPW ? TgBGS::EPW : 0, TgBGS::BYTES};\n", W, H, W, H, W, H, W, H, W, H);
  const std::string src = srcb;
  const std::vector<std::string> optv = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + dir, "-mllvm", "-amdgpu-kernarg-preload-count=16",
                                         "-mllvm", "-amdgpu-sched-strategy=max-ilp"};
  std::string keytext = all + src;
  for (auto& o : optv) keytext += o;
  for (auto& n : names) keytext += n;
  {  // what a cached code object depends on besides the sources: the compiler, the runtime, the device
    int maj = 0, min = 0, rt = 0;
    (void)hiprtcVersion(&maj, &min);
    (void)hipRuntimeGetVersion(&rt);
    hipDeviceProp_t prop;
    const bool okp = hipGetDeviceProperties(&prop, device) == hipSuccess;
    keytext += "|hiprtc " + std::to_string(maj) + "." + std::to_string(min) + "|runtime " + std::to_string(rt) + "|" +
               (okp ? prop.gcnArchName : "?");
  }
  char hashb[32];
  snprintf(hashb, sizeof hashb, "%016llx", (unsigned long long)fnv1a(keytext));
  // The cache directory must be ours alone: what sits there goes straight to hipModuleLoadData.  TG_CACHE_DIR, else
  // ~/.cache/tetris_gym_hip, else (no HOME) /tmp/tetris_gym_hip-<uid>; created 0700; a directory that is not owned by this
  // user or is writable by others is not used (every process then compiles for itself).
  std::string cdir;
  if (const char* e = getenv("TG_CACHE_DIR")) cdir = e;
  else if (const char* hm = getenv("HOME")) { mkdir((std::string(hm) + "/.cache").c_str(), 0755); cdir = std::string(hm) + "/.cache/tetris_gym_hip"; }
  else cdir = "/tmp/tetris_gym_hip-" + std::to_string((long)geteuid());
  bool cache_ok = false;
  {
    mkdir(cdir.c_str(), 0700);
    struct stat st;
    cache_ok = stat(cdir.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & 022) == 0;
  }
  const std::string cfile = cdir + "/" + std::to_string(W) + "x" + std::to_string(H) + "_" + std::to_string(RS) + "_" + hashb + ".bin";
  // cache file: lowered names (one per line) | blank line | checksum of the code object (16 hex digits) | code object
  auto from_cache = [&](std::vector<std::string>& lowered, std::string& code) -> bool {
    if (!cache_ok) return false;
    std::ifstream in(cfile, std::ios::binary);
    if (!in) return false;
    std::string line, sum;
    while (std::getline(in, line) && !line.empty()) lowered.push_back(line);
    if (!std::getline(in, sum)) return false;
    std::stringstream ss; ss << in.rdbuf(); code = ss.str();
    char want[32];
    snprintf(want, sizeof want, "%016llx", (unsigned long long)fnv1a(code));
    return lowered.size() == names.size() && !code.empty() && sum == want;
  };
  auto compile = [&](std::vector<std::string>& lowered, std::string& code) -> std::string {
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "tg_rtc.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return "hiprtcCreateProgram failed";
    for (auto& n : names) hiprtcAddNameExpression(prog, n.c_str());
    std::vector<const char*> opts;
    for (auto& o : optv) opts.push_back(o.c_str());
    const hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (r != HIPRTC_SUCCESS) {
      size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls);
      std::string log(ls, 0); if (ls) hiprtcGetProgramLog(prog, &log[0]);
      hiprtcDestroyProgram(&prog);
      const size_t e = log.find("error:");
      return std::string("compiling the kernels for this board size failed: ") + (e == std::string::npos ? log.substr(0, 300) : log.substr(e, 300));
    }
    for (auto& n : names) { const char* low = nullptr; hiprtcGetLoweredName(prog, n.c_str(), &low); lowered.push_back(low ? low : ""); }
    size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, &code[0]);
    hiprtcDestroyProgram(&prog);
    if (cache_ok) {
      const std::string tmp = cfile + ".tmp" + std::to_string((long)getpid());
      std::ofstream o(tmp, std::ios::binary);
      if (o) {
        char sum[32];
        snprintf(sum, sizeof sum, "%016llx", (unsigned long long)fnv1a(code));
        for (auto& l : lowered) o << l << "\n";
        o << "\n" << sum << "\n";
        o.write(code.data(), (std::streamsize)code.size());
        o.close();
        rename(tmp.c_str(), cfile.c_str());
      }
    }
    return "";
  };
  // load a code object and look every kernel up; false (nothing kept) when anything is missing
  auto load = [&](const std::vector<std::string>& lowered, const std::string& code, RtcSize* rs) -> bool {
    if (hipModuleLoadData(&rs->mod, code.data()) != hipSuccess) return false;
    size_t k = 0;
    auto fn = [&](hipFunction_t* f) { return hipModuleGetFunction(f, rs->mod, lowered[k++].c_str()) == hipSuccess; };
    bool ok = true;
    for (int ft : rtc_feats(RS, W, H)) { hipFunction_t f = nullptr; ok = fn(&f) && ok; rs->step[ft] = f; }
    ok = fn(&rs->reset) && ok; ok = fn(&rs->observe) && ok; ok = fn(&rs->prepare) && ok; ok = fn(&rs->features) && ok; ok = fn(&rs->placements) && ok;
    hipDeviceptr_t gp = nullptr; size_t gb = 0;
    ok = ok && hipModuleGetGlobal(&gp, &gb, rs->mod, "tg_rtc_geo") == hipSuccess && gb == sizeof(rs->geo) &&
         hipMemcpy(rs->geo, gp, sizeof(rs->geo), hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) { (void)hipModuleUnload(rs->mod); rs->mod = nullptr; rs->step.clear(); }
    return ok;
  };
  RtcSize* rs = new RtcSize();
  rs->W = W; rs->H = H; rs->RS = RS;
  bool loaded = false;
  {
    std::vector<std::string> lowered;
    std::string code;
    if (from_cache(lowered, code)) loaded = load(lowered, code, rs);  // (a stale or damaged file falls through to a compile)
  }
  if (!loaded) {
    std::vector<std::string> lowered;
    std::string code;
    const std::string err = compile(lowered, code);
    if (!err.empty()) { delete rs; return err; }
    if (!load(lowered, code, rs)) { delete rs; return "the compiled module lacks an expected kernel"; }
  }
  g_rtc[keyb] = rs;
  *out = rs;
  return "";
}

template <typename P>
void rtc_launch(hipFunction_t f, dim3 grid, dim3 block, size_t lds, hipStream_t s, const P& p) {
  void* args[] = {const_cast<P*>(&p)};
  (void)hipModuleLaunchKernel(f, grid.x, 1, 1, block.x, 1, 1, (unsigned)lds, s, args, nullptr);
}

Launchers rtc_launchers(const RtcSize* r) {
  Launchers L;
  const int EPW = r->geo[0], W = r->W, H = r->H, RS = r->RS;
  const bool dict_ok = r->geo[6] != 0, rgb_ok = r->geo[7] != 0;
  // (tg::bits_rgb_helper_count: 0 / 2 / 6 helper waves by what fits the 64 KiB beside BYTES_RGB)
  const int rgb_idb4 = 4 * ((((H + 4) * (W + 24)) + 15) / 16 * 16);
  const int rgb_helpers_max = !(rgb_ok && EPW == 64) ? 0 : (r->geo[3] + 6 * rgb_idb4 <= 65536 ? 6 : (r->geo[3] + 2 * rgb_idb4 <= 65536 ? 2 : 0));
  L.cols_bytes = (size_t)r->geo[8];
  L.dict_ok = dict_ok; L.rgb_ok = rgb_ok;
  L.step = [=](const tg::KParams& p, hipStream_t s) {
    constexpr int D = tg::kFeatDict, F = tg::kFeatFinal, R = tg::kFeatRgb;
    const int nrb = rng_blocks(p, RS);
    const uint4* const ring4 = reinterpret_cast<const uint4*>(p.ring);
    const size_t hrd = RS == TG_RULESET_NUMPY ? (size_t)p.hbuf_cur * p.ring_stride : 0, hwr = RS == TG_RULESET_NUMPY ? (size_t)(1 - p.hbuf_cur) * p.ring_stride : 0;
    tg::BitsParams bp{p.cols, p.ahead, ring4 + (size_t)p.ring_cur * p.ring_stride, const_cast<uint4*>(ring4) + (size_t)(1 - p.ring_cur) * p.ring_stride,
                      p.hbuf + hrd, p.hbuf + hwr, p.inl, nrb, p.ring_fill};
    const bool split_ok = r->geo[11] != 0;  // (BitsGeo::SPLIT_OK)
    const int feat = bits_feat_rt(p, RS, dict_ok, rgb_ok, EPW, split_ok, rgb_helpers_max);
    const bool seq = (feat & tg::kFeatSeq) && r->geo[12] > 0;
    const int epw_l = seq ? r->geo[12] : EPW;
    const int n_step_wgs = (p.batch + epw_l - 1) / epw_l;
    const size_t lds = seq ? (size_t)(r->geo[13] + (RS == TG_RULESET_FUNCTIONAL ? 128 * 12 : 0))  // (BitsGeo<.., 2>::BYTES / BYTES_FNX0)
                       : RS == TG_RULESET_FUNCTIONAL ? ((feat & F) ? r->geo[5] : r->geo[5] - 2 * r->geo[0] * 8)  // (BYTES_FNX / BYTES_FNX0)
                       : ((feat & D) ? r->geo[2] + ((feat & tg::kFeatSplit) ? 16 * (W + 8) * (H + 4) : 0)  // (BYTES_DICT_SPLIT: + GD * PC)
                                     : ((feat & R) ? r->geo[3] + ((feat & tg::kFeatMaskWave) ? ((feat & tg::kFeatMany) ? 6 : 2) * 4 * ((((H + 4) * (W + 24)) + 15) / 16 * 16) : 0)  // (BYTES_RGB_X: + 2 GR IDB)
                                                   : ((feat & F) ? r->geo[4] : r->geo[1])));
    auto it = r->step.find(feat);
    if (it == r->step.end()) return;
    uint8_t* a0 = p.pboard ? p.pboard : p.board; uint32_t* a1 = p.rec; void* a2 = p.cols; const uint4* a3 = bp.ring_rd; const uint32_t* a4 = bp.hbuf_rd;
    const void* a5 = p.actions; int a6 = p.batch;
    uint32_t a7 = bits_hot_misc(p, RS, n_step_wgs);
    tg::KParams kp = p;
    void* args[] = {&a0, &a1, &a2, &a3, &a4, &a5, &a6, &a7, &kp, &bp};
    // (threads per workgroup: tg::bits_stream_waves of this size and instantiation)
    (void)hipModuleLaunchKernel(it->second, (unsigned)(nrb + n_step_wgs), 1, 1, 64 * (1 + r->geo[10] * ((feat & tg::kFeatSplit) ? 2 : 1) + ((feat & tg::kFeatMaskWave) ? ((feat & R) ? ((feat & tg::kFeatMany) ? 6 : 2) : 1) : 0)), 1, 1, (unsigned)lds, s, args, nullptr);
  };
  L.step_name = [=](const tg::KParams& p, char* buf, size_t n) {
    snprintf(buf, n, "tg::step_kernel_bits<%d, %d, %d, %d>", W, H, RS, bits_feat_rt(p, RS, dict_ok, rgb_ok, EPW, r->geo[11] != 0, rgb_helpers_max));
  };
  L.reset = [=](const tg::ResetParams& p, hipStream_t s) { rtc_launch(r->reset, dim3(grid_for(p.batch)), dim3(tg::kBlock), 0, s, p); };
  L.observe = [=](const tg::ObserveParams& p, hipStream_t s) {
    const int per_block = tg::kWavesPerBlock * tg::kObserveEnvsPerWave;
    rtc_launch(r->observe, dim3((p.batch + per_block - 1) / per_block), dim3(tg::kBlock), 0, s, p);
  };
  L.prepare = [=](const tg::PrepareParams& p, hipStream_t s) { rtc_launch(r->prepare, dim3((p.batch + 255) / 256), dim3(256), 0, s, p); };
  L.features = [=](const tg::FeatureParams& p, hipStream_t s) { rtc_launch(r->features, dim3(grid_for(p.batch)), dim3(tg::kBlock), 0, s, p); };
  L.placements = [=](const tg::PlacementParams& p, hipStream_t s) { rtc_launch(r->placements, dim3(grid_for(p.batch)), dim3(tg::kBlock), (size_t)r->geo[9], s, p); };
  return L;
}

// TG_BUF_HOST_PACK: rewards f64 | episode returns f64 | lines i32 | episode lengths i32 | terminated u8, each 16-byte aligned
size_t host_pack_layout(size_t B, size_t* off) {
  auto up = [](size_t v) { return (v + 15) / 16 * 16; };
  size_t o[6];
  o[0] = 0;
  o[1] = up(o[0] + 8 * B);
  o[2] = up(o[1] + 8 * B);
  o[3] = up(o[2] + 4 * B);
  o[4] = up(o[3] + 4 * B);
  o[5] = up(o[4] + B);
  if (off) for (int i = 0; i < 6; ++i) off[i] = o[i];
  return o[5];
}

struct PackParams {
  const uint2* out;             // tg_step_out [B]
  const tg_episode_rec* ep;     // or null
  uint8_t* pack;
  size_t off[5];
  int batch;
  int custom;        // TG_FLAG_CUSTOM_REWARDS: rewards recomputed in double precision from tg_step_out.flags / lines
  int width;
  double r_alife, r_game_over, r_invalid;
};

__global__ __launch_bounds__(256) void host_pack_kernel(const PackParams p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.batch) return;
  const uint2 o = p.out[i];
  const float r = __uint_as_float(o.x);
  const uint32_t term = o.y & 0xFFu, lines = (o.y >> 8) & 0xFFu;
  double er = 0.0;
  int el = 0;
  if (p.ep && term) {
    const tg_episode_rec e = p.ep[i];
    er = (double)e.last_ret;
    el = (int)e.last_len;
  }
  // (the only negative reward is rewards.invalid_action of the grouped action set, mappings/rewards.py: the host hands out
  // the Python float -0.1, not float32(-0.1))
  double rd = r < 0.0f ? -0.1 : (double)r;
  if (p.custom) {  // what Python computes: int score + rewards.alife in double precision (envs/tetris.py:461-474, grouped.py:238)
    const uint32_t fl = (o.y >> 16) & 0xFFu;
    rd = (fl & TG_OUT_ILLEGAL) ? p.r_invalid
         : ((fl & TG_OUT_COMMITTED) ? (term ? p.r_game_over : (double)(lines * lines * (uint32_t)p.width) + p.r_alife) : 0.0);
  }
  reinterpret_cast<double*>(p.pack + p.off[0])[i] = rd;
  reinterpret_cast<double*>(p.pack + p.off[1])[i] = er;
  reinterpret_cast<int32_t*>(p.pack + p.off[2])[i] = (int32_t)lines;
  reinterpret_cast<int32_t*>(p.pack + p.off[3])[i] = el;
  (p.pack + p.off[4])[i] = (uint8_t)(term ? 1 : 0);
}

// tg_set_state_dev cannot range-check device-resident records on the host (tg_set_state does).  This pass does it on the device, right
// behind the copies: what tg_set_state would refuse is CLAMPED to something harmless (an out-of-range piece / position would index the
// kernels' tables and LDS slabs out of bounds) and flagged in *err; the next synchronising call on the handle reports it (TG_EINVAL).
struct SanitizeParams {
  uint32_t* rec;         // [B][8]
  uint8_t* board;        // [B][H*W] or null (not uploaded)
  int* err;              // flag word: bit 0 record field out of range, bit 1 board byte out of range
  int batch, W, H, Q, ruleset, packed;
};
__global__ __launch_bounds__(256) void sanitize_state_kernel(const SanitizeParams p) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= p.batch) return;
  uint32_t* r = p.rec + (size_t)env * 8;
  uint32_t d0 = r[0], d1 = r[1], d2 = r[2], d3 = r[3], d4 = r[4], d5 = r[5];
  int bad = 0;
  auto fix = [&](uint32_t& w, const int shift, const uint32_t limit, const uint32_t safe) {  // byte field <= limit, else := safe
    if (((w >> shift) & 0xFFu) > limit) { w = (w & ~(0xFFu << shift)) | (safe << shift); bad |= 1; }
  };
  fix(d0, 0, 6, 0);                                   // piece
  fix(d0, 8, 3, 0);                                   // rot
  // x <= Wp - n, n = the native matrix of the piece (I 4, O 2, the others 3; FUNCTIONAL: 4 x 4 boxes throughout): as far as the
  // reference's own slicing reaches (envs/tetris.py:408-427 raises beyond) — every such position lies wholly in the right bedrock
  {
    const uint32_t pc = d0 & 0xFFu;
    fix(d0, 16, (uint32_t)p.W + 4u + (p.ruleset == TG_RULESET_NUMPY ? (pc == 1u ? 2u : (pc == 0u ? 0u : 1u)) : 0u), 7);  // x
  }
  fix(d0, 24, (uint32_t)p.H + 3u, 0);                 // y
  if (((d1 >> 16) & 0xFFu) != TG_NONE) { fix(d1, 16, 6, TG_NONE); fix(d1, 24, 3, 0); }  // holder piece / rot
  fix(d2, 8, 7, 7);                                   // bag index
  fix(d2, 16, 6, 0); fix(d2, 24, 6, 0);               // bag[0..1]
  fix(d3, 0, 6, 0); fix(d3, 8, 6, 0); fix(d3, 16, 6, 0); fix(d3, 24, 6, 0);  // bag[2..5]
  fix(d4, 0, 6, 0);                                   // bag[6]
  if (p.ruleset == TG_RULESET_NUMPY) {                // visible queue (first Q entries)
    if (p.Q > 0) fix(d4, 8, 6, 0);
    if (p.Q > 1) fix(d4, 16, 6, 0);
    if (p.Q > 2) fix(d4, 24, 6, 0);
    if (p.Q > 3) fix(d5, 0, 6, 0);
    if (p.Q > 4) fix(d5, 8, 6, 0);
    if (p.Q > 5) fix(d5, 16, 6, 0);
    if (p.Q > 6) fix(d5, 24, 6, 0);
  }
  if (p.packed) { fix(d1, 0, 15, 2); if (((d1 >> 16) & 0xFFu) != TG_NONE) fix(d2, 0, 15, 2); }  // val / holder val as nibbles
  if (bad) { r[0] = d0; r[1] = d1; r[2] = d2; r[3] = d3; r[4] = d4; r[5] = d5; }
  if (p.board) {
    const uint32_t lim = p.packed ? 15u : (p.ruleset == TG_RULESET_FUNCTIONAL ? 0x7Fu : 0xFFu);
    if (lim < 0xFFu) {
      uint8_t* b = p.board + (size_t)env * p.W * p.H;
      for (int i = 0; i < p.W * p.H; ++i)
        if (b[i] > lim) { b[i] = 2; bad |= 2; }
    }
  }
  if (bad) atomicOr(p.err, bad);
}

// FUNCTIONAL handles, after a state upload: does any env's piece lie ON TOP of filled cells (and would be drawn: not game over)?  Only an
// upload makes such a state — every move into a collision is refused (envs/tetris_fn.py:161-273), and it lasts for as long as nothing
// moves the piece.  The step kernel draws the piece for the observation by plain byte writes and un-draws it with zeros, which is the
// reference's `(board > 0) + (-1) * piece` (envs/tetris_fn.py:148-155) only on free cells; doing it cell by cell in the kernel costs every
// launch 2-8 % (profiles/r05/ab_fn_exact_marker.log).  So the rare state is detected here and handled by the host: while a handle may hold
// one (tg_handle::fn_overlap), a transition is one launch of one transition (nothing is un-drawn) followed by the observe kernel, whose
// draw is exact (tg_common.h emit_obs).  Raises word 4 of tg_handle::d_err.
struct FnOverlapParams {
  const uint8_t* board;
  const uint32_t* rec;
  uint32_t* word;
  int batch, W, H;
  uint64_t shapes[7];  // KParams::shapes
};
__global__ __launch_bounds__(256) void fn_overlap_kernel(const FnOverlapParams q) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= q.batch) return;
  const uint32_t* rec = q.rec;
  const int W = q.W, H = q.H;
  const uint64_t* shapes = q.shapes;
  const uint8_t* board = q.board;
  uint32_t* word = q.word;
  const uint32_t d0 = rec[(size_t)env * 8], d1 = rec[(size_t)env * 8 + 1];
  const int piece = d0 & 0xFF, rot = (d0 >> 8) & 3, x = (d0 >> 16) & 0xFF, y = (d0 >> 24) & 0xFF;
  if (piece > 6 || ((d1 >> 8) & TG_F_GAME_OVER)) return;
  const uint32_t mask = (uint32_t)(shapes[piece] >> (16 * rot)) & 0xFFFFu;
  const uint8_t* b = board + (size_t)env * W * H;
  bool over = false;
  for (int c = 0; c < 16; ++c) {
    const int row = y + (c >> 2), col = x + (c & 3) - 4;
    if (((mask >> c) & 1u) && (unsigned)col < (unsigned)W && (unsigned)row < (unsigned)H) over = over || b[row * W + col] != 0;
  }
  if (over) atomicOr(word, 1u);
}

std::string built_sizes() {
  std::string s;
#define X(w, h) s += std::string(s.empty() ? "" : ", ") + #w "x" #h;
  TG_FOR_EACH_SIZE(X)
#undef X
  return s;
}

}  // namespace

struct tg_handle {
  tg_config cfg;
  int B, W, H, Q;
  size_t cells;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  Launchers L{};
  void* arena = nullptr;  // the one device allocation behind every buffer below
  size_t arena_size = 0, arena_off = 0;
  void* buf[TG_BUF_COUNT] = {};
  size_t bytes[TG_BUF_COUNT] = {};
  tg_rng_rec* d_new_rng = nullptr;
  uint8_t* d_reseed = nullptr;
  uint8_t* d_mask = nullptr;
  void* dbg = nullptr;  // phase-timing buffer (TG_PHASE_TIMING builds)
  void* d_cols = nullptr;   // bits kernel: column masks [B][W] (device-private acceleration state)
  uint4* d_ahead = nullptr; // bits kernel: generator states after the ring's shuffles [B][2 slots][2]
  uint64_t* d_ring = nullptr;  // bits kernel: permutation rings, double-buffered by launch parity [2][B]
  uint32_t* d_hbuf = nullptr;  // bits kernel: consumed counters + state location (+ FUNCTIONAL: bag index, chain epoch) [2][B]
  uint4* d_inl = nullptr;      // bits kernel: generator states after in-line shuffles [2 buffers][B][2]
  uint8_t* d_pboard = nullptr; // bits kernel, packed handles: the board as nibbles [B][H*W/2] (tg_kernel_bits.h kFeatPacked); the step
                               // kernel keeps THIS current, TG_BUF_BOARD is brought up to date on demand (ensure_board)
  bool board_stale = false;    // TG_BUF_BOARD is behind d_pboard
  uint32_t* h_nib_over = nullptr;  // packed FUNCTIONAL handles: page-locked flag word the step kernel raises (KParams::nib_over)
  uint32_t* h_fn_over = nullptr;   // FUNCTIONAL handles with an observation: page-locked copy of fn_overlap_kernel's word + the event behind
  hipEvent_t fn_ev = nullptr;      // that copy — the answer is picked up without waiting for the stream (resolve_fn_overlap)
  int* d_err = nullptr;        // flag word of sanitize_state_kernel (tg_set_state_dev); word 4: fn_overlap_kernel
  bool fn_overlap = false;         // FUNCTIONAL: the uploaded state holds a piece on top of filled cells (fn_overlap_kernel's word, read back)
  bool fn_overlap_pending = false; // ... a tg_set_state_dev has been enqueued whose word nobody has read yet (treated as "may hold one")
  bool dev_state_pending = false;  // a tg_set_state_dev has been enqueued whose flag word nobody has looked at yet
  int ring_cur = 0;            // ring parity the next step launch reads
  int hbuf_cur = 0;            // consumed-counter parity the next step launch reads
  int rng_every = 3;           // TG_RNG_EVERY: one-transition launches carry the RNG-role workgroups every n-th time (the ring
                               // looks 3 entries ahead and the consumer shuffles in line when it runs dry): 7.5 -> 7.3 us per launch
  unsigned launch_no = 0;
  int fuse_max = tg::kMaxFuse; // TG_FUSE: transitions per launch in tg_step_k (1..16)
  int ring_fill = tg::kRing - 1;  // TG_RING_FILL (tests): fewer look-ahead permutations → the consumer's in-line shuffle runs
  int fn_lazy = 0;             // TG_FN_LAZY (tests): KParams.fn_lazy
  uint64_t act_lut = tg::kActLutNumpy;           // the handle's ActionsMapping as the kernel's per-action table
  double rewards[4] = {1.0, 1.0, 0.0, -0.1};     // ... its RewardsMapping: alife, clear_line, game_over, invalid_action
  bool dict_in_kernel = false; // TG_OBS_DICT packed by the bits step kernel itself (no observe launch after a transition)
  // tg_config.streams = n (default 1) spreads the step launches of one transition over n streams (contiguous env shards).
  // Kept as a tested knob, off by default — measured on MI355X: every launch costs ≈2.8-3.4 µs of
  // command-processor time whichever stream it is on, so n shards cost n launch floors per transition (8.3 µs →
  // 9.3 µs at n = 2, 13.8 µs at n = 4, also when the launches are replayed from a HIP graph).
  // Every other operation joins the shards into `stream` first.
  int n_shards = 1;
  std::vector<hipStream_t> sh_stream;
  std::vector<hipEvent_t> sh_event;
  std::vector<int> sh_off, sh_cnt;
  hipEvent_t fork_ev = nullptr;
  bool forked = false;
  std::vector<uint8_t> rng_valid;
  std::vector<tg_rng_rec> h_new_rng;
  std::vector<uint8_t> h_reseed;
  std::string err;
};

namespace {

int fail(tg_handle* h, int code, const char* fmt, ...) {
  char msg[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, sizeof(msg), fmt, ap);
  va_end(ap);
  if (h) h->err = msg;
  else g_create_err = msg;
  return code;
}

#define HIPCHK(h, call)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess) return fail((h), TG_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

size_t obs_bytes_per_env(const tg_config* c, int obs_mode) {
  const size_t cells = (size_t)c->width * c->height;
  const size_t pc = (size_t)(c->width + 8) * (c->height + 4);
  switch (obs_mode) {
    case TG_OBS_BOARD:
    case TG_OBS_FN: return cells;
    case TG_OBS_DICT: return 2 * pc + 16 + 16 * (size_t)c->queue_size;
    case TG_OBS_RGB: return 3 * (size_t)(c->height + 4) * (size_t)(c->width + 8 + 4 * (c->queue_size > 1 ? c->queue_size : 1));
    default: return 0;
  }
}
size_t obs_bytes_per_env(const tg_config* c) { return obs_bytes_per_env(c, c->obs_mode); }

// numpy SeedSequence(seed).generate_state(4, uint64) → PCG64 seeding (numpy/random/bit_generator.pyx,
// _pcg64.pyx, src/pcg64/pcg64.h pcg_setseq_128_srandom_r).  Host-only set-up code, pinned by tests.
void seed_pcg64_host(uint64_t seed, tg_rng_rec* out) {
  typedef unsigned __int128 u128;
  const uint32_t INIT_A = 0x43b0d7e5u, MULT_A = 0x931e8875u, INIT_B = 0x8b51f9ddu, MULT_B = 0x58f38dedu;
  const uint32_t MIX_L = 0xca01f9ddu, MIX_R = 0x4973f715u;
  uint32_t ent[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const int n_ent = ent[1] ? 2 : 1;
  uint32_t hc = INIT_A;
  auto hashmix = [&](uint32_t v) {
    v ^= hc;
    hc *= MULT_A;
    v *= hc;
    v ^= v >> 16;
    return v;
  };
  auto mix = [&](uint32_t x, uint32_t y) {
    uint32_t r = MIX_L * x - MIX_R * y;
    r ^= r >> 16;
    return r;
  };
  uint32_t pool[4];
  for (int i = 0; i < 4; ++i) pool[i] = hashmix(i < n_ent ? ent[i] : 0u);
  for (int s = 0; s < 4; ++s)
    for (int d = 0; d < 4; ++d)
      if (s != d) pool[d] = mix(pool[d], hashmix(pool[s]));
  uint32_t w[8], hb = INIT_B;
  for (int i = 0; i < 8; ++i) {
    uint32_t v = pool[i & 3] ^ hb;
    hb *= MULT_B;
    v *= hb;
    v ^= v >> 16;
    w[i] = v;
  }
  uint64_t q[4];
  for (int i = 0; i < 4; ++i) q[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
  const u128 mult = ((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
  const u128 initstate = ((u128)q[0] << 64) | q[1];
  const u128 inc = ((((u128)q[2] << 64) | q[3]) << 1) | 1;
  u128 st = 0;
  st = st * mult + inc;
  st += initstate;
  st = st * mult + inc;
  std::memset(out, 0, sizeof(*out));
  out->state_lo = (uint64_t)st;
  out->state_hi = (uint64_t)(st >> 64);
  out->inc_lo = (uint64_t)inc;
  out->inc_hi = (uint64_t)(inc >> 64);
}

// Device buffers carry kTailPad bytes of zeroed padding: the step kernel's dword-per-lane slab loads are
// unmasked (lanes past the end of the last env's slab read padding instead of faulting).
constexpr size_t kTailPad = 4096;

// All device state of a handle comes from ONE allocation (`arena`), carved in 4 KiB-aligned pieces in a fixed order:
// where the buffers sit relative to each other is then the same for every handle (separate hipMalloc calls gave
// launch times that differed by up to 0.5 us between two handles of the same process: profiles/r02 placement runs).
hipError_t arena_take(tg_handle* h, void** out, size_t n) {
  const size_t off = ((h->arena_off + 4095) / 4096) * 4096;
  if (off + n > h->arena_size) return hipErrorOutOfMemory;
  *out = (uint8_t*)h->arena + off;
  h->arena_off = off + n;
  return hipSuccess;
}

int alloc_buf(tg_handle* h, int which, size_t n) {
  h->bytes[which] = n;
  if (n == 0) return TG_OK;
  HIPCHK(h, arena_take(h, &h->buf[which], n + kTailPad));
  return TG_OK;
}

// The per-action table of the step kernel for a handle's ActionsMapping (tg_config.action_ids: field values in declaration
// order move_left, move_right, move_down, rotate_clockwise, rotate_counterclockwise, hard_drop, swap, no_op).  Entry a =
// what Tetris.step does with action id a: the FIRST field of its elif chain (envs/tetris.py:222-256: left, right, down, cw,
// ccw, swap, hard_drop, no_op) whose value equals a; none → only gravity, like no_op.  A hard_drop that shares its value with an
// EARLIER field of the chain (round 6): that field's branch runs and the gravity block is skipped, because its guard compares the
// action with `self.actions.hard_drop` again (:259) — the entry of that operation with the hard-drop bit set beside its check / swap
// bit, which the logic wave reads as "this operation, then no gravity" (tg_kernel_bits.h).
void act_lut_from_ids(const int32_t ids[8], uint64_t* lut) {
  static const int chain[8] = {0, 1, 2, 3, 4, 6, 5, 7};  // declaration index of the chain's k-th comparison
  uint64_t v = 0;
  for (int a = 0; a < 8; ++a) {
    int op = 7;  // no_op
    for (int k = 0; k < 8; ++k)
      if (ids[chain[k]] == a) { op = chain[k]; break; }
    uint64_t e = (tg::kActLutNumpy >> (8 * op)) & 0xFFull;
    if (ids[5] == a && op != 5) e |= 0x40ull;  // (op is one of left / right / down / cw / ccw / swap: every one of them has bit 5 or 7)
    v |= e << (8 * a);
  }
  *lut = v;
}

tg::KParams step_params(tg_handle* h) {
  tg::KParams p{};
  p.act_lut = h->act_lut;
  p.r_alife = (float)h->rewards[0];
  p.r_game_over = (float)h->rewards[2];
  p.r_invalid = (float)h->rewards[3];
  p.board = (uint8_t*)h->buf[TG_BUF_BOARD];
  p.rec = (uint32_t*)h->buf[TG_BUF_REC];
  p.rng = (tg_rng_rec*)h->buf[TG_BUF_RNG];
  p.obs = (uint8_t*)h->buf[TG_BUF_OBS];
  p.final_obs = (uint8_t*)h->buf[TG_BUF_FINAL_OBS];
  p.out = (uint2*)h->buf[TG_BUF_STEP_OUT];
  p.episode = (tg_episode_rec*)h->buf[TG_BUF_EPISODE];
  {
    static constexpr tg::ShapeTable t = tg::make_shape_table();
    for (int i = 0; i < 7; ++i)
      p.shapes[i] = (uint64_t)t.mask[i][0] | ((uint64_t)t.mask[i][1] << 16) | ((uint64_t)t.mask[i][2] << 32) |
                    ((uint64_t)t.mask[i][3] << 48);
  }
  p.batch = h->B;
  p.gravity = h->cfg.gravity;
  p.autoreset = h->cfg.autoreset;
  p.obs_mode = h->cfg.obs_mode;
  p.queue_size = h->Q;
  p.action_set = h->cfg.action_set;
  p.grouped_term = h->cfg.grouped_terminate_on_illegal;
  p.stagger = g_stagger;
  p.dbg = (unsigned long long*)h->dbg;
  p.cols = h->d_cols;
  p.ahead = h->d_ahead;
  p.ring = h->d_ring;
  p.hbuf = h->d_hbuf;
  p.inl = h->d_inl;
  p.pboard = h->d_pboard;
  p.ring_cur = h->ring_cur;
  p.hbuf_cur = h->hbuf_cur;
  p.rng_role = 1;
  p.ring_stride = (uint32_t)h->B;
  p.cols_bytes = (uint32_t)h->L.cols_bytes;
  p.nsteps = 1;
  p.act_stride = (uint32_t)h->B;
  p.dict_in_kernel = h->dict_in_kernel ? 1 : 0;
  p.ring_fill = h->ring_fill;
  p.fn_lazy = h->fn_lazy;
  p.nib_over = h->h_nib_over;
  return p;
}

int join_shards(tg_handle* h);
int ensure_board(tg_handle* h);
int launch_observe_dict(tg_handle* h) {
  { const int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }  // (reads what the shard streams' step launches wrote)
  // (packed handles: the step kernel maintains the nibbles only — the observe kernel reads the uint8 board.  A no-op unless the handle is
  // packed and stale; without it the exact path of a packed FUNCTIONAL handle re-packed the boards of the last upload: ADVICE r05)
  { const int rcb = ensure_board(h); if (rcb != TG_OK) return rcb; }
  tg::ObserveParams o{};
  o.board = (const uint8_t*)h->buf[TG_BUF_BOARD];
  o.rec = (const uint32_t*)h->buf[TG_BUF_REC];
  o.obs = (uint8_t*)h->buf[TG_BUF_OBS];
  o.batch = h->B;
  o.obs_mode = h->cfg.obs_mode;
  o.queue_size = h->Q;
  o.ruleset = h->cfg.ruleset;
  h->L.observe(o, h->stream);
  HIPCHK(h, hipGetLastError());
  return TG_OK;
}

// FUNCTIONAL handles, behind a state upload (the whole batch is new): word 4 of d_err := does a piece now lie on top of filled cells?
int launch_fn_overlap(tg_handle* h) {
  if (h->cfg.ruleset != TG_RULESET_FUNCTIONAL || !h->d_cols) return TG_OK;
  HIPCHK(h, hipMemsetAsync((uint32_t*)h->d_err + 4, 0, sizeof(uint32_t), h->stream));
  const tg::KParams kp = step_params(h);
  FnOverlapParams q{};
  q.board = (const uint8_t*)h->buf[TG_BUF_BOARD];
  q.rec = (const uint32_t*)h->buf[TG_BUF_REC];
  q.word = (uint32_t*)h->d_err + 4;
  q.batch = h->B; q.W = h->W; q.H = h->H;
  for (int i = 0; i < 7; ++i) q.shapes[i] = kp.shapes[i];
  hipLaunchKernelGGL(fn_overlap_kernel, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, q);
  HIPCHK(h, hipGetLastError());
  if (h->h_fn_over) {  // the answer travels to a page-locked word; the event says when it is there
    HIPCHK(h, hipMemcpyAsync(h->h_fn_over, (uint32_t*)h->d_err + 4, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipEventRecord(h->fn_ev, h->stream));
  }
  h->fn_overlap_pending = true;
  return TG_OK;
}
// A pending answer is taken as soon as its copy has completed — no wait: a device-resident loop that restores states
// (tg_set_state_dev) and never synchronises used to stay on the exact path for good.
inline void resolve_fn_overlap(tg_handle* h) {
  if (!h->fn_overlap_pending || !h->fn_ev) return;
  if (hipEventQuery(h->fn_ev) == hipSuccess) {
    h->fn_overlap = *(volatile uint32_t*)h->h_fn_over != 0;
    h->fn_overlap_pending = false;
  } else {
    (void)hipGetLastError();  // (hipErrorNotReady is not an error of the call that asked)
  }
}
// ... while this holds, transitions take the exact path (one transition per launch + the observe kernel).  The flag is re-evaluated
// behind every transition of that path (step_impl / rollout_impl), so it falls by itself once no piece lies on filled cells any more
// (the pieces moved or locked) — it used to stay until the next full reset or upload.
inline bool fn_exact_path(tg_handle* h) {
  if (h->cfg.ruleset != TG_RULESET_FUNCTIONAL || h->cfg.obs_mode == TG_OBS_NONE) return false;
  resolve_fn_overlap(h);
  return h->fn_overlap || h->fn_overlap_pending;
}

// ---- shards: fork (step launches) / join (everything else) ----
int join_shards(tg_handle* h) {
  if (!h->forked) return TG_OK;
  for (int i = 0; i < h->n_shards; ++i) {
    HIPCHK(h, hipEventRecord(h->sh_event[i], h->sh_stream[i]));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->sh_event[i], 0));
  }
  h->forked = false;
  return TG_OK;
}

int fork_shards(tg_handle* h) {
  if (h->forked) return TG_OK;
  HIPCHK(h, hipEventRecord(h->fork_ev, h->stream));
  for (int i = 0; i < h->n_shards; ++i) HIPCHK(h, hipStreamWaitEvent(h->sh_stream[i], h->fork_ev, 0));
  h->forked = true;
  return TG_OK;
}

// one transition: a single launch on the handle's stream, or one launch per shard
int launch_step_all(tg_handle* h, const tg::KParams& p) {
  if (h->n_shards <= 1) {
    h->L.step(p, h->stream);
    return TG_OK;
  }
  int rc = fork_shards(h);
  if (rc != TG_OK) return rc;
  const size_t S = h->cells;
  for (int i = 0; i < h->n_shards; ++i) {
    const size_t o = (size_t)h->sh_off[i];
    tg::KParams q = p;
    q.batch = h->sh_cnt[i];
    q.board = p.board + o * S;
    q.rec = p.rec + o * 8;
    q.rng = p.rng + o;
    q.actions = p.act_i8 ? (const void*)((const int8_t*)p.actions + o) : (const void*)((const int32_t*)p.actions + o);
    if (p.obs) q.obs = p.obs + o * obs_bytes_per_env(&h->cfg);
    if (p.obs_traj) q.obs_traj = p.obs_traj + o * obs_bytes_per_env(&h->cfg);
    if (p.out_traj) q.out_traj = p.out_traj + o;
    if (p.act_traj) q.act_traj = p.act_traj + o;
    q.env_index_base = (uint32_t)o;
    q.out = p.out + o;
    if (p.episode) q.episode = p.episode + o;
    if (p.cols) q.cols = (uint8_t*)p.cols + o * p.cols_bytes;
    if (p.ahead) q.ahead = p.ahead + o * 2 * tg::kRing;
    if (p.ring) q.ring = p.ring + o * 2;
    if (p.hbuf) q.hbuf = p.hbuf + (h->cfg.ruleset == TG_RULESET_NUMPY ? o : 2 * o);  // (FUNCTIONAL: one u64 per env)
    if (p.inl) q.inl = p.inl + o * 2;
    h->L.step(q, h->sh_stream[i]);
  }
  return TG_OK;
}

// after a wait for the handle's stream: did a tg_set_state_dev since the last look upload something tg_set_state would have refused?
int check_dev_state(tg_handle* h) {
  if (h->h_nib_over && *(volatile uint32_t*)h->h_nib_over) {  // (host memory the kernel writes: visible once the launch is complete)
    *(volatile uint32_t*)h->h_nib_over = 0;
    return fail(h, TG_EINVAL, "a locked piece added its id onto filled cells and the sum left the nibble range of this handle's packed board "
                              "(only a state uploaded with the piece on top of filled cells does that): the boards are no longer what the "
                              "reference would hold; TG_PACKED=0 creates handles that keep bytes");
  }
  if (h->fn_overlap_pending) {  // (the stream has been waited for: fn_overlap_kernel's answer is there)
    uint32_t w = 0;
    if (h->h_fn_over) w = *(volatile uint32_t*)h->h_fn_over;
    else HIPCHK(h, hipMemcpy(&w, (uint32_t*)h->d_err + 4, sizeof(uint32_t), hipMemcpyDeviceToHost));
    h->fn_overlap = w != 0;
    h->fn_overlap_pending = false;
  }
  if (!h->dev_state_pending) return TG_OK;
  h->dev_state_pending = false;
  int flags = 0;
  HIPCHK(h, hipMemcpy(&flags, h->d_err, sizeof(int), hipMemcpyDeviceToHost));
  if (!flags) return TG_OK;
  HIPCHK(h, hipMemsetAsync(h->d_err, 0, sizeof(int), h->stream));
  if (flags & 4)
    return fail(h, TG_EUNSUPPORTED, "tg_set_state_dev: an uploaded board holds a row of bedrock cells (value 1 in every column) inside the four "
                                    "spawn rows: not supported (bedrock rows from row 4 down are); the engine holds the board as uploaded%s",
                (flags & 3) ? " (and other fields were out of range and clamped)" : "");
  return fail(h, TG_EINVAL, "tg_set_state_dev: the uploaded device state held values tg_set_state refuses (%s%s); they were clamped so that "
                            "nothing indexes out of range — the engine's state is not what was uploaded",
              (flags & 1) ? "record field out of range" : "", (flags & 2) ? ((flags & 1) ? ", board byte out of range" : "board byte out of range") : "");
}

// packed handles: bring the uint8 TG_BUF_BOARD up to date before anything reads it (the step kernel maintains the nibbles)
int ensure_board(tg_handle* h) {
  if (!h->d_pboard || !h->board_stale) return TG_OK;
  const size_t nb = (size_t)h->B * h->cells / 2;
  hipLaunchKernelGGL(tg::unpack_board_kernel, dim3((unsigned)((nb / 4 + 256) / 256)), dim3(256), 0, h->stream, h->d_pboard,
                     (uint8_t*)h->buf[TG_BUF_BOARD], nb);
  HIPCHK(h, hipGetLastError());
  h->board_stale = false;
  return TG_OK;
}

// bits kernel: rebuild column masks + look-ahead permutations from the reference-defined state
int launch_prepare_bits(tg_handle* h, const uint8_t* d_mask, int force) {
  if (!h->d_cols) return TG_OK;
  if (h->d_pboard) {  // TG_BUF_BOARD has just been (re)written by tg_reset / tg_set_state: the nibbles follow
    const size_t nb = (size_t)h->B * h->cells / 2;
    hipLaunchKernelGGL(tg::pack_board_kernel, dim3((unsigned)((nb / 4 + 256) / 256)), dim3(256), 0, h->stream,
                       (const uint8_t*)h->buf[TG_BUF_BOARD], h->d_pboard, nb);
    HIPCHK(h, hipGetLastError());
    h->board_stale = false;
  }
  tg::PrepareParams q{};
  q.board = (const uint8_t*)h->buf[TG_BUF_BOARD];
  q.rec = (uint32_t*)h->buf[TG_BUF_REC];
  q.rng = (tg_rng_rec*)h->buf[TG_BUF_RNG];
  q.cols = h->d_cols;
  q.ahead = h->d_ahead;
  q.ring = h->d_ring;
  q.hbuf = h->d_hbuf;
  q.cur = h->ring_cur;
  q.hcur = h->hbuf_cur;
  q.mask = d_mask;
  q.batch = h->B;
  q.ruleset = h->cfg.ruleset;
  q.force = force;
  q.queue_size = h->Q;
  q.err = h->d_err;
  h->L.prepare(q, h->stream);
  HIPCHK(h, hipGetLastError());
  return TG_OK;
}

// bits kernel, NUMPY rules: bring TG_BUF_RNG up to date with the generators the reference would hold now (a transition
// leaves the state in the permutation ring's `ahead` slots, tg_kernel_bits.h)
int launch_materialize_rng(tg_handle* h) {
  if (!h->d_cols || h->cfg.ruleset != TG_RULESET_NUMPY) return TG_OK;  // (FUNCTIONAL: the step kernel keeps State.rng_key current)
  tg::MaterializeParams q{};
  q.rng = (tg_rng_rec*)h->buf[TG_BUF_RNG];
  q.ahead = h->d_ahead;
  q.hbuf = h->d_hbuf + (size_t)h->hbuf_cur * (size_t)h->B;
  q.inl = h->d_inl;
  q.batch = h->B;
  hipLaunchKernelGGL(tg::bits_materialize_rng_kernel, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, q);
  HIPCHK(h, hipGetLastError());
  return TG_OK;
}

// Whether the next launch carries the RNG-role workgroups (p.nsteps is set): launches of several transitions consume several
// ring entries per env and always do; one-transition launches every rng_every-th time (TG_RNG_EVERY).
void next_rng_role(tg_handle* h, tg::KParams& p) {
  p.rng_role = p.nsteps > 1 || (h->launch_no++ % (unsigned)h->rng_every) == 0;
}
// ... and what flips afterwards: the ring parity when the role ran, the consumed-counter parity always
void after_launch(tg_handle* h, const tg::KParams& p) {
  if (p.rng_role) h->ring_cur ^= 1;
  h->hbuf_cur ^= 1;
  if (h->d_pboard) h->board_stale = true;
}

int run_grouped_mask(tg_handle* h);  // (defined with the grouped entry points below)

// TG_BUF_HOST_PACK is packed when somebody asks for it (the step path never pays for it)
int launch_host_pack(tg_handle* h) {
  PackParams q{};
  q.out = (const uint2*)h->buf[TG_BUF_STEP_OUT];
  q.ep = (const tg_episode_rec*)h->buf[TG_BUF_EPISODE];
  q.pack = (uint8_t*)h->buf[TG_BUF_HOST_PACK];
  size_t off[6];
  host_pack_layout((size_t)h->B, off);
  for (int i = 0; i < 5; ++i) q.off[i] = off[i];
  q.batch = h->B;
  q.custom = (h->cfg.flags & TG_FLAG_CUSTOM_REWARDS) ? 1 : 0;
  q.width = h->W;
  q.r_alife = h->rewards[0]; q.r_game_over = h->rewards[2]; q.r_invalid = h->rewards[3];
  hipLaunchKernelGGL(host_pack_kernel, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, q);
  HIPCHK(h, hipGetLastError());
  return TG_OK;
}

// the enqueue half of tg_fetch_many (no wait)
int enqueue_fetches(tg_handle* h, int m, const int* which, void* const* host_dst, const size_t* bytes, const size_t* src_off) {
  for (int i = 0; i < m; ++i) {
    if (which[i] < 0 || which[i] >= TG_BUF_COUNT || (!host_dst[i] && bytes[i])) return fail(h, TG_EINVAL, "tg_fetch_many: bad request %d", i);
    const size_t so = src_off ? src_off[i] : 0;
    if (so > h->bytes[which[i]] || bytes[i] > h->bytes[which[i]] - so)
      return fail(h, TG_EINVAL, "tg_fetch_many: %zu bytes at offset %zu requested, buffer %d holds %zu", bytes[i], so, which[i], h->bytes[which[i]]);
  }
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  for (int i = 0; i < m; ++i) {
    if (which[i] == TG_BUF_RNG) { int rc = launch_materialize_rng(h); if (rc != TG_OK) return rc; }
    if (which[i] == TG_BUF_HOST_PACK) { int rc = launch_host_pack(h); if (rc != TG_OK) return rc; }
    if (which[i] == TG_BUF_BOARD) { int rc = ensure_board(h); if (rc != TG_OK) return rc; }
    if (which[i] == TG_BUF_LEGAL_MASK && h->cfg.action_set == TG_ACTIONS_GROUPED) {  // the mask of the state as it stands now
      int rc = run_grouped_mask(h);
      if (rc != TG_OK) return rc;
    }
  }
  for (int i = 0; i < m;) {
    // neighbours in device memory and, at the same distance, in host memory leave as one copy (the gap between two pieces of
    // the arena is its tail padding: a few KiB)
    const char* d0 = (const char*)h->buf[which[i]] + (src_off ? src_off[i] : 0);
    char* h0 = (char*)host_dst[i];
    size_t len = bytes[i];
    int j = i + 1;
    while (j < m && bytes[j] && len) {
      const char* dj = (const char*)h->buf[which[j]] + (src_off ? src_off[j] : 0);
      const ptrdiff_t dd = dj - d0, hd = (char*)host_dst[j] - h0;
      if (dd != hd || dd < (ptrdiff_t)len || dd - (ptrdiff_t)len > 16384) break;
      len = (size_t)dd + bytes[j];
      ++j;
    }
    if (len) HIPCHK(h, hipMemcpyAsync(h0, d0, len, hipMemcpyDeviceToHost, h->stream));
    i = j;
  }
  return TG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Enqueue pool of tg_step_fetch_many: one persistent worker per additional handle, so that the launches of n devices are
// issued side by side instead of one after the other (a launch is ~3 us of host time; eight in a row are 27 us, measured
// with eight shards of one GPU: bench.py `vector_env_shards`).  A worker spins for its next job for a short while after it
// finished one (a stepping loop hands it the next within microseconds) and blocks on a condition variable otherwise, so
// an idle process burns nothing.
// ---------------------------------------------------------------------------------------------------------------------
class EnqueuePool {
 public:
  static EnqueuePool& get() { static EnqueuePool p; return p; }
  // run job(k) for k = 0 .. n - 1 (k = 0 on the calling thread); returns the first non-zero result
  int run(int n, const std::function<int(int)>& job) {
    if (n <= 1) {
      for (int k = 0; k < n; ++k) { const int rc = job(k); if (rc != TG_OK) return rc; }
      return TG_OK;
    }
    std::lock_guard<std::mutex> serial(call_mu_);  // (one tg_step_fetch_many at a time uses the workers)
    while ((int)workers_.size() < n - 1) workers_.emplace_back(new Worker());
    job_ = &job;
    for (int k = 1; k < n; ++k) workers_[k - 1]->post(this, k);
    int rc = job(0);
    for (int k = 1; k < n; ++k) { const int r = workers_[k - 1]->wait(); if (rc == TG_OK) rc = r; }
    return rc;
  }

 private:
  struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<int> state{0};  // 0 idle, 1 job posted, 2 done, 3 exit
    EnqueuePool* pool = nullptr;
    int k = 0, rc = 0;
    Worker() { th = std::thread([this] { loop(); }); }
    ~Worker() {
      { std::lock_guard<std::mutex> l(m); state.store(3, std::memory_order_release); }
      cv.notify_one();
      if (th.joinable()) th.join();
    }
    void post(EnqueuePool* p, int kk) {
      pool = p; k = kk;
      { std::lock_guard<std::mutex> l(m); state.store(1, std::memory_order_release); }
      cv.notify_one();
    }
    int wait() {
      // the job is normally a handful of asynchronous HIP calls (microseconds): spin briefly, then give the core away — with
      // pageable destinations a worker's hipMemcpyAsync blocks for milliseconds
      for (int spin = 0; state.load(std::memory_order_acquire) != 2; ++spin)
        if (spin > 4000) std::this_thread::yield();
      state.store(0, std::memory_order_release);
      return rc;
    }
    void loop() {
      for (;;) {
        int st = 0;
        for (int spin = 0; spin < 200000; ++spin) {  // (~100 us of polling before going to sleep)
          st = state.load(std::memory_order_acquire);
          if (st == 1 || st == 3) break;
        }
        if (st != 1 && st != 3) {
          std::unique_lock<std::mutex> l(m);
          cv.wait(l, [&] { const int v = state.load(std::memory_order_acquire); return v == 1 || v == 3; });
          st = state.load(std::memory_order_acquire);
        }
        if (st == 3) return;
        rc = (*pool->job_)(k);
        state.store(2, std::memory_order_release);
      }
    }
  };
  std::vector<std::unique_ptr<Worker>> workers_;
  const std::function<int(int)>* job_ = nullptr;
  std::mutex call_mu_;
};

}  // namespace

extern "C" {

int tg_abi_version(void) { return TG_ABI_VERSION; }

int tg_host_pack_layout(const tg_config* cfg, size_t offsets[6]) {
  if (!cfg || !offsets || cfg->batch <= 0) return TG_EINVAL;
  host_pack_layout((size_t)cfg->batch, offsets);
  return TG_OK;
}

int tg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int tg_device_pci_bus_id(int device, char* buf, int len) {
  if (!buf || len < 13) return TG_EINVAL;
  buf[0] = 0;
  return hipDeviceGetPCIBusId(buf, len, device) == hipSuccess ? TG_OK : TG_EHIP;
}

const char* tg_last_error(const tg_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int tg_obs_bytes_per_env(const tg_config* cfg) { return cfg ? (int)obs_bytes_per_env(cfg) : TG_EINVAL; }

int tg_num_actions(const tg_config* cfg) {
  if (!cfg) return TG_EINVAL;
  if (cfg->action_set == TG_ACTIONS_GROUPED) return 4 * cfg->width;
  return cfg->ruleset == TG_RULESET_NUMPY ? 8 : 7;
}

int tg_seed_pcg64(uint64_t seed, tg_rng_rec* out) {
  if (!out) return TG_EINVAL;
  seed_pcg64_host(seed, out);
  return TG_OK;
}

int tg_create(const tg_config* cfg, tg_handle** out) {
  if (!cfg || !out) return fail(nullptr, TG_EINVAL, "tg_create: null argument");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(tg_config))
    return fail(nullptr, TG_EINVAL, "tg_create: tg_config.struct_size %d != %zu (ABI mismatch)", cfg->struct_size,
                sizeof(tg_config));
  if (cfg->padding != 4)
    return fail(nullptr, TG_EUNSUPPORTED, "tg_create: padding must be 4 (stock tetromino set), got %d", cfg->padding);
  if (cfg->batch <= 0) return fail(nullptr, TG_EINVAL, "tg_create: batch must be > 0");
  if ((uint64_t)cfg->batch * (uint64_t)cfg->width * (uint64_t)cfg->height >= (1ull << 31))
    return fail(nullptr, TG_EINVAL, "tg_create: batch*width*height must stay below 2^31 (32-bit slab offsets)");
  if (cfg->ruleset != TG_RULESET_NUMPY && cfg->ruleset != TG_RULESET_FUNCTIONAL)
    return fail(nullptr, TG_EINVAL, "tg_create: unknown ruleset %d", cfg->ruleset);
  if (cfg->obs_mode < TG_OBS_NONE || cfg->obs_mode > TG_OBS_RGB)
    return fail(nullptr, TG_EINVAL, "tg_create: unknown obs_mode %d", cfg->obs_mode);
  if (cfg->autoreset < TG_AUTORESET_NONE || cfg->autoreset > TG_AUTORESET_NEXT_STEP)
    return fail(nullptr, TG_EINVAL, "tg_create: unknown autoreset mode %d", cfg->autoreset);
  if (cfg->ruleset == TG_RULESET_NUMPY && (cfg->queue_size < 1 || cfg->queue_size > 7))
    return fail(nullptr, TG_EINVAL, "tg_create: NUMPY queue_size must be 1..7, got %d", cfg->queue_size);
  // (functional/queue.py:35 permutes arange(queue_size): 1..7 deals the first queue_size pieces from a bag of that length; larger values
  // would index past the seven tetrominoes)
  if (cfg->ruleset == TG_RULESET_FUNCTIONAL && (cfg->queue_size < 1 || cfg->queue_size > 7))
    return fail(nullptr, TG_EUNSUPPORTED, "tg_create: FUNCTIONAL queue_size (the bag length) must be 1..7, got %d", cfg->queue_size);
  if ((cfg->obs_mode == TG_OBS_DICT || cfg->obs_mode == TG_OBS_RGB) && cfg->ruleset != TG_RULESET_NUMPY)
    return fail(nullptr, TG_EINVAL, "tg_create: TG_OBS_DICT / TG_OBS_RGB are observations of the NUMPY rule-set");
  if (cfg->action_set != TG_ACTIONS_BASE && cfg->action_set != TG_ACTIONS_GROUPED)
    return fail(nullptr, TG_EINVAL, "tg_create: unknown action_set %d", cfg->action_set);
  if (cfg->action_set == TG_ACTIONS_GROUPED && cfg->ruleset != TG_RULESET_NUMPY)
    return fail(nullptr, TG_EINVAL, "tg_create: the grouped action set wraps the NUMPY rule-set (wrappers/grouped.py)");
  if ((cfg->flags & (TG_FLAG_CUSTOM_ACTIONS | TG_FLAG_CUSTOM_REWARDS)) && cfg->ruleset != TG_RULESET_NUMPY)
    return fail(nullptr, TG_EINVAL, "tg_create: ActionsMapping / RewardsMapping belong to the NUMPY rule-set (envs/tetris.py:83-84)");
  uint64_t act_lut = tg::kActLutNumpy;
  if (cfg->flags & TG_FLAG_CUSTOM_ACTIONS) act_lut_from_ids(cfg->action_ids, &act_lut);
  if (cfg->flags & TG_FLAG_CUSTOM_REWARDS)
    for (int i = 0; i < 4; ++i)
      if (!(cfg->rewards[i] == cfg->rewards[i]) || cfg->rewards[i] > 3.0e38 || cfg->rewards[i] < -3.0e38)
        return fail(nullptr, TG_EINVAL, "tg_create: rewards[%d] is not a finite float32-range number", i);
  Launchers L{};
  const bool built_in = find_launchers(cfg->width, cfg->height, cfg->ruleset, &L);
  if (!built_in && (cfg->width < 4 || cfg->width > 32 || cfg->height < 4 || cfg->height > 58))
    return fail(nullptr, TG_EUNSUPPORTED, "tg_create: board %dx%d outside 4..32 x 4..58 (row masks are 32 bits, column masks 64)",
                cfg->width, cfg->height);
// (argument checks above need no device: a CPU-only machine gets the precise refusal, then the loud one below)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, TG_EHIP, "tg_create: no HIP device available (this engine has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(nullptr, TG_EINVAL, "tg_create: device %d out of range (0..%d)", cfg->device, ndev - 1);
  std::call_once(g_process_once, process_init, cfg->device);
  if (!g_process_err.empty()) return fail(nullptr, TG_EUNSUPPORTED, "%s", g_process_err.c_str());
  if (cfg->ruleset == TG_RULESET_FUNCTIONAL && cfg->queue_size != 7 && g_kernel_lanes != 16)
    return fail(nullptr, TG_EUNSUPPORTED, "tg_create: the earlier step kernels of the test-only library deal the 7-bag only");
  if ((cfg->flags & (TG_FLAG_CUSTOM_ACTIONS | TG_FLAG_CUSTOM_REWARDS)) && g_kernel_lanes != 16)
    return fail(nullptr, TG_EUNSUPPORTED, "tg_create: custom mappings need the default step kernel (TG_KERNEL=bits)");
  if (!built_in && g_kernel_lanes != 16)  // (the run-time compiled module holds the default step kernel only: launching it for a handle without column masks faults)
    return fail(nullptr, TG_EUNSUPPORTED, "tg_create: board %dx%d is not built in (%s); the earlier step kernels of the test-only library exist for the "
                "built-in sizes only", cfg->width, cfg->height, built_sizes().c_str());
  if (!built_in) {  // any other size: the same kernels, compiled for it now (or taken from the on-disk cache)
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, TG_EHIP, "tg_create: hipSetDevice failed");
    RtcSize* rs = nullptr;
    const std::string err = rtc_build(cfg->device, cfg->width, cfg->height, cfg->ruleset, &rs);
    if (!err.empty())
      return fail(nullptr, TG_EUNSUPPORTED, "tg_create: board %dx%d is not built in (%s) and %s", cfg->width, cfg->height,
                  built_sizes().c_str(), err.c_str());
    L = rtc_launchers(rs);
  }

  tg_handle* h = new (std::nothrow) tg_handle();
  if (!h) return fail(nullptr, TG_ENOMEM, "tg_create: out of host memory");
  h->cfg = *cfg;
  h->act_lut = act_lut;
  if (cfg->flags & TG_FLAG_CUSTOM_REWARDS) for (int i = 0; i < 4; ++i) h->rewards[i] = cfg->rewards[i];
  h->B = cfg->batch; h->W = cfg->width; h->H = cfg->height; h->Q = cfg->queue_size;
  h->cells = (size_t)h->W * h->H;
  h->L = L;
  auto bail = [&](int code) {
    g_create_err = h->err;
    tg_destroy(h);
    return code;
  };
#define CRCHK(call)                                                                               \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      h->err = std::string(#call " failed: ") + hipGetErrorString(e_);                            \
      return bail(TG_EHIP);                                                                       \
    }                                                                                             \
  } while (0)
  CRCHK(hipSetDevice(cfg->device));
  const int n_streams_env = cfg->streams > 1 ? cfg->streams : 1;  // tg_config.streams: spread a transition of a large batch over this many streams
  CRCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  CRCHK(hipEventCreate(&h->ev0));
  CRCHK(hipEventCreate(&h->ev1));
  const size_t B = (size_t)h->B;
  const bool want_final = (cfg->flags & TG_FLAG_FINAL_OBS) && cfg->autoreset == TG_AUTORESET_SAME_STEP &&
                          (cfg->obs_mode == TG_OBS_BOARD || cfg->obs_mode == TG_OBS_FN);
  const size_t sizes[TG_BUF_COUNT] = {
      /*OBS*/ B * obs_bytes_per_env(cfg), /*STEP_OUT*/ B * sizeof(tg_step_out), /*ACTIONS*/ B * 4,
      /*BOARD*/ B * h->cells, /*REC*/ B * sizeof(tg_env_rec), /*RNG*/ B * sizeof(tg_rng_rec),
      /*LEGAL_MASK*/ cfg->action_set == TG_ACTIONS_GROUPED ? B * 4 * (size_t)cfg->width : 0,
      /*FINAL_OBS*/ want_final ? B * h->cells : 0,
      /*EPISODE*/ (cfg->flags & TG_FLAG_EPISODE_STATS) ? B * sizeof(tg_episode_rec) : 0,
      /*PLACEMENT_FEATS*/ (cfg->flags & TG_FLAG_FEATURES) ? B * 4 * (size_t)cfg->width * (size_t)(cfg->width + 3) : 0,
      /*FEATS*/ (cfg->flags & TG_FLAG_FEATURES) ? B * (size_t)(cfg->width + 3) : 0,
      /*HOST_PACK*/ host_pack_layout(B, nullptr)};
  {
    size_t total = 0;
    for (int i = 0; i < TG_BUF_COUNT; ++i) total += sizes[i] + kTailPad + 4096;
    total += B * sizeof(tg_rng_rec) + 2 * B + B * L.cols_bytes + B * 32 * tg::kRing + 2 * B * 16 + 8 * B + 2 * B * 32 + B * h->cells / 2 + 64 + 11 * (kTailPad + 4096);
    CRCHK(hipMalloc(&h->arena, total));
    h->arena_size = total;
    CRCHK(hipMemsetAsync(h->arena, 0, total, h->stream));
  }
  {  // (fixed order; TG_BUF_HOST_PACK directly behind TG_BUF_OBS: a step's host-bound results are one device range)
    int order[TG_BUF_COUNT], k = 0;
    order[k++] = TG_BUF_OBS;
    order[k++] = TG_BUF_HOST_PACK;
    for (int i = 0; i < TG_BUF_COUNT; ++i)
      if (i != TG_BUF_OBS && i != TG_BUF_HOST_PACK) order[k++] = i;
    for (int i = 0; i < TG_BUF_COUNT; ++i) {
      int rc = alloc_buf(h, order[i], sizes[order[i]]);
      if (rc != TG_OK) return bail(rc);
    }
  }
  CRCHK(arena_take(h, (void**)&h->d_new_rng, B * sizeof(tg_rng_rec)));
  CRCHK(arena_take(h, (void**)&h->d_reseed, B));
  CRCHK(arena_take(h, (void**)&h->d_mask, B));
  CRCHK(arena_take(h, (void**)&h->d_err, 64));
  if (g_kernel_lanes == 16) {
    CRCHK(arena_take(h, &h->d_cols, B * L.cols_bytes + kTailPad));
    CRCHK(arena_take(h, (void**)&h->d_ahead, B * 32 * tg::kRing + kTailPad));
    CRCHK(arena_take(h, (void**)&h->d_ring, 2 * B * 16 + kTailPad));
    CRCHK(arena_take(h, (void**)&h->d_hbuf, 2 * B * sizeof(uint32_t) + kTailPad));
    CRCHK(arena_take(h, (void**)&h->d_inl, 2 * B * 32 + kTailPad));
    // Packed board (kFeatPacked): for the working sets that do not fit the Infinity Cache the launch is bound by bytes, and
    // the board as nibbles is 17-19 % fewer of them.  TG_PACKED=1 / 0 forces / forbids it (tests, A/B).
    {
      const bool can = packed_geometry(cfg->width, cfg->height) && !want_final &&
                       (cfg->obs_mode == TG_OBS_BOARD || cfg->obs_mode == TG_OBS_NONE ||
                        (cfg->obs_mode == TG_OBS_FN && cfg->ruleset == TG_RULESET_FUNCTIONAL)) &&
                       n_streams_env <= 1;
      // (measured, profiles/r04/ab_packed_board.log: config 5 24.3 -> 21.7 us, 524 288 envs at 10x20 46.7 -> 41.7, 131 072
      // 11.7 -> 11.1; at 65 536 x 10x20 — Infinity-Cache resident, bound by instruction issue — 6.30 -> 6.87: hence the size rule;
      // FUNCTIONAL, ab_fn_packed_board.log: 524 288 envs 49.8 -> 45.0, 262 144 26.2 -> 24.5, 131 072 14.7 -> 14.4)
      bool want = can && B * h->cells >= ((size_t)20 << 20);
      if (const char* e = getenv("TG_PACKED")) want = can && atoi(e) != 0;
      if (want) CRCHK(arena_take(h, (void**)&h->d_pboard, B * h->cells / 2 + kTailPad));
      if (want && cfg->ruleset == TG_RULESET_FUNCTIONAL) {
        CRCHK(hipHostMalloc((void**)&h->h_nib_over, 64, hipHostMallocDefault));
        *h->h_nib_over = 0;
      }
    }
    if (cfg->ruleset == TG_RULESET_FUNCTIONAL && cfg->obs_mode != TG_OBS_NONE) {
      CRCHK(hipHostMalloc((void**)&h->h_fn_over, 64, hipHostMallocDefault));
      *h->h_fn_over = 0;
      CRCHK(hipEventCreateWithFlags(&h->fn_ev, hipEventDisableTiming));
    }
  }
#ifdef TG_PHASE_TIMING
  CRCHK(hipMalloc(&h->dbg, ((B + 15) / 16) * 16 * 8));
  CRCHK(hipMemset(h->dbg, 0, ((B + 15) / 16) * 16 * 8));
#endif
  // a valid (if arbitrary) initial record so that a step before the first reset cannot index out of range
  {
    std::vector<tg_env_rec> init(B);
    std::memset(init.data(), 0, B * sizeof(tg_env_rec));
    for (size_t i = 0; i < B; ++i) { init[i].val = 2; init[i].x = 7; init[i].holder_piece = TG_NONE; }
    CRCHK(hipMemcpyAsync(h->buf[TG_BUF_REC], init.data(), B * sizeof(tg_env_rec), hipMemcpyHostToDevice, h->stream));
    CRCHK(hipStreamSynchronize(h->stream));
  }
  {
    h->dict_in_kernel = h->d_cols &&
                        ((cfg->obs_mode == TG_OBS_DICT && L.dict_ok) ||
                         (cfg->obs_mode == TG_OBS_RGB && L.rgb_ok && cfg->queue_size == 4));  // (the in-kernel RGB image is laid out for four queue pieces)
    // launch knobs: tg_config (0 = default), overridden by the environment (test hooks)
    if (cfg->ring_fill_p1 > 0) h->ring_fill = cfg->ring_fill_p1 - 1 > tg::kRing - 1 ? tg::kRing - 1 : cfg->ring_fill_p1 - 1;
    if (cfg->rng_every > 0) h->rng_every = cfg->rng_every > 3 ? 3 : cfg->rng_every;
    if (cfg->fuse > 0) h->fuse_max = cfg->fuse > tg::kMaxFuse ? tg::kMaxFuse : cfg->fuse;
    if (const char* e = getenv("TG_RING_FILL")) { const int v = atoi(e); h->ring_fill = v < 0 ? 0 : (v > tg::kRing - 1 ? tg::kRing - 1 : v); }
    if (const char* e = getenv("TG_RNG_EVERY")) { const int v = atoi(e); h->rng_every = v < 1 ? 1 : (v > 3 ? 3 : v); }
    if (const char* e = getenv("TG_FN_LAZY")) { const int v = atoi(e); h->fn_lazy = v < 0 ? 0 : (v & 7); }
    if (const char* e = getenv("TG_FUSE")) { const int v = atoi(e); h->fuse_max = v < 1 ? 1 : (v > tg::kMaxFuse ? tg::kMaxFuse : v); }
    // shards only for the plain step path of the bits kernel (observation written by the step kernel itself)
    int want = n_streams_env;
    if (!h->d_cols || cfg->obs_mode == TG_OBS_DICT || cfg->obs_mode == TG_OBS_RGB || (cfg->flags & TG_FLAG_FINAL_OBS)) want = 1;
    while (want > 1 && (B + want - 1) / want < 8192) --want;  // small shards only add launches
    if (want > 1) {
      const int chunk = (int)(((B + want - 1) / want + 127) / 128 * 128);  // RNG workgroups own 128-env ranges
      for (int o = 0; o < (int)B; o += chunk) {
        h->sh_off.push_back(o);
        h->sh_cnt.push_back((int)B - o < chunk ? (int)B - o : chunk);
      }
      h->n_shards = (int)h->sh_off.size();
      h->sh_stream.assign(h->n_shards, nullptr);
      h->sh_event.assign(h->n_shards, nullptr);
      for (int i = 0; i < h->n_shards; ++i) {
        CRCHK(hipStreamCreateWithFlags(&h->sh_stream[i], hipStreamNonBlocking));
        CRCHK(hipEventCreateWithFlags(&h->sh_event[i], hipEventDisableTiming));
      }
      CRCHK(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
    }
  }
#undef CRCHK
  h->rng_valid.assign(B, 0);
  *out = h;
  return TG_OK;
}

void tg_destroy(tg_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  for (hipStream_t st : h->sh_stream)
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  for (hipEvent_t ev : h->sh_event)
    if (ev) (void)hipEventDestroy(ev);
  if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->arena) (void)hipFree(h->arena);
  if (h->h_nib_over) (void)hipHostFree(h->h_nib_over);
  if (h->h_fn_over) (void)hipHostFree(h->h_fn_over);
  if (h->fn_ev) (void)hipEventDestroy(h->fn_ev);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int tg_reset(tg_handle* h, const uint64_t* seeds, const uint8_t* mask) {
  if (!h) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  { int rcm = launch_materialize_rng(h); if (rcm != TG_OK) return rcm; }  // an unseeded reset continues the running generator
  { int rcb = ensure_board(h); if (rcb != TG_OK) return rcb; }           // (a masked reset keeps the other envs' boards)
  const size_t B = (size_t)h->B;
  bool any_reseed = false;
  h->h_new_rng.resize(B);
  h->h_reseed.assign(B, 0);
  for (size_t i = 0; i < B; ++i) {
    if (mask && !mask[i]) continue;
    if (h->cfg.ruleset == TG_RULESET_NUMPY) {
      uint64_t sd = seeds ? seeds[i] : 0;
      if (sd == 0 && !h->rng_valid[i]) sd = 1 + (uint64_t)h->cfg.env_offset + i;  // never-seeded env
      if (sd > 0) {  // components/tetromino_randomizer.py:40-46: `if seed and seed > 0`
        seed_pcg64_host(sd, &h->h_new_rng[i]);
        h->h_reseed[i] = 1;
        h->rng_valid[i] = 1;
        any_reseed = true;
      }
    } else if (seeds) {
      std::memset(&h->h_new_rng[i], 0, sizeof(tg_rng_rec));
      h->h_new_rng[i].key[0] = (uint32_t)(seeds[i] >> 32);
      h->h_new_rng[i].key[1] = (uint32_t)seeds[i];
      h->h_reseed[i] = 1;
      any_reseed = true;
    }
  }
  tg::ResetParams p{};
  p.board = (uint8_t*)h->buf[TG_BUF_BOARD];
  p.rec = (uint32_t*)h->buf[TG_BUF_REC];
  p.rng = (tg_rng_rec*)h->buf[TG_BUF_RNG];
  p.obs = (uint8_t*)h->buf[TG_BUF_OBS];
  p.episode = (tg_episode_rec*)h->buf[TG_BUF_EPISODE];
  p.batch = h->B;
  p.obs_mode = h->cfg.obs_mode;
  p.queue_size = h->Q;
  if (any_reseed) {
    HIPCHK(h, hipMemcpyAsync(h->d_new_rng, h->h_new_rng.data(), B * sizeof(tg_rng_rec), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_reseed, h->h_reseed.data(), B, hipMemcpyHostToDevice, h->stream));
    p.new_rng = h->d_new_rng;
    p.reseed = h->d_reseed;
  }
  if (mask) {
    HIPCHK(h, hipMemcpyAsync(h->d_mask, mask, B, hipMemcpyHostToDevice, h->stream));
    p.mask = h->d_mask;
  }
  h->L.reset(p, h->stream);
  HIPCHK(h, hipGetLastError());
  if (!mask) h->fn_overlap = h->fn_overlap_pending = false;  // every env starts over: no uploaded piece is left on filled cells
  {
    int rc = launch_prepare_bits(h, p.mask, 0);
    if (rc != TG_OK) return rc;
  }
  if (h->cfg.obs_mode == TG_OBS_DICT || h->cfg.obs_mode == TG_OBS_RGB) {
    int rc = launch_observe_dict(h);
    if (rc != TG_OK) return rc;
  }
  // host staging vectors are reused by the next call: wait for the uploads
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return TG_OK;
}

// envs/tetris.py:215-217: the reference asserts action_space.contains(action)
static int check_host_actions(tg_handle* h, const int32_t* actions) {
  if (h->cfg.ruleset != TG_RULESET_NUMPY) return TG_OK;  // (functional: lax.switch clamps, tetris_fn.py:214-241)
  const int na = tg_num_actions(&h->cfg);
  for (int i = 0; i < h->B; ++i)
    if (actions[i] < 0 || actions[i] >= na)
      return fail(h, TG_EINVAL, "tg_step: action %d of env %d outside Discrete(%d)", actions[i], i, na);
  return TG_OK;
}

// `waited`: the caller waits for the handle's stream before it returns (tg_step_fetch_many with copies).  A small handle then reads
// page-locked host actions in place — one dependent copy less in front of the launch (3 us of a 23 us single-env step).
static int step_impl(tg_handle* h, const int32_t* actions, int kind, bool waited) {
  if (!h || !actions) return h ? fail(h, TG_EINVAL, "tg_step: null actions") : TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  tg::KParams p = step_params(h);
  if (kind == TG_MEM_HOST) {
    int rcj = join_shards(h);  // the staging copy runs on the handle's stream, behind the previous transition
    if (rcj != TG_OK) return rcj;
    { const int rca = check_host_actions(h, actions); if (rca != TG_OK) return rca; }
    const void* in_place = nullptr;
    if (waited && h->B <= 256) {
      hipPointerAttribute_t at{};
      if (hipPointerGetAttributes(&at, actions) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer)
        in_place = at.devicePointer;
      else
        (void)hipGetLastError();  // (pageable memory is "invalid value" to the query: not an error of this call)
    }
    if (!in_place) HIPCHK(h, hipMemcpyAsync(h->buf[TG_BUF_ACTIONS], actions, (size_t)h->B * 4, hipMemcpyHostToDevice, h->stream));
    p.actions = in_place ? in_place : h->buf[TG_BUF_ACTIONS];
  } else if (kind == TG_MEM_DEVICE) {
    p.actions = actions;
  } else {
    return fail(h, TG_EINVAL, "tg_step: unknown memory kind %d", kind);
  }
  next_rng_role(h, p);
  {
    int rc = launch_step_all(h, p);
    if (rc != TG_OK) return rc;
  }
  after_launch(h, p);
  HIPCHK(h, hipGetLastError());
  const bool exact = fn_exact_path(h);
  if (((h->cfg.obs_mode == TG_OBS_DICT || h->cfg.obs_mode == TG_OBS_RGB) && !h->dict_in_kernel) || exact) {
    const int rco = launch_observe_dict(h);
    if (rco != TG_OK) return rco;
    if (exact && h->fn_overlap) return launch_fn_overlap(h);  // (does a piece still lie on filled cells? — see fn_exact_path)
  }
  return TG_OK;
}

int tg_step(tg_handle* h, const int32_t* actions, int kind) { return step_impl(h, actions, kind, false); }

__global__ void narrow_actions_kernel(const int32_t* src, int8_t* dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (int8_t)src[i];
}

static int run_grouped(tg_handle* h, bool mask, bool sample, uint32_t seed, uint32_t counter);

// the common body of tg_step_k / tg_rollout (actions given) and tg_grouped_rollout (actions drawn on the device)
static int rollout_impl(tg_handle* h, const int8_t* actions_dev, bool sampled, uint32_t seed, uint32_t counter0, int K,
                        int8_t* act_traj_dev, uint8_t* obs_traj_dev, tg_step_out* out_traj_dev);

int tg_step_k(tg_handle* h, const int8_t* actions_dev, int K) { return tg_rollout(h, actions_dev, K, nullptr, nullptr); }

int tg_rollout(tg_handle* h, const int8_t* actions_dev, int K, uint8_t* obs_traj_dev, tg_step_out* out_traj_dev) {
  if (!h || !actions_dev || K <= 0) return h ? fail(h, TG_EINVAL, "tg_rollout: bad arguments") : TG_EINVAL;
  return rollout_impl(h, actions_dev, false, 0, 0, K, nullptr, obs_traj_dev, out_traj_dev);
}

int tg_grouped_rollout(tg_handle* h, uint32_t seed, uint32_t counter0, int K, int8_t* actions_out_dev,
                       uint8_t* obs_traj_dev, tg_step_out* out_traj_dev) {
  if (!h || K <= 0) return h ? fail(h, TG_EINVAL, "tg_grouped_rollout: bad arguments") : TG_EINVAL;
  if (h->cfg.action_set != TG_ACTIONS_GROUPED)
    return fail(h, TG_EINVAL, "tg_grouped_rollout: handle was not created with TG_ACTIONS_GROUPED");
  return rollout_impl(h, nullptr, true, seed, counter0, K, actions_out_dev, obs_traj_dev, out_traj_dev);
}

static int rollout_impl(tg_handle* h, const int8_t* actions_dev, bool sampled, uint32_t seed, uint32_t counter0, int K,
                        int8_t* act_traj_dev, uint8_t* obs_traj_dev, tg_step_out* out_traj_dev) {
  HIPCHK(h, hipSetDevice(h->cfg.device));
  if (h->B % 4 != 0) return fail(h, TG_EINVAL, "tg_rollout: batch must be a multiple of 4 (packed int8 action rows)");
  // Transitions per launch: the bits kernel keeps records, masks and slabs on chip across up to tg_config.fuse (default and
  // max 16) consecutive transitions of a rollout — one launch floor and one slab round trip per group instead of per
  // transition.  tg_step always launches one transition.
  const int fuse_max = h->fuse_max;
  const bool exact = fn_exact_path(h);
  const bool post_kernel = ((h->cfg.obs_mode == TG_OBS_DICT || h->cfg.obs_mode == TG_OBS_RGB) && !h->dict_in_kernel) || exact;
  const size_t obs_stride = (size_t)h->B * obs_bytes_per_env(&h->cfg);
  if (obs_traj_dev && h->cfg.obs_mode == TG_OBS_NONE) return fail(h, TG_EINVAL, "tg_rollout: the handle has no observation (TG_OBS_NONE)");
  // trajectories are written by the step kernel itself when it packs the observation (bits kernel, BOARD / FN);
  // otherwise every transition is one launch followed by device-to-device copies of the engine's buffers
  const bool in_kernel = h->d_cols && !post_kernel;
  const bool sample_in_kernel = sampled && in_kernel && !h->buf[TG_BUF_FINAL_OBS];  // (no sampling + final-observation instantiation)
  const int fuse = (in_kernel && (!sampled || sample_in_kernel)) ? fuse_max : 1;  // (actions drawn by the sampling kernel: one transition per launch)
  auto enqueue = [&]() -> int {  // K transitions, `fuse` per launch (per shard)
    tg::KParams p = step_params(h);
    p.act_i8 = 1;
    for (int k = 0; k < K; k += fuse) {
      if (sample_in_kernel) {
        p.actions = nullptr;
        p.sample_mode = 1;
        p.sample_seed = seed;
        p.sample_counter = counter0 + (uint32_t)k;
        p.act_traj = act_traj_dev ? act_traj_dev + (size_t)k * h->B : nullptr;
      } else if (sampled) {  // other kernels / surfaces: the sampling kernel, then one transition from its int32 actions
        int rcs = join_shards(h);
        if (rcs == TG_OK) rcs = run_grouped(h, false, true, seed, counter0 + (uint32_t)k);
        if (rcs != TG_OK) return rcs;
        if (act_traj_dev)
          hipLaunchKernelGGL(narrow_actions_kernel, dim3((h->B + 255) / 256), dim3(256), 0, h->stream,
                             (const int32_t*)h->buf[TG_BUF_ACTIONS], act_traj_dev + (size_t)k * h->B, h->B);
        p.actions = h->buf[TG_BUF_ACTIONS];
        p.act_i8 = 0;
      } else {
        p.actions = actions_dev + (size_t)k * h->B;
      }
      p.nsteps = K - k < fuse ? K - k : fuse;
      p.ring_cur = h->ring_cur;
      p.hbuf_cur = h->hbuf_cur;
      next_rng_role(h, p);
      if (in_kernel) {
        p.obs_traj = obs_traj_dev ? obs_traj_dev + (size_t)k * obs_stride : nullptr;
        p.out_traj = out_traj_dev ? (uint2*)out_traj_dev + (size_t)k * h->B : nullptr;
        p.obs_traj_stride = obs_stride;
      }
      int rc = launch_step_all(h, p);
      if (rc != TG_OK) return rc;
      after_launch(h, p);
      if (post_kernel) {  // packed by its own kernel after every transition
        rc = launch_observe_dict(h);
        if (rc != TG_OK) return rc;
      }
      if (!in_kernel && (obs_traj_dev || out_traj_dev)) {
        rc = join_shards(h);
        if (rc != TG_OK) return rc;
        if (obs_traj_dev)
          HIPCHK(h, hipMemcpyAsync(obs_traj_dev + (size_t)k * obs_stride, h->buf[TG_BUF_OBS], obs_stride, hipMemcpyDeviceToDevice, h->stream));
        if (out_traj_dev)
          HIPCHK(h, hipMemcpyAsync(out_traj_dev + (size_t)k * h->B, h->buf[TG_BUF_STEP_OUT], (size_t)h->B * sizeof(tg_step_out), hipMemcpyDeviceToDevice, h->stream));
      }
    }
    return TG_OK;
  };
  int rc = enqueue();
  if (rc != TG_OK) return rc;
  HIPCHK(h, hipGetLastError());
  if (exact && h->fn_overlap) return launch_fn_overlap(h);  // (re-evaluated behind the K transitions: see fn_exact_path)
  return TG_OK;
}

int tg_device_ptr(tg_handle* h, int which, void** ptr, size_t* bytes) {
  if (!h || which < 0 || which >= TG_BUF_COUNT) return h ? fail(h, TG_EINVAL, "tg_device_ptr: bad buffer id %d", which) : TG_EINVAL;
  if (which == TG_BUF_RNG) {  // (enqueued on the handle's stream: tg_sync before reading the buffer from elsewhere)
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rcj = join_shards(h);
    if (rcj == TG_OK) rcj = launch_materialize_rng(h);
    if (rcj != TG_OK) return rcj;
  }
  if (which == TG_BUF_HOST_PACK) {
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rcj = join_shards(h);
    if (rcj == TG_OK) rcj = launch_host_pack(h);
    if (rcj != TG_OK) return rcj;
  }
  if (which == TG_BUF_BOARD && h->d_pboard) {  // (packed handles: as for TG_BUF_RNG, the buffer is brought up to date first)
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rcj = join_shards(h);
    if (rcj == TG_OK) rcj = ensure_board(h);
    if (rcj != TG_OK) return rcj;
  }
  if (ptr) *ptr = h->buf[which];
  if (bytes) *bytes = h->bytes[which];
  return TG_OK;
}

int tg_fetch(tg_handle* h, int which, void* host_dst, size_t bytes) {
  if (!h || which < 0 || which >= TG_BUF_COUNT || !host_dst) return h ? fail(h, TG_EINVAL, "tg_fetch: bad arguments") : TG_EINVAL;
  if (bytes > h->bytes[which]) return fail(h, TG_EINVAL, "tg_fetch: %zu bytes requested, buffer %d holds %zu", bytes, which, h->bytes[which]);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  if (which == TG_BUF_RNG) { int rcm = launch_materialize_rng(h); if (rcm != TG_OK) return rcm; }
  if (which == TG_BUF_HOST_PACK) { int rcm = launch_host_pack(h); if (rcm != TG_OK) return rcm; }
  if (which == TG_BUF_BOARD) { int rcm = ensure_board(h); if (rcm != TG_OK) return rcm; }
  if (bytes) HIPCHK(h, hipMemcpyAsync(host_dst, h->buf[which], bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return check_dev_state(h);
}

int tg_fetch_many(tg_handle* h, int m, const int* which, void* const* host_dst, const size_t* bytes, const size_t* src_off) {
  if (!h || m < 0 || (m && (!which || !host_dst || !bytes))) return h ? fail(h, TG_EINVAL, "tg_fetch_many: bad arguments") : TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rc = enqueue_fetches(h, m, which, host_dst, bytes, src_off); if (rc != TG_OK) return rc; }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return check_dev_state(h);
}

int tg_step_fetch_many(int n, tg_handle* const* hs, const void* const* actions, int kind, int m, const int* which,
                       void* const* host_dst, const size_t* bytes, const size_t* src_off) {
  if (n <= 0 || !hs || m < 0 || (m && (!which || !host_dst || !bytes))) return TG_EINVAL;
  for (int k = 0; k < n; ++k)
    if (!hs[k]) return TG_EINVAL;
  // host actions of ALL handles are range-checked before anything is launched: a rejected action leaves no handle stepped
  if (actions && kind == TG_MEM_HOST)
    for (int k = 0; k < n; ++k) {
      if (!actions[k]) return fail(hs[k], TG_EINVAL, "tg_step_fetch_many: null actions for handle %d", k);
      const int rca = check_host_actions(hs[k], (const int32_t*)actions[k]);
      if (rca != TG_OK) return rca;
    }
  // phase 1: enqueue everything — handle 0 on this thread, the others on the pool's workers, side by side
  const std::function<int(int)> enqueue = [&](int k) -> int {
    tg_handle* h = hs[k];
    if (actions) {
      const int rc = step_impl(h, (const int32_t*)actions[k], kind, m > 0);
      if (rc != TG_OK) return rc;
    } else {
      HIPCHK(h, hipSetDevice(h->cfg.device));
    }
    if (m) return enqueue_fetches(h, m, which, host_dst + (size_t)k * m, bytes + (size_t)k * m, src_off ? src_off + (size_t)k * m : nullptr);
    return TG_OK;
  };
  {
    const int rc = EnqueuePool::get().run(n, enqueue);
    if (rc != TG_OK) {  // (a HIP error in one handle's enqueue: nothing of the others may still be in flight into the caller's arrays)
      for (int k = 0; k < n; ++k)
        if (hipSetDevice(hs[k]->cfg.device) == hipSuccess) (void)hipStreamSynchronize(hs[k]->stream);
      return rc;
    }
  }
  for (int k = 0; k < n; ++k) {  // phase 2: one wait per handle
    tg_handle* h = hs[k];
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (m) {
      HIPCHK(h, hipStreamSynchronize(h->stream));
      const int rc = check_dev_state(h);
      if (rc != TG_OK) return rc;
    }
  }
  return TG_OK;
}

int tg_get_state(tg_handle* h, uint8_t* board, tg_env_rec* rec, tg_rng_rec* rng) {
  if (!h) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  const size_t B = (size_t)h->B;
  if (rng) { int rcm = launch_materialize_rng(h); if (rcm != TG_OK) return rcm; }
  if (board) { int rcb = ensure_board(h); if (rcb != TG_OK) return rcb; }
  if (board) HIPCHK(h, hipMemcpyAsync(board, h->buf[TG_BUF_BOARD], B * h->cells, hipMemcpyDeviceToHost, h->stream));
  if (rec) HIPCHK(h, hipMemcpyAsync(rec, h->buf[TG_BUF_REC], B * sizeof(tg_env_rec), hipMemcpyDeviceToHost, h->stream));
  if (rng) HIPCHK(h, hipMemcpyAsync(rng, h->buf[TG_BUF_RNG], B * sizeof(tg_rng_rec), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  { const int rcc = check_dev_state(h); if (rcc != TG_OK) return rcc; }
  if (rec)
    for (size_t i = 0; i < B; ++i) {  // device-private: full-row hint, look-ahead permutation / (NUMPY) the mask of bedrock rows
      rec[i].flags &= (uint8_t)~TG_F_FULLROW;
      rec[i].reserved = 0;
      if (h->cfg.ruleset == TG_RULESET_NUMPY) rec[i].score = 0.0f;
    }
  return TG_OK;
}

int tg_set_state(tg_handle* h, const uint8_t* board, const tg_env_rec* rec, const tg_rng_rec* rng) {
  if (!h) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  if (!rng) { int rcm = launch_materialize_rng(h); if (rcm != TG_OK) return rcm; }  // the rings are rebuilt from TG_BUF_RNG below
  if (!board) { int rcb = ensure_board(h); if (rcb != TG_OK) return rcb; }         // (the boards stay: the nibbles are re-packed from them below)
  const size_t B = (size_t)h->B;
  if (h->d_pboard) {  // packed handles keep cells as nibbles: values the stock game never produces are refused
    if (board)
      for (size_t i = 0; i < B * h->cells; ++i)
        if (board[i] > 15)
          return fail(h, TG_EINVAL, "tg_set_state: board byte %zu is %d; this handle keeps its board as nibbles (cell values 0..15; TG_PACKED=0 "
                                    "creates handles without that limit)", i, board[i]);
    if (rec)
      for (size_t i = 0; i < B; ++i)
        if (rec[i].val > 15 || (rec[i].holder_piece != TG_NONE && rec[i].holder_val > 15))
          return fail(h, TG_EINVAL, "tg_set_state: record %zu holds a cell value > 15; this handle keeps its board as nibbles (TG_PACKED=0 "
                                    "creates handles without that limit)", i);
  }
  if (rec)
    for (size_t i = 0; i < B; ++i) {
      const tg_env_rec& r = rec[i];
      if (r.piece > 6 || r.rot > 3 || (r.holder_piece != TG_NONE && (r.holder_piece > 6 || r.holder_rot > 3)) ||
          r.bag_index > 7 || r.y > h->H + 3 ||
          // (x <= Wp - n: the reach of the reference's own slicing, see sanitize_state_kernel)
          r.x > h->W + 4 + (h->cfg.ruleset == TG_RULESET_NUMPY ? (r.piece == 1 ? 2 : (r.piece == 0 ? 0 : 1)) : 0))
        return fail(h, TG_EINVAL, "tg_set_state: record %zu out of range (piece %d rot %d holder %d bag_index %d x %d y %d)", i,
                    r.piece, r.rot, r.holder_piece, r.bag_index, r.x, r.y);
      for (int k = 0; k < 7; ++k)
        if (r.bag[k] > 6 || (k < h->Q && h->cfg.ruleset == TG_RULESET_NUMPY && r.queue[k] > 6))
          return fail(h, TG_EINVAL, "tg_set_state: record %zu holds a tetromino index > 6", i);
    }
  // The lane kernel looks for completed rows only among the rows a lock touched; a caller-provided board
  // may already hold a full row (the reference would clear it at the next lock, envs/tetris.py:493-494), so
  // such envs are marked for one full scan.
  std::vector<tg_env_rec> tmp;
  if (board) {
    if (h->cfg.ruleset == TG_RULESET_FUNCTIONAL)  // (the {-1, 0, 1} observation tells the drawn piece by its top bit: tg_kernel_bits.h bits_bin1)
      for (size_t i = 0; i < B * h->cells; ++i)
        if (board[i] >= 0x80)
          return fail(h, TG_EINVAL, "tg_set_state: board byte %zu is %d; cells of the functional rule-set are 0 or a tetromino value 2..8", i, board[i]);
    // (every refusal in front of the first upload: a refused call leaves the engine as it was)
    std::vector<uint8_t> any_full(B, 0);
    for (size_t i = 0; i < B; ++i) {
      const uint8_t* b = board + i * h->cells;
      for (int r = 0; r < h->H; ++r) {
        bool full = true, ones = true;
        for (int c = 0; c < h->W; ++c) { full &= b[r * h->W + c] != 0; ones &= b[r * h->W + c] == 1; }
        any_full[i] |= full ? 1 : 0;
        // (the reference never clears a row of bedrock cells, envs/tetris.py:493 `& ~(board == 1).all(axis=1)`.  The default kernel keeps
        // such rows as a mask in the record — bits_prepare_kernel / step_kernel_bits `brow` —; the earlier step kernels of the test-only
        // library test rows on bytes and would clear it with the next lock: refused there rather than stepped differently)
        if (ones && h->cfg.ruleset == TG_RULESET_NUMPY && !h->d_cols)
          return fail(h, TG_EUNSUPPORTED, "tg_set_state: row %d of board %zu consists of bedrock cells (value 1) only; the reference never clears such "
                                          "a row (envs/tetris.py:493): the earlier step kernels (TG_KERNEL) do not model that", r, i);
        // (... and one inside the four spawn rows is refused by the default kernel too: the grouped wrapper's legal mask — a drop from
        // y = 0 that ends on a value-1 cell, wrappers/grouped.py:147-166 — would depend on the reference's drop loop tunnelling through the
        // row from a colliding start, which the closed-form mask of every kernel here does not model)
        if (ones && r < 4 && h->cfg.ruleset == TG_RULESET_NUMPY)
          return fail(h, TG_EUNSUPPORTED, "tg_set_state: row %d of board %zu, one of the four spawn rows, consists of bedrock cells (value 1) only: "
                                          "not supported (bedrock rows from row 4 down are)", r, i);
      }
    }
    HIPCHK(h, hipMemcpyAsync(h->buf[TG_BUF_BOARD], board, B * h->cells, hipMemcpyHostToDevice, h->stream));
    tmp.resize(B);
    if (rec) std::memcpy(tmp.data(), rec, B * sizeof(tg_env_rec));
    else {
      HIPCHK(h, hipMemcpyAsync(tmp.data(), h->buf[TG_BUF_REC], B * sizeof(tg_env_rec), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    for (size_t i = 0; i < B; ++i)
      tmp[i].flags = (uint8_t)((tmp[i].flags & ~TG_F_FULLROW) | (any_full[i] ? TG_F_FULLROW : 0));
    rec = tmp.data();
  }
  if (rec) HIPCHK(h, hipMemcpyAsync(h->buf[TG_BUF_REC], rec, B * sizeof(tg_env_rec), hipMemcpyHostToDevice, h->stream));
  if (rng) {
    HIPCHK(h, hipMemcpyAsync(h->buf[TG_BUF_RNG], rng, B * sizeof(tg_rng_rec), hipMemcpyHostToDevice, h->stream));
    h->rng_valid.assign(B, 1);
  }
  { int rc = launch_fn_overlap(h); if (rc != TG_OK) return rc; }
  {
    int rc = launch_prepare_bits(h, nullptr, 1);
    if (rc != TG_OK) return rc;
  }
  // TG_BUF_OBS shows the uploaded state (as after tg_set_state_dev): re-packed for every surface — the DICT / RGB surfaces are only
  // kept up to date incrementally by the transitions (holder / queue strips are stored only when they change), and a BOARD / FN
  // observation left as the last transition wrote it would show a board that is gone
  if (h->cfg.obs_mode != TG_OBS_NONE) {
    int rc = launch_observe_dict(h);
    if (rc != TG_OK) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return check_dev_state(h);  // (reads fn_overlap_kernel's answer)
}

int tg_get_state_dev(tg_handle* h, uint8_t* board_dev, tg_env_rec* rec_dev, tg_rng_rec* rng_dev) {
  if (!h) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  const size_t B = (size_t)h->B;
  if (rng_dev) { int rcm = launch_materialize_rng(h); if (rcm != TG_OK) return rcm; }
  if (board_dev) { int rcb = ensure_board(h); if (rcb != TG_OK) return rcb; }
  if (board_dev) HIPCHK(h, hipMemcpyAsync(board_dev, h->buf[TG_BUF_BOARD], B * h->cells, hipMemcpyDeviceToDevice, h->stream));
  if (rec_dev) HIPCHK(h, hipMemcpyAsync(rec_dev, h->buf[TG_BUF_REC], B * sizeof(tg_env_rec), hipMemcpyDeviceToDevice, h->stream));
  if (rng_dev) HIPCHK(h, hipMemcpyAsync(rng_dev, h->buf[TG_BUF_RNG], B * sizeof(tg_rng_rec), hipMemcpyDeviceToDevice, h->stream));
  return TG_OK;
}

int tg_set_state_dev(tg_handle* h, const uint8_t* board_dev, const tg_env_rec* rec_dev, const tg_rng_rec* rng_dev) {
  if (!h) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  if (!rng_dev) { int rcm = launch_materialize_rng(h); if (rcm != TG_OK) return rcm; }  // the rings are rebuilt from TG_BUF_RNG below
  if (!board_dev) { int rcb = ensure_board(h); if (rcb != TG_OK) return rcb; }
  const size_t B = (size_t)h->B;
  if (board_dev) HIPCHK(h, hipMemcpyAsync(h->buf[TG_BUF_BOARD], board_dev, B * h->cells, hipMemcpyDeviceToDevice, h->stream));
  if (rec_dev) HIPCHK(h, hipMemcpyAsync(h->buf[TG_BUF_REC], rec_dev, B * sizeof(tg_env_rec), hipMemcpyDeviceToDevice, h->stream));
  if (rng_dev) {
    HIPCHK(h, hipMemcpyAsync(h->buf[TG_BUF_RNG], rng_dev, B * sizeof(tg_rng_rec), hipMemcpyDeviceToDevice, h->stream));
    h->rng_valid.assign(B, 1);
  }
  if (rec_dev || board_dev) {  // (range checks on the device: see sanitize_state_kernel)
    SanitizeParams q{};
    q.rec = (uint32_t*)h->buf[TG_BUF_REC];
    q.board = board_dev ? (uint8_t*)h->buf[TG_BUF_BOARD] : nullptr;
    q.err = h->d_err;
    q.batch = h->B; q.W = h->W; q.H = h->H; q.Q = h->Q; q.ruleset = h->cfg.ruleset; q.packed = h->d_pboard ? 1 : 0;
    hipLaunchKernelGGL(sanitize_state_kernel, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, q);
    HIPCHK(h, hipGetLastError());
    h->dev_state_pending = true;
  }
  { int rc = launch_fn_overlap(h); if (rc != TG_OK) return rc; }
  { int rc = launch_prepare_bits(h, nullptr, 1); if (rc != TG_OK) return rc; }
  if (h->cfg.obs_mode != TG_OBS_NONE) return launch_observe_dict(h);  // (the observation of the restored state)
  return TG_OK;
}

int tg_observe(tg_handle* h) {
  if (!h) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  if (h->cfg.obs_mode == TG_OBS_NONE) return TG_OK;
  { int rcb = ensure_board(h); if (rcb != TG_OK) return rcb; }
  return launch_observe_dict(h);
}

static int run_grouped(tg_handle* h, bool mask, bool sample, uint32_t seed, uint32_t counter) {
  tg::GroupedParams g{};
  g.rec = (const uint32_t*)h->buf[TG_BUF_REC];
  g.mask = mask ? (uint8_t*)h->buf[TG_BUF_LEGAL_MASK] : nullptr;
  g.actions = sample ? (int32_t*)h->buf[TG_BUF_ACTIONS] : nullptr;
  const tg::KParams kp = step_params(h);
  for (int i = 0; i < 7; ++i) g.shapes[i] = kp.shapes[i];
  g.seed = seed; g.counter = counter; g.batch = h->B; g.width = h->W;
  hipLaunchKernelGGL(tg::grouped_kernel, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, g);
  HIPCHK(h, hipGetLastError());
  return TG_OK;
}

}  // extern "C"
namespace { int run_grouped_mask(tg_handle* h) { return run_grouped(h, true, false, 0, 0); } }
extern "C" {

int tg_observe_as(tg_handle* h, int obs_mode, void* obs_dev, size_t bytes) {
  if (!h) return TG_EINVAL;
  if (!obs_dev) return fail(h, TG_EINVAL, "tg_observe_as: null buffer");
  if (obs_mode != TG_OBS_BOARD && obs_mode != TG_OBS_FN && obs_mode != TG_OBS_DICT && obs_mode != TG_OBS_RGB)
    return fail(h, TG_EINVAL, "tg_observe_as: unknown obs_mode %d", obs_mode);
  if ((obs_mode == TG_OBS_DICT || obs_mode == TG_OBS_RGB) && h->cfg.ruleset != TG_RULESET_NUMPY)
    return fail(h, TG_EINVAL, "tg_observe_as: TG_OBS_DICT / TG_OBS_RGB are observations of the NUMPY rule-set");
  const size_t need = (size_t)h->B * obs_bytes_per_env(&h->cfg, obs_mode);
  if (bytes < need) return fail(h, TG_EINVAL, "tg_observe_as: %zu bytes given, this surface takes %zu", bytes, need);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  { int rcb = ensure_board(h); if (rcb != TG_OK) return rcb; }
  tg::ObserveParams o{};
  o.board = (const uint8_t*)h->buf[TG_BUF_BOARD];
  o.rec = (const uint32_t*)h->buf[TG_BUF_REC];
  o.obs = (uint8_t*)obs_dev;
  o.batch = h->B;
  o.obs_mode = obs_mode;
  o.queue_size = h->Q;
  o.ruleset = h->cfg.ruleset;
  h->L.observe(o, h->stream);
  HIPCHK(h, hipGetLastError());
  return TG_OK;
}

int tg_features_as(tg_handle* h, uint8_t* placement_feats_dev, uint8_t* feats_dev) {
  if (!h) return TG_EINVAL;
  if (h->cfg.ruleset != TG_RULESET_NUMPY) return fail(h, TG_EINVAL, "tg_features_as: NUMPY rule-set only (wrappers/observation.py wraps envs/tetris.py)");
  if (!placement_feats_dev && !feats_dev) return TG_OK;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  { int rcb = ensure_board(h); if (rcb != TG_OK) return rcb; }
  tg::FeatureParams f{};
  f.board = (const uint8_t*)h->buf[TG_BUF_BOARD];
  f.rec = (const uint32_t*)h->buf[TG_BUF_REC];
  f.placement_feats = placement_feats_dev;
  f.feats = feats_dev;
  f.batch = h->B;
  h->L.features(f, h->stream);
  HIPCHK(h, hipGetLastError());
  return TG_OK;
}

int tg_grouped_enumerate(tg_handle* h, uint8_t* boards_dev) {
  if (!h) return TG_EINVAL;
  if (h->cfg.action_set != TG_ACTIONS_GROUPED) return fail(h, TG_EINVAL, "tg_grouped_enumerate: handle was not created with TG_ACTIONS_GROUPED");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  int rc = run_grouped(h, true, false, 0, 0);
  if (rc != TG_OK || !boards_dev) return rc;
  rc = ensure_board(h);
  if (rc != TG_OK) return rc;
  tg::PlacementParams q{};
  q.board = (const uint8_t*)h->buf[TG_BUF_BOARD];
  q.rec = (const uint32_t*)h->buf[TG_BUF_REC];
  q.boards = boards_dev;
  q.batch = h->B;
  h->L.placements(q, h->stream);
  HIPCHK(h, hipGetLastError());
  return TG_OK;
}

int tg_grouped_sample(tg_handle* h, uint32_t seed, uint32_t counter) {
  if (!h) return TG_EINVAL;
  if (h->cfg.action_set != TG_ACTIONS_GROUPED) return fail(h, TG_EINVAL, "tg_grouped_sample: handle was not created with TG_ACTIONS_GROUPED");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  return run_grouped(h, false, true, seed, counter);
}

int tg_features(tg_handle* h) {
  if (!h) return TG_EINVAL;
  if (!(h->cfg.flags & TG_FLAG_FEATURES)) return fail(h, TG_EINVAL, "tg_features: handle was created without TG_FLAG_FEATURES");
  if (h->cfg.ruleset != TG_RULESET_NUMPY) return fail(h, TG_EINVAL, "tg_features: NUMPY rule-set only (wrappers/observation.py wraps envs/tetris.py)");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  { int rcb = ensure_board(h); if (rcb != TG_OK) return rcb; }
  tg::FeatureParams f{};
  f.board = (const uint8_t*)h->buf[TG_BUF_BOARD];
  f.rec = (const uint32_t*)h->buf[TG_BUF_REC];
  f.placement_feats = (uint8_t*)h->buf[TG_BUF_PLACEMENT_FEATS];
  f.feats = (uint8_t*)h->buf[TG_BUF_FEATS];
  f.batch = h->B;
  h->L.features(f, h->stream);
  HIPCHK(h, hipGetLastError());
  return TG_OK;
}

int tg_sync(tg_handle* h) {
  if (!h) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  // short waits are polled (a blocking hipStreamSynchronize costs tens of microseconds of wake-up latency, which is
  // several step launches); after 2 ms of polling fall back to the blocking wait.  (Rounds 2-5 polled a fixed 4 000 queries
  // = ~140 us: a 20-launch region — the driver's bench command — ends right at that boundary, and the hand-over to the blocking
  // wait put a tail of +0.5 ... +1.3 us per launch on one region in ten: profiles/r06/region20_sync_poll.log)
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    for (int i = 0; i < 256; ++i) {
      const hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) return check_dev_state(h);
      if (q != hipErrorNotReady) return fail(h, TG_EHIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
    }
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return check_dev_state(h);
}

int tg_step_kernel_name(tg_handle* h, int transitions_per_launch, int trajectory, int sampled, char* buf, size_t n) {
  if (!h || !buf || n == 0) return TG_EINVAL;
  tg::KParams p = step_params(h);
  p.nsteps = transitions_per_launch > 1 ? transitions_per_launch : 1;
  p.sample_mode = sampled ? 1 : 0;
  if (trajectory) p.out_traj = (uint2*)h->buf[TG_BUF_STEP_OUT];  // (only its non-nullness selects the instantiation)
  h->L.step_name(p, buf, n);
  return TG_OK;
}

int tg_timer_start(tg_handle* h) {
  if (!h) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  return TG_OK;
}

int tg_timer_stop(tg_handle* h, float* elapsed_ms) {
  if (!h || !elapsed_ms) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  for (int i = 0; i < 4000; ++i) {  // short waits are polled (see tg_sync)
    const hipError_t q = hipEventQuery(h->ev1);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) return fail(h, TG_EHIP, "hipEventQuery failed: %s", hipGetErrorString(q));
  }
  HIPCHK(h, hipEventSynchronize(h->ev1));
  HIPCHK(h, hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
  return TG_OK;
}

int tg_dev_alloc(tg_handle* h, size_t bytes, void** ptr) {
  if (!h || !ptr) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipMalloc(ptr, bytes));
  return TG_OK;
}

int tg_host_alloc(tg_handle* h, size_t bytes, void** ptr) {
  if (!h || !ptr) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
  return TG_OK;
}

int tg_host_free(tg_handle* h, void* ptr) {
  if (!h) return hipHostFree(ptr) == hipSuccess ? TG_OK : TG_EHIP;  // (a block that outlived its engine: nothing to wait for)
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipHostFree(ptr));
  return TG_OK;
}

int tg_dev_free(tg_handle* h, void* ptr) {
  if (!h) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipFree(ptr));
  return TG_OK;
}

int tg_dev_upload(tg_handle* h, void* dst_dev, const void* src_host, size_t bytes) {
  if (!h || !dst_dev || !src_host) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  HIPCHK(h, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return TG_OK;
}

int tg_dev_download(tg_handle* h, void* dst_host, const void* src_dev, size_t bytes) {
  if (!h || !dst_host || !src_dev) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  HIPCHK(h, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return TG_OK;
}

int tg_dev_copy(tg_handle* h, void* dst_dev, const void* src_dev, size_t bytes) {
  if (!h || !dst_dev || !src_dev) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  if (bytes) HIPCHK(h, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, h->stream));
  return TG_OK;
}

int tg_stream_copy_gbps(tg_handle* h, size_t bytes, int iters, float* gbps) {
  if (!h || !gbps || iters <= 0) return TG_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  bytes &= ~(size_t)15;
  void *a = nullptr, *b = nullptr;
  HIPCHK(h, hipMalloc(&a, bytes));
  if (hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); return fail(h, TG_ENOMEM, "tg_stream_copy_gbps: hipMalloc failed"); }
  (void)hipMemsetAsync(a, 1, bytes, h->stream);
  const size_t n = bytes / 16;
  // the ceiling is the best of a few launch shapes (which one wins depends on the size: tools/ubench/copy_ceiling.hip)
  float best = 0.f;
  hipError_t e = hipSuccess;
  auto timed = [&](auto kern, int grid) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, h->stream, (const tg::copy_vec*)a, (tg::copy_vec*)b, n);
    (void)hipEventRecord(h->ev0, h->stream);
    for (int i = 0; i < iters; ++i)
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, h->stream, (const tg::copy_vec*)a, (tg::copy_vec*)b, n);
    (void)hipEventRecord(h->ev1, h->stream);
    hipError_t er = hipEventSynchronize(h->ev1);
    float ms = 0.f;
    if (er == hipSuccess) er = hipEventElapsedTime(&ms, h->ev0, h->ev1);
    if (er != hipSuccess) { e = er; return; }
    const float g = (float)(2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9);
    if (g > best) best = g;
  };
  timed(tg::stream_copy_kernel<4, false>, 2048);
  timed(tg::stream_copy_kernel<4, true>, 2048);
  timed(tg::stream_copy_kernel<8, true>, 2048);
  timed(tg::stream_copy_kernel<4, true>, 8192);
  timed(tg::stream_copy_kernel<4, true, true>, 16384);
  timed(tg::stream_copy_kernel<4, true, true>, 65536);
  (void)hipFree(a);
  (void)hipFree(b);
  if (e != hipSuccess) return fail(h, TG_EHIP, "tg_stream_copy_gbps: %s", hipGetErrorString(e));
  *gbps = best;
  return TG_OK;
}

// phase-timing buffer of the lane kernel (TG_PHASE_TIMING builds): u64 [waves][16] s_memtime stamps
int tg_debug_phase_times(tg_handle* h, unsigned long long* out, size_t n_waves) {
  if (!h || !h->dbg) return TG_EUNSUPPORTED;
  { int rcj = join_shards(h); if (rcj != TG_OK) return rcj; }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, h->dbg, n_waves * 16 * 8, hipMemcpyDeviceToHost));
  return TG_OK;
}

// host-side view of the shape table (tests compare it with the oracle's matrices)
int tg_shape_mask(int piece, int rot) {
  if (piece < 0 || piece > 6 || rot < 0 || rot > 3) return TG_EINVAL;
  static constexpr tg::ShapeTable t = tg::make_shape_table();
  return t.mask[piece][rot];
}

}  // extern "C"
