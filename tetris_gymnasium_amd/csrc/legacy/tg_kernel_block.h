// tg_kernel_block.h — third formulation of the step kernel: two-phase, workgroup-level compaction of the slow path.
//
// Measured on the lane kernels (tools/path_cost.py, profiles/r01): a 65 536-env step costs 7.0 µs when no env locks
// (loads + collision tests + observation pack) and 18.5 µs when EVERY env locks — and 16.9 µs for the real workload
// in which only ≈26 % of the envs need the slow path (lock, hard drop, effective swap, reset), because with 16 envs
// per wave almost every wave has at least one such env and then executes the whole slow path at ≈20 % lane
// utilisation.  Here a workgroup of NW waves owns 16·NW envs whose slabs all sit in LDS:
//   phase 1  every quad runs the branch-free fast path for its own env; envs that need more are appended to a
//            workgroup worklist (one LDS atomic per wave, ranks from a ballot);
//   phase 2  the worklist is processed DENSELY: quad i takes worklist entry i and runs the full rule logic
//            (lanes_logic(), the same code as the lane kernel) on that env's slab — ⌈26 % · 16·NW / 16⌉ waves' worth
//            of slow-path work instead of NW;
//   phase 3  the workgroup writes back changed slabs, stamps the active pieces and streams observations, records and
//            step outputs to HBM with fully coalesced 16-byte stores.
#pragma once
#include "tg_kernel_lanes.h"

namespace tg {

template <int W, int H, int NW>
struct BlockGeo {
  static constexpr int L = 4, EPW = 16, EPB = EPW * NW, NT = 64 * NW, S = W * H;
  static constexpr int SLABS = ((EPB * S + 15) / 16) * 16;
  static constexpr int NCH = SLABS / 16;
  static constexpr int NIT = (NCH + NT - 1) / NT;
  static constexpr int OFF_TAB = SLABS;                                   // 64 B cell table
  static constexpr int OFF_PAD = OFF_TAB + 64;                            // drop-probe overrun
  static constexpr int OFF_REC = OFF_PAD + ((4 * W + 16 + 15) / 16) * 16;  // uint4[2*EPB]: records (in, then out)
  static constexpr int OFF_ACT = OFF_REC + EPB * 32;                      // int[EPB]
  static constexpr int OFF_OUT = OFF_ACT + EPB * 4;                       // uint2[EPB] step outputs
  static constexpr int OFF_STP = OFF_OUT + EPB * 8;                       // uint2[EPB] stamp info
  static constexpr int OFF_DRT = OFF_STP + EPB * 8;                       // u8[EPB] slab changed
  static constexpr int OFF_WRK = OFF_DRT + EPB;                           // u8[EPB] worklist
  static constexpr int OFF_CNT = OFF_WRK + EPB;                           // int worklist length (+ pad)
  static constexpr int BYTES = OFF_CNT + 16;
};

template <int W, int H, int RULESET, int NW>
__global__ __launch_bounds__(64 * NW) void step_kernel_block(const KParams p) {
  using BG = BlockGeo<W, H, NW>;
  constexpr int L = BG::L, EPB = BG::EPB, NT = BG::NT, S = BG::S, NIT = BG::NIT;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* slabs = lds;
  uint4* lds4 = reinterpret_cast<uint4*>(lds);
  uint64_t* cell_tab = reinterpret_cast<uint64_t*>(lds + BG::OFF_TAB);
  uint4* recs = reinterpret_cast<uint4*>(lds + BG::OFF_REC);
  int* acts = reinterpret_cast<int*>(lds + BG::OFF_ACT);
  uint2* outs = reinterpret_cast<uint2*>(lds + BG::OFF_OUT);
  uint2* stamps = reinterpret_cast<uint2*>(lds + BG::OFF_STP);
  uint8_t* dirty = lds + BG::OFF_DRT;
  uint8_t* work = lds + BG::OFF_WRK;
  int* wcount = reinterpret_cast<int*>(lds + BG::OFF_CNT);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = rfl((int)(threadIdx.x >> 6));
  const int sub = lane & 3;
  const bool leader = sub == 0;
  const uint32_t B = (uint32_t)p.batch;
  const uint32_t base_env = blockIdx.x * (uint32_t)EPB;
  const uint32_t nlive = B - base_env < (uint32_t)EPB ? B - base_env : (uint32_t)EPB;
  const uint32_t nbytes = nlive * (uint32_t)S;
  const uint32_t nchunks = (nbytes + 15u) / 16u;

#ifdef TG_PHASE_TIMING
#define TG_BSTAMP(i) do { if (p.dbg && tid == 0) p.dbg[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TG_BSTAMP(i) do {} while (0)
#endif
  TG_BSTAMP(10);
  // ---- phase 0: the workgroup's slabs (one contiguous block), records and actions → LDS, all coalesced ----
  {
    const uint4* gsrc = reinterpret_cast<const uint4*>(p.board + (size_t)base_env * S);
    const uint4* grec = reinterpret_cast<const uint4*>(p.rec) + (size_t)base_env * 2;
    uint4 stage[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const uint32_t c = (uint32_t)tid + (uint32_t)NT * k;
      stage[k] = gsrc[c < nchunks ? c : nchunks - 1u];
    }
    uint4 rv = make_uint4(0, 0, 0, 0);
    if ((uint32_t)tid < 2u * nlive) rv = grec[tid];
    int av = 0;
    if ((uint32_t)tid < nlive)
      av = p.act_i8 ? (int)reinterpret_cast<const int8_t*>(p.actions)[base_env + tid]
                    : reinterpret_cast<const int32_t*>(p.actions)[base_env + tid];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const uint32_t c = (uint32_t)tid + (uint32_t)NT * k;
      if (c < nchunks) lds4[c] = stage[k];
    }
    if (tid < 2 * EPB) recs[tid] = rv;
    if (tid < EPB) { acts[tid] = av; dirty[tid] = 0; }
    if (tid < 7) cell_tab[tid] = c_cells.cells[tid];
    if (tid == 0) *wcount = 0;
  }
  __syncthreads();
  TG_BSTAMP(11);

  // ---- phase 1: branch-free fast path for this quad's own env ----
  const int eloc = wave * 16 + (lane >> 2);
  const bool live = (uint32_t)eloc < nlive;
  LaneBoard<W, H, L> bd;
  bd.my = slabs + eloc * S;
  bd.sub = sub;
  {
    LaneS s;
    lane_unpack(s, recs[2 * eloc], recs[2 * eloc + 1]);
    const int action = acts[eloc];
    uint32_t e;
    if (RULESET == TG_RULESET_NUMPY) {
      e = (uint32_t)(kActLutNumpy >> (8 * (action & 7))) & 0xFFu;
    } else {
      const int br = action < 0 ? 0 : (action > 6 ? 6 : action);
      e = (uint32_t)(kActLutFn >> (8 * br)) & 0xFFu;
    }
    const bool swap_eff = (e & 0x80u) && !(s.flags & TG_F_HAS_SWAPPED);
    const bool slow0 = (s.flags & (TG_F_GAME_OVER | TG_F_PENDING_RESET)) || (e & 0x40u) || swap_eff ||
                       p.action_set != TG_ACTIONS_BASE || p.stagger == -2;
    const int tx = s.x + (int)(e & 3u) - 1, ty = s.y + (int)((e >> 2) & 1u), tr = (s.rot + (int)((e >> 3) & 3u)) & 3;
    const uint32_t cellsC = lane_cells(cell_tab, s.piece, s.rot), cellsT = lane_cells(cell_tab, s.piece, tr);
    const int xa[2] = {tx, tx}, ya[2] = {ty, ty + 1}, xb[2] = {s.x, s.x}, yb[2] = {s.y, s.y + 1};
    const uint32_t ha = bd.template collide_n<2>(cellsT, xa, ya);
    const uint32_t hb = bd.template collide_n<2>(cellsC, xb, yb);
    const bool accept = (e & 0x20u) && !(ha & 1u);
    const uint32_t w = accept ? ha : hb;
    const bool blocked = p.gravity && (w & 2u);
    const bool slow = live && (slow0 || blocked);
    if (live && !slow && leader) {
      const int nx = accept ? tx : s.x, nr = accept ? tr : s.rot;
      const int ny = (accept ? ty : s.y) + (p.gravity ? 1 : 0);
      const bool coll = !p.gravity && (w & 1u);  // after a gravity move the piece is free
      const bool show = RULESET == TG_RULESET_NUMPY ? !coll : true;
      float reward = 0.0f;
      uint4 ra = recs[2 * eloc];
      ra.x = (uint32_t)s.piece | ((uint32_t)nr << 8) | ((uint32_t)nx << 16) | ((uint32_t)ny << 24);
      recs[2 * eloc] = ra;
      if (RULESET == TG_RULESET_FUNCTIONAL && accept && (e & 4u)) {  // soft drop +1 (tetris_fn.py:183-190)
        const float old = s.score;
        const float t = (old + 1.0f) + 0.0f;
        uint4 rb = recs[2 * eloc + 1];
        rb.z = __float_as_uint(t);
        recs[2 * eloc + 1] = rb;
        reward = t - old;
      }
      outs[eloc] = make_uint2(__float_as_uint(reward), 0u);
      stamps[eloc] = make_uint2((accept ? cellsT : cellsC) | ((uint32_t)s.val << 16),
                                (uint32_t)nx | ((uint32_t)ny << 8) | ((show ? 1u : 0u) << 16));
    }
    const uint64_t m = __ballot(slow && leader);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(wcount, (int)__popcll(m));
      base = rfl(base);
      if (slow && leader) work[base + (int)__popcll(m & ((1ULL << lane) - 1ULL))] = (uint8_t)eloc;
    }
  }
  __syncthreads();
  TG_BSTAMP(12);

  // ---- phase 2: the slow envs, densely packed: quad i ← worklist entry i ----
  {
    const int n = *wcount;
    for (int i0 = wave * 16; i0 < n; i0 += NW * 16) {
      const int i = i0 + (lane >> 2);
      const bool live2 = i < n;
      const int e2 = work[live2 ? i : n - 1];  // idle quads shadow the last entry (their results are discarded)
      const uint32_t env = base_env + (uint32_t)e2;
      LaneS s;
      lane_unpack(s, recs[2 * e2], recs[2 * e2 + 1]);
      LaneRng r;
      r.s_lo = r.s_hi = r.i_lo = r.i_hi = 0;
      r.uinteger = r.has32 = r.k0 = r.k1 = 0;
      r.src = p.rng + env;
      r.loaded = false;
      r.dirty = false;
      LaneBoard<W, H, L> b2;
      b2.my = slabs + e2 * S;
      b2.sub = sub;
      LaneOut lo;
      lanes_logic<W, H, RULESET, L>(p, s, r, b2, slabs, cell_tab, acts[e2], live2, e2 * S, lane, lo);
      if (RULESET == TG_RULESET_NUMPY) lane_resolve_shuffle(s, r, live2 && s.bidx >= 7);  // pending reshuffle
      if (live2 && leader) {
        uint4 oa, ob;
        lane_pack(s, oa, ob);
        recs[2 * e2] = oa;
        recs[2 * e2 + 1] = ob;
        outs[e2] = make_uint2(__float_as_uint(lo.reward), (uint32_t)lo.terminated | ((uint32_t)lo.lines << 8) |
                                                            ((lo.pending ? 1u : 0u) << 16));
        const bool show = RULESET == TG_RULESET_NUMPY ? !lo.final_coll : !(s.flags & TG_F_GAME_OVER);
        stamps[e2] = make_uint2(lane_cells(cell_tab, s.piece, s.rot) | ((uint32_t)s.val << 16),
                                (uint32_t)s.x | ((uint32_t)s.y << 8) | ((show ? 1u : 0u) << 16));
        dirty[e2] = lo.board_dirty ? 1 : 0;
        if (r.dirty) {
          uint4* rng_out = reinterpret_cast<uint4*>(p.rng + env);
          rng_out[0] = make_uint4((uint32_t)r.s_lo, (uint32_t)(r.s_lo >> 32), (uint32_t)r.s_hi, (uint32_t)(r.s_hi >> 32));
          rng_out[2] = make_uint4(r.uinteger, r.has32, r.k0, r.k1);
        }
      }
    }
  }
  __syncthreads();
  TG_BSTAMP(13);

  // ---- phase 3: changed slabs → board, stamp the pieces, observations / records / outputs → HBM ----
  {
    uint4* gdst = reinterpret_cast<uint4*>(p.board + (size_t)base_env * S);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const uint32_t c = (uint32_t)tid + (uint32_t)NT * k;
      if (c < nchunks) {
        const uint32_t e0 = (c * 16u) / (uint32_t)S, e1 = (c * 16u + 15u) / (uint32_t)S;
        if (dirty[e0] | dirty[e1 < (uint32_t)EPB ? e1 : (uint32_t)EPB - 1u]) gdst[c] = lds4[c];
      }
    }
  }
  const bool obs_on = p.obs_mode == TG_OBS_BOARD || p.obs_mode == TG_OBS_FN;
  if (obs_on) {
    __syncthreads();
    if (live) {
      const uint2 st = stamps[eloc];
      bd.stamp(st.x & 0xFFFFu, (int)(st.y & 0xFFu), (int)((st.y >> 8) & 0xFFu),
               p.obs_mode == TG_OBS_FN ? 0xFFu : ((st.x >> 16) & 0xFFu), false, ((st.y >> 16) & 1u) != 0);
    }
    __syncthreads();
    uint4* odst = reinterpret_cast<uint4*>(p.obs + (size_t)base_env * S);
    const bool fn = p.obs_mode == TG_OBS_FN;
    auto bin1 = [](uint32_t x) -> uint32_t {  // bytes: 0xFF (active) stays -1, other non-zero → 1  (tetris_fn.py:137-158)
      const uint32_t nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
      const uint32_t ff = ((x & 0x7F7F7F7Fu) + 0x01010101u) & x & 0x80808080u;
      return (nz >> 7) | ((ff >> 7) * 0xFFu);
    };
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const uint32_t c = (uint32_t)tid + (uint32_t)NT * k;
      if (c < nchunks) {
        uint4 v = lds4[c];
        if (fn) v = make_uint4(bin1(v.x), bin1(v.y), bin1(v.z), bin1(v.w));
        if (c * 16u + 16u <= nbytes) {
          odst[c] = v;
        } else {  // ragged tail of a partial workgroup
          for (uint32_t i = 0; c * 16u + i < nbytes; ++i) {
            const uint32_t w4 = i < 4 ? v.x : (i < 8 ? v.y : (i < 12 ? v.z : v.w));
            p.obs[(size_t)base_env * S + c * 16u + i] = (uint8_t)(w4 >> (8 * (i & 3)));
          }
        }
      }
    }
  }
  if ((uint32_t)tid < 2u * nlive) (reinterpret_cast<uint4*>(p.rec) + (size_t)base_env * 2)[tid] = recs[tid];
  if ((uint32_t)tid < nlive) {
    const uint2 o = outs[tid];
    p.out[base_env + tid] = make_uint2(o.x, o.y & 0xFFFFu);
    if (p.episode) {
      tg_episode_rec ep = p.episode[base_env + tid];
      const float reward = __uint_as_float(o.x);
      if ((o.y >> 16) & 1u) {  // the NEXT_STEP reset step belongs to no episode
        ep.ret = 0.0f; ep.len = 0;
      } else {
        ep.ret += reward;
        ep.len += 1;
        if (o.y & 0xFFu) { ep.last_ret = ep.ret; ep.last_len = ep.len; ep.ret = 0.0f; ep.len = 0; }
      }
      p.episode[base_env + tid] = ep;
    }
  }
  TG_BSTAMP(14);
}

}  // namespace tg
