// tile_gemv.h -- the multi-task GEMV launch for Q2_K weights in the tiled layout (tile_device.h): same launch descriptor,
// activation groups, staging prologue and epilogues as gemv_body (kernels_gemv.hip), the row products on the matrix pipe.
// A device function, so that the router launch's rider and the MLA second-stage launch with the cache write can run it too.
//
// A workgroup owns whole 16-row strips of its activation group's (padded) row space.  It works in ROUNDS of as many strips as
// the partials' LDS region holds (one round for every DeepSeek shape): the round's items (strip x 4-block item, or strip x
// block for rows of <= 8 blocks) are dealt to the waves as contiguous ranges - a wave streams one contiguous byte range of
// tiles -, every item leaves its partial in LDS, and after a barrier one WAVE per strip adds the strip's partials in the fixed
// order of tile_device.h and runs the epilogue for its 16 rows.  GLU launches treat w1 and w3 as two strips of the same rows.
#pragma once
#include "tile_device.h"
#define TILE_MAX_ROUND_STRIPS 256

template <bool GLU, int NW>
DEV void gemv_tile_body(const GemvLaunch* __restrict__ Lp, const void* h_a0, const void* h_a1, const void* h_a2, int h_n, int h_mode,
                        float h_eps, int h_gwgs, int h_gstride, const int bid, const float h_pre_scale) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float scratch[16];
  __shared__ bool comb_last;
  const GemvLaunch& L = *Lp;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned long long t_entry = wall_clock64();
  const int n_groups = L.n_groups;
  const int t_act = L.t_act;
  const bool hinted = h_n > 0;  // one activation group: its source came with the kernel arguments (ActSrc)
  if (hinted) {
    ActSrc S;
    S.act_mode = h_mode; S.n = h_n; S.eps = h_eps; S.pre_scale = h_pre_scale;
    S.a_qs = static_cast<const int8_t*>(h_a0); S.a_d = static_cast<const float*>(h_a1); S.a_bsums = static_cast<const int16_t*>(h_a2);
    S.a_f32 = static_cast<const float*>(h_a0); S.norm_w = static_cast<const float*>(h_a1);
    if (h_gwgs > 0) S.a_f32 += (size_t)(bid / h_gwgs) * h_gstride;  // equal groups, equally spaced f32 vectors
    stage_q8<LAY_TILE, NW>(S, smem, tid, scratch);
  }
  unsigned long long* tl = L.timeline && bid < DSK_TL_WGS ? L.timeline + (size_t)bid * 8 : nullptr;
  if (tl && tid == 0) tl[0] = t_entry;

  int t0 = 0, t1 = 1, wi, nwg, head = 0, grp_idx = 0;
  const bool bd = L.bd_heads > 0;
  if (bd) {  // block-diagonal stack: workgroup -> head
    head = bid / L.bd_wgs;
    wi = bid - head * L.bd_wgs;
    nwg = L.bd_wgs;
  } else {
    int g = 0, wg0 = 0;
#pragma unroll
    for (int k = 0; k < GEMV_MAX_TASKS - 1; ++k)
      if (k + 1 < n_groups && bid >= L.grp_wg_end[k]) { g = k + 1; wg0 = L.grp_wg_end[k]; }
    t0 = L.grp_t0[g];
    t1 = L.grp_t0[g + 1];
    wi = bid - wg0;
    nwg = L.grp_wg_end[g] - wg0;
    grp_idx = g;
  }
  auto task_of = [&](int ti) {
    GemvTask T = L.t[ti];
    if (bd) {  // re-base the one task to this workgroup's head (rows per head: a multiple of 16, gemv_plan)
      T.qs += (size_t)head * (size_t)(T.rows >> 4) * (T.n >> 8) * TILE_B;
      T.a_f32 += (size_t)head * T.n;
      T.out += (size_t)head * T.rows;
      T.vrow_begin = 0;
      T.vrow_end = T.rows;
    }
    return T;
  };
  if (!hinted) {
    const GemvTask Ta = task_of(t0);
    if (tl && tid == 0) tl[6] = wall_clock64();
    stage_q8<LAY_TILE, NW>(Ta, smem, tid, scratch, tl);
  } else if (tl && tid == 0) {
    tl[6] = wall_clock64();
  }
  if (tl && tid == 0) tl[7] = wall_clock64();
  __syncthreads();
  if (tl && tid == 0) tl[1] = wall_clock64();
#ifndef DSK_NO_TAPS
  if (L.tap_qs && !bd && wi == 0)  // parity tap: what this activation group staged
    dump_staged_q8<LAY_TILE>(smem, L.t[t0].n, L.tap_qs + (size_t)grp_idx * L.tap_stride, L.tap_d + (size_t)grp_idx * (L.tap_stride >> 8), tid, NW * 64);
#endif

  // this workgroup's share of the group's (padded) rows, in multiples of part_unit; sharded experts: only the present
  // tasks' rows count (GLU instantiations only, like gemv_body)
  const bool compact = GLU && L.compact_absent && !bd && L.comb_x == nullptr;
  int vtotal = bd ? (L.t[0].rows + 15) & ~15 : L.t[t1 - 1].vrow_end;
  int c_rows = 0, c_base = 0;
  if (compact) {
    if (lane < t1 - t0) {
      const GemvTask& Tk = L.t[t0 + lane];
      bool here = true;
      if (Tk.e_qs != 0) {
        const int le = (Tk.expert_ids ? Tk.expert_ids[Tk.slot] : Tk.slot) - Tk.expert_base;
        here = le >= 0 && le < Tk.local_experts;
      }
      c_rows = here ? Tk.vrow_end - Tk.vrow_begin : 0;
    }
    vtotal = 0;
#pragma unroll
    for (int j = 0; j < GEMV_MAX_TASKS; ++j) {
      const int rj = __builtin_amdgcn_readlane(c_rows, j);
      if (lane == j) c_base = vtotal;
      vtotal += rj;
    }
  }
  const int unit = L.part_unit;
  const long long units = (vtotal + unit - 1) / unit;
  const int r_lo = (int)(units * wi / nwg) * unit;
  int r_hi = (int)(units * (wi + 1) / nwg) * unit;
  if (r_hi > vtotal) r_hi = vtotal;
  const bool comb = !GLU && L.comb_x != nullptr;
  float* red = reinterpret_cast<float*>(smem + t_act);
  const TLane TL = tlane_init(lane);
  bool first = true;
  bool red_live = false;   // a previous round's partials may still be read by a slower wave (also across TASKS)
  int comb_rows = 0;       // combine: this workgroup's rows [r_lo, r_lo + comb_rows) of its ONE task
  float* comb_out = nullptr;

  for (int ti = t0; ti < t1; ++ti) {
    int vb, ve;
    if (compact) {
      vb = __builtin_amdgcn_readlane(c_base, ti - t0);
      ve = vb + __builtin_amdgcn_readlane(c_rows, ti - t0);
      if (r_lo >= ve || r_hi <= vb) continue;
    }
    const GemvTask T = task_of(ti);
    if (!compact) { vb = T.vrow_begin; ve = T.vrow_end; }
    const int lo = (r_lo > vb ? r_lo : vb) - vb, hi = (r_hi < ve ? r_hi : ve) - vb;  // multiples of 16
    if (lo >= hi) continue;
    const int rows_hi = hi < T.rows ? hi : T.rows;  // rows past T.rows are padding
    int le = 0;
    bool present = true;
    if (T.e_qs != 0) {  // slot -> expert on the device (the reference reads active_experts on the host, src/infer.cpp:854)
      const int e = T.expert_ids ? T.expert_ids[T.slot] : T.slot;
      le = e - T.expert_base;
      present = le >= 0 && le < T.local_experts;
    }
    if (comb) { comb_rows = rows_hi - lo; comb_out = T.out + lo; }
    if (!present) {  // the expert lives on another GPU
      if (L.zero_absent && !GLU)
        for (int rr = lo + tid; rr < rows_hi; rr += NW * 64) T.out[rr] = 0.f;
      continue;
    }
    const uint8_t* const W1 = T.qs + (size_t)le * T.e_qs;
    const uint8_t* const W3 = GLU ? T.qs2 + (size_t)le * T.e_qs : T.qs;
    const int nb = T.n >> 8, ips = tile_ips(nb);
    int cap = L.t_rcap / (ips * (GLU ? 2 : 1));  // strips (of each matrix) per round
    if (cap < 1) cap = 1;
    if (cap > TILE_MAX_ROUND_STRIPS) cap = TILE_MAX_ROUND_STRIPS;
    for (int tb = lo >> 4; tb < (hi >> 4); tb += cap) {
      const int nt = (hi >> 4) - tb < cap ? (hi >> 4) - tb : cap;
      // the partials region is reused by every round of every task of this workgroup: wait until the previous round's
      // readers (tile_strip_value) are done - a share that straddles two tasks used to skip this barrier
      if (red_live) __syncthreads();
      red_live = true;
      // static deal: the round's items as contiguous ranges (a wave streams one contiguous byte range of tiles), a barrier,
      // then one wave per strip adds its partials (the association of tile_device.h) and runs the epilogue.
      // (Measured and rejected: the waves pulling 8-step units from an LDS counter, the wave that delivers a
      // strip's last item reducing it, no barrier - with and without the next unit's loads issued ahead of the bookkeeping:
      // classifier 53.7 -> 65 us, experts' w1/w3 21.9 -> 23.5, first-stage projections 5.4 -> 7.5.)
      const int NS = nt * (GLU ? 2 : 1), I = NS * ips;
      const int i0 = (int)((long long)I * wave / NW), i1 = (int)((long long)I * (wave + 1) / NW);
      auto strip_of = [&](int s, rsrc_t& W, int& soff0, const uint8_t*& act) {
        const bool m3 = GLU && s >= nt;
        W = make_rsrc(m3 ? W3 : W1);
        soff0 = (tb + (m3 ? s - nt : s)) * nb * TILE_B;
        act = smem;
      };
      if (nb > 8) tile_items<4>(i0, i1, ips, nb, red, TL, lane, strip_of, [](int, int) {});  // (item size: tile_device.h)
      else tile_items<1>(i0, i1, ips, nb, red, TL, lane, strip_of, [](int, int) {});
      __syncthreads();
      for (int sp = wave; sp < nt; sp += NW) {
        const float v = tile_strip_value(red + (size_t)sp * ips * 64, ips, lane);
        const float v3 = GLU ? tile_strip_value(red + (size_t)(nt + sp) * ips * 64, ips, lane) : 0.f;
        const int row = (tb + sp) * 16 + lane;
        if (lane < 16 && row < T.rows) {
          float* o = T.out + row;
          if (GLU) *o = act_fn(v, L.act) * v3;                                                  // src/infer.cpp:859-872
          else if (comb) __hip_atomic_store(o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: the finisher sits on another CU
          else if (T.epilogue == EPI_ADD) *o += v;                                              // residual add, src/infer.cpp:832-834, 928-930
          else *o = v;
        }
      }
      if (tl && tid == 0 && first) { tl[2] = wall_clock64(); first = false; }
    }
  }
  if (comb) {
    // ---- fused MoE combine: one arrival per task on the counter of this workgroup's row share (every task's group has the
    // same geometry); the LAST task to arrive adds x += w_k * out_k in k order, then the shared expert ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(L.comb_counter + wi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      comb_last = old == (unsigned)L.n_tasks - 1;
      if (comb_last) {
        FINISHER_ACQUIRE();
        __hip_atomic_store(L.comb_counter + wi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
      }
    }
    __syncthreads();
    if (comb_last && comb_out) {
      const int row0 = (int)(comb_out - L.t[t0].out);
      for (int rr = row0 + tid; rr < row0 + comb_rows; rr += NW * 64) {
        float xv = L.comb_x[rr];
        for (int tj = 0; tj < L.n_tasks; ++tj) {
          const float v = __hip_atomic_load(L.t[tj].out + rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (L.t[tj].accum_w) xv = fmaf(v, *L.t[tj].accum_w, xv);  // src/infer.cpp:874-877
          else xv += v;                                             // src/infer.cpp:900-903
        }
        L.comb_x[rr] = xv;
      }
    }
  }
  if (tl && tid == 0) tl[3] = wall_clock64();
}
