// kernels_tile.hip -- the GEMV launch for Q2_K weights in the tiled layout: the reference's _matmul(block_q2_K*) / matmul_expert
// (src/infer.cpp:315-346, 381-469) with ggml_vec_dot_q2_K_q8_K's sub-block dots (src/quant.cpp:666-783) on v_mfma_i32_16x16x64_i8.
// Device code: tile_device.h (layout, row products, association), tile_gemv.h (the launch body); here the kernel, the
// planner and the launcher.  Same descriptor (GemvLaunch), activation groups, staging and epilogues as kernels_gemv.hip.
#include "dsk_internal.h"
#include "tile_gemv.h"
#include <algorithm>

template <bool GLU, int NW>
__global__ __launch_bounds__(NW * 64) void gemv_tile_kernel(const GemvLaunch* __restrict__ Lp, const void* h_a0, const void* h_a1,
                                                              const void* h_a2, int h_n, int h_mode, float h_eps, int h_gwgs, int h_gstride) {
  gemv_tile_body<GLU, NW>(Lp, h_a0, h_a1, h_a2, h_n, h_mode, h_eps, h_gwgs, h_gstride, (int)blockIdx.x, 0.f);
}

size_t tile_mat_bytes(size_t rows, size_t n) { return ((rows + 15) / 16) * (n / 256) * TILE_B; }

// ------------------------------------------------------------------------------------
// planner: workgroup size, grid, activation groups (whole 16-row strips per workgroup), LDS
// ------------------------------------------------------------------------------------
int gemv_plan_tile(GemvLaunch& h, int target_wgs) {
  const bool cg = h.comb_x != nullptr || h.comb_geometry != 0;  // fused-combine geometry: every task its own group, equal shares
  if (h.quant != DSK_QUANT_Q2_K) DSK_FAIL(DSK_ERR_INVALID, "tiled gemv: Q2_K only");
  if (h.n_tasks < 1 || h.n_tasks > GEMV_MAX_TASKS) DSK_FAIL(DSK_ERR_INVALID, "gemv: %d tasks", h.n_tasks);
  int nb_max = 0;
  long total_items = 0, total_strips = 0;
  double total_work = 0;
  for (int i = 0; i < h.n_tasks; ++i) {
    GemvTask& T = h.t[i];
    if (T.rows <= 0 || T.n <= 0) DSK_FAIL(DSK_ERR_INVALID, "gemv: empty shape %d x %d", T.rows, T.n);
    if (T.n % QK_K) DSK_FAIL(DSK_ERR_INVALID, "k-quant gemv: n=%d is not a multiple of 256 (quantizer.cpp:8)", T.n);
    if (cg && T.rows != h.t[0].rows) DSK_FAIL(DSK_ERR_INVALID, "gemv combine: tasks must share the row count");
    const int nb = T.n >> 8, strips = (T.rows + 15) >> 4;
    nb_max = nb > nb_max ? nb : nb_max;
    total_strips += strips;
    total_items += (long)strips * tile_ips(nb) * (h.glu ? 2 : 1);
    total_work += (double)strips * nb * (h.glu ? 2 : 1);
  }
  if (h.bd_heads > 0 && (h.t[0].rows & 15)) DSK_FAIL(DSK_ERR_UNSUPPORTED, "tiled gemv: %d rows per head (a multiple of 16 is needed)", h.t[0].rows);
  if (h.b0 < 1) h.b0 = 1;
  if (h.b1 < 1) h.b1 = 1;
  h.lpr_log2 = 4; h.R = 1; h.U = 4;  // (reported by dsk_plan_gemv; not used by the tiled kernel)
  // Workgroup size: 16 waves (one workgroup per CU: a quarter of the redundant staging prologues of 4-wave workgroups) as
  // soon as the launch has a wave's worth of items for most of them; small launches keep 4-wave workgroups
  // Workgroup size.  Big launches: 8-wave workgroups, two per CU (tools/mfma_gemv_probe.hip: experts' w1/w3 19.3 us with one
  // 16-wave workgroup per CU, 16.6 with two 8-wave ones, 16.1 with four 4-wave ones: a workgroup's barriers and its staging
  // prologue stall only half of the CU, while the prologue is redundant twice instead of four times).  Small launches
  // (the first-stage projections: 132 strips) are bound by the serial prologue: 16 waves share it.
  const long eff_items = h.bd_heads > 0 ? total_items * h.bd_heads : total_items;
  h.NW = 16;
  if (cg) h.NW = 8;
  (void)eff_items;
  if (h.force_NW == 4 || h.force_NW == 8 || h.force_NW == 16) h.NW = h.force_NW;
  h.part_unit = 16;
  h.t_act = (int)(((size_t)nb_max * TREC + 15) & ~(size_t)15);
  // partials of a round: big rounds balance the waves (items are dealt as contiguous ranges) and cost one barrier pair each
  h.t_rcap = h.NW == 16 ? 256 : (h.NW == 8 ? 128 : 48);
  {
    int ips = 0;  // a round holds at least one strip (pair) of any task
    for (int i = 0; i < h.n_tasks; ++i) ips = std::max(ips, tile_ips(h.t[i].n >> 8) * (h.glu ? 2 : 1));
    if (h.t_rcap < ips) h.t_rcap = ips;
  }
  h.lds_bytes = (size_t)h.t_act + (size_t)h.t_rcap * 256;
  if (h.lds_bytes > 150 * 1024) DSK_FAIL(DSK_ERR_UNSUPPORTED, "gemv: activation vector(s) need %zu B of LDS", h.lds_bytes);
  if (h.bd_heads > 0) {
    const int strips = h.t[0].rows >> 4;
    int per_head = target_wgs / h.bd_heads;
    if (per_head < 1) per_head = 1;
    if (per_head > strips) per_head = strips;
    h.bd_wgs = per_head;
    h.grid = per_head * h.bd_heads;
    h.t[0].wg_begin = 0; h.t[0].wg_end = h.grid;
    h.t[0].vrow_begin = 0; h.t[0].vrow_end = h.t[0].rows;
    h.n_groups = 1; h.grp_wg_end[0] = h.grid; h.grp_t0[0] = 0; h.grp_t0[1] = 1;
    return DSK_OK;
  }
  int W = h.NW == 16 ? 256 - (h.reserve_wgs > 0 && h.reserve_wgs < 128 ? h.reserve_wgs : 0) : (h.NW == 8 ? 512 : target_wgs);
  if (h.force_U > 8) W = h.force_U;  // micro-benchmarks (tools/kbench.py sweep): the workgroup count
  int wg = 0;
  h.n_groups = 0;
  for (int i = 0; i < h.n_tasks;) {
    int j = i + 1;
    while (!cg && j < h.n_tasks && h.t[j].act_mode == h.t[i].act_mode && h.t[j].a_f32 == h.t[i].a_f32 && h.t[j].a_qs == h.t[i].a_qs &&
           h.t[j].norm_w == h.t[i].norm_w && h.t[j].n == h.t[i].n)
      ++j;
    long rows_g = 0;
    double work_g = 0;
    for (int k = i; k < j; ++k) {
      const int padded = (h.t[k].rows + 15) & ~15;
      h.t[k].vrow_begin = (int)rows_g;
      rows_g += padded;
      h.t[k].vrow_end = (int)rows_g;
      work_g += (double)(padded >> 4) * (h.t[k].n >> 8) * (h.glu ? 2 : 1);
    }
    const long strips_g = rows_g >> 4;
    int share = (int)(W * (work_g / total_work) + 0.5);
    if (share > strips_g) share = (int)strips_g;
    if (share < 1) share = 1;
    if (h.fill_div > 0 && !cg) {  // a cap on the workgroups of a group (the rider of the router launch)
      const long items_g = strips_g * tile_ips(h.t[i].n >> 8) * (h.glu ? 2 : 1);
      long cap = (h.fill_div * items_g + h.NW - 1) / h.NW;
      if (cap < 1) cap = 1;
      if (share > cap) share = (int)cap;
    }
    if (cg) share = (int)((strips_g + (strips_g + share - 1) / share - 1) / ((strips_g + share - 1) / share));  // even shares
    for (int k = i; k < j; ++k) { h.t[k].wg_begin = wg; h.t[k].wg_end = wg + share; }
    wg += share;
    h.grp_t0[h.n_groups] = i;
    h.grp_wg_end[h.n_groups] = wg;
    ++h.n_groups;
    i = j;
  }
  h.grp_t0[h.n_groups] = h.n_tasks;
  h.grid = wg;
  return DSK_OK;
}

// ------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------
template <bool GLU, int NW>
static void launch_tile_one(hipStream_t st, const GemvLaunch* dev, const GemvLaunch& h) {
  dim3 grid(h.grid), block(NW * 64);
  // activation hint (gemv_device.h ActSrc): only when every workgroup stages the same vector, or equally spaced ones
  const void *a0 = nullptr, *a1 = nullptr, *a2 = nullptr;
  int hn = 0, hm = 0, gw = 0, gs = 0;
  float he = 0.f;
  if (h.bd_heads == 0 && h.n_groups == 1) {
    const GemvTask& T = h.t[h.grp_t0[0]];
    hn = T.n; hm = T.act_mode; he = T.eps;
    if (T.act_mode == ACT_Q8) { a0 = T.a_qs; a1 = T.a_d; a2 = T.a_bsums; }
    else { a0 = T.a_f32; a1 = T.norm_w; }
  } else if (h.bd_heads == 0 && h.n_groups > 1) {
    const GemvTask& T0 = h.t[h.grp_t0[0]];
    const int w0 = h.grp_wg_end[0];
    const ptrdiff_t sd = h.t[h.grp_t0[1]].a_f32 - T0.a_f32;
    bool ok = T0.act_mode == ACT_F32 && sd > 0 && sd < (1 << 30);
    for (int g = 0; ok && g < h.n_groups; ++g) {
      const GemvTask& T = h.t[h.grp_t0[g]];
      ok = T.act_mode == ACT_F32 && T.n == T0.n && T.a_f32 == T0.a_f32 + (ptrdiff_t)g * sd && h.grp_wg_end[g] == (g + 1) * w0;
    }
    if (ok) { a0 = T0.a_f32; hn = T0.n; hm = ACT_F32; gw = w0; gs = (int)sd; }
  }
  auto k = gemv_tile_kernel<GLU, NW>;
  if (h.lds_bytes > 64 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds_bytes);
  if (g_prof_start && g_prof_stop) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)h.lds_bytes, st, g_prof_start, g_prof_stop, 0u, dev, a0, a1, a2, hn, hm, he, gw, gs);
  else hipLaunchKernelGGL(k, grid, block, h.lds_bytes, st, dev, a0, a1, a2, hn, hm, he, gw, gs);
}

int gemv_launch_tile(hipStream_t st, const GemvLaunch* dev, const GemvLaunch& h) {
  if (h.glu) {
    if (h.NW == 16) launch_tile_one<true, 16>(st, dev, h);
    else if (h.NW == 8) launch_tile_one<true, 8>(st, dev, h);
    else launch_tile_one<true, 4>(st, dev, h);
  } else {
    if (h.NW == 16) launch_tile_one<false, 16>(st, dev, h);
    else if (h.NW == 8) launch_tile_one<false, 8>(st, dev, h);
    else launch_tile_one<false, 4>(st, dev, h);
  }
  return DSK_OK;
}
