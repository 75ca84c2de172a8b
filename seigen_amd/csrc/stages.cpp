// Stage launches of the C-ABI: regions of a split stage, the six fused launches of an LF4 step
// (seigen/elastic.py:283-313), hipGraph replay, un-fused operator applications, halo packs, timing.
#include "handle.hpp"

// ---- stage launches --------------------------------------------------------------------

static bool source_active(const sg_handle* h) {
  return h->src_nnz != 0 && (h->src_static || h->src_step < h->src_nsteps);
}
// launches of a capture: slice and weight of the step the device-side counter names (kernels.hpp SrcStep)
static SrcStep source_stepper(const sg_handle* h) {
  SrcStep ss;
  std::memset(&ss, 0, sizeof(ss));
  if (!h->capture_src) return ss;
  const int64_t dd = (int64_t)h->cfg.dim * h->cfg.dim;
  ss.ctr = h->src_ctr_d;
  ss.nsteps = h->src_nsteps;
  ss.is_static = h->src_static ? 1 : 0;
  ss.weights = h->src_weights.empty() ? nullptr : h->src_weights_d;
  ss.stride = (h->src_static || !h->src_weights.empty()) ? 0 : h->src_nnz * dd;
  if (ss.stride == 0 && ss.weights == nullptr) ss.is_static = 1;
  return ss;
}
// separable source: the one stored slice, scaled by this step's weight
static bool source_one_slice(const sg_handle* h) { return h->src_static || !h->src_weights.empty(); }
static double source_scale(const sg_handle* h) {
  return h->src_weights.empty() ? 1.0 : h->src_weights[(size_t)h->src_step];
}

static int run_op(sg_handle* h, int kind, int in_f, int out_f, int aux_f, int mode, double c_self, double c_aux,
                  double c_new, int region, int uabs_f = SG_FIELD_U, bool with_source = false, bool density = false,
                  double src_coef = 1.0) {
  StageArgs a;
  std::memset(&a, 0, sizeof(a));
  a.in = h->field[in_f];
  a.out = h->field[out_f];
  a.aux = aux_f >= 0 ? h->field[aux_f] : nullptr;
  a.uabs = h->field[uabs_f];
  for (int s = 0; s < 6; ++s) {
    a.ghost[s] = h->ghost[in_f][s];
    // required for interior launches too: masked boundary lanes still form (and load through) the pointer
    if (h->md.has_nbr[s] && !a.ghost[s])
      return fail(h, SG_ERR_STATE, "stage needs a halo buffer that was not attached (sg_halo_attach)");
  }
  a.Dt = h->Dt;
  a.Lt = h->Lt;
  a.md = h->md_dev;
  a.mk = h->mk_dev;
  a.ftab = h->ftab_dev;
  a.nbr_tab = h->nbr_tab;
  a.name_out = h->name_out;
  a.all_active = region == SG_REGION_ALL ? 1 : 0;
  a.tensor = h->re.kind == KIND_TENSOR ? 1 : 0;
  a.fragV = (kind == 0) ? h->fragF : h->fragG;
  a.fragL = h->fragL;
  if (kind == 1 && h->fragQ) {      // G stages with the factorised volume term
    a.fragV = h->fragP;
    a.fragQ = h->fragQ;
  }
  a.sym = h->sym ? 1 : 0;
  a.f32 = h->f32;
  a.dbg = h->dbg ? h->dbg + 8 * (kind * 2 + (mode ? 1 : 0)) : nullptr;
  a.sponge_slot = (kind == 0) ? h->sponge_slot : nullptr;
  a.sponge_B = h->sponge_B;
  a.sponge_sigma = (kind == 0) ? h->sponge_sigma : nullptr;
  a.sponge_pre = (kind == 0) ? h->sponge_pre : nullptr;
  // the first launch of an F stage - whichever region the caller starts with: the same stage again, or a region it has
  // already seen, is the next instance of the stage
  bool first_of_stage = false;
  if (kind == 0 && h->sponge_pre && !h->name_out) {
    const int key = out_f * 4 + mode;
    first_of_stage = key != h->sponge_pre_key || (h->sponge_pre_regions & (1 << region)) != 0 || region == SG_REGION_ALL;
    if (first_of_stage) h->sponge_pre_regions = 0;
    h->sponge_pre_key = key;
    h->sponge_pre_regions |= 1 << region;
  }
  if (first_of_stage && h->sponge_pre_field == uabs_f && h->sponge_pre_ver == h->fver[uabs_f]) first_of_stage = false;   // the buffer holds it
  if (first_of_stage) {
    h->sponge_pre_field = uabs_f;
    h->sponge_pre_ver = h->fver[uabs_f];
    // B_e u_abs of the cells with a sponge matrix, before anything of the stage writes
    if (launch_sponge_pre(a.uabs, h->sponge_B, h->sponge_cells, h->sponge_mat, h->sponge_mat_slots, h->sponge_pre, h->sponge_nmat_slots,
                          h->re.nd, h->cfg.dim, h->ncls, (int)h->md.gw, h->sponge_pre_lines, h->f32, h->stream) != 0)
      return fail(h, SG_ERR_DEVICE, "sponge pre-pass launch failed");
    // ... and of the cells whose sigma is affine in the reference coordinates: dim + 1 numbers per cell, element-constant matrices
    const int arc = h->sponge_aff_frag
                        ? launch_sponge_affine_mfma(h->cfg.degree, a.uabs, h->sponge_aff_frag, h->sponge_aff_items, h->sponge_aff_slots,
                                                    h->sponge_aff_coef, h->sponge_pre, h->sponge_aff_nitems, h->stream)
                        : launch_sponge_pre_affine(a.uabs, h->sponge_aff_X, h->sponge_aff_col, h->sponge_aff_W, h->sponge_aff_items,
                                                   h->sponge_aff_slots, h->sponge_aff_coef, h->sponge_pre, h->sponge_aff_nitems, h->re.nd,
                                                   h->cfg.dim, (int)h->md.gw, h->sponge_pre_lines, h->f32, h->stream);
    if (arc != 0) return fail(h, SG_ERR_DEVICE, "affine-sigma sponge pre-pass launch failed");
    // SECOND runs on its own stream after ev_stage - "everything before this stage's FIRST" - and reads the pre-pass too
    if (region == SG_REGION_FIRST && h->overlap && h->first_recorded_stage >= 0) HIPCHECK(h, hipEventRecord(h->ev_stage, h->stream));
  }
  if (!h->name_out) h->fver[out_f] += 1;     // (after the pre-pass decision: an in-place stage absorbs the state it overwrites)
  a.lam = h->lam_d;
  a.mu = h->mu_d;
  a.lam0 = h->lam0;
  a.mu0 = h->mu0;
  a.per_cell = h->per_cell;
  a.rho2 = (kind == 0 && mode == 1 && density) ? h->rho2_d : nullptr;   // stage U1 only
  a.src_coef = src_coef;
  a.mode = mode;
  a.c_self = c_self;
  a.c_aux = c_aux;
  a.c_new = c_new;
  if (kind == 0 && in_f == SG_FIELD_S && mode == 0 && h->capture_src && h->src_fused) {
    // stage UH1 of a captured step on the tile path: the launch that opens the step also names it
    a.src_step = source_stepper(h);
    a.src_bump = 1;
  }
  if (with_source && h->src_fused && h->capture_src) {     // ... of the step the device-side counter names
    a.src_slot = h->src_slot_d;
    a.src_idx = h->src_idx_d;
    a.src_vals = h->src_values;
    a.src_scale = 1.0;
    a.src_step = source_stepper(h);
  } else if (with_source && h->src_fused && source_active(h)) {  // tile path: the G kernel adds this step's source values
    a.src_slot = h->src_slot_d;
    a.src_idx = h->src_idx_d;
    a.src_vals = h->src_values + (size_t)(source_one_slice(h) ? 0 : h->src_step) * h->src_nnz * h->cfg.dim * h->cfg.dim;
    a.src_scale = source_scale(h);
  }
  std::vector<Box> boxes;
  region_boxes(h, region, boxes);
  if (h->use_mfma || h->use_lane || h->use_tile || h->use_hexm) {
    // one launch for the whole region: the kernels scan all cell groups and mask lanes by box
    a.nbox = 0;
    for (const Box& b : boxes) {
      if (b.n[0] <= 0 || b.n[1] <= 0 || b.n[2] <= 0) continue;
      if (a.nbox >= SG_MAX_BOXES) return fail(h, SG_ERR_STATE, "region has more boxes than a launch can carry");
      for (int k = 0; k < 3; ++k) {
        a.boxes_o[a.nbox][k] = b.o[k];
        a.boxes_n[a.nbox][k] = b.n[k];
      }
      a.nbox += 1;
    }
    if (a.nbox == 0) return SG_OK;
    a.spread = (region == SG_REGION_BOUNDARY) ? 1 : 0;
    // only launches that run while an exchange is in flight leave block slots to RCCL
    a.grid_blocks = (region == SG_REGION_INTERIOR || region == SG_REGION_SECOND) ? h->grid_blocks : h->grid_full;
    // (the sponge is part of F only, and only cells with a matrix of their own make items differ in cost: then one item per
    // wave on the prime-strided grid)
    if (h->use_tile) a.grid_blocks = (h->sponge_nslots > 0 && kind == 0) ? h->tile_grid_sponge : h->tile_grid;
    a.item_list = nullptr;
    a.nlist = 0;
    a.order_chunk = (h->use_mfma && !a.spread && kind == 0) ? h->order_chunk : 0;
    if (region != SG_REGION_ALL) {
      // Both regions of a split stage are static: list the (cell group, class) items that have
      // an active cube once.  The interior launch then splits ACTIVE items evenly over the XCDs
      // (a shell is whole z-layers of groups, i.e. the first items of XCD 0 and the last of XCD 7:
      // skipping them inside an even split of all items would leave the launch as long as before);
      // the shell launch deals its few items round-robin over all waves.
      int32_t*& list = h->region_items[region];
      int32_t& nlist = h->region_nitems[region];
      if (nlist < 0) {
        const int64_t gw = h->md.gw, ngroups = h->md.ncube_pad / gw;
        std::vector<char> hit((size_t)ngroups, 0);
        std::vector<int32_t> cnt((size_t)ngroups, 0);
        for (int bx = 0; bx < a.nbox; ++bx)
          for (int ck = a.boxes_o[bx][2]; ck < a.boxes_o[bx][2] + a.boxes_n[bx][2]; ++ck)
            for (int cj = a.boxes_o[bx][1]; cj < a.boxes_o[bx][1] + a.boxes_n[bx][1]; ++cj)
              for (int ci = a.boxes_o[bx][0]; ci < a.boxes_o[bx][0] + a.boxes_n[bx][0]; ++ci) {
                int64_t cube = ci + (int64_t)h->cfg.n[0] * (cj + (int64_t)h->cfg.n[1] * ck);
                hit[(size_t)(cube / gw)] = 1;
                cnt[(size_t)(cube / gw)] += 1;
              }
        // Whole groups only (always, on meshes whose rows are a multiple of the group width: a shell in x is one
        // group thick): the kernels then skip the cube coordinates and the box tests, as in a whole-block launch.
        // (The boxes of a region are disjoint, so the count of a group tells.)
        bool whole = true;
        for (int64_t g = 0; g < ngroups; ++g)
          if (hit[(size_t)g] && cnt[(size_t)g] != std::min<int64_t>(gw, h->md.ncube - g * gw)) whole = false;
        h->region_whole[region] = whole;
        std::vector<int32_t> items;
        for (int64_t g = 0; g < ngroups; ++g)
          if (hit[(size_t)g])
            for (int k = 0; k < h->ncls; ++k) items.push_back((int32_t)(g * h->ncls + k));
        nlist = (int32_t)items.size();
        if (!items.empty()) {
          HIPCHECK(h, hipMalloc((void**)&list, items.size() * sizeof(int32_t)));
          HIPCHECK(h, hipMemcpy(list, items.data(), items.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
      }
      a.item_list = list;
      a.nlist = nlist;
      if ((h->use_mfma || h->use_hexm) && h->region_whole[region] && !h->no_whole) a.all_active = 1;
    }
    a.nitems = a.item_list ? a.nlist : (int32_t)std::min<int64_t>((h->md.ncube_pad / h->md.gw) * h->ncls, INT32_MAX);
    int rc = h->use_mfma   ? launch_stage_mfma(kind, h->cfg.degree, a, h->stream)
             : h->use_hexm ? launch_stage_hexm(kind, h->cfg.degree, a, (long)(h->md.ncube_pad / 16), h->stream)
             : h->use_tile ? launch_stage_tile2d(kind, h->cfg.degree, a, h->t2c, (long)(h->md.ncube_pad / 16) * h->ncls, h->stream)
                           : launch_stage_lane(kind, h->cfg.dim, h->cfg.degree, a, (long)(h->md.ncube_pad / 64) * h->ncls, h->stream);
    if (rc != 0) return fail(h, SG_ERR_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)rc));
    return SG_OK;
  }
  for (const Box& b : boxes) {
    bool empty = false;
    for (int k = 0; k < 3; ++k) {
      a.box_o[k] = b.o[k];
      a.box_n[k] = b.n[k];
      empty = empty || (b.n[k] <= 0);
    }
    if (empty) continue;
    int rc = launch_stage(kind, h->cfg.dim, h->cfg.degree, a, h->stream);
    if (rc != 0) return fail(h, SG_ERR_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)rc));
  }
  return SG_OK;
}

// the source lives on single nodes: added to each part of a split stage right after the launch
// that wrote it (INTERIOR + BOUNDARY: all of it after the second launch)
static int add_source(sg_handle* h, int field, double coef, int region = SG_REGION_ALL) {
  if (h->src_fused || h->name_out) return SG_OK;  // added by the stage kernel (run_op with_source) / a naming pass launches nothing
  if (h->src_nnz == 0 || region == SG_REGION_INTERIOR) return SG_OK;
  if (!h->capture_src && !h->src_static && h->src_step >= h->src_nsteps) return SG_OK;
  const int d = h->cfg.dim;
  int64_t off = 0, cnt = h->src_nnz;
  if (region == SG_REGION_FIRST) cnt = h->src_nfirst;
  if (region == SG_REGION_SECOND) {
    off = h->src_nfirst;
    cnt = h->src_nnz - h->src_nfirst;
  }
  if (cnt == 0) return SG_OK;
  const SrcStep ss = source_stepper(h);
  const double* vals = h->src_values + ((size_t)((source_one_slice(h) || ss.ctr) ? 0 : h->src_step) * h->src_nnz + off) * d * d;
  int rc = launch_source(h->field[field], d * d, h->md.gw, cnt, h->src_nodes + off, vals, coef, ss.ctr ? 1.0 : source_scale(h), ss,
                         h->f32, h->stream);
  if (rc != 0) return fail(h, SG_ERR_DEVICE, "source kernel launch failed");
  return SG_OK;
}

static int run_stage_impl(sg_handle* h, int stage, int region) {
  const double dt = h->dt, c3 = dt * dt * dt / 24.0;
  int rc = SG_OK;
  switch (stage) {
    case SG_STAGE_UH1:
      return run_op(h, 0, SG_FIELD_S, SG_FIELD_UH, -1, 0, 0, 0, 0, region);
    case SG_STAGE_STEMP:
      rc = run_op(h, 1, SG_FIELD_UH, SG_FIELD_SH, -1, 0, 0, 0, 0, region, SG_FIELD_U, true);
      if (rc == SG_OK) rc = add_source(h, SG_FIELD_SH, 1.0, region);
      return rc;
    case SG_STAGE_U1:
      // explicit mode keeps only rhs(form_u1): u1 = rho*u0 + dt*uh1 + dt^3/24*uh2 (elastic.py:341-345, :354-356);
      // sg_set_density(physical = 1): u1 = u0 + (dt*uh1 + dt^3/24*uh2)/rho; per-cell density: factors in rho2
      if (h->rho2_d) return run_op(h, 0, SG_FIELD_SH, SG_FIELD_U, SG_FIELD_UH, 1, 1.0, dt, c3, region, SG_FIELD_U, false, true);
      if (h->rho_physical) return run_op(h, 0, SG_FIELD_SH, SG_FIELD_U, SG_FIELD_UH, 1, 1.0, dt / h->rho, c3 / h->rho, region);
      return run_op(h, 0, SG_FIELD_SH, SG_FIELD_U, SG_FIELD_UH, 1, h->rho, dt, c3, region);
    case SG_STAGE_SH1:
      rc = run_op(h, 1, SG_FIELD_U, SG_FIELD_SH, -1, 0, 0, 0, 0, region, SG_FIELD_U, true);
      if (rc == SG_OK) rc = add_source(h, SG_FIELD_SH, 1.0, region);
      return rc;
    case SG_STAGE_UTEMP:
      // utemp = F(sh1; u1) has one consumer, sh2 = G(utemp) in the stress update s1 = s0 + dt sh1 + dt^3/24 sh2 with
      // sh1 = G(u1) + S (elastic.py:300-303, :348-352) - and g is LINEAR in the velocity: dt G(u1) + dt^3/24 G(utemp) =
      // G(dt u1 + dt^3/24 utemp).  So this stage leaves w = dt u1 + dt^3/24 utemp in UH (one more operand in its fused
      // epilogue, on a stage that waits for the matrix pipe, not for memory) and stage S1 reads w and s0 ONLY: no sh1, no
      // second right-hand side - 6 of its 21 words per node gone (the halo exchanged after this stage is w's).
      // (mode 2: the fused form without the self term, out = c_aux aux + c_new rhs; kernel families without an instantiation
      // of their own run their mode-1 kernels with c_self = 0)
      return run_op(h, 0, SG_FIELD_SH, SG_FIELD_UH, SG_FIELD_U, 2, 0.0, dt, c3, region);
    case SG_STAGE_S1:
      // s1 = s0 + G(w) + (dt + dt^3/24) S   (G stage kernels, fused form: out = c_self out + c_new rhs, no second operand)
      rc = run_op(h, 1, SG_FIELD_UH, SG_FIELD_S, -1, 1, 1.0, 0.0, 1.0, region, SG_FIELD_U, true, false, dt + c3);
      if (rc == SG_OK) rc = add_source(h, SG_FIELD_S, dt + c3, region);
      return rc;
  }
  return fail(h, SG_ERR_ARG, "unknown stage");
}

int resolve_timing(sg_handle* h) {
  if (h->ev_stage_ids.empty()) return SG_OK;
  if (int rc = join_second(h)) return rc;
  HIPCHECK(h, sync_all(h));
  for (size_t k = 0; k < h->ev_stage_ids.size(); ++k) {
    float ms = 0;
    HIPCHECK(h, hipEventElapsedTime(&ms, h->ev_pool[2 * k], h->ev_pool[2 * k + 1]));
    const int id = h->ev_stage_ids[k], st = id & 15;
    if (st == 6) {
      h->counters.halo_pack_ms += ms;
    } else if (id & 16) {
      // FIRST of a stage whose SECOND runs beside it on the other stream: both pairs start at (about) the same
      // point and overlap, so the stage's device time is the LONGER of the two, not their sum
      if (h->first_ms_pending[st] >= 0) h->counters.kernel_ms[st] += h->first_ms_pending[st];  // a FIRST without SECOND
      h->first_ms_pending[st] = ms;
    } else if (id & 32) {
      // (a SECOND whose FIRST was not timed - timing switched on between the two - counts with its own time)
      const double f = h->first_ms_pending[st];
      h->counters.kernel_ms[st] += (f >= 0 && f > ms) ? f : (double)ms;
      h->first_ms_pending[st] = -1;
    } else {
      h->counters.kernel_ms[st] += ms;
    }
  }
  h->ev_stage_ids.clear();
  return SG_OK;
}

extern "C" {

int sg_run_stage(sg_handle* h, int stage, int region) {
  if (!h) return SG_ERR_ARG;
  if (!h->params_set) return fail(h, SG_ERR_STATE, "sg_set_params must be called before stepping");
  if (region < 0 || region > 4) return fail(h, SG_ERR_ARG, "unknown region");
  if (stage < 0 || stage > 5) return fail(h, SG_ERR_ARG, "unknown stage");
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  const bool second = h->overlap && region == SG_REGION_SECOND;
  hipStream_t const main_stream = h->stream;
  if (second) {
    // depends on everything before this stage's FIRST (ev_stage), not on FIRST itself - so FIRST of the SAME stage
    // must have been issued (it records ev_stage); anything else would wait on a stale event and race
    if (h->first_recorded_stage != stage)
      return fail(h, SG_ERR_STATE, "sg_run_stage(stage, SG_REGION_SECOND) must follow sg_run_stage(stage, SG_REGION_FIRST) of the same stage");
    h->first_recorded_stage = -1;
    HIPCHECK(h, hipStreamWaitEvent(h->stream2, h->ev_stage, 0));
    h->stream = h->stream2;
  } else {
    if (int rc = join_second(h)) return rc;
    if (h->overlap && region == SG_REGION_FIRST) {
      HIPCHECK(h, hipEventRecord(h->ev_stage, h->stream));
      h->first_recorded_stage = stage;
    }
  }
  size_t k = h->ev_stage_ids.size();
  int rc = SG_OK;
  if (h->timing) {
    if (k >= 8192) {
      h->stream = main_stream;
      rc = resolve_timing(h);
      if (rc != SG_OK) return rc;
      if (second) h->stream = h->stream2;
      k = 0;
    }
    while (rc == SG_OK && h->ev_pool.size() < 2 * k + 2) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) rc = fail(h, SG_ERR_DEVICE, "hipEventCreate failed");
      else h->ev_pool.push_back(e);
    }
    if (rc == SG_OK && hipEventRecord(h->ev_pool[2 * k], h->stream) != hipSuccess) rc = fail(h, SG_ERR_DEVICE, "hipEventRecord failed");
  }
  if (rc == SG_OK) rc = run_stage_impl(h, stage, region);
  if (rc == SG_OK && h->timing) {
    if (hipEventRecord(h->ev_pool[2 * k + 1], h->stream) != hipSuccess) rc = fail(h, SG_ERR_DEVICE, "hipEventRecord failed");
    else h->ev_stage_ids.push_back(stage + ((h->overlap && region == SG_REGION_FIRST) ? 16 : (second ? 32 : 0)));
  }
  if (second) {
    if (rc == SG_OK && hipEventRecord(h->ev_second, h->stream2) != hipSuccess) rc = fail(h, SG_ERR_DEVICE, "hipEventRecord failed");
    h->second_pending = rc == SG_OK;
    h->stream = main_stream;
  }
  if (rc != SG_OK) return rc;
  h->counters.launches[stage] += 1;
  return SG_OK;
}

int sg_end_step(sg_handle* h) {
  if (!h) return SG_ERR_ARG;
  h->src_step += 1;
  h->counters.steps += 1;
  return SG_OK;
}

// one LF4 step = six launches on the handle's stream (elastic.py:291-304)
static int enqueue_step(sg_handle* h) {
  for (int st = 0; st < 6; ++st) {
    int rc = run_stage_impl(h, st, SG_REGION_ALL);
    if (rc != SG_OK) return rc;
  }
  // the next step's slice (tile path: stage UH1 bumps the counter itself, run_op)
  if (h->capture_src && !h->src_fused && launch_step_counter(h->src_ctr_d, 1, 1, h->stream) != 0)
    return fail(h, SG_ERR_DEVICE, "step counter launch failed");
  return SG_OK;
}

// capture `steps` steps into an executable graph; on any failure graphs are switched off for the handle
static hipGraphExec_t capture_steps(sg_handle* h, int steps, bool with_src) {
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  {
    // a dry pass through the launch code (nothing is queued): what a launcher asks the runtime once per kernel
    // instantiation - the resident blocks of the 2-D tile kernels - is asked here, outside the capture
    std::string name;
    h->name_out = &name;
    for (int st = 0; st < 6; ++st) (void)run_stage_impl(h, st, SG_REGION_ALL);
    h->name_out = nullptr;
  }
  if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return nullptr;
  int rc = SG_OK;
  h->capture_src = with_src;
  h->sponge_pre_ver = ~0ull;      // a replay starts from whatever the buffer holds: the captured step computes its own
  for (int k = 0; k < steps && rc == SG_OK; ++k) rc = enqueue_step(h);
  h->sponge_pre_ver = ~0ull;      // nothing was launched: the buffer holds what it held
  h->capture_src = false;
  hipError_t e = hipStreamEndCapture(h->stream, &g);
  if (rc == SG_OK && e == hipSuccess && g) {
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) ge = nullptr;
  }
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  return ge;
}

int sg_step(sg_handle* h, int64_t nsteps) {
  if (!h || nsteps < 0) return SG_ERR_ARG;
  if (!h->params_set) return fail(h, SG_ERR_STATE, "sg_set_params must be called before stepping");
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  for (int s = 0; s < 6; ++s)
    if (h->md.has_nbr[s]) {
      if (h->comm) return comm_step(h, nsteps);     // the native exchange (comm.cpp)
      return fail(h, SG_ERR_STATE, "sg_step on a block with neighbours: attach a communicator (sg_comm_init) or drive "
                                   "stages + halo from the host");
    }
  int64_t k = 0;
  // launch-bound blocks: replay captured graphs (no per-stage timing).  A source that is still active is part of the
  // graphs: its launches take this step's slice and weight from a device-side step counter (kernels.hpp SrcStep),
  // set here to the step the replay starts from; one that has run out (or none) gives graphs without source launches.
  const bool graphs = h->graph_ok && !h->timing && nsteps >= 2;
  const bool with_src = source_active(h) && h->src_ctr_d != nullptr;
  if (graphs && source_active(h) && !with_src) return fail(h, SG_ERR_STATE, "source without a device-side step counter");
  if (graphs && (h->graph_epoch != h->epoch || h->graph_src != with_src)) {
    if (h->graph1) (void)hipGraphExecDestroy(h->graph1);
    if (h->graph8) (void)hipGraphExecDestroy(h->graph8);
    h->graph1 = capture_steps(h, 1, with_src);
    h->graph8 = h->graph1 ? capture_steps(h, 8, with_src) : nullptr;
    h->graph_epoch = h->epoch;
    h->graph_src = with_src;
    if (!h->graph1 || !h->graph8) h->graph_ok = false;  // same kernels, launched one by one below
  }
  HIPCHECK(h, hipEventRecord(h->ev0, h->stream));
  if (graphs && h->graph_ok) {
    // (tile path: the counter is bumped by the launch that OPENS a step, so it starts one short)
    if (with_src && launch_step_counter(h->src_ctr_d, h->src_step - (h->src_fused ? 1 : 0), 0, h->stream) != 0)
      return fail(h, SG_ERR_DEVICE, "step counter launch failed");
    for (; k + 8 <= nsteps; k += 8) HIPCHECK(h, hipGraphLaunch(h->graph8, h->stream));
    for (; k < nsteps; ++k) HIPCHECK(h, hipGraphLaunch(h->graph1, h->stream));
    for (int st = 0; st < 6; ++st) h->counters.launches[st] += nsteps;
    h->counters.steps += nsteps;
    h->src_step += nsteps;
    if (nsteps > 0) {
      for (int f = 0; f < 4; ++f) h->fver[f] += 1;
      h->sponge_pre_ver = ~0ull;
    }
  }
  for (; k < nsteps; ++k) {
    for (int st = 0; st < 6; ++st) {
      if (h->timing) {
        int rc = sg_run_stage(h, st, SG_REGION_ALL);
        if (rc != SG_OK) return rc;
      } else {
        int rc = run_stage_impl(h, st, SG_REGION_ALL);
        if (rc != SG_OK) return rc;
        h->counters.launches[st] += 1;
      }
    }
    h->src_step += 1;
    h->counters.steps += 1;
  }
  HIPCHECK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHECK(h, hipEventSynchronize(h->ev1));
  float ms = 0;
  HIPCHECK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms;
  return SG_OK;
}

int sg_last_step_ms(sg_handle* h, double* ms) {
  if (!h || !ms) return SG_ERR_ARG;
  *ms = h->last_ms;
  return SG_OK;
}

int sg_apply_F(sg_handle* h, int s_in, int u_abs, int u_out) {
  if (!h) return SG_ERR_ARG;
  if (!field_is_stress(s_in) || field_is_stress(u_out) || field_is_stress(u_abs) || u_abs == u_out)
    return fail(h, SG_ERR_ARG, "sg_apply_F: s_in must be a stress field, u_abs/u_out distinct velocity fields");
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  return run_op(h, 0, s_in, u_out, -1, 0, 0, 0, 0, SG_REGION_ALL, u_abs);
}

int sg_apply_G(sg_handle* h, int u_in, int s_out, int use_source) {
  if (!h) return SG_ERR_ARG;
  if (field_is_stress(u_in) || !field_is_stress(s_out))
    return fail(h, SG_ERR_ARG, "sg_apply_G: u_in must be a velocity field, s_out a stress field");
  if (!h->params_set) return fail(h, SG_ERR_STATE, "sg_set_params must be called first");
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  int rc = run_op(h, 1, u_in, s_out, -1, 0, 0, 0, 0, SG_REGION_ALL, SG_FIELD_U, use_source != 0);
  if (rc == SG_OK && use_source) rc = add_source(h, s_out, 1.0);
  return rc;
}

// ---- halo ---------------------------------------------------------------------------------

int sg_halo_bytes(const sg_handle* h, int field, int side, size_t* nbytes) {
  if (!h || !nbytes || field < 0 || field > 3 || side < 0 || side >= 2 * h->cfg.dim) return SG_ERR_ARG;
  const int d = h->cfg.dim;
  int axis = side >> 1;
  size_t n2 = 1;
  for (int a = 0; a < 3; ++a)
    if (a != axis) n2 *= (size_t)h->cfg.n[a];
  // dim components per facet node for every field: a stress trace travels as T_i,axis (kernels.hip pack_one)
  *nbytes = n2 * h->md.halo_per_cube * h->re.nf * (size_t)d * (h->f32 ? sizeof(float) : sizeof(double));
  (void)field;
  return SG_OK;
}

// pack launches are timed like stage launches (event pairs resolved lazily; stage id 6 = halo pack)
static int pack_begin(sg_handle* h, size_t& k) {
  if (int rc = join_second(h)) return rc;
  k = h->ev_stage_ids.size();
  if (!h->timing) return SG_OK;
  if (k >= 8192) {
    int rc = resolve_timing(h);
    if (rc != SG_OK) return rc;
    k = 0;
  }
  while (h->ev_pool.size() < 2 * k + 2) {
    hipEvent_t e;
    HIPCHECK(h, hipEventCreate(&e));
    h->ev_pool.push_back(e);
  }
  HIPCHECK(h, hipEventRecord(h->ev_pool[2 * k], h->stream));
  return SG_OK;
}

static int pack_end(sg_handle* h, size_t k, size_t nbytes) {
  if (h->timing) {
    HIPCHECK(h, hipEventRecord(h->ev_pool[2 * k + 1], h->stream));
    h->ev_stage_ids.push_back(6);
  }
  h->counters.halo_pack_launches += 1;
  h->counters.halo_bytes_packed += (int64_t)nbytes;
  return SG_OK;
}

int sg_halo_pack(sg_handle* h, int field, int side, void* dev_out) {
  if (!h || !dev_out || field < 0 || field > 3 || side < 0 || side >= 2 * h->cfg.dim) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  const int d = h->cfg.dim;
  int comps = field_is_stress(field) ? d * d : d;
  void* out = dev_out;
  size_t k = 0, nb = 0;
  int rc = pack_begin(h, k);
  if (rc != SG_OK) return rc;
  rc = launch_pack(h->md_dev, h->md, h->field[field], comps, 1, &side, &out,
                   (h->sym && field_is_stress(field)) ? 1 : 0, h->f32, h->stream);
  if (rc != 0) return fail(h, SG_ERR_DEVICE, "pack kernel launch failed");
  (void)sg_halo_bytes(h, field, side, &nb);
  return pack_end(h, k, nb);
}

int sg_halo_pack_sides(sg_handle* h, int field, void* const* dev_out) {
  if (!h || !dev_out || field < 0 || field > 3) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  const int d = h->cfg.dim;
  int comps = field_is_stress(field) ? d * d : d;
  int sides[6], n = 0;
  void* outs[6];
  for (int s = 0; s < 2 * d; ++s)
    if (dev_out[s]) {
      sides[n] = s;
      outs[n] = dev_out[s];
      n += 1;
    }
  size_t k = 0, total = 0;
  int rc = pack_begin(h, k);
  if (rc != SG_OK) return rc;
  rc = launch_pack(h->md_dev, h->md, h->field[field], comps, n, sides, outs,
                   (h->sym && field_is_stress(field)) ? 1 : 0, h->f32, h->stream);
  if (rc != 0) return fail(h, SG_ERR_DEVICE, "pack kernel launch failed");
  for (int i = 0; i < n; ++i) {
    size_t nb = 0;
    (void)sg_halo_bytes(h, field, sides[i], &nb);
    total += nb;
  }
  return pack_end(h, k, total);
}

// ---- instrumentation ----------------------------------------------------------------------------

int sg_enable_timing(sg_handle* h, int on) {
  if (!h) return SG_ERR_ARG;
  if (!on) {
    int rc = resolve_timing(h);
    if (rc != SG_OK) return rc;
  }
  h->timing = on != 0;
  return SG_OK;
}

int sg_get_counters(sg_handle* h, sg_counters_t* out) {
  if (!h || !out) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  int rc = resolve_timing(h);
  if (rc != SG_OK) return rc;
  for (int st = 0; st < 6; ++st)      // a FIRST launch whose SECOND was never issued
    if (h->first_ms_pending[st] >= 0 && !h->second_pending) {
      h->counters.kernel_ms[st] += h->first_ms_pending[st];
      h->first_ms_pending[st] = -1;
    }
  *out = h->counters;
  return SG_OK;
}

int sg_stage_kernel_name(sg_handle* h, int stage, int region, char* buf, size_t n) {
  if (!h || !buf || n == 0) return SG_ERR_ARG;
  if (region < 0 || region > 4) return fail(h, SG_ERR_ARG, "unknown region");
  if (stage < 0 || stage > 5) return fail(h, SG_ERR_ARG, "unknown stage");
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  // the stage's own launch code with StageArgs::name_out set: the dispatch that picks the instantiation is the one
  // that would launch it (kernels.hpp SG_LAUNCH); nothing is queued on the stream
  std::string name;
  h->name_out = &name;
  const int rc = run_stage_impl(h, stage, region);
  h->name_out = nullptr;
  if (rc != SG_OK) return rc;
  std::snprintf(buf, n, "%s", name.c_str());
  return SG_OK;
}

}  // extern "C"
