// C-ABI of libseigen_hip.so (see include/seigen_hip.h): handle life cycle (device memory, operator tables, kernel
// family choice), parameters, sponge, source and the table exports.  Field transfers: transfer.cpp; stage launches,
// the fused six-launch LF4 step (seigen/elastic.py:283-313) and halo packs: stages.cpp.
#include <limits>
#include <string>
#include <unordered_map>

#include "handle.hpp"
#include "sponge_tables.hpp"

static void free_affine_sponge(sg_handle* h);

extern "C" {

const char* sg_last_error(const sg_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

void sg_destroy(sg_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)sync_all(h);
  for (int f = 0; f < 4; ++f)
    if (h->field[f]) (void)hipFree(h->field[f]);
  if (h->md_dev) (void)hipFree(h->md_dev);
  comm_release(h);
  if (h->mk_dev) (void)hipFree(h->mk_dev);
  if (h->ftab_dev) (void)hipFree(h->ftab_dev);
  if (h->fragQ) (void)hipFree(h->fragQ);
  if (h->fragP) (void)hipFree(h->fragP);
  if (h->nbr_tab) (void)hipFree(h->nbr_tab);
  if (h->Dt) (void)hipFree(h->Dt);
  if (h->Lt) (void)hipFree(h->Lt);
  if (h->fragF) (void)hipFree(h->fragF);
  if (h->fragG) (void)hipFree(h->fragG);
  if (h->fragL) (void)hipFree(h->fragL);
  if (h->staging) (void)hipFree(h->staging);
  for (int i = 0; i < 2; ++i) {
    if (h->pin[i]) (void)hipHostFree(h->pin[i]);
    if (h->dstage[i]) (void)hipFree(h->dstage[i]);
    if (h->xfer_ev[i]) (void)hipEventDestroy(h->xfer_ev[i]);
  }
  if (h->sym_flag) (void)hipFree(h->sym_flag);
  if (h->graph1) (void)hipGraphExecDestroy(h->graph1);
  if (h->graph8) (void)hipGraphExecDestroy(h->graph8);
  for (int r = 0; r < 5; ++r)
    if (h->region_items[r]) (void)hipFree(h->region_items[r]);
  if (h->dbg) {
    unsigned long long v[32];
    if (hipMemcpy(v, h->dbg, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess)
      for (int k = 0; k < 4; ++k)
        std::fprintf(stderr, "[seigen_hip stamps] %s mode %d: items %llu  cycles/item: setup %.0f volume %.0f lifts %.0f epilogue %.0f"
                             "   (lifts: neighbour set-up %.0f, facet 0 %.0f, facets 1-3 %.0f)\n",
                     k < 2 ? "F" : "G", k & 1, v[8 * k + 4], v[8 * k + 4] ? (double)v[8 * k + 0] / v[8 * k + 4] : 0.0,
                     v[8 * k + 4] ? (double)v[8 * k + 1] / v[8 * k + 4] : 0.0, v[8 * k + 4] ? (double)v[8 * k + 2] / v[8 * k + 4] : 0.0,
                     v[8 * k + 4] ? (double)v[8 * k + 3] / v[8 * k + 4] : 0.0, v[8 * k + 4] ? (double)v[8 * k + 5] / v[8 * k + 4] : 0.0,
                     v[8 * k + 4] ? (double)v[8 * k + 6] / v[8 * k + 4] : 0.0, v[8 * k + 4] ? (double)v[8 * k + 7] / v[8 * k + 4] : 0.0);
    (void)hipFree(h->dbg);
  }
  if (h->lam_d) (void)hipFree(h->lam_d);
  if (h->mu_d) (void)hipFree(h->mu_d);
  if (h->rho2_d) (void)hipFree(h->rho2_d);
  if (h->sponge_slot) (void)hipFree(h->sponge_slot);
  if (h->sponge_B) (void)hipFree(h->sponge_B);
  if (h->sponge_sigma) (void)hipFree(h->sponge_sigma);
  if (h->sponge_cells) (void)hipFree(h->sponge_cells);
  if (h->sponge_mat) (void)hipFree(h->sponge_mat);
  if (h->sponge_pre) (void)hipFree(h->sponge_pre);
  free_affine_sponge(h);
  if (h->src_nodes) (void)hipFree(h->src_nodes);
  if (h->src_values) (void)hipFree(h->src_values);
  if (h->src_slot_d) (void)hipFree(h->src_slot_d);
  if (h->src_idx_d) (void)hipFree(h->src_idx_d);
  if (h->src_ctr_d) (void)hipFree(h->src_ctr_d);
  if (h->src_weights_d) (void)hipFree(h->src_weights_d);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  if (h->stream2) {
    (void)hipStreamSynchronize(h->stream2);
    (void)hipStreamDestroy(h->stream2);
  }
  if (h->ev_stage) (void)hipEventDestroy(h->ev_stage);
  if (h->ev_second) (void)hipEventDestroy(h->ev_second);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

static int create_impl(const sg_config* cfg, sg_handle* h) {
  if (cfg->dim < 1 || cfg->dim > 3) return fail(h, SG_ERR_ARG, "dim must be 1, 2 or 3");
  if (cfg->degree < 1 || cfg->degree > 4) return fail(h, SG_ERR_ARG, "degree must be 1..4");
  for (int a = 0; a < cfg->dim; ++a) {
    if (cfg->n[a] < 1) return fail(h, SG_ERR_ARG, "n[axis] must be >= 1");
    if (!(cfg->h[a] > 0.0)) return fail(h, SG_ERR_ARG, "h[axis] must be > 0");
  }
  h->cfg = *cfg;
  for (int a = cfg->dim; a < 3; ++a) {
    h->cfg.n[a] = 1;
    h->cfg.h[a] = 1.0;
    h->cfg.origin[a] = 0.0;
    h->cfg.cube0[a] = 0;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(h, SG_ERR_DEVICE, "no HIP device available (libseigen_hip has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(h, SG_ERR_ARG, "device ordinal out of range");
  HIPCHECK(h, hipSetDevice(cfg->device));

  if (cfg->diagonal == SG_DIAGONAL_QUAD && cfg->dim == 1) return fail(h, SG_ERR_ARG, "tensor-product cells need dim 2 or 3");
  try {
    h->re = make_refelem(cfg->dim, cfg->degree, cfg->diagonal == SG_DIAGONAL_QUAD ? KIND_TENSOR : KIND_SIMPLEX);
    std::memset(&h->md, 0, sizeof(MeshDev));
    h->md.nd = h->re.nd;
    h->md.nf = h->re.nf;
    build_mesh_tables(cfg->dim, cfg->degree, cfg->diagonal, h->cfg.h, h->re.fnode.data(), h->re.lattice.data(), h->md);
  } catch (const std::exception& e) {
    return fail(h, SG_ERR_ARG, e.what());
  }
  for (int a = 0; a < 3; ++a) h->md.n[a] = h->cfg.n[a];
  for (int s = 0; s < 6; ++s) h->md.has_nbr[s] = (s < 2 * cfg->dim) ? ((cfg->nbr_mask >> s) & 1) : 0;
  h->ncls = h->md.ncls;
  h->ncells = (int64_t)h->cfg.n[0] * h->cfg.n[1] * h->cfg.n[2] * h->ncls;
  const KernelPath kp = choose_kernel_path(*cfg);
  h->use_mfma = kp.mfma;
  h->use_lane = kp.lane;
  h->use_tile = kp.tile;
  h->use_hexm = kp.hexm;
  if (cfg->dtype != 0 && cfg->dtype != 1) return fail(h, SG_ERR_ARG, "dtype must be 0 (f64) or 1 (f32)");
  h->f32 = cfg->dtype;
  if (h->f32 && !h->use_mfma && !h->use_tile)
    return fail(h, SG_ERR_ARG, "dtype f32 is implemented on the MFMA paths (3-D blocks: degree 1 from 65536 cells; 2-D "
                               "blocks on the tile kernels)");
  h->md.gw = kp.gw;
  h->md.ncube = (int64_t)h->cfg.n[0] * h->cfg.n[1] * h->cfg.n[2];
  h->md.ncube_pad = (h->md.ncube + h->md.gw - 1) / h->md.gw * h->md.gw;
  if (h->use_tile) h->t2c = tile2d_const(h->md);
  for (int f = 0; f < 4; ++f)
    for (int s = 0; s < 6; ++s) h->ghost[f][s] = nullptr;
  std::memset(&h->counters, 0, sizeof(h->counters));

  const int d = cfg->dim, nd = h->re.nd, nf = h->re.nf, nfaces = h->re.nfaces;
  // transposed operators: Dt[r][b][a], Lt[f][b'][a]
  std::vector<double> Dt((size_t)d * nd * nd), Lt((size_t)nfaces * nf * nd);
  for (int r = 0; r < d; ++r)
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b < nd; ++b) Dt[((size_t)r * nd + b) * nd + a] = h->re.D[((size_t)r * nd + a) * nd + b];
  for (int f = 0; f < nfaces; ++f)
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b < nf; ++b) Lt[((size_t)f * nf + b) * nd + a] = h->re.L[((size_t)f * nd + a) * nf + b];
  if (h->re.kind == KIND_TENSOR && cfg->dim == 3) {
    // hexahedra, both kernel families (kernels_lane.hip hex_stage, kernels.hip stage_kernel TP = 2): the 1-D factors D1 [n1][n1] and lift1 [2][n1] of the
    // tensor-product operators, read off the full tables along the first axis and checked against ALL of D_r, L_f
    const int n1 = cfg->degree + 1;
    auto node = [&](int a0, int a1, int a2) { return a0 + n1 * (a1 + n1 * a2); };
    Dt.assign((size_t)n1 * n1 + 2 * n1, 0.0);
    for (int m = 0; m < n1; ++m) {
      for (int n = 0; n < n1; ++n) Dt[(size_t)m * n1 + n] = h->re.D[((size_t)0 * nd + node(m, 0, 0)) * nd + node(n, 0, 0)];
      for (int s = 0; s < 2; ++s) Dt[(size_t)n1 * n1 + s * n1 + m] = h->re.L[((size_t)s * nd + node(m, 0, 0)) * nf + 0];
    }
    double worst = 0.0;
    for (int a = 0; a < nd; ++a) {
      const int ai[3] = {a % n1, (a / n1) % n1, a / (n1 * n1)};
      for (int r = 0; r < 3; ++r) {
        for (int b = 0; b < nd; ++b) {
          const int bi[3] = {b % n1, (b / n1) % n1, b / (n1 * n1)};
          const bool line = ai[(r + 1) % 3] == bi[(r + 1) % 3] && ai[(r + 2) % 3] == bi[(r + 2) % 3];
          const double want = line ? Dt[(size_t)ai[r] * n1 + bi[r]] : 0.0;
          worst = std::max(worst, std::fabs(h->re.D[((size_t)r * nd + a) * nd + b] - want));
        }
        for (int s = 0; s < 2; ++s)
          for (int bp = 0; bp < nf; ++bp) {
            const int b = h->re.fnode[(size_t)(2 * r + s) * nf + bp];
            const int bi[3] = {b % n1, (b / n1) % n1, b / (n1 * n1)};
            // facet nodes in ascending order: bp = lower transverse index + n1 * the upper one
            const int lo = r == 0 ? 1 : 0, hi = r == 2 ? 1 : 2;
            // ... and the neighbour's matching node is the one across the cube, at the same place of its facet list
            int ni[3] = {bi[0], bi[1], bi[2]};
            ni[r] = s ? 0 : n1 - 1;
            if (bp != bi[lo] + n1 * bi[hi] || bi[r] != (s ? n1 - 1 : 0) || h->md.fnode[2 * r + s][bp] != b ||
                h->md.nb_node[0][2 * r + s][bp] != node(ni[0], ni[1], ni[2]) || h->md.nb_fnode[0][2 * r + s][bp] != bp ||
                h->md.nb_face[0][2 * r + s] != 2 * r + (1 - s) || h->md.nb_axis[0][2 * r + s] != r)
              return fail(h, SG_ERR_ARG, "hexahedral element: unexpected facet node order");
            const bool same = ai[lo] == bi[lo] && ai[hi] == bi[hi];
            const double want = same ? Dt[(size_t)n1 * n1 + s * n1 + ai[r]] : 0.0;
            worst = std::max(worst, std::fabs(h->re.L[((size_t)(2 * r + s) * nd + a) * nf + bp] - want));
          }
      }
    }
    if (worst > 1e-11) return fail(h, SG_ERR_ARG, "hexahedral element: the operator tables do not factorise");
    if (h->use_hexm) {     // kernels_hexm.hip: line operators with the own-trace half folded in, x-pass A operands
      const std::vector<double> d1l(Dt);
      Dt = hexm_table(cfg->degree, d1l.data(), d1l.data() + (size_t)n1 * n1, h->md);
    }
  } else if (h->use_lane) {
    // lane path: E_r = D_r - (L_0 R_0 - L_{r+1} R_{r+1}) / (2 (d-1)!) row-major (own-trace half of the
    // central flux folded into the volume operator, see mfma_tables.cpp), and L_f row-major
    double fact = 1.0;
    for (int i = 2; i <= d - 1; ++i) fact *= i;
    const double cfold = 1.0 / (2.0 * fact);
    for (int r = 0; r < d; ++r)
      for (int a = 0; a < nd; ++a)
        for (int b = 0; b < nd; ++b) {
          double v = h->re.D[((size_t)r * nd + a) * nd + b];
          for (int bf = 0; bf < nf; ++bf) {
            if (h->re.fnode[(size_t)0 * nf + bf] == b) v -= cfold * h->re.L[((size_t)0 * nd + a) * nf + bf];
            if (h->re.fnode[(size_t)(r + 1) * nf + bf] == b) v += cfold * h->re.L[((size_t)(r + 1) * nd + a) * nf + bf];
          }
          Dt[((size_t)r * nd + a) * nd + b] = v;
        }
    for (size_t i = 0; i < Lt.size(); ++i) Lt[i] = h->re.L[i];
  }
  HIPCHECK(h, hipMalloc((void**)&h->Dt, Dt.size() * sizeof(double)));
  HIPCHECK(h, hipMalloc((void**)&h->Lt, Lt.size() * sizeof(double)));
  HIPCHECK(h, hipMalloc((void**)&h->md_dev, sizeof(MeshDev)));
  HIPCHECK(h, hipMemcpy(h->Dt, Dt.data(), Dt.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHECK(h, hipMemcpy(h->Lt, Lt.data(), Lt.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHECK(h, hipMemcpy(h->md_dev, &h->md, sizeof(MeshDev), hipMemcpyHostToDevice));
  if (h->use_mfma) {
    const MfmaConst mk = mfma_const(h->md);
    HIPCHECK(h, hipMalloc((void**)&h->mk_dev, sizeof(MfmaConst)));
    HIPCHECK(h, hipMemcpy(h->mk_dev, &mk, sizeof(MfmaConst), hipMemcpyHostToDevice));
    {
      std::vector<int32_t> ft;
      mfma_trace_offsets(h->md, 9, ft);
      HIPCHECK(h, hipMalloc((void**)&h->ftab_dev, ft.size() * sizeof(int32_t)));
      HIPCHECK(h, hipMemcpy(h->ftab_dev, ft.data(), ft.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if ((h->md.ncube_pad / 16) * 6 * 16 >= ((int64_t)1 << 31))     // cell slots are int32 (288 GB hold far fewer cells)
      return fail(h, SG_ERR_ARG, "block too large for the MFMA path's neighbour table");
    {
      std::vector<int32_t> tab;
      build_nbr_table(h->md, tab);
      HIPCHECK(h, hipMalloc((void**)&h->nbr_tab, tab.size() * sizeof(int32_t)));
      HIPCHECK(h, hipMemcpy(h->nbr_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
  }

  for (int f = 0; f < 4; ++f) {
    size_t comps = field_is_stress(f) ? (size_t)d * d : (size_t)d;
    h->field_len[f] = (size_t)h->ncells * nd * comps;
    h->field_alloc[f] = (size_t)h->md.ncube_pad * h->ncls * nd * comps;
    const size_t es = h->f32 ? sizeof(float) : sizeof(double);
    if (hipMalloc((void**)&h->field[f], h->field_alloc[f] * es) != hipSuccess)
      return fail(h, SG_ERR_NOMEM, "hipMalloc of a field buffer failed");
    HIPCHECK(h, hipMemset(h->field[f], 0, h->field_alloc[f] * es));
  }
  if ((h->use_mfma || h->use_tile) && h->f32) {
    std::vector<float> fF, fG, fL;
    if (h->use_tile) {
      fF = tile2d_frags32_V(h->re, -1.0);
      fG = tile2d_frags32_V(h->re, 1.0);
      fL = tile2d_frags32_L(h->re);
    } else {
      fF = mfma32_frags_F(h->re);
      fG = mfma32_frags_G(h->re);
      fL = mfma32_frags_L(h->re);
    }
    HIPCHECK(h, hipMalloc((void**)&h->fragF, fF.size() * sizeof(float)));
    HIPCHECK(h, hipMalloc((void**)&h->fragG, fG.size() * sizeof(float)));
    HIPCHECK(h, hipMalloc((void**)&h->fragL, fL.size() * sizeof(float)));
    HIPCHECK(h, hipMemcpy(h->fragF, fF.data(), fF.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(h->fragG, fG.data(), fG.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(h->fragL, fL.data(), fL.size() * sizeof(float), hipMemcpyHostToDevice));
  } else if (h->use_mfma || h->use_tile) {
    std::vector<double> fF, fG, fL;
    if (h->use_tile) {
      fF = tile2d_frags_V(h->re, -1.0);
      fG = tile2d_frags_V(h->re, 1.0);
      fL = tile2d_frags_L(h->re);
    } else {
      fF = mfma_frags_F(h->re);
      fG = mfma_frags_G(h->re);
      fL = mfma_frags_L(h->re);
    }
    HIPCHECK(h, hipMalloc((void**)&h->fragF, fF.size() * sizeof(double)));
    HIPCHECK(h, hipMalloc((void**)&h->fragG, fG.size() * sizeof(double)));
    HIPCHECK(h, hipMalloc((void**)&h->fragL, fL.size() * sizeof(double)));
    HIPCHECK(h, hipMemcpy(h->fragF, fF.data(), fF.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(h->fragG, fG.data(), fG.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(h->fragL, fL.data(), fL.size() * sizeof(double), hipMemcpyHostToDevice));
    // factorised G volume (mfma_tables.hpp): D_r = P_r Q
    const char* gq = std::getenv("SEIGEN_HIP_GQ");
    // default: degree 4 only (G<4,0> -2 %, step -0.7 .. -1 %; at degree 3, rank 10 of 20 on 4-row tiles, it is 12 % slower:
    // profiles/r04/kernel_experiments.txt); SEIGEN_HIP_GQ=0 / 1 forces it off / on for degrees 3 and 4
    if (h->use_mfma && cfg->degree >= 3 && (gq ? std::atoi(gq) != 0 : cfg->degree >= SG_GQ_FROM_DEGREE)) {
      std::vector<double> fQ, fP;
      try {
        fQ = mfma_frags_Q(h->re);
        fP = mfma_frags_P(h->re);
      } catch (const std::exception& e) {
        return fail(h, SG_ERR_ARG, e.what());
      }
      HIPCHECK(h, hipMalloc((void**)&h->fragQ, fQ.size() * sizeof(double)));
      HIPCHECK(h, hipMalloc((void**)&h->fragP, fP.size() * sizeof(double)));
      HIPCHECK(h, hipMemcpy(h->fragQ, fQ.data(), fQ.size() * sizeof(double), hipMemcpyHostToDevice));
      HIPCHECK(h, hipMemcpy(h->fragP, fP.data(), fP.size() * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  if (h->use_mfma || h->use_lane || h->use_tile || h->use_hexm) {
    // symmetric-stress mode: fields start at zero, g only produces symmetric tensors;
    // left for good as soon as the user uploads a non-symmetric stress or source (SEIGEN_HIP_SYM=0: never entered)
    const char* sym_env = std::getenv("SEIGEN_HIP_SYM");
    h->sym = !(sym_env && std::strcmp(sym_env, "0") == 0);
    HIPCHECK(h, hipMalloc((void**)&h->sym_flag, sizeof(int)));
    HIPCHECK(h, hipMemset(h->sym_flag, 0, sizeof(int)));
  }
  if (std::getenv("SEIGEN_HIP_STAMPS")) {
    HIPCHECK(h, hipMalloc((void**)&h->dbg, 32 * sizeof(unsigned long long)));
    HIPCHECK(h, hipMemset(h->dbg, 0, 32 * sizeof(unsigned long long)));
  }
  {
    // Persistent grid of the MFMA stage kernels: two blocks per CU fill every CU (registers and
    // LDS allow exactly two).  A block with halo neighbours leaves 1/16 of those slots empty, so
    // that RCCL's send/receive kernels can start WHILE an interior launch runs: behind a full
    // grid they only start when it drains (tools/overlap_probe.py), and a stage kernel that found
    // some of its own slots taken would run the late blocks' static shares one after the other.
    hipDeviceProp_t prop;
    HIPCHECK(h, hipGetDeviceProperties(&prop, cfg->device));
    const int slots = (h->use_mfma ? mfma_blocks_per_cu(cfg->degree, h->f32) : (h->use_hexm ? hexm_blocks_per_cu(cfg->degree) : 2)) *
                      prop.multiProcessorCount;
    h->grid_full = slots / 8 * 8;
    h->grid_blocks = (cfg->nbr_mask != 0 ? slots - slots / 16 : slots) / 8 * 8;
    if (const char* gb = std::getenv("SEIGEN_HIP_GRID_BLOCKS")) h->grid_blocks = std::max(8, std::atoi(gb) / 8 * 8);
    if (cfg->nbr_mask == 0) h->grid_full = h->grid_blocks;
    // F stages of a whole 3-D block: items dealt to the XCDs in chunks of 1/8 of a z-layer of cubes, so that all XCDs
    // sweep the block layer by layer together (the z-neighbour traces then meet the own rows of the next layer in the
    // Infinity Cache: F stages -3 %, profiles/r03/order_chunk_sweep.txt; the G stages do not gain and keep one
    // contiguous range per XCD).  SEIGEN_HIP_ORDER_CHUNK overrides (0 = off).
    h->order_chunk = 0;
    if (h->use_mfma && cfg->dim == 3) {
      const int64_t per_layer = ((int64_t)cfg->n[0] * cfg->n[1] + 15) / 16 * 6;
      if (cfg->n[2] >= 16) h->order_chunk = (int)std::max<int64_t>(6, (per_layer + 7) / 8);
    }
    if (const char* oc = std::getenv("SEIGEN_HIP_ORDER_CHUNK")) h->order_chunk = std::max(0, std::atoi(oc));
    h->no_whole = std::getenv("SEIGEN_HIP_NO_WHOLE") != nullptr;
    // 2-D tile kernels: a persistent grid of exactly the blocks the device holds of the stage's kernel (0 = the launcher
    // asks the runtime per instantiation, kernels_tile2d.hip) - a wave sets up once and works through its share of the
    // items: 2-D P4 at N = 256 (the reference's benchmark protocol) 84.5 -> 93.4 G DoF-updates/s against one item per
    // wave, P3 +8 %, level elsewhere (profiles/r05/tile_grid_sweep.txt).  With a sponge the items differ in cost and a
    // static share can collect the expensive ones: then 251 blocks per XCD label - with an odd (prime) stride of 4 * 251
    // items a wave's items do not keep falling on the same column of the mesh, i.e. on the sponge strips at both ends of
    // every row (config 2: 0.228 ms per step with 768 blocks, 0.213 with 2008).
    h->tile_grid = 0;
    h->tile_grid_sponge = 2008;
    if (const char* tg = std::getenv("SEIGEN_HIP_TILE_GRID")) h->tile_grid = h->tile_grid_sponge = std::max(8, std::atoi(tg) / 8 * 8);
  }
  {
    const char* ge = std::getenv("SEIGEN_HIP_GRAPH");  // 0/1 overrides (measurements)
    const int64_t dofs = h->ncells * (int64_t)h->re.nd * (cfg->dim + cfg->dim * cfg->dim);
    h->graph_ok = ge ? (std::strcmp(ge, "0") != 0) : (dofs <= (int64_t)1 << 23);
  }
  {
    const char* ov = std::getenv("SEIGEN_HIP_OVERLAP");
    h->overlap = cfg->nbr_mask != 0 && !(ov && std::strcmp(ov, "0") == 0);
  }
  int prio_lo = 0, prio_hi = 0;
  HIPCHECK(h, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  if (cfg->stream) {
    h->stream = (hipStream_t)cfg->stream;
  } else {
    HIPCHECK(h, hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, h->overlap ? prio_hi : 0));
    h->own_stream = true;
  }
  if (h->overlap) {
    HIPCHECK(h, hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio_lo));
    HIPCHECK(h, hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming));
    HIPCHECK(h, hipEventCreateWithFlags(&h->ev_second, hipEventDisableTiming));
  }
  HIPCHECK(h, hipEventCreate(&h->ev0));
  HIPCHECK(h, hipEventCreate(&h->ev1));
  HIPCHECK(h, hipDeviceSynchronize());
  return SG_OK;
}

int sg_create(const sg_config* cfg, sg_handle** out) {
  if (!cfg || !out) {
    g_create_err = "null argument";
    return SG_ERR_ARG;
  }
  *out = nullptr;
  sg_handle* h = new sg_handle();
  int rc = create_impl(cfg, h);
  if (rc != SG_OK) {
    g_create_err = h->err;
    sg_destroy(h);
    return rc;
  }
  *out = h;
  return SG_OK;
}

int sg_get_stream(const sg_handle* h, void** stream) {
  if (!h || !stream) return SG_ERR_ARG;
  *stream = (void*)h->stream;
  return SG_OK;
}

int sg_get_second_stream(const sg_handle* h, void** stream) {
  if (!h || !stream) return SG_ERR_ARG;
  *stream = (void*)h->stream2;
  return SG_OK;
}

int sg_get_info(const sg_handle* h, sg_info_t* out) {
  if (!h || !out) return SG_ERR_ARG;
  const int d = h->cfg.dim;
  out->dim = d;
  out->degree = h->cfg.degree;
  out->nd = h->re.nd;
  out->nf = h->re.nf;
  out->nfaces = h->re.nfaces;
  out->nclasses = h->ncls;
  out->ncells = h->ncells;
  out->u_dofs = (int64_t)d * h->re.nd * h->ncells;
  out->s_dofs = (int64_t)d * d * h->re.nd * h->ncells;
  for (int s = 0; s < 6; ++s) {
    int axis = s >> 1;
    if (axis >= d) {
      out->halo_faces[s] = 0;
      continue;
    }
    int64_t n2 = 1;
    for (int a = 0; a < 3; ++a)
      if (a != axis) n2 *= h->cfg.n[a];
    out->halo_faces[s] = (int32_t)(n2 * h->md.halo_per_cube);
  }
  return SG_OK;
}

int sg_sync(sg_handle* h) {
  if (!h) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (int rc = join_second(h)) return rc;
  HIPCHECK(h, sync_all(h));
  return SG_OK;
}

int sg_node_coords(const sg_handle* h, int degree, double* out, size_t nbytes) {
  if (!h) return SG_ERR_ARG;
  return sg_block_node_coords(&h->cfg, degree, out, nbytes);
}

int sg_set_params(sg_handle* h, double density, double dt, const double* lambda, const double* mu, int per_cell) {
  if (h) h->epoch += 1;
  if (!h || !lambda || !mu) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  h->rho = density;
  h->rho_physical = 0;
  if (h->rho2_d) {
    HIPCHECK(h, sync_all(h));
    (void)hipFree(h->rho2_d);
    h->rho2_d = nullptr;
  }
  h->dt = dt;
  h->per_cell = per_cell ? 1 : 0;
  if (per_cell) {
    size_t nb = (size_t)h->ncells * sizeof(double);
    if (!h->lam_d) HIPCHECK(h, hipMalloc((void**)&h->lam_d, nb));
    if (!h->mu_d) HIPCHECK(h, hipMalloc((void**)&h->mu_d, nb));
    HIPCHECK(h, sync_all(h));
    HIPCHECK(h, hipMemcpy(h->lam_d, lambda, nb, hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(h->mu_d, mu, nb, hipMemcpyHostToDevice));
    h->lam0 = lambda[0];
    h->mu0 = mu[0];
  } else {
    h->lam0 = lambda[0];
    h->mu0 = mu[0];
  }
  h->params_set = true;
  return SG_OK;
}

int sg_set_density(sg_handle* h, const double* rho, int per_cell, int physical) {
  if (h) h->epoch += 1;
  if (!h || !rho) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  HIPCHECK(h, sync_all(h));
  if (h->rho2_d) {
    (void)hipFree(h->rho2_d);
    h->rho2_d = nullptr;
  }
  h->rho_physical = physical ? 1 : 0;
  if (!per_cell) {
    if (physical && rho[0] == 0.0) return fail(h, SG_ERR_ARG, "sg_set_density: zero density");
    h->rho = rho[0];
    return SG_OK;
  }
  std::vector<double> r2((size_t)h->ncells * 2);
  for (int64_t e = 0; e < h->ncells; ++e) {
    if (physical && rho[e] == 0.0) return fail(h, SG_ERR_ARG, "sg_set_density: zero density");
    r2[2 * (size_t)e] = physical ? 1.0 : rho[e];
    r2[2 * (size_t)e + 1] = physical ? 1.0 / rho[e] : 1.0;
  }
  h->rho = rho[0];
  HIPCHECK(h, hipMalloc((void**)&h->rho2_d, r2.size() * sizeof(double)));
  HIPCHECK(h, hipMemcpy(h->rho2_d, r2.data(), r2.size() * sizeof(double), hipMemcpyHostToDevice));
  return SG_OK;
}

static void free_affine_sponge(sg_handle* h) {
  for (void** p : {(void**)&h->sponge_mat_slots, (void**)&h->sponge_aff_items, (void**)&h->sponge_aff_slots, (void**)&h->sponge_aff_coef,
                   (void**)&h->sponge_aff_X, (void**)&h->sponge_aff_col, (void**)&h->sponge_aff_frag})
    if (*p) {
      (void)hipFree(*p);
      *p = nullptr;
    }
  h->sponge_nmat_slots = 0;
  h->sponge_aff_nitems = 0;
  h->sponge_aff_W = 0;
}

// What each cell gets - nothing, a scalar, dim + 1 numbers, a matrix - is decided by plan_sponge (sponge_tables.cpp: plain
// C++, under the CPU sanitizers); this function frees the old tables and uploads the new ones.
int sg_set_absorption(sg_handle* h, const double* sigma_nodes, int sigma_degree) {
  if (h) h->epoch += 1;
  if (!h) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  HIPCHECK(h, sync_all(h));
  for (void** p : {(void**)&h->sponge_slot, (void**)&h->sponge_B, (void**)&h->sponge_sigma, (void**)&h->sponge_cells,
                   (void**)&h->sponge_mat, (void**)&h->sponge_pre})
    if (*p) {
      (void)hipFree(*p);
      *p = nullptr;
    }
  h->sponge_nslots = 0;
  h->sponge_pre_key = -1;
  h->sponge_pre_regions = 0;
  h->sponge_pre_ver = ~0ull;
  h->sponge_pre_field = -1;
  h->sponge_pre_lines = 0;
  free_affine_sponge(h);
  if (!sigma_nodes) return SG_OK;
  if (sigma_degree < 1 || sigma_degree > 6) return fail(h, SG_ERR_ARG, "sigma_degree must be 1..6");
  const int d = h->cfg.dim, nd = h->re.nd;
  SpongeRequest rq;
  rq.dim = d;
  rq.degree = h->cfg.degree;
  rq.kind = h->re.kind;
  rq.sigma_degree = sigma_degree;
  rq.ncells = h->ncells;
  rq.ncls = h->ncls;
  rq.gw = (int)h->md.gw;
  // 2-D tile and 3-D matrix kernels, lane kernels: a sigma that is one value on all nodes of a cell (the piecewise-constant
  // sponges of the reference's problem scripts, explosive_source_lf4.py:42-45) is applied as sigma u at the node
  rq.want_scalar = h->use_tile || h->use_mfma || h->use_hexm || h->use_lane;
  // the 3-D matrix kernels and the lane kernels read B u_abs from a pre-pass (kernels.hpp launch_sponge_pre); the 2-D tile
  // kernels work their small matrices off themselves: a launch more per F stage costs them more
  rq.pre_family = h->use_mfma || h->use_hexm || h->use_lane;
  // Affine cells take dim + 1 numbers (kernels.hip sponge_pre_affine_kernel, kernels_mfma.hip sponge_affine_mfma);
  // SEIGEN_HIP_SPONGE_AFFINE=0 sends them through their matrices (tests: the two must agree).  The lane kernels' cells -
  // hexahedra DQ_1 / DQ_2, gw = 64 - have matrices of at most 27 x 27 shared through the caches: there the matrix pre-pass
  // is the faster one (64.5 against 61.9 G at 96^3 DQ_2, profiles/r06/affine_sponge.txt); '1' forces the affine path
  const char* aff_env = std::getenv("SEIGEN_HIP_SPONGE_AFFINE");
  rq.try_affine = rq.pre_family && !(aff_env && aff_env[0] == '0') && (!h->use_lane || (aff_env && aff_env[0] == '1'));
  // 3-D MFMA family: the pre-pass results live in LINE layout like the fields (a record per cell cost the affine pre-pass
  // scattered 24-byte stores and the F stage scattered loads)
  rq.line_layout = h->use_mfma;
  SpongePlan pl;
  try {
    pl = plan_sponge(rq, sigma_nodes);
  } catch (const std::exception& e) {
    return fail(h, SG_ERR_ARG, std::string("sg_set_absorption: ") + e.what());
  }
  auto up = [&](void** dst, const void* src, size_t bytes) {
    if (hipMalloc(dst, bytes ? bytes : 8) != hipSuccess) return false;
    return bytes == 0 || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  bool ok = up((void**)&h->sponge_slot, pl.slot.data(), pl.slot.size() * sizeof(int32_t));
  h->sponge_nslots = pl.nslots;
  h->sponge_pre_lines = (rq.line_layout && pl.nslots > 0) ? 1 : 0;
  if (rq.pre_family && pl.nslots > 0) {
    const size_t pre_bytes = (size_t)pl.nslots * nd * d * (h->f32 ? sizeof(float) : sizeof(double));
    ok = ok && up((void**)&h->sponge_cells, pl.cells.data(), pl.cells.size() * sizeof(int32_t)) &&
         up((void**)&h->sponge_mat, pl.mat_of.data(), pl.mat_of.size() * sizeof(int32_t)) && hipMalloc(&h->sponge_pre, pre_bytes) == hipSuccess &&
         hipMemset(h->sponge_pre, 0, pre_bytes) == hipSuccess;
    // the cells with a matrix: every slot in order, or - line layout, affine cells among them - a list
    h->sponge_nmat_slots = (int32_t)pl.mat_slots.size();
    if (h->sponge_pre_lines || pl.naffine > 0)
      ok = ok && up((void**)&h->sponge_mat_slots, pl.mat_slots.data(), pl.mat_slots.size() * sizeof(int32_t));
  }
  if (rq.pre_family && pl.naffine > 0) {
    const size_t lds = sponge_pre_affine_lds(pl.W, !pl.dense, nd, d, rq.gw);
    if (lds > ((size_t)150 << 10))
      return fail(h, SG_ERR_STATE, "sg_set_absorption: the affine-sigma tables of this element do not fit the LDS (set SEIGEN_HIP_SPONGE_AFFINE=0)");
    if (prepare_sponge_pre_affine(d, h->f32, lds) != 0)
      return fail(h, SG_ERR_DEVICE, "sg_set_absorption: the affine-sigma pre-pass cannot have its LDS");
    ok = ok && up((void**)&h->sponge_aff_items, pl.items.data(), pl.items.size() * sizeof(int32_t)) &&
         up((void**)&h->sponge_aff_slots, pl.item_slots.data(), pl.item_slots.size() * sizeof(int32_t)) &&
         up((void**)&h->sponge_aff_coef, pl.aff_coef.data(), pl.aff_coef.size() * sizeof(double)) &&
         up((void**)&h->sponge_aff_X, pl.X.data(), pl.X.size() * sizeof(double)) &&
         (pl.dense || up((void**)&h->sponge_aff_col, pl.col.data(), pl.col.size() * sizeof(int32_t)));
    if (h->use_mfma && !h->f32 && d == 3) {      // on the matrix pipe (kernels_mfma.hip sponge_affine_mfma)
      const std::vector<double> fX = mfma_frags_dense(h->re, pl.Xd.data(), 3);
      ok = ok && up((void**)&h->sponge_aff_frag, fX.data(), fX.size() * sizeof(double)) && prepare_sponge_affine_mfma(h->cfg.degree) == 0;
    }
    h->sponge_aff_nitems = (int32_t)pl.items.size();
    h->sponge_aff_W = pl.W;
  }
  if (!pl.sig.empty()) ok = ok && up((void**)&h->sponge_sigma, pl.sig.data(), pl.sig.size() * sizeof(double));
  ok = ok && up((void**)&h->sponge_B, pl.B.data(), pl.B.size() * sizeof(double));
  if (!ok) return fail(h, SG_ERR_NOMEM, "sg_set_absorption: hipMalloc / upload of the sponge tables failed");
  return SG_OK;
}

int sg_set_source(sg_handle* h, int64_t nnz, const int64_t* nodes, int64_t nsteps, const double* values) {
  if (h) h->epoch += 1;
  if (!h || nnz < 0) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  HIPCHECK(h, sync_all(h));
  if (h->src_nodes) {
    (void)hipFree(h->src_nodes);
    h->src_nodes = nullptr;
  }
  if (h->src_values) {
    (void)hipFree(h->src_values);
    h->src_values = nullptr;
  }
  if (h->src_slot_d) {
    (void)hipFree(h->src_slot_d);
    h->src_slot_d = nullptr;
  }
  if (h->src_idx_d) {
    (void)hipFree(h->src_idx_d);
    h->src_idx_d = nullptr;
  }
  if (h->src_weights_d) {
    (void)hipFree(h->src_weights_d);
    h->src_weights_d = nullptr;
  }
  h->src_fused = false;
  h->src_nnz = 0;
  h->src_nsteps = 0;
  h->src_step = 0;
  h->src_static = false;
  h->src_weights.clear();
  if (nnz == 0 || nsteps == 0) return SG_OK;
  if (!nodes || !values || nsteps < -1) return SG_ERR_ARG;
  const bool is_static = nsteps == -1;
  if (is_static) nsteps = 1;
  const int d = h->cfg.dim;
  int64_t nscalar = h->ncells * h->re.nd;
  for (int64_t k = 0; k < nnz; ++k)
    if (nodes[k] < 0 || nodes[k] >= nscalar) return fail(h, SG_ERR_ARG, "sg_set_source: node index out of range");
  // A node listed more than once: its entries add up (in the order listed), merged here once so that every node is
  // written by one thread - the sum is then the same on every run and on every partition of the mesh (an atomic add
  // per entry gave the right sum in an arbitrary order, i.e. results that differed in the last bit from run to run).
  std::vector<int64_t> merged_nodes;
  std::vector<double> merged_values;
  {
    std::unordered_map<int64_t, int64_t> slot_of;
    slot_of.reserve((size_t)nnz * 2);
    std::vector<int64_t> to((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) {
      auto it = slot_of.find(nodes[k]);
      if (it == slot_of.end()) {
        it = slot_of.emplace(nodes[k], (int64_t)merged_nodes.size()).first;
        merged_nodes.push_back(nodes[k]);
      }
      to[(size_t)k] = it->second;
    }
    if ((int64_t)merged_nodes.size() != nnz) {
      const int64_t nm = (int64_t)merged_nodes.size(), dd = (int64_t)d * d;
      merged_values.assign((size_t)(nsteps * nm * dd), 0.0);
      for (int64_t s = 0; s < nsteps; ++s)
        for (int64_t k = 0; k < nnz; ++k)
          for (int64_t c = 0; c < dd; ++c) merged_values[(size_t)((s * nm + to[(size_t)k]) * dd + c)] += values[(s * nnz + k) * dd + c];
      nodes = merged_nodes.data();
      values = merged_values.data();
      nnz = nm;
    }
  }
  if (h->sym) {
    bool symmetric = true;
    for (int64_t i = 0; i < nsteps * nnz && symmetric; ++i)
      for (int a = 0; a < d; ++a)
        for (int b = a + 1; b < d; ++b) symmetric = symmetric && values[i * d * d + a * d + b] == values[i * d * d + b * d + a];
    if (!symmetric) {
      int rc = leave_sym_mode(h);
      if (rc != SG_OK) return rc;
    }
  }
  size_t vbytes = (size_t)nsteps * nnz * d * d * sizeof(double);
  HIPCHECK(h, hipMalloc((void**)&h->src_nodes, (size_t)nnz * sizeof(int64_t)));
  HIPCHECK(h, hipMalloc((void**)&h->src_values, vbytes));
  // Order the nodes so that those in cells of SG_REGION_FIRST come first: a split stage adds the
  // source to each part right after the launch that wrote it (the traces of FIRST are packed
  // before SECOND has run).  Then: device offset of component 0 of each node in the field layout.
  std::vector<int64_t> order((size_t)nnz), offs((size_t)nnz);
  {
    const int64_t nd = h->re.nd, ncls = h->ncls, gw = h->md.gw, nc = (int64_t)d * d;
    std::vector<Box> first;
    region_boxes(h, SG_REGION_FIRST, first);
    auto in_first = [&](int64_t node) {
      const int64_t cube = node / nd / ncls;
      const int64_t c[3] = {cube % h->cfg.n[0], (cube / h->cfg.n[0]) % h->cfg.n[1], cube / ((int64_t)h->cfg.n[0] * h->cfg.n[1])};
      for (const Box& b : first) {
        bool in = true;
        for (int k = 0; k < 3; ++k) in = in && c[k] >= b.o[k] && c[k] < b.o[k] + b.n[k];
        if (in) return true;
      }
      return false;
    };
    int64_t n1 = 0;
    for (int64_t i = 0; i < nnz; ++i)
      if (in_first(nodes[i])) order[(size_t)n1++] = i;
    h->src_nfirst = n1;
    for (int64_t i = 0; i < nnz; ++i)
      if (!in_first(nodes[i])) order[(size_t)n1++] = i;
    for (int64_t j = 0; j < nnz; ++j) {
      const int64_t node = nodes[order[(size_t)j]];
      int64_t e = node / nd, b = node % nd;
      int64_t cube = e / ncls, cls = e % ncls;
      offs[(size_t)j] = ((((cube / gw) * ncls + cls) * nd + b) * nc) * gw + cube % gw;
    }
  }
  std::vector<double> vals((size_t)nsteps * nnz * d * d);
  for (int64_t k = 0; k < nsteps; ++k)
    for (int64_t j = 0; j < nnz; ++j)
      std::memcpy(&vals[((size_t)k * nnz + j) * d * d], &values[((size_t)k * nnz + order[(size_t)j]) * d * d], sizeof(double) * d * d);
  HIPCHECK(h, hipMemcpy(h->src_nodes, offs.data(), (size_t)nnz * sizeof(int64_t), hipMemcpyHostToDevice));
  HIPCHECK(h, hipMemcpy(h->src_values, vals.data(), vbytes, hipMemcpyHostToDevice));
  if (h->use_tile && !std::getenv("SEIGEN_HIP_SOURCE_LAUNCH")) {
    // tile kernels: item (16 squares of one class) -> slot, and per slot a dense (node, cell) -> value-row table, so
    // that the G stages add the source themselves (one launch less per G stage).  (Nodes are unique here: entries of a
    // node listed twice were merged above.)
    const int64_t nd = h->re.nd, ncl = h->ncls, nitems = h->md.ncube_pad / 16 * ncl;
    std::vector<int32_t> slot((size_t)nitems, -1), idx;
    bool dup = false;
    for (int64_t j = 0; j < nnz && !dup; ++j) {
      const int64_t node = nodes[order[(size_t)j]];
      const int64_t e = node / nd, b = node % nd, cube = e / ncl, cls = e % ncl;
      const int64_t item = (cube / 16) * ncl + cls;
      if (slot[(size_t)item] < 0) {
        slot[(size_t)item] = (int32_t)(idx.size() / (size_t)(nd * 16));
        idx.resize(idx.size() + (size_t)(nd * 16), -1);
      }
      int32_t& cell = idx[((size_t)slot[(size_t)item] * nd + b) * 16 + cube % 16];
      dup = cell >= 0;
      cell = (int32_t)j;
    }
    if (!dup) {
      HIPCHECK(h, hipMalloc((void**)&h->src_slot_d, slot.size() * sizeof(int32_t)));
      HIPCHECK(h, hipMalloc((void**)&h->src_idx_d, idx.size() * sizeof(int32_t)));
      HIPCHECK(h, hipMemcpy(h->src_slot_d, slot.data(), slot.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      HIPCHECK(h, hipMemcpy(h->src_idx_d, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      h->src_fused = true;
    }
  }
  if (!h->src_ctr_d) {   // device-side step counter for graph replay (stages.cpp sg_step)
    HIPCHECK(h, hipMalloc((void**)&h->src_ctr_d, sizeof(int64_t)));
    HIPCHECK(h, hipMemset(h->src_ctr_d, 0, sizeof(int64_t)));
  }
  h->src_nnz = nnz;
  h->src_nsteps = nsteps;
  h->src_static = is_static;
  return SG_OK;
}

int sg_set_source_separable(sg_handle* h, int64_t nnz, const int64_t* nodes, const double* pattern, int64_t nsteps,
                            const double* weights) {
  if (!h || nnz < 0 || nsteps < 0) return SG_ERR_ARG;
  if (nnz == 0 || nsteps == 0) return sg_set_source(h, 0, nullptr, 0, nullptr);
  if (!weights) return SG_ERR_ARG;
  int rc = sg_set_source(h, nnz, nodes, 1, pattern);      // one slice: order, offsets, symmetry check, fused tables
  if (rc != SG_OK) return rc;
  h->src_weights.assign(weights, weights + nsteps);
  h->src_nsteps = nsteps;
  HIPCHECK(h, hipMalloc((void**)&h->src_weights_d, (size_t)nsteps * sizeof(double)));
  HIPCHECK(h, hipMemcpy(h->src_weights_d, weights, (size_t)nsteps * sizeof(double), hipMemcpyHostToDevice));
  return SG_OK;
}

int sg_set_source_box_ricker(sg_handle* h, const double* lo, const double* hi, double a, double t0, double t_first,
                             double dt_step, int64_t nsteps) {
  if (!h || !lo || !hi || nsteps < 0) return SG_ERR_ARG;
  const int d = h->cfg.dim;
  NodeGeom G;
  if (!G.init(&h->cfg, h->cfg.degree)) return SG_ERR_ARG;
  // cubes that can hold a node of the box: those overlapping it (closed on both sides)
  int c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
  for (int i = 0; i < d; ++i) {
    if (!(lo[i] <= hi[i])) return fail(h, SG_ERR_ARG, "source box: lo must not exceed hi");
    const double t0c = std::floor((lo[i] - h->cfg.origin[i]) / h->cfg.h[i]) - (double)h->cfg.cube0[i] - 1.0;
    const double t1c = std::floor((hi[i] - h->cfg.origin[i]) / h->cfg.h[i]) - (double)h->cfg.cube0[i] + 1.0;
    c0[i] = (int)std::max(0.0, std::min(t0c, (double)h->cfg.n[i]));
    c1[i] = (int)std::max(-1.0, std::min(t1c, (double)h->cfg.n[i] - 1.0));
  }
  std::vector<int64_t> nodes;
  for (int ck = c0[2]; ck <= c1[2]; ++ck)
    for (int cj = c0[1]; cj <= c1[1]; ++cj)
      for (int ci = c0[0]; ci <= c1[0]; ++ci) {
        const int c[3] = {ci, cj, ck};
        const int64_t cube = ci + (int64_t)h->cfg.n[0] * (cj + (int64_t)h->cfg.n[1] * ck);
        for (int k = 0; k < G.ncls; ++k)
          for (int b = 0; b < G.nq; ++b) {
            double x[3];
            G.node(c, k, b, x);
            bool in = true;
            for (int i = 0; i < d; ++i) in = in && x[i] >= lo[i] && x[i] <= hi[i];
            if (in) nodes.push_back((cube * G.ncls + k) * G.nq + b);
          }
      }
  if (nodes.empty() || nsteps == 0) return sg_set_source(h, 0, nullptr, 0, nullptr);
  std::vector<double> pattern(nodes.size() * (size_t)d * d, 0.0), w((size_t)nsteps);
  for (size_t j = 0; j < nodes.size(); ++j)
    for (int i = 0; i < d; ++i) pattern[(j * d + i) * d + i] = 1.0;
  for (int64_t k = 0; k < nsteps; ++k) {
    const double t = t_first + (double)k * dt_step, q = (t - t0) * (t - t0);
    w[(size_t)k] = (-1.0 + 2.0 * a * q) * std::exp(-a * q);
  }
  return sg_set_source_separable(h, (int64_t)nodes.size(), nodes.data(), pattern.data(), nsteps, w.data());
}

int sg_get_sym(const sg_handle* h, int* sym) {
  if (!h || !sym) return SG_ERR_ARG;
  *sym = h->sym ? 1 : 0;
  return SG_OK;
}

int sg_leave_sym(sg_handle* h) {
  if (!h) return SG_ERR_ARG;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  return leave_sym_mode(h);
}

int sg_halo_attach(sg_handle* h, int field, int side, const void* dev_in) {
  if (!h || field < 0 || field > 3 || side < 0 || side >= 2 * h->cfg.dim) return SG_ERR_ARG;
  h->ghost[field][side] = (const double*)dev_in;
  return SG_OK;
}

}  // extern "C"
