// The handle behind the C-ABI (include/seigen_hip.h) and what its translation units share:
//   api.cpp      create / destroy, parameters, sponge, source, table exports
//   transfer.cpp host <-> device field transfers (layout conversion, pinned pipeline)
//   stages.cpp   regions, stage launches, the LF4 step, graphs, halo packs, timing
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/seigen_hip.h"
#include "hostlogic.hpp"
#include "kernels.hpp"
#include "mesh_tables.hpp"
#include "mfma_tables.hpp"
#include "refelem.hpp"

using namespace sg;

struct sg_comm_state;   // comm.cpp: RCCL communicator, peers and halo buffers of the native exchange

struct sg_handle {
  sg_config cfg;
  sg_comm_state* comm = nullptr;
  RefElem re;
  MeshDev md;
  MeshDev* md_dev = nullptr;
  double* Dt = nullptr;
  double* Lt = nullptr;
  double* field[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t field_len[4] = {0, 0, 0, 0};    // doubles, host layout (ncells * nd * comps)
  size_t field_alloc[4] = {0, 0, 0, 0};  // doubles allocated on the device (layout padding included)
  bool use_mfma = false;
  bool use_lane = false;
  bool use_tile = false;    // 2-D MFMA tile kernels (kernels_tile2d.hip), gw = 16
  bool use_hexm = false;    // hexahedra DQ_3 / DQ_4 (kernels_hexm.hip), gw = 16
  int f32 = 0;              // sg_config.dtype = 1: fields, halo buffers, operator tiles and arithmetic are float (MFMA path)
  bool sym = false;         // MFMA path: all stress fields symmetric -> kernels touch only the i <= j lines
  int* sym_flag = nullptr;  // device word set by an upload that is not symmetric
  // active (cell group, class) items of each region of a split stage (MFMA / lane paths), by sg_region
  int32_t* region_items[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int32_t region_nitems[5] = {-1, -1, -1, -1, -1};  // -1: not built yet
  bool region_whole[5] = {false, false, false, false, false};  // no listed cell group is cut by the region's boxes
  double* fragF = nullptr;  // MFMA operator fragment tables (device)
  double* fragG = nullptr;
  double* fragL = nullptr;
  // G stages with the factorised volume term (kernels_mfma.hip mfma_stage_G<.., FACT = 1>; double, degrees 3 and 4;
  // SEIGEN_HIP_GQ): the Q tiles and the P_r tiles, or null
  double* fragQ = nullptr;
  double* fragP = nullptr;
  double* staging = nullptr;  // host-layout staging buffer for layout conversion
  // large transfers: two pinned host slots + two device slots, so that the DMA of one chunk, the
  // layout kernel of the next and the host-side copy of the previous one overlap
  double* pin[2] = {nullptr, nullptr};
  double* dstage[2] = {nullptr, nullptr};
  hipEvent_t xfer_ev[2] = {nullptr, nullptr};
  unsigned long long* dbg = nullptr;  // SEIGEN_HIP_STAMPS=1 (diagnostic builds): [kind][8] cycle sums
  size_t staging_len = 0;
  int64_t ncells = 0;
  int ncls = 0;
  // parameters
  bool params_set = false;
  double rho = 1.0, dt = 0.0, lam0 = 0.0, mu0 = 0.0;
  int per_cell = 0;
  double* lam_d = nullptr;
  double* mu_d = nullptr;
  double* rho2_d = nullptr;  // per-cell density factors [cell][2] (kernels.hpp), or null
  int rho_physical = 0;      // scalar density: 0 = rho*u0 + ..., 1 = u0 + (...)/rho
  // sponge
  int32_t* sponge_slot = nullptr;
  double* sponge_B = nullptr;
  double* sponge_sigma = nullptr;   // 2-D tile and 3-D MFMA kernels only (kernels.hpp StageArgs::sponge_sigma)
  int32_t sponge_nslots = 0;        // cells with a sponge matrix of their own
  int32_t* sponge_cells = nullptr;  // [slot] -> cell: the pre-pass of the F stages (kernels.hpp launch_sponge_pre)
  int32_t* sponge_mat = nullptr;    // [slot] -> matrix in sponge_B (cells with the same nodal sigma share one)
  void* sponge_pre = nullptr;       // [slot][nd][dim] in the field type
  // cells whose sigma is affine in the reference coordinates take dim + 1 numbers instead of a matrix (kernels.hpp
  // launch_sponge_pre_affine); the cells that keep a matrix are then a LIST of slots
  int sponge_pre_lines = 0;             // sponge_pre in line layout, slot = item * gw + w (3-D MFMA family), else a record per slot
  int32_t sponge_nmat_slots = 0;        // slots with a matrix (all of them where no cell is affine: sponge_mat_slots stays null)
  int32_t* sponge_mat_slots = nullptr;
  int32_t sponge_aff_nitems = 0, sponge_aff_W = 0;
  int32_t* sponge_aff_items = nullptr;  // [item] -> (cube group) * ncls + class
  int32_t* sponge_aff_slots = nullptr;  // [item][gw] -> slot or -1
  double* sponge_aff_coef = nullptr;    // [slot][dim + 1]
  double* sponge_aff_X = nullptr;       // [dim][nd][W]
  int32_t* sponge_aff_col = nullptr;    // [nd][W], null where the rows are dense
  double* sponge_aff_frag = nullptr;    // 3-D MFMA family in double: the X_k as row tiles (mfma_frags_dense)
  int sponge_pre_key = -1, sponge_pre_regions = 0;   // the F stage (output field, mode) whose pre-pass ran last, and the regions launched since
  // The pre-pass is B_e u_abs of a FIELD STATE: stages UH1 and U1 both absorb u0 (elastic.py:206-208 in form_uh1 and form_uh2),
  // so the second of them finds the first one's result - two pre-passes per step instead of three.  fver counts the writes to
  // each field (stage outputs, uploads); sponge_pre_ver / _field name the state the buffer holds (~0: none).
  uint64_t fver[4] = {0, 0, 0, 0};
  uint64_t sponge_pre_ver = ~0ull;
  int sponge_pre_field = -1;
  // source
  int64_t src_nnz = 0;
  int64_t src_nfirst = 0;  // source nodes are stored with those in cells of SG_REGION_FIRST first
  int64_t* src_nodes = nullptr;
  double* src_values = nullptr;  // [nsteps][nnz][dim*dim]
  int64_t src_nsteps = 0;
  int64_t src_step = 0;
  bool src_static = false;  // one time slice that holds at every step
  std::vector<double> src_weights;  // separable source: src_values is one slice, scaled by src_weights[src_step]
  // graph replay with a source: the step index lives in a device word that the captured launches read and a one-thread
  // launch bumps at the end of every step (kernels.hpp SrcStep); sg_step sets it to src_step before it replays
  int64_t* src_ctr_d = nullptr;
  double* src_weights_d = nullptr;
  bool capture_src = false;   // stage launches issued now (a capture) take slice and weight from src_ctr_d
  bool graph_src = false;     // the captured graphs contain the source launches
  // 2-D tile path: the source is added inside the G stage kernels (StageArgs::src_slot / src_idx)
  bool src_fused = false;
  int32_t* src_slot_d = nullptr;
  int32_t* src_idx_d = nullptr;
  // halo
  const double* ghost[4][6];
  // execution
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // Split stages (blocks with neighbours): SG_REGION_SECOND of a stage depends on the stage before it, not on the
  // FIRST launch of its own stage, so it may run on a second (lower-priority) stream and fill the slots that FIRST's
  // persistent blocks free as they drain (default; SEIGEN_HIP_OVERLAP=0: one stream).  ev_stage: everything before this stage's FIRST;
  // ev_second: the SECOND launch, which every later piece of work on `stream` waits for.
  bool overlap = false;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_stage = nullptr, ev_second = nullptr;
  bool second_pending = false;
  int32_t* nbr_tab = nullptr;   // MFMA path: per-item neighbour table (StageArgs::nbr_tab)
  MfmaConst* mk_dev = nullptr;  // MFMA path: scalar-load copy of the mesh constants (kernels.hpp MfmaConst)
  int32_t* ftab_dev = nullptr;  // MFMA path, F stages: tabulated trace offsets (kernels.hpp mfma_trace_offsets)
  int grid_blocks = 0;  // persistent grid of the MFMA stage kernels: while an exchange is in flight ...
  int order_chunk = 0;  // MFMA path: items per XCD chunk of whole-block launches (StageArgs::order_chunk)
  int grid_full = 0;    // ... and otherwise (every block slot of the device)
  T2Const t2c;          // 2-D tile kernels: kernarg copy of the mesh tables
  int tile_grid = 0, tile_grid_sponge = 0;    // 2-D tile kernels: cap of the grid in blocks of four waves, without / with a sponge (SEIGEN_HIP_TILE_GRID)
  // small blocks are launch-bound (config 1: six 5-us launches per step): sg_step replays captured
  // hipGraphs of one and of eight steps there; any setter that changes kernel arguments bumps the epoch
  bool graph_ok = false;
  uint64_t epoch = 0, graph_epoch = ~0ull;
  hipGraphExec_t graph1 = nullptr, graph8 = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double last_ms = 0.0;
  bool timing = false;
  std::vector<hipEvent_t> ev_pool;   // per-launch event pairs, resolved lazily (no sync in the hot loop)
  // stage of pair k = events 2k, 2k+1 (6 = halo pack); + 16 for the FIRST, + 32 for the SECOND launch of a stage
  // that runs its two regions side by side on two streams (their pairs overlap in time: the stage counts the longer)
  std::vector<int> ev_stage_ids;
  double first_ms_pending[6] = {-1, -1, -1, -1, -1, -1};
  int first_recorded_stage = -1;         // stage whose FIRST launch recorded ev_stage last (SECOND must follow it)
  sg_counters_t counters;
  std::string* name_out = nullptr;   // sg_stage_kernel_name: stage launches only name their kernel (StageArgs::name_out)
  bool no_whole = false;             // SEIGEN_HIP_NO_WHOLE (diagnostic): region launches always test the boxes
  std::string err;
};

static_assert(SG_MAX_BOXES == SG_MAX_REGION_BOXES, "kernels.hpp and seigen_hip.h disagree on the box limit");

#define HIPCHECK(h, expr)                                                                        \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                              \
      return SG_ERR_DEVICE;                                                                      \
    }                                                                                            \
  } while (0)

inline int fail(sg_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

inline bool field_is_stress(int f) { return f == SG_FIELD_S || f == SG_FIELD_SH; }

// work queued on `stream` from here on comes after the SECOND launch that may still run on stream2
inline int join_second(sg_handle* h) {
  if (h->second_pending) {
    HIPCHECK(h, hipStreamWaitEvent(h->stream, h->ev_second, 0));
    h->second_pending = false;
  }
  return SG_OK;
}

// the host waits for everything the handle has queued (both streams)
inline hipError_t sync_all(sg_handle* h) {
  if (h->second_pending) {
    hipError_t e = hipStreamWaitEvent(h->stream, h->ev_second, 0);
    if (e != hipSuccess) return e;
    h->second_pending = false;
  }
  return hipStreamSynchronize(h->stream);
}

// transfer.cpp: make the (i > j) lines of both stress buffers valid again and continue with the full-tensor kernels
int leave_sym_mode(sg_handle* h);

// hostapi.cpp (hostlogic.hpp): kernel-family choice and the regions of a split stage
inline void region_boxes(const sg_handle* h, int region, std::vector<Box>& out) {
  region_boxes(h->cfg.dim, h->cfg.n, h->md.has_nbr, region, out,
               shell_width_x(h->md.gw, h->cfg.n[0], h->md.has_nbr[0] != 0, h->md.has_nbr[1] != 0));
}
int resolve_timing(sg_handle* h);
// comm.cpp
int comm_step(sg_handle* h, int64_t nsteps);
void comm_release(sg_handle* h);
