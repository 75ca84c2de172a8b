/*
 * seigen_hip.h - C-ABI of libseigen_hip.so: the MI355X (gfx950) replacement for
 * the per-timestep hot path of devitocodes/seigen.
 *
 * Every entry point below replaces something Firedrake/PyOP2 generate or run
 * for `seigen/elastic.py` (paths relative to the reference checkout).  The
 * host side (seigen_amd/elastic.py) binds these with ctypes; INTEGRATION.md
 * shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures (streams and
 *     device buffers cross as void*).
 *   - every call returns SG_OK (0) or a negative error code; the message is
 *     available from sg_last_error().
 *   - one host thread drives a handle.  The handle owns its device memory;
 *     host buffers are caller-owned and copied synchronously.
 *   - there is NO CPU fallback: sg_create fails if no HIP device is usable.
 *
 * Host field layout = the reference's `Function.dat.data` layout [upstream]:
 *   velocity  [cell][node][dim]        (VectorFunctionSpace, elastic.py:82)
 *   stress    [cell][node][dim][dim]   (TensorFunctionSpace, elastic.py:81)
 * cells ordered cube-major (x fastest) / simplex-class-minor, nodes on the
 * equispaced lattice ordered with the first reference coordinate fastest
 * (see DESIGN.md "Mesh and numbering").
 */
#ifndef SEIGEN_HIP_H
#define SEIGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever the MEANING of an existing entry point, argument or field changes (new entry points
 * alone do not bump it).  A host that drives stages itself must check it: it is the only sign of such a change.
 *   1  rounds 1-4: SG_FIELD_UH holds utemp after stage UTEMP; stage S1 reads sh1 (SG_FIELD_SH) and utemp
 *   2  round 5: stage UTEMP leaves w = dt u1 + dt^3/24 utemp in SG_FIELD_UH, stage S1 = s0 + Minv g(w) reads
 *      SG_FIELD_UH and SG_FIELD_S only (enum sg_stage below); (u1, s1) unchanged to round-off */
#define SG_ABI_VERSION 2
int sg_abi_version(void);   /* the SG_ABI_VERSION the loaded library was built with (device-free) */

#define SG_OK 0
#define SG_ERR_ARG (-1)      /* bad argument */
#define SG_ERR_DEVICE (-2)   /* HIP runtime error / no device */
#define SG_ERR_STATE (-3)    /* call sequence error (e.g. params not set) */
#define SG_ERR_NOMEM (-4)

/* device-resident fields.  The ten Functions of elastic.py:93-103 map onto
 * four buffers because the fused stages never materialise uh2 / sh2 and
 * u0.assign(u1) / s0.assign(s1) (elastic.py:296,304) are in-place updates. */
enum sg_field {
  SG_FIELD_U = 0,  /* VelocityOld / VelocityNew   (u0, u1)      */
  SG_FIELD_UH = 1, /* VelocityHalf1 / VelocityTemp (uh1; after stage UTEMP: w = dt u1 + dt^3/24 utemp, see enum sg_stage) */
  SG_FIELD_S = 2,  /* StressOld / StressNew       (s0, s1)      */
  SG_FIELD_SH = 3  /* StressTemp / StressHalf1    (stemp, sh1)  */
};

/* the six fused launches of one LF4 step (elastic.py:291-304) */
enum sg_stage {
  SG_STAGE_UH1 = 0,   /* uh1   = Minv f(s0; u0)                          :292 */
  SG_STAGE_STEMP = 1, /* stemp = Minv g(uh1)                              :293 */
  SG_STAGE_U1 = 2,    /* u1 = rho*u0 + dt*uh1 + dt^3/24 * Minv f(stemp; u0); u0<-u1   :294-296 */
  SG_STAGE_SH1 = 3,   /* sh1   = Minv g(u1)                               :300 */
  SG_STAGE_UTEMP = 4, /* utemp = Minv f(sh1; u1)                          :301   - left in UH as w = dt*u1 + dt^3/24*utemp */
  SG_STAGE_S1 = 5     /* s1 = s0 + dt*sh1 + dt^3/24 * Minv g(utemp); s0<-s1           :302-304 - computed as s0 + Minv g(w) */
};
/* g is linear in the velocity, and utemp has no other consumer than sh2 = Minv g(utemp): dt*sh1 + dt^3/24*sh2 =
 * Minv g(dt*u1 + dt^3/24*utemp) (+ (dt + dt^3/24) S with a source).  Stage UTEMP therefore leaves that ONE velocity w in
 * the UH buffer (its fused epilogue reads u1 beside its result) and stage S1 reads w and s0 only - neither sh1 nor a
 * second right-hand side: 6 of the 21 words per node the stress update moved.  (u1, s1) are the reference's to round-off;
 * what a caller finds in SG_FIELD_UH after a step is w, not utemp.  SG_FIELD_SH holds sh1 as before. */

/* which cubes of the block a stage launch covers (halo overlap) */
enum sg_region {
  SG_REGION_ALL = 0,
  SG_REGION_INTERIOR = 1, /* cubes whose stencil needs no remote trace */
  SG_REGION_BOUNDARY = 2, /* the one-cube shell next to sides that have a neighbour block */
  /* the same stage cut differently, for the pipelined exchange (seigen_amd/parallel.py): */
  SG_REGION_FIRST = 3,    /* the shell and the lower half of the interior: everything whose results
                             the neighbours wait for, inside one large launch */
  SG_REGION_SECOND = 4    /* the rest of the interior: runs while the traces of FIRST travel */
};

typedef struct sg_handle sg_handle;

typedef struct sg_config {
  int32_t dim;       /* 1, 2 or 3          (ElasticLF4.create(..., dimension), elastic.py:28) */
  int32_t degree;    /* 1..4               (ElasticLF4.create(..., degree)) */
  int32_t n[3];      /* squares / cubes per axis in THIS block */
  double h[3];       /* cell size per axis */
  double origin[3];  /* physical coordinate of the block's low corner (of cube -cube0, see below) */
  int32_t diagonal;  /* 2-D: 0 = "left" (Firedrake default), 1 = "right": how each square is cut into two triangles;
                        2 = not at all: quadrilateral cells (dim 2) or hexahedral cells (dim 3) with
                        the tensor-product element DQ_k, what FunctionSpace(mesh, "DG", k) (elastic.py:81-82) is on
                        a quadrilateral / hexahedral mesh.  3-D simplicial blocks ignore 0 / 1 (one Kuhn cut). */
  int32_t nbr_mask;  /* bit (2*axis + side) set: that side touches another block (halo), else free surface */
  int32_t device;    /* HIP device ordinal */
  int32_t dtype;     /* 0: FP64 storage and arithmetic - the reference's precision (elastic.py:442 'double'),
                        the parity and headline mode; 1: FP32 storage and arithmetic, the separately reported
                        second mode of SURVEY 8d (32 B per DoF-update; 3-D blocks on the MFMA path, 2-D blocks on the MFMA
                        tile kernels; SG_ERR_ARG elsewhere: 1-D blocks, forced generic / lane kernels).  The C-ABI
                        keeps double on the host side in both modes; halo buffers hold the device type. */
  void* stream;      /* hipStream_t to launch on, or NULL for the handle's own stream */
  int32_t cube0[3];  /* index of the block's first square / cube counted from `origin` (0: origin is the block's own low
                        corner).  Node coordinates are origin + (cube0 + cube + lattice) * h with the integers added
                        first: a block of a partitioned mesh, or a slab / sub-box of a block evaluated on its own
                        (sg_block_node_coords), given the MESH's origin and its integer offset, gets bit for bit the
                        coordinates the whole mesh gets - a source box whose faces lie on node lines
                        (explosive_source_lf4.py:36-38) then selects the same nodes under every partition. */
  int32_t pad_;
} sg_config;

typedef struct sg_info {
  int32_t dim, degree;
  int32_t nd;        /* scalar nodes per cell */
  int32_t nf;        /* nodes per facet */
  int32_t nfaces;    /* facets per cell */
  int32_t nclasses;  /* simplices per square / cube */
  int64_t ncells;
  int64_t u_dofs;    /* dim   * nd * ncells */
  int64_t s_dofs;    /* dim^2 * nd * ncells */
  int32_t halo_faces[6]; /* cell-facets lying on each block side (2*axis+side) */
} sg_info_t;

typedef struct sg_counters {
  double kernel_ms[6];   /* accumulated device time per stage (hipEvent), only when timing is enabled */
  int64_t launches[6];
  int64_t steps;
  /* halo layer (the role of ParLoopHaloEnd, tests/tiling/utils.py:144) */
  double halo_pack_ms;       /* device time of the trace-pack launches, only when timing is enabled */
  int64_t halo_pack_launches;
  int64_t halo_bytes_packed; /* bytes written to send buffers = bytes this block hands to the transport */
} sg_counters_t;

/* ---- lifecycle ------------------------------------------------------------------- */
/* replaces: function-space/field construction (elastic.py:66-103) and
 * ExplicitElasticLF4.setup (elastic.py:369-385: element-wise inverse mass). */
int sg_create(const sg_config* cfg, sg_handle** out);
void sg_destroy(sg_handle* h);
const char* sg_last_error(const sg_handle* h); /* h may be NULL: error of the last failed sg_create */
int sg_get_info(const sg_handle* h, sg_info_t* out);
int sg_sync(sg_handle* h);
/* the hipStream_t the handle launches on (its own or the one given in sg_config): the host layer
 * makes it the current stream around its torch.distributed calls, which order themselves against
 * the current stream (the role PyOP2's implicit halo/compute ordering has, elastic.py:404-436) */
int sg_get_stream(const sg_handle* h, void** stream);
/* Blocks with neighbours launch SG_REGION_SECOND of a split stage on a second, lower-priority stream beside
 * SG_REGION_FIRST of the same stage (it depends on the stage before, not on FIRST; SEIGEN_HIP_OVERLAP=0 switches
 * this off); everything queued later on the main stream waits for it.  NULL if the handle has no such stream.
 * Call order: sg_run_stage(stage, SG_REGION_SECOND) must directly follow sg_run_stage(stage, SG_REGION_FIRST) of
 * the SAME stage (FIRST records the event SECOND's stream waits for); anything else returns SG_ERR_STATE.
 * Timing counters (sg_get_counters kernel_ms) count such a stage once, with the longer of its two launches.
 * For instrumentation: the host layer records an event on it to tell how long the next stage really waited
 * for traces (what ParLoopHaloEnd times in the reference, tests/tiling/utils.py:144). */
int sg_get_second_stream(const sg_handle* h, void** stream);

/* physical coordinates of the DG nodes, [cell][node][dim]; `degree` may differ
 * from the solver's (e.g. 4 for the DG4 sponge space of
 * tests/explosive_source/explosive_source_lf4.py:43).  Replaces what
 * Function.interpolate(Expression) needs from the mesh (elastic.py:141,154). */
int sg_node_coords(const sg_handle* h, int degree, double* out, size_t nbytes);
/* the same from a configuration alone (device-free; cfg->device / stream ignored) */
int sg_block_node_coords(const sg_config* cfg, int degree, double* out, size_t nbytes);

/* ---- parameters (plain attributes density, dt, mu, l: eigenmode_2d.py:17-20) ------- */
/* per_cell = 0: lambda/mu point to one value each; 1: one value per cell
 * (build-defined heterogeneous extension). */
int sg_set_params(sg_handle* h, double density, double dt, const double* lambda, const double* mu, int per_cell);
/* Density beyond the scalar of sg_set_params (which it overrides until the next sg_set_params):
 *   per_cell = 0: rho[0];  1: one value per cell (SURVEY 8 f2 "(and rho)", build-defined like per-cell
 *   lambda/mu: each cell's velocity update uses its own density).
 *   physical = 0: the explicit reference's update u1 = rho*u0 + dt*uh1 + dt^3/24*uh2 - explicit
 *   mode keeps only rhs(form_u1) and applies the unweighted mass inverse (elastic.py:341-345,
 *   :354-356, :376), which is the physical update only for rho = 1;
 *   physical = 1: u1 = u0 + (dt*uh1 + dt^3/24*uh2)/rho, what the implicit form solves (elastic.py:175-178). */
int sg_set_density(sg_handle* h, const double* rho, int per_cell, int physical);

/* ---- field transfer (u0.assign(...), s0.assign(...): eigenmode_2d.py:32,36) -------- */
int sg_set_field(sg_handle* h, int field, const double* host, size_t nbytes);
int sg_get_field(sg_handle* h, int field, double* host, size_t nbytes);

/* the same for `ncells` consecutive cells starting at `cell0` (chunked transfer of large fields) */
int sg_set_field_range(sg_handle* h, int field, int64_t cell0, int64_t ncells, const double* host, size_t nbytes);
int sg_get_field_range(sg_handle* h, int field, int64_t cell0, int64_t ncells, double* host, size_t nbytes);

/* ---- absorption (elastic.py:136-141, :207-208) ------------------------------------ */
/* sigma_nodes: [cell][nd(sigma_degree)] nodal values of the DG_q sponge field, or NULL to disable. */
int sg_set_absorption(sg_handle* h, const double* sigma_nodes, int sigma_degree);

/* ---- source (elastic.py:149-154, :217-218, :285-288) ------------------------------- */
/* Sparse nodal source: `nnz` scalar DG nodes (flat index = cell*nd + node) carry a
 * source; values[k][i][dim*dim] is the nodal S_ij of entry i during step k
 * (k = 0 .. nsteps-1 counted from this call; no source afterwards).  This is the
 * re-interpolated `source_function` of elastic.py:285-288 restricted to its
 * support.  nsteps = -1: a time-independent source, values[0][i][dim*dim] holds at
 * every step.  nnz = 0 disables.  A node listed more than once receives the SUM of its
 * entries (added up in the order listed, once, on the host: deterministic). */
int sg_set_source(sg_handle* h, int64_t nnz, const int64_t* nodes, int64_t nsteps, const double* values);
/* The same for a SEPARABLE source S_ij(node, step k) = weights[k] * pattern[node][dim*dim] - what every source of
 * the reference's tests is (the indicator of a box times a Ricker wavelet re-interpolated at every step,
 * tests/explosive_source/explosive_source_lf4.py:36-40 with elastic.py:285-288): one slice of nodal values and
 * one factor per step (k = 0 .. nsteps-1 counted from this call; no source afterwards) instead of a table of
 * nsteps * nnz * dim^2 values.  The product weights[k] * pattern is rounded before it is added, so the result is
 * bitwise what sg_set_source gives for the table of those products (node lists without repeated nodes; the entries
 * of a repeated node are summed first, here of the pattern, there of the products). */
int sg_set_source_separable(sg_handle* h, int64_t nnz, const int64_t* nodes, const double* pattern, int64_t nsteps,
                            const double* weights);
/* The reference's explosive source itself, from its parameters (tests/explosive_source/explosive_source_lf4.py:36-40):
 *   S_ij(x, t) = delta_ij w(t) at the DG nodes x inside the CLOSED box lo <= x <= hi (the nodal interpolation of the
 *   box indicator, elastic.py:149-154), w(t) = (-1 + 2 a (t - t0)^2) exp(-a (t - t0)^2),
 * for the steps k = 0 .. nsteps-1 counted from this call, evaluated at t = t_first + k * dt_step (elastic.py:285-288
 * sets t to the time of the step before re-interpolating: t_first = dt, dt_step = dt for a run from t = 0).
 * lo, hi: dim doubles each.  Node coordinates are those of sg_node_coords.  A convenience over
 * sg_set_source_separable (same storage, same rounding rule); a box that contains no node of this block disables
 * the source of the block. */
int sg_set_source_box_ricker(sg_handle* h, const double* lo, const double* hi, double a, double t0, double t_first,
                             double dt_step, int64_t nsteps);

/* ---- the hot path ---------------------------------------------------------------- */
/* whole steps (all six stages, source included); replaces the body of
 * ElasticLF4.run's while-loop (elastic.py:283-313).  A block with neighbours needs a communicator
 * (sg_comm_init): the exchanges then run inside this call; without one, drive stages and halo from the host
 * (sg_run_stage + sg_halo_pack_sides + your transport). */
int sg_step(sg_handle* h, int64_t nsteps);
/* one fused stage over a region (multi-block overlap and stage-level tests). */
int sg_run_stage(sg_handle* h, int stage, int region);
/* advance the source-amplitude index after a manually staged step */
int sg_end_step(sg_handle* h);

/* un-fused operators for stage-level parity tests:
 *   out = Minv f(w; s_in, u_abs)   (elastic.py:204-209 + :358-367)
 *   out = Minv g(v; u_in)          (elastic.py:211-219 + :358-367)
 * `use_source` adds the current step's source to g. */
int sg_apply_F(sg_handle* h, int s_in, int u_abs, int u_out);
int sg_apply_G(sg_handle* h, int u_in, int s_out, int use_source);

/* ---- halo layer (replaces PyOP2's implicit halo exchange, elastic.py:404-436) ------- */
/* Facet traces of `field` on block side `side` (2*axis + hi), packed as
 * [facet][facet-node][dim]: the velocity components, or for a stress field the column
 * T_i,axis (i < dim) - on an axis-aligned block side the only part of the neighbour's tensor that
 * enters `avg(s0)*n` (elastic.py:206; SURVEY 8e "send T.n").  Written to the DEVICE buffer
 * `dev_out` (caller-allocated, e.g. a torch tensor). */
int sg_halo_bytes(const sg_handle* h, int field, int side, size_t* nbytes);
int sg_halo_pack(sg_handle* h, int field, int side, void* dev_out);
/* register the DEVICE buffer holding the neighbour block's packed traces of
 * `field` for `side`; read by the next stages that consume `field`. */
/* the same for several sides in ONE launch: dev_out[side] = send buffer of that side or NULL, 6 entries */
int sg_halo_pack_sides(sg_handle* h, int field, void* const* dev_out);
int sg_halo_attach(sg_handle* h, int field, int side, const void* dev_in);
/* ---- native exchange over RCCL (csrc/comm.cpp) --------------------------------------------
 * The reference's halo exchange is implicit in every assemble (elastic.py:364; set-up :404-436).  With a
 * communicator attached, sg_step on a block with neighbours runs the whole pipelined schedule itself - per stage
 * FIRST, pack, grouped ncclSend/ncclRecv with the face neighbours on the handle's stream, SECOND beside them -
 * for all six stages of all `nsteps` steps without returning to the caller; the handle owns the send / receive
 * buffers.  One process per GPU: every rank of the block grid calls sg_comm_init (collective) with the SAME unique
 * id - made by one rank with sg_comm_get_unique_id and distributed by whatever the host has (MPI_Bcast in the
 * reference's world, torch.distributed here) - its rank, the number of ranks, and per block side the rank of the
 * face neighbour (-1: none; must agree with sg_config::nbr_mask). */
#define SG_COMM_ID_BYTES 128
typedef struct sg_comm_stats {
  int64_t exchanges;        /* grouped send/receive calls issued */
  int64_t bytes_sent;       /* bytes handed to RCCL */
  double exposed_wait_ms;   /* timing enabled: time the receives lasted beyond the SECOND launch beside them */
} sg_comm_stats_t;
int sg_comm_get_unique_id(void* id, size_t nbytes);
int sg_comm_init(sg_handle* h, const void* id, size_t nbytes, int rank, int nranks, const int32_t* peers);
/* RCCL is bound at run time, on first use: the copy already in the process (torch brings its own) or the system's
 * librccl.so - or the one file the environment variable SEIGEN_RCCL_LIB names (a site's own build; the transport
 * double of tests/fake_rccl) - and must report the major version of the rccl.h this library was compiled against;
 * SG_ERR_STATE where there is none or the wrong one (single blocks and the device-free entry points do not need it).
 * sg_comm_version: its version as RCCL encodes it (e.g. 22707).  sg_comm_library: the path of the shared object the
 * bound entry points come from, NUL-terminated into buf[n] - which MPI / which RCCL moved the traces is part of a
 * run's record (the reference prints its MPI through PyOP2's configuration, tests/tiling/utils.py:143-144 timers). */
int sg_comm_version(int* version);
int sg_comm_library(char* buf, size_t n);
/* Everything sg_comm_init can refuse WITHOUT another rank (RCCL present, rank / nranks, peers[] against
 * sg_config::nbr_mask; two sides may share a peer only as the two ends of one axis).  Hosts call it on every rank and
 * agree on the result BEFORE the collective sg_comm_init: a rank that failed alone would leave the others waiting. */
int sg_comm_check(sg_handle* h, int rank, int nranks, const int32_t* peers);
/* Collective self-test of the exchange, after sg_comm_init: every rank sends a pattern naming (rank, side, position)
 * from each send buffer and counts the values that did not arrive from the FACING side of the right neighbour
 * (0 = every side receives its neighbour's trace; the role of a halo-exchange consistency check before a run).
 * Receives are posted in the order of the facing sides, so two faces between the same pair of ranks - a block that is
 * its own neighbour across an axis - are paired correctly, not mirrored. */
int sg_comm_selftest(sg_handle* h, int64_t* mismatches);   /* SG_ERR_STATE once traces have been exchanged: it overwrites the ghost buffers */
int sg_comm_finalize(sg_handle* h);
int sg_comm_get_stats(sg_handle* h, sg_comm_stats_t* out, int reset);
/* one exchange of `field`'s traces on its own (tests), and the device addresses of a side's buffers */
int sg_comm_exchange(sg_handle* h, int field);
int sg_comm_buffers(sg_handle* h, int kind, int side, void** send, void** recv, size_t* nbytes);

/* Symmetric-stress storage is left automatically when THIS block is handed a
 * non-symmetric stress or source; blocks of one mesh must agree, so the host layer reads the
 * state of every block (sg_get_sym) and makes all of them leave together (sg_leave_sym). */
int sg_get_sym(const sg_handle* h, int* sym);
int sg_leave_sym(sg_handle* h);

/* ---- instrumentation --------------------------------------------------------------- */
int sg_enable_timing(sg_handle* h, int on);
int sg_get_counters(sg_handle* h, sg_counters_t* out);
/* ms of the last sg_step call measured with hipEvents on the launch stream */
int sg_last_step_ms(sg_handle* h, double* ms);
/* Name of the kernel a launch of `stage` over `region` runs on this handle, as rocprofv3 --kernel-trace prints it
 * (e.g. "void sg::mfma_stage_G<double, 4, 0, 1, 1>(sg::StageArgs)"), NUL-terminated into buf[n] (truncated if longer).
 * Launches nothing: the stage's own dispatch code runs and reports the instantiation it selected, so profiles and
 * bench.py's `roofline.kernel` name what actually runs - the role of PyOP2's per-parloop kernel names in the reference's
 * timing summaries (tests/tiling/utils.py:143-144 [upstream get_timers]).  Blocks with neighbours: after sg_halo_attach /
 * sg_comm_init (the choice depends on the attached halo buffers). */
int sg_stage_kernel_name(sg_handle* h, int stage, int region, char* buf, size_t n);

/* ---- device-free setup queries (host logic; usable without a GPU) ------------------------ */
/* Reference-element operators of equispaced Lagrange P_degree on the dim-simplex
 * (what `assemble(..., inverse=True)`, elastic.py:376-382, and the generated
 * element kernels tabulate [upstream]).  which = 0: D[r][a][b] = (Mhat^-1 Shat_r)
 * (dim*nd*nd), 1: L[f][a][b'] facet lifts (nfaces*nd*nf), 2: Mhat (nd*nd),
 * 3: sponge tensor A[a][c][b] for a DG_q sigma (nd*nq*nd), 4: facet node lists
 * fnode[f][b'] as doubles (nfaces*nf).  out = NULL: size query.
 * Returns the number of doubles or a negative error. */
int64_t sg_reference_operator(int dim, int degree, int which, int q, double* out, size_t nbytes);
/* phi[p][a]: Lagrange basis of P_degree (degree <= 8) at reference points xi[p][dim] - the
 * tabulation behind Function evaluation / the error functional of eigenmode_2d.py:49-63. */
int sg_tabulate(int dim, int degree, int64_t npts, const double* xi, double* phi);
/* The same two for a cell type: 0 = simplex (the functions above), 1 = tensor-product cell (the unit square / cube,
 * (degree+1)^dim equispaced nodes with the first coordinate fastest; facet 2m: x_m = 0, facet 2m+1: x_m = 1) -
 * the element of sg_config::diagonal = 2. */
int64_t sg_reference_operator_cell(int cell_type, int dim, int degree, int which, int q, double* out, size_t nbytes);
int sg_tabulate_cell(int cell_type, int dim, int degree, int64_t npts, const double* xi, double* phi);
/* The disjoint boxes of cubes {origin[3], extent[3]} a region of a split stage covers in the block
 * `cfg` describes (n, dim, nbr_mask); returns their number (at most SG_MAX_REGION_BOXES; a stage
 * launch carries that many), writes the first `max_boxes` of them to boxes[][6]. */
#define SG_MAX_REGION_BOXES 7
int sg_region_boxes(const sg_config* cfg, int region, int32_t* boxes, int max_boxes);
/* Translation-invariant neighbour tables of the structured simplicial mesh:
 * nb [cls][face][5] = {axis crossed (-1: same cube), direction, neighbour class,
 * neighbour facet, ordinal on the cube side}; nb_node [cls][face][nf] neighbour
 * element node matching each facet node; cn [cls][face][3] = |F|/|detJ| * outward
 * normal; jinv [cls][3][3]. */
int sg_mesh_tables(int dim, int degree, int diagonal, const double* h, int32_t* nb, int32_t* nb_node, double* cn,
                   double* jinv);

#ifdef __cplusplus
}
#endif
#endif /* SEIGEN_HIP_H */
