// Host <-> device field transfers of the C-ABI (sg_set_field / sg_get_field and their range forms): layout
// conversion between the reference's [cell][node][comp] order and the device layout (MeshDev::gw), the pinned
// two-slot pipeline for large downloads, and the symmetric-stress bookkeeping of uploads.
#include "handle.hpp"

// Leave symmetric-stress mode: make the (i > j) lines of both stress buffers valid again.
int leave_sym_mode(sg_handle* h) {
  h->epoch += 1;
  if (!h->sym) return SG_OK;
  for (int f : {SG_FIELD_S, SG_FIELD_SH})
    if (launch_mirror(h->md, h->field[f], h->f32, h->stream) != 0) return fail(h, SG_ERR_DEVICE, "mirror kernel launch failed");
  HIPCHECK(h, sync_all(h));
  h->sym = false;
  return SG_OK;
}

// host-to-host copy on several threads (one thread moves about 10 GB/s, the PCIe link 50+)
static void parallel_memcpy(void* dst, const void* src, size_t nbytes) {
  unsigned nt = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (nbytes < ((size_t)4 << 20)) nt = 1;
  if (nt == 1) {
    std::memcpy(dst, src, nbytes);
    return;
  }
  std::vector<std::thread> th;
  const size_t per = (nbytes / nt + 4095) & ~(size_t)4095;
  for (unsigned t = 0; t < nt; ++t) {
    const size_t o = (size_t)t * per;
    if (o >= nbytes) break;
    const size_t n = std::min(per, nbytes - o);
    th.emplace_back([=]() { std::memcpy((char*)dst + o, (const char*)src + o, n); });
  }
  for (auto& t : th) t.join();
}

static constexpr size_t XFER_CHUNK_BYTES = (size_t)64 << 20;

// Large transfers (sg_set_field / sg_get_field of hundreds of MB): chunks of 64 MB go through pinned
// host slots.  Download: layout kernel -> device slot -> async DMA -> pinned slot, while the host
// copies the previous pinned slot into the caller's (pageable) array on several threads.  Downloads only:
// uploads from pageable memory already run at 47 GB/s inside the runtime.  A plain hipMemcpy to pageable memory runs at 11 GB/s (one staging thread inside the
// runtime); this pipeline is bound by the link.
static int download_pipelined(sg_handle* h, int field, int64_t cell0, int64_t ncells, double* host) {
  const size_t per_cell = h->field_len[field] / (size_t)h->ncells;
  const int comps = (int)(per_cell / h->re.nd);
  const size_t chunk_cells = std::max<size_t>(1, XFER_CHUNK_BYTES / (per_cell * sizeof(double)));
  const size_t slot_len = chunk_cells * per_cell;
  for (int i = 0; i < 2; ++i) {
    if (!h->pin[i]) HIPCHECK(h, hipHostMalloc((void**)&h->pin[i], slot_len * sizeof(double), hipHostMallocDefault));
    if (!h->dstage[i] && h->md.gw != 1) HIPCHECK(h, hipMalloc((void**)&h->dstage[i], slot_len * sizeof(double)));
    if (!h->xfer_ev[i]) HIPCHECK(h, hipEventCreateWithFlags(&h->xfer_ev[i], hipEventDisableTiming));
  }
  const int64_t nchunks = (ncells + (int64_t)chunk_cells - 1) / (int64_t)chunk_cells;
  auto range = [&](int64_t c, int64_t& c0, int64_t& n) {
    c0 = c * (int64_t)chunk_cells;
    n = std::min<int64_t>((int64_t)chunk_cells, ncells - c0);
  };
  const int symdl = (h->sym && field_is_stress(field)) ? 1 : 0;
  for (int64_t c = 0; c <= nchunks; ++c) {
    const int sl = (int)(c & 1);
    int64_t c0, n;
    if (c < nchunks) {
      range(c, c0, n);
      const size_t nb = (size_t)n * per_cell * sizeof(double);
      if (h->md.gw == 1) {
        HIPCHECK(h, hipMemcpyAsync(h->pin[sl], h->field[field] + (size_t)(cell0 + c0) * per_cell, nb, hipMemcpyDeviceToHost, h->stream));
      } else {
        if (launch_layout(h->md, comps, 1, h->field[field], h->dstage[sl], cell0 + c0, n, symdl, nullptr, h->f32, h->stream) != 0)
          return fail(h, SG_ERR_DEVICE, "layout kernel launch failed");
        HIPCHECK(h, hipMemcpyAsync(h->pin[sl], h->dstage[sl], nb, hipMemcpyDeviceToHost, h->stream));
      }
      HIPCHECK(h, hipEventRecord(h->xfer_ev[sl], h->stream));
    }
    if (c > 0) {  // the previous chunk has arrived (or is arriving) in the other slot: hand it to the caller
      range(c - 1, c0, n);
      HIPCHECK(h, hipEventSynchronize(h->xfer_ev[sl ^ 1]));
      parallel_memcpy(host + (size_t)c0 * per_cell, h->pin[sl ^ 1], (size_t)n * per_cell * sizeof(double));
    }
  }
  HIPCHECK(h, sync_all(h));
  return SG_OK;
}

// Copy `ncells` cells from `cell0` between a host array in the reference layout and the device
// field.  gw == 1: the layouts coincide; otherwise go through a staging buffer + layout kernel.
static int transfer(sg_handle* h, int field, int64_t cell0, int64_t ncells, double* host, bool to_device) {
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (int rc = join_second(h)) return rc;
  HIPCHECK(h, sync_all(h));
  if (to_device) h->fver[field] += 1;      // what was derived from the field's old state (the sponge pre-pass) is stale
  const size_t per_cell = h->field_len[field] / (size_t)h->ncells;
  // downloads only: uploads from pageable memory already run at 47 GB/s inside the runtime (measured,
  // tools/transfer_rate.py: 35 GB/s through this pipeline)
  if (!to_device && (size_t)ncells * per_cell * sizeof(double) >= ((size_t)16 << 20) && !std::getenv("SEIGEN_HIP_PLAIN_COPY")) {
    return download_pipelined(h, field, cell0, ncells, host);
  }
  if (h->md.gw == 1) {
    double* dev = h->field[field] + (size_t)cell0 * per_cell;
    size_t nb = (size_t)ncells * per_cell * sizeof(double);
    if (to_device)
      HIPCHECK(h, hipMemcpy(dev, host, nb, hipMemcpyHostToDevice));
    else
      HIPCHECK(h, hipMemcpy(host, dev, nb, hipMemcpyDeviceToHost));
    return SG_OK;
  }
  // The staging buffer is allocated once, by whichever field is moved first: size it for the widest field (stress),
  // or a velocity-sized buffer of a one-cell block holds no whole cell of a stress field (chunk = 0: no progress).
  const size_t per_cell_max = (size_t)h->re.nd * h->cfg.dim * h->cfg.dim;
  const size_t cap_cells = std::max<size_t>(1, ((size_t)32 << 20) / per_cell_max);  // 256 MB staging
  if (!h->staging) {
    h->staging_len = std::min(cap_cells, (size_t)h->ncells) * per_cell_max;
    HIPCHECK(h, hipMalloc((void**)&h->staging, h->staging_len * sizeof(double)));
  }
  const size_t chunk = h->staging_len / per_cell;
  if (chunk == 0) return fail(h, SG_ERR_STATE, "staging buffer smaller than one cell");
  const int comps = (int)(per_cell / h->re.nd);
  for (int64_t done = 0; done < ncells; done += (int64_t)chunk) {
    int64_t n = std::min<int64_t>((int64_t)chunk, ncells - done);
    size_t nb = (size_t)n * per_cell * sizeof(double);
    if (to_device) {
      HIPCHECK(h, hipMemcpy(h->staging, host + (size_t)done * per_cell, nb, hipMemcpyHostToDevice));
      int* flag = (h->sym && field_is_stress(field)) ? h->sym_flag : nullptr;
      if (launch_layout(h->md, comps, 0, h->field[field], h->staging, cell0 + done, n, 0, flag, h->f32, h->stream) != 0)
        return fail(h, SG_ERR_DEVICE, "layout kernel launch failed");
      HIPCHECK(h, sync_all(h));
    } else {
      const int symdl = (h->sym && field_is_stress(field)) ? 1 : 0;
      if (launch_layout(h->md, comps, 1, h->field[field], h->staging, cell0 + done, n, symdl, nullptr, h->f32, h->stream) != 0)
        return fail(h, SG_ERR_DEVICE, "layout kernel launch failed");
      HIPCHECK(h, sync_all(h));
      HIPCHECK(h, hipMemcpy(host + (size_t)done * per_cell, h->staging, nb, hipMemcpyDeviceToHost));
    }
  }
  if (to_device && h->sym && field_is_stress(field)) {
    int flag = 0;
    HIPCHECK(h, hipMemcpy(&flag, h->sym_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) {
      // a non-symmetric stress arrived: make every (i > j) line of both stress buffers valid (they
      // are stale wherever kernels ran in symmetric mode), then repeat this upload in full mode
      int rc = leave_sym_mode(h);
      if (rc != SG_OK) return rc;
      return transfer(h, field, cell0, ncells, host, true);
    }
  }
  return SG_OK;
}

extern "C" {

int sg_set_field(sg_handle* h, int field, const double* host, size_t nbytes) {
  if (!h || !host || field < 0 || field > 3) return SG_ERR_ARG;
  if (nbytes != h->field_len[field] * sizeof(double)) return fail(h, SG_ERR_ARG, "sg_set_field: size mismatch");
  return transfer(h, field, 0, h->ncells, const_cast<double*>(host), true);
}

int sg_get_field(sg_handle* h, int field, double* host, size_t nbytes) {
  if (!h || !host || field < 0 || field > 3) return SG_ERR_ARG;
  if (nbytes != h->field_len[field] * sizeof(double)) return fail(h, SG_ERR_ARG, "sg_get_field: size mismatch");
  return transfer(h, field, 0, h->ncells, host, false);
}

static int field_range(sg_handle* h, int field, int64_t cell0, int64_t ncells, size_t nbytes, const char* who) {
  if (!h || field < 0 || field > 3) return SG_ERR_ARG;
  if (cell0 < 0 || ncells < 0 || cell0 + ncells > h->ncells) return fail(h, SG_ERR_ARG, std::string(who) + ": cell range out of bounds");
  size_t per_cell = h->field_len[field] / (size_t)h->ncells;
  if (nbytes != (size_t)ncells * per_cell * sizeof(double)) return fail(h, SG_ERR_ARG, std::string(who) + ": size mismatch");
  return SG_OK;
}

int sg_set_field_range(sg_handle* h, int field, int64_t cell0, int64_t ncells, const double* host, size_t nbytes) {
  int rc = field_range(h, field, cell0, ncells, nbytes, "sg_set_field_range");
  if (rc != SG_OK || !host) return rc != SG_OK ? rc : SG_ERR_ARG;
  return transfer(h, field, cell0, ncells, const_cast<double*>(host), true);
}

int sg_get_field_range(sg_handle* h, int field, int64_t cell0, int64_t ncells, double* host, size_t nbytes) {
  int rc = field_range(h, field, cell0, ncells, nbytes, "sg_get_field_range");
  if (rc != SG_OK || !host) return rc != SG_OK ? rc : SG_ERR_ARG;
  return transfer(h, field, cell0, ncells, host, false);
}

}  // extern "C"
