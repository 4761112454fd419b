// hip/dist.hip — the multi-GPU seam (SPEC §MG): one process per GPU, reads sharded by rank with no collective on
// the mapping path; after mapping ONE exchange of the ranks' equivalence-class tables over RCCL (all-gather of the packed fields,
// xGMI inside a node) folded in where they land (sq_eq_merge_device: integer counts and fixed-point weight sums, exact in any
// order), the per-transcript model state reduced by a defined rule, and posterior replicates sharded by rank.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a single-GPU user of libsalmon_hip.so never loads it.
#include "ctx.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>
#include <vector>

namespace {
struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi* rccl() {
  static RcclApi api; static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { api.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (api.h) break; }
    if (api.h) {
      api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.h, "ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.h, "ncclCommInitRank");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy");
      api.AllGather = (decltype(api.AllGather))dlsym(api.h, "ncclAllGather");
      api.AllReduce = (decltype(api.AllReduce))dlsym(api.h, "ncclAllReduce");
      api.Broadcast = (decltype(api.Broadcast))dlsym(api.h, "ncclBroadcast");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
      if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce || !api.Broadcast) { dlclose(api.h); api.h = nullptr; }
    }
  }
  return api.h ? &api : nullptr;
}
#define SQ_NCCL(expr)                                                                                         \
  do { ncclResult_t _r = (expr); if (_r != ncclSuccess) {                                                     \
      sq_set_error("%s failed: %s", #expr, rccl()->GetErrorString ? rccl()->GetErrorString(_r) : "RCCL error"); return SQ_ERR_DEVICE; } } while (0)
}  // namespace

struct sq_dist {
  ncclComm_t comm = nullptr; int rank = 0, world = 1, device = 0; hipStream_t st = nullptr;
  sq_dbuf<uint8_t> send, recv;   // staging for the padded all-gathers
};

extern "C" int sq_dist_make_id(uint8_t* id128) {
  if (!id128) { sq_set_error("sq_dist_make_id: null buffer"); return SQ_ERR_ARG; }
  RcclApi* R = rccl(); if (!R) { sq_set_error("librccl.so.1 not found: multi-GPU needs RCCL (there is no other transport)"); return SQ_ERR_DEVICE; }
  static_assert(sizeof(ncclUniqueId) == SQ_DIST_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id; SQ_NCCL(R->GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return SQ_OK;
}
extern "C" int sq_dist_init(const uint8_t* id128, int rank, int world, int device, sq_dist** out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) { sq_set_error("sq_dist_init: bad arguments"); return SQ_ERR_ARG; }
  RcclApi* R = rccl(); if (!R) { sq_set_error("librccl.so.1 not found: multi-GPU needs RCCL (there is no other transport)"); return SQ_ERR_DEVICE; }
  SQ_HIP_CHECK(hipSetDevice(device));
  sq_dist* d = new sq_dist(); d->rank = rank; d->world = world; d->device = device;
  ncclUniqueId id; memcpy(&id, id128, sizeof(id));
  if (hipStreamCreate(&d->st) != hipSuccess) { delete d; sq_set_error("sq_dist_init: stream creation failed"); return SQ_ERR_DEVICE; }
  ncclResult_t r = R->CommInitRank(&d->comm, world, id, rank);
  if (r != ncclSuccess) { sq_set_error("ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(r) : "?"); (void)hipStreamDestroy(d->st); delete d; return SQ_ERR_DEVICE; }
  *out = d;
  return SQ_OK;
}
extern "C" void sq_dist_free(sq_dist* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->comm && rccl()) (void)rccl()->CommDestroy(d->comm);
  d->send.free_(); d->recv.free_();
  if (d->st) (void)hipStreamDestroy(d->st);
  delete d;
}
extern "C" int sq_dist_rank(const sq_dist* d) { return d ? d->rank : 0; }
extern "C" int sq_dist_world(const sq_dist* d) { return d ? d->world : 1; }

// all ranks contribute `bytes` (any size, possibly different per rank); returns every rank's size and a device buffer holding
// rank r's bytes at r * stride
static int gather_var(sq_dist* d, const void* dev_src, size_t bytes, std::vector<uint64_t>& sizes, size_t* stride, uint8_t** base) {
  RcclApi* R = rccl();
  const int W = d->world;
  sizes.assign(W, 0);
  if (d->send.ensure(64 + 16) || d->recv.ensure((size_t)W * 16 + 64)) { sq_set_error("device allocation failed (dist sizes)"); return SQ_ERR_NOMEM; }
  uint64_t mine = bytes;
  SQ_HIP_CHECK(hipMemcpyAsync(d->send.p, &mine, 8, hipMemcpyHostToDevice, d->st));
  SQ_NCCL(R->AllGather(d->send.p, d->recv.p, 8, ncclChar, d->comm, d->st));
  SQ_HIP_CHECK(hipMemcpyAsync(sizes.data(), d->recv.p, (size_t)W * 8, hipMemcpyDeviceToHost, d->st));
  SQ_HIP_CHECK(hipStreamSynchronize(d->st));
  size_t mx = 16; for (uint64_t s : sizes) mx = std::max<size_t>(mx, (size_t)s);
  mx = (mx + 15) & ~(size_t)15;
  if (d->send.ensure(mx) || d->recv.ensure(mx * W)) { sq_set_error("device allocation failed (dist all-gather of %zu bytes x %d ranks)", mx, W); return SQ_ERR_NOMEM; }
  if (bytes) SQ_HIP_CHECK(hipMemcpyAsync(d->send.p, dev_src, bytes, hipMemcpyDeviceToDevice, d->st));
  SQ_NCCL(R->AllGather(d->send.p, d->recv.p, mx, ncclChar, d->comm, d->st));
  SQ_HIP_CHECK(hipStreamSynchronize(d->st));
  *stride = mx; *base = d->recv.p;
  return SQ_OK;
}

// One rank's table as it travels: [E, L | off (E+1) | count E | h1 E | h2 E | wq L | tid L (u32) | bins L (u32)], packed from the
// canonical-order export resident in HBM (device-to-device copies on the communicator's stream).
static size_t packed_bytes(uint64_t E, uint64_t L) { return 16 + (E ? (E + 1) * 8 + 3 * E * 8 + L * 8 + L * 4 + L * 4 : 0); }
static int pack_table(sq_dist* d, sq_ctx* c, sq_dbuf<uint8_t>& pack, size_t* bytes_out) {
  sq_eq_table mine; int rc = sq_eq_export_device(c, &mine); if (rc) return rc;
  const uint64_t E = mine.num_classes, L = mine.num_labels;
  const size_t bytes = packed_bytes(E, L);
  if (pack.ensure(bytes + 16)) { sq_set_error("device allocation failed (dist pack)"); return SQ_ERR_NOMEM; }
  uint64_t hdr[2] = {E, L}; SQ_HIP_CHECK(hipMemcpyAsync(pack.p, hdr, 16, hipMemcpyHostToDevice, d->st));
  uint8_t* p = pack.p + 16;
  auto put = [&](const void* src, size_t n) -> int { if (n && hipMemcpyAsync(p, src, n, hipMemcpyDeviceToDevice, d->st) != hipSuccess) return 1; p += n; return 0; };
  if (E && (put(mine.off, (E + 1) * 8) || put(mine.count, E * 8) || put(mine.h1, E * 8) || put(mine.h2, E * 8) || put(mine.wq, L * 8) ||
            put(mine.tid, L * 4) || put(mine.bins, L * 4))) { sq_set_error("dist pack copy failed"); return SQ_ERR_DEVICE; }
  SQ_HIP_CHECK(hipStreamSynchronize(d->st));
  *bytes_out = bytes;
  return SQ_OK;
}
// a received packed table (device memory, `size` bytes as announced by its sender) folded into c's class table
static int merge_packed(sq_ctx* c, uint8_t* base, uint64_t size, int from_rank) {
  uint64_t hdr[2]; SQ_HIP_CHECK(hipMemcpy(hdr, base, 16, hipMemcpyDeviceToHost));
  const uint64_t Er = hdr[0], Lr = hdr[1];
  if (packed_bytes(Er, Lr) != size) { sq_set_error("sq_dist_merge_eq: rank %d sent a malformed table (%llu classes, %llu labels, %llu bytes)", from_rank,
      (unsigned long long)Er, (unsigned long long)Lr, (unsigned long long)size); return SQ_ERR_STATE; }
  if (!Er) return SQ_OK;
  uint8_t* p = base + 16;
  sq_eq_table t; memset(&t, 0, sizeof(t)); t.num_classes = Er; t.num_labels = Lr;
  t.off = (uint64_t*)p; p += (Er + 1) * 8; t.count = (uint64_t*)p; p += Er * 8; t.h1 = (uint64_t*)p; p += Er * 8; t.h2 = (uint64_t*)p; p += Er * 8;
  t.wq = (uint64_t*)p; p += Lr * 8; t.tid = (uint32_t*)p; p += Lr * 4; t.bins = (uint32_t*)p;
  return sq_eq_merge_device(c, &t);
}

// The eq-class exchange: every rank ends with the union of all ranks' classes (counts and fixed-point weight sums added exactly).
// A communicator of one rank runs the same collectives (the sizes' and the payload's all-gather) and merges nothing.
extern "C" int sq_dist_merge_eq(sq_dist* d, sq_ctx* c) {
  if (!d || !c) { sq_set_error("sq_dist_merge_eq: bad arguments"); return SQ_ERR_ARG; }
  SQ_HIP_CHECK(hipSetDevice(d->device));
  sq_dbuf<uint8_t> pack; size_t bytes = 0;
  int rc = pack_table(d, c, pack, &bytes); if (rc) { pack.free_(); return rc; }
  std::vector<uint64_t> sizes; size_t stride = 0; uint8_t* base = nullptr;
  rc = gather_var(d, pack.p, bytes, sizes, &stride, &base);
  pack.free_();
  if (rc) return rc;
  for (int r = 0; r < d->world; ++r) {
    if (r == d->rank) continue;
    rc = merge_packed(c, base + (size_t)r * stride, sizes[r], r); if (rc) return rc;
  }
  return SQ_OK;
}

// Loop-back form for a box with fewer GPUs than ranks (tests, bring-up): `n` contexts on this communicator's device stand for the ranks of an
// n-rank job.  Every context's table takes the path a rank's table takes — packed, its size and its bytes all-gathered over RCCL on the
// one-rank communicator, landed in a receive slot with the stride an n-rank all-gather would use — and every context then merges the
// slots of the others, exactly as sq_dist_merge_eq does with what arrived over xGMI.
extern "C" int sq_dist_merge_eq_loopback(sq_dist* d, sq_ctx* const* ctxs, uint32_t n) {
  if (!d || !ctxs || n == 0) { sq_set_error("sq_dist_merge_eq_loopback: bad arguments"); return SQ_ERR_ARG; }
  if (d->world != 1) { sq_set_error("sq_dist_merge_eq_loopback: needs a communicator of one rank (it stands in for %u)", n); return SQ_ERR_STATE; }
  SQ_HIP_CHECK(hipSetDevice(d->device));
  std::vector<sq_dbuf<uint8_t>> slot(n); std::vector<uint64_t> bytes_of(n, 0);
  auto release = [&]() { for (auto& s : slot) s.free_(); };
  for (uint32_t v = 0; v < n; ++v) {
    sq_dbuf<uint8_t> pack; size_t bytes = 0;
    int rc = pack_table(d, ctxs[v], pack, &bytes); if (rc) { pack.free_(); release(); return rc; }
    std::vector<uint64_t> sizes; size_t stride = 0; uint8_t* base = nullptr;
    rc = gather_var(d, pack.p, bytes, sizes, &stride, &base);
    pack.free_();
    if (rc) { release(); return rc; }
    if (sizes.size() != 1 || sizes[0] != bytes) { release(); sq_set_error("sq_dist_merge_eq_loopback: the size all-gather returned %llu for %zu bytes", (unsigned long long)(sizes.empty() ? 0 : sizes[0]), bytes); return SQ_ERR_STATE; }
    if (slot[v].ensure(stride + 16)) { release(); sq_set_error("device allocation failed (loop-back slot)"); return SQ_ERR_NOMEM; }
    SQ_HIP_CHECK(hipMemcpy(slot[v].p, base, stride, hipMemcpyDeviceToDevice));
    bytes_of[v] = sizes[0];
  }
  for (uint32_t u = 0; u < n; ++u) for (uint32_t v = 0; v < n; ++v) {
    if (u == v) continue;
    int rc = merge_packed(ctxs[u], slot[v].p, bytes_of[v], (int)v); if (rc) { release(); return rc; }
  }
  release();
  return SQ_OK;
}

extern "C" int sq_dist_allreduce_u64(sq_dist* d, uint64_t* host, size_t n) {   // element-wise sums over the ranks (counters)
  if (!d || (!host && n)) { sq_set_error("sq_dist_allreduce_u64: bad arguments"); return SQ_ERR_ARG; }
  if (n == 0) return SQ_OK;
  SQ_HIP_CHECK(hipSetDevice(d->device));
  if (d->send.ensure(n * 8 + 16)) { sq_set_error("device allocation failed (dist all-reduce)"); return SQ_ERR_NOMEM; }
  SQ_HIP_CHECK(hipMemcpyAsync(d->send.p, host, n * 8, hipMemcpyHostToDevice, d->st));
  SQ_NCCL(rccl()->AllReduce(d->send.p, d->send.p, n, ncclUint64, ncclSum, d->comm, d->st));
  SQ_HIP_CHECK(hipMemcpyAsync(host, d->send.p, n * 8, hipMemcpyDeviceToHost, d->st));
  SQ_HIP_CHECK(hipStreamSynchronize(d->st));
  return SQ_OK;
}
extern "C" int sq_dist_bcast(sq_dist* d, void* host, size_t bytes, int root) {
  if (!d || (!host && bytes) || root < 0 || root >= d->world) { sq_set_error("sq_dist_bcast: bad arguments"); return SQ_ERR_ARG; }
  if (bytes == 0) return SQ_OK;
  SQ_HIP_CHECK(hipSetDevice(d->device));
  if (d->send.ensure(bytes + 16)) { sq_set_error("device allocation failed (dist broadcast)"); return SQ_ERR_NOMEM; }
  if (d->rank == root) SQ_HIP_CHECK(hipMemcpyAsync(d->send.p, host, bytes, hipMemcpyHostToDevice, d->st));
  SQ_NCCL(rccl()->Broadcast(d->send.p, d->send.p, bytes, ncclChar, root, d->comm, d->st));
  SQ_HIP_CHECK(hipMemcpyAsync(host, d->send.p, bytes, hipMemcpyDeviceToHost, d->st));
  SQ_HIP_CHECK(hipStreamSynchronize(d->st));
  return SQ_OK;
}
// every rank's `bytes` of host data to every rank: out holds rank r's block at r * bytes (equal sizes)
extern "C" int sq_dist_allgather(sq_dist* d, const void* host_in, size_t bytes, void* host_out) {
  if (!d || !host_in || !host_out) { sq_set_error("sq_dist_allgather: bad arguments"); return SQ_ERR_ARG; }
  if (bytes == 0) return SQ_OK;
  SQ_HIP_CHECK(hipSetDevice(d->device));
  if (d->send.ensure(bytes + 16) || d->recv.ensure(bytes * d->world + 16)) { sq_set_error("device allocation failed (dist all-gather)"); return SQ_ERR_NOMEM; }
  SQ_HIP_CHECK(hipMemcpyAsync(d->send.p, host_in, bytes, hipMemcpyHostToDevice, d->st));
  SQ_NCCL(rccl()->AllGather(d->send.p, d->recv.p, bytes, ncclChar, d->comm, d->st));
  SQ_HIP_CHECK(hipMemcpyAsync(host_out, d->recv.p, bytes * d->world, hipMemcpyDeviceToHost, d->st));
  SQ_HIP_CHECK(hipStreamSynchronize(d->st));
  return SQ_OK;
}
extern "C" int sq_dist_barrier(sq_dist* d) { uint64_t x = 1; return sq_dist_allreduce_u64(d, &x, 1); }

// SPEC §MG: the per-transcript online state of R ranks -> the state the inference tail sees, identical on every rank:
//   unique / total counts add; masses combine by logAdd in rank order 0..R-1 (LOG_0 = "no mass" is the identity); effective lengths
//   are rank 0's (its fragment-length distribution).
extern "C" int sq_dist_reduce_model(sq_dist* d, uint32_t M, double* log_mass, uint64_t* unique_count, uint64_t* total_count, double* log_eff_len) {
  if (!d || !log_mass || !unique_count || !total_count || !log_eff_len) { sq_set_error("sq_dist_reduce_model: bad arguments"); return SQ_ERR_ARG; }
  int rc = sq_dist_allreduce_u64(d, unique_count, M); if (rc) return rc;
  rc = sq_dist_allreduce_u64(d, total_count, M); if (rc) return rc;
  std::vector<double> all((size_t)M * d->world);
  rc = sq_dist_allgather(d, log_mass, (size_t)M * 8, all.data()); if (rc) return rc;
  sq_merge_log_masses(M, (uint32_t)d->world, all.data(), log_mass);
  return sq_dist_bcast(d, log_eff_len, (size_t)M * 8, 0);
}

// replicates / Gibbs chains by rank: the contiguous share [first, first + count) of `total` items cut at multiples of `unit`
// (unit = 1 for bootstrap replicates; unit = sq_gibbs_chain_step(total) for Gibbs samples, where a chain stays on one GPU and the
// samples past the last chain start belong to the last chain — CollapsedGibbsSampler.cpp:452-455)
extern "C" void sq_dist_share(const sq_dist* d, uint32_t total, uint32_t unit, uint32_t* first, uint32_t* count) {
  const uint32_t W = d ? (uint32_t)d->world : 1u, r = d ? (uint32_t)d->rank : 0u;
  if (unit == 0) unit = 1;
  const uint32_t nu = std::max(1u, total / unit);                // whole units; a remainder rides with the last one
  const uint32_t lo = (uint32_t)(((uint64_t)nu * r) / W), hi = (uint32_t)(((uint64_t)nu * (r + 1)) / W);
  const uint32_t a = std::min(total, lo * unit), b = (hi == nu) ? total : std::min(total, hi * unit);
  *first = a; *count = b > a ? b - a : 0;
}
