// hip/index_dev.hip — sq_index_load / sq_index_to_device / sq_index_free (seam B0).
#include "device_index.h"

template <class T>
static int up(sq_device_index* d, const std::vector<T>& v, const T** out) {
  size_t n = v.size() * sizeof(T);
  void* p = nullptr;
  SQ_HIP_CHECK(hipMalloc(&p, n ? n : 16));
  if (n) SQ_HIP_CHECK(hipMemcpy(p, v.data(), n, hipMemcpyHostToDevice));
  d->allocs.push_back(p); d->bytes += n; *out = (const T*)p;
  return SQ_OK;
}

void sq_device_index_free(sq_device_index* d) {
  if (!d) return;
  if (d->device >= 0) (void)hipSetDevice(d->device);
  for (void* p : d->allocs) (void)hipFree(p);
  delete d;
}

extern "C" int sq_index_to_device(sq_index* idx, int device) {
  if (!idx || device < 0) { sq_set_error("sq_index_to_device: bad arguments"); return SQ_ERR_ARG; }
  if (idx->dev && idx->dev->device == device) return SQ_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) {
    sq_set_error("no HIP device %d (found %d): the mapping/EM path has no CPU fallback", device, ndev);
    return SQ_ERR_DEVICE;
  }
  SQ_HIP_CHECK(hipSetDevice(device));
  if (idx->dev) { sq_device_index_free(idx->dev); idx->dev = nullptr; }
  sq_device_index* d = new sq_device_index(); d->device = device;
  int rc = 0;
  sq_dict_view& v = d->dict;
  v.k = idx->k; v.m = idx->m; v.n_parts = idx->n_parts; v.num_unitigs = idx->uoff.size() - 1;
  v.skew_mask = idx->skew_keys.empty() ? 0 : idx->skew_keys.size() - 1;
  rc |= up(d, idx->part_slot_off, &v.part_slot_off); rc |= up(d, idx->part_bkt_off, &v.part_bkt_off);
  rc |= up(d, idx->pilots, &v.pilots); rc |= up(d, idx->slots, &v.slots); rc |= up(d, idx->entries, &v.entries);
  rc |= up(d, idx->skew_keys, &v.skew_keys); rc |= up(d, idx->skew_vals, &v.skew_vals);
  rc |= up(d, idx->useq, &v.useq); rc |= up(d, idx->uoff, &v.uoff);
  rc |= up(d, idx->ref_accum, &d->ref_accum); rc |= up(d, idx->ref_len, &d->ref_len); rc |= up(d, idx->ref_clen, &d->ref_clen);
  rc |= up(d, idx->refseq, &d->refseq); rc |= up(d, idx->ctab_off, &d->ctab_off); rc |= up(d, idx->ctab, &d->ctab);
  if (rc) { sq_device_index_free(d); return SQ_ERR_DEVICE; }
  d->k = idx->k; d->first_decoy = idx->first_decoy; d->num_refs = (uint32_t)idx->names.size();
  idx->dev = d;
  return SQ_OK;
}

extern "C" int sq_index_load(const char* dir, int device, sq_index** out) {
  if (!dir || !out) { sq_set_error("sq_index_load: bad arguments"); return SQ_ERR_ARG; }
  sq_index* idx = nullptr;
  int rc = sq_index_load_host(dir, &idx);
  if (rc) return rc;
  if (device >= 0) { rc = sq_index_to_device(idx, device); if (rc) { delete idx; return rc; } }
  *out = idx;
  return SQ_OK;
}

extern "C" void sq_index_free(sq_index* idx) {
  if (!idx) return;
  if (idx->dev) sq_device_index_free(idx->dev);
  delete idx;
}

extern "C" uint64_t sq_index_device_bytes(const sq_index* idx) { return idx && idx->dev ? idx->dev->bytes : 0; }
