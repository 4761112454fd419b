// hip/index_dev.hip — sq_index_load / sq_index_to_device / sq_index_free (seam B0).
#include "device_index.h"
#include <algorithm>
#include <cstdlib>

template <class T>
static int up(sq_device_index* d, const std::vector<T>& v, const T** out) {
  size_t n = v.size() * sizeof(T);
  void* p = nullptr;
  SQ_HIP_CHECK(hipMalloc(&p, n ? n : 16));
  if (n) SQ_HIP_CHECK(hipMemcpy(p, v.data(), n, hipMemcpyHostToDevice));
  d->allocs.push_back(p); d->bytes += n; *out = (const T*)p;
  return SQ_OK;
}

// one wave per unitig (grid-stride): every k-mer of the string pool sets its four signature bits in its filter word
__global__ void k_build_kfilter(const uint64_t* __restrict__ useq, const uint64_t* __restrict__ uoff, uint64_t num_unitigs, uint32_t k, uint32_t m,
                                unsigned long long* __restrict__ filter, uint64_t nblocks) {
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  for (uint64_t u = wave; u < num_unitigs; u += nwaves) {
    const uint64_t b = uoff[u], e = uoff[u + 1];
    if (e - b < k) continue;
    for (uint64_t p = b + lane; p + k <= e; p += 64) {
      const uint64_t km = sq_fetch_bases(useq, p, k), rc = sq_revcomp(km, k);
      const uint64_t h = sq_kf_hash(km < rc ? km : rc);
      atomicOr(&filter[sq_kf_word_of(sq_minimizer(km, rc, k, m), h, nblocks)], (unsigned long long)sq_kf_mask(h));   // the block of the k-mer's minimizer (sq_internal.h)
    }
  }
}

// [r5] the minimizer table (sq_internal.h: sq_mtab_*): one wave per unitig walks its k-mers; where the minimizer changes from one k-mer to the next the
// new one is entered with the record the MPHF gives for it (claimed with one compare-and-swap on the key; a key met again is left alone — its
// record is the same).  A bucket is filled front to back and never emptied, which is what lets a search stop at the first bucket with a free place.
__global__ void k_build_mtab(sq_dict_view d, unsigned long long* __restrict__ tab, uint64_t nbuckets, unsigned long long* __restrict__ fail) {
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t lane = threadIdx.x & 63, k = d.k, m = d.m;
  for (uint64_t u = wave; u < d.num_unitigs; u += nwaves) {
    const uint64_t b = d.uoff[u], e = d.uoff[u + 1];
    if (e - b < k) continue;
    for (uint64_t p0 = b; p0 + k <= e; p0 += 64) {   // wave-uniform trip count: the shuffle below needs every lane
      const uint64_t p = p0 + lane; const bool valid = p + k <= e;
      uint64_t mini = ~0ULL;
      if (valid) { const uint64_t km = sq_fetch_bases(d.useq, p, k), rc = sq_revcomp(km, k); mini = sq_minimizer(km, rc, k, m); }
      const uint64_t prev = (uint64_t)__shfl_up((unsigned long long)mini, 1, 64);
      if (!valid || (lane != 0 && prev == mini)) continue;
      const uint64_t rec = d.slots[sq_mphf_slot(d, mini)];
      uint64_t bk = sq_mtab_bucket_of(mini, nbuckets); bool placed = false;
      for (uint64_t tries = 0; tries < nbuckets && !placed; ++tries) {
        for (uint32_t j = 0; j < SQ_MTAB_BUCKET && !placed; ++j) {
          unsigned long long* kp = tab + (bk * SQ_MTAB_BUCKET + j) * 2;
          const unsigned long long old = atomicCAS(kp, (unsigned long long)SQ_MTAB_EMPTY, (unsigned long long)mini);
          if (old == SQ_MTAB_EMPTY) { kp[1] = rec; placed = true; }
          else if (old == mini) placed = true;
        }
        if (++bk == nbuckets) bk = 0;
      }
      if (!placed) atomicAdd(fail, 1ULL);
    }
  }
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
  rc |= up(d, idx->refseq, &d->refseq); rc |= up(d, sq_index_gc_prefix(idx), &d->gcpre); rc |= up(d, idx->ctab_off, &d->ctab_off); rc |= up(d, idx->ctab, &d->ctab);
  if (rc) { sq_device_index_free(d); return SQ_ERR_DEVICE; }
  v.uinfo = nullptr;
  {   // unitig bounds of the string pool and of the contig table, interleaved (sq_internal.h)
    std::vector<uint64_t> ui(2 * idx->uoff.size());
    for (size_t u = 0; u < idx->uoff.size(); ++u) { ui[2 * u] = idx->uoff[u]; ui[2 * u + 1] = idx->ctab_off[u]; }
    rc |= up(d, ui, &v.uinfo);
    if (rc) { sq_device_index_free(d); return SQ_ERR_DEVICE; }
  }
  v.kfilter = nullptr; v.kfilter_words = 0;
  if (idx->num_kmers > 0) {   // k-mer membership filter (sq_internal.h): SQ_KF_BITS_PER_KEY bits per distinct k-mer
    const uint64_t nblocks = std::max<uint64_t>(1024, (idx->num_kmers * SQ_KF_BITS_PER_KEY + 511) / 512);   // 64-byte blocks, one per ~2.5 minimizers
    const uint64_t nwords = nblocks * SQ_KF_BLOCK_WORDS;
    void* p = nullptr;
    SQ_HIP_CHECK(hipMalloc(&p, nwords * 8));
    d->allocs.push_back(p); d->bytes += nwords * 8;
    SQ_HIP_CHECK(hipMemset(p, 0, nwords * 8));
    k_build_kfilter<<<4096, 256>>>(v.useq, v.uoff, v.num_unitigs, idx->k, idx->m, (unsigned long long*)p, nblocks);
    SQ_HIP_CHECK(hipDeviceSynchronize());
    v.kfilter = (const uint64_t*)p; v.kfilter_words = nwords;
  }
  v.mtab = nullptr; v.mtab_buckets = 0;
  if (v.kfilter && v.uinfo && idx->num_kmers > 0 && idx->k == 31 && v.m == 20) {   // what k_seed2 needs (map.hip); other k / m take the general kernel and the MPHF   // minimizer table (sq_internal.h): 1.6 minimizers per 64-byte bucket of four
    uint64_t nmin = 0; for (uint64_t r : idx->slots) nmin += r != SQ_SLOT_EMPTY;
    const uint64_t nb = nmin * 5 / 8 + 1024;
    void* p = nullptr; unsigned long long* fail = nullptr;
    SQ_HIP_CHECK(hipMalloc(&p, nb * 64));
    d->allocs.push_back(p); d->bytes += nb * 64;
    SQ_HIP_CHECK(hipMemset(p, 0xFF, nb * 64));
    SQ_HIP_CHECK(hipMalloc((void**)&fail, 8)); SQ_HIP_CHECK(hipMemset(fail, 0, 8));
    k_build_mtab<<<4096, 256>>>(v, (unsigned long long*)p, nb, fail);
    unsigned long long hfail = 0;
    SQ_HIP_CHECK(hipMemcpy(&hfail, fail, 8, hipMemcpyDeviceToHost)); (void)hipFree(fail);
    if (hfail) { sq_set_error("internal: %llu minimizers found no place in the minimizer table", hfail); sq_device_index_free(d); return SQ_ERR_STATE; }
    v.mtab = (const uint64_t*)p; v.mtab_buckets = nb;
  }
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
