// host/index.h — in-memory form of the salmon-hip index (host side) and its device mirror.
#pragma once
#include <string>
#include <vector>
#include <cstdint>
#include "../sq_internal.h"
#include "../../../include/salmon_hip.h"

struct sq_device_index;  // defined in hip/device_index.h

struct sq_index {
  uint32_t k = 31, m = 20;
  uint32_t first_decoy = 0;
  uint64_t num_kmers = 0;
  std::vector<std::string> names;
  std::vector<uint32_t> ref_len, ref_clen;
  std::vector<uint64_t> ref_accum;  // [nrefs+1]
  std::vector<uint32_t> gcpre;      // G/C count before every word of refseq (built on demand: sq_index_gc_prefix)
  std::vector<uint64_t> refseq;     // 2-bit, padded by 2 words
  std::vector<uint64_t> useq;       // unitig pool, padded by 2 words
  std::vector<uint64_t> uoff;       // [U+1]
  std::vector<uint64_t> ctab_off;   // [U+1]
  std::vector<uint64_t> ctab;       // tid<<32 | fw<<31 | pos
  // dictionary
  uint32_t n_parts = 0;
  std::vector<uint64_t> part_slot_off;
  std::vector<uint32_t> part_bkt_off;
  std::vector<uint16_t> pilots;
  std::vector<uint64_t> slots, entries, skew_keys, skew_vals;
  // build statistics (info.json)
  uint64_t num_minimizers = 0, num_superkmers = 0, num_skew_kmers = 0, max_bucket = 0;
  std::vector<std::pair<std::string, std::string>> duplicates;  // retained, duplicate
  // SHA-2 digests of the input records (SalmonIndex.hpp:94-98): [0] SeqHash, [1] NameHash, [2] SeqHash512, [3] NameHash512 over the targets,
  // [4] DecoySeqHash, [5] DecoyNameHash over the decoys; info.json carries them, meta_info.json repeats them
  std::string hashes[6];
  bool keep_duplicates = false;
  // device mirror (owned; created by sq_index_to_device)
  sq_device_index* dev = nullptr;

  sq_dict_view host_view() const {
    sq_dict_view v;
    v.k = k; v.m = m; v.n_parts = n_parts;
    v.part_slot_off = part_slot_off.data(); v.part_bkt_off = part_bkt_off.data();
    v.pilots = pilots.data(); v.slots = slots.data(); v.entries = entries.data();
    v.skew_keys = skew_keys.data(); v.skew_vals = skew_vals.data();
    v.skew_mask = skew_keys.empty() ? 0 : skew_keys.size() - 1;
    v.useq = useq.data(); v.uoff = uoff.data(); v.num_unitigs = uoff.empty() ? 0 : uoff.size() - 1;
    v.kfilter = nullptr; v.kfilter_words = 0; v.uinfo = nullptr; v.mtab = nullptr; v.mtab_buckets = 0;   // the filter and the interleaved unitig bounds live on the device only (built at upload)
    return v;
  }
};

void sq_set_error(const char* fmt, ...);
int sq_index_save(const sq_index& idx, const std::string& dir);
int sq_index_load_host(const std::string& dir, sq_index** out);
void sq_device_index_free(sq_device_index*);  // hip side
const std::vector<uint32_t>& sq_index_gc_prefix(sq_index* idx);   // host/bias.cpp: builds idx->gcpre once

// small parallel-for helper (dynamic chunks)
#include <thread>
#include <atomic>
#include <functional>
template <class F>
static inline void sq_parallel_for(uint64_t n, uint32_t nthreads, uint64_t chunk, F fn) {
  if (nthreads <= 1 || n <= chunk) { fn(0, n, 0u); return; }
  std::atomic<uint64_t> next(0);
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < nthreads; ++t)
    th.emplace_back([&, t]() {
      for (;;) {
        uint64_t b = next.fetch_add(chunk);
        if (b >= n) break;
        uint64_t e = b + chunk < n ? b + chunk : n;
        fn(b, e, t);
      }
    });
  for (auto& x : th) x.join();
}
