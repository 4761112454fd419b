// hip/gzip_dev.h — [r6] an ordinary gzip file (one deflate stream per member, no index, no block table) inflated ON THE DEVICE, segment by segment (gzip_dev.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>

struct sq_gzdev;
// `data` / `bytes`: the whole file, readable for the life of the object (a mapping).  `st`: the stream its work is queued on.  seg_bytes: compressed bytes per segment
// (0: 64 MB — some four thousand spans, a launch that fills the chip).  SQ_ERR_IO when the data does not start with a gzip member header.
int sq_gzdev_open(const uint8_t* data, size_t bytes, int device, hipStream_t st, size_t seg_bytes, sq_gzdev** out, std::string* err);
// the next segment is decoded as far as its size: *n = bytes of text it holds (0: the end of the file).  Synchronises with the stream.
int sq_gzdev_next(sq_gzdev*, size_t* n, std::string* err);
// its text is written to d_dst (room for the *n of sq_gzdev_next) by work queued on the decoder's own stream: returns at once, `done` (may be null) is recorded behind the
// text.  The members' checksums (CRC-32, length against the trailer) are looked at by a later sq_gzdev_next or by sq_gzdev_wait: a damaged file is refused a segment late.
int sq_gzdev_emit(sq_gzdev*, uint8_t* d_dst, hipEvent_t done, std::string* err);
// everything queued so far is complete and checked
int sq_gzdev_wait(sq_gzdev*, std::string* err);
void sq_gzdev_close(sq_gzdev*);
struct sq_gzdev_counters { uint64_t segments, spans, members, retries; };
sq_gzdev_counters sq_gzdev_stats(const sq_gzdev*);
