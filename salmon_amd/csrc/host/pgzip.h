// host/pgzip.h — a plain gzip file inflated by several threads (the gzip floor of the read pipeline).
//
// A deflate stream is sequential in two ways: block boundaries are only known by decoding, and a block may copy from the 32 KB of text before
// it.  Both are worked around the way pugz does (Kerbiriou & Chikhi 2019, "Parallel decompression of gzip-compressed files and random access
// to DNA sequences"): the compressed file is cut into pieces; every piece but the first FINDS a block start by trying bit offsets until one
// parses as a dynamic block whose literals are text, then decodes into 16-bit symbols where "byte k of the unknown 32 KB window" is a symbol
// of its own, so copies out of the unknown window stay symbolic; once the pieces before it are done their last 32 KB resolve the symbols.
// Nothing is taken on trust: a piece must END exactly on the bit where its successor started (a successor whose start was not a real block
// boundary is simply decoded through), and every member's CRC-32 and length are checked against its trailer.
// The reference reads .gz through one zlib stream per file (include/salmon/internal/io/FastxReader.hpp:13-32).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>

struct PgzStream;
// `submit` runs a task on the caller's worker pool.  Returns nullptr when the data does not start with a gzip member header.
PgzStream* pgz_open(const uint8_t* data, size_t bytes, std::function<void(std::function<void()>)> submit, unsigned threads, size_t piece_bytes);
// up to `want` bytes of text in file order; 0 at the end; -1 on error (*err says what)
long pgz_read(PgzStream*, char* dst, size_t want, std::string* err);
// the same text without the copy: the next whole buffer of the stream (a few megabytes; 1), the end (0) or an error (-1).  `hold` keeps the
// buffer alive — also past pgz_close — and hands it back for reuse when the last reference goes.  Do not mix with pgz_read.
// [r4] a whole raw deflate stream of known output size (a BGZF member) into dst, which has room for isize + 32 bytes (or is followed by the next stream's place):
// the bytes written (== isize for a sound member), -1 for a damaged stream.  ~2 x zlib's inflate on sequence text.
long pgz_inflate_raw(const uint8_t* deflate, size_t n, char* dst, size_t isize);
struct PgzBuf { const char* p = nullptr; size_t n = 0; std::shared_ptr<void> hold; };
int pgz_next(PgzStream*, PgzBuf* out, std::string* err);
void pgz_close(PgzStream*);
// what happened, for tests and SQ_TIMING: pieces decoded, pieces whose start was not confirmed by the predecessor (decoded twice), members
struct pgz_counters { uint64_t pieces, resynced, members, rounds; };
pgz_counters pgz_stats(const PgzStream*);
