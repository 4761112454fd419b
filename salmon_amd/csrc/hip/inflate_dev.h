// hip/inflate_dev.h — the device BGZF inflater as the reader sees it (hip/inflate_dev.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
// one member: its raw deflate stream (the bytes between the gzip header and the 8-byte trailer) at comp + coff, csize bytes; its text goes to text + voff,
// isize bytes (from the trailer) with the CRC-32 the trailer names.  flags & SQ_BGZF_LINE_END: no stream — the reader's pseudo-member, isize (<= 64) line ends are
// written.  A member WITHOUT that flag and without a stream (csize == 0) that claims text is damaged input and is reported as such
constexpr uint32_t SQ_BGZF_LINE_END = 1u;
struct sq_bgzf_member { uint64_t coff; uint64_t voff; uint32_t csize, isize, crc, flags; };
// status: two words, [0] preset to 0xFFFFFFFF; after the kernel [0] = index + 1 of the first damaged member (unchanged if none), [1] = what was wrong with a damaged member (sq_bgzf_status_text; with several damaged members not necessarily the first one's reason)
int sq_bgzf_inflate_launch(const uint8_t* d_comp, const sq_bgzf_member* d_mem, uint32_t nmem, uint8_t* d_text, uint32_t* d_status, hipStream_t st);
const char* sq_bgzf_status_text(uint32_t what);
