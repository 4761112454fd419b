// hip/inflate_dev.h — the device BGZF inflater as the reader sees it (hip/inflate_dev.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
// one member: its raw deflate stream (the bytes between the gzip header and the 8-byte trailer) at comp + coff, csize bytes; its text goes to text + voff,
// isize bytes (from the trailer) with the CRC-32 the trailer names.  csize == 0: no stream — isize (<= 64) line ends are written
struct sq_bgzf_member { uint64_t coff; uint64_t voff; uint32_t csize, isize, crc, _pad; };
// status: two words, [0] preset to 0xFFFFFFFF; after the kernel [0] = index + 1 of the first damaged member (unchanged if none), [1] = what was wrong (sq_bgzf_status_text)
int sq_bgzf_inflate_launch(const uint8_t* d_comp, const sq_bgzf_member* d_mem, uint32_t nmem, uint8_t* d_text, uint32_t* d_status, hipStream_t st);
const char* sq_bgzf_status_text(uint32_t what);
