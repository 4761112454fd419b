// tools/inflate_fuzz.cpp — the decoder of the device BGZF inflater (salmon_amd/csrc/hip/inflate_core.h: bit reader, tables, block parsing — the part that is one source
// for host and device) under AddressSanitizer + UBSan on the host: deflate streams of every strategy and level, then with flipped bits, cut short, or with a
// lied-about text size, each in exact-size heap buffers.  A sound stream must come back as its text; a damaged one may be refused or decode to something else
// (the member's CRC-32 is what catches that on the device), but no byte may be read or written out of place.  Built with hipcc --cuda-host-only (`make -C tools
// inflate_fuzz`), run by tests/test_inflate.py.  What it cannot see: the device's own output stage (tokens, lanes) — tests/test_inflate.py's GPU test does.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <zlib.h>
#include "hip/inflate_core.h"
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000; unsigned seed = 12345; auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  static sqinf::Tables T; long ok = 0, refused = 0, wrong = 0;
  for (int it = 0; it < iters; ++it) {
    // a text with repeats, compressed; then bytes flipped / cut / size lied about
    const size_t n = 1 + rnd() % 65535; std::vector<unsigned char> text(n);
    for (size_t i = 0; i < n; ++i) text[i] = (rnd() % 3 == 0 && i > 40) ? text[i - 1 - rnd() % 40] : (unsigned char)("ACGT\nI@+"[rnd() % 8]);
    z_stream zs{}; deflateInit2(&zs, (int)(rnd() % 10), Z_DEFLATED, -15, 8, rnd() % 7 == 0 ? Z_FIXED : (rnd() % 5 == 0 ? Z_HUFFMAN_ONLY : Z_DEFAULT_STRATEGY));
    std::vector<unsigned char> comp(deflateBound(&zs, n) + 64); zs.next_in = text.data(); zs.avail_in = (uInt)n; zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
    deflate(&zs, Z_FINISH); size_t cn = zs.total_out; deflateEnd(&zs);
    const int mode = (int)(rnd() % 4); size_t use = cn; unsigned isize = (unsigned)n;
    if (mode == 1) for (int k = 0; k < 1 + (int)(rnd() % 4); ++k) comp[rnd() % cn] ^= (unsigned char)(1u << (rnd() % 8));
    if (mode == 2) use = rnd() % cn;
    if (mode == 3) isize = (unsigned)(rnd() % 65537);
    // exact-size heap buffers so that the sanitizer sees any byte out of place (input: the decoder may read whole words: 3 bytes of slack, as the device buffers have)
    const size_t shift = rnd() % 4; unsigned char* in = (unsigned char*)malloc(use + shift + 4); memcpy(in + shift, comp.data(), use); unsigned char* out = (unsigned char*)malloc(isize ? isize : 1);
    const int rc = sqinf::inflate_member(in + shift, use, out, isize, T);
    if (rc == 0) { if (isize == n && !memcmp(out, text.data(), n)) ++ok; else if (mode == 0) ++wrong; else ++ok; } else { ++refused; if (mode == 0) ++wrong; }
    free(in); free(out);
  }
  printf("ok %ld refused %ld wrong %ld\n", ok, refused, wrong); return wrong ? 1 : 0;
}
