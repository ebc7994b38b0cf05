// rx_flowgraph_example.cpp -- the back half of apps/dvbt_rx_demo.grc (2k / QAM16 / rate 1/2) written
// against the C++ mirror of the gr::dvbt block API: same make() calls, same connection order.
// Reads a file of bit-de-interleaver output bytes (one constellation label per byte) and writes TS.
// Compile check only on machines without a GPU (the blocks' constructors throw there).
#include <cstdio>
#include <vector>
#include "dvbt_blocks.hpp"

using namespace gr::dvbt_amd;

int main(int argc, char **argv)
{
  if (argc < 3) { std::printf("usage: %s <bitdeint_bytes.bin> <out.ts>\n", argv[0]); return 2; }
  try {
    // parameters exactly as in apps/dvbt_rx_demo.grc
    viterbi_decoder::sptr vit = viterbi_decoder::make(DVBT_QAM16, DVBT_NH, DVBT_C1_2, 768, 0, -1);
    convolutional_deinterleaver::sptr dei = convolutional_deinterleaver::make(136, 12, 17);
    reed_solomon_dec::sptr rs = reed_solomon_dec::make(2, 8, 0x11d, 255, 239, 8, 51, 8);
    energy_descramble::sptr des = energy_descramble::make(8);

    std::FILE *f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("open"); return 1; }
    std::vector<unsigned char> in;
    unsigned char buf[65536]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) in.insert(in.end(), buf, buf + n);
    std::fclose(f);

    const int d_nsym = 768 * 2 / 4, d_nout = 768 / 8;
    int nblocks = (int)(in.size() / d_nsym);
    std::vector<unsigned char> v((size_t)nblocks * d_nout), d, r, ts;
    std::vector<tag_t> tin{{0, DVBT_TAG_SUPERFRAME_START, 0xaa}}, tout;
    int consumed = 0;
    int nv = vit->general_work(nblocks * d_nout, nblocks * d_nsym, in.data(), v.data(), tin, tout, consumed);
    int items = (nv / 1632) & ~1;
    d.resize((size_t)items * 1632); r.resize((size_t)items * 1504); ts.resize((size_t)items * 1504);
    dei->general_work(items, items * 1632, v.data(), d.data(), tout, tout, consumed);
    rs->general_work(items, items, d.data(), r.data(), {}, tout, consumed);
    int nts = des->general_work((items / 4) * 4 * 1504, items, r.data(), ts.data(), {}, tout, consumed);
    std::FILE *o = std::fopen(argv[2], "wb");
    std::fwrite(ts.data(), 1, (size_t)nts, o);
    std::fclose(o);
    std::printf("%d bytes in -> %d TS bytes\n", (int)in.size(), nts);
  } catch (const std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
  return 0;
}
