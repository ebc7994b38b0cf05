// rx_flowgraph_example.cpp -- the back half of apps/dvbt_rx_demo.grc (2k / QAM16 / rate 1/2) written
// against the C++ mirror of the gr::dvbt block API: same make() calls, same connection order.
// Reads a file of bit-de-interleaver output bytes (one constellation label per byte) and writes TS.
// The four blocks are chained on DEVICE buffers (general_work_device = dvbt_<blk>_work_device): the items
// go from the Viterbi decoder to the descrambler in HBM; only the input file and the TS cross PCIe.
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
    const int nblocks = (int)(in.size() / d_nsym);
    const size_t cap = (size_t)nblocks * d_nout + 4096;
    // device buffers between the blocks (libdvbt_hip's own allocator: this program does not link the HIP runtime)
    unsigned char *d_in = (unsigned char *)dvbt_device_malloc(in.size() + 64), *d_v = (unsigned char *)dvbt_device_malloc(cap),
                  *d_d = (unsigned char *)dvbt_device_malloc(cap), *d_r = (unsigned char *)dvbt_device_malloc(cap), *d_ts = (unsigned char *)dvbt_device_malloc(cap);
    if (!d_in || !d_v || !d_d || !d_r || !d_ts) throw std::runtime_error(dvbt_last_error());
    check(dvbt_copy_to_device(d_in, in.data(), in.size()));
    std::vector<tag_t> tin{{0, DVBT_TAG_SUPERFRAME_START, 0xaa}}, tout, none;
    int consumed = 0;
    const int nv = vit->general_work_device(nblocks * d_nout, nblocks * d_nsym, d_in, d_v, tin, tout, consumed);
    const int items = (nv / 1632) & ~1;
    dei->general_work_device(items, items * 1632, d_v, d_d, tout, tout, consumed);
    rs->general_work_device(items, items, d_d, d_r, none, tout, consumed);
    const int nts = des->general_work_device((items / 4) * 4 * 1504, items, d_r, d_ts, none, tout, consumed);
    check(dvbt_synchronize(nullptr));
    std::vector<unsigned char> ts((size_t)(nts > 0 ? nts : 0));
    if (nts > 0) check(dvbt_copy_to_host(ts.data(), d_ts, ts.size()));
    for (unsigned char *q : {d_in, d_v, d_d, d_r, d_ts}) dvbt_device_free(q);
    std::FILE *o = std::fopen(argv[2], "wb");
    std::fwrite(ts.data(), 1, ts.size(), o);
    std::fclose(o);
    std::printf("%d bytes in -> %d TS bytes\n", (int)in.size(), nts);
  } catch (const std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
  return 0;
}
