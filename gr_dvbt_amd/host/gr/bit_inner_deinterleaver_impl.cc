/* bit_inner_deinterleaver_impl.cc -- gr::dvbt::bit_inner_deinterleaver on libdvbt_hip (replaces lib/bit_inner_deinterleaver_impl.cc).
 * Non-hierarchical modes: one output stream (the second output of :73-75 exists only for hierarchical transmission, which no RX
 * flowgraph of gr-dvbt uses; dvbt_bit_inner_deinterleaver_create refuses it). */
#include "bit_inner_deinterleaver_impl.h"

namespace gr {
  namespace dvbt {

    bit_inner_deinterleaver::sptr
    bit_inner_deinterleaver::make(int nsize, dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_transmission_mode_t transmission)
    { return gnuradio::get_initial_sptr(new bit_inner_deinterleaver_impl(nsize, constellation, hierarchy, transmission)); }

    static dvbt_bit_inner_deinterleaver_params bit_params(int nsize, int c, int h, int t)
    { dvbt_bit_inner_deinterleaver_params p = { nsize, c, h, t }; return p; }

    /* io signatures: lib/bit_inner_deinterleaver_impl.cc:73-75 */
    bit_inner_deinterleaver_impl::bit_inner_deinterleaver_impl(int nsize, dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy,
                                                               dvbt_transmission_mode_t transmission)
      : block("bit_inner_deinterleaver", io_signature::make(1, 1, sizeof(unsigned char) * nsize), io_signature::make(1, 2, sizeof(unsigned char) * nsize)),
        DVBT_HIP_CORE_INIT(bit_inner_deinterleaver, bit_params(nsize, (int)constellation, (int)hierarchy, (int)transmission))
    {
    }

  } /* namespace dvbt */
} /* namespace gr */
