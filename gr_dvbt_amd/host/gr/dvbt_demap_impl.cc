/* dvbt_demap_impl.cc -- gr::dvbt::dvbt_demap on libdvbt_hip (replaces lib/dvbt_demap_impl.cc). */
#include "dvbt_demap_impl.h"

namespace gr {
  namespace dvbt {

    dvbt_demap::sptr
    dvbt_demap::make(int nsize, dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_transmission_mode_t transmission, float gain)
    { return gnuradio::get_initial_sptr(new dvbt_demap_impl(nsize, constellation, hierarchy, transmission, gain)); }

    static dvbt_demap_params demap_params(int nsize, int c, int h, int t, float gain)
    { dvbt_demap_params p = { nsize, c, h, t, gain }; return p; }

    /* io signatures: lib/dvbt_demap_impl.cc:57-59 */
    dvbt_demap_impl::dvbt_demap_impl(int nsize, dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy,
                                     dvbt_transmission_mode_t transmission, float gain)
      : block("dvbt_demap", io_signature::make(1, 1, sizeof(gr_complex) * nsize), io_signature::make(1, 1, sizeof(unsigned char) * nsize)),
        DVBT_HIP_CORE_INIT(demap, demap_params(nsize, (int)constellation, (int)hierarchy, (int)transmission, gain))
    {
    }

  } /* namespace dvbt */
} /* namespace gr */
