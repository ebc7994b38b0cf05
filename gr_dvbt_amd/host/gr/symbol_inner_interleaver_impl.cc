/* symbol_inner_interleaver_impl.cc -- gr::dvbt::symbol_inner_interleaver on libdvbt_hip (replaces lib/symbol_inner_interleaver_impl.cc).
 * direction 0 (RX) reads one symbol_index tag per item (:172-176,199); direction 1 (TX) counts by itself. */
#include "symbol_inner_interleaver_impl.h"

namespace gr {
  namespace dvbt {

    symbol_inner_interleaver::sptr
    symbol_inner_interleaver::make(int nsize, dvbt_transmission_mode_t transmission, int direction)
    { return gnuradio::get_initial_sptr(new symbol_inner_interleaver_impl(nsize, transmission, direction)); }

    static dvbt_symbol_inner_interleaver_params sym_params(int nsize, int t, int direction)
    { dvbt_symbol_inner_interleaver_params p = { nsize, t, direction }; return p; }

    /* io signatures: lib/symbol_inner_interleaver_impl.cc:111-113 */
    symbol_inner_interleaver_impl::symbol_inner_interleaver_impl(int nsize, dvbt_transmission_mode_t transmission, int direction)
      : block("symbol_inner_interleaver", io_signature::make(1, 1, sizeof(unsigned char) * nsize), io_signature::make(1, 1, sizeof(unsigned char) * nsize)),
        DVBT_HIP_CORE_INIT(symbol_inner_interleaver, sym_params(nsize, (int)transmission, direction))
    {
    }

  } /* namespace dvbt */
} /* namespace gr */
