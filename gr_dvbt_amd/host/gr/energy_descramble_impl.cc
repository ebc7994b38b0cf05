/* energy_descramble_impl.cc -- gr::dvbt::energy_descramble on libdvbt_hip (replaces lib/energy_descramble_impl.cc). */
#include "energy_descramble_impl.h"

namespace gr {
  namespace dvbt {

    energy_descramble::sptr
    energy_descramble::make(int nblocks)
    { return gnuradio::get_initial_sptr(new energy_descramble_impl(nblocks)); }

    static dvbt_energy_descramble_params ed_params(int nblocks)
    { dvbt_energy_descramble_params p = { nblocks }; return p; }

    /* io signatures, rate and output multiple: lib/energy_descramble_impl.cc:79-87 (items of nblocks packets of 188 bytes in, bytes out) */
    energy_descramble_impl::energy_descramble_impl(int nblocks)
      : block("energy_descramble", io_signature::make(1, 1, sizeof(unsigned char) * nblocks * 188), io_signature::make(1, 1, sizeof(unsigned char))),
        DVBT_HIP_CORE_INIT(energy_descramble, ed_params(nblocks))
    {
      set_relative_rate((double)(nblocks * 188));
      set_output_multiple(4 * nblocks * 188);
    }

  } /* namespace dvbt */
} /* namespace gr */
