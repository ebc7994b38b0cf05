/* ofdm_sym_acquisition_impl.cc -- gr::dvbt::ofdm_sym_acquisition on libdvbt_hip (replaces lib/ofdm_sym_acquisition_impl.cc). */
#include "ofdm_sym_acquisition_impl.h"

namespace gr {
  namespace dvbt {

    ofdm_sym_acquisition::sptr
    ofdm_sym_acquisition::make(int blocks, int fft_length, int occupied_tones, int cp_length, float snr)
    { return gnuradio::get_initial_sptr(new ofdm_sym_acquisition_impl(blocks, fft_length, occupied_tones, cp_length, snr)); }

    static dvbt_ofdm_sym_acquisition_params acq_params(int blocks, int fft_length, int occupied_tones, int cp_length, float snr)
    { dvbt_ofdm_sym_acquisition_params p = { blocks, fft_length, occupied_tones, cp_length, snr }; return p; }

    /* io signatures and rate: lib/ofdm_sym_acquisition_impl.cc:380-388 */
    ofdm_sym_acquisition_impl::ofdm_sym_acquisition_impl(int blocks, int fft_length, int occupied_tones, int cp_length, float snr)
      : block("ofdm_sym_acquisition",
              io_signature::make(1, 1, sizeof(gr_complex) * blocks),
              io_signature::make(1, 1, sizeof(gr_complex) * blocks * fft_length)),
        DVBT_HIP_CORE_INIT(ofdm_sym_acquisition, acq_params(blocks, fft_length, occupied_tones, cp_length, snr))
    {
      set_relative_rate(1.0 / (double)(cp_length + fft_length));
      /* unlike the reference (at most one item per call) the HIP block turns a whole input window into items: the scheduler
       * is free to offer many symbols at once; the sync_start tag is attached as in :353-360,507 */
    }

  } /* namespace dvbt */
} /* namespace gr */
