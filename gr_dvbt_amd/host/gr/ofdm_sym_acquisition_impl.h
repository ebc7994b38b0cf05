/* ofdm_sym_acquisition_impl.h -- HIP-backed body of gr::dvbt::ofdm_sym_acquisition (replaces lib/ofdm_sym_acquisition_impl.h of gr-dvbt; see hip_shell.h) */
#ifndef INCLUDED_DVBT_OFDM_SYM_ACQUISITION_IMPL_HIP_H
#define INCLUDED_DVBT_OFDM_SYM_ACQUISITION_IMPL_HIP_H

#include <dvbt/ofdm_sym_acquisition.h>
#include "hip_shell.h"

namespace gr {
  namespace dvbt {

    class ofdm_sym_acquisition_impl : public ofdm_sym_acquisition
    {
      DVBT_HIP_SHELL_MEMBERS(ofdm_sym_acquisition)
    public:
      ofdm_sym_acquisition_impl(int blocks, int fft_length, int occupied_tones, int cp_length, float snr);
      ~ofdm_sym_acquisition_impl() {}
    };

  } // namespace dvbt
} // namespace gr

#endif
