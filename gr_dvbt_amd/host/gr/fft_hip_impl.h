/* fft_hip_impl.h -- HIP-backed body of gr::dvbt::fft_hip (replaces lib/fft_hip_impl.h of gr-dvbt; see hip_shell.h) */
#ifndef INCLUDED_DVBT_FFT_HIP_IMPL_HIP_H
#define INCLUDED_DVBT_FFT_HIP_IMPL_HIP_H

#include <dvbt/fft_hip.h>
#include "hip_shell.h"

namespace gr {
  namespace dvbt {

    class fft_hip_impl : public fft_hip
    {
      DVBT_HIP_SHELL_MEMBERS(fft)
    public:
      fft_hip_impl(int fft_size, bool forward, bool shift);
      ~fft_hip_impl() {}
    };

  } // namespace dvbt
} // namespace gr

#endif
