/* fft_hip_impl.cc -- gr::dvbt::fft_hip: the forward, shifted FFT of the RX flowgraphs (fft_vxx_0) on libdvbt_hip. */
#include "fft_hip_impl.h"

namespace gr {
  namespace dvbt {

    fft_hip::sptr
    fft_hip::make(int fft_size, bool forward, bool shift)
    { return gnuradio::get_initial_sptr(new fft_hip_impl(fft_size, forward, shift)); }

    static dvbt_fft_params fft_params(int fft_size, bool forward, bool shift)
    { dvbt_fft_params p = { fft_size, forward ? 1 : 0, shift ? 1 : 0 }; return p; }

    /* item layout of gr::fft::fft_vcc: vectors of fft_size complex in and out */
    fft_hip_impl::fft_hip_impl(int fft_size, bool forward, bool shift)
      : block("fft_hip", io_signature::make(1, 1, sizeof(gr_complex) * fft_size), io_signature::make(1, 1, sizeof(gr_complex) * fft_size)),
        DVBT_HIP_CORE_INIT(fft, fft_params(fft_size, forward, shift))
    {
    }

  } /* namespace dvbt */
} /* namespace gr */
