/* dvbt/fft_hip.h -- public interface of the one block of this directory that gr-dvbt itself does not declare: the forward FFT
 * between ofdm_sym_acquisition and demod_reference_signals, which the flowgraphs take from gr-fft (fft_vxx_0 = gr::fft::fft_vcc,
 * forward, rectangular window, shift=True: apps/dvbt_rx_demo*.grc).  Same item layout in and out; drop-in for that instance. */
#ifndef INCLUDED_DVBT_FFT_HIP_H
#define INCLUDED_DVBT_FFT_HIP_H

#include <dvbt/api.h>
#include <gnuradio/block.h>

namespace gr {
  namespace dvbt {

    class DVBT_API fft_hip : virtual public block
    {
    public:
      typedef boost::shared_ptr<fft_hip> sptr;
      /* fft_size: 64..8192, a power of two; forward and shift must be true (the only configuration the RX flowgraphs use) */
      static sptr make(int fft_size, bool forward = true, bool shift = true);
    };

  } // namespace dvbt
} // namespace gr

#endif
