/* convolutional_deinterleaver_impl.cc -- gr::dvbt::convolutional_deinterleaver on libdvbt_hip (replaces
 * lib/convolutional_deinterleaver_impl.cc).  superframe_start in: the input is realigned to the tag (:109-120), the FIFO history
 * is kept (B-16). */
#include "convolutional_deinterleaver_impl.h"

namespace gr {
  namespace dvbt {

    convolutional_deinterleaver::sptr
    convolutional_deinterleaver::make(int nsize, int I, int M)
    { return gnuradio::get_initial_sptr(new convolutional_deinterleaver_impl(nsize, I, M)); }

    static dvbt_convolutional_deinterleaver_params cdi_params(int blocks, int I, int M)
    { dvbt_convolutional_deinterleaver_params p = { blocks, I, M }; return p; }

    /* io signatures, rate and output multiple: lib/convolutional_deinterleaver_impl.cc:55-61 */
    convolutional_deinterleaver_impl::convolutional_deinterleaver_impl(int blocks, int I, int M)
      : block("convolutional_deinterleaver", io_signature::make(1, 1, sizeof(unsigned char)), io_signature::make(1, 1, sizeof(unsigned char) * I * blocks)),
        DVBT_HIP_CORE_INIT(convolutional_deinterleaver, cdi_params(blocks, I, M))
    {
      set_relative_rate(1.0 / (double)(I * blocks));
      set_output_multiple(2);
    }

  } /* namespace dvbt */
} /* namespace gr */
