/* convolutional_deinterleaver_impl.h -- HIP-backed body of gr::dvbt::convolutional_deinterleaver (replaces lib/convolutional_deinterleaver_impl.h of gr-dvbt; see hip_shell.h) */
#ifndef INCLUDED_DVBT_CONVOLUTIONAL_DEINTERLEAVER_IMPL_HIP_H
#define INCLUDED_DVBT_CONVOLUTIONAL_DEINTERLEAVER_IMPL_HIP_H

#include <dvbt/convolutional_deinterleaver.h>
#include "hip_shell.h"

namespace gr {
  namespace dvbt {

    class convolutional_deinterleaver_impl : public convolutional_deinterleaver
    {
      DVBT_HIP_SHELL_MEMBERS(convolutional_deinterleaver)
    public:
      convolutional_deinterleaver_impl(int nsize, int I, int M);
      ~convolutional_deinterleaver_impl() {}
    };

  } // namespace dvbt
} // namespace gr

#endif
