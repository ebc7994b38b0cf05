/* viterbi_decoder_impl.h -- HIP-backed body of gr::dvbt::viterbi_decoder (replaces lib/viterbi_decoder_impl.h of gr-dvbt; see hip_shell.h) */
#ifndef INCLUDED_DVBT_VITERBI_DECODER_IMPL_HIP_H
#define INCLUDED_DVBT_VITERBI_DECODER_IMPL_HIP_H

#include <dvbt/viterbi_decoder.h>
#include "hip_shell.h"

namespace gr {
  namespace dvbt {

    class viterbi_decoder_impl : public viterbi_decoder
    {
      DVBT_HIP_SHELL_MEMBERS(viterbi_decoder)
    public:
      viterbi_decoder_impl(dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_code_rate_t coderate, int bsize, int S0, int SK);
      ~viterbi_decoder_impl() {}
    };

  } // namespace dvbt
} // namespace gr

#endif
