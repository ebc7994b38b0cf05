/* bit_inner_deinterleaver_impl.h -- HIP-backed body of gr::dvbt::bit_inner_deinterleaver (replaces lib/bit_inner_deinterleaver_impl.h of gr-dvbt; see hip_shell.h) */
#ifndef INCLUDED_DVBT_BIT_INNER_DEINTERLEAVER_IMPL_HIP_H
#define INCLUDED_DVBT_BIT_INNER_DEINTERLEAVER_IMPL_HIP_H

#include <dvbt/bit_inner_deinterleaver.h>
#include "hip_shell.h"

namespace gr {
  namespace dvbt {

    class bit_inner_deinterleaver_impl : public bit_inner_deinterleaver
    {
      DVBT_HIP_SHELL_MEMBERS(bit_inner_deinterleaver)
    public:
      bit_inner_deinterleaver_impl(int nsize, dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_transmission_mode_t transmission);
      ~bit_inner_deinterleaver_impl() {}
    };

  } // namespace dvbt
} // namespace gr

#endif
