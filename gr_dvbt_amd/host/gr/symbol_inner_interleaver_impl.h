/* symbol_inner_interleaver_impl.h -- HIP-backed body of gr::dvbt::symbol_inner_interleaver (replaces lib/symbol_inner_interleaver_impl.h of gr-dvbt; see hip_shell.h) */
#ifndef INCLUDED_DVBT_SYMBOL_INNER_INTERLEAVER_IMPL_HIP_H
#define INCLUDED_DVBT_SYMBOL_INNER_INTERLEAVER_IMPL_HIP_H

#include <dvbt/symbol_inner_interleaver.h>
#include "hip_shell.h"

namespace gr {
  namespace dvbt {

    class symbol_inner_interleaver_impl : public symbol_inner_interleaver
    {
      DVBT_HIP_SHELL_MEMBERS(symbol_inner_interleaver)
    public:
      symbol_inner_interleaver_impl(int nsize, dvbt_transmission_mode_t transmission, int direction);
      ~symbol_inner_interleaver_impl() {}
    };

  } // namespace dvbt
} // namespace gr

#endif
