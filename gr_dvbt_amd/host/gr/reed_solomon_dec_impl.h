/* reed_solomon_dec_impl.h -- HIP-backed body of gr::dvbt::reed_solomon_dec (replaces lib/reed_solomon_dec_impl.h of gr-dvbt; see hip_shell.h) */
#ifndef INCLUDED_DVBT_REED_SOLOMON_DEC_IMPL_HIP_H
#define INCLUDED_DVBT_REED_SOLOMON_DEC_IMPL_HIP_H

#include <dvbt/reed_solomon_dec.h>
#include "hip_shell.h"

namespace gr {
  namespace dvbt {

    class reed_solomon_dec_impl : public reed_solomon_dec
    {
      DVBT_HIP_SHELL_MEMBERS(reed_solomon_dec)
    public:
      reed_solomon_dec_impl(int p, int m, int gfpoly, int n, int k, int t, int s, int blocks);
      ~reed_solomon_dec_impl() {}
    };

  } // namespace dvbt
} // namespace gr

#endif
