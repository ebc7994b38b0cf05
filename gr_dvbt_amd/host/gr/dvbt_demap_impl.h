/* dvbt_demap_impl.h -- HIP-backed body of gr::dvbt::dvbt_demap (replaces lib/dvbt_demap_impl.h of gr-dvbt; see hip_shell.h) */
#ifndef INCLUDED_DVBT_DVBT_DEMAP_IMPL_HIP_H
#define INCLUDED_DVBT_DVBT_DEMAP_IMPL_HIP_H

#include <dvbt/dvbt_demap.h>
#include "hip_shell.h"

namespace gr {
  namespace dvbt {

    class dvbt_demap_impl : public dvbt_demap
    {
      DVBT_HIP_SHELL_MEMBERS(demap)
    public:
      dvbt_demap_impl(int nsize, dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_transmission_mode_t transmission, float gain);
      ~dvbt_demap_impl() {}
    };

  } // namespace dvbt
} // namespace gr

#endif
