/* energy_descramble_impl.h -- HIP-backed body of gr::dvbt::energy_descramble (replaces lib/energy_descramble_impl.h of gr-dvbt; see hip_shell.h) */
#ifndef INCLUDED_DVBT_ENERGY_DESCRAMBLE_IMPL_HIP_H
#define INCLUDED_DVBT_ENERGY_DESCRAMBLE_IMPL_HIP_H

#include <dvbt/energy_descramble.h>
#include "hip_shell.h"

namespace gr {
  namespace dvbt {

    class energy_descramble_impl : public energy_descramble
    {
      DVBT_HIP_SHELL_MEMBERS(energy_descramble)
    public:
      energy_descramble_impl(int nblocks);
      ~energy_descramble_impl() {}
    };

  } // namespace dvbt
} // namespace gr

#endif
