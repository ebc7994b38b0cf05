/* demod_reference_signals_impl.h -- HIP-backed body of gr::dvbt::demod_reference_signals (replaces lib/demod_reference_signals_impl.h of gr-dvbt; see hip_shell.h) */
#ifndef INCLUDED_DVBT_DEMOD_REFERENCE_SIGNALS_IMPL_HIP_H
#define INCLUDED_DVBT_DEMOD_REFERENCE_SIGNALS_IMPL_HIP_H

#include <dvbt/demod_reference_signals.h>
#include "hip_shell.h"

namespace gr {
  namespace dvbt {

    class demod_reference_signals_impl : public demod_reference_signals
    {
      DVBT_HIP_SHELL_MEMBERS(demod_reference_signals)
    public:
      demod_reference_signals_impl(int itemsize, int ninput, int noutput, dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_code_rate_t code_rate_HP, dvbt_code_rate_t code_rate_LP, dvbt_guard_interval_t guard_interval, dvbt_transmission_mode_t transmission_mode, int include_cell_id, int cell_id);
      ~demod_reference_signals_impl() {}
    };

  } // namespace dvbt
} // namespace gr

#endif
