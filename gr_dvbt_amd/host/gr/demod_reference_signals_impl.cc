/* demod_reference_signals_impl.cc -- gr::dvbt::demod_reference_signals on libdvbt_hip (replaces lib/demod_reference_signals_impl.cc
 * and the RX half of lib/reference_signals_impl.cc). */
#include "demod_reference_signals_impl.h"

namespace gr {
  namespace dvbt {

    demod_reference_signals::sptr
    demod_reference_signals::make(int itemsize, int ninput, int noutput, dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy,
                                  dvbt_code_rate_t code_rate_HP, dvbt_code_rate_t code_rate_LP, dvbt_guard_interval_t guard_interval,
                                  dvbt_transmission_mode_t transmission_mode, int include_cell_id, int cell_id)
    {
      return gnuradio::get_initial_sptr(new demod_reference_signals_impl(itemsize, ninput, noutput, constellation, hierarchy, code_rate_HP,
                                                                         code_rate_LP, guard_interval, transmission_mode, include_cell_id, cell_id));
    }

    static dvbt_demod_reference_signals_params demod_params(int itemsize, int ninput, int noutput, int c, int h, int hp, int lp, int gi, int tm, int inc, int id)
    { dvbt_demod_reference_signals_params p = { itemsize, ninput, noutput, c, h, hp, lp, gi, tm, inc, id }; return p; }

    /* io signatures: lib/demod_reference_signals_impl.cc:61-63.  Tags in: sync_start; out: superframe_start (once per hunt) and
     * symbol_index (one per produced item; the reference attaches one per call, which is the same thing in its one-item regime: B-7) */
    demod_reference_signals_impl::demod_reference_signals_impl(int itemsize, int ninput, int noutput, dvbt_constellation_t constellation,
                                                               dvbt_hierarchy_t hierarchy, dvbt_code_rate_t code_rate_HP, dvbt_code_rate_t code_rate_LP,
                                                               dvbt_guard_interval_t guard_interval, dvbt_transmission_mode_t transmission_mode,
                                                               int include_cell_id, int cell_id)
      : block("demod_reference_signals", io_signature::make(1, 1, itemsize * ninput), io_signature::make(1, 1, itemsize * noutput)),
        DVBT_HIP_CORE_INIT(demod_reference_signals, demod_params(itemsize, ninput, noutput, (int)constellation, (int)hierarchy, (int)code_rate_HP,
                                                                 (int)code_rate_LP, (int)guard_interval, (int)transmission_mode, include_cell_id, cell_id))
    {
    }

  } /* namespace dvbt */
} /* namespace gr */
