/* rx_hip_impl.h -- gr::dvbt::rx_hip over dvbt_rx_stream_* (see include/dvbt/rx_hip.h, hip_shell.h) */
#ifndef INCLUDED_DVBT_RX_HIP_IMPL_H
#define INCLUDED_DVBT_RX_HIP_IMPL_H

#include <dvbt/rx_hip.h>
#include <gnuradio/block_detail.h>
#include <gnuradio/buffer.h>
#include "hip_shell.h"

namespace gr {
  namespace dvbt {

    class rx_hip_impl : public rx_hip
    {
      ::dvbt_rx_stream *d_stream;
      bool d_finished;                                   /* dvbt_rx_stream_finish has run: only draining is left */
      enum { RX_HIP_MAX_BACKLOG = 8 << 20 };             /* decoded TS bytes that may wait for the sink before the block stops taking input */
      bool input_ended();
    public:
      rx_hip_impl(dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_code_rate_t code_rate, dvbt_guard_interval_t guard_interval,
                  dvbt_transmission_mode_t transmission_mode, float snr, int bsize, int segment_superframes, bool soft_decision);
      ~rx_hip_impl();
      void forecast(int noutput_items, gr_vector_int &ninput_items_required);
      int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items);
      bool stop();
    };

  } // namespace dvbt
} // namespace gr

#endif
