/* rx_hip_impl.cc -- gr::dvbt::rx_hip: cfloat in, TS bytes out, the ten receive blocks of apps/dvbt_rx_demo*.grc in one (dvbt_rx_stream_*).
 * general_work pushes whatever the scheduler offers (any size) and hands out the TS bytes that are ready; the library cuts the stream into
 * pieces of whole superframes, decodes them device resident and stitches the packets, byte-identical to the chain of single blocks. */
#include "rx_hip_impl.h"

namespace gr {
  namespace dvbt {

    rx_hip::sptr
    rx_hip::make(dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_code_rate_t code_rate, dvbt_guard_interval_t guard_interval,
                 dvbt_transmission_mode_t transmission_mode, float snr, int bsize, int segment_superframes, bool soft_decision)
    { return gnuradio::get_initial_sptr(new rx_hip_impl(constellation, hierarchy, code_rate, guard_interval, transmission_mode, snr, bsize, segment_superframes, soft_decision)); }

    rx_hip_impl::rx_hip_impl(dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_code_rate_t code_rate, dvbt_guard_interval_t guard_interval,
                             dvbt_transmission_mode_t transmission_mode, float snr, int bsize, int segment_superframes, bool soft_decision)
      : block("rx_hip", io_signature::make(1, 1, sizeof(gr_complex)), io_signature::make(1, 1, sizeof(unsigned char))), d_stream(0), d_finished(false)
    {
      dvbt_rx_stream_params p = {};
      p.rx.constellation = (int)constellation; p.rx.hierarchy = (int)hierarchy; p.rx.code_rate = (int)code_rate; p.rx.guard_interval = (int)guard_interval;
      p.rx.transmission_mode = (int)transmission_mode; p.rx.snr_db = snr; p.rx.viterbi_bsize = bsize; p.rx.descramble = 1;
      p.rx.soft_decision = soft_decision ? 1 : 0; p.segment_superframes = segment_superframes;
      if (dvbt_rx_stream_create(&p, &d_stream) < 0) throw std::runtime_error(std::string("libdvbt_hip: ") + dvbt_last_error());
      dvbt_dims d;
      if (dvbt_get_dims((int)constellation, (int)hierarchy, (int)code_rate, (int)guard_interval, (int)transmission_mode, &d) < 0) throw std::runtime_error(dvbt_last_error());
      /* TS bytes per input sample: info bits of a symbol x 188/204 / 8 over N + cp samples */
      set_relative_rate((double)d.info_bits_per_symbol * 188.0 / 204.0 / 8.0 / (double)(d.fft_length + d.cp_length));
      set_output_multiple(188);
    }

    rx_hip_impl::~rx_hip_impl() { if (d_stream) dvbt_rx_stream_destroy(d_stream); }

    /* Upstream has finished and its buffer is empty: the end of the stream.  gr::block_detail / gr::buffer_reader are public runtime API
     * (gnuradio/block_detail.h, gnuradio/buffer.h of the 3.7 series: buffer_reader::done(), items_available()). */
    bool rx_hip_impl::input_ended()
    {
      block_detail_sptr d = detail();
      return d && d->ninputs() > 0 && d->input(0)->done() && d->input(0)->items_available() == 0;
    }

    void rx_hip_impl::forecast(int noutput_items, gr_vector_int &ninput_items_required)
    {
      /* The block takes what it is given and delivers when a piece has been decoded: any amount of input lets it make progress.  Once the input has
       * ended it asks for NOTHING: the scheduler's executor (gnuradio-runtime/lib/block_executor.cc) marks a block done only when it is blocked on an
       * input whose upstream is done; a block that requires 0 items is never blocked on input, so general_work keeps being called (with 0 items) until
       * it returns WORK_DONE -- that is where the stream's tail is decoded and drained (below).  The reference's chain ends the same way, block by block,
       * with two 1504-byte items of energy_descramble held back (lib/energy_descramble_impl.cc:121-141); nothing more is held back here. */
      const int need = input_ended() ? 0 : 1;
      for (size_t i = 0; i < ninput_items_required.size(); i++) ninput_items_required[i] = need;
    }

    int rx_hip_impl::general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
    {
      dvbt_rx_stream_info inf;
      if (dvbt_rx_stream_status(d_stream, &inf) < 0) throw std::runtime_error(std::string("libdvbt_hip: ") + dvbt_last_error());
      const int nin = ninput_items[0];
      if (nin == 0 && input_ended() && !d_finished) {
        /* end of the stream: decode what is left (the one piece that has not been launched runs to the stream's end; blocks for that decode) */
        if (dvbt_rx_stream_finish(d_stream) < 0) throw std::runtime_error(std::string("libdvbt_hip: ") + dvbt_last_error());
        d_finished = true;
      } else if (nin > 0 && !d_finished && inf.ts_bytes_ready < RX_HIP_MAX_BACKLOG) {
        /* a sink slower than the source must not let the library's FIFO grow without bound: while more than RX_HIP_MAX_BACKLOG decoded bytes wait, the
         * input is left where it is (back-pressure reaches the source through the scheduler) and only output is handed out */
        if (dvbt_rx_stream_push(d_stream, input_items[0], (size_t)nin) < 0) throw std::runtime_error(std::string("libdvbt_hip: ") + dvbt_last_error());
        consume_each(nin);
      } else
        consume_each(0);
      const int64_t n = dvbt_rx_stream_pull(d_stream, output_items[0], (size_t)noutput_items);
      if (n < 0) throw std::runtime_error(std::string("libdvbt_hip: ") + dvbt_last_error());
      if (n == 0 && d_finished) return WORK_DONE;               /* finished and drained: the block is done, downstream sees the end */
      return (int)n;
    }

    /* A flowgraph stopped from outside (tb.stop(): a live source) never reaches the end-of-stream path above; what has been pushed is decoded here so that
     * dvbt_rx_stream_status reports the whole stream, but GNU Radio calls no work() after stop(): those bytes (at most one piece, segment_superframes + ~1.3
     * superframes) are not delivered, exactly as the reference's blocks drop what sits in their buffers when a flowgraph is stopped. */
    bool rx_hip_impl::stop()
    {
      if (d_stream && !d_finished) { dvbt_rx_stream_finish(d_stream); d_finished = true; }
      return true;
    }

  } /* namespace dvbt */
} /* namespace gr */
