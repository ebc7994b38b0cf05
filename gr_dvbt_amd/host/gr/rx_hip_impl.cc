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
      : block("rx_hip", io_signature::make(1, 1, sizeof(gr_complex)), io_signature::make(1, 1, sizeof(unsigned char))), d_stream(0)
    {
      dvbt_rx_stream_params p;
      p.rx.constellation = (int)constellation; p.rx.hierarchy = (int)hierarchy; p.rx.code_rate = (int)code_rate; p.rx.guard_interval = (int)guard_interval;
      p.rx.transmission_mode = (int)transmission_mode; p.rx.include_cell_id = 0; p.rx.cell_id = 0; p.rx.snr_db = snr; p.rx.viterbi_bsize = bsize;
      p.rx.rs_oracle_compat = 0; p.rx.descramble = 1; p.rx.max_samples = 0; p.rx.device = 0; p.rx.viterbi_chunk_bytes = 0;
      p.rx.resample_interp = 0; p.rx.resample_decim = 0; p.rx.front_scale = 0.f; p.rx.soft_decision = soft_decision ? 1 : 0;
      p.segment_superframes = segment_superframes; p.rank = 0; p.world = 0;
      if (dvbt_rx_stream_create(&p, &d_stream) < 0) throw std::runtime_error(std::string("libdvbt_hip: ") + dvbt_last_error());
      dvbt_dims d;
      if (dvbt_get_dims((int)constellation, (int)hierarchy, (int)code_rate, (int)guard_interval, (int)transmission_mode, &d) < 0) throw std::runtime_error(dvbt_last_error());
      /* TS bytes per input sample: info bits of a symbol x 188/204 / 8 over N + cp samples */
      set_relative_rate((double)d.info_bits_per_symbol * 188.0 / 204.0 / 8.0 / (double)(d.fft_length + d.cp_length));
      set_output_multiple(188);
    }

    rx_hip_impl::~rx_hip_impl() { if (d_stream) dvbt_rx_stream_destroy(d_stream); }

    void rx_hip_impl::forecast(int noutput_items, gr_vector_int &ninput_items_required)
    {
      /* the block takes what it is given and delivers when a piece has been decoded: any amount of input lets it make progress */
      for (size_t i = 0; i < ninput_items_required.size(); i++) ninput_items_required[i] = 1;
    }

    int rx_hip_impl::general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
    {
      if (dvbt_rx_stream_push(d_stream, input_items[0], (size_t)ninput_items[0]) < 0) throw std::runtime_error(std::string("libdvbt_hip: ") + dvbt_last_error());
      consume_each(ninput_items[0]);
      const int64_t n = dvbt_rx_stream_pull(d_stream, output_items[0], (size_t)noutput_items);
      if (n < 0) throw std::runtime_error(std::string("libdvbt_hip: ") + dvbt_last_error());
      return (int)n;
    }

    /* end of the flowgraph's run: what is left of the stream is decoded (energy_descramble's two-item hold-back applies here, as at the end of the
     * reference's run); the bytes still inside can be fetched with dvbt_rx_stream_pull by a host that wants them (GNU Radio calls no work() after stop()) */
    bool rx_hip_impl::stop()
    {
      if (d_stream) dvbt_rx_stream_finish(d_stream);
      return true;
    }

  } /* namespace dvbt */
} /* namespace gr */
