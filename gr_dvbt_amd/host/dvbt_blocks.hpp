// dvbt_blocks.hpp -- C++ host-side mirror of the gr::dvbt block interface over the C ABI.
//
// GNU Radio is not available where this repository is built, so these classes do not derive
// from gr::block; they keep the reference's names, make() argument lists and the
// forecast()/general_work() contract (include/dvbt/*.h) so that the real block
// shells are a few lines each: general_work() gathers the visible tags into a
// dvbt_sideband, calls work(), re-emits the returned tags and calls consume_each(n_consumed).
//
// Header-only; link with libdvbt_hip.so.  All computation happens on the GPU behind the C ABI.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/dvbt_hip.h"

namespace gr { namespace dvbt_amd {

typedef dvbt_constellation_t dvbt_constellation_t;
typedef dvbt_hierarchy_t dvbt_hierarchy_t;
typedef dvbt_code_rate_t dvbt_code_rate_t;
typedef dvbt_guard_interval_t dvbt_guard_interval_t;
typedef dvbt_transmission_mode_t dvbt_transmission_mode_t;

struct tag_t { long long offset; int key; int value; };

inline void check(int r) { if (r < 0) throw std::runtime_error(std::string("libdvbt_hip: ") + dvbt_last_error()); }

// CRTP-free helper: every block is (handle, forecast fn, work fn, destroy fn)
template <class H, class P> class block_base {
 public:
  typedef std::shared_ptr<block_base> sptr;
  typedef int (*create_fn)(const P *, H **);
  typedef int (*forecast_fn)(const H *, int, int *);
  typedef int (*work_fn)(H *, int, int, const void *, void *, dvbt_sideband *);
  typedef int (*work_device_fn)(H *, int, int, const void *, void *, dvbt_sideband *, void *);
  typedef void (*destroy_fn)(H *);
  block_base(const P &p, create_fn c, forecast_fn f, work_fn w, destroy_fn d, work_device_fn wd = nullptr) : d_forecast(f), d_work(w), d_destroy(d), d_work_device(wd)
  { check(c(&p, &d_h)); }
  ~block_base() { if (d_h) d_destroy(d_h); }
  block_base(const block_base &) = delete;
  // gr::block::forecast
  void forecast(int noutput_items, std::vector<int> &ninput_items_required)
  { int n = 0; check(d_forecast(d_h, noutput_items, &n)); for (auto &x : ninput_items_required) x = n; }
  // gr::block::general_work for one input and one output stream.  tags_in: tags visible in the input
  // window (offsets relative to its first item); tags_out: tags to attach (relative to the first output
  // item); n_consumed: argument for consume_each().  Returns the number of items produced.
  int general_work(int noutput_items, int ninput_items, const void *in, void *out, const std::vector<tag_t> &tags_in,
                   std::vector<tag_t> &tags_out, int &n_consumed)
  {
    std::vector<dvbt_tag> ti(tags_in.size()), to(4096);
    for (size_t i = 0; i < tags_in.size(); i++) { ti[i].rel_offset = tags_in[i].offset; ti[i].key = tags_in[i].key; ti[i].value = tags_in[i].value; }
    dvbt_sideband sb; sb.in_tags = ti.data(); sb.n_in_tags = (int)ti.size(); sb.out_tags = to.data(); sb.out_cap = (int)to.size();
    sb.n_out_tags = 0; sb.n_consumed = 0;
    int r = d_work(d_h, noutput_items, ninput_items, in, out, &sb);
    check(r);
    tags_out.clear();
    for (int i = 0; i < sb.n_out_tags && i < sb.out_cap; i++) tags_out.push_back(tag_t{to[i].rel_offset, to[i].key, to[i].value});
    n_consumed = sb.n_consumed;
    return r;
  }
  // the same call on DEVICE buffers and a HIP stream (dvbt_<blk>_work_device): adjacent HIP blocks hand items over in HBM
  int general_work_device(int noutput_items, int ninput_items, const void *in_device, void *out_device, const std::vector<tag_t> &tags_in,
                          std::vector<tag_t> &tags_out, int &n_consumed, void *stream = nullptr)
  {
    if (!d_work_device) throw std::runtime_error("block has no device entry");
    std::vector<dvbt_tag> ti(tags_in.size()), to(4096);
    for (size_t i = 0; i < tags_in.size(); i++) { ti[i].rel_offset = tags_in[i].offset; ti[i].key = tags_in[i].key; ti[i].value = tags_in[i].value; }
    dvbt_sideband sb; sb.in_tags = ti.data(); sb.n_in_tags = (int)ti.size(); sb.out_tags = to.data(); sb.out_cap = (int)to.size();
    sb.n_out_tags = 0; sb.n_consumed = 0;
    int r = d_work_device(d_h, noutput_items, ninput_items, in_device, out_device, &sb, stream);
    check(r);
    tags_out.clear();
    for (int i = 0; i < sb.n_out_tags && i < sb.out_cap; i++) tags_out.push_back(tag_t{to[i].rel_offset, to[i].key, to[i].value});
    n_consumed = sb.n_consumed;
    return r;
  }
 private:
  H *d_h = nullptr; forecast_fn d_forecast; work_fn d_work; destroy_fn d_destroy; work_device_fn d_work_device;
};

#define DVBT_AMD_BLOCK(NAME, PARAMS, MAKE_ARGS, PARAM_INIT)                                                     \
  class NAME : public block_base<::dvbt_##NAME, PARAMS> {                                                       \
   public:                                                                                                      \
    typedef std::shared_ptr<NAME> sptr;                                                                         \
    static sptr make MAKE_ARGS { PARAMS prm__ = PARAM_INIT; return sptr(new NAME(prm__)); }                             \
   private:                                                                                                     \
    explicit NAME(const PARAMS &p)                                                                              \
        : block_base<::dvbt_##NAME, PARAMS>(p, dvbt_##NAME##_create, dvbt_##NAME##_forecast, dvbt_##NAME##_work, dvbt_##NAME##_destroy, dvbt_##NAME##_work_device) {} \
  };

// include/dvbt/ofdm_sym_acquisition.h:49
DVBT_AMD_BLOCK(ofdm_sym_acquisition, dvbt_ofdm_sym_acquisition_params, (int blocks, int fft_length, int occupied_tones, int cp_length, float snr),
               (dvbt_ofdm_sym_acquisition_params{blocks, fft_length, occupied_tones, cp_length, snr}))
// gr::fft::fft_vcc(fft_size, forward, window, shift) as used by apps/dvbt_rx_demo*.grc
DVBT_AMD_BLOCK(fft, dvbt_fft_params, (int fft_size, bool forward, bool shift), (dvbt_fft_params{fft_size, forward ? 1 : 0, shift ? 1 : 0}))
// include/dvbt/demod_reference_signals.h:50-54
DVBT_AMD_BLOCK(demod_reference_signals, dvbt_demod_reference_signals_params,
               (int itemsize, int ninput, int noutput, dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_code_rate_t code_rate_HP,
                dvbt_code_rate_t code_rate_LP, dvbt_guard_interval_t guard_interval, dvbt_transmission_mode_t transmission_mode,
                int include_cell_id, int cell_id),
               (dvbt_demod_reference_signals_params{itemsize, ninput, noutput, constellation, hierarchy, code_rate_HP, code_rate_LP, guard_interval,
                                                    transmission_mode, include_cell_id, cell_id}))
// include/dvbt/dvbt_demap.h:50
DVBT_AMD_BLOCK(demap, dvbt_demap_params, (int nsize, dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_transmission_mode_t transmission, float gain),
               (dvbt_demap_params{nsize, constellation, hierarchy, transmission, gain}))
typedef demap dvbt_demap;   // reference class name
// include/dvbt/symbol_inner_interleaver.h:50-51
DVBT_AMD_BLOCK(symbol_inner_interleaver, dvbt_symbol_inner_interleaver_params, (int ninput, dvbt_transmission_mode_t transmission, int direction),
               (dvbt_symbol_inner_interleaver_params{ninput, transmission, direction}))
// include/dvbt/bit_inner_deinterleaver.h:50-51
DVBT_AMD_BLOCK(bit_inner_deinterleaver, dvbt_bit_inner_deinterleaver_params,
               (int nsize, dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_transmission_mode_t transmission),
               (dvbt_bit_inner_deinterleaver_params{nsize, constellation, hierarchy, transmission}))
// include/dvbt/viterbi_decoder.h:51-52
DVBT_AMD_BLOCK(viterbi_decoder, dvbt_viterbi_decoder_params,
               (dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_code_rate_t coderate, int bsize, int S0, int SK),
               (dvbt_viterbi_decoder_params{constellation, hierarchy, coderate, bsize, S0, SK}))
// include/dvbt/convolutional_deinterleaver.h:49
DVBT_AMD_BLOCK(convolutional_deinterleaver, dvbt_convolutional_deinterleaver_params, (int nsize, int I, int M),
               (dvbt_convolutional_deinterleaver_params{nsize, I, M}))
// include/dvbt/reed_solomon_dec.h:49
DVBT_AMD_BLOCK(reed_solomon_dec, dvbt_reed_solomon_dec_params, (int p, int m, int gfpoly, int n, int k, int t, int s, int blocks),
               (dvbt_reed_solomon_dec_params{p, m, gfpoly, n, k, t, s, blocks, 0}))
// include/dvbt/energy_descramble.h
DVBT_AMD_BLOCK(energy_descramble, dvbt_energy_descramble_params, (int nblocks), (dvbt_energy_descramble_params{nblocks}))
// stock gr::filter::rational_resampler_ccc(interpolation, decimation, taps=None) followed by blocks::multiply_const_cc(scale)
// (rational_resampler_xxx_0 + blocks_multiply_const_vxx_0 of apps/dvbt_rx_demo*.grc), one block here
DVBT_AMD_BLOCK(resampler, dvbt_resampler_params, (int interpolation, int decimation, float scale), (dvbt_resampler_params{interpolation, decimation, scale}))

#undef DVBT_AMD_BLOCK

// gr::dvbt::rx_hip (gr_dvbt_amd/host/gr/include/dvbt/rx_hip.h): the ten receive blocks of apps/dvbt_rx_demo*.grc behind one block, over the
// streaming entry of the C ABI (dvbt_rx_stream_*).  The same forecast / general_work logic as gr_dvbt_amd/host/gr/rx_hip_impl.cc, with the one thing the
// shell asks the GNU Radio runtime (detail()->input(0)->done() && items_available() == 0) supplied by the host loop through input_ended():
//   forecast: 1 item while the input lives, 0 once it has ended (so that the scheduler keeps calling general_work with nothing to read);
//   general_work: pushes the input unless more than MAX_BACKLOG decoded bytes wait for the sink, hands out the TS bytes that are ready; at the end of the
//   input it decodes the stream's tail (dvbt_rx_stream_finish), keeps delivering, and returns WORK_DONE when everything is out.
class rx_hip {
 public:
  typedef std::shared_ptr<rx_hip> sptr;
  enum { WORK_DONE = -1, MAX_BACKLOG = 8 << 20 };
  static sptr make(dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_code_rate_t code_rate, dvbt_guard_interval_t guard_interval,
                   dvbt_transmission_mode_t transmission_mode, float snr = 30.0f, int bsize = 768, int segment_superframes = 0, bool soft_decision = false)
  { return sptr(new rx_hip(constellation, hierarchy, code_rate, guard_interval, transmission_mode, snr, bsize, segment_superframes, soft_decision)); }
  ~rx_hip() { if (d_s) dvbt_rx_stream_destroy(d_s); }
  rx_hip(const rx_hip &) = delete;
  void input_ended(bool ended) { d_input_ended = ended; }     // the runtime's "upstream done and its buffer empty"
  void forecast(int, std::vector<int> &ninput_items_required) { for (auto &x : ninput_items_required) x = d_input_ended ? 0 : 1; }
  // in: ninput_items complex64 samples; out: room for noutput_items bytes.  n_consumed: all of the input, or 0 under back-pressure.
  // Returns the TS bytes produced, or WORK_DONE
  int general_work(int noutput_items, int ninput_items, const void *in, void *out, int &n_consumed)
  {
    n_consumed = 0;
    const dvbt_rx_stream_info inf = info();
    if (ninput_items == 0 && d_input_ended && !d_finished) { check(dvbt_rx_stream_finish(d_s)); d_finished = true; }
    else if (ninput_items > 0 && !d_finished && inf.ts_bytes_ready < MAX_BACKLOG) { check(dvbt_rx_stream_push(d_s, in, (size_t)ninput_items)); n_consumed = ninput_items; }
    const long long n = dvbt_rx_stream_pull(d_s, out, (size_t)noutput_items);
    check((int)(n < 0 ? n : 0));
    if (n == 0 && d_finished) return WORK_DONE;
    return (int)n;
  }
  // a flowgraph stopped from outside: what was pushed is decoded, nothing more is delivered (no work() call follows stop())
  bool stop() { if (!d_finished) { check(dvbt_rx_stream_finish(d_s)); d_finished = true; } return true; }
  dvbt_rx_stream_info info() const { dvbt_rx_stream_info i; check(dvbt_rx_stream_status(d_s, &i)); return i; }
 private:
  rx_hip(dvbt_constellation_t c, dvbt_hierarchy_t h, dvbt_code_rate_t r, dvbt_guard_interval_t g, dvbt_transmission_mode_t m, float snr, int bsize, int seg, bool soft)
  {
    dvbt_rx_stream_params p{};
    p.rx.constellation = (int)c; p.rx.hierarchy = (int)h; p.rx.code_rate = (int)r; p.rx.guard_interval = (int)g; p.rx.transmission_mode = (int)m;
    p.rx.snr_db = snr; p.rx.viterbi_bsize = bsize; p.rx.descramble = 1; p.rx.soft_decision = soft ? 1 : 0; p.segment_superframes = seg;
    check(dvbt_rx_stream_create(&p, &d_s));
  }
  dvbt_rx_stream *d_s = nullptr; bool d_input_ended = false, d_finished = false;
};

}}  // namespace gr::dvbt_amd
