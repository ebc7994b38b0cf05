/* dvbt/rx_hip.h -- public interface of gr::dvbt::rx_hip: the WHOLE receive chain of apps/dvbt_rx_demo*.grc behind one block
 *   ofdm_sym_acquisition -> fft_vxx -> demod_reference_signals -> dvbt_demap -> symbol_inner_interleaver(0) -> bit_inner_deinterleaver ->
 *   vector_to_stream -> viterbi_decoder -> convolutional_deinterleaver -> reed_solomon_dec -> energy_descramble
 * complex samples at the OFDM elementary rate in (what ofdm_sym_acquisition takes), MPEG-TS bytes out (what the file sink takes).  The body is
 * the streaming entry of libdvbt_hip (dvbt_rx_stream_*, include/dvbt_hip.h): the bytes are those of the ten blocks it stands for, at the
 * segment API's speed whatever the scheduler's buffer sizes.  gr-dvbt itself has no such block; the arguments are those of its make() calls
 * that describe the transmission (include/dvbt/demod_reference_signals.h:50-54, ofdm_sym_acquisition.h:49, viterbi_decoder.h:51-52). */
#ifndef INCLUDED_DVBT_RX_HIP_H
#define INCLUDED_DVBT_RX_HIP_H

#include <dvbt/api.h>
#include <dvbt/dvbt_config.h>
#include <gnuradio/block.h>

namespace gr {
  namespace dvbt {

    class DVBT_API rx_hip : virtual public block
    {
    public:
      typedef boost::shared_ptr<rx_hip> sptr;
      /* snr: ofdm_sym_acquisition's parameter (30 in the demo flowgraphs); bsize: viterbi_decoder's (768);
       * segment_superframes: superframes decoded per launch sequence (0 = 16).  The TS lags the input by about segment_superframes + 1 superframes (a piece is decoded
       * when the next one is known to be viable): 4 (the GRC default) is ~1.4 s of signal at 8k, 0.35 s at 2k, and still > 200x real time; 16 is for files.  The end of a
       * finite stream is delivered whatever the value (rx_hip_impl.cc);
       * soft_decision: soft demapper + soft-input Viterbi instead of the reference's hard decisions (2-3 dB less SNR needed, ~1.6x the time) */
      static sptr make(dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_code_rate_t code_rate,
                       dvbt_guard_interval_t guard_interval, dvbt_transmission_mode_t transmission_mode,
                       float snr = 30.0f, int bsize = 768, int segment_superframes = 0, bool soft_decision = false);
    };

  } // namespace dvbt
} // namespace gr

#endif
