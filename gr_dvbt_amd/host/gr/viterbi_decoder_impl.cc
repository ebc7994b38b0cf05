/* viterbi_decoder_impl.cc -- gr::dvbt::viterbi_decoder on libdvbt_hip (replaces lib/viterbi_decoder_impl.cc, lib/d_viterbi.c,
 * lib/d_tab.c).  Unlike the reference (file-scope decoder state, lib/viterbi_decoder_impl.cc:46-52) any number of instances may
 * live in one process.  Tags: superframe_start in (reset + skip to the tag, :213-229), re-emitted with value 1 (:298-307). */
#include "viterbi_decoder_impl.h"

namespace gr {
  namespace dvbt {

    viterbi_decoder::sptr
    viterbi_decoder::make(dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_code_rate_t coderate, int bsize, int S0, int SK)
    { return gnuradio::get_initial_sptr(new viterbi_decoder_impl(constellation, hierarchy, coderate, bsize, S0, SK)); }

    static dvbt_viterbi_decoder_params vit_params(int c, int h, int r, int bsize, int S0, int SK)
    { dvbt_viterbi_decoder_params p = { c, h, r, bsize, S0, SK }; return p; }

    /* io signatures: lib/viterbi_decoder_impl.cc:79-81; output multiple :141; the relative rate of :138 is an integer division that
     * yields 0 there (SURVEY B-17): the intended k m / (8 n) is set here */
    viterbi_decoder_impl::viterbi_decoder_impl(dvbt_constellation_t constellation, dvbt_hierarchy_t hierarchy, dvbt_code_rate_t coderate,
                                               int bsize, int S0, int SK)
      : block("viterbi_decoder", io_signature::make(1, 1, sizeof(unsigned char)), io_signature::make(1, 1, sizeof(unsigned char))),
        DVBT_HIP_CORE_INIT(viterbi_decoder, vit_params((int)constellation, (int)hierarchy, (int)coderate, bsize, S0, SK))
    {
      dvbt_dims d;
      if (dvbt_get_dims((int)constellation, (int)hierarchy, (int)coderate, 0, 0, &d) < 0) throw std::runtime_error(dvbt_last_error());
      set_relative_rate((double)(d.cr_k * d.m) / (double)(8 * d.cr_n));
      set_output_multiple(bsize * d.cr_k / 8);
      /* the library emits superframe_start itself (behind the ntraceback delay and the dropped partial block); with the intended rate GNU Radio's
       * default all-to-all propagation would add a second, misplaced copy of the upstream tag (the reference's rate of 0 sends it to offset 0) */
      set_tag_propagation_policy(TPP_DONT);
    }

  } /* namespace dvbt */
} /* namespace gr */
