/* hip_shell.h -- what the ten GNU Radio block shells of this directory share.
 *
 * A shell IS the reference's block class: it derives from the public interface gr-dvbt declares in include/dvbt/<block>.h
 * (same name, same make() signature, same io signatures and scheduler hints), so apps/*.grc, grc/*.xml and swig/dvbt_swig.i of
 * gr-dvbt keep working unchanged.  Its body is a pass-through to the C ABI of libdvbt_hip.so (include/dvbt_hip.h):
 *   forecast()      -> dvbt_<block>_forecast
 *   general_work()  -> stream tags of the input window into a dvbt_sideband, dvbt_<block>_work, the sideband's output tags
 *                      back into stream tags, consume_each(n_consumed), return the items produced
 * Tag keys: SURVEY Appendix D ("sync_start", "superframe_start", "symbol_index", values pmt longs).
 *
 * GNU Radio 3.7 is absent from the authoring container: this directory is built only where find_package(Gnuradio) succeeds
 * (CMakeLists.txt); tests/test_gr_shells.py checks the sources against the reference's public headers with g++ -fsyntax-only and
 * throw-away declarations of the few GNU Radio names used (tests/gr_syntax/, declarations only -- nothing is executed or linked).
 */
#ifndef INCLUDED_DVBT_HIP_SHELL_H
#define INCLUDED_DVBT_HIP_SHELL_H

#include <gnuradio/block.h>
#include <gnuradio/io_signature.h>
#include <pmt/pmt.h>
#include <stdexcept>
#include <string>
#include <vector>
#define DVBT_HIP_NO_ENUMS      /* gr-dvbt's dvbt/dvbt_config.h owns the enum type names in a shell */
#include <dvbt_hip.h>

namespace gr {
  namespace dvbt {
    namespace hip {

      inline pmt::pmt_t tag_symbol(int key)
      {
        static const pmt::pmt_t k_sync = pmt::string_to_symbol("sync_start"), k_sf = pmt::string_to_symbol("superframe_start"),
                                k_si = pmt::string_to_symbol("symbol_index");
        return key == DVBT_TAG_SYNC_START ? k_sync : key == DVBT_TAG_SUPERFRAME_START ? k_sf : k_si;
      }

      /* Owner of one C-ABI handle + the tag marshalling.  H: opaque handle type, P: its params struct. */
      template <class H, class P>
      class core
      {
      public:
        typedef int (*create_fn)(const P *, H **);
        typedef int (*forecast_fn)(const H *, int, int *);
        typedef int (*work_fn)(H *, int, int, const void *, void *, dvbt_sideband *);
        typedef void (*destroy_fn)(H *);

        core(const P &p, create_fn c, forecast_fn f, work_fn w, destroy_fn d) : d_h(0), d_forecast(f), d_work(w), d_destroy(d), d_out(4096)
        {
          if (c(&p, &d_h) < 0)                       /* no CPU fallback: a box without a usable GPU cannot run the flowgraph */
            throw std::runtime_error(std::string("libdvbt_hip: ") + dvbt_last_error());
        }
        ~core() { if (d_h) d_destroy(d_h); }

        void forecast(int noutput_items, gr_vector_int &required) const
        {
          int n = noutput_items;
          d_forecast(d_h, noutput_items, &n);
          for (size_t i = 0; i < required.size(); i++) required[i] = n;
        }

        /* one general_work call of block b; returns the items produced on output 0 */
        int work(gr::block *b, int noutput_items, int ninput_items, const void *in, void *out)
        {
          const uint64_t r0 = b->nitems_read(0);
          std::vector<tag_t> tags;
          b->get_tags_in_range(tags, 0, r0, r0 + (uint64_t)ninput_items);
          d_in.clear();
          for (size_t i = 0; i < tags.size(); i++) {
            dvbt_tag t; t.rel_offset = (int64_t)(tags[i].offset - r0); t.value = (int32_t)pmt::to_long(tags[i].value);
            if (pmt::eqv(tags[i].key, tag_symbol(DVBT_TAG_SYNC_START))) t.key = DVBT_TAG_SYNC_START;
            else if (pmt::eqv(tags[i].key, tag_symbol(DVBT_TAG_SUPERFRAME_START))) t.key = DVBT_TAG_SUPERFRAME_START;
            else if (pmt::eqv(tags[i].key, tag_symbol(DVBT_TAG_SYMBOL_INDEX))) t.key = DVBT_TAG_SYMBOL_INDEX;
            else continue;
            d_in.push_back(t);
          }
          dvbt_sideband sb;
          sb.in_tags = d_in.empty() ? 0 : &d_in[0]; sb.n_in_tags = (int)d_in.size();
          sb.out_tags = &d_out[0]; sb.out_cap = (int)d_out.size(); sb.n_out_tags = 0; sb.n_consumed = 0;
          const int produced = d_work(d_h, noutput_items, ninput_items, in, out, &sb);
          if (produced < 0) throw std::runtime_error(std::string("libdvbt_hip: ") + dvbt_last_error());
          const uint64_t w0 = b->nitems_written(0);
          for (int i = 0; i < sb.n_out_tags && i < sb.out_cap; i++)
            b->add_item_tag(0, w0 + (uint64_t)d_out[i].rel_offset, tag_symbol(d_out[i].key), pmt::from_long(d_out[i].value));
          b->consume_each(sb.n_consumed);
          return produced;
        }

        H *handle() const { return d_h; }

      private:
        core(const core &); core &operator=(const core &);
        H *d_h; forecast_fn d_forecast; work_fn d_work; destroy_fn d_destroy;
        std::vector<dvbt_tag> d_in, d_out;
      };

    } // namespace hip
  } // namespace dvbt
} // namespace gr

/* body of a shell class: members + the two virtuals, given the C-ABI stem (e.g. viterbi_decoder) */
#define DVBT_HIP_SHELL_MEMBERS(stem) \
  private: \
    hip::core< ::dvbt_##stem, ::dvbt_##stem##_params> d_core;   /* global names: gr::dvbt::dvbt_demap is the block class */ \
  public: \
    void forecast(int noutput_items, gr_vector_int &ninput_items_required) { d_core.forecast(noutput_items, ninput_items_required); } \
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) \
    { return d_core.work(this, noutput_items, ninput_items[0], input_items[0], output_items[0]); }

#define DVBT_HIP_CORE_INIT(stem, params) \
  d_core(params, ::dvbt_##stem##_create, ::dvbt_##stem##_forecast, ::dvbt_##stem##_work, ::dvbt_##stem##_destroy)

#endif /* INCLUDED_DVBT_HIP_SHELL_H */
