/* syntax-check scaffolding only (tests/gr_syntax/README.md): the members of gr::block the shells call, declared, not defined */
#ifndef GRSYN_BLOCK_H
#define GRSYN_BLOCK_H
#include <complex>
#include <vector>
#include <string>
#include <stdint.h>
#include <gnuradio/io_signature.h>
#include <pmt/pmt.h>
#include <gnuradio/buffer.h>
typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;
namespace gr {
  struct tag_t { uint64_t offset; pmt::pmt_t key, value, srcid; };
  class block {
  public:
    enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
    enum tag_propagation_policy_t { TPP_DONT = 0, TPP_ALL_TO_ALL = 1, TPP_ONE_TO_ONE = 2 };
    void set_tag_propagation_policy(tag_propagation_policy_t p);
    block(const std::string &name, io_signature::sptr in, io_signature::sptr out);
    virtual ~block();
    virtual void forecast(int noutput_items, gr_vector_int &ninput_items_required);
    virtual int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items);
    virtual bool stop();
    void consume_each(int how_many_items);
    block_detail_sptr detail() const;
    void set_output_multiple(int multiple);
    void set_relative_rate(double relative_rate);
    uint64_t nitems_read(unsigned int which_input);
    uint64_t nitems_written(unsigned int which_output);
    void add_item_tag(unsigned int which_output, uint64_t abs_offset, const pmt::pmt_t &key, const pmt::pmt_t &value);
    void get_tags_in_range(std::vector<tag_t> &v, unsigned int which_input, uint64_t abs_start, uint64_t abs_end);
  protected:
    block();
  };
}
namespace gnuradio { template <class T> boost::shared_ptr<T> get_initial_sptr(T *p) { return boost::shared_ptr<T>(p); } }
#endif
