/* syntax-check scaffolding only (tests/gr_syntax/README.md): the two members of gr::buffer_reader the rx_hip shell calls, declared, not defined */
#ifndef GRSYN_BUFFER_H
#define GRSYN_BUFFER_H
#include <gnuradio/io_signature.h>
namespace gr {
  class buffer_reader { public: int items_available() const; bool done() const; };
  typedef boost::shared_ptr<buffer_reader> buffer_reader_sptr;
  class block_detail;
  typedef boost::shared_ptr<block_detail> block_detail_sptr;
}
#endif
