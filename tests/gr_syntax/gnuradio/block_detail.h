/* syntax-check scaffolding only (tests/gr_syntax/README.md): the two members of gr::block_detail the rx_hip shell calls, declared, not defined */
#ifndef GRSYN_BLOCK_DETAIL_H
#define GRSYN_BLOCK_DETAIL_H
#include <gnuradio/buffer.h>
namespace gr {
  class block_detail { public: int ninputs() const; buffer_reader_sptr input(unsigned int which); };
}
#endif
