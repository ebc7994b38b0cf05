/* syntax-check scaffolding only (tests/gr_syntax/README.md) */
#ifndef GRSYN_IO_SIGNATURE_H
#define GRSYN_IO_SIGNATURE_H
#include <memory>
namespace boost { using std::shared_ptr; }
namespace gr {
  class io_signature { public: typedef boost::shared_ptr<io_signature> sptr; static sptr make(int min_streams, int max_streams, int sizeof_stream_item); };
}
#endif
