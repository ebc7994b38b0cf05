/* reed_solomon_dec_impl.cc -- gr::dvbt::reed_solomon_dec on libdvbt_hip (replaces lib/reed_solomon_dec_impl.cc and the decoder of
 * lib/reed_solomon.cc).  Decoding failures are silent, as in the reference (:100-102): the bytes pass through.  The decoder is the
 * correct one; oracle_compat = 1 in the params would reproduce the as-compiled reference quirk (SURVEY B-1). */
#include "reed_solomon_dec_impl.h"

namespace gr {
  namespace dvbt {

    reed_solomon_dec::sptr
    reed_solomon_dec::make(int p, int m, int gfpoly, int n, int k, int t, int s, int blocks)
    { return gnuradio::get_initial_sptr(new reed_solomon_dec_impl(p, m, gfpoly, n, k, t, s, blocks)); }

    static dvbt_reed_solomon_dec_params rs_params(int p, int m, int gfpoly, int n, int k, int t, int s, int blocks)
    { dvbt_reed_solomon_dec_params q = { p, m, gfpoly, n, k, t, s, blocks, 0 }; return q; }

    /* io signatures: lib/reed_solomon_dec_impl.cc:48-50 */
    reed_solomon_dec_impl::reed_solomon_dec_impl(int p, int m, int gfpoly, int n, int k, int t, int s, int blocks)
      : block("reed_solomon_dec", io_signature::make(1, 1, sizeof(unsigned char) * blocks * (n - s)), io_signature::make(1, 1, sizeof(unsigned char) * blocks * (k - s))),
        DVBT_HIP_CORE_INIT(reed_solomon_dec, rs_params(p, m, gfpoly, n, k, t, s, blocks))
    {
    }

  } /* namespace dvbt */
} /* namespace gr */
