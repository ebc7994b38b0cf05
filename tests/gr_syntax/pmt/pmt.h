/* syntax-check scaffolding only (tests/gr_syntax/README.md) */
#ifndef GRSYN_PMT_H
#define GRSYN_PMT_H
#include <memory>
#include <string>
namespace pmt {
  class pmt_base; typedef std::shared_ptr<pmt_base> pmt_t;
  pmt_t string_to_symbol(const std::string &s); pmt_t from_long(long v); long to_long(pmt_t p); bool eqv(const pmt_t &a, const pmt_t &b);
}
#endif
