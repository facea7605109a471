/* syntax-check stand-in for <pmt/pmt.h> (see ../README.md): names and signatures only */
#ifndef MOCK_PMT_PMT_H
#define MOCK_PMT_PMT_H
#include <memory>
#include <string>
namespace pmt {
class pmt_base {};
typedef std::shared_ptr<pmt_base> pmt_t;
inline pmt_t mp(const std::string&) { return pmt_t(); }
inline pmt_t mp(const char*) { return pmt_t(); }
inline pmt_t string_to_symbol(const std::string&) { return pmt_t(); }
inline pmt_t intern(const std::string&) { return pmt_t(); }
inline pmt_t make_dict() { return pmt_t(); }
inline pmt_t dict_add(const pmt_t&, const pmt_t&, const pmt_t&) { return pmt_t(); }
inline pmt_t from_long(long) { return pmt_t(); }
inline pmt_t from_uint64(unsigned long long) { return pmt_t(); }
inline pmt_t from_double(double) { return pmt_t(); }
inline pmt_t from_bool(bool) { return pmt_t(); }
} // namespace pmt
#endif
