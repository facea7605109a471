/* Minimal FUNCTIONAL stand-in for <pmt/pmt.h> (see ../README.md): the polymorphic types the shim blocks build --
 * interned symbols, integers, doubles, booleans, dictionaries, complex vectors -- with GNU Radio 3.10's function names
 * and signatures, so that a test harness can read back what a block put into its tags and messages.  Not GNU Radio:
 * no serialisation, no other types. */
#ifndef MOCK_PMT_PMT_H
#define MOCK_PMT_PMT_H
#include <complex>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
namespace pmt {
class pmt_base
{
public:
    enum kind_t { NIL, SYMBOL, LONG, UINT64, DOUBLE, BOOL, DICT, C32VECTOR };
    kind_t kind = NIL;
    std::string sym;
    long l = 0;
    uint64_t u = 0;
    double d = 0.0;
    bool b = false;
    std::vector<std::pair<std::shared_ptr<pmt_base>, std::shared_ptr<pmt_base>>> items;   /* DICT, insertion order */
    std::vector<std::complex<float>> c32;
};
typedef std::shared_ptr<pmt_base> pmt_t;
class wrong_type : public std::invalid_argument
{
public:
    explicit wrong_type(const std::string& what) : std::invalid_argument("pmt::wrong_type: " + what) {}
};

inline pmt_t get_PMT_NIL() { static pmt_t nil = std::make_shared<pmt_base>(); return nil; }
#define PMT_NIL pmt::get_PMT_NIL()
inline pmt_t get_PMT_T() { static pmt_t t = [] { auto p = std::make_shared<pmt_base>(); p->kind = pmt_base::BOOL; p->b = true; return p; }(); return t; }
inline pmt_t get_PMT_F() { static pmt_t f = [] { auto p = std::make_shared<pmt_base>(); p->kind = pmt_base::BOOL; p->b = false; return p; }(); return f; }
#define PMT_T pmt::get_PMT_T()
#define PMT_F pmt::get_PMT_F()

/* symbols are interned: two symbols of one name are the same object, so tag keys compare with == as in GNU Radio */
inline pmt_t string_to_symbol(const std::string& s)
{
    static std::mutex mu;
    static std::map<std::string, pmt_t> table;
    std::lock_guard<std::mutex> lk(mu);
    auto it = table.find(s);
    if (it != table.end()) return it->second;
    auto p = std::make_shared<pmt_base>();
    p->kind = pmt_base::SYMBOL;
    p->sym = s;
    table.emplace(s, p);
    return p;
}
inline pmt_t intern(const std::string& s) { return string_to_symbol(s); }
inline pmt_t mp(const std::string& s) { return string_to_symbol(s); }
inline pmt_t mp(const char* s) { return string_to_symbol(s); }
inline bool is_symbol(const pmt_t& p) { return p && p->kind == pmt_base::SYMBOL; }
inline std::string symbol_to_string(const pmt_t& p)
{
    if (!is_symbol(p)) throw wrong_type("symbol_to_string");
    return p->sym;
}
inline pmt_t from_long(long v) { auto p = std::make_shared<pmt_base>(); p->kind = pmt_base::LONG; p->l = v; return p; }
inline pmt_t from_uint64(uint64_t v) { auto p = std::make_shared<pmt_base>(); p->kind = pmt_base::UINT64; p->u = v; return p; }
inline pmt_t from_double(double v) { auto p = std::make_shared<pmt_base>(); p->kind = pmt_base::DOUBLE; p->d = v; return p; }
inline pmt_t from_bool(bool v) { return v ? get_PMT_T() : get_PMT_F(); }
inline bool is_integer(const pmt_t& p) { return p && p->kind == pmt_base::LONG; }
inline bool is_real(const pmt_t& p) { return p && p->kind == pmt_base::DOUBLE; }
inline bool is_bool(const pmt_t& p) { return p && p->kind == pmt_base::BOOL; }
inline long to_long(const pmt_t& p)
{
    if (!is_integer(p)) throw wrong_type("to_long");
    return p->l;
}
inline uint64_t to_uint64(const pmt_t& p)
{
    if (!p || p->kind != pmt_base::UINT64) throw wrong_type("to_uint64");
    return p->u;
}
inline double to_double(const pmt_t& p)
{
    if (is_integer(p)) return double(p->l);
    if (!is_real(p)) throw wrong_type("to_double");
    return p->d;
}
inline bool to_bool(const pmt_t& p)
{
    if (!is_bool(p)) throw wrong_type("to_bool");
    return p->b;
}
inline bool eqv(const pmt_t& a, const pmt_t& b)
{
    if (a == b) return true;
    if (!a || !b || a->kind != b->kind) return false;
    switch (a->kind) {
    case pmt_base::LONG: return a->l == b->l;
    case pmt_base::UINT64: return a->u == b->u;
    case pmt_base::DOUBLE: return a->d == b->d;
    case pmt_base::BOOL: return a->b == b->b;
    default: return false;     /* symbols are interned: equal symbols were caught by a == b */
    }
}
inline pmt_t make_dict() { auto p = std::make_shared<pmt_base>(); p->kind = pmt_base::DICT; return p; }
inline bool is_dict(const pmt_t& p) { return p && p->kind == pmt_base::DICT; }
/* functional update, as in GNU Radio: returns a new dictionary */
inline pmt_t dict_add(const pmt_t& dict, const pmt_t& key, const pmt_t& value)
{
    if (!is_dict(dict)) throw wrong_type("dict_add");
    auto p = std::make_shared<pmt_base>(*dict);
    for (auto& kv : p->items)
        if (eqv(kv.first, key)) { kv.second = value; return p; }
    p->items.emplace_back(key, value);
    return p;
}
inline bool dict_has_key(const pmt_t& dict, const pmt_t& key)
{
    if (!is_dict(dict)) throw wrong_type("dict_has_key");
    for (auto& kv : dict->items) if (eqv(kv.first, key)) return true;
    return false;
}
inline pmt_t dict_ref(const pmt_t& dict, const pmt_t& key, const pmt_t& not_found)
{
    if (!is_dict(dict)) throw wrong_type("dict_ref");
    for (auto& kv : dict->items) if (eqv(kv.first, key)) return kv.second;
    return not_found;
}
inline size_t length(const pmt_t& p)
{
    if (is_dict(p)) return p->items.size();
    if (p && p->kind == pmt_base::C32VECTOR) return p->c32.size();
    throw wrong_type("length");
}
inline pmt_t init_c32vector(size_t k, const std::complex<float>* data)
{
    auto p = std::make_shared<pmt_base>();
    p->kind = pmt_base::C32VECTOR;
    p->c32.assign(data, data + k);
    return p;
}
inline pmt_t init_c32vector(size_t k, const std::vector<std::complex<float>>& data) { return init_c32vector(k, data.data()); }
inline bool is_c32vector(const pmt_t& p) { return p && p->kind == pmt_base::C32VECTOR; }
inline std::vector<std::complex<float>> c32vector_elements(const pmt_t& p)
{
    if (!is_c32vector(p)) throw wrong_type("c32vector_elements");
    return p->c32;
}
} // namespace pmt
#endif
