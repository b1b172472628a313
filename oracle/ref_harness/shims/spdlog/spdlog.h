// TEST INFRASTRUCTURE ONLY (oracle/_ref build).  Not part of the product.
//
// Stand-in for spdlog (nothing is written) so the unmodified reference translation units
// compile without the (absent, un-pinned) third-party logger.  spdlog is used
// by the reference for CSV logging only; it carries no
// hot-path arithmetic.
#ifndef ORACLE_SHIM_SPDLOG_H
#define ORACLE_SHIM_SPDLOG_H

#include <climits>
#include <exception>
#include <map>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

namespace spdlog {

class spdlog_ex : public std::exception {
    std::string m_;
public:
    explicit spdlog_ex(const std::string& m) : m_(m) {}
    const char* what() const noexcept override { return m_.c_str(); }
};

// Records are not formatted or written anywhere; the numeric arguments of each call are kept so that the harness
// can read back what the reference logged (profit_log rows of Intraday::LogProfit, intraday.cpp:438-451).
class logger {
    static void put(std::vector<double>&) {}
    template <typename T, typename... R>
    static typename std::enable_if<std::is_arithmetic<T>::value>::type put(std::vector<double>& row, const T& v, const R&... rest) {
        row.push_back((double)v);
        put(row, rest...);
    }
    template <typename T, typename... R>
    static typename std::enable_if<!std::is_arithmetic<T>::value>::type put(std::vector<double>& row, const T&, const R&... rest) {
        put(row, rest...);
    }
public:
    unsigned long n_records = 0;
    std::vector<std::vector<double>> rows;
    template <typename... A> void info(const A&... a) {
        n_records++;
        std::vector<double> row;
        put(row, a...);
        rows.push_back(row);
    }
};

inline std::map<std::string, std::shared_ptr<logger>>& registry() {
    static std::map<std::string, std::shared_ptr<logger>> r;
    return r;
}

inline std::shared_ptr<logger> get(const std::string& name) {
    auto it = registry().find(name);
    return it == registry().end() ? nullptr : it->second;
}

inline std::shared_ptr<logger> rotating_logger_mt(const std::string& name, const std::string&,
                                                   size_t, size_t) {
    if (registry().count(name)) throw spdlog_ex("logger exists: " + name);
    auto l = std::make_shared<logger>();
    registry()[name] = l;
    return l;
}

inline void set_pattern(const std::string&) {}
inline void drop_all() { registry().clear(); }

}  // namespace spdlog

#endif
