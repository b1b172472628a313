// TEST INFRASTRUCTURE ONLY (oracle/_ref build).  Not part of the product.
//
// No-op stand-in for spdlog so the unmodified reference translation units
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

namespace spdlog {

class spdlog_ex : public std::exception {
    std::string m_;
public:
    explicit spdlog_ex(const std::string& m) : m_(m) {}
    const char* what() const noexcept override { return m_.c_str(); }
};

class logger {
public:
    unsigned long n_records = 0;
    template <typename... A> void info(const A&...) { n_records++; }
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
