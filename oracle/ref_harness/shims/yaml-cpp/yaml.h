// TEST INFRASTRUCTURE ONLY (oracle/_ref build).  Not part of the product.
//
// Minimal in-memory stand-in for yaml-cpp's YAML::Node, sufficient for the
// key/value reads the reference performs through `Config`
// (reference include/utilities/config.h:7-12 and every `c[...]...as<T>()`
// call site).  yaml-cpp carries no hot-path arithmetic; this
// shim only lets the *unmodified* reference translation units compile in a
// container that has no yaml-cpp and no network.
//
// Supported YAML subset (enough for config/example.yaml): block maps by
// indentation, flow sequences `[a, b, c]`, plain / quoted scalars, `#`
// comments.
#ifndef ORACLE_SHIM_YAML_H
#define ORACLE_SHIM_YAML_H

#include <cstdlib>
#include <fstream>
#include <list>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace YAML {

class Exception : public std::runtime_error {
public:
    explicit Exception(const std::string& m) : std::runtime_error(m) {}
};

namespace detail {
struct Impl {
    bool defined = false;
    bool is_scalar = false;
    std::string scalar;
    std::map<std::string, std::shared_ptr<Impl>> map;
    std::vector<std::shared_ptr<Impl>> seq;
};

template <typename T> struct conv;

template <> struct conv<std::string> {
    static std::string get(const Impl& n) { return n.scalar; }
};
template <> struct conv<bool> {
    static bool get(const Impl& n) {
        const std::string& s = n.scalar;
        if (s == "true" || s == "True" || s == "yes" || s == "on" || s == "1") return true;
        if (s == "false" || s == "False" || s == "no" || s == "off" || s == "0") return false;
        throw Exception("bad bool: " + s);
    }
};
#define SHIM_NUM_CONV(T, EXPR)                                   \
    template <> struct conv<T> {                                 \
        static T get(const Impl& n) {                            \
            const std::string& s = n.scalar;                     \
            if (s.empty()) throw Exception("empty scalar");      \
            return (T)(EXPR);                                    \
        }                                                        \
    };
SHIM_NUM_CONV(int, std::stol(s))
SHIM_NUM_CONV(long, std::stol(s))
SHIM_NUM_CONV(unsigned, std::stoul(s))
SHIM_NUM_CONV(unsigned long, std::stoull(s))
SHIM_NUM_CONV(float, std::stof(s))
SHIM_NUM_CONV(double, std::stod(s))
#undef SHIM_NUM_CONV
}  // namespace detail

class Node {
    std::shared_ptr<detail::Impl> p_;

public:
    Node() : p_(std::make_shared<detail::Impl>()) {}
    explicit Node(std::shared_ptr<detail::Impl> p) : p_(std::move(p)) {}

    bool IsDefined() const { return p_->defined; }
    explicit operator bool() const { return p_->defined; }
    bool operator!() const { return !p_->defined; }

    // Map access.  Like yaml-cpp, a missing key yields an undefined node that
    // becomes defined on assignment.
    Node operator[](const std::string& key) {
        auto it = p_->map.find(key);
        if (it == p_->map.end()) {
            auto child = std::make_shared<detail::Impl>();
            p_->map[key] = child;
            return Node(child);
        }
        return Node(it->second);
    }
    Node operator[](const char* key) { return (*this)[std::string(key)]; }
    Node operator[](int idx) {
        if (idx < 0 || (size_t)idx >= p_->seq.size())
            return Node(std::make_shared<detail::Impl>());
        return Node(p_->seq[idx]);
    }

    template <typename T> T as() const {
        if (!p_->defined) throw Exception("bad conversion of undefined node");
        return get_(static_cast<T*>(nullptr));
    }
    template <typename T, typename D> T as(const D& dflt) const {
        if (!p_->defined) return (T)dflt;
        try {
            return get_(static_cast<T*>(nullptr));
        } catch (...) {
            return (T)dflt;
        }
    }

    template <typename T> Node& operator=(const T& v) {
        std::ostringstream os;
        os.precision(17);
        os << v;
        p_->defined = true;
        p_->is_scalar = true;
        p_->scalar = os.str();
        return *this;
    }
    Node& operator=(const std::vector<std::string>& v) {
        p_->defined = true;
        p_->is_scalar = false;
        p_->seq.clear();
        for (auto& s : v) {
            auto c = std::make_shared<detail::Impl>();
            c->defined = true;
            c->is_scalar = true;
            c->scalar = s;
            p_->seq.push_back(c);
        }
        return *this;
    }
    Node& operator=(const Node& o) = default;
    Node(const Node&) = default;

    size_t size() const { return p_->is_scalar ? 0 : (p_->seq.size() + p_->map.size()); }
    detail::Impl& impl() { return *p_; }

private:
    template <typename T> T get_(T*) const { return detail::conv<T>::get(*p_); }
    template <typename E> std::list<E> get_(std::list<E>*) const {
        std::list<E> out;
        for (auto& c : p_->seq) out.push_back(detail::conv<E>::get(*c));
        return out;
    }
    template <typename E> std::vector<E> get_(std::vector<E>*) const {
        std::vector<E> out;
        for (auto& c : p_->seq) out.push_back(detail::conv<E>::get(*c));
        return out;
    }
};

namespace detail {
inline std::string strip(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}
inline std::string unquote(const std::string& s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
        return s.substr(1, s.size() - 2);
    return s;
}
inline std::string strip_comment(const std::string& line) {
    bool in_s = false, in_d = false;
    for (size_t i = 0; i < line.size(); i++) {
        char ch = line[i];
        if (ch == '"' && !in_s) in_d = !in_d;
        else if (ch == '\'' && !in_d) in_s = !in_s;
        else if (ch == '#' && !in_s && !in_d && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t'))
            return line.substr(0, i);
    }
    return line;
}
inline void set_value(Impl& n, const std::string& raw) {
    std::string v = strip(raw);
    n.defined = true;
    if (!v.empty() && v.front() == '[' && v.back() == ']') {
        n.is_scalar = false;
        std::string body = v.substr(1, v.size() - 2);
        std::string cur;
        bool in_s = false, in_d = false;
        auto flush = [&]() {
            std::string e = strip(cur);
            if (!e.empty()) {
                auto c = std::make_shared<Impl>();
                c->defined = true;
                c->is_scalar = true;
                c->scalar = unquote(e);
                n.seq.push_back(c);
            }
            cur.clear();
        };
        for (char ch : body) {
            if (ch == '"' && !in_s) in_d = !in_d;
            if (ch == '\'' && !in_d) in_s = !in_s;
            if (ch == ',' && !in_s && !in_d) flush();
            else cur.push_back(ch);
        }
        flush();
    } else {
        n.is_scalar = true;
        n.scalar = unquote(v);
    }
}
}  // namespace detail

inline Node Load(std::istream& in) {
    using detail::Impl;
    auto root = std::make_shared<Impl>();
    root->defined = true;
    std::vector<std::pair<int, std::shared_ptr<Impl>>> stack;  // (indent, map node)
    stack.push_back({-1, root});
    std::string line;
    while (std::getline(in, line)) {
        line = detail::strip_comment(line);
        if (detail::strip(line).empty()) continue;
        int indent = 0;
        while ((size_t)indent < line.size() && line[indent] == ' ') indent++;
        std::string body = detail::strip(line);
        size_t colon = body.find(':');
        if (colon == std::string::npos) throw Exception("yaml shim: unsupported line: " + line);
        std::string key = detail::unquote(detail::strip(body.substr(0, colon)));
        std::string val = detail::strip(body.substr(colon + 1));
        while (stack.size() > 1 && stack.back().first >= indent) stack.pop_back();
        auto child = std::make_shared<Impl>();
        stack.back().second->map[key] = child;
        child->defined = true;
        if (val.empty()) {
            child->is_scalar = false;
            stack.push_back({indent, child});
        } else {
            detail::set_value(*child, val);
        }
    }
    return Node(root);
}

inline Node Load(const std::string& text) {
    std::istringstream is(text);
    return Load(is);
}

inline Node LoadFile(const std::string& path) {
    std::ifstream f(path);
    if (!f.is_open()) throw Exception("yaml shim: cannot open " + path);
    return Load(f);
}

}  // namespace YAML

#endif
