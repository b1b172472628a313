// TEST INFRASTRUCTURE ONLY.  A minimal stand-in for the Catch single header, just enough to
// compile and run the reference's own unit tests (test/test_*.cpp of the reference checkout,
// compiled in place by oracle/Makefile `make reftests`) without the un-pinned, un-vendored Catch the
// reference's build system downloads.  Implements the part of Catch those files use:
//   SCENARIO / GIVEN / WHEN / THEN (nested sections, every leaf path run in a fresh execution of
//   the test function, like Catch does), REQUIRE, REQUIRE_THROWS, REQUIRE_NOTHROW, Approx
//   (epsilon = 100 * float epsilon, .scale(), .epsilon(), .margin()).
// With CATCH_CONFIG_MAIN it provides main(): runs everything, prints a summary, returns the
// number of failed assertions.
#ifndef MINI_CATCH_HPP
#define MINI_CATCH_HPP

#include <cmath>
#include <cstdio>
#include <exception>
#include <limits>
#include <string>
#include <vector>

namespace mini_catch {

struct TestCase { const char* name; void (*fn)(); };
inline std::vector<TestCase>& registry() { static std::vector<TestCase> r; return r; }
struct Registrar { Registrar(const char* n, void (*f)()) { TestCase t; t.name = n; t.fn = f; registry().push_back(t); } };

// Section tracking: a tree of sections discovered while running; each run of the test function
// enters, at every level, only the first child that is not completed yet.
struct Node {
    std::string name;
    std::vector<Node> kids;
    bool done;
    explicit Node(const std::string& n = std::string()) : name(n), done(false) {}
};
struct Run {
    Node root;
    std::vector<Node*> stack;       // sections currently open
    std::vector<bool> level_taken;  // per open level: has a child been entered during this run
    int passed, failed;
    std::string current;
    Run() : passed(0), failed(0) {}
};
inline Run& run() { static Run r; return r; }

inline Node* find_or_add(Node& parent, const std::string& name) {
    for (auto& k : parent.kids) if (k.name == name) return &k;
    parent.kids.push_back(Node(name));
    return &parent.kids.back();
}

struct Section {
    bool active;
    Section(const char* prefix, const char* name) : active(false) {
        Run& r = run();
        Node& parent = *r.stack.back();
        Node* n = find_or_add(parent, std::string(prefix) + name);
        // NB: find_or_add may reallocate parent.kids; pointers in `stack` refer to ancestors only.
        if (n->done || r.level_taken.back()) return;  // finished earlier, or a sibling runs in this pass
        r.level_taken.back() = true;
        r.stack.push_back(n);
        r.level_taken.push_back(false);
        active = true;
    }
    ~Section() {
        if (!active) return;
        Run& r = run();
        Node* n = r.stack.back();
        // a section is complete once all the children discovered inside it are complete
        bool all = true;
        for (auto& k : n->kids) all = all && k.done;
        if (all) n->done = true;
        r.stack.pop_back();
        r.level_taken.pop_back();
    }
    explicit operator bool() const { return active; }
};

class Approx {
public:
    explicit Approx(double v) : v_(v), eps_(std::numeric_limits<float>::epsilon() * 100.0), scale_(1.0), margin_(0.0) {}
    Approx& epsilon(double e) { eps_ = e; return *this; }
    Approx& scale(double s) { scale_ = s; return *this; }
    Approx& margin(double m) { margin_ = m; return *this; }
    bool matches(double x) const {
        // Catch 1.x: |x - v| < epsilon * (scale + max(|x|, |v|)); newer versions add the margin
        const double d = std::fabs(x - v_);
        return d < eps_ * (scale_ + std::fmax(std::fabs(x), std::fabs(v_))) || d <= margin_;
    }
    double value() const { return v_; }
private:
    double v_, eps_, scale_, margin_;
};
template <class T> bool operator==(const T& a, const Approx& b) { return b.matches((double)a); }
template <class T> bool operator==(const Approx& b, const T& a) { return b.matches((double)a); }
template <class T> bool operator!=(const T& a, const Approx& b) { return !b.matches((double)a); }

inline void report(bool ok, const char* expr, const char* file, int line) {
    Run& r = run();
    if (ok) { r.passed++; return; }
    r.failed++;
    std::string path;
    for (size_t i = 1; i < r.stack.size(); i++) path += " / " + r.stack[i]->name;
    std::printf("FAILED %s:%d  %s\n    in %s%s\n", file, line, expr, r.current.c_str(), path.c_str());
}

inline int run_all() {
    Run& r = run();
    int cases = 0;
    for (auto& tc : registry()) {
        cases++;
        r.root = Node(tc.name);
        r.current = tc.name;
        int guard = 0;
        do {
            r.stack.assign(1, &r.root);
            r.level_taken.assign(1, false);
            try {
                tc.fn();
            } catch (const std::exception& e) {
                r.failed++;
                std::printf("FAILED %s: unexpected exception: %s\n", tc.name, e.what());
                break;
            } catch (...) {
                r.failed++;
                std::printf("FAILED %s: unexpected exception\n", tc.name);
                break;
            }
            bool all = true;
            for (auto& k : r.root.kids) all = all && k.done;
            r.root.done = all;
        } while (!r.root.done && ++guard < 100000);
    }
    std::printf("mini-catch: %d test cases, %d assertions passed, %d failed\n", cases, r.passed, r.failed);
    return r.failed;
}

}  // namespace mini_catch

using mini_catch::Approx;

#define MC_CAT2(a, b) a##b
#define MC_CAT(a, b) MC_CAT2(a, b)
#define SCENARIO(name, ...)                                                              \
    static void MC_CAT(mc_test_, __LINE__)();                                            \
    static mini_catch::Registrar MC_CAT(mc_reg_, __LINE__)("Scenario: " name, &MC_CAT(mc_test_, __LINE__)); \
    static void MC_CAT(mc_test_, __LINE__)()
#define TEST_CASE(name, ...)                                                             \
    static void MC_CAT(mc_test_, __LINE__)();                                            \
    static mini_catch::Registrar MC_CAT(mc_reg_, __LINE__)(name, &MC_CAT(mc_test_, __LINE__)); \
    static void MC_CAT(mc_test_, __LINE__)()
#define MC_SECTION(prefix, name) if (mini_catch::Section MC_CAT(mc_sec_, __LINE__){prefix, name})
#define SECTION(name) MC_SECTION("", name)
#define GIVEN(name) MC_SECTION("Given: ", name)
#define WHEN(name) MC_SECTION("When: ", name)
#define AND_WHEN(name) MC_SECTION("And when: ", name)
#define THEN(name) MC_SECTION("Then: ", name)
#define AND_THEN(name) MC_SECTION("And: ", name)
#define REQUIRE(...) mini_catch::report(static_cast<bool>(__VA_ARGS__), #__VA_ARGS__, __FILE__, __LINE__)
#define CHECK(...) REQUIRE(__VA_ARGS__)
#define REQUIRE_FALSE(...) mini_catch::report(!static_cast<bool>(__VA_ARGS__), "!(" #__VA_ARGS__ ")", __FILE__, __LINE__)
#define REQUIRE_THROWS(...)                                                              \
    do {                                                                                 \
        bool mc_threw = false;                                                           \
        try { (void)(__VA_ARGS__); } catch (...) { mc_threw = true; }                    \
        mini_catch::report(mc_threw, "throws: " #__VA_ARGS__, __FILE__, __LINE__);       \
    } while (0)
#define REQUIRE_NOTHROW(...)                                                             \
    do {                                                                                 \
        bool mc_threw = false;                                                           \
        try { (void)(__VA_ARGS__); } catch (...) { mc_threw = true; }                    \
        mini_catch::report(!mc_threw, "does not throw: " #__VA_ARGS__, __FILE__, __LINE__); \
    } while (0)

#ifdef CATCH_CONFIG_MAIN
int main() { return mini_catch::run_all() ? 1 : 0; }
#endif

#endif  // MINI_CATCH_HPP
