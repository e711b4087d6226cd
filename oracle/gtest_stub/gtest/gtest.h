// oracle/gtest_stub/gtest/gtest.h -- TEST INFRASTRUCTURE ONLY.
// A ~150-line stand-in for the GoogleTest surface the reference's tests/test_all.cpp uses (gtest 1.14 is a network fetch in the
// reference's CMakeLists.txt:140-146 and is not in this image): TEST / TEST_F, EXPECT_* / ASSERT_* with streamed messages,
// GTEST_SKIP, SUCCEED, ::testing::Test fixtures, InitGoogleTest / RUN_ALL_TESTS.  It lets the reference's OWN unit tests run
// against the real reference sources compiled on the axiom stand-in (oracle/Makefile target `reftests`), which is how the
// stand-in itself is checked (tests/test_reference_suite.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace testing {

class Test {
  public:
    virtual ~Test() = default;
    virtual void SetUp() {}
    virtual void TearDown() {}
    virtual void TestBody() = 0;
};

struct State {
    bool failed = false, skipped = false;
    std::string skip_msg;
};
inline State &cur() {
    static State s;
    return s;
}
struct Case {
    std::string name;
    std::function<Test *()> make;
};
inline std::vector<Case> &registry() {
    static std::vector<Case> r;
    return r;
}
struct Registrar {
    Registrar(const char *suite, const char *name, std::function<Test *()> f) { registry().push_back({std::string(suite) + "." + name, f}); }
};

template <class T, class = void> struct streamable : std::false_type {};
template <class T> struct streamable<T, std::void_t<decltype(std::declval<std::ostream &>() << std::declval<const T &>())>> : std::true_type {};
template <class T> std::string show(const T &v) {
    if constexpr (streamable<T>::value) {
        std::ostringstream os;
        os << v;
        return os.str();
    } else {
        return "<" + std::to_string(sizeof(T)) + "-byte object>";
    }
}

class Message {
  public:
    template <class T> Message &operator<<(const T &v) {
        os_ << show(v);
        return *this;
    }
    std::string str() const { return os_.str(); }

  private:
    std::ostringstream os_;
};

enum Kind { kFail, kFatal, kSkip };
struct Helper {
    Kind kind;
    const char *file;
    int line;
    std::string what;
    void operator=(const Message &m) const {
        if (kind == kSkip) {
            cur().skipped = true;
            cur().skip_msg = m.str();
            return;
        }
        cur().failed = true;
        std::printf("%s:%d: Failure\n%s\n%s\n", file, line, what.c_str(), m.str().c_str());
    }
};

inline bool float_eq(float a, float b) {  // within 4 ULPs, as EXPECT_FLOAT_EQ
    if (std::isnan(a) || std::isnan(b)) return false;
    int32_t ia, ib;
    std::memcpy(&ia, &a, 4);
    std::memcpy(&ib, &b, 4);
    if (ia < 0) ia = (int32_t)0x80000000 - ia;
    if (ib < 0) ib = (int32_t)0x80000000 - ib;
    return std::llabs((long long)ia - (long long)ib) <= 4;
}

inline void InitGoogleTest(int *, char **) {}
inline int run_all(const char *filter) {
    int passed = 0, failed = 0, skipped = 0;
    for (auto &c : registry()) {
        if (filter && *filter && c.name.find(filter) == std::string::npos) continue;
        cur() = State{};
        std::printf("[ RUN      ] %s\n", c.name.c_str());
        try {
            std::unique_ptr<Test> t(c.make());
            t->SetUp();
            if (!cur().skipped && !cur().failed) t->TestBody();
            t->TearDown();
        } catch (const std::exception &e) {
            cur().failed = true;
            std::printf("unexpected exception: %s\n", e.what());
        }
        if (cur().failed) {
            ++failed;
            std::printf("[  FAILED  ] %s\n", c.name.c_str());
        } else if (cur().skipped) {
            ++skipped;
            std::printf("[  SKIPPED ] %s (%s)\n", c.name.c_str(), cur().skip_msg.c_str());
        } else {
            ++passed;
            std::printf("[       OK ] %s\n", c.name.c_str());
        }
    }
    std::printf("[==========] %d passed, %d failed, %d skipped\n", passed, failed, skipped);
    return failed ? 1 : 0;
}

}  // namespace testing

#define RUN_ALL_TESTS() ::testing::run_all(std::getenv("GTEST_STUB_FILTER"))

#define GT_CAT_(a, b) a##_##b##_Test
#define GT_TEST_(suite, name, base)                                                                               \
    class GT_CAT_(suite, name) : public base {                                                                    \
      public:                                                                                                     \
        void TestBody() override;                                                                                 \
    };                                                                                                            \
    static ::testing::Registrar gt_reg_##suite##_##name(#suite, #name, [] { return new GT_CAT_(suite, name)(); }); \
    void GT_CAT_(suite, name)::TestBody()
#define TEST(suite, name) GT_TEST_(suite, name, ::testing::Test)
#define TEST_F(fixture, name) GT_TEST_(fixture, name, fixture)

#define GT_NONFATAL_(cond, text) \
    if (cond)                    \
        ;                        \
    else                         \
        ::testing::Helper{::testing::kFail, __FILE__, __LINE__, text} = ::testing::Message()
#define GT_FATAL_(cond, text) \
    if (cond)                 \
        ;                     \
    else                      \
        return ::testing::Helper{::testing::kFatal, __FILE__, __LINE__, text} = ::testing::Message()
#define GT_CMP_TEXT_(op, a, b) (std::string("Expected: (" #a ") " #op " (" #b "), actual: ") + ::testing::show(a) + " vs " + ::testing::show(b))

#define EXPECT_TRUE(c) GT_NONFATAL_((c), "Value of: " #c "\n  Actual: false\nExpected: true")
#define EXPECT_FALSE(c) GT_NONFATAL_(!(c), "Value of: " #c "\n  Actual: true\nExpected: false")
#define ASSERT_TRUE(c) GT_FATAL_((c), "Value of: " #c "\n  Actual: false\nExpected: true")
#define ASSERT_FALSE(c) GT_FATAL_(!(c), "Value of: " #c "\n  Actual: true\nExpected: false")
#define EXPECT_EQ(a, b) GT_NONFATAL_((a) == (b), GT_CMP_TEXT_(==, a, b))
#define EXPECT_NE(a, b) GT_NONFATAL_((a) != (b), GT_CMP_TEXT_(!=, a, b))
#define EXPECT_LT(a, b) GT_NONFATAL_((a) < (b), GT_CMP_TEXT_(<, a, b))
#define EXPECT_LE(a, b) GT_NONFATAL_((a) <= (b), GT_CMP_TEXT_(<=, a, b))
#define EXPECT_GT(a, b) GT_NONFATAL_((a) > (b), GT_CMP_TEXT_(>, a, b))
#define EXPECT_GE(a, b) GT_NONFATAL_((a) >= (b), GT_CMP_TEXT_(>=, a, b))
#define ASSERT_EQ(a, b) GT_FATAL_((a) == (b), GT_CMP_TEXT_(==, a, b))
#define ASSERT_NE(a, b) GT_FATAL_((a) != (b), GT_CMP_TEXT_(!=, a, b))
#define ASSERT_GT(a, b) GT_FATAL_((a) > (b), GT_CMP_TEXT_(>, a, b))
#define ASSERT_GE(a, b) GT_FATAL_((a) >= (b), GT_CMP_TEXT_(>=, a, b))
#define ASSERT_LT(a, b) GT_FATAL_((a) < (b), GT_CMP_TEXT_(<, a, b))
#define EXPECT_FLOAT_EQ(a, b) GT_NONFATAL_(::testing::float_eq((a), (b)), GT_CMP_TEXT_(~=, a, b))
#define ASSERT_FLOAT_EQ(a, b) GT_FATAL_(::testing::float_eq((a), (b)), GT_CMP_TEXT_(~=, a, b))
#define EXPECT_NEAR(a, b, tol) GT_NONFATAL_(std::fabs((double)(a) - (double)(b)) <= (double)(tol), GT_CMP_TEXT_(~, a, b))
#define ASSERT_NEAR(a, b, tol) GT_FATAL_(std::fabs((double)(a) - (double)(b)) <= (double)(tol), GT_CMP_TEXT_(~, a, b))
#define EXPECT_THROW(stmt, ex)                       \
    do {                                             \
        bool gt_thrown = false;                      \
        try {                                        \
            stmt;                                    \
        } catch (const ex &) { gt_thrown = true; }   \
        EXPECT_TRUE(gt_thrown) << #stmt " did not throw " #ex; \
    } while (0)
#define EXPECT_NO_THROW(stmt)                                   \
    do {                                                        \
        try {                                                   \
            stmt;                                               \
        } catch (...) { EXPECT_TRUE(false) << #stmt " threw"; } \
    } while (0)
#define SUCCEED() ::testing::Message()
#define FAIL() GT_FATAL_(false, "Failed")
#define GTEST_SKIP() return ::testing::Helper{::testing::kSkip, __FILE__, __LINE__, ""} = ::testing::Message()
