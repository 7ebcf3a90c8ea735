// tests/refprog/minigtest/gtest/gtest.h — TEST INFRASTRUCTURE, never part of a product path.
//
// GoogleTest is not in this image.  The reference's own unit-test files (src/test/*.cc) are built by
// tests/refprog/build_refprogs.py against the drop-in headers of include/ (and, for the golden record, against the
// reference's headers), and this header supplies the subset of the GoogleTest interface those files use: TEST,
// TYPED_TEST_CASE / TYPED_TEST over ::testing::Types<...>, ::testing::Test, EXPECT_* / ASSERT_* {EQ, NE, LT, LE, GT, GE,
// NEAR, TRUE, FALSE, DOUBLE_EQ} with `<< message`, ::testing::InitGoogleTest, RUN_ALL_TESTS.  Output follows GoogleTest's
// layout ([ RUN ], [ OK ], [ FAILED ], the final [ PASSED ] / [ FAILED ] counts) so that the harness can read it; the process
// exits non-zero when a test fails.  -DMINIGTEST_MAIN stands for linking GTest::Main (the reference's CMake does that for
// every test; files that define their own main() are built without it).
#ifndef TESTS_REFPROG_MINIGTEST_GTEST_GTEST_H_
#define TESTS_REFPROG_MINIGTEST_GTEST_GTEST_H_

#include <cmath>
#include <cstdio>
#include <functional>
#include <iostream>
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

template <class... Ts>
struct Types {};

namespace internal {

struct Registered {
  std::string name;
  std::function<Test*()> make;
};
inline std::vector<Registered>& Registry() {
  static std::vector<Registered> r;
  return r;
}
inline bool& CurrentFailed() {
  static bool failed = false;
  return failed;
}
inline int Register(const std::string& name, std::function<Test*()> make) {
  Registry().push_back({name, std::move(make)});
  return 0;
}

// One failed expectation: prints file:line, the expression text and whatever was streamed into it.
class Failure {
 public:
  Failure(const char* file, int line, const std::string& what, bool fatal) : fatal_(fatal) {
    text_ << file << ":" << line << ": Failure\n" << what;
  }
  template <class T>
  Failure& operator<<(const T& v) {
    extra_ << v;
    return *this;
  }
  Failure& operator<<(std::ostream& (*manip)(std::ostream&)) {
    extra_ << manip;
    return *this;
  }
  void Report() const {
    CurrentFailed() = true;
    std::cout << text_.str();
    const std::string e = extra_.str();
    if (!e.empty()) std::cout << "\n" << e;
    std::cout << std::endl;
  }
  bool fatal() const { return fatal_; }

 private:
  std::ostringstream text_, extra_;
  bool fatal_;
};
// `Reporter() = Failure(...) << a << b;` — assignment binds looser than <<, so the message is complete when it reports.
struct Reporter {
  void operator=(const Failure& f) const { f.Report(); }
};

template <class T>
std::string Show(const T& v) {
  if constexpr (std::is_same_v<T, bool>) {
    return v ? "true" : "false";
  } else if constexpr (std::is_enum_v<T>) {
    return std::to_string(static_cast<long long>(v));
  } else if constexpr (std::is_arithmetic_v<T>) {
    std::ostringstream os;
    os.precision(17);
    os << v;
    return os.str();
  } else {
    return "<value>";
  }
}
template <class A, class B>
std::string Compared(const char* op, const char* ea, const char* eb, const A& a, const B& b) {
  return std::string("Expected: (") + ea + ") " + op + " (" + eb + "), actual: " + Show(a) + " vs " + Show(b);
}
inline bool AlmostEqualDoubles(double a, double b) {   // EXPECT_DOUBLE_EQ: within 4 ULPs
  if (std::isnan(a) || std::isnan(b)) return false;
  if (a == b) return true;
  const double scale = std::fmax(std::fabs(a), std::fabs(b));
  return std::fabs(a - b) <= 4.0 * scale * 2.220446049250313e-16;
}

template <template <class> class Fixture, class List>
struct TypedRegistrar;
template <template <class> class Fixture, class... Ts>
struct TypedRegistrar<Fixture, Types<Ts...>> {
  static int Go(const char* suite, const char* name) {
    int index = 0;
    (void)std::initializer_list<int>{
        (Register(std::string(suite) + "/" + std::to_string(index++) + "." + name, [] { return static_cast<Test*>(new Fixture<Ts>()); }), 0)...};
    return 0;
  }
};

inline int RunAll() {
  int failed = 0;
  std::vector<std::string> failed_names;
  std::cout << "[==========] Running " << Registry().size() << " tests." << std::endl;
  for (auto& t : Registry()) {
    std::cout << "[ RUN      ] " << t.name << std::endl;
    CurrentFailed() = false;
    Test* test = t.make();
    test->SetUp();
    test->TestBody();
    test->TearDown();
    delete test;
    if (CurrentFailed()) {
      ++failed;
      failed_names.push_back(t.name);
      std::cout << "[  FAILED  ] " << t.name << std::endl;
    } else {
      std::cout << "[       OK ] " << t.name << std::endl;
    }
  }
  std::cout << "[==========] " << Registry().size() << " tests ran." << std::endl;
  std::cout << "[  PASSED  ] " << (Registry().size() - static_cast<size_t>(failed)) << " tests." << std::endl;
  if (failed) {
    std::cout << "[  FAILED  ] " << failed << " tests, listed below:" << std::endl;
    for (auto& n : failed_names) std::cout << "[  FAILED  ] " << n << std::endl;
  }
  return failed ? 1 : 0;
}

}  // namespace internal

inline void InitGoogleTest(int*, char**) {}
inline void InitGoogleTest() {}

}  // namespace testing

#define RUN_ALL_TESTS() ::testing::internal::RunAll()

#define MINIGTEST_CLASS_(suite, name) suite##_##name##_Test

#define TEST(suite, name)                                                                              \
  class MINIGTEST_CLASS_(suite, name) : public ::testing::Test {                                       \
   public:                                                                                             \
    void TestBody() override;                                                                          \
  };                                                                                                   \
  static int minigtest_reg_##suite##_##name = ::testing::internal::Register(                           \
      #suite "." #name, [] { return static_cast<::testing::Test*>(new MINIGTEST_CLASS_(suite, name)()); }); \
  void MINIGTEST_CLASS_(suite, name)::TestBody()

#define TEST_F(fixture, name)                                                                          \
  class MINIGTEST_CLASS_(fixture, name) : public fixture {                                             \
   public:                                                                                             \
    void TestBody() override;                                                                          \
  };                                                                                                   \
  static int minigtest_reg_##fixture##_##name = ::testing::internal::Register(                         \
      #fixture "." #name, [] { return static_cast<::testing::Test*>(new MINIGTEST_CLASS_(fixture, name)()); }); \
  void MINIGTEST_CLASS_(fixture, name)::TestBody()

#define TYPED_TEST_CASE(suite, types) using minigtest_types_##suite = types
#define TYPED_TEST_SUITE(suite, types) TYPED_TEST_CASE(suite, types)

#define TYPED_TEST(suite, name)                                                                        \
  template <class minigtest_T>                                                                         \
  class MINIGTEST_CLASS_(suite, name) : public suite<minigtest_T> {                                    \
   public:                                                                                             \
    using TypeParam = minigtest_T;                                                                     \
    void TestBody() override;                                                                          \
  };                                                                                                   \
  static int minigtest_reg_##suite##_##name =                                                          \
      ::testing::internal::TypedRegistrar<MINIGTEST_CLASS_(suite, name), minigtest_types_##suite>::Go(#suite, #name); \
  template <class minigtest_T>                                                                         \
  void MINIGTEST_CLASS_(suite, name)<minigtest_T>::TestBody()

// An expectation is `if (holds) ; else [return] Reporter() = Failure(...) << message`: the dangling-else form GoogleTest
// uses, so that `EXPECT_X(...) << "text";` parses and a failed ASSERT_X leaves the test body.
#define MINIGTEST_CHECK_(cond, what, fatal_return)       \
  if (cond)                                              \
    ;                                                    \
  else                                                   \
    fatal_return ::testing::internal::Reporter() = ::testing::internal::Failure(__FILE__, __LINE__, what, false)

#define MINIGTEST_CMP_(op, a, b, fatal_return)                                                    \
  MINIGTEST_CHECK_(((a)op(b)), ::testing::internal::Compared(#op, #a, #b, (a), (b)), fatal_return)

#define EXPECT_TRUE(c) MINIGTEST_CHECK_(static_cast<bool>(c), std::string("Value of: " #c "\n  Actual: false\nExpected: true"), )
#define EXPECT_FALSE(c) MINIGTEST_CHECK_(!static_cast<bool>(c), std::string("Value of: " #c "\n  Actual: true\nExpected: false"), )
#define ASSERT_TRUE(c) MINIGTEST_CHECK_(static_cast<bool>(c), std::string("Value of: " #c "\n  Actual: false\nExpected: true"), return)
#define ASSERT_FALSE(c) MINIGTEST_CHECK_(!static_cast<bool>(c), std::string("Value of: " #c "\n  Actual: true\nExpected: false"), return)

#define EXPECT_EQ(a, b) MINIGTEST_CMP_(==, a, b, )
#define EXPECT_NE(a, b) MINIGTEST_CMP_(!=, a, b, )
#define EXPECT_LT(a, b) MINIGTEST_CMP_(<, a, b, )
#define EXPECT_LE(a, b) MINIGTEST_CMP_(<=, a, b, )
#define EXPECT_GT(a, b) MINIGTEST_CMP_(>, a, b, )
#define EXPECT_GE(a, b) MINIGTEST_CMP_(>=, a, b, )
#define ASSERT_EQ(a, b) MINIGTEST_CMP_(==, a, b, return)
#define ASSERT_NE(a, b) MINIGTEST_CMP_(!=, a, b, return)
#define ASSERT_LT(a, b) MINIGTEST_CMP_(<, a, b, return)
#define ASSERT_LE(a, b) MINIGTEST_CMP_(<=, a, b, return)
#define ASSERT_GT(a, b) MINIGTEST_CMP_(>, a, b, return)
#define ASSERT_GE(a, b) MINIGTEST_CMP_(>=, a, b, return)

#define MINIGTEST_NEAR_(a, b, tol, fatal_return)                                                                  \
  MINIGTEST_CHECK_(std::fabs(static_cast<double>(a) - static_cast<double>(b)) <= static_cast<double>(tol),        \
                   std::string("The difference between " #a " and " #b " is ") +                                  \
                       ::testing::internal::Show(std::fabs(static_cast<double>(a) - static_cast<double>(b))) +    \
                       ", which exceeds " #tol " (" + ::testing::internal::Show(static_cast<double>(a)) + " vs " + \
                       ::testing::internal::Show(static_cast<double>(b)) + ")",                                   \
                   fatal_return)
#define EXPECT_NEAR(a, b, tol) MINIGTEST_NEAR_(a, b, tol, )
#define ASSERT_NEAR(a, b, tol) MINIGTEST_NEAR_(a, b, tol, return)
#define EXPECT_DOUBLE_EQ(a, b) MINIGTEST_CHECK_(::testing::internal::AlmostEqualDoubles((a), (b)), ::testing::internal::Compared("==", #a, #b, (a), (b)), )
#define ASSERT_DOUBLE_EQ(a, b) MINIGTEST_CHECK_(::testing::internal::AlmostEqualDoubles((a), (b)), ::testing::internal::Compared("==", #a, #b, (a), (b)), return)

#ifdef MINIGTEST_MAIN
int main(int argc, char** argv) {
  ::testing::InitGoogleTest(&argc, argv);
  return RUN_ALL_TESTS();
}
#endif

#endif  // TESTS_REFPROG_MINIGTEST_GTEST_GTEST_H_
