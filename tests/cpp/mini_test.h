// Minimal test macros (GoogleTest is not available in this image).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>

static int g_failures = 0;
#define EXPECT_TRUE(c)                                                        \
  do {                                                                        \
    if (!(c)) {                                                               \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c);               \
      std::fflush(stdout);                                                    \
      ++g_failures;                                                           \
    }                                                                         \
  } while (0)
#define EXPECT_EQ(a, b) EXPECT_TRUE((a) == (b))
#define EXPECT_NEAR(a, b, tol) EXPECT_TRUE(std::fabs(double(a) - double(b)) <= (tol))
#define TEST_MAIN_END()                                                       \
  do {                                                                        \
    if (g_failures) {                                                         \
      std::printf("%d FAILURE(S)\n", g_failures);                            \
      std::fflush(stdout);                                                    \
      return 1;                                                               \
    }                                                                         \
    std::printf("ALL PASSED\n");                                             \
    std::fflush(stdout);                                                      \
    return 0;                                                                 \
  } while (0)
