// tests/refprog/prelude.h — TEST INFRASTRUCTURE, force-included (`g++ -include`) into BOTH builds of a reference program
// (over the reference's headers and over include/), never part of a product path.
//
// The reference's programs print with the stream's default six significant digits, and its PrintProgressCallback
// switches the stream to fixed / six decimals for everything that follows (solver/solver.h:62-66): too coarse to compare
// two builds at 1e-6.  This prelude makes every number both builds print carry 17 digits — the stream default here, and
// every later `std::setprecision(p)` turned into `std::setprecision(17)` — so that tests/test_reference_programs.py
// compares what was computed, not what was rounded for display.  Layout and wording of the output are untouched.
#ifndef TESTS_REFPROG_PRELUDE_H_
#define TESTS_REFPROG_PRELUDE_H_
#include <iomanip>
#include <iostream>
#include <sstream>
namespace refprog_prelude {
struct FullPrecision {
  FullPrecision() { std::cout << std::setprecision(17); }
};
static FullPrecision g_full_precision;
}  // namespace refprog_prelude
#define setprecision(p) setprecision(17)
#endif  // TESTS_REFPROG_PRELUDE_H_
