// cppoptlib/mi355/context.h — RAII owner of an engine context (mi355_lbfgs_ctx).
// Solver objects share it through a shared_ptr, so they stay copy-constructible
// like the reference's solvers (AugmentedLagrangian clones its inner solver,
// solver/augmented_lagrangian.h:347).
#ifndef CPPOPTLIB_MI355_CONTEXT_H_
#define CPPOPTLIB_MI355_CONTEXT_H_

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>

#include "../../mi355_lbfgs.h"

namespace cppoptlib::mi355 {

// Engine failures surface like the reference's only error path
// (function_base.h:108-115): an exception, or abort() when built -fno-exceptions.
[[noreturn]] inline void Fail(const std::string& what) {
#if defined(__cpp_exceptions) || defined(__EXCEPTIONS) || defined(_CPPUNWIND)
  throw std::runtime_error(what);
#else
  std::fprintf(stderr, "cppoptlib::mi355: %s\n", what.c_str());
  std::abort();
#endif
}
inline void Check(int rc, const char* where) {
  if (rc != MI355_OK) Fail(std::string(where) + ": " + mi355_lbfgs_last_error());
}

class Context {
 public:
  explicit Context(int device = 0) { Check(mi355_lbfgs_create(device, &ctx_), "mi355_lbfgs_create"); }
  ~Context() { mi355_lbfgs_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  mi355_lbfgs_ctx* get() const { return ctx_; }

  // One lazily created context per process for `device` (default device 0).
  static std::shared_ptr<Context> Default(int device = 0) {
    static std::mutex mu;
    static std::shared_ptr<Context> slots[16];
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0 || device >= 16) Fail("device index out of range");
    if (!slots[device]) slots[device] = std::make_shared<Context>(device);
    return slots[device];
  }

 private:
  mi355_lbfgs_ctx* ctx_ = nullptr;
};

}  // namespace cppoptlib::mi355
#endif  // CPPOPTLIB_MI355_CONTEXT_H_
