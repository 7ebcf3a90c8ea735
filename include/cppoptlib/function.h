// cppoptlib/function.h — umbrella header and the four aliases (reference: include/cppoptlib/function.h:26-47).
#ifndef INCLUDE_CPPOPTLIB_FUNCTION_H_
#define INCLUDE_CPPOPTLIB_FUNCTION_H_

#include "function_base.h"
#include "function_expressions.h"
#include "function_penalty.h"
#include "function_problem.h"
#include "mi355/objectives.h"

namespace cppoptlib::function {
template <class F>
using FunctionXf = FunctionCRTP<F, float, DifferentiabilityMode::First>;
template <class F>
using FunctionXd = FunctionCRTP<F, double, DifferentiabilityMode::First>;

using FunctionExprXf = FunctionExpr<float, DifferentiabilityMode::First>;
using FunctionExprXd = FunctionExpr<double, DifferentiabilityMode::First>;
}  // namespace cppoptlib::function
#endif  // INCLUDE_CPPOPTLIB_FUNCTION_H_
