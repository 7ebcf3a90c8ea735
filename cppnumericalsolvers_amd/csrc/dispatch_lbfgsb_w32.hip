// dispatch_lbfgsb_w32.hip — the L-BFGS-B kernels with 32 lanes per problem (m = 9, 10; see engine_internal.hpp).
#define MI355_DISPATCH_TU 1
#define MI355_DISPATCH_LBFGSB_W32_TU 1
#include "engine_internal.hpp"
