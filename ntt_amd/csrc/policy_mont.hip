// Kernels of the Mont<false> arithmetic policy (arith.hpp), compiled on their own.
#include "launch_impl.hpp"

namespace nttx {
NTTX_INSTANTIATE_POLICY(Mont<false>)
}
