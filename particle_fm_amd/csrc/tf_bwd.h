// Backward kernels of the Full-Transformer vector field (gfx950).
#pragma once
#include "tf_common.h"

namespace pfm {
namespace tf {
inline int attn_bwd_set_lds() { return 0; }
}  // namespace tf
}  // namespace pfm
