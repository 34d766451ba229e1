// Fused split + LOD pyramid + apron kernels (placeholder until the fused path lands; the generic
// reference-shaped kernels in bt_kernels.hip are the complete path).
#include "bt_internal.hpp"

namespace bt {

bool fused_plan(bt_preprocessor*, bt_atlas*, std::vector<TaskDev>&, std::vector<Launch>&) { return false; }

bt_status fused_launch(bt_preprocessor*, bt_atlas*, const Launch&) {
    set_error("fused launch requested but no fused plan exists");
    return BT_ERR_UNSUPPORTED;
}

}  // namespace bt
