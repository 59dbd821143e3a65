// Factory of the device-side setup accelerator (see host/spops.hpp: SetupAccel).
#pragma once
#include <hip/hip_runtime.h>
#include <memory>
#include "../host/common.hpp"
#include "../host/spops.hpp"

namespace airhip {
std::unique_ptr<SetupAccel> make_device_setup_accel(hipStream_t stream, int64_t min_nnz = -1);
}
