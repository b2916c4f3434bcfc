#include "vrt_launch.h"
hipError_t vrt_launch_wavefront(const DevScene&, const DevCam&, const DevParams&, bool, bool, const WfWorkspace&, float4*, int*, unsigned long long*, hipStream_t) { return hipErrorNotSupported; }
