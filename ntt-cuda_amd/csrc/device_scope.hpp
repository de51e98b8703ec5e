// device_scope.hpp -- run an entry point on the device its context was created on.
//
// Every launching entry point of the C ABI runs on the CONTEXT's device, whatever the caller's current device is: the
// scope switches to it when they differ and switches back on exit (a process that holds contexts for several GPUs, or
// torch code that has moved on to another device, would otherwise launch against another device's tables).  It also
// drops a stale error of an unrelated earlier HIP call, so that the hipGetLastError() behind each launch reports that
// launch only.
#pragma once
#include <hip/hip_runtime.h>

namespace mi355ntt {

struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = (err == hipSuccess);
        }
        if (err == hipSuccess) (void)hipGetLastError();
    }
    ~DeviceScope()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

}  // namespace mi355ntt
