// device_scope.hpp -- RAII "make this HIP device current, restore the caller's on exit".
//
// Every host entry point that touches a robot's / chain's GPU binds that GPU for the calling
// thread.  The reference's host API (crates/optik-cpp/src/lib.rs:26-183) has no notion of a
// current device, so a caller that also drives torch / RCCL on the same thread must find its own
// device current again when the call returns -- on every exit path, failed ones included.
#pragma once

#include <hip/hip_runtime.h>

namespace optik {

class DeviceScope {
public:
    explicit DeviceScope(int device) {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        ok_ = (prev_ == device) || hipSetDevice(device) == hipSuccess;
        changed_ = ok_ && prev_ != device;
    }
    ~DeviceScope() {
        if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
    bool ok() const { return ok_; }

private:
    int prev_ = -1;
    bool ok_ = false, changed_ = false;
};

}  // namespace optik
