// csic_device_guard.h -- scoped "make this device current, put the caller's device back on exit".
//
// Every csic_* entry point that touches a plan bound to device k needs k to be the calling thread's
// current HIP device for the duration of the call, and must leave the thread on the device it arrived
// on: the host (PyTorch, a JVM service that owns several GPUs) keeps allocating and launching after we
// return.  The guard is a template over the runtime calls so that its logic is unit-tested without a GPU
// (tests/cpp/host_test.cpp drives it with a fake runtime); the .hip files instantiate it with HipDeviceApi.
#pragma once

namespace csic {

// Api must provide:  static int get(int *device);  static int set(int device);  both returning 0 on success.
template <class Api>
class BasicDeviceGuard {
public:
    explicit BasicDeviceGuard(int device) : prev_(-1), switched_(false), status_(0)
    {
        status_ = Api::get(&prev_);
        if (status_ == 0 && prev_ != device) {
            status_ = Api::set(device);
            switched_ = (status_ == 0);
        }
    }
    ~BasicDeviceGuard()
    {
        if (switched_) (void)Api::set(prev_);
    }
    BasicDeviceGuard(const BasicDeviceGuard &) = delete;
    BasicDeviceGuard &operator=(const BasicDeviceGuard &) = delete;

    int status() const { return status_; }      // runtime error code of the get/set that failed, 0 = ok
    int previous() const { return prev_; }
    bool switched() const { return switched_; }

private:
    int prev_;
    bool switched_;
    int status_;
};

} // namespace csic
