// Host-side helpers shared by the C-ABI translation units (frt_api.cpp, frt_jpeg_api.cpp): error plumbing behind frt_last_error().
#pragma once
#include <hip/hip_runtime.h>

#include <exception>
#include <string>

#include "../../include/frt.h"

namespace frthost {

std::string &last_error();  // thread-local message behind frt_last_error() (defined in frt_api.cpp)

struct FrtError {
    int code;
    std::string msg;
};

[[noreturn]] inline void raise(int code, const std::string &m) { throw FrtError{code, m}; }

template <typename Fn>
int guarded(Fn &&fn) {
    try {
        fn();
        last_error().clear();
        return FRT_OK;
    } catch (const FrtError &e) {
        last_error() = e.msg;
        return e.code;
    } catch (const std::exception &e) {
        last_error() = e.what();
        return FRT_ERR_FORMAT;
    }
}

inline void use_device(int dev);

}  // namespace frthost

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) ::frthost::raise(FRT_ERR_DEVICE, std::string("HIP API failed: ") + #expr + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace frthost {
inline void use_device(int dev) { HIPCHK(hipSetDevice(dev)); }
}  // namespace frthost
