// Error plumbing shared by the C-ABI translation units: C++ exceptions never cross the ABI,
// they become a negative return code plus a thread-local message (dcvc_last_error()).
#pragma once

#include <exception>
#include <string>

namespace dcvc {

inline std::string& last_error()
{
    static thread_local std::string msg;
    return msg;
}

template <typename F>
inline int guarded(F&& f)
{
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        last_error() = e.what();
    } catch (...) {
        last_error() = "unknown C++ exception";
    }
    return -1;
}

}  // namespace dcvc
