// Shared by the host translation units of the library (wbc_api.cpp, wbc_multi.cpp): the thread-local error string.
#pragma once
#include <string>
namespace wbc {
int fail(int code, const std::string& msg);   // records msg for wbc_last_error(), returns code
}
