// set_last_error for the CPU-only sanitizer build of azc_contour.cpp (the real one lives in vs_api.cpp)
#include <string>
namespace vsd {
static std::string g_err;
void set_last_error(const std::string& m) { g_err = m; }
}
