// TEST INFRASTRUCTURE: a silent stand-in for the part of spdlog the reference's C++ names (see ../README.md).
#pragma once
#include <cstring>
#include <memory>
#include <string>

namespace spdlog {
namespace level { enum level_enum { trace, debug, info, warn, err, critical, off }; }
class logger {
 public:
    template <class... A> void trace(const char*, A&&...) {}
    template <class... A> void debug(const char*, A&&...) {}
    template <class... A> void info(const char*, A&&...) {}
    template <class... A> void warn(const char*, A&&...) {}
    template <class... A> void error(const char*, A&&...) {}
    template <class... A> void critical(const char*, A&&...) {}
};
inline std::shared_ptr<logger> default_logger() {
    static std::shared_ptr<logger> l = std::make_shared<logger>();
    return l;
}
inline void set_pattern(const std::string&) {}
inline void set_level(level::level_enum) {}
}  // namespace spdlog
