// Logger.hpp -- minimal logging with the reference's macro names (Logger.h:38-43) and "{}" / "{:.Nf}"
// placeholders, writing to stderr.  The reference logs through spdlog; logging is out of the hot
// path's scope, so this is just enough for the host mirror to say what it does.
#pragma once

#include <cstdio>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>

namespace hvrt_log {

enum Level { TRACE = 0, DEBUG, INFO, WARN, ERROR, CRITICAL, OFF };
inline Level& threshold() { static Level l = INFO; return l; }

inline void emit_rest(std::ostringstream& os, const char* f) { os << f; }

template <typename T, typename... R>
void emit_rest(std::ostringstream& os, const char* f, const T& v, const R&... rest) {
    for (; *f; ++f) {
        if (*f == '{') {
            const char* e = f;
            while (*e && *e != '}') ++e;
            if (*e == '}') {
                std::string spec(f + 1, e);              // "", ":.3f", ":.1f", ":.6f", ":.0f"
                if (spec.size() >= 3 && spec[0] == ':' && spec[1] == '.') {
                    std::ios::fmtflags fl = os.flags();
                    std::streamsize pr = os.precision();
                    os << std::fixed << std::setprecision(std::atoi(spec.c_str() + 2)) << v;
                    os.flags(fl);
                    os.precision(pr);
                } else {
                    os << v;
                }
                emit_rest(os, e + 1, rest...);
                return;
            }
        }
        os << *f;
    }
}

template <typename... A>
void log(Level lvl, const char* tag, const char* fmt, const A&... a) {
    if (lvl < threshold()) return;
    std::ostringstream os;
    os << "[" << tag << "] ";
    emit_rest(os, fmt, a...);
    os << "\n";
    std::cerr << os.str();
}

}  // namespace hvrt_log

struct Logger {
    static void init() {}
    static void setLevel(hvrt_log::Level l) { hvrt_log::threshold() = l; }
};

#define LOG_TRACE(...) hvrt_log::log(hvrt_log::TRACE, "trace", __VA_ARGS__)
#define LOG_DEBUG(...) hvrt_log::log(hvrt_log::DEBUG, "debug", __VA_ARGS__)
#define LOG_INFO(...) hvrt_log::log(hvrt_log::INFO, "info", __VA_ARGS__)
#define LOG_WARN(...) hvrt_log::log(hvrt_log::WARN, "warning", __VA_ARGS__)
#define LOG_ERROR(...) hvrt_log::log(hvrt_log::ERROR, "error", __VA_ARGS__)
#define LOG_CRITICAL(...) hvrt_log::log(hvrt_log::CRITICAL, "critical", __VA_ARGS__)
