// FATAL(): print and abort.  API contract: reference include/madrona/crash.hpp.
#pragma once

#include <madrona/macros.hpp>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

namespace madrona {

[[noreturn]] inline void fatal(const char *file, int line, const char *func,
                               const char *fmt, ...)
{
    fprintf(stderr, "Error at %s:%d in %s\n", file, line, func);
    va_list args;
    va_start(args, fmt);
    vfprintf(stderr, fmt, args);
    va_end(args);
    fprintf(stderr, "\n");
    fflush(stderr);
    abort();
}

}

#define FATAL(fmt, ...) ::madrona::fatal(__FILE__, __LINE__, __func__, \
                                         fmt __VA_OPT__(,) __VA_ARGS__)
