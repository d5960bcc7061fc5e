// az_host.h -- host-side helpers shared by the translation units of libaz_amd.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/az_amd.h"

void az_set_error(const char *fmt, ...);

#define AZ_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            az_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return AZ_EHIP;                                                                   \
        }                                                                                     \
    } while (0)

#define AZ_REQUIRE(cond, code, ...)     \
    do {                                \
        if (!(cond)) {                  \
            az_set_error(__VA_ARGS__);  \
            return (code);              \
        }                               \
    } while (0)

struct GameDesc;
// fills gd for (game,H,W); returns AZ_OK or AZ_EINVAL with the reference's constructor conditions
// (othello.py:87-88 odd size, connect4.py:90-91 smaller than 4x4)
int az_make_game_desc(int game, int H, int W, GameDesc *gd);
