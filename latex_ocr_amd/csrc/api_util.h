#pragma once
#include <string.h>
// propagate a non-zero launcher status
#define RC(expr) do { int rc__ = (expr); if (rc__ != 0) return rc__ < 0 ? rc__ : -rc__; } while (0)
#define HIPRC(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return -100 - (int)e__; } while (0)
