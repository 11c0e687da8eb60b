#pragma once
#include <string.h>
// propagate a non-zero launcher status
#define RC(expr) do { int rc__ = (expr); if (rc__ != 0) return rc__ < 0 ? rc__ : -rc__; } while (0)
