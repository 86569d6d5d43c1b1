// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include "aphantasia_hip.h"
#include "aphantasia_hip_test.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

int aph_fail(int code, const char* fmt, ...);
int aph_check_launch(const char* where);

#define APH_TRY try {
#define APH_CATCH                                                            \
  }                                                                          \
  catch (const std::exception& e) { return aph_fail(APH_ERR_INTERNAL, "%s", e.what()); } \
  catch (...) { return aph_fail(APH_ERR_INTERNAL, "unknown C++ exception"); }

// a HIP runtime call whose failure must not be ignored (inside APH_TRY ... APH_CATCH)
#ifndef APH_EMU
#define APH_HIP(expr)                                                                                             \
  do {                                                                                                            \
    const hipError_t aph_e_ = (expr);                                                                             \
    if (aph_e_ != hipSuccess) throw std::runtime_error(std::string(#expr ": ") + hipGetErrorString(aph_e_));      \
  } while (0)
#else
#define APH_HIP(expr) (void)(expr)
#endif
