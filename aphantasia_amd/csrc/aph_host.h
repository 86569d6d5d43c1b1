// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include "aphantasia_hip.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <exception>
#include <vector>

int aph_fail(int code, const char* fmt, ...);
int aph_check_launch(const char* where);

#define APH_TRY try {
#define APH_CATCH                                                            \
  }                                                                          \
  catch (const std::exception& e) { return aph_fail(APH_ERR_INTERNAL, "%s", e.what()); } \
  catch (...) { return aph_fail(APH_ERR_INTERNAL, "unknown C++ exception"); }
