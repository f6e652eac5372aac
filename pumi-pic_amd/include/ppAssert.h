// ppAssert.h -- support/ppAssert.h:7-22, ppAssert.cpp:10-13: PS_ALWAYS_ASSERT / pumipic::Assert_Fail
#pragma once
#include <cstdio>
#include <cstdlib>
#define PS_ALWAYS_ASSERT(cond)                                               \
  do {                                                                       \
    if (!(cond)) {                                                           \
      char omsg[2048];                                                       \
      snprintf(omsg, sizeof(omsg), "%s failed at %s + %d \n", #cond, __FILE__, __LINE__); \
      pumipic::Assert_Fail(omsg);                                            \
    }                                                                        \
  } while (0)
namespace pumipic {
inline void Assert_Fail(const char* msg) {
  fprintf(stderr, "%s", msg);
  abort();
}
}  // namespace pumipic
