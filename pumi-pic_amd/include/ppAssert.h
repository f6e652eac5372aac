// ppAssert.h -- the always-on check of the reference's support library (support/ppAssert.h:7-22, ppAssert.cpp:10-13):
// PS_ALWAYS_ASSERT(cond) reports "<cond> failed at <file> + <line>" on stderr and aborts, in release builds too.
#pragma once
#include <cstdio>
#include <cstdlib>
namespace pumipic {
[[noreturn]] inline void Assert_Fail(const char* msg) {
  fputs(msg, stderr);
  abort();
}
inline void always_assert(bool holds, const char* what, const char* file, int line) {
  if (holds) return;
  char text[2048];
  snprintf(text, sizeof(text), "%s failed at %s + %d \n", what, file, line);
  Assert_Fail(text);
}
}  // namespace pumipic
#define PS_ALWAYS_ASSERT(cond) ::pumipic::always_assert(static_cast<bool>(cond), #cond, __FILE__, __LINE__)
