// pumipic_version.hpp -- src/pumipic_version.hpp.in: the version string a driver prints.  The API mirrored here is
// that of pumipic 2.1.6 (the reference tree's CMakeLists.txt:3).
#pragma once
#define PUMIPIC_VERSION_MAJOR 2
#define PUMIPIC_VERSION_MINOR 1
#define PUMIPIC_VERSION_PATCH 6
namespace pumipic {
inline const char* pumipic_version() { return "2.1.6 (pumi-pic_amd, gfx950)"; }
}
