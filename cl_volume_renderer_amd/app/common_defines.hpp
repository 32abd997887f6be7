// frame image size of the shipped app (reference app/common_defines.hpp:3-4); overridable at build time
#pragma once
#ifndef SCREEN_WIDTH
#define SCREEN_WIDTH 2048
#endif
#ifndef SCREEN_HEIGHT
#define SCREEN_HEIGHT 1024
#endif
