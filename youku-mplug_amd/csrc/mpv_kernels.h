// mpv_kernels.h -- internal: pulls the public C ABI into the kernel translation units.
#pragma once
#include "../../include/mpv.h"
