/* every kernel header at once: host emulation (tests/emu) and the resource report */
#pragma once
#include "smol_kernels.hip.h"
#include "smol_kernel_tile.hip.h"
#include "smol_kernel_wtile.hip.h"
#include "smol_kernel_pow2x.hip.h"
#include "smol_kernel_up.hip.h"
#include "smol_kernel_lind.hip.h"
#include "smol_kernel_box2.hip.h"
#include "smol_kernel_hspan.hip.h"
