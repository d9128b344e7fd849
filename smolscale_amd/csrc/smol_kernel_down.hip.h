/* smol_kernel_down.hip.h -- bilinear minification on both axes in the 128bpp class: since round 4 every flavour of it
 * runs through k_lind (smol_kernel_lind.hip.h): premul16 with and without sRGB (32-bit channels) and the sRGB-linear
 * flavours (11-bit channels), 32- and 24-bit rows at any byte alignment. Rounds 1-3 had k_down (premul16 fast path,
 * one or four columns per lane) and k_downg (table-driven flavours) here; k_lind measured faster than both on every row
 * (profiles/r04_lind_ab.txt: gamma-correct 99 -> 63 us, gamma-correct photographs 118 -> 62, premul16 + sRGB 90 -> 67,
 * premul16 57 -> 49) and they were removed. What is left in this file is the host emulation's entry point. */

#pragma once

#include "smol_kernel_lind.hip.h"

namespace smolhip {

#ifdef SMOL_KERNEL_EMU
/* host emulation: one entry that switches at run time */
__global__ void k_down_one (const JobDev J)
{
    const int in = J.p16 ? 2 : J.bpp_in == 3 ? 1 : 0;
    if (dispatch::lind_narrow (J))              /* tiles of at most 64 columns: the one-column flavour */
    {
        switch (J.halv_h * 3 + in)
        {
            case 3: lind_entry<1, false, false, true> (J, smol_dyn_lds); break;
            case 4: lind_entry<1, true, false, true> (J, smol_dyn_lds); break;
            case 5: lind_entry<1, false, true, true> (J, smol_dyn_lds); break;
            case 6: lind_entry<2, false, false, true> (J, smol_dyn_lds); break;
            case 7: lind_entry<2, true, false, true> (J, smol_dyn_lds); break;
            default: lind_entry<2, false, true, true> (J, smol_dyn_lds); break;
        }
        return;
    }
    switch (J.halv_h * 3 + in)
    {
        case 0: lind_entry<0, false, false> (J, smol_dyn_lds); break;
        case 1: lind_entry<0, true, false> (J, smol_dyn_lds); break;
        case 2: lind_entry<0, false, true> (J, smol_dyn_lds); break;
        case 3: lind_entry<1, false, false> (J, smol_dyn_lds); break;
        case 4: lind_entry<1, true, false> (J, smol_dyn_lds); break;
        case 5: lind_entry<1, false, true> (J, smol_dyn_lds); break;
        case 6: lind_entry<2, false, false> (J, smol_dyn_lds); break;
        case 7: lind_entry<2, true, false> (J, smol_dyn_lds); break;
        default: lind_entry<2, false, true> (J, smol_dyn_lds); break;
    }
}
#endif

} /* namespace smolhip */
