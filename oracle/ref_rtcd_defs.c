/*
 * ref_rtcd_defs.c -- instantiates the reference's run-time-dispatch function pointers for the kernel-level
 * reference library (oracle/_ref/libsvtref_kernels.so).  TEST INFRASTRUCTURE ONLY; contains no code.
 *
 * The reference defines its RTCD pointers by including VPX/vpx_dsp_rtcd.h and VPX/vp9_rtcd.h with RTCD_C
 * defined in exactly one translation unit (Codec/EbEncHandle.c:74).  EbEncHandle.c itself cannot be linked
 * here (it drags in the whole encoder incl. the yasm-only symbols), so this file is that one translation
 * unit: the two #includes below are the reference's own headers and they produce the definitions.
 * The pointers stay NULL; the tests only call the `_c` kernels directly (VPX/vp9_idct.c's iht*_add_c need
 * the symbols to exist because the eob-dispatch wrappers of the same file reference them).
 */
#define RTCD_C
#include "vpx_dsp_rtcd.h"
#include "vp9_rtcd.h"
