/*
 * ref_pa_defs.c -- TEST INFRASTRUCTURE ONLY (compiled only where /root/reference exists, into oracle/_ref/libsvtref_pa.so).
 *
 * The reference's picture-analysis leaf functions (eb_vp9_decimation_2d, eb_vp9_generate_padding, the 8x8 mean kernels)
 * live in translation units that also mention the encoder's global bookkeeping variables.  In the encoder these
 * variables are defined by Codec/EbEncHandle.c (the dispatch selector eb_vp9_ASM_TYPES, the allocation-tracking table of
 * EB_MALLOC); this file plays that role for the test library: it defines the variables, with the reference's own
 * declarations (Codec/EbDefinitions.h:215, 464-468), and contains no algorithmic code.
 */
#include "EbDefinitions.h"

uint32_t          eb_vp9_ASM_TYPES = 0; /* `-asm 0`: C kernels */
EbMemoryMapEntry *memory_map       = 0;
uint32_t         *memory_map_index = 0;
uint64_t         *total_lib_memory = 0;
uint32_t          lib_malloc_count = 0;
