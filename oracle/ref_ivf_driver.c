/*
 * ref_ivf_driver.c -- harness that runs the REFERENCE application's own IVF header writers, write_ivf_stream_header and
 * write_ivf_frame_header (Source/App/EbAppProcessCmd.c:515-559).  They are static functions of that file, so the file is
 * included as it lies (the way ref_rate_driver.c includes vp9_rd.c); nothing else of it is called -- the encoder library
 * it would talk to cannot be built here (yasm), its symbols stay unresolved and untouched.  TEST INFRASTRUCTURE ONLY.
 *
 * usage: ref_ivf_headers out.bin width height frame_rate_q16 numerator denominator [byte_count pts]...
 * writes the 32-byte stream header, then one 12-byte frame header per (byte_count, pts) pair.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "EbAppProcessCmd.c"

int main(int argc, char **argv) {
    if (argc < 7) return 2;
    static EbConfig cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.bitstream_file = fopen(argv[1], "wb");
    if (!cfg.bitstream_file) return 2;
    cfg.source_width           = (uint32_t)strtoul(argv[2], 0, 0);
    cfg.source_height          = (uint32_t)strtoul(argv[3], 0, 0);
    cfg.frame_rate             = (int32_t)strtoul(argv[4], 0, 0);
    cfg.frame_rate_numerator   = (int32_t)strtoul(argv[5], 0, 0);
    cfg.frame_rate_denominator = (int32_t)strtoul(argv[6], 0, 0);
    write_ivf_stream_header(&cfg);
    for (int i = 7; i + 1 < argc; i += 2) write_ivf_frame_header(&cfg, (uint32_t)strtoul(argv[i], 0, 0), strtoull(argv[i + 1], 0, 0));
    fclose(cfg.bitstream_file);
    return 0;
}
