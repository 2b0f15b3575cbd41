/*
 * ivf.c -- the IVF container layer of the reference's sample application, as buffer-to-buffer functions
 * (App/EbAppProcessCmd.c:515-567 stream and frame headers, :621-650 the splitting of a packet that ends in a
 * show-existing-frame group).  Host-side C, no GPU involved: this is the "data format behind the path" a caller needs to
 * keep when the application around the encoder library is replaced.
 */
#include <string.h>
#include "../../include/svtvp9_hip.h"

static void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put32(uint8_t *p, uint32_t v) { put16(p, v); put16(p + 2, v >> 16); }

/* 32-byte stream header: "DKIF", version 0, header size 32, fourcc "VP90", width, height, time base, frame count 0
 * (the reference never patches the count).  With a numerator / denominator pair the time base is that pair; otherwise it
 * is derived from the Q16 frame rate as (frame_rate >> 16) * 1000 over 1000 (App/EbAppProcessCmd.c:526-532). */
int32_t svt_ivf_stream_header(uint8_t out[SVT_IVF_STREAM_HEADER_BYTES], uint32_t width, uint32_t height, uint32_t frame_rate_q16,
                              uint32_t rate_numerator, uint32_t rate_denominator) {
    if (!out) return SVT_HIP_ERR_BAD_PARAMETER;
    memcpy(out, "DKIF", 4);
    put16(out + 4, 0);
    put16(out + 6, 32);
    put32(out + 8, 0x30395056u); /* "VP90" */
    put16(out + 12, width);
    put16(out + 14, height);
    if (rate_numerator != 0 && rate_denominator != 0) { put32(out + 16, rate_numerator); put32(out + 20, rate_denominator); }
    else { put32(out + 16, (frame_rate_q16 >> 16) * 1000u); put32(out + 20, 1000u); }
    put32(out + 24, 0);
    put32(out + 28, 0);
    return SVT_HIP_OK;
}

/* 12-byte frame header: payload size, 64-bit presentation time stamp (App/EbAppProcessCmd.c:545-559) */
static uint8_t *frame(uint8_t *o, const uint8_t *payload, uint32_t n, uint64_t pts) {
    put32(o, n);
    put32(o + 4, (uint32_t)(pts & 0xffffffffu));
    put32(o + 8, (uint32_t)(pts >> 32));
    memcpy(o + 12, payload, n);
    return o + 12 + n;
}

/* One output packet of the encoder -> IVF frames.  A packet flagged SHOW_EXT carries, after the coded frame, four
 * one-byte show-existing-frame headers; the reference writes the coded frame with the packet's pts and then the four
 * bytes as IVF frames of their own with pts - 2, pts - 1, pts, pts + 1 (App/EbAppProcessCmd.c:621-646).
 * Returns the number of bytes written to out, or a negative SVT_HIP_ERR_* (capacity: len + 12, or len + 60 with show_ext). */
int64_t svt_ivf_packetize(const uint8_t *packet, uint32_t len, uint64_t pts, int32_t show_ext, uint8_t *out, size_t capacity) {
    if (!packet || !out) return SVT_HIP_ERR_BAD_PARAMETER;
    if (!show_ext) {
        if (capacity < (size_t)len + 12) return SVT_HIP_ERR_BAD_PARAMETER;
        return frame(out, packet, len, pts) - out;
    }
    if (len < 4 || capacity < (size_t)len + 12 + 4 * 12) return SVT_HIP_ERR_BAD_PARAMETER;
    uint8_t *o = frame(out, packet, len - 4, pts);
    static const int64_t dpts[4] = {-2, -1, 0, 1};
    for (int i = 0; i < 4; i++) o = frame(o, packet + len - 4 + i, 1, pts + (uint64_t)dpts[i]);
    return o - out;
}
