/*
 * ref_api_driver.c -- harness around the REFERENCE's public API definitions (Source/API/EbSvtVp9Enc.h) and the parts of
 * Source/Lib/Codec/EbEncHandle.c that define the library's defaults and parameter checks.  TEST INFRASTRUCTURE ONLY
 * (rules: ref_me_driver.c).  The reference's file is compiled as part of this translation unit because
 * copy_api_from_app / verify_settings are static; nothing of the encoder runs.
 *
 *   ref_api layout            -> "name size" / "name.field offset size" lines for the four public structs, the enum values
 *   ref_api defaults          -> the bytes of an EbSvtVp9EncConfiguration pre-filled with 0xAA after eb_vp9_svt_enc_init_parameter
 *   ref_api levels            -> max_luma_picture_size[13], max_luma_sample_rate[13]
 *   ref_api verify < lines    -> per line "field=value field=value ..." applied to the defaults (+ 1920x1080): the return code of
 *                                set_default_configuration_parameters + copy_api_from_app + verify_settings, as
 *                                eb_vp9_svt_enc_set_parameter runs them (:2700-2706)
 */
#include <stddef.h>
#include "EbEncHandle.c"

#define F(T, f) printf(#T "." #f " %zu %zu\n", offsetof(T, f), sizeof(((T *)0)->f))

static int set_field(EbSvtVp9EncConfiguration *c, const char *k, long long v) {
#define S(f) if (!strcmp(k, #f)) { c->f = v; return 0; }
    S(enc_mode) S(tune) S(intra_period) S(pred_structure) S(base_layer_switch_mode) S(source_width) S(source_height) S(frame_rate)
    S(frame_rate_numerator) S(frame_rate_denominator) S(encoder_bit_depth) S(partition_depth) S(qp) S(use_qp_file)
    S(enable_qp_scaling_flag) S(loop_filter) S(use_default_me_hme) S(enable_hme_flag) S(search_area_width) S(search_area_height)
    S(rate_control_mode) S(target_bit_rate) S(max_qp_allowed) S(min_qp_allowed) S(profile) S(level) S(asm_type) S(channel_id)
    S(active_channel_count) S(speed_control_flag) S(injector_frame_rate) S(logical_processors) S(target_socket) S(recon_file)
    S(input_picture_stride) S(vbv_max_rate) S(vbv_buf_size) S(frames_to_be_encoded)
#undef S
    return -1;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    if (!strcmp(argv[1], "layout")) {
        printf("EbComponentType %zu\n", sizeof(EbComponentType));
        F(EbComponentType, n_size); F(EbComponentType, p_component_private); F(EbComponentType, p_application_private);
        printf("EbSvtEncInput %zu\n", sizeof(EbSvtEncInput));
        F(EbSvtEncInput, luma); F(EbSvtEncInput, cb); F(EbSvtEncInput, cr); F(EbSvtEncInput, luma_ext); F(EbSvtEncInput, cb_ext);
        F(EbSvtEncInput, cr_ext); F(EbSvtEncInput, y_stride); F(EbSvtEncInput, cr_stride); F(EbSvtEncInput, cb_stride);
        printf("EbBufferHeaderType %zu\n", sizeof(EbBufferHeaderType));
        F(EbBufferHeaderType, size); F(EbBufferHeaderType, p_buffer); F(EbBufferHeaderType, n_filled_len); F(EbBufferHeaderType, n_alloc_len);
        F(EbBufferHeaderType, p_app_private); F(EbBufferHeaderType, wrapper_ptr); F(EbBufferHeaderType, n_tick_count); F(EbBufferHeaderType, dts);
        F(EbBufferHeaderType, pts); F(EbBufferHeaderType, qp); F(EbBufferHeaderType, pic_type); F(EbBufferHeaderType, flags);
        printf("EbSvtVp9EncConfiguration %zu\n", sizeof(EbSvtVp9EncConfiguration));
#define C_(f) F(EbSvtVp9EncConfiguration, f)
        C_(enc_mode); C_(tune); C_(intra_period); C_(pred_structure); C_(base_layer_switch_mode); C_(source_width); C_(source_height);
        C_(frame_rate); C_(frame_rate_numerator); C_(frame_rate_denominator); C_(encoder_bit_depth); C_(partition_depth); C_(qp);
        C_(use_qp_file); C_(enable_qp_scaling_flag); C_(loop_filter); C_(use_default_me_hme); C_(enable_hme_flag); C_(search_area_width);
        C_(search_area_height); C_(rate_control_mode); C_(target_bit_rate); C_(max_qp_allowed); C_(min_qp_allowed); C_(profile);
        C_(level); C_(asm_type); C_(channel_id); C_(active_channel_count); C_(speed_control_flag); C_(injector_frame_rate);
        C_(logical_processors); C_(target_socket); C_(recon_file); C_(input_picture_stride); C_(vbv_max_rate); C_(vbv_buf_size);
        C_(frames_to_be_encoded);
        printf("enum.EB_ErrorNone %lld 4\nenum.EB_ErrorInsufficientResources %lld 4\nenum.EB_ErrorUndefined %lld 4\nenum.EB_ErrorInvalidComponent %lld 4\n"
               "enum.EB_ErrorBadParameter %lld 4\nenum.EB_NoErrorEmptyQueue %lld 4\nenum.EB_ErrorMax %lld 4\nenum.EB_BUFFERFLAG_EOS %lld 4\n"
               "enum.EB_BUFFERFLAG_SHOW_EXT %lld 4\n",
               (long long)EB_ErrorNone, (long long)EB_ErrorInsufficientResources, (long long)EB_ErrorUndefined, (long long)EB_ErrorInvalidComponent,
               (long long)EB_ErrorBadParameter, (long long)EB_NoErrorEmptyQueue, (long long)EB_ErrorMax, (long long)EB_BUFFERFLAG_EOS,
               (long long)EB_BUFFERFLAG_SHOW_EXT);
        return 0;
    }
    if (!strcmp(argv[1], "defaults")) {
        EbSvtVp9EncConfiguration c;
        memset(&c, 0xAA, sizeof c);
        if (eb_vp9_svt_enc_init_parameter(&c) != EB_ErrorNone) return 3;
        fprintf(stderr, "DEFAULTS");
        for (size_t i = 0; i < sizeof c; i++) fprintf(stderr, " %u", ((unsigned char *)&c)[i]);
        fprintf(stderr, "\n");
        return 0;
    }
    if (!strcmp(argv[1], "levels")) {
        fprintf(stderr, "LEVELS");
        for (int i = 0; i < TOTAL_LEVEL_COUNT; i++) fprintf(stderr, " %llu", (unsigned long long)max_luma_picture_size[i]);
        for (int i = 0; i < TOTAL_LEVEL_COUNT; i++) fprintf(stderr, " %llu", (unsigned long long)max_luma_sample_rate[i]);
        fprintf(stderr, "\n");
        return 0;
    }
    if (!strcmp(argv[1], "verify")) {
        char line[1024];
        while (fgets(line, sizeof line, stdin)) {
            EbSvtVp9EncConfiguration c;
            memset(&c, 0, sizeof c);
            eb_vp9_svt_enc_init_parameter(&c);
            c.source_width = 1920; c.source_height = 1080;
            for (char *tok = strtok(line, " \n"); tok; tok = strtok(NULL, " \n")) {
                char *eq = strchr(tok, '=');
                if (!eq) continue;
                *eq = 0;
                if (set_field(&c, tok, atoll(eq + 1))) return 4;
            }
            SequenceControlSet *scs = (SequenceControlSet *)calloc(1, sizeof *scs);
            scs->video_usability_info_ptr = (AppVideoUsabilityInfo *)calloc(1, sizeof(AppVideoUsabilityInfo));
            scs->encode_context_ptr = (EncodeContext *)calloc(1, sizeof(EncodeContext));
            set_default_configuration_parameters(scs);
            copy_api_from_app(scs, &c);
            const EbErrorType rc = verify_settings(scs);
            fprintf(stderr, "VERIFY %lld\n", (long long)rc);
        }
        return 0;
    }
    return 2;
}
