// One-call replay of a recorded forward (include/bts_hip.h: bts_plan_run).  Pure host code: it patches the per-call
// tensor pointers into the recorded argument lists and calls the same extern "C" entry points the recording saw, in
// order, on the given stream -- one crossing of the language boundary instead of one per launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "common.h"

extern "C" int bts_plan_run(bts_op* ops, int n_ops, const bts_plan_patch* patches, int n_patches, void* const* slots,
                            int n_slots, bts_stream_t stream) {
    if (!ops || n_ops < 0 || n_patches < 0 || (n_patches > 0 && (!patches || !slots))) return BTS_ERR_INVALID;
    for (int i = 0; i < n_patches; ++i) {
        const bts_plan_patch& p = patches[i];
        if (p.op < 0 || p.op >= n_ops || p.slot < 0 || p.slot >= n_slots || p.field_offset < 0 ||
            p.field_offset + (int)sizeof(void*) > (int)sizeof(bts_op) || (p.field_offset & 7))
            return BTS_ERR_INVALID;
        char* v = slots[p.slot] ? static_cast<char*>(slots[p.slot]) + p.delta : nullptr;
        memcpy(reinterpret_cast<char*>(&ops[p.op]) + p.field_offset, &v, sizeof v);
    }
    for (int i = 0; i < n_ops; ++i) {
        bts_op& o = ops[i];
        int rc;
        switch (o.kind) {
            case BTS_OP_CONV: rc = bts_conv_fwd_f32(&o.u.conv, stream); break;
            case BTS_OP_REDUC: {
                auto& a = o.u.reduc;
                rc = bts_reduc_fwd_f32(a.x, a.x_pix_stride, a.npix, a.c_in, a.c_first_out, a.w_frag, a.w_frag_floats, a.max_depth,
                                       a.is_final, a.normalize, a.out, stream);
                break;
            }
            case BTS_OP_REDUC_LPG: {
                auto& a = o.u.reduc_lpg;
                rc = bts_reduc_lpg_fwd_f32(a.x, a.x_pix_stride, a.B, a.h, a.w, a.c_in, a.c_first_out, a.w_frag, a.w_frag_floats,
                                           a.max_depth, a.upratio, a.plane4, a.depth_scaled, a.ds_out, a.abs_min, stream);
                break;
            }
            case BTS_OP_LPG_FUSED: {
                auto& a = o.u.lpg_fused;
                rc = bts_lpg_fused_fwd_f32(a.plane4, a.B, a.h, a.w, a.upratio, a.normalize, a.max_depth, a.depth_scaled, a.ds_out,
                                           a.ds_factor, a.ds_pix_stride, a.abs_min, stream);
                break;
            }
            case BTS_OP_LPG: {
                auto& a = o.u.lpg;
                rc = bts_lpg_fwd_f32(a.plane_eq, a.B, a.h, a.w, a.upratio, a.depth, a.abs_min, stream);
                break;
            }
            case BTS_OP_NCHW_TO_NHWC: {
                auto& a = o.u.nchw_to_nhwc;
                rc = bts_nchw_to_nhwc_f32(a.src, a.B, a.C, a.HW, a.dst, a.dst_pix_stride, a.relu, stream);
                break;
            }
            case BTS_OP_NHWC_TO_NCHW: {
                auto& a = o.u.nhwc_to_nchw;
                rc = bts_nhwc_to_nchw_f32(a.src, a.src_pix_stride, a.B, a.C, a.HW, a.dst, stream);
                break;
            }
            case BTS_OP_MAXPOOL: {
                auto& a = o.u.maxpool;
                rc = bts_maxpool3x3s2_nhwc_f32(a.src, a.src_pix_stride, a.B, a.h, a.w, a.C, a.dst, a.dst_pix_stride, a.dst2,
                                               a.dst2_pix_stride, stream);
                break;
            }
            case BTS_OP_BN_RELU_AVGPOOL: {
                auto& a = o.u.avgpool;
                rc = bts_bn_relu_avgpool2_nhwc_f32(a.src, a.src_pix_stride, a.B, a.h, a.w, a.C, a.scale, a.shift, a.dst,
                                                   a.dst_pix_stride, stream);
                break;
            }
            case BTS_OP_GET_DEPTH: {
                auto& a = o.u.get_depth;
                rc = bts_get_depth_f32(a.iconv1, a.w, a.B, a.C, a.H, a.W, a.max_depth, a.focal, a.final_depth, stream);
                break;
            }
            case BTS_OP_UPCONV_COMBINE: {
                auto& a = o.u.upconv_combine;
                rc = bts_upconv_combine_f32(a.taps, a.taps_pix_stride, a.B, a.h, a.w, a.c, a.e2_scale, a.e2_shift, a.act, a.y,
                                            a.y_pix_stride, stream);
                break;
            }
            default: rc = BTS_ERR_INVALID;
        }
        o.failed_code = rc;
        if (rc != 0) return rc;
    }
    return 0;
}
