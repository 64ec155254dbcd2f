// torch operator boundary over the C ABI of libbts_hip.so (include/bts_hip.h) -- `TORCH_LIBRARY(bts_hip, ...)`.
//
// The reference's own native side registers its hot op with the framework it runs under:
// tensorflow/custom_layer/local_planar_guidance.cc:31-72 (REGISTER_OP "LocalPlanarGuidance" + shape function),
// :116-156 (the OpKernel: validates `upratio`, allocates the output, hands raw pointers + dims to the functor) and
// :234-239 (the gradient op).  This file is the same layer for PyTorch-ROCm: every operator
//   * checks device / dtype / shape / contiguity / alignment with TORCH_CHECK (the C ABI itself only returns codes),
//   * sets a c10::hip device guard for the tensors' device and launches on the CURRENT HIP stream,
//   * hands raw device pointers and dims to the extern "C" entry point -- no torch type crosses that line.
// Built with g++ only (no device code): `make -C bts_amd/csrc torch` -> bts_amd/libbts_torch.so, loaded by
// bts_amd/_lib.py with torch.ops.load_library.  Autograd for `bts_hip::lpg` is registered in bts_amd/ops.py
// (torch.library.register_autograd) on top of `bts_hip::lpg_backward`.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>      // PyTorch-ROCm tensors carry the device type "cuda": its guard / stream
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>         // accessors are the *MasqueradingAsCUDA forms
#include <torch/library.h>

#include <cstdint>
#include <tuple>
#include <vector>

#include "../../include/bts_hip.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<Tensor>;

const char* err_text(int rc) { return bts_hip_error_string(rc); }

void need_f32_cuda(const Tensor& t, const char* op, const char* what) {
    TORCH_CHECK(t.is_cuda(), op, ": ", what, " must be a CUDA(ROCm) tensor (no CPU fallback)");
    TORCH_CHECK(t.scalar_type() == at::kFloat, op, ": ", what, " must be float32, got ", t.scalar_type());
}

void same_device(const Tensor& a, const Tensor& b, const char* op, const char* what) {
    TORCH_CHECK(a.device() == b.device(), op, ": ", what, " lives on ", b.device(), ", expected ", a.device());
}

// [rows, C] view of an NHWC buffer: unit channel stride, any row stride that is a multiple of 4 floats, 16-byte aligned
long rows2d_stride(const Tensor& t, const char* op, const char* what) {
    need_f32_cuda(t, op, what);
    TORCH_CHECK(t.dim() == 2 && t.stride(1) == 1, op, ": ", what, " must be a 2-D view with unit channel stride");
    const long s = t.size(0) > 1 ? t.stride(0) : t.size(1);
    TORCH_CHECK(s % 4 == 0 && (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0, op, ": ", what,
                " must have a pixel stride that is a multiple of 4 floats and a 16-byte aligned base");
    return s;
}

float* opt_ptr(const OptTensor& t) { return t.has_value() && t->defined() ? t->data_ptr<float>() : nullptr; }

bts_stream_t current_stream() { return (bts_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

void check_rc(int rc, const char* op) { TORCH_CHECK(rc == 0, op, " failed: ", err_text(rc), " (code ", rc, ")"); }

// ---------------------------------------------------------------------------------------------- local_planar_guidance
// pytorch/bts.py:138-173; returns (depth [B, h*k, w*k], abs_min 0-d).  Bit-exact module-level op (csrc/lpg.hip).
std::tuple<Tensor, Tensor> lpg(const Tensor& plane_eq, int64_t upratio) {
    need_f32_cuda(plane_eq, "bts_hip::lpg", "plane_eq");
    TORCH_CHECK(plane_eq.dim() == 4 && plane_eq.size(1) == 4, "bts_hip::lpg: plane_eq must be [B,4,h,w], got ", plane_eq.sizes());
    TORCH_CHECK(upratio == 1 || upratio == 2 || upratio == 4 || upratio == 8,
                "bts_hip::lpg: upratio must be 1, 2, 4 or 8 (local_planar_guidance.cc:36-44 rejects the rest), got ", upratio);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(plane_eq.device());
    const Tensor pe = plane_eq.contiguous();
    const int B = (int)pe.size(0), h = (int)pe.size(2), w = (int)pe.size(3), k = (int)upratio;
    Tensor depth = at::empty({B, h * k, w * k}, pe.options());
    Tensor abs_min = at::empty({}, pe.options());
    check_rc(bts_lpg_fwd_f32(pe.data_ptr<float>(), B, h, w, k, depth.data_ptr<float>(), abs_min.data_ptr<float>(), current_stream()),
             "bts_hip::lpg");
    return {depth, abs_min};
}

// gradient w.r.t. plane_eq (autograd of bts.py:149-173 incl. the clamp mask; native statement: local_planar_guidance.cu:95-150)
Tensor lpg_backward(const Tensor& plane_eq, const Tensor& grad_depth, int64_t upratio) {
    need_f32_cuda(plane_eq, "bts_hip::lpg_backward", "plane_eq");
    need_f32_cuda(grad_depth, "bts_hip::lpg_backward", "grad_depth");
    same_device(plane_eq, grad_depth, "bts_hip::lpg_backward", "grad_depth");
    TORCH_CHECK(plane_eq.dim() == 4 && plane_eq.size(1) == 4, "bts_hip::lpg_backward: plane_eq must be [B,4,h,w]");
    const int B = (int)plane_eq.size(0), h = (int)plane_eq.size(2), w = (int)plane_eq.size(3), k = (int)upratio;
    TORCH_CHECK(grad_depth.numel() == (int64_t)B * h * k * w * k, "bts_hip::lpg_backward: grad_depth must have B*h*k*w*k elements");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(plane_eq.device());
    const Tensor pe = plane_eq.contiguous(), gd = grad_depth.contiguous();
    Tensor g = at::empty_like(pe);
    check_rc(bts_lpg_bwd_f32(pe.data_ptr<float>(), gd.data_ptr<float>(), B, h, w, k, g.data_ptr<float>(), current_stream()),
             "bts_hip::lpg_backward");
    return g;
}

// ------------------------------------------------------------------------------------------------------ reduction_1x1
// pytorch/bts.py:97-136, the whole chain in one launch; x2d: NHWC [npix, >= c_in] view; out: [npix*4] (or [npix] when final)
void reduction_1x1(const Tensor& x2d, int64_t c_in, int64_t c_first_out, const Tensor& w_frag, double max_depth, bool is_final,
                   bool normalize, Tensor out) {
    const long xs = rows2d_stride(x2d, "bts_hip::reduction_1x1", "x2d");
    need_f32_cuda(w_frag, "bts_hip::reduction_1x1", "w_frag");
    need_f32_cuda(out, "bts_hip::reduction_1x1", "out");
    same_device(x2d, w_frag, "bts_hip::reduction_1x1", "w_frag");
    same_device(x2d, out, "bts_hip::reduction_1x1", "out");
    TORCH_CHECK(x2d.size(1) >= c_in, "bts_hip::reduction_1x1: x2d has ", x2d.size(1), " channels, the chain reads ", c_in);
    TORCH_CHECK(w_frag.is_contiguous() && out.is_contiguous(), "bts_hip::reduction_1x1: w_frag and out must be contiguous");
    const int64_t npix = x2d.size(0);
    TORCH_CHECK(out.numel() == npix * (is_final ? 1 : 4), "bts_hip::reduction_1x1: out must hold ", npix * (is_final ? 1 : 4), " floats");
    TORCH_CHECK(max_depth > 0, "bts_hip::reduction_1x1: max_depth must be positive");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x2d.device());
    check_rc(bts_reduc_fwd_f32(x2d.data_ptr<float>(), xs, npix, (int)c_in, (int)c_first_out, w_frag.data_ptr<float>(), w_frag.numel(),
                               (float)max_depth, is_final ? 1 : 0, normalize ? 1 : 0, out.data_ptr<float>(), current_stream()),
             "bts_hip::reduction_1x1");
}

// reduction_1x1 -> F.normalize -> LPG -> /max_depth (+ nearest-downsampled plane, abs_min): bts.py:249-256 / 263-270 / 277-283
void reduc_lpg(const Tensor& x2d, int64_t B, int64_t h, int64_t w, int64_t c_in, int64_t c_first_out, const Tensor& w_frag,
               double max_depth, int64_t upratio, Tensor depth_scaled, OptTensor ds_out, OptTensor abs_min, OptTensor plane4) {
    const long xs = rows2d_stride(x2d, "bts_hip::reduc_lpg", "x2d");
    need_f32_cuda(w_frag, "bts_hip::reduc_lpg", "w_frag");
    need_f32_cuda(depth_scaled, "bts_hip::reduc_lpg", "depth_scaled");
    same_device(x2d, w_frag, "bts_hip::reduc_lpg", "w_frag");
    same_device(x2d, depth_scaled, "bts_hip::reduc_lpg", "depth_scaled");
    const int64_t npix = B * h * w, k = upratio;
    TORCH_CHECK(B > 0 && h > 0 && w > 0 && x2d.size(0) == npix && x2d.size(1) >= c_in, "bts_hip::reduc_lpg: x2d ", x2d.sizes(),
                " does not match B=", B, " ", h, "x", w, " with ", c_in, " channels");
    TORCH_CHECK(depth_scaled.is_contiguous() && depth_scaled.numel() == npix * k * k, "bts_hip::reduc_lpg: depth_scaled must be contiguous [B,1,h*k,w*k]");
    for (const OptTensor* t : {&ds_out, &abs_min, &plane4})
        if (t->has_value() && (*t)->defined()) {
            need_f32_cuda(**t, "bts_hip::reduc_lpg", "optional output");
            same_device(x2d, **t, "bts_hip::reduc_lpg", "optional output");
            TORCH_CHECK((*t)->is_contiguous(), "bts_hip::reduc_lpg: optional outputs must be contiguous");
        }
    if (ds_out.has_value() && ds_out->defined()) TORCH_CHECK(k != 2 && ds_out->numel() == npix * 4, "bts_hip::reduc_lpg: ds_out must be a [B,2h,2w] plane (k = 8 or 4)");
    if (plane4.has_value() && plane4->defined()) TORCH_CHECK(plane4->numel() == npix * 4, "bts_hip::reduc_lpg: plane4 must be [B*h*w,4]");
    if (abs_min.has_value() && abs_min->defined()) TORCH_CHECK(abs_min->numel() == 1, "bts_hip::reduc_lpg: abs_min must hold one float");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x2d.device());
    check_rc(bts_reduc_lpg_fwd_f32(x2d.data_ptr<float>(), xs, (int)B, (int)h, (int)w, (int)c_in, (int)c_first_out, w_frag.data_ptr<float>(),
                                   w_frag.numel(), (float)max_depth, (int)k, opt_ptr(plane4), depth_scaled.data_ptr<float>(), opt_ptr(ds_out),
                                   opt_ptr(abs_min), current_stream()),
             "bts_hip::reduc_lpg");
}

// ------------------------------------------------------------------------------------------------------------- conv
// One fused convolution = one bts_conv_desc (include/bts_hip.h): atrous_conv's 1x1 and dilated 3x3 (bts.py:65-80), upconv
// (bts.py:83-94), conv5..conv1, daspp_conv.  geom = {x_pix_stride, c_in_ld, k_pad, B, h_in, w_in, up, ksize, dil, stride, pad,
// c_out, c_out_pad, pre_relu, act, y_pix_stride, y_nchw, subpixel, y2_pix_stride, res_pix_stride, n_bundles, precision,
// fill_frames}.
constexpr int kGeomLen = 23;

void conv_fwd(const Tensor& x, const Tensor& w, OptTensor pre_scale, OptTensor pre_shift, OptTensor e1_scale, OptTensor e1_shift,
              OptTensor e2_scale, OptTensor e2_shift, Tensor y, OptTensor y2, OptTensor res, OptTensor splitk_ws,
              std::vector<Tensor> tail_planes, OptTensor w_split, OptTensor w_wino, std::vector<int64_t> geom) {
    const char* op = "bts_hip::conv_fwd";
    TORCH_CHECK((int)geom.size() == kGeomLen, op, ": geom must hold ", kGeomLen, " integers, got ", geom.size());
    need_f32_cuda(x, op, "x");
    need_f32_cuda(w, op, "w");
    need_f32_cuda(y, op, "y");
    same_device(x, w, op, "w");
    same_device(x, y, op, "y");
    TORCH_CHECK(w.is_contiguous(), op, ": packed weights must be contiguous");
    bts_conv_desc d{};
    int i = 0;
    d.x_pix_stride = geom[i++]; d.c_in_ld = (int)geom[i++]; d.k_pad = (int)geom[i++];
    d.B = (int)geom[i++]; d.h_in = (int)geom[i++]; d.w_in = (int)geom[i++]; d.up = (int)geom[i++];
    d.ksize = (int)geom[i++]; d.dil = (int)geom[i++]; d.stride = (int)geom[i++]; d.pad = (int)geom[i++];
    d.c_out = (int)geom[i++]; d.c_out_pad = (int)geom[i++]; d.pre_relu = (int)geom[i++]; d.act = (int)geom[i++];
    d.y_pix_stride = geom[i++]; d.y_nchw = (int)geom[i++]; d.subpixel = (int)geom[i++]; d.y2_pix_stride = geom[i++];
    d.res_pix_stride = geom[i++]; d.n_bundles = (int)geom[i++]; d.precision = (int)geom[i++]; d.fill_frames = (int)geom[i++];
    TORCH_CHECK(d.B > 0 && d.h_in > 0 && d.w_in > 0 && d.c_in_ld > 0 && d.c_out > 0 && d.x_pix_stride > 0, op, ": non-positive dimension");
    const int nb = d.n_bundles > 1 ? d.n_bundles : 1;
    // the input view must cover the last pixel's channels; the weights the packed [classes][c_out_pad][k_pad] block
    const int64_t in_pix = (int64_t)d.B * d.h_in * d.w_in;
    const int64_t x_need = (in_pix - 1) * d.x_pix_stride + (int64_t)(d.c_in_ld - ((int)tail_planes.size() > 0 ? 4 : 0)) * nb;
    TORCH_CHECK(x.dim() >= 1 && x.stride(-1) == 1, op, ": x must have unit channel stride");
    TORCH_CHECK((int64_t)x.numel() > 0 && x_need <= (int64_t)(x.storage().nbytes() / 4 - x.storage_offset()), op, ": x (", x.sizes(),
                ") is smaller than B*h_in*w_in pixels of ", d.c_in_ld, " channels at pixel stride ", d.x_pix_stride);
    const int64_t classes = d.subpixel ? 4 : nb;
    TORCH_CHECK(w.numel() == classes * (int64_t)d.c_out_pad * d.k_pad, op, ": packed weights hold ", w.numel(), " floats, expected ",
                classes * (int64_t)d.c_out_pad * d.k_pad, " ([classes][c_out_pad][k_pad])");
    d.x = x.data_ptr<float>();
    d.w = w.data_ptr<float>();
    auto vec = [&](const OptTensor& t, int64_t n, const char* what) -> const float* {
        if (!(t.has_value() && t->defined())) return nullptr;
        need_f32_cuda(*t, op, what);
        same_device(x, *t, op, what);
        TORCH_CHECK(t->is_contiguous() && t->numel() == n, op, ": ", what, " must be a contiguous vector of ", n, " floats, got ", t->sizes());
        return t->data_ptr<float>();
    };
    d.pre_scale = vec(pre_scale, (int64_t)d.c_in_ld * nb, "pre_scale");
    d.pre_shift = vec(pre_shift, (int64_t)d.c_in_ld * nb, "pre_shift");
    d.e1_scale = vec(e1_scale, (int64_t)d.c_out_pad * nb, "e1_scale");
    d.e1_shift = vec(e1_shift, (int64_t)d.c_out_pad * nb, "e1_shift");
    d.e2_scale = vec(e2_scale, (int64_t)d.c_out_pad * nb, "e2_scale");
    d.e2_shift = vec(e2_shift, (int64_t)d.c_out_pad * nb, "e2_shift");
    // output extent: the library computes H, W from the geometry; mirror it for the bounds check
    const int Hs = d.h_in * d.up, Ws = d.w_in * d.up;
    int H = (Hs + 2 * d.pad - d.dil * (d.ksize - 1) - 1) / (d.stride > 0 ? d.stride : 1) + 1;
    int W = (Ws + 2 * d.pad - d.dil * (d.ksize - 1) - 1) / (d.stride > 0 ? d.stride : 1) + 1;
    if (d.subpixel) { H = 2 * d.h_in; W = 2 * d.w_in; }
    TORCH_CHECK(H > 0 && W > 0, op, ": empty output");
    const int64_t out_pix = (int64_t)d.B * H * W;
    auto out_check = [&](const Tensor& t, int64_t pix_stride, bool nchw, const char* what) {
        const int64_t have = (int64_t)(t.storage().nbytes() / 4) - t.storage_offset();
        const int64_t need = nchw ? out_pix * d.c_out : (out_pix - 1) * pix_stride + (int64_t)d.c_out * nb;
        TORCH_CHECK(need <= have, op, ": ", what, " (", t.sizes(), ") is smaller than the ", d.B, "x", H, "x", W, "x", d.c_out * nb, " result");
    };
    out_check(y, d.y_pix_stride, d.y_nchw != 0, "y");
    d.y = y.data_ptr<float>();
    if (y2.has_value() && y2->defined()) { need_f32_cuda(*y2, op, "y2"); same_device(x, *y2, op, "y2"); out_check(*y2, d.y2_pix_stride, false, "y2"); d.y2 = y2->data_ptr<float>(); }
    if (res.has_value() && res->defined()) { need_f32_cuda(*res, op, "res"); same_device(x, *res, op, "res"); out_check(*res, d.res_pix_stride, false, "res"); d.res = res->data_ptr<float>(); }
    if (splitk_ws.has_value() && splitk_ws->defined()) {
        need_f32_cuda(*splitk_ws, op, "splitk_ws");
        same_device(x, *splitk_ws, op, "splitk_ws");
        TORCH_CHECK(splitk_ws->is_contiguous(), op, ": splitk_ws must be contiguous");
        d.splitk_ws = splitk_ws->data_ptr<float>();
        d.splitk_ws_floats = splitk_ws->numel();
    }
    TORCH_CHECK(tail_planes.size() <= 4, op, ": at most 4 tail planes");
    d.n_tail = (int)tail_planes.size();
    for (int j = 0; j < d.n_tail; ++j) {
        need_f32_cuda(tail_planes[j], op, "tail plane");
        same_device(x, tail_planes[j], op, "tail plane");
        TORCH_CHECK(tail_planes[j].is_contiguous() && tail_planes[j].numel() == in_pix, op, ": every tail plane must be a contiguous [B,1,h_in,w_in] map");
        d.tail_planes[j] = tail_planes[j].data_ptr<float>();
    }
    if (w_split.has_value() && w_split->defined()) {
        TORCH_CHECK(w_split->is_cuda() && w_split->scalar_type() == at::kShort && w_split->is_contiguous() && w_split->numel() == 3 * w.numel(), op,
                    ": w_split must be the contiguous int16 [classes][3][c_out_pad][k_pad] split of w (ops.split_bf16x3)");
        same_device(x, *w_split, op, "w_split");
        d.w_split = w_split->data_ptr();
    }
    if (w_wino.has_value() && w_wino->defined()) {
        need_f32_cuda(*w_wino, op, "w_wino");
        same_device(x, *w_wino, op, "w_wino");
        const int64_t cm = d.n_tail > 0 ? d.c_in_ld - 4 : d.c_in_ld;
        TORCH_CHECK(w_wino->is_contiguous() && (w_wino->numel() == (int64_t)16 * d.c_out_pad * cm || w_wino->numel() == (int64_t)16 * d.c_out * cm), op,
                    ": w_wino must be the contiguous Winograd form of w, 16 * c_out_pad * (buffer channels) floats (ops.pack_wino_weight)");
        d.w_wino = w_wino->data_ptr<float>();
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    check_rc(bts_conv_fwd_f32(&d, current_stream()), op);
}

}  // namespace

TORCH_LIBRARY(bts_hip, m) {
    m.def("lpg(Tensor plane_eq, int upratio) -> (Tensor, Tensor)");
    m.def("lpg_backward(Tensor plane_eq, Tensor grad_depth, int upratio) -> Tensor");
    m.def("reduction_1x1(Tensor x2d, int c_in, int c_first_out, Tensor w_frag, float max_depth, bool is_final, bool normalize, Tensor(a!) out) -> ()");
    m.def("reduc_lpg(Tensor x2d, int B, int h, int w, int c_in, int c_first_out, Tensor w_frag, float max_depth, int upratio, "
          "Tensor(a!) depth_scaled, Tensor(b!)? ds_out, Tensor(c!)? abs_min, Tensor(d!)? plane4) -> ()");
    m.def("conv_fwd(Tensor x, Tensor w, Tensor? pre_scale, Tensor? pre_shift, Tensor? e1_scale, Tensor? e1_shift, Tensor? e2_scale, "
          "Tensor? e2_shift, Tensor(a!) y, Tensor(b!)? y2, Tensor? res, Tensor(c!)? splitk_ws, Tensor[] tail_planes, Tensor? w_split, "
          "Tensor? w_wino, int[] geom) -> ()");
}

TORCH_LIBRARY_IMPL(bts_hip, CUDA, m) {
    m.impl("lpg", &lpg);
    m.impl("lpg_backward", &lpg_backward);
    m.impl("reduction_1x1", &reduction_1x1);
    m.impl("reduc_lpg", &reduc_lpg);
    m.impl("conv_fwd", &conv_fwd);
}
