// torch_ext.cpp -- the PyTorch-ROCm extension over the C ABI of include/gsr.h (libgsr_hip.so).
//
// north_star: "Python host code calls HIP through a PyTorch-ROCm C++/HIP extension".  This file is that extension's host
// side: plain C++ (no kernels -- they live in *.hip behind the C ABI), built by torch.utils.cpp_extension into
// csrc/torch_build/gsr_torch.so and loaded with torch.ops.load_library.  It stands where the public module's
// `rasterize_points.cu / ext.cpp` pair stands (RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA / markVisible,
// called from diff_gaussian_rasterization/__init__.py as at /root/reference/scene/gaussian_model_ht.py:871-880):
//
//   gsr::rasterize            autograd-enabled entry (C++ torch::autograd::Function): what GaussianRasterizer.forward calls
//   gsr::rasterize_forward    -> gsr_forward   (buffers come from at::empty inside the allocator callback: no Python)
//   gsr::rasterize_backward   -> gsr_backward
//   gsr::rasterize_backward_fused  -> gsr_backward with the in-kernel Adam step (parameters / moments updated in place)
//   gsr::mark_visible         -> gsr_mark_visible
//   gsr::photometric_loss_forward / _backward -> gsr_loss_forward / gsr_loss_backward
//   gsr::adam_step            -> gsr_adam_step
//   gsr::pose_step            -> gsr_pose_step (stage A: tangent-space Adam + exponential map between two renders)
//   gsr::knn_mean_dist2       -> gsr_knn_mean_dist2
//
// Registered with TORCH_LIBRARY so the ops are visible to the dispatcher (torch.ops.gsr.*, fake kernels in _ext.py for
// torch.compile).  An empty tensor (numel 0) stands for "None".  Everything runs on the current HIP stream of the
// inputs' device under a device guard.  No CPU kernels are registered: CPU tensors fail in the dispatcher.
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // a ROCm build of torch presents its HIP devices under the "cuda"
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>      // device type: these are its guard and stream accessors for them
#include <torch/library.h>
#include <torch/torch.h>

#include <cmath>
#include <mutex>
#include <vector>

#include "../../include/gsr.h"

namespace {

using at::Tensor;

inline const float* fp(const Tensor& t) { return (t.defined() && t.numel() > 0) ? t.data_ptr<float>() : nullptr; }
inline float* fpm(Tensor& t) { return (t.defined() && t.numel() > 0) ? t.data_ptr<float>() : nullptr; }
inline bool has(const Tensor& t) { return t.defined() && t.numel() > 0; }

inline Tensor f32c(const Tensor& t)
{
    if (!has(t)) return t;
    return t.to(at::kFloat).contiguous();   // no-ops when already float32 + contiguous
}

void check(int rc, const char* what)
{
    TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", gsr_last_error());
}

// GsrBatch from an op's `int[] batch_first_block` (B + 1 block offsets; fewer than three entries = a single model)
struct BatchArg {
    std::vector<int32_t> fb;
    GsrBatch b{};
    explicit BatchArg(at::IntArrayRef first_block)
    {
        if (first_block.size() >= 3) {
            fb.assign(first_block.begin(), first_block.end());
            b.B = (int32_t)fb.size() - 1;
            b.first_block = fb.data();
        }
    }
    const GsrBatch* ptr() const { return b.B > 1 ? &b : nullptr; }
    int64_t B() const { return b.B > 1 ? b.B : 1; }
};

struct AllocCtx {
    at::TensorOptions opts;
    Tensor binning;
    std::vector<Tensor> scratch;
};
void* alloc_cb(size_t bytes, int tag, void* user)
{
    AllocCtx* c = static_cast<AllocCtx*>(user);
    try {
        Tensor t = at::empty({(int64_t)bytes}, c->opts);
        if (tag == GSR_ALLOC_BINNING) c->binning = t;
        else c->scratch.push_back(t);
        return t.data_ptr();
    } catch (...) {
        return nullptr;   // out of memory -> GSR_ERR_ALLOC
    }
}

// diagnostics of the most recent forward of this process (bench.py reads R / R_eff from it, the binning test the list):
// {image workspace, binning, meta, [W, H]}
std::mutex g_last_mutex;
std::vector<Tensor> g_last;

std::vector<Tensor> debug_last()
{
    std::lock_guard<std::mutex> lk(g_last_mutex);
    return g_last;
}

// ---------------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_forward(
    const Tensor& means3D_, const Tensor& sh_, const Tensor& colors_, const Tensor& opacities_, const Tensor& scales_,
    const Tensor& rotations_, const Tensor& cov3D_, const Tensor& sh_rest_, const Tensor& viewmatrix_, const Tensor& projmatrix_,
    const Tensor& campos_, const Tensor& bg_, const Tensor& xf_, int64_t H, int64_t W, double tanfovx, double tanfovy,
    double scale_modifier, int64_t sh_degree, bool raw_params, bool prefiltered, bool debug, const Tensor& prepared,
    at::IntArrayRef batch_first_block, int64_t view_id, int64_t extras)
{
    TORCH_CHECK(means3D_.is_cuda(), "GaussianRasterizer: tensors must be on a ROCm/HIP device (no CPU fallback)");
    const BatchArg batch(batch_first_block);
    const int64_t NB = batch.B();
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D_.device());
    const Tensor means3D = f32c(means3D_), sh = f32c(sh_), colors = f32c(colors_), opac = f32c(opacities_), scales = f32c(scales_),
                 rots = f32c(rotations_), cov = f32c(cov3D_), rest = f32c(sh_rest_), vm = f32c(viewmatrix_), pm = f32c(projmatrix_),
                 campos = f32c(campos_), bg = f32c(bg_), xf = f32c(xf_);
    const int64_t N = means3D.size(0);
    const int64_t M = has(sh) ? sh.size(1) + (has(rest) ? rest.size(1) : 0) : 0;
    const auto fo = means3D.options().dtype(at::kFloat);
    const auto bo = means3D.options().dtype(at::kByte);
    // batched render (GsrBatch): one image per model -- [B,3,H,W] / [B,1,H,W]; cameras are [B,4,4] / [B,3] / [B,3,4]
    if (NB > 1)
        TORCH_CHECK(vm.numel() == 16 * NB && pm.numel() == 16 * NB && (!has(campos) || campos.numel() == 3 * NB) && (!has(xf) || xf.numel() == 12 * NB),
                    "batch: viewmatrix / projmatrix must be [B,4,4], campos [B,3], points_transform [B,3,4]");
    Tensor color = NB > 1 ? at::empty({NB, 3, H, W}, fo) : at::empty({3, H, W}, fo);
    Tensor depth = NB > 1 ? at::empty({NB, 1, H, W}, fo) : at::empty({1, H, W}, fo);
    Tensor alpha = NB > 1 ? at::empty({NB, 1, H, W}, fo) : at::empty({1, H, W}, fo);
    // (prepared: the radii already sit in the hand-over buffer -- an int32 view of it, no copy)
    Tensor radii = has(prepared) ? prepared.slice(0, (int64_t)gsr_prepared_radii_offset((int32_t)N), (int64_t)gsr_prepared_radii_offset((int32_t)N) + 4 * N).view(at::kInt)
                                 : at::empty({N}, means3D.options().dtype(at::kInt));
    // "prepare in backward": the preceding backward already produced this render's splat records (+ keys, tile records) in
    // `prepared`; that buffer then IS the geometry workspace and the preprocess kernel is skipped
    if (has(prepared))
        TORCH_CHECK(prepared.is_contiguous() && prepared.scalar_type() == at::kByte &&
                        prepared.numel() == (int64_t)gsr_prepared_bytes((int32_t)N), "prepared buffer does not belong to this model");
    Tensor geom = has(prepared) ? prepared : at::empty({(int64_t)gsr_geom_bytes((int32_t)N)}, bo);
    Tensor image = at::empty({(int64_t)gsr_image_bytes_batched((int32_t)W, (int32_t)H, (int32_t)NB)}, bo);
    AllocCtx actx{bo, Tensor(), {}};

    GsrForwardArgs a{};
    a.N = (int32_t)N; a.M = (int32_t)M; a.D = (int32_t)sh_degree; a.W = (int32_t)W; a.H = (int32_t)H;
    a.prefiltered = prefiltered; a.debug = debug;
    a.scale_modifier = (float)scale_modifier; a.tanfovx = (float)tanfovx; a.tanfovy = (float)tanfovy;
    a.means3D = fp(means3D); a.scales = fp(scales); a.rotations = fp(rots); a.cov3D_precomp = fp(cov);
    a.opacities = fp(opac); a.shs = fp(sh); a.colors_precomp = fp(colors);
    a.viewmatrix = fp(vm); a.projmatrix = fp(pm); a.campos = fp(campos); a.bg = fp(bg);
    a.out_color = color.data_ptr<float>(); a.out_depth = depth.data_ptr<float>(); a.out_alpha = alpha.data_ptr<float>();
    a.radii = N ? radii.data_ptr<int32_t>() : nullptr;
    a.geom = geom.data_ptr(); a.image = image.data_ptr();
    a.alloc = alloc_cb; a.alloc_user = &actx;
    a.shs_rest = fp(rest); a.raw_params = raw_params;
    a.points_transform = fp(xf);
    a.prepared = has(prepared) ? prepared.data_ptr() : nullptr;
    a.batch = batch.ptr();
    a.view_id = view_id;
    // extras: bit 0 = the clamped colour image (written by the blend kernel), bit 1 = the visibility bytes radii > 0 (written by the
    // preprocess; not available from a prepared buffer) -- what the reference's wrapper derives with one torch launch each
    Tensor clamped = (extras & 1) ? at::empty_like(color) : at::empty({0}, fo);
    Tensor visible = ((extras & 2) && !has(prepared)) ? at::empty({N}, bo) : at::empty({0}, bo);
    a.out_color_clamped = has(clamped) ? clamped.data_ptr<float>() : nullptr;
    a.visible = has(visible) ? visible.data_ptr<uint8_t>() : nullptr;
    GsrForwardOut out{};
    check(gsr_forward(&a, &out, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_forward");
    // (scratch tensors die here: stream-ordered reuse by the caching allocator is safe, same stream)
    Tensor meta = at::empty({3}, at::TensorOptions().dtype(at::kLong));   // CPU: R, capacity, flags
    int64_t* mp = meta.data_ptr<int64_t>();
    mp[0] = out.num_rendered; mp[1] = out.binning_capacity; mp[2] = out.forward_flags;
    Tensor binning = actx.binning.defined() ? actx.binning : at::empty({0}, bo);
    {
        Tensor dims = at::empty({3}, at::TensorOptions().dtype(at::kLong));
        dims.data_ptr<int64_t>()[0] = W; dims.data_ptr<int64_t>()[1] = H; dims.data_ptr<int64_t>()[2] = NB;
        std::lock_guard<std::mutex> lk(g_last_mutex);
        g_last = {image, binning, meta, dims};
    }
    return {color, radii, depth, alpha, geom, image, binning, meta, clamped, visible};
}

struct BwdCommon {
    Tensor means3D, sh, colors, opac, scales, rots, cov, rest, vm, pm, campos, bg, xf, gc, gd, ga;
};

void fill_backward_args(GsrBackwardArgs& a, const BwdCommon& b, const Tensor& geom, const Tensor& image, const Tensor& binning,
                        const Tensor& meta, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier,
                        int64_t sh_degree, bool raw_params)
{
    const int64_t N = b.means3D.size(0);
    const int64_t M = has(b.sh) ? b.sh.size(1) + (has(b.rest) ? b.rest.size(1) : 0) : 0;
    const int64_t* mp = meta.data_ptr<int64_t>();
    a.N = (int32_t)N; a.M = (int32_t)M; a.D = (int32_t)sh_degree; a.W = (int32_t)W; a.H = (int32_t)H;
    a.scale_modifier = (float)scale_modifier; a.tanfovx = (float)tanfovx; a.tanfovy = (float)tanfovy;
    a.means3D = fp(b.means3D); a.scales = fp(b.scales); a.rotations = fp(b.rots); a.cov3D_precomp = fp(b.cov);
    a.opacities = fp(b.opac); a.shs = fp(b.sh); a.colors_precomp = fp(b.colors);
    a.viewmatrix = fp(b.vm); a.projmatrix = fp(b.pm); a.campos = fp(b.campos); a.bg = fp(b.bg);
    a.geom = geom.data_ptr(); a.image = image.data_ptr(); a.binning = has(binning) ? binning.data_ptr() : nullptr;
    a.num_rendered = mp[0]; a.binning_capacity = mp[1]; a.forward_flags = mp[2];
    a.grad_color = fp(b.gc); a.grad_depth = fp(b.gd); a.grad_alpha = fp(b.ga);
    a.shs_rest = fp(b.rest); a.raw_params = raw_params;
    a.points_transform = fp(b.xf);
}

// densification statistics accumulated by the backward kernel (GsrDensifyStats): stats = {xyz_gradient_accum, denom, max_radii2D}
// (N float32 elements each, updated in place), radii = the forward's int32 output.  Empty list = none.
bool fill_densify(GsrDensifyStats& ds, at::TensorList stats, const Tensor& radii, int64_t N)
{
    if (stats.empty()) return false;
    TORCH_CHECK(stats.size() == 3 && has(radii) && radii.scalar_type() == at::kInt && radii.numel() == N && radii.is_contiguous(),
                "densify_stats: three statistics tensors and the forward's radii expected");
    for (const Tensor& t : stats)
        TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous() && t.numel() == N, "densify_stats: contiguous float32 tensors of N elements");
    ds.radii = radii.data_ptr<int32_t>();
    ds.xyz_gradient_accum = stats[0].data_ptr<float>(); ds.denom = stats[1].data_ptr<float>(); ds.max_radii2D = stats[2].data_ptr<float>();
    return true;
}

// returns {d_means3D, d_means2D, d_sh, d_colors, d_opac, d_scales, d_rot, d_cov, d_sh_rest, d_vm, d_pm, d_campos, d_xf}
std::vector<Tensor> rasterize_backward(
    const Tensor& means3D, const Tensor& sh, const Tensor& colors, const Tensor& opac, const Tensor& scales, const Tensor& rots,
    const Tensor& cov, const Tensor& rest, const Tensor& vm, const Tensor& pm, const Tensor& campos, const Tensor& bg, const Tensor& xf,
    const Tensor& geom, const Tensor& image, const Tensor& binning, const Tensor& meta, const Tensor& grad_color, const Tensor& grad_depth,
    const Tensor& grad_alpha, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier, int64_t sh_degree,
    bool raw_params, bool need_vm, bool need_pm, bool need_campos, bool need_xf, at::TensorList densify_stats, const Tensor& radii,
    at::IntArrayRef batch_first_block)
{
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
    const BatchArg batch(batch_first_block);
    const int64_t NB = batch.B();
    BwdCommon b{means3D, sh, colors, opac, scales, rots, cov, rest, vm, pm, campos, bg, xf, f32c(grad_color), f32c(grad_depth), f32c(grad_alpha)};
    const int64_t N = means3D.size(0);
    const int64_t M = has(sh) ? sh.size(1) + (has(rest) ? rest.size(1) : 0) : 0;
    const auto fo = means3D.options().dtype(at::kFloat);
    Tensor none;
    Tensor d_means3D = at::empty({N, 3}, fo), d_means2D = at::empty({N, 3}, fo), d_opac = at::empty({N, 1}, fo);
    Tensor d_sh = has(sh) ? at::empty({N, has(rest) ? 1 : M, 3}, fo) : none;
    Tensor d_rest = (has(sh) && has(rest)) ? at::empty({N, M - 1, 3}, fo) : none;
    Tensor d_col = has(colors) ? at::empty({N, 3}, fo) : none;
    Tensor d_scales = has(scales) ? at::empty({N, 3}, fo) : none, d_rot = has(scales) ? at::empty({N, 4}, fo) : none;
    Tensor d_cov = has(cov) ? at::empty({N, 6}, fo) : none;
    auto cam_shape = [&](std::vector<int64_t> sh) { if (NB > 1) sh.insert(sh.begin(), NB); return sh; };
    Tensor d_vm = need_vm ? at::empty(cam_shape({4, 4}), fo) : none, d_pm = need_pm ? at::empty(cam_shape({4, 4}), fo) : none;
    Tensor d_cp = need_campos ? at::empty(cam_shape({3}), fo) : none;
    Tensor d_xf = (need_xf && has(xf)) ? at::zeros(cam_shape({3, 4}), fo) : none;
    Tensor scratch = at::empty({(int64_t)gsr_backward_scratch_bytes((int32_t)N)}, means3D.options().dtype(at::kByte));
    GsrBackwardArgs a{};
    fill_backward_args(a, b, geom, image, binning, meta, H, W, tanfovx, tanfovy, scale_modifier, sh_degree, raw_params);
    a.batch = batch.ptr();
    a.d_means3D = fpm(d_means3D); a.d_means2D = fpm(d_means2D); a.d_opacities = fpm(d_opac);
    a.d_colors_precomp = fpm(d_col); a.d_shs = fpm(d_sh); a.d_scales = fpm(d_scales); a.d_rotations = fpm(d_rot);
    a.d_cov3D_precomp = fpm(d_cov); a.d_shs_rest = fpm(d_rest);
    a.d_viewmatrix = fpm(d_vm); a.d_projmatrix = fpm(d_pm); a.d_campos = fpm(d_cp); a.d_points_transform = fpm(d_xf);
    a.scratch = scratch.data_ptr();
    GsrDensifyStats dstat{};
    if (fill_densify(dstat, densify_stats, radii, N)) a.densify_stats = &dstat;
    check(gsr_backward(&a, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_backward");
    return {d_means3D, d_means2D, d_sh, d_col, d_opac, d_scales, d_rot, d_cov, d_rest, d_vm, d_pm, d_cp, d_xf};
}

// Optimizer-in-backward (GsrFusedAdam): parameters (xyz, f_dc, f_rest, opacity, scaling, rotation) and their moments are
// updated in place; returns {d_means2D, d_vm, d_pm, d_campos, d_xf}.
std::vector<Tensor> rasterize_backward_fused(
    Tensor means3D, Tensor sh, Tensor rest, Tensor opac, Tensor scales, Tensor rots, const Tensor& vm, const Tensor& pm, const Tensor& campos,
    const Tensor& bg, const Tensor& xf, const Tensor& geom, const Tensor& image, const Tensor& binning, const Tensor& meta,
    const Tensor& grad_color, const Tensor& grad_depth, const Tensor& grad_alpha, int64_t H, int64_t W, double tanfovx, double tanfovy,
    double scale_modifier, int64_t sh_degree, bool need_vm, bool need_pm, bool need_campos, bool need_xf, at::TensorList adam_m,
    at::TensorList adam_v, at::ArrayRef<double> adam_lr, double beta1, double beta2, double eps, int64_t step, const Tensor& next_vm,
    const Tensor& next_pm, const Tensor& next_campos, int64_t next_H, int64_t next_W, double next_tanfovx, double next_tanfovy,
    Tensor prepared_out, const Tensor& next_xf, int64_t next_sh_degree, at::TensorList densify_stats, const Tensor& radii,
    at::IntArrayRef batch_first_block)
{
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
    const BatchArg batch(batch_first_block);
    const int64_t NB = batch.B();
    // adam_lr: six learning rates, optionally followed by six step lags (GsrFusedAdam::step_lag; whole numbers)
    // deferred application (GsrFusedAdam::param_out): adam_m = six moments + six moment outputs + six parameter outputs, adam_v = six + six
    const bool deferred = adam_m.size() == 18 && adam_v.size() == 12;
    TORCH_CHECK(((adam_m.size() == 6 && adam_v.size() == 6) || deferred) && (adam_lr.size() == 6 || adam_lr.size() == 12), "fused_adam: six groups expected");
    Tensor none;
    BwdCommon b{means3D, sh, none, opac, scales, rots, none, rest, vm, pm, campos, bg, xf, f32c(grad_color), f32c(grad_depth), f32c(grad_alpha)};
    const int64_t N = means3D.size(0);
    const auto fo = means3D.options().dtype(at::kFloat);
    Tensor d_means2D = at::empty({N, 3}, fo);
    auto cam_shape = [&](std::vector<int64_t> sh) { if (NB > 1) sh.insert(sh.begin(), NB); return sh; };
    Tensor d_vm = need_vm ? at::empty(cam_shape({4, 4}), fo) : none, d_pm = need_pm ? at::empty(cam_shape({4, 4}), fo) : none;
    Tensor d_cp = need_campos ? at::empty(cam_shape({3}), fo) : none;
    Tensor d_xf = (need_xf && has(xf)) ? at::zeros(cam_shape({3, 4}), fo) : none;
    Tensor scratch = at::empty({(int64_t)gsr_backward_scratch_bytes((int32_t)N)}, means3D.options().dtype(at::kByte));
    GsrFusedAdam fa{};
    fa.beta1 = (float)beta1; fa.beta2 = (float)beta2; fa.eps = (float)eps; fa.step = step;
    for (int q = 0; q < 6; q++) {
        TORCH_CHECK(adam_m[q].is_contiguous() && adam_v[q].is_contiguous() && adam_m[q].scalar_type() == at::kFloat, "fused_adam: moments must be contiguous float32");
        fa.lr[q] = (float)adam_lr[q];
        fa.step_lag[q] = adam_lr.size() == 12 ? (int32_t)adam_lr[6 + q] : 0;
        fa.exp_avg[q] = adam_m[q].numel() ? adam_m[q].data_ptr<float>() : nullptr;     // (empty = the group is skipped: GsrFusedAdam)
        fa.exp_avg_sq[q] = adam_v[q].numel() ? adam_v[q].data_ptr<float>() : nullptr;
        if (deferred && adam_m[q].numel()) {
            TORCH_CHECK(adam_m[6 + q].is_contiguous() && adam_v[6 + q].is_contiguous() && adam_m[12 + q].is_contiguous() &&
                        adam_m[6 + q].numel() == adam_m[q].numel() && adam_v[6 + q].numel() == adam_m[q].numel() && adam_m[12 + q].numel() == adam_m[q].numel(),
                        "fused_adam (deferred): output buffers must be contiguous and shaped like their group");
            fa.exp_avg_out[q] = adam_m[6 + q].data_ptr<float>();
            fa.exp_avg_sq_out[q] = adam_v[6 + q].data_ptr<float>();
            fa.param_out[q] = adam_m[12 + q].data_ptr<float>();
        }
    }
    GsrBackwardArgs a{};
    fill_backward_args(a, b, geom, image, binning, meta, H, W, tanfovx, tanfovy, scale_modifier, sh_degree, true);
    a.d_means2D = fpm(d_means2D);
    a.d_viewmatrix = fpm(d_vm); a.d_projmatrix = fpm(d_pm); a.d_campos = fpm(d_cp); a.d_points_transform = fpm(d_xf);
    a.scratch = scratch.data_ptr();
    a.fused_adam = &fa;
    a.batch = batch.ptr();
    GsrNextView nv{};
    // (the next render's own pose transform when its frame has one; otherwise it shares this render's)
    const Tensor nvm = f32c(next_vm), npm = f32c(next_pm), ncp = f32c(next_campos),
                 nxf = has(next_xf) ? f32c(NB > 1 ? next_xf : next_xf.slice(0, 0, 3)) : xf;
    if (has(prepared_out) && NB > 1)
        TORCH_CHECK(nvm.numel() == 16 * NB && npm.numel() == 16 * NB && ncp.numel() == 3 * NB, "batch: the next view's cameras must be [B,4,4] / [B,3]");
    if (has(prepared_out)) {   // "prepare in backward": this kernel also runs the NEXT render's preprocess on the updated parameters
        nv.W = (int32_t)next_W; nv.H = (int32_t)next_H; nv.D = (int32_t)(next_sh_degree >= 0 ? next_sh_degree : sh_degree);
        nv.scale_modifier = (float)scale_modifier; nv.tanfovx = (float)next_tanfovx; nv.tanfovy = (float)next_tanfovy;
        nv.viewmatrix = fp(nvm); nv.projmatrix = fp(npm); nv.campos = fp(ncp); nv.points_transform = fp(nxf);
        a.next_view = &nv;
        a.prepared_out = prepared_out.data_ptr();
    }
    GsrDensifyStats dstat{};
    if (fill_densify(dstat, densify_stats, radii, N)) a.densify_stats = &dstat;
    check(gsr_backward(&a, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_backward");
    return {d_means2D, d_vm, d_pm, d_cp, d_xf};
}

// ---------------------------------------------------------------------------------------------------------------
// autograd
// inputs (16 tensors): 0 means3D 1 means2D 2 sh 3 colors 4 opacities 5 scales 6 rotations 7 cov3D 8 sh_rest
//                      9 viewmatrix 10 projmatrix 11 campos 12 bg 13 points_transform
struct Cfg {
    int64_t H, W, sh_degree, adam_step;
    double tanfovx, tanfovy, scale_modifier, beta1, beta2, eps;
    bool raw_params, prefiltered, debug, cam_grad;
    std::vector<double> adam_lr;
    std::vector<Tensor> adam_m, adam_v;   // optimizer moments: plain buffers, not autograd inputs
    std::vector<int64_t> batch;           // first 128-Gaussian block of each model + the total (B + 1 entries), or empty: GsrBatch
    std::vector<Tensor> densify_stats;    // {xyz_gradient_accum, denom, max_radii2D} or empty: accumulated by the backward kernel
    Tensor adam_commit;                   // CPU int64 [1]: number of in-kernel Adam steps this optimizer's backwards have applied.  The
                                          // step count advances when a backward RUNS (a forward whose graph is dropped leaves no trace);
                                          // adam_step is the optimizer's step count at forward time, with adam_commit[0] as it was then
    Tensor prepared;                      // input: hand-over buffer of the preceding backward (or undefined)
    Tensor next_vm, next_pm, next_campos; // camera of the NEXT render (or undefined): the backward prepares it
    Tensor next_xf;                       // ... and its points_transform, when it differs from this render's (per-frame poses)
    int64_t next_H = 0, next_W = 0, next_D = -1;   // next_D: SH degree of the next render (-1 = this render's)
    int64_t view_id = 0;                           // GsrForwardArgs::view_id
    int64_t extras = 0;                            // bit 0: clamped colour output, bit 1: visibility bytes
    double next_tanfovx = 0, next_tanfovy = 0;
};

class RasterizeFn : public torch::autograd::Function<RasterizeFn> {
   public:
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& means3D, const Tensor& means2D,
                                                  const Tensor& sh, const Tensor& colors, const Tensor& opac, const Tensor& scales,
                                                  const Tensor& rots, const Tensor& cov, const Tensor& rest, const Tensor& vm,
                                                  const Tensor& pm, const Tensor& campos, const Tensor& bg, const Tensor& xf,
                                                  const Cfg& cfg)
    {
        (void)means2D;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("gsr::rasterize_forward", "").typed<decltype(rasterize_forward)>();
        // float32 + contiguous once, here: the SAME tensors are saved for the backward
        const Tensor m3 = f32c(means3D), s = f32c(sh), c = f32c(colors), o = f32c(opac), sc = f32c(scales), r = f32c(rots), cv = f32c(cov),
                     rs = f32c(rest), v = f32c(vm), p = f32c(pm), cp = f32c(campos), b = f32c(bg),
                     x = f32c((has(xf) && xf.dim() == 2) ? xf.slice(0, 0, 3) : xf);   // [4,4] -> rows 0..2; a batch hands [B,3,4]
        Tensor none;
        auto out = op.call(m3, s, c, o, sc, r, cv, rs, v, p, cp, b, x, cfg.H, cfg.W, cfg.tanfovx, cfg.tanfovy, cfg.scale_modifier,
                           cfg.sh_degree, cfg.raw_params, cfg.prefiltered, cfg.debug, cfg.prepared.defined() ? cfg.prepared : x.new_empty({0}, x.options().dtype(at::kByte)),
                           cfg.batch, cfg.view_id, cfg.extras);
        // hand-over buffer for the NEXT render, filled by this render's backward (stream-ordered): allocated here so that it
        // can be returned to the caller as an ordinary output
        Tensor prep_out = has(cfg.next_vm) ? at::empty({(int64_t)gsr_prepared_bytes((int32_t)m3.size(0))}, m3.options().dtype(at::kByte))
                                           : at::empty({0}, m3.options().dtype(at::kByte));
        // NOTE: depth is deliberately NOT saved -- the caller mutates it in place (ht3dgs_trainer.py:1290-1292)
        std::vector<Tensor> saved = {m3, s, c, o, sc, r, cv, rs, v, p, cp, b, x, std::get<4>(out), std::get<5>(out), std::get<6>(out),
                                     std::get<7>(out)};
        const Tensor clamped = std::get<8>(out), visible = std::get<9>(out);
        if (has(clamped)) saved.push_back(std::get<0>(out));   // the raw colour: where it lies in [0, 1] a gradient on the clamped image passes
        ctx->saved_data["has_clamped"] = has(clamped);
        ctx->save_for_backward(saved);
        ctx->saved_data["adam_m"] = cfg.adam_m; ctx->saved_data["adam_v"] = cfg.adam_v;
        ctx->saved_data["dens"] = cfg.densify_stats;
        ctx->saved_data["batch"] = cfg.batch;
        ctx->saved_data["radii"] = cfg.densify_stats.empty() ? Tensor(at::empty({0}, m3.options().dtype(at::kInt))) : std::get<1>(out);
        ctx->saved_data["H"] = cfg.H; ctx->saved_data["W"] = cfg.W; ctx->saved_data["D"] = cfg.sh_degree;
        ctx->saved_data["tfx"] = cfg.tanfovx; ctx->saved_data["tfy"] = cfg.tanfovy; ctx->saved_data["smod"] = cfg.scale_modifier;
        ctx->saved_data["raw"] = cfg.raw_params; ctx->saved_data["cam_grad"] = cfg.cam_grad;
        ctx->saved_data["n_adam"] = (int64_t)cfg.adam_m.size();
        ctx->saved_data["lr"] = cfg.adam_lr; ctx->saved_data["b1"] = cfg.beta1; ctx->saved_data["b2"] = cfg.beta2;
        ctx->saved_data["eps"] = cfg.eps; ctx->saved_data["step"] = cfg.adam_step;
        ctx->saved_data["commit"] = cfg.adam_commit.defined() ? cfg.adam_commit : at::zeros({1}, at::TensorOptions().dtype(at::kLong));
        ctx->saved_data["commit_seen"] = cfg.adam_commit.defined() ? cfg.adam_commit.data_ptr<int64_t>()[0] : (int64_t)0;
        ctx->saved_data["xf_rows"] = (has(xf) && xf.dim() == 2) ? xf.size(0) : (int64_t)0;
        ctx->saved_data["done"] = false;
        ctx->saved_data["prep_out"] = prep_out;
        ctx->saved_data["next_cam"] = std::vector<Tensor>{has(cfg.next_vm) ? f32c(cfg.next_vm) : x, has(cfg.next_vm) ? f32c(cfg.next_pm) : x,
                                                          has(cfg.next_vm) ? f32c(cfg.next_campos) : x,
                                                          has(cfg.next_xf) ? f32c(cfg.next_xf) : x.new_empty({0})};
        ctx->saved_data["next_H"] = cfg.next_H; ctx->saved_data["next_W"] = cfg.next_W;
        ctx->saved_data["next_tfx"] = cfg.next_tanfovx; ctx->saved_data["next_tfy"] = cfg.next_tanfovy;
        ctx->saved_data["next_D"] = cfg.next_D;
        ctx->mark_non_differentiable({std::get<1>(out), prep_out, visible});
        ctx->set_materialize_grads(false);   // unused depth / alpha outputs arrive undefined -> specialised backward
        return {std::get<0>(out), std::get<1>(out), std::get<2>(out), std::get<3>(out), prep_out, clamped, visible};
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list g)
    {
        auto sv = ctx->get_saved_variables();
        const int64_t n_adam = ctx->saved_data["n_adam"].toInt();
        Tensor gc = g[0];
        const Tensor &gd = g[2], &ga = g[3];
        torch::autograd::variable_list out(15);   // 14 tensor inputs + the Cfg argument
        if (g.size() > 5 && g[5].defined() && ctx->saved_data["has_clamped"].toBool()) {
            // a gradient arrived on the CLAMPED image (a consumer that did not take the fused-clamp route): torch.clamp's rule
            const Tensor raw = sv.back();
            const Tensor pass = g[5] * raw.ge(0).logical_and(raw.le(1)).to(g[5].scalar_type());
            gc = gc.defined() ? gc + pass : pass;
        }
        if (!gc.defined() && !gd.defined() && !ga.defined()) return out;
        const int64_t H = ctx->saved_data["H"].toInt(), W = ctx->saved_data["W"].toInt(), D = ctx->saved_data["D"].toInt();
        const double tfx = ctx->saved_data["tfx"].toDouble(), tfy = ctx->saved_data["tfy"].toDouble(), smod = ctx->saved_data["smod"].toDouble();
        const bool raw = ctx->saved_data["raw"].toBool(), cam = ctx->saved_data["cam_grad"].toBool();
        const bool need_vm = cam && ctx->needs_input_grad(9), need_pm = cam && ctx->needs_input_grad(10), need_cp = cam && ctx->needs_input_grad(11);
        const bool need_xf = ctx->needs_input_grad(13);
        const int64_t xf_rows = ctx->saved_data["xf_rows"].toInt();
        Tensor e;
        auto orE = [&](const Tensor& t) { return t.defined() ? t : e; };
        Tensor d_xf;
        std::vector<Tensor> dens = ctx->saved_data["dens"].toTensorVector();
        const std::vector<int64_t> bfb = ctx->saved_data["batch"].toIntVector();
        const Tensor radii = ctx->saved_data["radii"].toTensor();
        if (n_adam) {
            // a second backward through the same forward would apply the optimizer step twice
            TORCH_CHECK(!ctx->saved_data["done"].toBool(), "fused_adam: backward() ran twice on the same render (retain_graph); the in-kernel "
                                                           "Adam step can be applied once per forward");
            ctx->saved_data["done"] = true;
            static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("gsr::rasterize_backward_fused", "").typed<decltype(rasterize_backward_fused)>();
            std::vector<Tensor> m = ctx->saved_data["adam_m"].toTensorVector(), v = ctx->saved_data["adam_v"].toTensorVector();
            auto lr = ctx->saved_data["lr"].toDoubleVector();
            auto nc = ctx->saved_data["next_cam"].toTensorVector();
            // the 1-based step of THIS update: the optimizer's count at forward time + the in-kernel steps applied since + 1
            Tensor commit = ctx->saved_data["commit"].toTensor();
            int64_t* commit_p = commit.data_ptr<int64_t>();
            const int64_t step_now = ctx->saved_data["step"].toInt() + (commit_p[0] - ctx->saved_data["commit_seen"].toInt()) + 1;
            auto r = op.call(sv[0], sv[1], sv[7], sv[3], sv[4], sv[5], sv[8], sv[9], sv[10], sv[11], sv[12], sv[13], sv[14], sv[15], sv[16],
                             orE(gc), orE(gd), orE(ga), H, W, tfx, tfy, smod, D, need_vm, need_pm, need_cp, need_xf, m, v, lr,
                             ctx->saved_data["b1"].toDouble(), ctx->saved_data["b2"].toDouble(), ctx->saved_data["eps"].toDouble(),
                             step_now, nc[0], nc[1], nc[2], ctx->saved_data["next_H"].toInt(),
                             ctx->saved_data["next_W"].toInt(), ctx->saved_data["next_tfx"].toDouble(), ctx->saved_data["next_tfy"].toDouble(),
                             ctx->saved_data["prep_out"].toTensor(), nc[3], ctx->saved_data["next_D"].toInt(), dens, radii, bfb);
            commit_p[0] += 1;   // the update has been enqueued: the optimizer's step count advances (FusedAdam reconciles from this)
            out[1] = r[0]; out[9] = r[1]; out[10] = r[2]; out[11] = r[3]; d_xf = r[4];
        } else {
            static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("gsr::rasterize_backward", "").typed<decltype(rasterize_backward)>();
            auto r = op.call(sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7], sv[8], sv[9], sv[10], sv[11], sv[12], sv[13], sv[14],
                             sv[15], sv[16], orE(gc), orE(gd), orE(ga), H, W, tfx, tfy, smod, D, raw, need_vm, need_pm, need_cp, need_xf, dens, radii, bfb);
            out[0] = r[0]; out[1] = r[1]; out[2] = r[2]; out[3] = r[3]; out[4] = r[4]; out[5] = r[5]; out[6] = r[6]; out[7] = r[7]; out[8] = r[8];
            out[9] = r[9]; out[10] = r[10]; out[11] = r[11]; d_xf = r[12];
        }
        if (d_xf.defined() && xf_rows == 4) d_xf = at::cat({d_xf, at::zeros({1, 4}, d_xf.options())}, 0);   // a [4,4] input keeps a zero last row
        out[13] = d_xf;
        return out;
    }
};

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize(
    const Tensor& means3D, const Tensor& means2D, const Tensor& sh, const Tensor& colors, const Tensor& opac, const Tensor& scales,
    const Tensor& rots, const Tensor& cov, const Tensor& rest, const Tensor& vm, const Tensor& pm, const Tensor& campos, const Tensor& bg,
    const Tensor& xf, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier, int64_t sh_degree, bool raw_params,
    bool prefiltered, bool debug, bool cam_grad, at::TensorList adam_m, at::TensorList adam_v, at::ArrayRef<double> adam_lr, double beta1,
    double beta2, double eps, int64_t step, const Tensor& prepared, const Tensor& next_vm, const Tensor& next_pm, const Tensor& next_campos,
    int64_t next_H, int64_t next_W, double next_tanfovx, double next_tanfovy, const Tensor& next_xf, int64_t next_sh_degree,
    const Tensor& adam_commit, at::TensorList densify_stats, at::IntArrayRef batch_first_block, int64_t view_id, int64_t extras)
{
    Cfg cfg{H, W, sh_degree, step, tanfovx, tanfovy, scale_modifier, beta1, beta2, eps, raw_params, prefiltered, debug, cam_grad,
            std::vector<double>(adam_lr.begin(), adam_lr.end()), adam_m.vec(), adam_v.vec()};
    if (has(prepared)) cfg.prepared = prepared;
    cfg.densify_stats = densify_stats.vec();
    cfg.batch.assign(batch_first_block.begin(), batch_first_block.end());
    cfg.view_id = view_id;
    cfg.extras = extras;
    if (!adam_m.empty()) {
        TORCH_CHECK(has(adam_commit) && adam_commit.is_cpu() && adam_commit.scalar_type() == at::kLong, "fused_adam: adam_commit must be a CPU int64 tensor");
        cfg.adam_commit = adam_commit;
    }
    if (has(next_vm)) {
        TORCH_CHECK(!adam_m.empty(), "prepare_next needs fused_adam (the backward that applies the update prepares the next render)");
        cfg.next_vm = next_vm; cfg.next_pm = next_pm; cfg.next_campos = next_campos;
        cfg.next_H = next_H; cfg.next_W = next_W; cfg.next_tanfovx = next_tanfovx; cfg.next_tanfovy = next_tanfovy;
        cfg.next_D = next_sh_degree;
        if (has(next_xf)) cfg.next_xf = next_xf;
    }
    auto r = RasterizeFn::apply(means3D, means2D, sh, colors, opac, scales, rots, cov, rest, vm, pm, campos, bg, xf, cfg);
    return {r[0], r[1], r[2], r[3], r[4], r[5], r[6]};
}

// tensors without an autograd key (torch.inference_mode): the forward alone
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_forward_only(
    const Tensor& means3D, const Tensor& means2D, const Tensor& sh, const Tensor& colors, const Tensor& opac, const Tensor& scales,
    const Tensor& rots, const Tensor& cov, const Tensor& rest, const Tensor& vm, const Tensor& pm, const Tensor& campos, const Tensor& bg,
    const Tensor& xf, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier, int64_t sh_degree, bool raw_params,
    bool prefiltered, bool debug, bool cam_grad, at::TensorList adam_m, at::TensorList adam_v, at::ArrayRef<double> adam_lr, double beta1,
    double beta2, double eps, int64_t step, const Tensor& prepared, const Tensor& next_vm, const Tensor& next_pm, const Tensor& next_campos,
    int64_t next_H, int64_t next_W, double next_tanfovx, double next_tanfovy, const Tensor& next_xf, int64_t next_sh_degree,
    const Tensor& adam_commit, at::TensorList densify_stats, at::IntArrayRef batch_first_block, int64_t view_id, int64_t extras)
{
    (void)means2D; (void)next_xf; (void)next_sh_degree; (void)adam_commit; (void)densify_stats; (void)cam_grad; (void)adam_m; (void)adam_v; (void)adam_lr; (void)beta1; (void)beta2; (void)eps; (void)step;
    (void)next_vm; (void)next_pm; (void)next_campos; (void)next_H; (void)next_W; (void)next_tanfovx; (void)next_tanfovy;
    auto out = rasterize_forward(means3D, sh, colors, opac, scales, rots, cov, rest, vm, pm, campos, bg, (has(xf) && xf.dim() == 2) ? xf.slice(0, 0, 3) : xf, H, W,
                                 tanfovx, tanfovy, scale_modifier, sh_degree, raw_params, prefiltered, debug, prepared, batch_first_block, view_id, extras);
    return {std::get<0>(out), std::get<1>(out), std::get<2>(out), std::get<3>(out), at::empty({0}, means3D.options().dtype(at::kByte)), std::get<8>(out),
            std::get<9>(out)};
}

// ---------------------------------------------------------------------------------------------------------------
Tensor mark_visible(const Tensor& means3D_, const Tensor& vm_, const Tensor& pm_)
{
    TORCH_CHECK(means3D_.is_cuda(), "markVisible: tensors must be on a ROCm/HIP device (no CPU fallback)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D_.device());
    const Tensor p = f32c(means3D_), vm = f32c(vm_), pm = f32c(pm_);
    Tensor present = at::empty({p.size(0)}, p.options().dtype(at::kByte));
    check(gsr_mark_visible((int32_t)p.size(0), fp(p), fp(vm), fp(pm), p.size(0) ? present.data_ptr<uint8_t>() : nullptr,
                           c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_mark_visible");
    return present.to(at::kBool);
}

// render / target: [C,H,W], or a stack [B,C,H,W] of independent images (a batched render): then the loss is the SUM of the images'
// losses, each normalised by its own C H W, and every image receives the gradient of its own loss
std::tuple<Tensor, Tensor> photometric_loss_forward(const Tensor& render_, const Tensor& target_, double lambda_dssim, bool clamp)
{
    TORCH_CHECK(render_.is_cuda(), "fused_photometric_loss: tensors must be on a ROCm/HIP device (no CPU fallback)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(render_.device());
    const Tensor render = f32c(render_), target = f32c(target_);
    TORCH_CHECK((render.dim() == 3 || render.dim() == 4) && render.sizes() == target.sizes(), "fused_photometric_loss: [C,H,W] or [B,C,H,W] render and target of one shape");
    const int o = render.dim() == 4 ? 1 : 0;
    const int32_t B = o ? (int32_t)render.size(0) : 1, C = (int32_t)render.size(o), H = (int32_t)render.size(o + 1), W = (int32_t)render.size(o + 2);
    Tensor ws = at::empty({(int64_t)gsr_loss_workspace_bytes_batched(B, C, H, W)}, render.options().dtype(at::kByte));
    Tensor out = at::empty({3}, render.options());
    check(gsr_loss_forward_batched(fp(render), fp(target), B, C, H, W, (float)lambda_dssim, clamp, ws.data_ptr(), out.data_ptr<float>(),
                                   c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_loss_forward");
    return {out, ws};
}

// the same with the six-term result vector of gsr_loss_forward_terms (single image)
std::tuple<Tensor, Tensor, Tensor> photometric_loss_forward_terms(const Tensor& render_, const Tensor& target_, double lambda_dssim, bool clamp)
{
    TORCH_CHECK(render_.is_cuda(), "fused_photometric_loss: tensors must be on a ROCm/HIP device (no CPU fallback)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(render_.device());
    const Tensor render = f32c(render_), target = f32c(target_);
    TORCH_CHECK((render.dim() == 3 || render.dim() == 4) && render.sizes() == target.sizes(), "fused_photometric_loss: [C,H,W] or [B,C,H,W] render and target of one shape");
    const int o = render.dim() == 4 ? 1 : 0;
    const int32_t B = o ? (int32_t)render.size(0) : 1, C = (int32_t)render.size(o), H = (int32_t)render.size(o + 1), W = (int32_t)render.size(o + 2);
    Tensor ws = at::empty({(int64_t)gsr_loss_workspace_bytes_batched(B, C, H, W)}, render.options().dtype(at::kByte));
    Tensor out = at::empty({6}, render.options()), loss = at::empty({}, render.options());
    check(gsr_loss_forward_terms(fp(render), fp(target), B, C, H, W, (float)lambda_dssim, clamp, ws.data_ptr(), out.data_ptr<float>(),
                                 loss.data_ptr<float>(), c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_loss_forward_terms");
    return {out, ws, loss};
}

Tensor photometric_loss_backward(const Tensor& render_, const Tensor& target_, const Tensor& ws, const Tensor& grad_loss_, double lambda_dssim,
                                 bool clamp)
{
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(render_.device());
    const Tensor render = f32c(render_), target = f32c(target_), g = f32c(grad_loss_);
    const int o = render.dim() == 4 ? 1 : 0;
    const int32_t B = o ? (int32_t)render.size(0) : 1, C = (int32_t)render.size(o), H = (int32_t)render.size(o + 1), W = (int32_t)render.size(o + 2);
    Tensor d = at::empty_like(render);
    check(gsr_loss_backward_batched(fp(render), fp(target), B, C, H, W, (float)lambda_dssim, clamp, ws.data_ptr(), fp(g), d.data_ptr<float>(),
                                    c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_loss_backward");
    return d;
}

void adam_step(at::TensorList params, at::TensorList grads, at::TensorList exp_avg, at::TensorList exp_avg_sq, at::ArrayRef<double> lr,
               double beta1, double beta2, double eps, int64_t step)
{
    if (params.empty()) return;
    TORCH_CHECK(params[0].is_cuda(), "FusedAdam: parameters must be on a ROCm/HIP device (no CPU fallback)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(params[0].device());
    std::vector<Tensor> keep;
    for (size_t lo = 0; lo < params.size(); lo += GSR_ADAM_MAX_TENSORS) {
        GsrAdamTensor arr[GSR_ADAM_MAX_TENSORS];
        const size_t n = std::min<size_t>(GSR_ADAM_MAX_TENSORS, params.size() - lo);
        for (size_t k = 0; k < n; k++) {
            const Tensor& p = params[lo + k];
            TORCH_CHECK(p.is_contiguous() && p.scalar_type() == at::kFloat, "FusedAdam: parameters must be contiguous float32");
            Tensor g = grads[lo + k].contiguous();
            keep.push_back(g);
            arr[k].param = p.data_ptr<float>(); arr[k].grad = g.data_ptr<float>();
            arr[k].exp_avg = exp_avg[lo + k].data_ptr<float>(); arr[k].exp_avg_sq = exp_avg_sq[lo + k].data_ptr<float>();
            arr[k].n = (uint64_t)p.numel(); arr[k].lr = (float)lr[lo + k];
        }
        check(gsr_adam_step(arr, (int32_t)n, (float)beta1, (float)beta2, (float)eps, step, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
              "gsr_adam_step");
    }
}

// The loss with its autograd node in C++ (one dispatcher call in the forward, no Python frame in the backward)
class PhotometricLossFn : public torch::autograd::Function<PhotometricLossFn> {
   public:
    static Tensor forward(torch::autograd::AutogradContext* ctx, const Tensor& render, const Tensor& target, double lambda_dssim, bool clamp)
    {
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("gsr::photometric_loss_forward", "").typed<decltype(photometric_loss_forward)>();
        const Tensor r = f32c(render), t = f32c(target.device() == render.device() ? target : target.to(render.device()));
        auto out = op.call(r, t, lambda_dssim, clamp);
        ctx->save_for_backward({r, t, std::get<1>(out)});
        ctx->saved_data["lam"] = lambda_dssim; ctx->saved_data["clamp"] = clamp;
        return std::get<0>(out).select(0, 0);
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list g)
    {
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("gsr::photometric_loss_backward", "").typed<decltype(photometric_loss_backward)>();
        auto sv = ctx->get_saved_variables();
        Tensor d = op.call(sv[0], sv[1], sv[2], g[0], ctx->saved_data["lam"].toDouble(), ctx->saved_data["clamp"].toBool());
        return {d, Tensor(), Tensor(), Tensor()};
    }
};
Tensor photometric_loss(const Tensor& render, const Tensor& target, double lambda_dssim, bool clamp)
{
    return PhotometricLossFn::apply(render, target, lambda_dssim, clamp);
}

// (loss, terms): `loss` is the differentiable scalar, `terms` the six-float vector {loss, mean SSIM, mean L1, loss_rgb, loss_dssim,
// loss_depth = 0} -- everything Loss.forward of the reference returns, from ONE forward; the caller slices it (views, no kernels).
// Undefined upstream gradients are not materialised: the backward sees d loss alone (no zero fills for the unused vector).
class PhotometricTermsFn : public torch::autograd::Function<PhotometricTermsFn> {
   public:
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& render, const Tensor& target, double lambda_dssim,
                                                  bool clamp)
    {
        const Tensor r = f32c(render), t = f32c(target.device() == render.device() ? target : target.to(render.device()));
        auto out = photometric_loss_forward_terms(r, t, lambda_dssim, clamp);
        ctx->save_for_backward({r, t, std::get<1>(out)});
        ctx->saved_data["lam"] = lambda_dssim; ctx->saved_data["clamp"] = clamp;
        Tensor terms = std::get<0>(out), loss = std::get<2>(out);   // (the scalar in storage of its own, written by the same kernel)
        ctx->mark_non_differentiable({terms});
        ctx->set_materialize_grads(false);
        return {loss, terms};
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list g)
    {
        if (!g[0].defined()) return {Tensor(), Tensor(), Tensor(), Tensor()};
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("gsr::photometric_loss_backward", "").typed<decltype(photometric_loss_backward)>();
        auto sv = ctx->get_saved_variables();
        Tensor d = op.call(sv[0], sv[1], sv[2], g[0], ctx->saved_data["lam"].toDouble(), ctx->saved_data["clamp"].toBool());
        return {d, Tensor(), Tensor(), Tensor()};
    }
};
std::tuple<Tensor, Tensor> photometric_loss_terms(const Tensor& render, const Tensor& target, double lambda_dssim, bool clamp)
{
    auto r = PhotometricTermsFn::apply(render, target, lambda_dssim, clamp);
    return {r[0], r[1]};
}
std::tuple<Tensor, Tensor> photometric_loss_terms_no_grad(const Tensor& render, const Tensor& target, double lambda_dssim, bool clamp)
{
    auto out = photometric_loss_forward_terms(render, target.device() == render.device() ? target : target.to(render.device()), lambda_dssim, clamp);
    return {std::get<2>(out), std::get<0>(out)};
}
Tensor photometric_loss_no_grad(const Tensor& render, const Tensor& target, double lambda_dssim, bool clamp)
{
    return std::get<0>(photometric_loss_forward(render, target.device() == render.device() ? target : target.to(render.device()), lambda_dssim, clamp)).select(0, 0);
}

// Pose step of stage A: delta / exp_avg / exp_avg_sq ([6] float32) updated in place from dL/dM, the new M = Exp(delta) * base written
// into `xf` ([3,4] or [4,4] float32, rows 0..2) -- the tensor the next render reads as points_transform.  step = 0: only evaluates M.
void pose_step(Tensor delta, Tensor exp_avg, Tensor exp_avg_sq, const Tensor& d_xf, const Tensor& base, Tensor xf, double lr,
               double beta1, double beta2, double eps, int64_t step)
{
    TORCH_CHECK(delta.is_cuda(), "pose_step: tensors must be on a ROCm/HIP device (no CPU fallback)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(delta.device());
    auto ok6 = [](const Tensor& t) { return t.is_contiguous() && t.scalar_type() == at::kFloat && t.numel() == 6; };
    TORCH_CHECK(ok6(delta) && (step == 0 || (ok6(exp_avg) && ok6(exp_avg_sq))), "pose_step: delta / moments must be contiguous float32 [6]");
    TORCH_CHECK(xf.is_contiguous() && xf.scalar_type() == at::kFloat && xf.numel() >= 12, "pose_step: xf must be contiguous float32 [3,4] or [4,4]");
    Tensor g = has(d_xf) ? f32c(d_xf) : d_xf, b = has(base) ? f32c(base) : base;
    TORCH_CHECK(step == 0 || (has(g) && g.numel() >= 12), "pose_step: d_xf must hold dL/dM (12 floats)");
    TORCH_CHECK(!has(b) || b.numel() >= 12, "pose_step: base must hold a 3x4 (or 4x4) matrix");
    check(gsr_pose_step(delta.data_ptr<float>(), step ? exp_avg.data_ptr<float>() : nullptr, step ? exp_avg_sq.data_ptr<float>() : nullptr,
                        fp(g), fp(b), xf.data_ptr<float>(), (float)lr, (float)beta1, (float)beta2, (float)eps, step,
                        c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_pose_step");
    // written through raw pointers: tell autograd's version counters (a hand-over buffer prepared for the old transform is then
    // recognised as stale, train_step._same_transform; a plain TORCH_LIBRARY op with a (a!) schema does not bump them itself)
    torch::autograd::impl::bump_version(xf);
    torch::autograd::impl::bump_version(delta);
}

// The pose matrix as an autograd node (round 6; what gsr_autopatch puts where the unmodified trainer evaluates `P[k].retr()`,
// /root/reference/scene/gaussian_model_ht.py:135-148): M = Exp(delta) * base as a [3,4] tensor from the pose's six tangent numbers,
// one one-wave kernel forward (gsr_pose_step with step 0), one backward (gsr_pose_grad: dL/dM -> dL/d(delta)); `delta` keeps its own
// shape ([6] or lietorch's [1,6]) and receives its .grad the usual way, so whichever optimizer owns it -- the FusedPoseAdam
// gsr_autopatch hands out, or a stock torch.optim.Adam -- steps it unchanged.  base: [3,4] / [4,4] or empty (identity).
Tensor pose_matrix_forward(const Tensor& delta, const Tensor& base)
{
    TORCH_CHECK(delta.is_cuda(), "pose_matrix: tensors must be on a ROCm/HIP device (no CPU fallback)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(delta.device());
    TORCH_CHECK(delta.scalar_type() == at::kFloat && delta.numel() == 6, "pose_matrix: delta must hold six float32 numbers (tau, phi)");
    const Tensor d = delta.contiguous(), b = has(base) ? f32c(base) : base;
    TORCH_CHECK(!has(b) || b.numel() >= 12, "pose_matrix: base must hold a 3x4 (or 4x4) matrix");
    Tensor xf = at::empty({3, 4}, d.options());
    check(gsr_pose_step(const_cast<float*>(d.data_ptr<float>()), nullptr, nullptr, nullptr, fp(b), xf.data_ptr<float>(), 0.f, 0.9f, 0.999f, 1e-8f, 0,
                        c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_pose_step");
    return xf;
}
Tensor pose_matrix_backward(const Tensor& delta, const Tensor& base, const Tensor& d_xf)
{
    TORCH_CHECK(delta.is_cuda(), "pose_matrix: tensors must be on a ROCm/HIP device (no CPU fallback)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(delta.device());
    const Tensor d = delta.contiguous(), b = has(base) ? f32c(base) : base, g = f32c(d_xf);
    TORCH_CHECK(d.scalar_type() == at::kFloat && d.numel() == 6 && g.numel() >= 12, "pose_matrix backward: delta [6], dL/dM [3,4]");
    Tensor out = at::empty(delta.sizes(), d.options());
    check(gsr_pose_grad(d.data_ptr<float>(), g.data_ptr<float>(), fp(b), out.data_ptr<float>(),
                        c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_pose_grad");
    return out;
}
class PoseMatrixFn : public torch::autograd::Function<PoseMatrixFn> {
   public:
    static Tensor forward(torch::autograd::AutogradContext* ctx, const Tensor& delta, const Tensor& base)
    {
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("gsr::pose_matrix_forward", "").typed<decltype(pose_matrix_forward)>();
        ctx->save_for_backward({delta, base});
        ctx->set_materialize_grads(false);
        return op.call(delta, base);
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list g)
    {
        if (!g[0].defined()) return {Tensor(), Tensor()};
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("gsr::pose_matrix_backward", "").typed<decltype(pose_matrix_backward)>();
        auto sv = ctx->get_saved_variables();
        return {op.call(sv[0], sv[1], g[0]), Tensor()};
    }
};
Tensor pose_matrix(const Tensor& delta, const Tensor& base) { return PoseMatrixFn::apply(delta, base); }

// Camera-route pose step: the frame's viewmatrix / projmatrix / campos tensors are rewritten in place from their own gradients.
void pose_step_camera(Tensor delta, Tensor exp_avg, Tensor exp_avg_sq, const Tensor& d_vm, const Tensor& d_pm, const Tensor& d_cp,
                      const Tensor& projT, const Tensor& base, Tensor vm, Tensor pm, Tensor cp, double lr, double beta1, double beta2,
                      double eps, int64_t step)
{
    TORCH_CHECK(delta.is_cuda(), "pose_step_camera: tensors must be on a ROCm/HIP device (no CPU fallback)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(delta.device());
    auto okn = [](const Tensor& t, int64_t n) { return t.is_contiguous() && t.scalar_type() == at::kFloat && t.numel() == n; };
    TORCH_CHECK(okn(delta, 6) && (step == 0 || (okn(exp_avg, 6) && okn(exp_avg_sq, 6))), "pose_step_camera: delta / moments must be contiguous float32 [6]");
    TORCH_CHECK(okn(vm, 16) && okn(pm, 16) && okn(cp, 3), "pose_step_camera: viewmatrix / projmatrix [4,4] and campos [3] must be contiguous float32");
    const Tensor gv = has(d_vm) ? f32c(d_vm) : d_vm, gp = has(d_pm) ? f32c(d_pm) : d_pm, gc = has(d_cp) ? f32c(d_cp) : d_cp;
    const Tensor pt = f32c(projT), b = has(base) ? f32c(base) : base;
    TORCH_CHECK(pt.numel() == 16 && (!has(b) || b.numel() >= 12), "pose_step_camera: projection_T must be [4,4], base [3,4] or [4,4]");
    check(gsr_pose_step_camera(delta.data_ptr<float>(), step ? exp_avg.data_ptr<float>() : nullptr, step ? exp_avg_sq.data_ptr<float>() : nullptr,
                               fp(gv), fp(gp), fp(gc), fp(pt), fp(b), vm.data_ptr<float>(), pm.data_ptr<float>(), cp.data_ptr<float>(),
                               (float)lr, (float)beta1, (float)beta2, (float)eps, step,
                               c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_pose_step_camera");
    // (see pose_step: train_step._camera_versions compares these counters)
    torch::autograd::impl::bump_version(vm);
    torch::autograd::impl::bump_version(pm);
    torch::autograd::impl::bump_version(cp);
    torch::autograd::impl::bump_version(delta);
}

Tensor knn_mean_dist2(const Tensor& points_)
{
    TORCH_CHECK(points_.is_cuda(), "distCUDA2: points must be on a ROCm/HIP device (no CPU fallback)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(points_.device());
    const Tensor p = f32c(points_);
    const int32_t N = (int32_t)p.size(0);
    Tensor out = at::empty({N}, p.options());
    const size_t sb = gsr_knn_scratch_bytes(N);
    Tensor scratch = at::empty({(int64_t)sb}, p.options().dtype(at::kByte));
    check(gsr_knn_mean_dist2(fp(p), N, N ? out.data_ptr<float>() : nullptr, scratch.data_ptr(), sb, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
          "gsr_knn_mean_dist2");
    return out;
}

}  // namespace

// ---- the trainer's per-iteration bookkeeping (gsr_masked_max / gsr_densify_stats_add / gsr_psnr; include/gsr.h) ------------------
void masked_max_(Tensor dst, const Tensor& src, const Tensor& mask)
{
    TORCH_CHECK(dst.is_cuda(), "masked_max_: tensors must be on a ROCm/HIP device (no CPU fallback)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dst.device());
    TORCH_CHECK(dst.is_contiguous() && dst.scalar_type() == at::kFloat && src.is_contiguous() && src.scalar_type() == at::kInt &&
                mask.is_contiguous() && mask.scalar_type() == at::kBool && dst.dim() == 1 && src.numel() == dst.numel() && mask.numel() == dst.numel(),
                "masked_max_: dst float32 [n], src int32 [n], mask bool [n], contiguous");
    check(gsr_masked_max(dst.data_ptr<float>(), src.data_ptr<int32_t>(), reinterpret_cast<const uint8_t*>(mask.data_ptr<bool>()), (int32_t)dst.numel(),
                         c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_masked_max");
    torch::autograd::impl::bump_version(dst);
}

void densify_stats_add_(Tensor accum, Tensor denom, const Tensor& grad, const Tensor& mask)
{
    TORCH_CHECK(accum.is_cuda(), "densify_stats_add_: tensors must be on a ROCm/HIP device (no CPU fallback)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(accum.device());
    const int64_t n = mask.numel();
    TORCH_CHECK(accum.is_contiguous() && denom.is_contiguous() && grad.is_contiguous() && mask.is_contiguous() && accum.scalar_type() == at::kFloat &&
                denom.scalar_type() == at::kFloat && grad.scalar_type() == at::kFloat && mask.scalar_type() == at::kBool && accum.numel() == n &&
                denom.numel() == n && grad.numel() == 3 * n, "densify_stats_add_: accum / denom float32 [n(,1)], grad float32 [n,3], mask bool [n], contiguous");
    check(gsr_densify_stats_add(accum.data_ptr<float>(), denom.data_ptr<float>(), grad.data_ptr<float>(),
                                reinterpret_cast<const uint8_t*>(mask.data_ptr<bool>()), (int32_t)n,
                                c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_densify_stats_add");
    torch::autograd::impl::bump_version(accum);
    torch::autograd::impl::bump_version(denom);
}

Tensor psnr(const Tensor& a, const Tensor& b)
{
    TORCH_CHECK(a.is_cuda(), "psnr: tensors must be on a ROCm/HIP device (no CPU fallback)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(a.device());
    const Tensor x = f32c(a), y = f32c(b);
    TORCH_CHECK(x.dim() >= 2 && x.sizes() == y.sizes(), "psnr: two images of the same shape [C, ...]");
    const int64_t C = x.size(0), P = x.numel() / C;
    Tensor out = at::empty({C, 1}, x.options());
    Tensor scratch = at::empty({(int64_t)gsr_psnr_scratch_bytes((int32_t)C)}, x.options().dtype(at::kByte));
    check(gsr_psnr(x.data_ptr<float>(), y.data_ptr<float>(), (int32_t)C, P, out.data_ptr<float>(), scratch.data_ptr(),
                   c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()), "gsr_psnr");
    return out;
}

TORCH_LIBRARY(gsr, m)
{
    m.def("rasterize_forward(Tensor means3D, Tensor sh, Tensor colors_precomp, Tensor opacities, Tensor scales, Tensor rotations, "
          "Tensor cov3D_precomp, Tensor sh_rest, Tensor viewmatrix, Tensor projmatrix, Tensor campos, Tensor bg, Tensor points_transform, "
          "int image_height, int image_width, float tanfovx, float tanfovy, float scale_modifier, int sh_degree, bool raw_params, "
          "bool prefiltered, bool debug, Tensor prepared, int[] batch_first_block, int view_id=0, int extras=0) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("rasterize_backward(Tensor means3D, Tensor sh, Tensor colors_precomp, Tensor opacities, Tensor scales, Tensor rotations, "
          "Tensor cov3D_precomp, Tensor sh_rest, Tensor viewmatrix, Tensor projmatrix, Tensor campos, Tensor bg, Tensor points_transform, "
          "Tensor geom, Tensor image, Tensor binning, Tensor meta, Tensor grad_color, Tensor grad_depth, Tensor grad_alpha, "
          "int image_height, int image_width, float tanfovx, float tanfovy, float scale_modifier, int sh_degree, bool raw_params, "
          "bool need_viewmatrix, bool need_projmatrix, bool need_campos, bool need_points_transform, Tensor(a!)[] densify_stats, Tensor radii, int[] batch_first_block) -> Tensor[]");
    m.def("rasterize_backward_fused(Tensor(a!) means3D, Tensor(b!) sh, Tensor(c!) sh_rest, Tensor(d!) opacities, Tensor(e!) scales, "
          "Tensor(f!) rotations, Tensor viewmatrix, Tensor projmatrix, Tensor campos, Tensor bg, Tensor points_transform, Tensor geom, "
          "Tensor image, Tensor binning, Tensor meta, Tensor grad_color, Tensor grad_depth, Tensor grad_alpha, int image_height, "
          "int image_width, float tanfovx, float tanfovy, float scale_modifier, int sh_degree, bool need_viewmatrix, bool need_projmatrix, "
          "bool need_campos, bool need_points_transform, Tensor(g!)[] adam_m, Tensor(h!)[] adam_v, float[] adam_lr, float beta1, "
          "float beta2, float eps, int step, Tensor next_viewmatrix, Tensor next_projmatrix, Tensor next_campos, int next_height, "
          "int next_width, float next_tanfovx, float next_tanfovy, Tensor(i!) prepared_out, Tensor next_points_transform, int next_sh_degree, Tensor(j!)[] densify_stats, Tensor radii, int[] batch_first_block) -> Tensor[]");
    m.def("rasterize(Tensor means3D, Tensor means2D, Tensor sh, Tensor colors_precomp, Tensor opacities, Tensor scales, Tensor rotations, "
          "Tensor cov3D_precomp, Tensor sh_rest, Tensor viewmatrix, Tensor projmatrix, Tensor campos, Tensor bg, Tensor points_transform, "
          "int image_height, int image_width, float tanfovx, float tanfovy, float scale_modifier, int sh_degree, bool raw_params, "
          "bool prefiltered, bool debug, bool cam_grad, Tensor[] adam_m, Tensor[] adam_v, float[] adam_lr, float beta1, float beta2, "
          "float eps, int step, Tensor prepared, Tensor next_viewmatrix, Tensor next_projmatrix, Tensor next_campos, int next_height, "
          "int next_width, float next_tanfovx, float next_tanfovy, Tensor next_points_transform, int next_sh_degree, Tensor adam_commit, Tensor[] densify_stats, int[] batch_first_block, int view_id=0, int extras=0) -> "
          "(Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("mark_visible(Tensor means3D, Tensor viewmatrix, Tensor projmatrix) -> Tensor");
    m.def("photometric_loss_forward(Tensor render, Tensor target, float lambda_dssim, bool clamp) -> (Tensor, Tensor)");
    m.def("photometric_loss_backward(Tensor render, Tensor target, Tensor workspace, Tensor grad_loss, float lambda_dssim, bool clamp) -> Tensor");
    m.def("photometric_loss(Tensor render, Tensor target, float lambda_dssim, bool clamp) -> Tensor");
    m.def("photometric_loss_terms(Tensor render, Tensor target, float lambda_dssim, bool clamp) -> (Tensor, Tensor)");
    m.def("adam_step(Tensor(a!)[] params, Tensor[] grads, Tensor(b!)[] exp_avg, Tensor(c!)[] exp_avg_sq, float[] lr, float beta1, "
          "float beta2, float eps, int step) -> ()");
    m.def("pose_step(Tensor(a!) delta, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, Tensor d_xf, Tensor base, Tensor(d!) xf, float lr, "
          "float beta1, float beta2, float eps, int step) -> ()");
    m.def("pose_step_camera(Tensor(a!) delta, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, Tensor d_viewmatrix, Tensor d_projmatrix, "
          "Tensor d_campos, Tensor projection_T, Tensor base, Tensor(d!) viewmatrix, Tensor(e!) projmatrix, Tensor(f!) campos, float lr, "
          "float beta1, float beta2, float eps, int step) -> ()");
    m.def("pose_matrix_forward(Tensor delta, Tensor base) -> Tensor");
    m.def("pose_matrix_backward(Tensor delta, Tensor base, Tensor d_xf) -> Tensor");
    m.def("pose_matrix(Tensor delta, Tensor base) -> Tensor");
    m.def("knn_mean_dist2(Tensor points) -> Tensor");
    m.def("masked_max_(Tensor(a!) dst, Tensor src, Tensor mask) -> ()");
    m.def("densify_stats_add_(Tensor(a!) accum, Tensor(b!) denom, Tensor grad, Tensor mask) -> ()");
    m.def("psnr(Tensor a, Tensor b) -> Tensor");
    m.def("debug_last() -> Tensor[]", &debug_last);
}

TORCH_LIBRARY_IMPL(gsr, CUDA, m)   // the dispatch key of HIP tensors on a ROCm build
{
    m.impl("rasterize_forward", &rasterize_forward);
    m.impl("rasterize_backward", &rasterize_backward);
    m.impl("rasterize_backward_fused", &rasterize_backward_fused);
    m.impl("mark_visible", &mark_visible);
    m.impl("photometric_loss_forward", &photometric_loss_forward);
    m.impl("photometric_loss_backward", &photometric_loss_backward);
    m.impl("adam_step", &adam_step);
    m.impl("pose_step", &pose_step);
    m.impl("pose_step_camera", &pose_step_camera);
    m.impl("pose_matrix_forward", &pose_matrix_forward);
    m.impl("pose_matrix_backward", &pose_matrix_backward);
    m.impl("pose_matrix", &pose_matrix_forward);
    m.impl("knn_mean_dist2", &knn_mean_dist2);
    m.impl("masked_max_", &masked_max_);
    m.impl("densify_stats_add_", &densify_stats_add_);
    m.impl("psnr", &psnr);
    m.impl("rasterize", &rasterize_forward_only);
    m.impl("photometric_loss", &photometric_loss_no_grad);
    m.impl("photometric_loss_terms", &photometric_loss_terms_no_grad);
}

TORCH_LIBRARY_IMPL(gsr, Autograd, m)
{
    m.impl("rasterize", &rasterize);
    m.impl("photometric_loss", &photometric_loss);
    m.impl("photometric_loss_terms", &photometric_loss_terms);
    m.impl("pose_matrix", &pose_matrix);
}
