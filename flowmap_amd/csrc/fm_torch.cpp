// libflowmap_torch.so — the torch binding of the C ABI in include/flowmap_hip.h.
//
// The reference (dcharatan/flowmap) is pure Python: its hot path is autograd walking chains of
// ATen ops.  Here the per-step operators of that path — IntrinsicsRegressed's K
// (flowmap/model/intrinsics/common.py:6-20), align_surfaces + align_rigid + get_extrinsics
// (flowmap/model/projection.py:187-252, flowmap/model/procrustes.py:7-51), LossFlow
// (flowmap/loss/loss_flow.py:31-70), LossTracking (flowmap/loss/loss_tracking.py:28-61) and the Adam
// update (flowmap/model/model_wrapper_overfit.py:104-105) — are C++ torch::autograd::Functions
// registered with TORCH_LIBRARY under the namespace `flowmap_amd`: at::Tensor shims that check their
// arguments (TORCH_CHECK -> RuntimeError), allocate outputs / workspaces with the caching allocator,
// select the tensor's device and launch the hand-written HIP kernels through the C ABI on
// the current HIP stream (c10::hip::getCurrentHIPStreamMasqueradingAsCUDA: PyTorch-ROCm names its devices "cuda").  No kernel lives here (this file is host-only C++, built with g++);
// no state lives here either, beyond the table of C-ABI entry points: what one step hands from one
// operator to the next (the shared dense dL/ddepth, the persistent dL/dweights storage) travels in
// explicit objects (DepthSink, GradArena) that the Python layer creates and passes in.
//
// The C ABI is resolved with dlopen / dlsym so that the CPU test-suite can point the same binding at
// tests/host_sim's serial build of the ABI (set_library); the product only ever loads libflowmap_hip.so.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>  // PyTorch-ROCm calls its HIP devices "cuda": these are the
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>     // guard / stream types that go with that device type
#include <dlfcn.h>
#include <torch/autograd.h>
#include <torch/custom_class.h>
#include <torch/library.h>

#include <cmath>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/flowmap_hip.h"

namespace fmt {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::Function;
using torch::autograd::variable_list;
using OptTensor = std::optional<Tensor>;

// ------------------------------------------------------------------------------------------
// The C ABI, resolved at run time
// ------------------------------------------------------------------------------------------
#define FM_API_LIST(X)                                                                                                                    \
  X(fm_flow_loss_fused) X(fm_flow_loss_fused_adam) X(fm_adam_step_elements) X(fm_flow_loss_finalize) X(fm_scale_if_needed) X(fm_intrinsics_inverse) X(fm_intrinsics_inverse_bwd)              \
  X(fm_focal_intrinsics_fwd) X(fm_focal_intrinsics_bwd) X(fm_pose_chain_fwd) X(fm_pose_chain_bwd) X(fm_relative_pose_fwd)                  \
  X(fm_relative_pose_bwd) X(fm_procrustes_fit) X(fm_procrustes_fit_chain) X(fm_depth_gather_kgrad) X(fm_pose_solve_bwd) X(fm_pose_solve_bwd_kinv) X(fm_procrustes_scatter) X(fm_procrustes_scatter_dense) X(fm_procrustes_bwd_planned) X(fm_flow_loss_fused_views) X(fm_procrustes_fit_chain_views) X(fm_procrustes_fit_views) X(fm_procrustes_scatter_views)                \
  X(fm_depth_gather) X(fm_extrinsics_inverse) X(fm_track_loss_fused_fwd) X(fm_track_loss_bwd) X(fm_adam_step)                 \
  X(fm_adam_step_capturable) X(fm_softmin_score_fwd) X(fm_softmin_score_bwd) X(fm_softmin_blend_fwd) X(fm_softmin_blend_bwd)         \
  X(fm_random_subset) X(fm_random_subset_stateful) X(fm_abi_version) X(fm_flow_loss_fused_taps) X(fm_track_loss_fused_fwd_taps) X(fm_tap_grad_apply)

struct Api {
#define X(name) decltype(&::name) name = nullptr;
  FM_API_LIST(X)
#undef X
  void* handle = nullptr;
  bool test_double = false;
  std::string path;
};

static Api& api_storage() {
  static Api a;
  return a;
}
static std::mutex& api_mutex() {
  static std::mutex m;
  return m;
}

static void set_library(const std::string& path, bool test_double) {
  std::lock_guard<std::mutex> lock(api_mutex());
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  TORCH_CHECK(h != nullptr, "flowmap_amd: cannot load ", path, ": ", dlerror(),
              " — build it with `python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU or eager fallback");
  Api a;
  a.handle = h;
  a.test_double = test_double;
  a.path = path;
#define X(name)                                                                    \
  a.name = reinterpret_cast<decltype(&::name)>(dlsym(h, #name));                   \
  TORCH_CHECK(a.name != nullptr, "flowmap_amd: ", path, " does not export ", #name);
  FM_API_LIST(X)
#undef X
  TORCH_CHECK(a.fm_abi_version() == FM_ABI_VERSION, "flowmap_amd: ", path, " implements version ", a.fm_abi_version(),
              " of the C ABI, this binding was built against version ", FM_ABI_VERSION, " (include/flowmap_hip.h): rebuild both");
  api_storage() = a;
}

static const Api& api() {
  const Api& a = api_storage();
  TORCH_CHECK(a.handle != nullptr, "flowmap_amd: the native library is not loaded (flowmap_amd._lib loads it on import)");
  return a;
}

static void fm_check(int status, const char* name) {
  TORCH_CHECK(status == 0, "flowmap_amd: ", name, " failed: ", status == 1 ? "invalid argument" : "HIP launch/runtime failure");
}
#define FM_CALL(fn, ...) fm_check(api().fn(__VA_ARGS__), #fn)

// ------------------------------------------------------------------------------------------
// Small helpers
// ------------------------------------------------------------------------------------------
// The C ABI takes dense row-major buffers.  A view is accepted and copied; a copy that costs as much as a pass of the
// step is said out loud, once per argument kind, instead of silently (SURVEY.md §8b "Ownership": the reference's own
// tensors are contiguous at b = 1 — `depth[None]`, the Flows fields — so this is the day a caller passes a real view).
static Tensor f32c(const Tensor& t, const char* what) {
  TORCH_CHECK(t.scalar_type() == at::kFloat, "flowmap_amd: ", what, " must be float32 (got ", t.scalar_type(), ")");
  if (!t.is_contiguous() && t.numel() >= (int64_t(16) << 20))
    TORCH_WARN("flowmap_amd: ", what, " is a non-contiguous view of ", t.numel() * 4 / (1 << 20), " MB and is copied on every call; pass a contiguous tensor");
  return t.contiguous();
}

// An image stack (batch, frame, ...) read IN PLACE when it is a frame window of a larger tensor — x[:, s:s+f], earlier(x) / later(x),
// a slice of a pretraining batch: every frame dense, frames and batch entries any (non-negative) number of elements apart.  Anything
// else (a transposed or channel-sliced view) is copied like f32c does.  `lay` stays {0, 0} for a dense tensor.
static int64_t& view_copy_counter() {
  static int64_t n = 0;
  return n;
}
struct ImageStack {
  Tensor t;
  fm_layout lay{0, 0};
  bool is_view() const { return lay.frame_stride != 0 || lay.batch_stride != 0; }
};
static ImageStack image_stack(const Tensor& t, const char* what) {
  TORCH_CHECK(t.scalar_type() == at::kFloat, "flowmap_amd: ", what, " must be float32 (got ", t.scalar_type(), ")");
  ImageStack out;
  if (t.is_contiguous()) {
    out.t = t;
    return out;
  }
  bool frames_dense = t.dim() >= 3;
  int64_t per_frame = 1;
  for (int64_t d = t.dim() - 1; d >= 2 && frames_dense; --d) {
    if (t.size(d) != 1 && t.stride(d) != per_frame) frames_dense = false;
    per_frame *= t.size(d);
  }
  if (frames_dense && (t.size(1) == 1 || t.stride(1) >= per_frame)) {
    const int64_t frame_stride = t.size(1) == 1 ? per_frame : t.stride(1);
    // batch entries must not overlap — the launchers (flow_loss_launch, pack_launch) refuse a batch stride below the extent of one entry, so
    // a batch-EXPANDED stack (stride 0 over the batch) is copied here instead of failing there with "invalid argument"
    if (t.size(0) == 1 || t.stride(0) >= frame_stride * (t.size(1) - 1) + per_frame) {
      out.t = t;
      out.lay.frame_stride = frame_stride;
      out.lay.batch_stride = t.size(0) == 1 ? frame_stride * t.size(1) : t.stride(0);
      return out;
    }
  }
  ++view_copy_counter();
  out.t = f32c(t, what);
  return out;
}

static c10::Device check_device(std::initializer_list<const Tensor*> tensors) {
  std::optional<c10::Device> dev;
  for (const Tensor* t : tensors) {
    if (t == nullptr || !t->defined()) continue;
    if (!dev) dev = t->device();
    else TORCH_CHECK(t->device() == *dev, "flowmap_amd: tensors on different devices (", *dev, " vs ", t->device(), ")");
  }
  TORCH_CHECK(dev.has_value(), "flowmap_amd: no tensor arguments");
  const bool dbl = api().test_double;
  TORCH_CHECK(dev->is_cuda() || dbl, "flowmap_amd: tensors are on ", *dev,
              "; the HIP path needs a GPU (device 'cuda' on ROCm). There is no CPU fallback.");
  TORCH_CHECK(!(dev->is_cuda() && dbl), "flowmap_amd: the host test double cannot take GPU tensors");
  return *dev;
}

struct DeviceScope {  // select the tensors' GPU for the launches inside; no-op for the host double
  std::optional<c10::hip::HIPGuardMasqueradingAsCUDA> guard;
  void* stream = nullptr;
  explicit DeviceScope(const c10::Device& dev) {
    if (dev.is_cuda()) {
      guard.emplace(dev);
      stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    }
  }
};

template <class T = float>
static T* ptr(const Tensor& t) {
  return t.defined() ? static_cast<T*>(t.data_ptr()) : nullptr;
}
template <class T = float>
static T* ptr(const OptTensor& t) {
  return (t.has_value() && t->defined()) ? static_cast<T*>(t->data_ptr()) : nullptr;
}
static Tensor opt(const OptTensor& t) { return t.has_value() ? *t : Tensor(); }

static Tensor empty_like_shape(at::IntArrayRef shape, const Tensor& like, at::ScalarType dtype = at::kFloat) {
  return at::empty(shape, like.options().dtype(dtype));
}

// ------------------------------------------------------------------------------------------
// Objects one optimisation step hands between its operators
// ------------------------------------------------------------------------------------------
// DepthSink: the dense dL/ddepth of a step has three producers — the fused flow loss (every pixel),
// the tracking loss and the Procrustes fit (sparse) — and autograd would sum three full-size
// tensors (1.7 GB of traffic at C1).  The fit's node always runs after the two losses (they consume
// its poses), so the losses park what they have here and the fit returns ONE buffer.  Created per
// step by align_surfaces (Python), passed to the three operators explicitly; nothing global.
struct DepthSink : torch::CustomClassHolder {
  bool active = false;        // the fit node will run and return dL/ddepth for `depth`
  const void* depth_ptr = nullptr;
  int64_t depth_version = -1;
  std::vector<int64_t> depth_sizes;
  Tensor carried;                                         // dense gradient parked by the flow loss
  Tensor carried_k;                                       // ... and its dL/dK (B,F,3,3): the fit adds its own part into it
  const void* k_ptr = nullptr;                            // identity of the K tensor the fit was given
  std::vector<std::function<void(Tensor&)>> pending;      // sparse scatters into the final buffer (tracking loss)
  // what the fit returned for this tensor, WEAKLY (a strong reference would make AccumulateGrad deep-copy the
  // 553 MB gradient instead of adopting it): LeadingFrames adds its frames into it while autograd still holds it
  using WeakImpl = c10::weak_intrusive_ptr<c10::TensorImpl, c10::UndefinedTensorImpl>;
  std::optional<WeakImpl> final_buffer;
  std::function<void(const Tensor&, int64_t)> on_leading_add;  // told when LeadingFrames has added `count` frames into the final buffer
  bool expect_leading = false;
  void note_final(const Tensor& t) {
    if (expect_leading && t.defined()) final_buffer.emplace(t.getIntrusivePtr());
  }
  Tensor take_final() {
    Tensor t;
    if (final_buffer.has_value() && !final_buffer->expired()) t = Tensor(final_buffer->lock());
    final_buffer.reset();
    return t;
  }
  const torch::autograd::Node* fit_node = nullptr;        // the fit's autograd node (identity only): a loss parks its gradient
                                                          // only when its poses come from this node, i.e. the node WILL run later
  int64_t leading_in_place = 0, leading_dense = 0, planned_steps = 0;  // which path ran (tests)
  // frame sharding (FrameShard.enable_early_halo): the flow loss leaves its dense dL/ddepth here right after its FORWARD pass (that is when it
  // exists), so that the boundary frames' exchange overlaps the rest of the step; `unit_flag` (one int32 on the device) is raised by its
  // backward when the upstream gradient turns out not to be 1 (the early copy is then stale)
  bool want_early = false;
  Tensor early_dense, unit_flag;
  bool in_pass_confirmed = false;  // the backward of a flow loss that applied the in-pass Adam update has run (FusedAdam.step checks)
  // The tap exchange (include/flowmap_hip.h: fm_flow_taps).  A tracking loss evaluated BEFORE the flow pass offers its unscaled dL/ddepth
  // at its static taps (tap_grad, M floats) with its normaliser (tap_scale); the flow pass adds scale·tap_grad into the dense gradient it
  // writes (tap_absorbed) — assuming both losses reach backward() with the same upstream gradient.  Whether that held is settled where
  // everything is known: in the fit's backward, which runs after both (settle_taps), with one conditional launch when it did not.
  Tensor tap_grad, tap_scale, tap_pixels;
  bool tap_absorbed = false;
  Tensor tap_flow_upstream, tap_track_upstream;  // defined once the respective node's backward has run in this pass
  bool tap_confirmed = false;                    // the tracking node whose gradient was absorbed has run its backward (FusedAdam.step checks)
  // The absorbing pass also applied the in-pass Adam update at the taps (they left the element list): the update used scale·tap_grad at
  // factor 1 and cannot be corrected afterwards.  This is the optimiser's scaled-loss flag; settle_taps raises it ON THE DEVICE when the
  // two upstream gradients turn out to differ (`flow + 3·tracking`), and FusedAdam.step reports it like a scaled loss.
  Tensor tap_adam_flag;
  int64_t taps_settled_free = 0, taps_settled_launch = 0;  // (tests)
  void offer_taps(const Tensor& grad, const Tensor& scale, const Tensor& pixels) {
    tap_grad = grad, tap_scale = scale, tap_pixels = pixels;
    tap_absorbed = false;
    tap_adam_flag = Tensor();
    tap_flow_upstream = Tensor(), tap_track_upstream = Tensor();
  }
  bool offers_taps() const { return tap_grad.defined() && !tap_absorbed; }
  void settle_taps(Tensor& buffer);

  void arm(const Tensor& depth) {
    active = true;
    depth_ptr = depth.data_ptr();
    depth_version = (int64_t)depth._version();
    depth_sizes = depth.sizes().vec();
  }
  bool accepts(const Tensor& depth) const {
    return active && depth.data_ptr() == depth_ptr && (int64_t)depth._version() == depth_version && depth.sizes().vec() == depth_sizes;
  }
};

// GradArena: the dL/dweights of a sparse fit (P points per pair) is a dense (B,F-1,H,W) tensor that is
// zero except at P·(F-1) slots — 549 MB of zeros written per step at C1.  The arena keeps ONE storage
// across steps: zeroed once; every step the planned per-correspondence pass overwrites the same slots
// (constant index set) and autograd is handed a fresh alias of the storage, which AccumulateGrad adopts
// without a copy.  If a previous alias is still alive (gradient accumulation without zero_grad), or
// somebody wrote into the gradient in place (version counter moved), the step falls back to fresh zeros.
struct GradArena : torch::CustomClassHolder {
  Tensor buffer;
  const void* indices_ptr = nullptr;
  int64_t indices_version = -1, version = -1;
  int64_t leading = 0;               // images at the front that another node adds into after the fit (the softmin sweep): re-zeroed per step
  int64_t reused = 0, refilled = 0;  // (tests)

  // a tensor shaped like `like`, zero everywhere except (possibly) at the slots `indices` selects in every pair
  Tensor acquire(const Tensor& like, const Tensor& indices) {
    const bool same_layout = buffer.defined() && buffer.sizes() == like.sizes() && buffer.device() == like.device();
    const bool same_slots = indices.data_ptr() == indices_ptr && (int64_t)indices._version() == indices_version;
    const bool untouched = same_layout && (int64_t)buffer._version() == version && buffer.storage().use_count() == 1;
    if (same_layout && same_slots && untouched) {
      ++reused;
      if (leading > 0) buffer.narrow(1, 0, std::min<int64_t>(leading, buffer.size(1))).zero_();  // (3.7 MB per image at 720p, not 549 MB)
    } else {
      if (!(same_layout && buffer.storage().use_count() == 1)) buffer = at::empty_like(like);  // never write under a live alias
      buffer.zero_();
      ++refilled;
    }
    indices_ptr = indices.data_ptr();
    indices_version = (int64_t)indices._version();
    version = (int64_t)buffer._version();
    return buffer;
  }
  Tensor alias() const { return buffer.alias(); }  // a new tensor over the same storage: autograd's to keep
  // LeadingFrames added `count` leading images into the alias it was handed: an expected edit — the storage still is
  // "zero outside the slots and the first `leading` images"
  void note_leading_add(const Tensor& into, int64_t count) {
    if (!buffer.defined() || into.data_ptr() != buffer.data_ptr()) return;
    leading = std::max(leading, count);
    version = (int64_t)buffer._version();
  }
  // is `grad` this arena's storage exactly as the last backward left it (zero outside the slots, nothing added or edited)?
  bool holds(const Tensor& grad) const {
    return buffer.defined() && grad.defined() && grad.data_ptr() == buffer.data_ptr() && grad.numel() == buffer.numel() &&
           (int64_t)grad._version() == version && grad.is_contiguous();
  }
};

// What the flow pass absorbed for the tracking loss was right if both losses reached backward() with the same upstream gradient — the
// same tensor object when they are summed into one root (AddBackward hands its gradient to both unchanged): nothing to do.  Otherwise
// buffer += scale·(plus − minus)·tap_grad at the taps, one launch whose blocks leave at once when the factor turns out to be zero.
void DepthSink::settle_taps(Tensor& buffer) {
  if (!tap_absorbed || !tap_grad.defined()) return;
  const Tensor plus = tap_track_upstream, minus = tap_flow_upstream;
  tap_flow_upstream = Tensor(), tap_track_upstream = Tensor();  // (a second backward through a retained graph starts clean)
  if (!plus.defined() && !minus.defined()) return;
  if (plus.defined() && minus.defined() && plus.data_ptr() == minus.data_ptr()) {
    ++taps_settled_free;
    return;
  }
  ++taps_settled_launch;
  TORCH_CHECK(buffer.defined() && buffer.is_contiguous(), "flowmap_amd: the tap exchange needs a dense dL/ddepth buffer");
  DeviceScope scope(buffer.device());
  FM_CALL(fm_tap_grad_apply, ptr(tap_grad), ptr<int64_t>(tap_pixels), (long)tap_pixels.numel(), ptr(tap_scale), ptr(plus), ptr(minus), ptr(buffer),
          ptr<int>(tap_adam_flag), scope.stream);
}

// tests and A/B timing: the planned sparse fit's backward as one launch (default) or as the three launches it replaces
static bool& one_launch_backward_flag() {
  static bool on = true;
  return on;
}
static bool use_one_launch_backward() { return one_launch_backward_flag(); }
static void set_one_launch_backward(bool on) { one_launch_backward_flag() = on; }

// The root of a step's backward: `loss.backward()` makes autograd fill a fresh ones_like(loss) and hands it down as the seed — one launch — and
// the flow loss then has to find out ON THE DEVICE that the seed is 1 (fm_scale_if_needed: a second launch that does nothing).  The losses
// of this package are returned as `RootLoss` tensors (flowmap_amd/_ops.py) whose backward() seeds with a ones tensor made once and registered
// here: a seed that IS that tensor (same memory, never written: version counter 0) is known to be 1 on the host and nothing is launched.
static std::vector<Tensor>& unit_seeds() {
  static auto* seeds = new std::vector<Tensor>();  // (never destroyed: device tensors must not be freed during static destruction, after the allocator)
  return *seeds;
}
static void register_unit_seed(const Tensor& seed) {
  TORCH_CHECK(seed.defined() && seed.numel() == 1 && seed.scalar_type() == at::kFloat && seed._version() == 0 && !seed.requires_grad(),
              "flowmap_amd: a unit seed is one float32 that nobody has written since it was made");
  for (const Tensor& known : unit_seeds())
    if (known.data_ptr() == seed.data_ptr()) return;
  unit_seeds().push_back(seed);
}
static bool is_unit_seed(const Tensor& g) {
  if (!g.defined() || g.numel() != 1) return false;
  for (const Tensor& known : unit_seeds())
    if (known.data_ptr() == g.data_ptr() && known._version() == 0 && g.scalar_type() == at::kFloat) return true;
  return false;
}
static int64_t& unit_seed_uses() {
  static int64_t count = 0;
  return count;
}

// does the autograd graph above `from` contain `target` within `depth` hops?  (poses -> [chain ->] fit)
static bool reaches(const std::shared_ptr<torch::autograd::Node>& from, const torch::autograd::Node* target, int depth) {
  if (!from || target == nullptr) return false;
  if (from.get() == target) return true;
  if (depth == 0) return false;
  for (const auto& edge : from->next_edges())
    if (edge.function && reaches(edge.function, target, depth - 1)) return true;
  return false;
}

// ------------------------------------------------------------------------------------------
// Intrinsics
// ------------------------------------------------------------------------------------------
static Tensor intrinsics_inverse(const Tensor& k_in) {
  const auto dev = check_device({&k_in});
  TORCH_CHECK(k_in.dim() >= 2 && k_in.size(-1) == 3 && k_in.size(-2) == 3, "flowmap_amd: intrinsics must be (..., 3, 3)");
  const Tensor k = f32c(k_in, "intrinsics");
  Tensor out = at::empty_like(k);
  DeviceScope scope(dev);
  FM_CALL(fm_intrinsics_inverse, ptr(k), (int)(k.numel() / 9), ptr(out), scope.stream);
  return out;
}

// focal_lengths_to_intrinsics spread over the frames (intrinsics/common.py:6-20 as used by
// intrinsics_regressed.py:34-41): focal (*lead) -> K (*lead, *repeat_shape, 3, 3) in one launch that also
// leaves K^-1 behind for the step's consumers; the backward is one reduction.
struct FocalIntrinsics : public Function<FocalIntrinsics> {
  static variable_list forward(AutogradContext* ctx, const Tensor& focal_in, std::vector<int64_t> repeat_shape, int64_t h, int64_t w) {
    const auto dev = check_device({&focal_in});
    const Tensor focal = f32c(focal_in, "focal lengths");
    int64_t repeat = 1;
    for (auto d : repeat_shape) repeat *= d;
    std::vector<int64_t> shape = focal.sizes().vec();
    shape.insert(shape.end(), repeat_shape.begin(), repeat_shape.end());
    shape.push_back(3);
    shape.push_back(3);
    Tensor k = empty_like_shape(shape, focal), kinv = empty_like_shape(shape, focal);
    DeviceScope scope(dev);
    FM_CALL(fm_focal_intrinsics_fwd, ptr(focal), (long)focal.numel(), (long)repeat, (int)h, (int)w, ptr(k), ptr(kinv), scope.stream);
    ctx->saved_data["geometry"] = std::vector<int64_t>{focal.numel(), repeat, h, w};
    ctx->saved_data["shape"] = focal.sizes().vec();
    ctx->mark_non_differentiable({kinv});
    ctx->set_materialize_grads(false);  // (else the engine builds a zeros tensor for K⁻¹'s absent gradient in every backward: one fill launch per step)
    return {k, kinv};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    if (!grads[0].defined()) return {Tensor(), Tensor(), Tensor(), Tensor()};
    const auto geo = ctx->saved_data["geometry"].toIntVector();
    const Tensor g_k = f32c(grads[0], "grad");
    Tensor g_focal = empty_like_shape(ctx->saved_data["shape"].toIntVector(), g_k);
    DeviceScope scope(g_k.device());
    FM_CALL(fm_focal_intrinsics_bwd, ptr(g_k), (long)geo[0], (long)geo[1], (int)geo[2], (int)geo[3], ptr(g_focal), scope.stream);
    return {g_focal, Tensor(), Tensor(), Tensor()};
  }
};

// ------------------------------------------------------------------------------------------
// Pose plumbing
// ------------------------------------------------------------------------------------------
struct PoseChain : public Function<PoseChain> {  // get_extrinsics (projection.py:187-210)
  static Tensor forward(AutogradContext* ctx, const Tensor& rel_in) {
    const auto dev = check_device({&rel_in});
    const Tensor rel = f32c(rel_in, "relative transformations");
    TORCH_CHECK(rel.dim() >= 3 && rel.size(-1) == 4 && rel.size(-2) == 4, "flowmap_amd: relative transformations must be (..., steps, 4, 4)");
    const int64_t steps = rel.size(-3);
    const int64_t nb = steps > 0 ? rel.numel() / (steps * 16) : 1;
    std::vector<int64_t> shape = rel.sizes().vec();
    shape[shape.size() - 3] = steps + 1;
    Tensor ext = empty_like_shape(shape, rel);
    DeviceScope scope(dev);
    FM_CALL(fm_pose_chain_fwd, ptr(rel), (int)nb, (int)steps, ptr(ext), scope.stream);
    ctx->save_for_backward({rel, ext});
    ctx->saved_data["dims"] = std::vector<int64_t>{nb, steps};
    return ext;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    if (!grads[0].defined()) return {Tensor()};
    const auto saved = ctx->get_saved_variables();
    const auto dims = ctx->saved_data["dims"].toIntVector();
    const Tensor g_ext = f32c(grads[0], "grad");
    Tensor g_rel = at::empty_like(saved[0]);
    DeviceScope scope(g_ext.device());
    FM_CALL(fm_pose_chain_bwd, ptr(saved[0]), ptr(saved[1]), ptr(g_ext), (int)dims[0], (int)dims[1], ptr(g_rel), scope.stream);
    return {g_rel};
  }
};

// later(E).inverse() @ earlier(E) and earlier(E).inverse() @ later(E) (projection.py:154,176)
struct RelativePoses : public Function<RelativePoses> {
  static variable_list forward(AutogradContext* ctx, const Tensor& ext_in) {
    const auto dev = check_device({&ext_in});
    const Tensor ext = f32c(ext_in, "extrinsics");
    TORCH_CHECK(ext.dim() == 4 && ext.size(1) >= 2 && ext.size(2) == 4 && ext.size(3) == 4, "flowmap_amd: extrinsics must be (batch, frame >= 2, 4, 4)");
    const int64_t b = ext.size(0), f = ext.size(1);
    Tensor fwd = empty_like_shape({b, f - 1, 4, 4}, ext), bwd = empty_like_shape({b, f - 1, 4, 4}, ext);
    DeviceScope scope(dev);
    FM_CALL(fm_relative_pose_fwd, ptr(ext), (int)b, (int)f, ptr(fwd), ptr(bwd), scope.stream);
    ctx->save_for_backward({ext});
    return {fwd, bwd};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const Tensor ext = ctx->get_saved_variables()[0];
    if (!grads[0].defined() && !grads[1].defined()) return {Tensor()};
    const Tensor g_fwd = grads[0].defined() ? f32c(grads[0], "grad") : Tensor();
    const Tensor g_bwd = grads[1].defined() ? f32c(grads[1], "grad") : Tensor();
    Tensor g_ext = at::empty_like(ext);
    DeviceScope scope(ext.device());
    FM_CALL(fm_relative_pose_bwd, ptr(ext), ptr(g_fwd), ptr(g_bwd), (int)ext.size(0), (int)ext.size(1), ptr(g_ext), scope.stream);
    return {g_ext};
  }
};

// ------------------------------------------------------------------------------------------
// Procrustes fit of adjacent frames: align_surfaces up to (not including) the pose chain
// (projection.py:213-249) with align_rigid (procrustes.py:7-51) inside.
//   depth (B,F,H,W) + k, kinv (B·R,F,3,3)  [surfaces never materialised]   or   surfaces (B,F,H,W,3)
//   weights (B,F-1,H,W)  (LOGITS when weight_sens != 0: w = sigmoid(weight_sens·logit), evaluated at the
//   gathered points only — BackboneExplicitDepth fused into the gather; the gradient is then w.r.t. the logits)
//   -> t_bwd (B·R,F-1,4,4) later -> earlier camera ("inverse relative transformations"), t_fwd its rigid inverse.
// Backward plans (built by the Python layer from the constant flows / indices, or undefined):
//   sparse: plan_pixels, plan_first, plan_vectors, plan_weights (fm_procrustes_scatter_plan, sorted) -> fm_depth_gather
//   dense:  dense_first, dense_list (fm_procrustes_dense_plan) -> fm_procrustes_scatter_dense
// ------------------------------------------------------------------------------------------
struct ProcrustesFit : public Function<ProcrustesFit> {
  // NB: needs_input_grad(i) indexes the i-th tensor argument that is PRESENT (an empty optional<Tensor> is not an
  // autograd input; an undefined plain Tensor is not accepted at all), so the indices are worked out in forward.
  static variable_list forward(AutogradContext* ctx, const OptTensor& depth_o, const OptTensor& k_o, const OptTensor& kinv_o,
                               const OptTensor& surfaces_o, const Tensor& weights_in, const Tensor& bwd_flow_in, const OptTensor& indices_o,
                               double weight_sens, int64_t rep, const c10::intrusive_ptr<DepthSink>& sink,
                               const c10::intrusive_ptr<DepthSink>& wsink, const c10::intrusive_ptr<GradArena>& arena,
                               const OptTensor& plan_pixels, const OptTensor& plan_first, const OptTensor& plan_vectors,
                               const OptTensor& plan_weights, const OptTensor& plan_frame_first, const OptTensor& plan_tap_records,
                               const OptTensor& dense_first, const OptTensor& dense_list, const OptTensor& work_o, bool want_ext, bool grad_enabled) {  // (forward runs with grad mode off: the caller's mode is passed in)
    ctx->set_materialize_grads(false);  // an unused output (the extrinsics of a flow-only step) must not cost a zeros tensor + the chain's backward
    Tensor depth = opt(depth_o), k = opt(k_o), kinv = opt(kinv_o), surfaces = opt(surfaces_o), indices = opt(indices_o);
    const bool from_depth = !surfaces.defined();
    {  // edge index of depth / k / surfaces / weights among the present tensor arguments
      int64_t at_depth = 0, at_k = depth.defined(), at_surf = at_k + k.defined() + kinv.defined(), at_w = at_surf + surfaces.defined();
      ctx->saved_data["edges"] = std::vector<int64_t>{at_depth, at_k, at_surf, at_w};
    }
    const Tensor& src_in = from_depth ? depth : surfaces;
    TORCH_CHECK(src_in.defined(), "flowmap_amd: the Procrustes fit needs depth + intrinsics or surfaces");
    const auto dev = check_device({&src_in, &weights_in, &bwd_flow_in, &indices, &k, &kinv});
    // sparse, un-repeated fits read frame windows in place (fm_procrustes_fit_chain_views / _fit_views / _scatter_views); the tiled dense
    // kernels and the softmin sweep's repeated batches take dense stacks
    const bool views_ok = indices_o.has_value() && indices_o->defined() && rep == 1;
    const Tensor weights = views_ok ? image_stack(weights_in, "weights").t : f32c(weights_in, "weights");
    const Tensor bwd_flow = views_ok ? image_stack(bwd_flow_in, "backward flow").t : f32c(bwd_flow_in, "backward flow");
    TORCH_CHECK(!bwd_flow_in.requires_grad(), "flowmap_amd: gradients w.r.t. optical flow are not supported (flows are constants)");
    TORCH_CHECK(rep >= 1, "flowmap_amd: batch_repeat must be >= 1");
    int64_t bd, f, h, w, b;
    if (from_depth) {
      depth = views_ok ? image_stack(depth, "depth").t : f32c(depth, "depth");
      TORCH_CHECK(depth.dim() == 4, "flowmap_amd: depth must be (batch, frame, height, width)");
      bd = depth.size(0), f = depth.size(1), h = depth.size(2), w = depth.size(3);
      b = bd * rep;  // pose / intrinsics batch: every image-batch entry serves `rep` candidates
      TORCH_CHECK(k.defined() && kinv.defined(), "flowmap_amd: depth-sourced surfaces need intrinsics and their inverse");
      k = f32c(k, "intrinsics");
      kinv = f32c(kinv, "inverse intrinsics");
      TORCH_CHECK(k.sizes() == at::IntArrayRef({b, f, 3, 3}) && kinv.sizes() == k.sizes(),
                  "flowmap_amd: intrinsics shape does not match depth (x batch_repeat)");
    } else {
      TORCH_CHECK(rep == 1, "flowmap_amd: batch_repeat needs depth-sourced surfaces");
      surfaces = views_ok ? image_stack(surfaces, "surfaces").t : f32c(surfaces, "surfaces");
      TORCH_CHECK(surfaces.dim() == 5 && surfaces.size(4) == 3, "flowmap_amd: surfaces must be (batch, frame, height, width, 3)");
      bd = surfaces.size(0), f = surfaces.size(1), h = surfaces.size(2), w = surfaces.size(3);
      b = bd;
    }
    TORCH_CHECK(f >= 2, "flowmap_amd: at least two frames are needed");
    TORCH_CHECK(weights.sizes() == at::IntArrayRef({bd, f - 1, h, w}) && bwd_flow.sizes() == at::IntArrayRef({bd, f - 1, h, w, 2}),
                "flowmap_amd: weights/backward-flow shapes do not match the surfaces");
    int64_t points = h * w;
    if (indices.defined()) {
      TORCH_CHECK(indices.scalar_type() == at::kLong, "flowmap_amd: indices must be int64");
      indices = indices.contiguous();
      points = indices.numel();
    }
    const int64_t pairs = b * (f - 1);
    Tensor aux = at::empty({pairs, FM_AUX_STRIDE}, weights.options().dtype(at::kDouble));
    Tensor t_bwd = empty_like_shape({b, f - 1, 4, 4}, weights), t_fwd = empty_like_shape({b, f - 1, 4, 4}, weights);
    // With a persistent workspace (sparse index set, no repeat): moments, finish + solve and the pose chain in ONE launch
    Tensor work = opt(work_o), ext, stats, corr;
    const bool chained = work.defined() && indices.defined() && rep == 1;
    fm_layout lay[4] = {};  // depth, surfaces, bwd_flow, weights
    bool any_view = false;
    if (views_ok) {
      const Tensor* stacks[4] = {&depth, &surfaces, &bwd_flow, &weights};
      for (int i = 0; i < 4; ++i)
        if (stacks[i]->defined()) {
          const ImageStack st = image_stack(*stacks[i], "image stack");  // (in place already: no copy happens here)
          lay[i] = st.lay;
          any_view = any_view || st.is_view();
        }
    }
    {
      DeviceScope scope(dev);
      if (chained) {
        TORCH_CHECK(work.scalar_type() == at::kDouble && work.is_contiguous() && work.numel() >= pairs * FM_STAT_STRIDE + (pairs + 2) / 2 + 1,
                    "flowmap_amd: the fit workspace is too small");
        // want_ext = false (lazy extrinsics, round 6): the caller chains the poses only if something asks for them — the flow loss takes the
        // relative poses, and the chain is the last block's ~7 us at 149 poses with the rest of the GPU idle (profiles/r06_fit_microbench.jsonl)
        if (want_ext) ext = empty_like_shape({b, f, 4, 4}, weights);
        // the planned backward (one launch, fm_procrustes_bwd_planned) reads the correspondences back instead of re-gathering them
        const bool wants_records = from_depth && grad_enabled && plan_frame_first.has_value() && plan_frame_first->defined() && points <= FM_FIT_BWD_MAX_POINTS &&
                                   use_one_launch_backward() && (depth.requires_grad() || weights_in.requires_grad() || (k_o.has_value() && k_o->requires_grad()));
        const bool taps_ok = from_depth && plan_tap_records.has_value() && plan_tap_records->defined() && plan_tap_records->scalar_type() == at::kFloat &&
                             plan_tap_records->is_contiguous() && plan_tap_records->numel() == pairs * points * 8 && points <= 4096;
        if (wants_records && taps_ok) corr = at::empty({pairs * points, 8}, weights.options());
        if (any_view)
          FM_CALL(fm_procrustes_fit_chain_views, from_depth ? ptr(depth) : nullptr, ptr(kinv), ptr(surfaces), ptr(bwd_flow), ptr(weights),
                  (float)weight_sens, ptr<int64_t>(indices), (long)points, (int)b, (int)f, (int)h, (int)w, ptr<double>(work), ptr(t_bwd), ptr(t_fwd),
                  ptr<double>(aux), ptr(ext), ptr(corr), taps_ok ? ptr(plan_tap_records) : nullptr, lay, scope.stream);
        else
          FM_CALL(fm_procrustes_fit_chain, from_depth ? ptr(depth) : nullptr, ptr(kinv), ptr(surfaces), ptr(bwd_flow), ptr(weights),
                  (float)weight_sens, ptr<int64_t>(indices), (long)points, (int)b, (int)f, (int)h, (int)w, ptr<double>(work), ptr(t_bwd), ptr(t_fwd),
                  ptr<double>(aux), ptr(ext), ptr(corr), taps_ok ? ptr(plan_tap_records) : nullptr, scope.stream);
      } else {
        stats = at::empty({pairs, FM_STAT_STRIDE}, weights.options().dtype(at::kDouble));
        if (any_view)
          FM_CALL(fm_procrustes_fit_views, from_depth ? ptr(depth) : nullptr, ptr(kinv), ptr(surfaces), ptr(bwd_flow), ptr(weights), (float)weight_sens,
                  ptr<int64_t>(indices), (long)points, (int)b, (int)f, (int)h, (int)w, ptr<double>(stats), ptr(t_bwd), ptr(t_fwd), ptr<double>(aux), lay,
                  scope.stream);
        else
          FM_CALL(fm_procrustes_fit, from_depth ? ptr(depth) : nullptr, ptr(kinv), ptr(surfaces), ptr(bwd_flow), ptr(weights), (float)weight_sens,
                  ptr<int64_t>(indices), (long)points, (int)b, (int)rep, (int)f, (int)h, (int)w, ptr<double>(stats), ptr(t_bwd), ptr(t_fwd),
                  ptr<double>(aux), scope.stream);
      }
    }
    ctx->save_for_backward({from_depth ? depth : surfaces, kinv, weights, bwd_flow, indices, t_bwd, aux, opt(plan_pixels), opt(plan_first),
                            opt(plan_vectors), opt(plan_weights), opt(dense_first), opt(dense_list), ext, opt(plan_frame_first), corr});
    ctx->saved_data["dims"] = std::vector<int64_t>{b, f, h, w, points, rep, from_depth ? 1 : 0};
    ctx->saved_data["weight_sens"] = weight_sens;
    if (sink) ctx->saved_data["sink"] = sink;
    if (wsink) ctx->saved_data["wsink"] = wsink;
    if (arena) ctx->saved_data["arena"] = arena;
    if (wsink && grad_enabled && weights_in.requires_grad()) {  // identity of the weights tensor LeadingFrames was given
      wsink->depth_ptr = weights_in.data_ptr();
      wsink->depth_version = (int64_t)weights_in._version();
    }
    if (sink && from_depth && rep == 1 && depth.requires_grad() && grad_enabled) {
      sink->arm(depth);
      sink->k_ptr = (k_o.has_value() && k_o->requires_grad()) ? k.data_ptr() : nullptr;
    }
    if (!chained || !ext.defined()) ext = at::empty({0}, weights.options());  // placeholder output: the caller chains the poses itself
    return {t_bwd, t_fwd, ext};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &src = saved[0], &kinv = saved[1], &weights = saved[2], &bwd_flow = saved[3], &indices = saved[4], &t_bwd = saved[5],
                 &aux = saved[6];
    const Tensor &plan_pixels = saved[7], &plan_first = saved[8], &plan_vectors = saved[9], &plan_weights = saved[10],
                 &dense_first = saved[11], &dense_list = saved[12], &ext = saved[13], &plan_frame_first = saved[14], &corr = saved[15];
    const auto dims = ctx->saved_data["dims"].toIntVector();
    const int64_t b = dims[0], f = dims[1], h = dims[2], w = dims[3], points = dims[4], rep = dims[5];
    const bool from_depth = dims[6] != 0;
    const float sens = (float)ctx->saved_data["weight_sens"].toDouble();
    auto sink = ctx->saved_data.count("sink") ? ctx->saved_data["sink"].toCustomClass<DepthSink>() : c10::intrusive_ptr<DepthSink>();
    auto arena = ctx->saved_data.count("arena") ? ctx->saved_data["arena"].toCustomClass<GradArena>() : c10::intrusive_ptr<GradArena>();
    auto wsink = ctx->saved_data.count("wsink") ? ctx->saved_data["wsink"].toCustomClass<DepthSink>() : c10::intrusive_ptr<DepthSink>();
    const int64_t pairs = b * (f - 1);
    const auto dev = weights.device();
    Tensor g_t = grads[0].defined() ? f32c(grads[0], "grad") : Tensor();
    const Tensor g_t_fwd = grads[1].defined() ? f32c(grads[1], "grad") : Tensor();
    if (grads.size() > 2 && grads[2].defined() && ext.defined() && ext.numel() > 0) {  // the chain's backward (get_extrinsics), then summed into dL/dT
      const Tensor g_ext = f32c(grads[2], "grad");
      Tensor g_rel = at::empty_like(t_bwd);
      DeviceScope scope(weights.device());
      FM_CALL(fm_pose_chain_bwd, ptr(t_bwd), ptr(ext), ptr(g_ext), (int)b, (int)(f - 1), ptr(g_rel), scope.stream);
      g_t = g_t.defined() ? g_t + g_rel : g_rel;
    }
    const auto edges = ctx->saved_data["edges"].toIntVector();  // depth, k, surfaces, weights
    const bool need_src = ctx->needs_input_grad(from_depth ? edges[0] : edges[2]);
    const bool need_k = from_depth && ctx->needs_input_grad(edges[1]);
    const bool need_w = ctx->needs_input_grad(edges[3]);
    Tensor pair_grad;
    Tensor g_src, g_k, g_w;

    // The dense dL/ddepth: what the losses parked (the sink is armed only for depth-sourced, un-repeated fits)
    Tensor carried, carried_k;
    std::vector<std::function<void(Tensor&)>> pending;
    if (sink) {
      carried = std::move(sink->carried);
      sink->carried = Tensor();
      carried_k = std::move(sink->carried_k);
      sink->carried_k = Tensor();
      pending.swap(sink->pending);
      sink->active = false;
    }
    if (need_src) {
      g_src = carried.defined() ? carried : at::zeros(src.sizes(), src.options());  // (dense, whatever the strides of a window)
      for (auto& scatter : pending) scatter(g_src);
      if (sink) sink->settle_taps(g_src);
    }
    // every pixel a correspondence: the tiled dense kernels — one fused pass with atomics, or, when the caller built the static tap lists
    // (flowmap_amd.set_dense_procrustes_planned), the planned pair of kernels whose dL/ddepth is bit-reproducible
    const bool dense = from_depth && !indices.defined() && rep == 1 && h <= 65535 && w <= 65535 &&
                       (!need_src || dense_first.defined() || (src.is_contiguous() && bwd_flow.is_contiguous() && weights.is_contiguous()));
    const bool planned = from_depth && indices.defined() && rep == 1 && plan_pixels.defined();
    bool arena_used = false;
    if (need_w) {
      if (dense) g_w = at::empty(weights.sizes(), weights.options());  // every element stored exactly once
      else if (planned && arena && g_src.defined()) {  // (the planned pass STORES dL/dweights at its slots)
        g_w = arena->acquire(weights, indices);         // zero except at the slots this very pass overwrites
        arena_used = true;
      } else g_w = at::zeros(weights.sizes(), weights.options());
    }
    // dL/dK⁻¹ is linear in the statistics the forward pass left in aux: written by the pose-solve backward itself, and the per-point
    // passes carry no sums for it (repeated batches — the softmin sweep's candidates — keep the per-point accumulation)
    const bool k_closed_form = need_k && from_depth && rep == 1;
    // A planned fit of up to FM_FIT_BWD_MAX_POINTS points per pair: pose-solve backward, per-correspondence gradients, the planned
    // gather and dL/dK in ONE launch, one workgroup per frame (fm_procrustes_bwd_planned)
    const bool one_launch = planned && g_src.defined() && plan_frame_first.defined() && corr.defined() && points <= FM_FIT_BWD_MAX_POINTS;
    if (one_launch) {
      const bool add_k = need_k && carried_k.defined() && carried_k.sizes() == kinv.sizes() && carried_k.is_contiguous();
      if (need_k) g_k = add_k ? carried_k : at::empty_like(kinv);
      {
        DeviceScope scope(dev);
        FM_CALL(fm_procrustes_bwd_planned, ptr(corr), ptr(kinv), sens, (long)points, (int)b, (int)f,
                (int)h, (int)w, ptr<double>(aux), ptr(t_bwd), ptr(g_t), ptr(g_t_fwd), ptr<int64_t>(plan_pixels), ptr<int32_t>(plan_first),
                ptr<int32_t>(plan_vectors), ptr(plan_weights), ptr<int32_t>(plan_frame_first), ptr(g_src), ptr(g_w), ptr(g_k), add_k ? 1 : 0,
                scope.stream);
      }
      if (carried_k.defined() && !add_k) g_k = g_k.defined() ? g_k + carried_k : carried_k;  // (unexpected layout: plain sum)
      if (sink) ++sink->planned_steps;
    }
    Tensor kinv_acc = (need_k && !one_launch) ? at::empty({b * f, 9}, weights.options().dtype(at::kDouble)) : Tensor();
    Tensor point_grads = (planned && g_src.defined() && !one_launch) ? at::empty({pairs * points, 2, 3}, weights.options()) : Tensor();
    if (!one_launch) {
      DeviceScope scope(dev);
      pair_grad = at::empty({pairs, FM_PAIR_GRAD_STRIDE}, weights.options().dtype(at::kDouble));
      if (k_closed_form) {
        FM_CALL(fm_pose_solve_bwd_kinv, ptr(g_t), ptr(g_t_fwd), ptr(t_bwd), ptr<double>(aux), ptr(kinv), (int)b, (int)f, ptr<double>(pair_grad),
                ptr<double>(kinv_acc), scope.stream);
      } else {  // (kinv_acc is zeroed by the same launch)
        FM_CALL(fm_pose_solve_bwd, ptr(g_t), ptr(g_t_fwd), ptr(t_bwd), ptr<double>(aux), (int)pairs, ptr<double>(pair_grad), ptr<double>(kinv_acc),
                kinv_acc.defined() ? (long)kinv_acc.numel() : 0L, scope.stream);
      }
      double* per_point_k = k_closed_form ? nullptr : ptr<double>(kinv_acc);
      if (dense) {  // every pixel a correspondence: tiled (fused pass, or planned without atomics)
        TORCH_CHECK(!need_k || k_closed_form, "flowmap_amd: the dense Procrustes backward derives dL/dK from the forward statistics (no batch repeat)");
        Tensor consts = at::empty({pairs, FM_DENSE_CONST_STRIDE}, weights.options().dtype(at::kDouble));
        FM_CALL(fm_procrustes_scatter_dense, ptr(src), ptr(kinv), ptr(bwd_flow), ptr(weights), sens, (int)b, (int)f, (int)h, (int)w,
                ptr<double>(aux), ptr<double>(pair_grad), ptr(g_src), ptr(g_w), ptr<int64_t>(dense_first), ptr<uint32_t>(dense_list),
                ptr<double>(consts), scope.stream);
      } else {
        fm_layout lay[4] = {};  // depth, surfaces, bwd_flow, weights: the saved inputs may be frame windows read in place
        bool any_view = false;
        if (indices.defined() && rep == 1) {
          const ImageStack s_src = image_stack(src, "image stack"), s_flow = image_stack(bwd_flow, "image stack"), s_w = image_stack(weights, "image stack");
          lay[from_depth ? 0 : 1] = s_src.lay, lay[2] = s_flow.lay, lay[3] = s_w.lay;
          any_view = s_src.is_view() || s_flow.is_view() || s_w.is_view();
        }
        if (any_view)
          FM_CALL(fm_procrustes_scatter_views, from_depth ? ptr(src) : nullptr, ptr(kinv), from_depth ? nullptr : ptr(src), ptr(bwd_flow), ptr(weights),
                  sens, ptr<int64_t>(indices), (long)points, (int)b, (int)f, (int)h, (int)w, ptr<double>(aux), ptr<double>(pair_grad),
                  from_depth ? ptr(g_src) : nullptr, from_depth ? nullptr : ptr(g_src), ptr(g_w), per_point_k, ptr(point_grads), lay, scope.stream);
        else
          FM_CALL(fm_procrustes_scatter, from_depth ? ptr(src) : nullptr, ptr(kinv), from_depth ? nullptr : ptr(src), ptr(bwd_flow), ptr(weights),
                  sens, ptr<int64_t>(indices), (long)points, (int)b, (int)rep, (int)f, (int)h, (int)w, ptr<double>(aux), ptr<double>(pair_grad),
                  from_depth ? ptr(g_src) : nullptr, from_depth ? nullptr : ptr(g_src), ptr(g_w), per_point_k, ptr(point_grads),
                  nullptr, scope.stream);
      }
      // dL/dK = −K⁻ᵀ·dK⁻¹·K⁻ᵀ, added to the flow loss's own dL/dK when that was parked here (one gradient for autograd,
      // no separate add); with a planned scatter it rides in the gather's launch
      const bool add_k = need_k && carried_k.defined() && carried_k.sizes() == kinv.sizes() && carried_k.is_contiguous();
      if (need_k) g_k = add_k ? carried_k : at::empty_like(kinv);
      if (point_grads.defined()) {  // the planned scatter runs as a gather: one plain read-modify-write per touched pixel
        FM_CALL(fm_depth_gather_kgrad, ptr(point_grads), ptr<int64_t>(plan_pixels), ptr<int32_t>(plan_first), ptr<int32_t>(plan_vectors),
                ptr(plan_weights), (long)plan_pixels.numel(), ptr(kinv), (int)h, (int)w, ptr(g_src), ptr<double>(kinv_acc),
                need_k ? (int)(b * f) : 0, ptr(g_k), add_k ? 1 : 0, scope.stream);
        if (sink) ++sink->planned_steps;
      } else if (need_k) {
        FM_CALL(fm_intrinsics_inverse_bwd, ptr<double>(kinv_acc), ptr(kinv), (int)(b * f), ptr(g_k), add_k ? 1 : 0, scope.stream);
      }
      if (carried_k.defined() && !add_k) g_k = g_k.defined() ? g_k + carried_k : carried_k;  // (unexpected layout: plain sum)
    }
    if (arena_used) g_w = arena->alias();
    if (sink) sink->note_final(g_src);
    if (wsink) {
      wsink->note_final(g_w);
      if (arena_used) wsink->on_leading_add = [arena](const Tensor& into, int64_t count) { arena->note_leading_add(into, count); };
    }
    variable_list out(23);
    if (from_depth) {
      out[0] = g_src;
      out[1] = g_k;
    } else {
      out[3] = g_src;
    }
    out[4] = g_w;
    return out;
  }
};

// ------------------------------------------------------------------------------------------
// Fused flow loss: weight · LossFlow.compute_unweighted_loss (loss_flow.py:31-70, loss.py:47) from depth +
// intrinsics + relative poses, with the analytic gradient of every input produced in the same HBM pass.
// The gradients are computed in forward (one pass over 4.4 GB at C1) and handed out by backward; a second
// backward through a retained graph recomputes them (one more launch) from the saved inputs.
// ------------------------------------------------------------------------------------------
static std::mutex& timing_mutex() {
  static std::mutex m;
  return m;
}
struct FlowTimings {
  bool enabled = false;
  std::vector<std::pair<void*, void*>> events;        // hipEvent_t pairs around every fused flow kernel launch
  std::vector<std::pair<void*, void*>> track_events;  // ... and around every fm_track_loss_fwd (track_pairs + its reduction)
  void* (*create)() = nullptr;
  void (*record)(void*, void*) = nullptr;
  float (*elapsed)(void*, void*) = nullptr;
};
static FlowTimings& flow_timings() {
  static FlowTimings t;
  return t;
}

struct FlowLaunch {
  Tensor loss, g_depth, small, g_tf, g_tb, g_k;
};
// the tap exchange of one launch (fm_flow_taps): the static mask / rank table, what a tracking loss offered, where the tap depths go
struct FlowTapArgs {
  Tensor chunk_base, pixel, grad, scale, depth_out, stale;
  bool on() const { return chunk_base.defined(); }
};

static FlowLaunch flow_launch(const Tensor& depth, const Tensor& k, const Tensor& kinv, const Tensor& t_fwd, const Tensor& t_bwd,
                              const Tensor& flow_fwd, const Tensor& flow_bwd, const Tensor& mask_fwd, const Tensor& mask_bwd, const Tensor& norm,
                              const Tensor& packed, int64_t kind, double delta, int64_t items, bool need, bool need_depth, const Tensor& acc_work,
                              const Tensor& exp_avg = Tensor(), const Tensor& exp_avg_sq = Tensor(), const Tensor& touched = Tensor(),
                              int64_t adam_step = 0, const std::vector<double>& adam = {}, const FlowTapArgs& taps = FlowTapArgs()) {
  const int64_t b = depth.size(0), f = depth.size(1), h = depth.size(2), w = depth.size(3);
  const auto dev = depth.device();
  FlowLaunch o;
  // the per-(frame, direction) sums: a persistent workspace the finalize launch leaves zero, or fresh zeros
  const bool persistent = acc_work.defined() && acc_work.scalar_type() == at::kDouble && acc_work.is_contiguous() &&
                          acc_work.numel() == b * f * 2 * FM_FLOW_ACC_STRIDE && acc_work.device() == dev;
  Tensor acc = persistent ? acc_work : at::zeros({b * f * 2 * FM_FLOW_ACC_STRIDE}, depth.options().dtype(at::kDouble));
  o.loss = at::empty({1}, depth.options());
  if (need && need_depth) o.g_depth = at::empty(depth.sizes(), depth.options());  // (dense, whatever the strides of a depth window)
  // the three small gradients share one allocation so one launch rescales them in backward
  o.small = at::empty({2 * t_fwd.numel() + k.numel()}, depth.options());
  o.g_tf = o.small.narrow(0, 0, t_fwd.numel()).view_as(t_fwd);
  o.g_tb = o.small.narrow(0, t_fwd.numel(), t_fwd.numel()).view_as(t_bwd);
  o.g_k = o.small.narrow(0, 2 * t_fwd.numel(), k.numel()).view_as(k);
  const float scale = std::sqrt((float)(h * w));
  DeviceScope scope(dev);
  FlowTimings& tm = flow_timings();
  void *e0 = nullptr, *e1 = nullptr;
  if (tm.enabled && dev.is_cuda() && tm.create) {
    e0 = tm.create();
    e1 = tm.create();
    tm.record(e0, scope.stream);
  }
  // `acc` is self-cleaning (the finalize launch leaves it zero): if anything between the two launches fails, a persistent
  // workspace would carry dirty sums into every later step — zero it before the error leaves this function
  struct AccGuard {
    Tensor acc;
    bool armed;
    ~AccGuard() {
      if (armed && acc.defined()) {
        try {
          acc.zero_();
        } catch (...) {
        }
      }
    }
  } acc_guard{persistent ? acc : Tensor(), true};
  const bool pk = packed.defined();  // (the originals may then be placeholders without storage of their own)
  fm_layout lay[5] = {};
  bool any_view = false;
  {
    const Tensor* stacks[5] = {&depth, &flow_fwd, &flow_bwd, &mask_fwd, &mask_bwd};
    for (int i = 0; i < (pk ? 1 : 5); ++i) {
      const ImageStack st = image_stack(*stacks[i], "image stack");  // (already in place or copied by the caller: no copy happens here)
      lay[i] = st.lay;
      any_view = any_view || st.is_view();
    }
  }
  const float *p_ff = pk ? nullptr : ptr(flow_fwd), *p_fb = pk ? nullptr : ptr(flow_bwd), *p_mf = pk ? nullptr : ptr(mask_fwd),
              *p_mb = pk ? nullptr : ptr(mask_bwd);
  if (taps.on()) {  // the tap exchange with the tracking loss, with or without the in-pass Adam update (fm_flow_loss_fused_taps)
    TORCH_CHECK(!any_view, "flowmap_amd: the tap exchange reads dense image stacks (the caller hands frame windows to the plain pass)");
    const fm_flow_taps t{ptr<int32_t>(taps.chunk_base), ptr<int32_t>(taps.pixel), ptr(taps.grad), ptr(taps.scale), ptr(taps.depth_out),
                         exp_avg.defined() ? nullptr : ptr<int32_t>(taps.stale)};
    const bool ad = exp_avg.defined();
    FM_CALL(fm_flow_loss_fused_taps, ptr(depth), ptr(k), ptr(kinv), ptr(t_fwd), ptr(t_bwd), p_ff, p_fb, p_mf, p_mb, ptr(packed), ptr(norm), (int)b,
            (int)f, (int)h, (int)w, (int)kind, (float)delta, (float)w / scale, (float)h / scale, ptr(o.g_depth), ptr<double>(acc), (int)items, &t,
            ptr(exp_avg), ptr(exp_avg_sq), ptr<uint8_t>(touched), (long)adam_step, ad ? adam[0] : 0.0, ad ? adam[1] : 0.0, ad ? adam[2] : 0.0,
            ad ? adam[3] : 0.0, scope.stream);
  } else if (exp_avg.defined()) {  // the depth parameter's Adam update applied by the same pass (fm_flow_loss_fused_adam)
    FM_CALL(fm_flow_loss_fused_adam, ptr(depth), ptr(k), ptr(kinv), ptr(t_fwd), ptr(t_bwd), p_ff, p_fb, p_mf,
            p_mb, ptr(packed), ptr(norm), (int)b, (int)f, (int)h, (int)w, (int)kind, (float)delta, (float)w / scale, (float)h / scale,
            ptr(o.g_depth), ptr<double>(acc), (int)items, ptr(exp_avg), ptr(exp_avg_sq), ptr<uint8_t>(touched), (long)adam_step, adam[0], adam[1],
            adam[2], adam[3], scope.stream);
  } else if (any_view) {
    FM_CALL(fm_flow_loss_fused_views, ptr(depth), ptr(k), ptr(kinv), ptr(t_fwd), ptr(t_bwd), p_ff, p_fb, p_mf, p_mb,
            ptr(packed), need ? ptr(norm) : nullptr, (int)b, (int)f, (int)h, (int)w, (int)kind, (float)delta, (float)w / scale, (float)h / scale,
            ptr(o.g_depth), ptr<double>(acc), (int)items, lay, scope.stream);
  } else {
    FM_CALL(fm_flow_loss_fused, ptr(depth), ptr(k), ptr(kinv), ptr(t_fwd), ptr(t_bwd), p_ff, p_fb, p_mf, p_mb,
            ptr(packed), need ? ptr(norm) : nullptr, (int)b, (int)f, (int)h, (int)w, (int)kind, (float)delta, (float)w / scale, (float)h / scale,
            ptr(o.g_depth), ptr<double>(acc), (int)items, scope.stream);
  }
  if (e0) {
    tm.record(e1, scope.stream);
    std::lock_guard<std::mutex> lock(timing_mutex());
    tm.events.emplace_back(e0, e1);
  }
  FM_CALL(fm_flow_loss_finalize, ptr<double>(acc), ptr(k), ptr(kinv), ptr(t_fwd), ptr(t_bwd), ptr(norm), (int)b, (int)f, (float)w / scale,
          (float)h / scale, ptr(o.loss), ptr(o.g_tf), ptr(o.g_tb), ptr(o.g_k), scope.stream);
  acc_guard.armed = false;
  return o;
}

struct FlowLossFused : public Function<FlowLossFused> {
  static Tensor forward(AutogradContext* ctx, const Tensor& depth_in, const Tensor& k_in, const Tensor& kinv_in, const Tensor& t_fwd_in,
                        const Tensor& t_bwd_in, const Tensor& flow_fwd_in, const Tensor& flow_bwd_in, const Tensor& mask_fwd_in,
                        const Tensor& mask_bwd_in, const Tensor& norm, const OptTensor& packed_o, int64_t kind, double delta,
                        const c10::intrusive_ptr<DepthSink>& sink, int64_t items, const OptTensor& acc_work, const OptTensor& exp_avg_o,
                        const OptTensor& exp_avg_sq_o, const OptTensor& touched_o, int64_t adam_step, std::vector<double> adam,
                        const OptTensor& adam_flag_o, const OptTensor& tap_chunk_base_o, const OptTensor& tap_pixel_o, const OptTensor& tap_depth_o,
                        const OptTensor& tap_stale_o, bool tap_require, bool grad_enabled, bool park) {
    check_device({&depth_in, &k_in, &kinv_in, &t_fwd_in, &t_bwd_in, &flow_fwd_in, &flow_bwd_in, &mask_fwd_in, &mask_bwd_in, &norm});
    TORCH_CHECK(flow_fwd_in.scalar_type() == at::kFloat && flow_bwd_in.scalar_type() == at::kFloat && mask_fwd_in.scalar_type() == at::kFloat &&
                    mask_bwd_in.scalar_type() == at::kFloat,
                "flowmap_amd: flows and masks must be float32");
    const Tensor depth = image_stack(depth_in, "depth").t, k = f32c(k_in, "intrinsics"), kinv = f32c(kinv_in, "inverse intrinsics");
    const Tensor t_fwd = f32c(t_fwd_in, "forward poses"), t_bwd = f32c(t_bwd_in, "backward poses");
    // With a packed copy (fm_flow_pack_inputs) the kernel reads that and nothing else: the four originals are only checked for
    // their shapes — they may be storage-free placeholders (release_flow_originals) — and reach the C ABI as NULL
    const bool has_packed = packed_o.has_value() && packed_o->defined();
    // (frame windows of larger tensors are read in place: flow_launch hands their strides to fm_flow_loss_fused_views)
    const Tensor flow_fwd = has_packed ? flow_fwd_in : image_stack(flow_fwd_in, "forward flow").t, flow_bwd = has_packed ? flow_bwd_in : image_stack(flow_bwd_in, "backward flow").t;
    const Tensor mask_fwd = has_packed ? mask_fwd_in : image_stack(mask_fwd_in, "forward mask").t, mask_bwd = has_packed ? mask_bwd_in : image_stack(mask_bwd_in, "backward mask").t;
    TORCH_CHECK(depth.dim() == 4, "flowmap_amd: depth must be (batch, frame, height, width)");
    const int64_t b = depth.size(0), f = depth.size(1), h = depth.size(2), w = depth.size(3);
    // the C ABI's hard limits (include/flowmap_hip.h: fm_flow_loss_fused), named here instead of surfacing as "invalid argument"
    TORCH_CHECK(b * f <= 65535, "flowmap_amd: the fused flow loss handles at most 65 535 source frames per call (batch x frames = ", b * f,
                "): split the batch");
    TORCH_CHECK(h * w < (int64_t(1) << 30), "flowmap_amd: the fused flow loss indexes pixels inside a frame with 32 bits: height x width = ", h * w,
                " must stay below 2^30 (frames x height x width is not limited)");
    TORCH_CHECK(flow_fwd.sizes() == at::IntArrayRef({b, f - 1, h, w, 2}) && flow_bwd.sizes() == flow_fwd.sizes(),
                "flowmap_amd: flow shape does not match depth");
    TORCH_CHECK(mask_fwd.sizes() == at::IntArrayRef({b, f - 1, h, w}) && mask_bwd.sizes() == mask_fwd.sizes(),
                "flowmap_amd: mask shape does not match depth");
    TORCH_CHECK(k.sizes() == at::IntArrayRef({b, f, 3, 3}) && kinv.sizes() == k.sizes() && t_fwd.sizes() == at::IntArrayRef({b, f - 1, 4, 4}) &&
                    t_bwd.sizes() == t_fwd.sizes(),
                "flowmap_amd: intrinsics / pose shapes do not match depth");
    TORCH_CHECK(norm.scalar_type() == at::kFloat && norm.numel() >= 1, "flowmap_amd: the normaliser must be a float32 device tensor");
    TORCH_CHECK(!flow_fwd_in.requires_grad() && !flow_bwd_in.requires_grad(), "flowmap_amd: gradients w.r.t. optical flow are not supported (flows are constants)");
    Tensor packed = opt(packed_o);
    if (packed.defined())
      TORCH_CHECK(packed.scalar_type() == at::kFloat && packed.is_contiguous() && w % 4 == 0 &&
                      packed.sizes() == at::IntArrayRef({b * f, (h * w / 4 + 63) / 64, 6, 64, 4}),
                  "flowmap_amd: packed flow inputs do not match the depth shape");
    const bool need = grad_enabled && (depth_in.requires_grad() || k_in.requires_grad() || t_fwd_in.requires_grad() || t_bwd_in.requires_grad());
    // In-pass Adam (FusedAdam.fuse_depth_update): depth itself is rewritten, so it must be the caller's memory, not a copy;
    // the update assumes the gradient is final as computed, i.e. that the loss reaches backward() unscaled
    const Tensor exp_avg = opt(exp_avg_o), exp_avg_sq = opt(exp_avg_sq_o), touched = opt(touched_o);
    const bool in_pass_adam = exp_avg.defined();
    if (in_pass_adam) {
      TORCH_CHECK(need && park && depth_in.requires_grad() && depth.data_ptr() == depth_in.data_ptr() && depth.is_contiguous() && w % 4 == 0 && adam.size() == 4 && adam_step >= 1,
                  "flowmap_amd: the in-pass Adam update needs a contiguous float32 depth parameter whose gradient travels through the step's DepthSink");
      TORCH_CHECK(exp_avg.sizes() == depth.sizes() && exp_avg_sq.sizes() == depth.sizes() && exp_avg.is_contiguous() && exp_avg_sq.is_contiguous() &&
                      exp_avg.scalar_type() == at::kFloat && exp_avg_sq.scalar_type() == at::kFloat && touched.defined() &&
                      touched.scalar_type() == at::kByte && touched.is_contiguous() && touched.numel() * 4 == depth.numel(),
                  "flowmap_amd: Adam state / touched-pixel mask do not match the depth tensor");
    }
    // The tap exchange: the static tap set of the tracking loss (mask + rank table from the caller), its offered gradient (from the sink, when
    // a tracking loss of this step ran before this pass and the dense gradient is parked with the fit, whose backward settles the account)
    // and the compact image the tap depths are left in
    FlowTapArgs taps;
    if (tap_chunk_base_o.has_value() && tap_chunk_base_o->defined() && need && depth_in.requires_grad() && depth.is_contiguous() && w % 4 == 0 &&
        depth.data_ptr() == depth_in.data_ptr()) {
      taps.chunk_base = *tap_chunk_base_o;
      taps.pixel = opt(tap_pixel_o);
      taps.depth_out = opt(tap_depth_o);
      taps.stale = opt(tap_stale_o);
      TORCH_CHECK(!taps.stale.defined() || (taps.stale.scalar_type() == at::kInt && taps.stale.numel() == 1 && taps.stale.device() == depth.device()),
                  "flowmap_amd: the stale-image flag is one int32 on the depth tensor's device");
      const int64_t chunks = b * f * ((h * w / 4 + 63) / 64);
      TORCH_CHECK(taps.chunk_base.scalar_type() == at::kInt && taps.chunk_base.is_contiguous() && taps.chunk_base.numel() == chunks + 1 &&
                      taps.pixel.defined() && taps.pixel.scalar_type() == at::kInt && taps.pixel.is_contiguous() &&
                      taps.chunk_base.device() == depth.device() && taps.pixel.device() == depth.device(),
                  "flowmap_amd: the tap rank table / pixel list do not match the depth tensor");
      TORCH_CHECK(!taps.depth_out.defined() || (taps.depth_out.scalar_type() == at::kFloat && taps.depth_out.is_contiguous() &&
                                                taps.depth_out.device() == depth.device() && taps.depth_out.numel() == taps.pixel.numel()),
                  "flowmap_amd: the compact tap image is a contiguous float32 tensor of one value per tap on the depth tensor's device");
      if (sink && park && sink->accepts(depth) && sink->offers_taps() &&
          taps.pixel.numel() == sink->tap_grad.numel()) {
        taps.grad = sink->tap_grad;
        taps.scale = sink->tap_scale;
      }
    }
    // (the caller left the tracking loss's taps to this pass's in-pass Adam update: it must really absorb their gradient — checked BEFORE the
    // launch, which rewrites the parameter and the optimiser state in place)
    TORCH_CHECK(!tap_require || taps.grad.defined(), "flowmap_amd: the flow pass was to absorb the tracking loss's gradient at its taps (tap exchange + in-pass "
                "Adam) but the step's DepthSink offers none that fits");
    FlowLaunch run = flow_launch(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, norm, packed, kind, delta, items, need,
                                 depth_in.requires_grad(), opt(acc_work), exp_avg, exp_avg_sq, touched, adam_step, adam, taps);
    if (taps.grad.defined()) {
      sink->tap_absorbed = true;
      sink->tap_adam_flag = (in_pass_adam && adam_flag_o.has_value() && adam_flag_o->defined()) ? *adam_flag_o : Tensor();
    }
    ctx->saved_data["tap_absorbed"] = taps.grad.defined();
    ctx->saved_data["in_pass_adam"] = in_pass_adam;
    if (in_pass_adam && adam_flag_o.has_value() && adam_flag_o->defined()) {
      TORCH_CHECK(adam_flag_o->scalar_type() == at::kInt && adam_flag_o->numel() == 1 && adam_flag_o->device() == depth.device(),
                  "flowmap_amd: the in-pass Adam flag is one int32 on the depth tensor's device");
      ctx->saved_data["adam_flag"] = *adam_flag_o;
    }
    ctx->save_for_backward({depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, norm, packed});
    ctx->saved_data["acc_work"] = opt(acc_work);
    ctx->saved_data["cfg"] = std::vector<int64_t>{kind, items};
    ctx->saved_data["delta"] = delta;
    if (sink && park) ctx->saved_data["sink"] = sink;
    ctx->saved_data["g_depth"] = run.g_depth;
    if (sink && sink->want_early && run.g_depth.defined()) {
      // (the same memory backward hands on: the caller copies the frames it sends.  With the in-pass Adam update only the pixels the pass
      // keeps are written — the frames shared with a neighbour are kept whole, and they are all the caller reads)
      sink->early_dense = run.g_depth;
      if (sink->unit_flag.defined() && !in_pass_adam) ctx->saved_data["adam_flag"] = sink->unit_flag;  // (in-pass: the optimiser's own flag watches the same condition)
    }
    ctx->saved_data["small"] = run.small;
    ctx->saved_data["fresh"] = need;
    return run.loss.reshape({});
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    variable_list out(29);
    if (!grads[0].defined()) return out;
    const auto saved = ctx->get_saved_variables();
    const Tensor &depth = saved[0], &k = saved[1], &t_fwd = saved[3];
    Tensor g_depth, small;
    const bool delivers_taps = ctx->saved_data["fresh"].toBool() && ctx->saved_data["tap_absorbed"].toBool();
    if (ctx->saved_data["fresh"].toBool()) {
      if (ctx->saved_data["g_depth"].isTensor()) g_depth = ctx->saved_data["g_depth"].toTensor();
      small = ctx->saved_data["small"].toTensor();
      ctx->saved_data["fresh"] = false;  // the buffers are scaled in place below and belong to autograd afterwards
      ctx->saved_data["g_depth"] = Tensor();
      ctx->saved_data["small"] = Tensor();
    } else {  // a second backward through a retained graph: one more pass from the saved inputs
      TORCH_CHECK(!ctx->saved_data["in_pass_adam"].toBool(), "flowmap_amd: a step whose flow loss already applied the Adam update cannot be differentiated twice");
      const auto cfg = ctx->saved_data["cfg"].toIntVector();
      FlowLaunch run = flow_launch(saved[0], saved[1], saved[2], saved[3], saved[4], saved[5], saved[6], saved[7], saved[8], saved[9], saved[10],
                                   cfg[0], ctx->saved_data["delta"].toDouble(), cfg[1], true, ctx->needs_input_grad(0),
                                   ctx->saved_data["acc_work"].isTensor() ? ctx->saved_data["acc_work"].toTensor() : Tensor());
      g_depth = run.g_depth;
      small = run.small;
    }
    TORCH_CHECK(small.defined(), "flowmap_amd: the flow loss was evaluated without gradients (no input required grad)");
    const bool unit = is_unit_seed(grads[0]);  // the step's own root seed (RootLoss.backward): 1 by construction, nothing to rescale
    const Tensor g = grads[0].reshape({1}).to(at::kFloat).contiguous();
    if (unit) {
      ++unit_seed_uses();
    } else {
      DeviceScope scope(g.device());
      // (with the in-pass Adam update the gradient has ALREADY been used, unscaled: a scalar other than 1 raises the caller's flag)
      Tensor flag = ctx->saved_data.count("adam_flag") ? ctx->saved_data["adam_flag"].toTensor() : Tensor();
      FM_CALL(fm_scale_if_needed, ptr(g_depth), g_depth.defined() ? (long)g_depth.numel() : 0L, ptr(small), (long)small.numel(), ptr(g),
              ptr<int>(flag), scope.stream);
    }
    auto sink = ctx->saved_data.count("sink") ? ctx->saved_data["sink"].toCustomClass<DepthSink>() : c10::intrusive_ptr<DepthSink>();
    if (sink && ctx->saved_data["in_pass_adam"].toBool()) sink->in_pass_confirmed = true;
    if (sink && delivers_taps) sink->tap_flow_upstream = g;  // the absorbed tap gradient travels with this buffer, multiplied by g
    if (sink && g_depth.defined() && sink->accepts(depth) && !sink->carried.defined()) {
      sink->carried = g_depth;  // returned (summed with the sparse parts) by the Procrustes fit's node, which runs later
      g_depth = Tensor();
    }
    const int64_t nt = t_fwd.numel();
    if (ctx->needs_input_grad(0)) out[0] = g_depth;
    if (ctx->needs_input_grad(1)) {
      Tensor g_k = small.narrow(0, 2 * nt, k.numel()).view_as(k);
      if (sink && sink->active && sink->k_ptr == k.data_ptr() && !sink->carried_k.defined()) sink->carried_k = g_k;  // the fit adds its part and returns the sum
      else out[1] = g_k;
    }
    if (ctx->needs_input_grad(3)) out[3] = small.narrow(0, 0, nt).view_as(t_fwd);
    if (ctx->needs_input_grad(4)) out[4] = small.narrow(0, nt, nt).view_as(t_fwd);
    return out;
  }
};

// ------------------------------------------------------------------------------------------
// Fused tracking loss: weight · LossTracking.compute_unweighted_loss (loss_tracking.py:28-61, loss.py:47) over
// all segments, from depth + intrinsics + extrinsics.  The track arrays are the packed form the Python layer
// builds once per track set (PackedTracks); `depth` may be a window of the video starting at frame0 (frame
// sharding) while k / ext cover the whole video.  Outputs: loss (differentiable), scale = [weight/max(count,1),
// count] and totals = [Σρ, count] (fp64) — a sharded caller all-reduces totals and overwrites `scale` in place
// before backward; the gradients follow the scale they find.
// ------------------------------------------------------------------------------------------
struct TrackArrays {
  Tensor xy, vis, seg, blocks, tiles;
  int64_t nblocks, ntiles, pmax, fmax, total, partial;
};

struct TrackLossFused : public Function<TrackLossFused> {
  static variable_list forward(AutogradContext* ctx, const Tensor& depth_in, const Tensor& k_in, const Tensor& kinv_in, const Tensor& ext_in,
                               const Tensor& xy, const Tensor& vis, const Tensor& seg, const Tensor& blocks, const Tensor& tiles,
                               std::vector<int64_t> counts, double weight, int64_t kind, double delta,
                               const c10::intrusive_ptr<DepthSink>& sink, int64_t frame0, const OptTensor& plan_pixels,
                               const OptTensor& plan_first, const OptTensor& plan_entries, const OptTensor& plan_weights, const OptTensor& tap_slot_o,
                               const OptTensor& tap_depth_o, const OptTensor& tap_shared_o, bool offer_taps, bool grad_enabled, bool park) {
    const auto dev = check_device({&depth_in, &k_in, &kinv_in, &ext_in, &xy});
    const Tensor depth = f32c(depth_in, "depth"), k = f32c(k_in, "intrinsics"), kinv = f32c(kinv_in, "inverse intrinsics"),
                 ext = f32c(ext_in, "extrinsics");
    TORCH_CHECK(counts.size() == 9, "flowmap_amd: packed track counts");
    const int64_t ntiles = counts[1], pmax = counts[2], fmax = counts[3], total = counts[4], partial = counts[5], last_frame = counts[6];
    TORCH_CHECK(depth.dim() == 4 && depth.size(0) == 1, "flowmap_amd: the fused tracking loss supports batch size 1 (as the reference asserts)");
    const int64_t f_local = depth.size(1), h = depth.size(2), w = depth.size(3), f = ext.size(1);
    TORCH_CHECK(k.sizes() == at::IntArrayRef({1, f, 3, 3}) && kinv.sizes() == k.sizes() && ext.sizes() == at::IntArrayRef({1, f, 4, 4}) &&
                    frame0 >= 0 && frame0 + f_local <= f,
                "flowmap_amd: intrinsics / extrinsics must cover the whole video and depth a window of it");
    TORCH_CHECK(last_frame <= f, "flowmap_amd: a track segment extends past the last frame");
    const int64_t own_first = std::max<int64_t>(counts[7], frame0), own_end = std::min<int64_t>(counts[8] < 0 ? f : counts[8], frame0 + f_local);
    const auto fopt = depth.options();
    Tensor ext_inv = at::empty_like(ext);
    Tensor ws = at::empty({total, 9}, fopt);
    Tensor flag = partial ? at::zeros({total}, fopt.dtype(at::kByte)) : at::empty({total}, fopt.dtype(at::kByte));
    Tensor acc = at::empty({f * 20}, fopt.dtype(at::kDouble));
    Tensor loss = at::empty({1}, fopt), scale = at::empty({2}, fopt), totals = at::empty({2}, fopt.dtype(at::kDouble));
    const bool need = (depth_in.requires_grad() || k_in.requires_grad() || ext_in.requires_grad()) && grad_enabled;
    // every residual is evaluated once: the (unscaled) gradients come out of the same launch
    Tensor gws = need ? at::empty({total, 3}, fopt) : Tensor();
    Tensor acc2 = need ? at::empty({f * 24}, fopt.dtype(at::kDouble)) : Tensor();
    Tensor tgt = at::empty({f, 12}, fopt);
    Tensor part = at::empty({std::max<int64_t>(ntiles, 1) * ((pmax + 63) / 64) * (fmax * 14 + FM_TRACK_TILE * 21)}, fopt);
    // The tap exchange (fm_flow_taps): sample from the compact tap image the last flow pass left (tap_slot + tap_depth, when the caller
    // vouches that depth has not moved since), and / or compact this loss's dL/ddepth at the taps for the flow pass that follows (offer_taps).
    // Whole video local (no frame sharding), gradients on, the fit's node downstream.
    const bool whole = frame0 == 0 && f_local == f && !partial && counts[7] == 0 && (counts[8] < 0 || counts[8] >= f);
    const Tensor tap_slot = (whole && tap_slot_o.has_value() && tap_slot_o->defined()) ? *tap_slot_o : Tensor();
    const Tensor tap_depth = (tap_slot.defined() && tap_depth_o.has_value() && tap_depth_o->defined()) ? *tap_depth_o : Tensor();
    const Tensor tap_shared = (tap_slot.defined() && tap_shared_o.has_value() && tap_shared_o->defined()) ? *tap_shared_o : Tensor();
    if (tap_slot.defined())
      TORCH_CHECK(tap_slot.scalar_type() == at::kInt && tap_slot.is_contiguous() && tap_slot.numel() == total * 4 && tap_slot.device() == depth.device() &&
                      (!tap_depth.defined() || (tap_depth.scalar_type() == at::kFloat && tap_depth.is_contiguous() && tap_depth.device() == depth.device())) &&
                      (!tap_shared.defined() || (tap_shared.scalar_type() == at::kInt && tap_shared.is_contiguous() && tap_shared.device() == depth.device())),
                  "flowmap_amd: tap slots (total, 4) int32, a float32 compact tap image and an int32 list of shared taps on the depth tensor's device");
    const bool offer = offer_taps && whole && need && depth_in.requires_grad() && ntiles > 0 && sink && park && sink->accepts(depth) &&
                       plan_pixels.has_value() && plan_pixels->defined() && plan_pixels->numel() > 0;
    Tensor tap_grad = offer ? at::empty({plan_pixels->numel()}, fopt) : Tensor();
    const bool use_taps = tap_depth.defined() || tap_grad.defined();
    const float sc = std::sqrt((float)(h * w));
    {
      DeviceScope scope(dev);
      FM_CALL(fm_extrinsics_inverse, ptr(ext), (int)f, ptr(ext_inv), scope.stream);
      if (ntiles > 0) {
        FlowTimings& tm = flow_timings();
        void *e0 = nullptr, *e1 = nullptr;
        if (tm.enabled && dev.is_cuda() && tm.create) {
          e0 = tm.create();
          e1 = tm.create();
          tm.record(e0, scope.stream);
        }
        if (use_taps) {
          FM_CALL(fm_track_loss_fused_fwd_taps, ptr(depth), ptr(kinv), ptr(ext), ptr(ext_inv), ptr(k), (int)f, ptr(xy), ptr<uint8_t>(vis),
                  ptr<int32_t>(seg), ptr<int32_t>(tiles), (int)ntiles, (int)pmax, (int)fmax, (int)h, (int)w, (int)kind, (float)delta, (float)w / sc,
                  (float)h / sc, (float)weight, ptr(ws), ptr<uint8_t>(flag), ptr(tgt), ptr(part), ptr<double>(acc), ptr(loss), ptr(scale),
                  ptr<double>(totals), ptr(gws), ptr<double>(acc2), ptr<int32_t>(tap_slot), ptr(tap_depth), ptr<int64_t>(opt(plan_pixels)),
                  ptr<int32_t>(opt(plan_first)), ptr<int32_t>(opt(plan_entries)), ptr(opt(plan_weights)),
                  tap_grad.defined() ? (long)tap_grad.numel() : 0L, tap_grad.defined() ? ptr<int32_t>(tap_shared) : nullptr,
                  (tap_grad.defined() && tap_shared.defined()) ? (long)tap_shared.numel() : 0L,
                  ptr(tap_grad), scope.stream);
        } else {
          FM_CALL(fm_track_loss_fused_fwd, ptr(depth), (int)frame0, (int)own_first, (int)own_end, ptr(kinv), ptr(ext), ptr(ext_inv), ptr(k), (int)f,
                  ptr(xy), ptr<uint8_t>(vis), ptr<int32_t>(seg), ptr<int32_t>(tiles), (int)ntiles, (int)pmax, (int)fmax, (int)h, (int)w, (int)kind,
                  (float)delta, (float)w / sc, (float)h / sc, (float)weight, ptr(ws), ptr<uint8_t>(flag), ptr(tgt), ptr(part), ptr<double>(acc),
                  ptr(loss), ptr(scale), ptr<double>(totals), ptr(gws), ptr<double>(acc2), scope.stream);
        }
        if (e0) {
          tm.record(e1, scope.stream);
          std::lock_guard<std::mutex> lock(timing_mutex());
          tm.track_events.emplace_back(e0, e1);
        }
      }
    }
    if (ntiles == 0) {  // this rank owns no source frame of any segment
      acc.zero_();
      totals.zero_();
      loss.zero_();
      scale.zero_();
      scale.narrow(0, 0, 1).fill_(weight);
      if (need) acc2.zero_();
    }
    // (`scale` is deliberately NOT a version-checked saved tensor: a sharded caller overwrites it with the global normaliser)
    ctx->save_for_backward({k, kinv, ext_inv, acc, opt(plan_pixels), opt(plan_first), opt(plan_entries), opt(plan_weights), depth});
    ctx->saved_data["scale"] = scale;
    ctx->saved_data["tap_grad"] = tap_grad;
    if (tap_grad.defined()) sink->offer_taps(tap_grad, scale, *plan_pixels);
    ctx->saved_data["gws"] = gws;
    ctx->saved_data["acc2"] = acc2;
    ctx->saved_data["dims"] = std::vector<int64_t>{f, h, w, frame0};
    if (sink && park) ctx->saved_data["sink"] = sink;
    ctx->mark_non_differentiable({scale, totals});
    ctx->set_materialize_grads(false);  // (no zeros tensors for the two non-differentiable outputs in every backward: two fill launches per step)
    return {loss.reshape({}), scale, totals};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    variable_list out(25);
    if (!grads[0].defined()) return out;
    const auto saved = ctx->get_saved_variables();
    const Tensor &k = saved[0], &kinv = saved[1], &ext_inv = saved[2], &acc = saved[3];
    const Tensor scale = ctx->saved_data["scale"].toTensor();
    const Tensor &plan_pixels = saved[4], &plan_first = saved[5], &plan_entries = saved[6], &plan_weights = saved[7], &depth = saved[8];
    const Tensor gws = ctx->saved_data["gws"].toTensor(), acc2 = ctx->saved_data["acc2"].toTensor();
    TORCH_CHECK(gws.defined(), "flowmap_amd: the tracking loss was evaluated without gradients (no input required grad)");
    const auto dims = ctx->saved_data["dims"].toIntVector();
    const int64_t f = dims[0], h = dims[1], w = dims[2], frame0 = dims[3];
    const auto dev = kinv.device();
    const Tensor g = grads[0].reshape({1}).to(at::kFloat).contiguous();
    Tensor g_ext = at::empty({1, f, 4, 4}, kinv.options()), g_k = at::empty_like(k);
    {
      DeviceScope scope(dev);
      FM_CALL(fm_track_loss_bwd, ptr<double>(acc), ptr<double>(acc2), ptr(scale), ptr(g), ptr(ext_inv), ptr(k), ptr(kinv), (int)f, ptr(g_ext),
              ptr(g_k), scope.stream);
    }
    // the depth part: a planned gather over the touched pixels (no atomics; gws / scale stay untouched, so a second
    // backward through a retained graph repeats it)
    auto scatter = [=](Tensor& buffer) {
      if (!plan_pixels.defined()) return;
      DeviceScope scope(dev);
      FM_CALL(fm_depth_gather, ptr(gws), ptr<int64_t>(plan_pixels), ptr<int32_t>(plan_first), ptr<int32_t>(plan_entries), ptr(plan_weights),
              (long)plan_pixels.numel(), ptr(kinv), ptr(scale), ptr(g), (int)h, (int)w, (long)frame0, ptr(buffer), scope.stream);
    };
    if (ctx->needs_input_grad(0)) {
      auto sink = ctx->saved_data.count("sink") ? ctx->saved_data["sink"].toCustomClass<DepthSink>() : c10::intrusive_ptr<DepthSink>();
      const Tensor tap_grad = ctx->saved_data["tap_grad"].isTensor() ? ctx->saved_data["tap_grad"].toTensor() : Tensor();
      if (sink && sink->accepts(depth) && tap_grad.defined() && sink->tap_absorbed && sink->tap_grad.defined() &&
          sink->tap_grad.data_ptr() == tap_grad.data_ptr()) {
        // the flow pass of this step added scale·tap_grad into the dense gradient it wrote: nothing is scattered — the fit's node compares
        // the two upstream gradients (DepthSink::settle_taps)
        sink->tap_track_upstream = g;
        sink->tap_confirmed = true;
      } else if (sink && sink->accepts(depth)) {
        sink->pending.push_back(scatter);  // lands in the buffer the Procrustes fit's node returns
      } else {
        Tensor g_depth = at::zeros_like(depth);
        scatter(g_depth);
        out[0] = g_depth;
      }
    }
    if (ctx->needs_input_grad(1)) out[1] = g_k;
    if (ctx->needs_input_grad(3)) out[3] = g_ext;
    return out;
  }
};

// ------------------------------------------------------------------------------------------
// IntrinsicsSoftmin's score + blend (intrinsics_softmin.py:105-141): per candidate the pose-induced backward-flow error at
// the sampled pixels, straight from the images (fm_softmin_score_fwd), then softmin((err − min)·10), the blended K of every
// frame and its inverse (fm_softmin_blend_fwd).  Two launches forward, three backward.
//   depth (B,2,H,W) frames 0/1; weights (B,1,H,W) of pair 0 (logits when weight_sens != 0); bwd_flow (B,1,H,W,2);
//   indices (P) int64 distinct; k (N,3,3) candidates; rel (B·N,4,4) fitted poses later -> earlier.
//   -> K (B,frames,3,3), soft (B,N) [not differentiable: the window's input], K⁻¹ (B,frames,3,3) [not differentiable]
// ------------------------------------------------------------------------------------------
struct SoftminIntrinsics : public Function<SoftminIntrinsics> {
  static variable_list forward(AutogradContext* ctx, const Tensor& depth_in, const Tensor& weights_in, const Tensor& bwd_flow_in,
                               const Tensor& indices_in, const Tensor& k_in, const Tensor& rel_in, double weight_sens, int64_t frames) {
    const auto dev = check_device({&depth_in, &weights_in, &bwd_flow_in, &indices_in, &k_in, &rel_in});
    const Tensor depth = f32c(depth_in, "depth"), weights = f32c(weights_in, "weights"), bwd_flow = f32c(bwd_flow_in, "backward flow");
    const Tensor k = f32c(k_in, "intrinsics"), rel = f32c(rel_in, "poses");
    TORCH_CHECK(depth.dim() == 4 && depth.size(1) == 2, "flowmap_amd: the softmin score expects depth (b,2,h,w)");
    const int64_t b = depth.size(0), h = depth.size(2), w = depth.size(3), n = k.size(0);
    TORCH_CHECK(weights.sizes() == at::IntArrayRef({b, 1, h, w}) && bwd_flow.sizes() == at::IntArrayRef({b, 1, h, w, 2}),
                "flowmap_amd: the softmin score expects weights (b,1,h,w) and backward flow (b,1,h,w,2)");
    TORCH_CHECK(k.sizes() == at::IntArrayRef({n, 3, 3}) && rel.sizes() == at::IntArrayRef({b * n, 4, 4}) && indices_in.scalar_type() == at::kLong &&
                    frames >= 1,
                "flowmap_amd: the softmin score expects intrinsics (n,3,3), poses (b*n,4,4), int64 indices");
    TORCH_CHECK(!k_in.requires_grad() && !bwd_flow_in.requires_grad(), "flowmap_amd: the softmin candidates and the optical flow are constants");
    const Tensor indices = indices_in.contiguous();
    const Tensor kinv = intrinsics_inverse(k);
    Tensor err = at::empty({b * n}, depth.options().dtype(at::kDouble));
    Tensor soft = at::empty({b, n}, depth.options());
    Tensor out = at::empty({b, frames, 3, 3}, depth.options()), kinv_out = at::empty({b, frames, 3, 3}, depth.options());
    {
      DeviceScope scope(dev);
      FM_CALL(fm_softmin_score_fwd, ptr(depth), ptr(weights), (float)weight_sens, ptr(bwd_flow), ptr<int64_t>(indices), (long)indices.numel(), ptr(k),
              ptr(kinv), ptr(rel), (int)b, (int)n, (int)h, (int)w, ptr<double>(err), scope.stream);
      FM_CALL(fm_softmin_blend_fwd, ptr<double>(err), ptr(k), (int)b, (int)n, (int)frames, ptr(soft), ptr(out), ptr(kinv_out), scope.stream);
    }
    ctx->save_for_backward({depth, weights, bwd_flow, indices, k, kinv, rel, soft});
    ctx->saved_data["weight_sens"] = weight_sens;
    ctx->saved_data["frames"] = frames;
    ctx->saved_data["needs"] = std::vector<int64_t>{depth_in.requires_grad(), weights_in.requires_grad(), rel_in.requires_grad()};
    ctx->mark_non_differentiable({soft, kinv_out});
    ctx->set_materialize_grads(false);
    return {out, soft, kinv_out};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    variable_list out(8);
    if (!grads[0].defined()) return out;
    const auto saved = ctx->get_saved_variables();
    const Tensor &depth = saved[0], &weights = saved[1], &bwd_flow = saved[2], &indices = saved[3], &k = saved[4], &kinv = saved[5], &rel = saved[6],
                 &soft = saved[7];
    const auto needs = ctx->saved_data["needs"].toIntVector();
    const int64_t b = depth.size(0), h = depth.size(2), w = depth.size(3), n = k.size(0), frames = ctx->saved_data["frames"].toInt();
    const Tensor g_k = f32c(grads[0], "grad");
    Tensor g_err = at::empty({b * n}, depth.options());
    Tensor g_depth = needs[0] ? at::zeros_like(depth) : Tensor(), g_weights = needs[1] ? at::zeros_like(weights) : Tensor();
    Tensor acc = at::empty({b * n, 12}, depth.options().dtype(at::kDouble)), g_rel = at::empty_like(rel);
    DeviceScope scope(g_k.device());
    FM_CALL(fm_softmin_blend_bwd, ptr(g_k), ptr(soft), ptr(k), (int)b, (int)n, (int)frames, ptr(g_err), scope.stream);
    FM_CALL(fm_softmin_score_bwd, ptr(depth), ptr(weights), (float)ctx->saved_data["weight_sens"].toDouble(), ptr(bwd_flow), ptr<int64_t>(indices),
            (long)indices.numel(), ptr(k), ptr(kinv), ptr(rel), (int)b, (int)n, (int)h, (int)w, ptr(g_err), ptr(g_depth), ptr(g_weights),
            ptr<double>(acc), ptr(g_rel), scope.stream);
    out[0] = g_depth;
    out[1] = g_weights;
    if (needs[2]) out[5] = g_rel;
    return out;
  }
};

// `count` distinct pseudo-random indices of [0, n) (what torch.randperm(n)[:count] is used for): one launch.  With `state` (a
// one-element int64 device tensor) the seed lives in device memory and the call advances it — capturable in a hipGraph.
static Tensor random_subset(int64_t n, int64_t count, c10::Device device, int64_t seed, const OptTensor& state) {
  TORCH_CHECK(count >= 1 && count <= n, "flowmap_amd: random_subset needs 1 <= count <= n");
  Tensor out = at::empty({count}, at::TensorOptions().dtype(at::kLong).device(device));
  const auto dev = check_device({&out});
  DeviceScope scope(dev);
  if (state.has_value() && state->defined()) {
    TORCH_CHECK(state->scalar_type() == at::kLong && state->numel() == 1 && state->device() == out.device(), "flowmap_amd: the sampler state is one int64 on the device");
    FM_CALL(fm_random_subset_stateful, static_cast<unsigned long long*>(state->data_ptr()), (long)n, (long)count, ptr<int64_t>(out), scope.stream);
  } else {
    FM_CALL(fm_random_subset, (unsigned long long)seed, (long)n, (long)count, ptr<int64_t>(out), scope.stream);
  }
  return out;
}

// ------------------------------------------------------------------------------------------
// LeadingFrames: x[:, :count].contiguous() for (b, F, H, W) image stacks.  The softmin sweep reads two of the
// 150 depth frames; autograd's slice backward would zero-fill a full-size tensor and add it densely to the main
// path's gradient (1.7 GB of traffic at C1).  Autograd runs this node's backward AFTER the nodes that consume
// the intrinsics it helped to produce, so when the step's DepthSink says the fit has already returned the dense
// gradient of `x`, the `count` frames are added into that buffer and nothing is returned.
// ------------------------------------------------------------------------------------------
struct LeadingFrames : public Function<LeadingFrames> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, int64_t count, const c10::intrusive_ptr<DepthSink>& sink) {
    TORCH_CHECK(x.dim() == 4 && count >= 1 && count <= x.size(1), "flowmap_amd: LeadingFrames expects (batch, frame, height, width) and 1 <= count <= frame");
    ctx->saved_data["shape"] = x.sizes().vec();
    ctx->saved_data["count"] = count;
    if (sink) ctx->saved_data["sink"] = sink;
    ctx->saved_data["ident"] = std::vector<int64_t>{(int64_t)(uintptr_t)x.data_ptr(), (int64_t)x._version()};
    if (sink) sink->expect_leading = true;
    return x.narrow(1, 0, count).contiguous();
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    if (!grads[0].defined()) return {Tensor(), Tensor(), Tensor()};
    const auto shape = ctx->saved_data["shape"].toIntVector();
    const int64_t count = ctx->saved_data["count"].toInt();
    const auto ident = ctx->saved_data["ident"].toIntVector();
    auto sink = ctx->saved_data.count("sink") ? ctx->saved_data["sink"].toCustomClass<DepthSink>() : c10::intrusive_ptr<DepthSink>();
    const Tensor& g = grads[0];
    Tensor buffer = sink ? sink->take_final() : Tensor();
    if (buffer.defined() && (int64_t)(uintptr_t)sink->depth_ptr == ident[0] && sink->depth_version == ident[1] && buffer.sizes().vec() == shape &&
        buffer.scalar_type() == g.scalar_type() && buffer.device() == g.device()) {
      buffer.narrow(1, 0, count).add_(g);
      if (sink->on_leading_add) sink->on_leading_add(buffer, count);
      sink->on_leading_add = nullptr;
      ++sink->leading_in_place;
      return {Tensor(), Tensor(), Tensor()};
    }
    if (sink) ++sink->leading_dense;
    if (count == shape[1]) return {g, Tensor(), Tensor()};
    Tensor full = at::zeros(shape, g.options());
    full.narrow(1, 0, count).copy_(g);
    return {full, Tensor(), Tensor()};
  }
};

// ------------------------------------------------------------------------------------------
// Adam (model_wrapper_overfit.py:104-105), in place; the version counters are bumped as an in-place ATen op would
// ------------------------------------------------------------------------------------------
static void adam_step(Tensor p, const Tensor& grad_in, Tensor m, Tensor v, int64_t step, const OptTensor& step_tensor, double lr, double beta1,
                      double beta2, double eps, double weight_decay) {
  const auto dev = check_device({&p, &grad_in, &m, &v});
  TORCH_CHECK(p.scalar_type() == at::kFloat && grad_in.scalar_type() == at::kFloat, "flowmap_amd.FusedAdam: parameters and gradients must be float32");
  TORCH_CHECK(p.is_contiguous() && m.is_contiguous() && v.is_contiguous(), "flowmap_amd.FusedAdam: parameters must be contiguous");
  const Tensor grad = grad_in.contiguous();
  DeviceScope scope(dev);
  if (step_tensor.has_value() && step_tensor->defined()) {
    TORCH_CHECK(step_tensor->device() == p.device(), "flowmap_amd.FusedAdam: capturable=True needs the step counter on the parameter's device");
    FM_CALL(fm_adam_step_capturable, ptr(p), ptr(grad), ptr(m), ptr(v), (long)p.numel(), ptr(*step_tensor), lr, beta1, beta2, eps, weight_decay,
            scope.stream);
  } else {
    FM_CALL(fm_adam_step, ptr(p), ptr(grad), ptr(m), ptr(v), (long)p.numel(), (long)step, lr, beta1, beta2, eps, weight_decay, scope.stream);
  }
  p.unsafeGetTensorImpl()->bump_version();
  m.unsafeGetTensorImpl()->bump_version();
  v.unsafeGetTensorImpl()->bump_version();
}

// ------------------------------------------------------------------------------------------
// Kernel timing for bench.py: hipEvents recorded on the launch stream around every fused flow kernel
// ------------------------------------------------------------------------------------------
static void flow_timing_enable(bool on) {
  FlowTimings& tm = flow_timings();
  std::lock_guard<std::mutex> lock(timing_mutex());
  if (on && tm.create == nullptr) {
    // the HIP runtime is already in the process (torch links it); resolve the three calls we need
    using CreateFn = int (*)(void**);
    using RecordFn = int (*)(void*, void*);
    using ElapsedFn = int (*)(float*, void*, void*);
    static CreateFn create = reinterpret_cast<CreateFn>(dlsym(RTLD_DEFAULT, "hipEventCreate"));
    static RecordFn record = reinterpret_cast<RecordFn>(dlsym(RTLD_DEFAULT, "hipEventRecord"));
    static ElapsedFn elapsed = reinterpret_cast<ElapsedFn>(dlsym(RTLD_DEFAULT, "hipEventElapsedTime"));
    TORCH_CHECK(create && record && elapsed, "flowmap_amd: the HIP runtime is not loaded");
    tm.create = []() -> void* {
      void* e = nullptr;
      create(&e);
      return e;
    };
    tm.record = [](void* e, void* stream) { record(e, stream); };
    tm.elapsed = [](void* a, void* b) -> float {
      float ms = 0.f;
      elapsed(&ms, a, b);
      return ms;
    };
  }
  tm.enabled = on;
  tm.events.clear();
  tm.track_events.clear();
}

// milliseconds of every launch recorded since the last call (synchronise the device first)
static std::vector<double> flow_timing_collect(bool tracking) {
  FlowTimings& tm = flow_timings();
  std::lock_guard<std::mutex> lock(timing_mutex());
  auto& events = tracking ? tm.track_events : tm.events;
  std::vector<double> ms;
  for (auto& ev : events) ms.push_back((double)tm.elapsed(ev.first, ev.second));
  events.clear();
  return ms;
}

// ------------------------------------------------------------------------------------------
// Registration
// ------------------------------------------------------------------------------------------
static std::tuple<Tensor, Tensor> focal_intrinsics_op(const Tensor& focal, std::vector<int64_t> repeat_shape, int64_t h, int64_t w) {
  auto out = FocalIntrinsics::apply(focal, repeat_shape, h, w);
  return {out[0], out[1]};
}
static Tensor pose_chain_op(const Tensor& rel) { return PoseChain::apply(rel); }
static std::tuple<Tensor, Tensor> relative_poses_op(const Tensor& ext) {
  auto out = RelativePoses::apply(ext);
  return {out[0], out[1]};
}
using OptSink = std::optional<c10::intrusive_ptr<DepthSink>>;
using OptArena = std::optional<c10::intrusive_ptr<GradArena>>;
static c10::intrusive_ptr<DepthSink> sink_of(const OptSink& s) { return s.has_value() ? *s : c10::intrusive_ptr<DepthSink>(); }

static std::tuple<Tensor, Tensor, Tensor> procrustes_fit_op(const OptTensor& depth, const OptTensor& k, const OptTensor& kinv, const OptTensor& surfaces,
                                                    const Tensor& weights, const Tensor& bwd_flow, const OptTensor& indices, double weight_sens,
                                                    int64_t batch_repeat, const OptSink& sink, const OptSink& wsink, const OptArena& arena,
                                                    const OptTensor& plan_pixels,
                                                    const OptTensor& plan_first, const OptTensor& plan_vectors, const OptTensor& plan_weights,
                                                    const OptTensor& plan_frame_first, const OptTensor& plan_tap_records, const OptTensor& dense_first,
                                                    const OptTensor& dense_list, const OptTensor& work, bool want_ext) {
  auto out = ProcrustesFit::apply(depth, k, kinv, surfaces, weights, bwd_flow, indices, weight_sens, batch_repeat, sink_of(sink), sink_of(wsink),
                                  arena.has_value() ? *arena : c10::intrusive_ptr<GradArena>(), plan_pixels, plan_first, plan_vectors, plan_weights,
                                  plan_frame_first, plan_tap_records, dense_first, dense_list, work, want_ext, at::GradMode::is_enabled());
  if (sink.has_value() && *sink) (*sink)->fit_node = out[0].grad_fn().get();  // null when no graph is being built
  return {out[0], out[1], out[2]};
}
static Tensor flow_loss_op(const Tensor& depth, const Tensor& k, const Tensor& kinv, const Tensor& t_fwd, const Tensor& t_bwd, const Tensor& flow_fwd,
                           const Tensor& flow_bwd, const Tensor& mask_fwd, const Tensor& mask_bwd, const Tensor& norm, const OptTensor& packed,
                           int64_t kind, double delta, const OptSink& sink, int64_t items, const OptTensor& acc_work, const OptTensor& exp_avg,
                           const OptTensor& exp_avg_sq, const OptTensor& touched, int64_t adam_step, std::vector<double> adam,
                           const OptTensor& adam_flag, const OptTensor& tap_chunk_base, const OptTensor& tap_pixel, const OptTensor& tap_depth,
                           const OptTensor& tap_stale, bool tap_require) {
  auto s = sink_of(sink);
  // park dL/ddepth in the sink only when both pose tensors come from the fit that armed it: that node then runs after this one
  const bool park = s && s->fit_node != nullptr && reaches(t_fwd.grad_fn(), s->fit_node, 3) && reaches(t_bwd.grad_fn(), s->fit_node, 3);
  return FlowLossFused::apply(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, norm, packed, kind, delta, s, items, acc_work,
                              exp_avg, exp_avg_sq, touched, adam_step, adam, adam_flag, tap_chunk_base, tap_pixel, tap_depth, tap_stale, tap_require,
                              at::GradMode::is_enabled(), park);
}
static std::tuple<Tensor, Tensor, Tensor> track_loss_op(const Tensor& depth, const Tensor& k, const Tensor& kinv, const Tensor& ext, const Tensor& xy,
                                                        const Tensor& vis, const Tensor& seg, const Tensor& blocks, const Tensor& tiles,
                                                        std::vector<int64_t> counts, double weight, int64_t kind, double delta,
                                                        const OptSink& sink, int64_t frame0, const OptTensor& plan_pixels,
                                                        const OptTensor& plan_first, const OptTensor& plan_entries, const OptTensor& plan_weights,
                                                        const OptTensor& anchor, const OptTensor& tap_slot, const OptTensor& tap_depth, const OptTensor& tap_shared,
                                                        bool offer_taps) {
  auto s = sink_of(sink);
  // `anchor`: the tensor whose history leads to the fit (the local extrinsics under frame sharding, where `ext` is the gathered chain)
  const Tensor& from = (anchor.has_value() && anchor->defined()) ? *anchor : ext;
  const bool park = s && s->fit_node != nullptr && reaches(from.grad_fn(), s->fit_node, 3);
  auto out = TrackLossFused::apply(depth, k, kinv, ext, xy, vis, seg, blocks, tiles, counts, weight, kind, delta, s, frame0, plan_pixels, plan_first,
                                   plan_entries, plan_weights, tap_slot, tap_depth, tap_shared, offer_taps, at::GradMode::is_enabled(), park);
  return {out[0], out[1], out[2]};
}
// would a flow loss fed these poses hand its dL/ddepth to the sink's fit (i.e. is the in-pass Adam update possible)?
// `leading_covered`: the frames a LeadingFrames node of this step reads (the softmin sweep: random pixels of frames 0 / 1, new every
// step) are wholly in the caller's touched set, so the sweep's gradient — added in place into the final buffer later — lands
// on elements the in-pass update leaves alone
static bool flow_loss_parks(const Tensor& t_fwd, const Tensor& t_bwd, const c10::intrusive_ptr<DepthSink>& sink, bool leading_covered) {
  return sink && sink->active && (!sink->expect_leading || leading_covered) && sink->fit_node != nullptr &&
         reaches(t_fwd.grad_fn(), sink->fit_node, 3) && reaches(t_bwd.grad_fn(), sink->fit_node, 3);
}

// Adam on a list of elements (the touched pixels of the in-pass update), versions bumped like adam_step
static void adam_step_elements(Tensor p, const Tensor& grad_in, Tensor m, Tensor v, const Tensor& elements, int64_t step, double lr, double beta1,
                               double beta2, double eps, double weight_decay) {
  const auto dev = check_device({&p, &grad_in, &m, &v, &elements});
  TORCH_CHECK(p.scalar_type() == at::kFloat && grad_in.scalar_type() == at::kFloat && elements.scalar_type() == at::kLong,
              "flowmap_amd.FusedAdam: float32 parameters / gradients and int64 element indices");
  TORCH_CHECK(p.is_contiguous() && m.is_contiguous() && v.is_contiguous() && grad_in.is_contiguous() && elements.is_contiguous() &&
                  grad_in.numel() == p.numel(),
              "flowmap_amd.FusedAdam: contiguous tensors of one size");
  DeviceScope scope(dev);
  FM_CALL(fm_adam_step_elements, ptr(p), ptr(grad_in), ptr(m), ptr(v), ptr<int64_t>(elements), (long)elements.numel(), (long)step, lr, beta1, beta2,
          eps, weight_decay, scope.stream);
  p.unsafeGetTensorImpl()->bump_version();
  m.unsafeGetTensorImpl()->bump_version();
  v.unsafeGetTensorImpl()->bump_version();
}

static std::tuple<Tensor, Tensor, Tensor> softmin_intrinsics_op(const Tensor& depth, const Tensor& weights, const Tensor& bwd_flow, const Tensor& indices,
                                                                 const Tensor& k, const Tensor& rel, double weight_sens, int64_t frames) {
  auto out = SoftminIntrinsics::apply(depth, weights, bwd_flow, indices, k, rel, weight_sens, frames);
  return {out[0], out[1], out[2]};
}

static Tensor leading_frames_op(const Tensor& x, int64_t count, const OptSink& sink) {
  return LeadingFrames::apply(x, count, sink_of(sink));
}

}  // namespace fmt

TORCH_LIBRARY(flowmap_amd, m) {
  m.class_<fmt::DepthSink>("DepthSink")
      .def(torch::init<>())
      .def("leading_in_place", [](const c10::intrusive_ptr<fmt::DepthSink>& s) { return s->leading_in_place; })
      .def("leading_dense", [](const c10::intrusive_ptr<fmt::DepthSink>& s) { return s->leading_dense; })
      .def("planned_steps", [](const c10::intrusive_ptr<fmt::DepthSink>& s) { return s->planned_steps; })
      .def("is_active", [](const c10::intrusive_ptr<fmt::DepthSink>& s) { return s->active; })
      .def("in_pass_confirmed", [](const c10::intrusive_ptr<fmt::DepthSink>& s) { return s->in_pass_confirmed; })
      .def("tap_absorbed", [](const c10::intrusive_ptr<fmt::DepthSink>& s) { return s->tap_absorbed; })
      .def("tap_confirmed", [](const c10::intrusive_ptr<fmt::DepthSink>& s) { return s->tap_confirmed; })
      .def("offers_taps", [](const c10::intrusive_ptr<fmt::DepthSink>& s) { return s->offers_taps(); })
      .def("taps_settled", [](const c10::intrusive_ptr<fmt::DepthSink>& s) { return std::vector<int64_t>{s->taps_settled_free, s->taps_settled_launch}; })
      .def("request_early_dense", [](const c10::intrusive_ptr<fmt::DepthSink>& s, const at::Tensor& unit_flag) {
        s->want_early = true;
        s->unit_flag = unit_flag;
      })
      .def("take_early_dense", [](const c10::intrusive_ptr<fmt::DepthSink>& s) -> std::optional<at::Tensor> {
        std::optional<at::Tensor> t;
        if (s->early_dense.defined()) t = s->early_dense;
        s->early_dense = at::Tensor();
        s->want_early = false;
        return t;
      });
  m.class_<fmt::GradArena>("GradArena")
      .def(torch::init<>())
      .def("reused", [](const c10::intrusive_ptr<fmt::GradArena>& a) { return a->reused; })
      .def("refilled", [](const c10::intrusive_ptr<fmt::GradArena>& a) { return a->refilled; })
      .def("holds", [](const c10::intrusive_ptr<fmt::GradArena>& a, const at::Tensor& grad) { return a->holds(grad); });
  m.def("set_library(str path, bool test_double) -> ()", fmt::set_library);
  m.def("intrinsics_inverse(Tensor k) -> Tensor", fmt::intrinsics_inverse);
  m.def("focal_intrinsics(Tensor focal, int[] repeat_shape, int height, int width) -> (Tensor, Tensor)", fmt::focal_intrinsics_op);
  m.def("pose_chain(Tensor rel) -> Tensor", fmt::pose_chain_op);
  m.def("relative_poses(Tensor ext) -> (Tensor, Tensor)", fmt::relative_poses_op);
  m.def(
      "procrustes_fit(Tensor? depth, Tensor? k, Tensor? kinv, Tensor? surfaces, Tensor weights, Tensor bwd_flow, Tensor? indices, float weight_sens, "
      "int batch_repeat, __torch__.torch.classes.flowmap_amd.DepthSink? sink, __torch__.torch.classes.flowmap_amd.DepthSink? wsink, "
      "__torch__.torch.classes.flowmap_amd.GradArena? arena, "
      "Tensor? plan_pixels, Tensor? plan_first, Tensor? plan_vectors, Tensor? plan_weights, Tensor? plan_frame_first, Tensor? plan_tap_records, Tensor? dense_first, "
      "Tensor? dense_list, Tensor? work, bool want_ext=True) "
      "-> (Tensor, Tensor, Tensor)",
      fmt::procrustes_fit_op);
  m.def(
      "flow_loss(Tensor depth, Tensor k, Tensor kinv, Tensor t_fwd, Tensor t_bwd, Tensor flow_fwd, Tensor flow_bwd, Tensor mask_fwd, Tensor mask_bwd, "
      "Tensor norm, Tensor? packed, int kind, float delta, __torch__.torch.classes.flowmap_amd.DepthSink? sink, int items, Tensor? acc_work, "
      "Tensor? exp_avg, Tensor? exp_avg_sq, Tensor? touched, int adam_step, float[] adam, Tensor? adam_flag, Tensor? tap_chunk_base=None, "
      "Tensor? tap_pixel=None, Tensor? tap_depth=None, Tensor? tap_stale=None, bool tap_require=False) -> Tensor",
      fmt::flow_loss_op);
  m.def(
      "track_loss(Tensor depth, Tensor k, Tensor kinv, Tensor ext, Tensor xy, Tensor vis, Tensor seg, Tensor blocks, Tensor tiles, int[] counts, "
      "float weight, int kind, float delta, __torch__.torch.classes.flowmap_amd.DepthSink? sink, int frame0, Tensor? plan_pixels, Tensor? plan_first, "
      "Tensor? plan_entries, Tensor? plan_weights, Tensor? anchor, Tensor? tap_slot=None, Tensor? tap_depth=None, Tensor? tap_shared=None, "
      "bool offer_taps=False) "
      "-> (Tensor, Tensor, Tensor)",
      fmt::track_loss_op);
  m.def("leading_frames(Tensor x, int count, __torch__.torch.classes.flowmap_amd.DepthSink? sink) -> Tensor", fmt::leading_frames_op);
  m.def("adam_step(Tensor(a!) p, Tensor grad, Tensor(b!) m, Tensor(c!) v, int step, Tensor? step_tensor, float lr, float beta1, float beta2, float eps, "
        "float weight_decay) -> ()",
        fmt::adam_step);
  m.def("flow_loss_parks(Tensor t_fwd, Tensor t_bwd, __torch__.torch.classes.flowmap_amd.DepthSink sink, bool leading_covered) -> bool",
        fmt::flow_loss_parks);
  m.def("adam_step_elements(Tensor(a!) p, Tensor grad, Tensor(b!) m, Tensor(c!) v, Tensor elements, int step, float lr, float beta1, float beta2, "
        "float eps, float weight_decay) -> ()",
        fmt::adam_step_elements);
  m.def("softmin_intrinsics(Tensor depth, Tensor weights, Tensor bwd_flow, Tensor indices, Tensor k, Tensor rel, float weight_sens, int frames) "
        "-> (Tensor, Tensor, Tensor)",
        fmt::softmin_intrinsics_op);
  m.def("random_subset(int n, int count, Device device, int seed, Tensor? state) -> Tensor", fmt::random_subset);
  m.def("set_one_launch_backward(bool on) -> ()", fmt::set_one_launch_backward);
  m.def("register_unit_seed(Tensor seed) -> ()", fmt::register_unit_seed);
  m.def("unit_seed_uses() -> int", []() { return fmt::unit_seed_uses(); });
  m.def("view_copies() -> int", []() { return fmt::view_copy_counter(); });
  m.def("flow_timing_enable(bool on) -> ()", fmt::flow_timing_enable);
  m.def("flow_timing_collect(bool tracking) -> float[]", fmt::flow_timing_collect);
}
