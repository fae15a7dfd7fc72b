// pnsfm_seq.cpp -- the compiled layer between autograd and the kernels (round 6).
//
// Every block of the step used to be sequenced in Python: per autograd node a few dozen attribute look-ups, 3-6 torch.empty, 2-4 ctypes
// calls with 11-18 marshalled arguments each, stream forks as Python calls -- 14.7 ms of host time to enqueue a 24 ms step
// (profiles/r06_host_profile.txt; the reference's own path is stock ATen dispatch, C++ all the way: models/model_wrapper.py
// training_step -> SelfSupModel.forward, models/SelfSupModel.py:63-97).  This extension holds the BODIES of the hot autograd nodes --
// Conv2D block (conv -> GroupNorm statistics -> normalise + ELU), its backward (GroupNorm backward -> fork -> weight gradient on the side
// stream -> backward-data (+ gradient tap) -> join), the plain convolution, GroupNorm with a residual, the strip plumbing of the packing
// block -- as ONE call each: argument checks, output allocation (torch's caching allocator, on torch's current stream), the C-ABI
// launches of include/pnsfm.h and the stream fork / join in between.  Nothing here computes: it calls the same entry points of
// libpnsfm_hip.so the ctypes wrappers of packnet_sfm/hip/ops.py call, through function pointers handed over at start-up (bind()), so
// the host-emulated build of the kernels (tests/emu) runs underneath it as well.  hip/functional.py keeps the autograd Functions,
// the decisions (side stream or not, gradient slots, taps) and a pure-Python body for every node (PNSFM_SEQ=0).
#include <torch/extension.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/pnsfm.h"

namespace {

struct Abi {
  decltype(&pnsfm_last_error) last_error = nullptr;
  decltype(&pnsfm_conv2d_forward) conv2d_forward = nullptr;
  decltype(&pnsfm_conv2d_forward_cat) conv2d_forward_cat = nullptr;
  decltype(&pnsfm_conv2d_backward_data) conv2d_backward_data = nullptr;
  decltype(&pnsfm_conv2d_backward_data_add) conv2d_backward_data_add = nullptr;
  decltype(&pnsfm_conv2d_backward_weight) conv2d_backward_weight = nullptr;
  decltype(&pnsfm_conv2d_backward_weight_cat) conv2d_backward_weight_cat = nullptr;
  decltype(&pnsfm_groupnorm_act_forward) groupnorm_act_forward = nullptr;
  decltype(&pnsfm_groupnorm_act_backward) groupnorm_act_backward = nullptr;
  decltype(&pnsfm_stream_wait_stream) stream_wait_stream = nullptr;
  decltype(&pnsfm_region_ops) region_ops = nullptr;
  bool require_cuda = true;
  bool bound = false;
} g;

template <class F>
void take(F& slot, const py::dict& d, const char* name) {
  if (!d.contains(name)) throw std::runtime_error(std::string("pnsfm_seq.bind: missing symbol ") + name);
  slot = reinterpret_cast<F>(static_cast<uintptr_t>(d[name].cast<uint64_t>()));
}

void bind(const py::dict& addrs, bool require_cuda) {
  take(g.last_error, addrs, "pnsfm_last_error");
  take(g.conv2d_forward, addrs, "pnsfm_conv2d_forward");
  take(g.conv2d_forward_cat, addrs, "pnsfm_conv2d_forward_cat");
  take(g.conv2d_backward_data, addrs, "pnsfm_conv2d_backward_data");
  take(g.conv2d_backward_data_add, addrs, "pnsfm_conv2d_backward_data_add");
  take(g.conv2d_backward_weight, addrs, "pnsfm_conv2d_backward_weight");
  take(g.conv2d_backward_weight_cat, addrs, "pnsfm_conv2d_backward_weight_cat");
  take(g.groupnorm_act_forward, addrs, "pnsfm_groupnorm_act_forward");
  take(g.groupnorm_act_backward, addrs, "pnsfm_groupnorm_act_backward");
  take(g.stream_wait_stream, addrs, "pnsfm_stream_wait_stream");
  take(g.region_ops, addrs, "pnsfm_region_ops");
  g.require_cuda = require_cuda;
  g.bound = true;
}

void rc_check(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + " failed (rc=" + std::to_string(rc) + "): " + (g.last_error ? g.last_error() : "?"));
}

// what ops._chk / ops._f32 check: device tensors (no CPU fallback), fp32, contiguous, one device
void chk(const at::Tensor& t, const char* what, const at::Tensor* ref = nullptr) {
  if (g.require_cuda && !t.is_cuda())
    throw std::runtime_error(std::string("packnet_sfm HIP op got a ") + t.device().str() + " tensor (" + what +
                             "): the HIP kernels run on MI355X only, there is no CPU fallback");
  if (t.scalar_type() != at::kFloat) throw std::runtime_error(std::string("packnet_sfm HIP op needs float32 tensors (") + what + ")");
  if (!t.is_contiguous()) throw std::runtime_error(std::string("packnet_sfm HIP op needs contiguous tensors (") + what + ")");
  if (ref && t.device() != ref->device()) throw std::runtime_error("packnet_sfm HIP op got tensors on different devices");
}
const float* cptr(const at::Tensor& t) { return t.data_ptr<float>(); }
const float* cptr(const c10::optional<at::Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr<float>() : nullptr; }
float* mptr(at::Tensor& t) { return t.data_ptr<float>(); }
void* sp(uint64_t s) { return reinterpret_cast<void*>(static_cast<uintptr_t>(s)); }

// conv(cat(xs, 1)) into y (xs: 1..3 NCHW tensors)
void launch_conv_fwd(const std::vector<at::Tensor>& xs, const at::Tensor& wp, const c10::optional<at::Tensor>& bias, at::Tensor& y,
                     int Cout, int ks, uint64_t stream) {
  const auto& x0 = xs[0];
  const int B = (int)x0.size(0), H = (int)x0.size(2), W = (int)x0.size(3);
  if (xs.size() == 1) {
    rc_check(g.conv2d_forward(cptr(x0), cptr(wp), cptr(bias), mptr(y), B, (int)x0.size(1), Cout, H, W, ks, sp(stream)), "conv2d_forward");
  } else {
    const int C0 = (int)xs[0].size(1), C1 = (int)xs[1].size(1), C2 = xs.size() > 2 ? (int)xs[2].size(1) : 0;
    rc_check(g.conv2d_forward_cat(cptr(xs[0]), C0, cptr(xs[1]), C1, xs.size() > 2 ? cptr(xs[2]) : nullptr, C2, cptr(wp), cptr(bias), mptr(y), B,
                                  Cout, H, W, ks, sp(stream)),
             "conv2d_forward_cat");
  }
}

void check_inputs(const std::vector<at::Tensor>& xs, int Cin, const char* what) {
  if (xs.empty() || xs.size() > 3) throw std::runtime_error(std::string(what) + ": 1..3 input tensors");
  int c = 0;
  for (auto& t : xs) {
    chk(t, what, &xs[0]);
    if (t.dim() != 4 || t.size(0) != xs[0].size(0) || t.size(2) != xs[0].size(2) || t.size(3) != xs[0].size(3))
      throw std::runtime_error(std::string(what) + ": the input tensors must be NCHW of equal batch and size");
    c += (int)t.size(1);
  }
  if (Cin >= 0 && c != Cin) throw std::runtime_error(std::string(what) + ": inputs have " + std::to_string(c) + " channels, weight expects " + std::to_string(Cin));
}

// ---- Conv2D block forward: (out, y, [mean | rstd])
py::tuple conv_gn_act_forward(const std::vector<at::Tensor>& xs, const at::Tensor& wp_fwd, const c10::optional<at::Tensor>& bias,
                              const at::Tensor& gamma, const at::Tensor& beta, int64_t Cin, int64_t Cout, int64_t ks, int64_t G, double eps,
                              int64_t act, uint64_t stream) {
  check_inputs(xs, (int)Cin, "conv_gn_act");
  chk(wp_fwd, "packed weight", &xs[0]); chk(gamma, "gamma", &xs[0]); chk(beta, "beta", &xs[0]);
  if (bias.has_value() && bias->defined()) chk(*bias, "bias", &xs[0]);
  const int64_t B = xs[0].size(0), H = xs[0].size(2), W = xs[0].size(3);
  at::Tensor y = at::empty({B, Cout, H, W}, xs[0].options());
  launch_conv_fwd(xs, wp_fwd, bias, y, (int)Cout, (int)ks, stream);
  at::Tensor out = at::empty_like(y);
  at::Tensor ms = at::empty({2, B * G}, xs[0].options());
  float* mean = ms.data_ptr<float>();
  rc_check(g.groupnorm_act_forward(cptr(y), nullptr, cptr(gamma), cptr(beta), mptr(out), mean, mean + B * G, nullptr, (int)B, (int)Cout,
                                   (int)(H * W), (int)G, (float)eps, (int)act, sp(stream)),
           "groupnorm_act_forward");
  return py::make_tuple(out, y, ms);
}

// weight gradient of conv(cat(xs)) into (dw, db): fresh tensors (db right behind dw in one allocation) or the caller's slots
std::pair<at::Tensor, at::Tensor> launch_wgrad(const std::vector<at::Tensor>& xs, const at::Tensor& dy, int Cin, int Cout, int ks, bool has_bias,
                                               const c10::optional<at::Tensor>& dw_out, const c10::optional<at::Tensor>& db_out, uint64_t stream) {
  at::Tensor dw, db;
  if (dw_out.has_value() && dw_out->defined()) {
    chk(*dw_out, "gradient slot", &dy);
    if (dw_out->numel() != (int64_t)Cout * Cin * ks * ks || (has_bias && (!db_out.has_value() || db_out->numel() != Cout)))
      throw std::runtime_error("conv2d_backward_weight: gradient slot has the wrong shape");
    // fresh view objects: AccumulateGrad adopts a gradient only if nobody else holds the tensor object
    dw = dw_out->view({Cout, Cin, ks, ks});
    if (has_bias) db = db_out->view({Cout});
  } else {
    const int64_t n = (int64_t)Cout * Cin * ks * ks;
    at::Tensor buf = at::empty({n + (has_bias ? Cout : 0)}, dy.options());
    dw = buf.narrow(0, 0, n).view({Cout, Cin, ks, ks});
    if (has_bias) db = buf.narrow(0, n, Cout);
  }
  const int B = (int)dy.size(0), H = (int)dy.size(2), W = (int)dy.size(3);
  float* dbp = has_bias ? db.data_ptr<float>() : nullptr;
  if (xs.size() == 1) {
    rc_check(g.conv2d_backward_weight(cptr(xs[0]), cptr(dy), mptr(dw), dbp, B, Cin, Cout, H, W, ks, sp(stream)), "conv2d_backward_weight");
  } else {
    const int C0 = (int)xs[0].size(1), C1 = (int)xs[1].size(1), C2 = xs.size() > 2 ? (int)xs[2].size(1) : 0;
    rc_check(g.conv2d_backward_weight_cat(cptr(xs[0]), C0, cptr(xs[1]), C1, xs.size() > 2 ? cptr(xs[2]) : nullptr, C2, cptr(dy), mptr(dw), dbp, B,
                                          Cout, H, W, ks, sp(stream)),
             "conv2d_backward_weight_cat");
  }
  return {dw, db};
}

// backward-data of conv(cat(xs)) (+ the tap's gradient in the epilogue): the gradient of the whole concatenation
at::Tensor launch_dgrad(const at::Tensor& dy, const at::Tensor& wp_bwd, int Cin, int Cout, int ks, const c10::optional<at::Tensor>& addend,
                        uint64_t stream) {
  const int64_t B = dy.size(0), H = dy.size(2), W = dy.size(3);
  at::Tensor dx = at::empty({B, Cin, H, W}, dy.options());
  if (addend.has_value() && addend->defined()) {
    at::Tensor ad = *addend;
    if (ad.scalar_type() != at::kFloat || ad.device() != dy.device()) throw std::runtime_error("conv2d_backward_data: bad addend");
    if (ad.dim() != 4 || ad.size(0) != B || ad.size(1) != Cin || ad.size(2) != H || ad.size(3) != W)
      throw std::runtime_error("conv2d_backward_data: addend does not match dx");
    // dense, or a channel slice of a wider dense tensor (any sample stride)
    if (!(ad.stride(3) == 1 && ad.stride(2) == W && ad.stride(1) == H * W && (B == 1 || ad.stride(0) >= (int64_t)Cin * H * W))) ad = ad.contiguous();
    const long long bs = B > 1 ? (long long)ad.stride(0) : (long long)Cin * H * W;
    rc_check(g.conv2d_backward_data_add(cptr(dy), cptr(wp_bwd), mptr(dx), ad.data_ptr<float>(), bs, (int)B, Cin, Cout, (int)H, (int)W, ks, sp(stream)),
             "conv2d_backward_data_add");
  } else {
    rc_check(g.conv2d_backward_data(cptr(dy), cptr(wp_bwd), mptr(dx), (int)B, Cin, Cout, (int)H, (int)W, ks, sp(stream)), "conv2d_backward_data");
  }
  return dx;
}

// the shared tail of the two conv backward bodies: [fork ->] weight gradient [on the side stream] / backward-data [-> join]
//   side_stream != 0: the weight gradient is enqueued there after it has been made to wait for everything on main_stream so far; the
//   inputs are recorded on it for the caching allocator (side_obj: the torch.cuda.Stream of that handle); detached == false: the
//   compute stream waits for it again before this returns (a gradient something on the compute stream reads during this backward pass)
py::tuple conv_backward_tail(const at::Tensor& dy, const std::vector<at::Tensor>& xs, const c10::optional<at::Tensor>& wp_bwd, int Cin, int Cout,
                             int ks, bool has_bias, bool need_dx, bool want_w, const c10::optional<at::Tensor>& dw_out,
                             const c10::optional<at::Tensor>& db_out, const c10::optional<at::Tensor>& g_tap, uint64_t main_stream,
                             uint64_t side_stream, bool detached, const py::object& side_obj) {
  at::Tensor dx, dw, db;
  const bool on_side = want_w && side_stream != 0 && side_stream != main_stream;
  if (on_side) {
    rc_check(g.stream_wait_stream(sp(side_stream), sp(main_stream)), "stream_wait_stream");
    auto r = launch_wgrad(xs, dy, Cin, Cout, ks, has_bias, dw_out, db_out, side_stream);
    dw = r.first; db = r.second;
    if (!side_obj.is_none()) {
      const c10::Stream st = side_obj.cast<c10::Stream>();
      dy.record_stream(st);
      for (auto& t : xs) t.record_stream(st);
    }
  }
  if (need_dx) {
    if (!wp_bwd.has_value() || !wp_bwd->defined() || wp_bwd->numel() == 0) throw std::runtime_error("conv backward: no packed backward-data weight");
    dx = launch_dgrad(dy, *wp_bwd, Cin, Cout, ks, g_tap, main_stream);
  }
  if (want_w && !on_side) {
    auto r = launch_wgrad(xs, dy, Cin, Cout, ks, has_bias, dw_out, db_out, main_stream);
    dw = r.first; db = r.second;
  }
  if (on_side && !detached) rc_check(g.stream_wait_stream(sp(main_stream), sp(side_stream)), "stream_wait_stream");
  auto opt = [](const at::Tensor& t) -> py::object { return t.defined() ? py::cast(t) : py::none(); };
  return py::make_tuple(opt(dx), opt(dw), opt(db));
}

// ---- Conv2D block backward: (dx | None, dw | None, db | None, dgamma, dbeta); dx is the gradient of the whole concatenation
py::tuple conv_gn_act_backward(const at::Tensor& dout, const at::Tensor& y, const at::Tensor& gamma, const at::Tensor& beta, const at::Tensor& ms,
                               const std::vector<at::Tensor>& xs, const c10::optional<at::Tensor>& wp_bwd, int64_t Cin, int64_t Cout, int64_t ks,
                               int64_t G, int64_t act, bool has_bias, bool need_dx, bool want_w, const c10::optional<at::Tensor>& dw_out,
                               const c10::optional<at::Tensor>& db_out, const c10::optional<at::Tensor>& g_tap, uint64_t main_stream,
                               uint64_t side_stream, bool detached, const py::object& side_obj) {
  at::Tensor dz = dout.is_contiguous() ? dout : dout.contiguous();
  chk(dz, "dout"); chk(y, "y", &dz); chk(gamma, "gamma", &dz); chk(beta, "beta", &dz); chk(ms, "mean / rstd", &dz);
  check_inputs(xs, (int)Cin, "conv_gn_act backward");
  const int64_t B = y.size(0), H = y.size(2), W = y.size(3);
  at::Tensor dy = at::empty_like(y);
  at::Tensor dgb = at::empty({2, Cout}, y.options());
  float* dgamma = dgb.data_ptr<float>();
  const float* mean = ms.data_ptr<float>();
  rc_check(g.groupnorm_act_backward(cptr(dz), cptr(y), nullptr, cptr(gamma), cptr(beta), mean, mean + B * G, mptr(dy), dgamma, dgamma + Cout, nullptr,
                                    (int)B, (int)Cout, (int)(H * W), (int)G, (int)act, sp(main_stream)),
           "groupnorm_act_backward");
  py::tuple t = conv_backward_tail(dy, xs, wp_bwd, (int)Cin, (int)Cout, (int)ks, has_bias, need_dx, want_w, dw_out, db_out, g_tap, main_stream,
                                   side_stream, detached, side_obj);
  return py::make_tuple(t[0], t[1], t[2], dgb.select(0, 0).view_as(gamma), dgb.select(0, 1).view_as(beta));
}

// ---- plain convolution
at::Tensor conv2d_forward(const std::vector<at::Tensor>& xs, const at::Tensor& wp_fwd, const c10::optional<at::Tensor>& bias, int64_t Cin,
                          int64_t Cout, int64_t ks, uint64_t stream) {
  check_inputs(xs, (int)Cin, "conv2d");
  chk(wp_fwd, "packed weight", &xs[0]);
  if (bias.has_value() && bias->defined()) chk(*bias, "bias", &xs[0]);
  at::Tensor y = at::empty({xs[0].size(0), Cout, xs[0].size(2), xs[0].size(3)}, xs[0].options());
  launch_conv_fwd(xs, wp_fwd, bias, y, (int)Cout, (int)ks, stream);
  return y;
}

py::tuple conv2d_backward(const at::Tensor& dy_in, const std::vector<at::Tensor>& xs, const c10::optional<at::Tensor>& wp_bwd, int64_t Cin,
                          int64_t Cout, int64_t ks, bool has_bias, bool need_dx, bool want_w, const c10::optional<at::Tensor>& dw_out,
                          const c10::optional<at::Tensor>& db_out, const c10::optional<at::Tensor>& g_tap, uint64_t main_stream,
                          uint64_t side_stream, bool detached, const py::object& side_obj) {
  at::Tensor dy = dy_in.is_contiguous() ? dy_in : dy_in.contiguous();
  chk(dy, "dy");
  check_inputs(xs, (int)Cin, "conv2d backward");
  return conv_backward_tail(dy, xs, wp_bwd, (int)Cin, (int)Cout, (int)ks, has_bias, need_dx, want_w, dw_out, db_out, g_tap, main_stream, side_stream,
                            detached, side_obj);
}

// ---- GroupNorm (+ residual in front) + activation
py::tuple gn_act_forward(const at::Tensor& x_in, const c10::optional<at::Tensor>& res_in, const at::Tensor& gamma, const at::Tensor& beta, int64_t G,
                         double eps, int64_t act, uint64_t stream) {
  at::Tensor x = x_in.is_contiguous() ? x_in : x_in.contiguous();
  at::Tensor res;
  if (res_in.has_value() && res_in->defined()) res = res_in->is_contiguous() ? *res_in : res_in->contiguous();
  chk(x, "x"); chk(gamma, "gamma", &x); chk(beta, "beta", &x);
  if (res.defined()) { chk(res, "res", &x); if (res.sizes() != x.sizes()) throw std::runtime_error("groupnorm_act: residual of another shape"); }
  const int64_t B = x.size(0), C = x.size(1), HW = x.numel() / (B * C);
  at::Tensor y = at::empty_like(x);
  at::Tensor ms = at::empty({2, B * G}, x.options());
  float* mean = ms.data_ptr<float>();
  rc_check(g.groupnorm_act_forward(cptr(x), res.defined() ? cptr(res) : nullptr, cptr(gamma), cptr(beta), mptr(y), mean, mean + B * G, nullptr, (int)B,
                                   (int)C, (int)HW, (int)G, (float)eps, (int)act, sp(stream)),
           "groupnorm_act_forward");
  return py::make_tuple(y, ms, x, res.defined() ? py::cast(res) : py::none());
}

py::tuple gn_act_backward(const at::Tensor& dy_in, const at::Tensor& x, const c10::optional<at::Tensor>& res, const at::Tensor& gamma,
                          const at::Tensor& beta, const at::Tensor& ms, int64_t G, int64_t act, uint64_t stream) {
  at::Tensor dy = dy_in.is_contiguous() ? dy_in : dy_in.contiguous();
  chk(dy, "dy"); chk(x, "x", &dy); chk(gamma, "gamma", &dy); chk(beta, "beta", &dy); chk(ms, "mean / rstd", &dy);
  const bool has_res = res.has_value() && res->defined();
  if (has_res) chk(*res, "res", &dy);
  const int64_t B = x.size(0), C = x.size(1), HW = x.numel() / (B * C);
  at::Tensor dx = at::empty_like(x);
  at::Tensor dgb = at::empty({2, C}, x.options());
  float* dgamma = dgb.data_ptr<float>();
  const float* mean = ms.data_ptr<float>();
  rc_check(g.groupnorm_act_backward(cptr(dy), cptr(x), has_res ? cptr(*res) : nullptr, cptr(gamma), cptr(beta), mean, mean + B * G, mptr(dx), dgamma,
                                    dgamma + C, nullptr, (int)B, (int)C, (int)HW, (int)G, (int)act, sp(stream)),
           "groupnorm_act_backward");
  return py::make_tuple(dx, dgb.select(0, 0).view_as(gamma), dgb.select(0, 1).view_as(beta));
}

// ---- strided window operations (the packing block's border strips): items = [(op, dst, src | None)], <= 12 per launch
void region_ops(const std::vector<std::tuple<int64_t, at::Tensor, c10::optional<at::Tensor>>>& items, uint64_t stream) {
  constexpr size_t kMax = 12;        // MAX_REGION_OPS of hip/ops.py, include/pnsfm.h
  for (size_t i0 = 0; i0 < items.size(); i0 += kMax) {
    pnsfm_region_op arr[kMax];
    const size_t n = std::min(kMax, items.size() - i0);
    const at::Tensor& ref = std::get<1>(items[i0]);
    for (size_t k = 0; k < n; ++k) {
      const auto& it = items[i0 + k];
      const at::Tensor& dst = std::get<1>(it);
      const auto& src = std::get<2>(it);
      const bool has_src = src.has_value() && src->defined();
      if (dst.scalar_type() != at::kFloat || (has_src && src->scalar_type() != at::kFloat)) throw std::runtime_error("region_ops needs float32 tensors");
      if (g.require_cuda && !dst.is_cuda())
        throw std::runtime_error("packnet_sfm HIP op got a " + dst.device().str() + " tensor: the HIP kernels run on MI355X only, there is no CPU fallback");
      if (dst.dim() > 4 || (has_src && src->sizes() != dst.sizes()) || dst.device() != ref.device())
        throw std::runtime_error("region_ops: windows must be <= 4-D, of equal shape, on one device");
      const int pad = 4 - (int)dst.dim();
      arr[k].dst = dst.data_ptr<float>();
      arr[k].src = has_src ? src->data_ptr<float>() : nullptr;
      arr[k].op = (int)std::get<0>(it);
      for (int d = 0; d < 4; ++d) {
        const int64_t nd = d < pad ? 1 : dst.size(d - pad);
        if (nd == 0) throw std::runtime_error("region_ops: empty window");
        arr[k].n[d] = (int)nd;
        arr[k].dst_stride[d] = d < pad ? 0 : dst.stride(d - pad);
        arr[k].src_stride[d] = (!has_src || d < pad) ? 0 : src->stride(d - pad);
      }
    }
    rc_check(g.region_ops(arr, (int)n, sp(stream)), "region_ops");
  }
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "packnet_sfm block sequencer: bodies of the hot autograd nodes as single calls over the C ABI of libpnsfm_hip.so";
  m.def("bind", &bind, py::arg("addresses"), py::arg("require_cuda"));
  m.def("conv_gn_act_forward", &conv_gn_act_forward);
  m.def("conv_gn_act_backward", &conv_gn_act_backward);
  m.def("conv2d_forward", &conv2d_forward);
  m.def("conv2d_backward", &conv2d_backward);
  m.def("gn_act_forward", &gn_act_forward);
  m.def("gn_act_backward", &gn_act_backward);
  m.def("region_ops", &region_ops);
}
