// Test infrastructure only: pybind entry points over the REFERENCE's own spconv-1.x CPU code.
//
// This translation unit is ours; it contains no reference code.  It includes the reference's
// include/spconv/geometry.h and is compiled together with the reference's src/reordering.cc WHERE
// THEY LIE under /root/reference/mmdet3d/ops/spconv (see oracle/build_ref.py), so that
//   * getIndicePairsSubM / getIndicePairsConv / getValidOutPos (include/spconv/geometry.h:24-85,
//     144-194, 247-297) and
//   * SparseGatherFunctor / SparseScatterAddFunctor<tv::CPU> (src/reordering.cc:20-50)
// run as the reference wrote them.  What is restated here is only the driver glue that cannot be
// compiled in this image: getIndicePair<3>'s CPU branch (include/spconv/spconv_ops.h:27-141, which
// reaches geometry.h through the three-line functors of src/indice.cc:23-64) and
// indiceConv<float>'s CPU branch (spconv_ops.h:260-361).  spconv_ops.h (and src/indice.cc, which
// includes it) pull in torch_utils.h -> ATen/cuda/CUDAContext.h -> <cusparse.h>, which neither this
// image nor the GPU box has.
#include <torch/extension.h>

#include <spconv/geometry.h>
#include <spconv/reordering.h>

#include <algorithm>
#include <vector>

namespace {

template <typename T>
tv::TensorView<T> view_of(const torch::Tensor &t) {
  tv::Shape shape;
  for (auto s : t.sizes()) shape.push_back(int(s));
  return tv::TensorView<T>(t.data_ptr<std::remove_const_t<T>>(), shape);
}

tv::SimpleVector<int, 3> vec3(const std::vector<int64_t> &v) {
  TORCH_CHECK(v.size() == 3, "expected 3 values");
  tv::SimpleVector<int, 3> r;
  for (auto x : v) r.push_back(int(x));
  return r;
}

// spconv_ops.h:27-141, CPU branch: allocation values (-1 pairs, 0 counts, -1 grid), SubM forcing
// stride 1 / padding k/2 (:76-79), outputs sliced to numActOut (:139).
std::vector<torch::Tensor> get_indice_pairs(torch::Tensor indices, int64_t batch_size,
                                            std::vector<int64_t> out_shape,
                                            std::vector<int64_t> ksize,
                                            std::vector<int64_t> stride,
                                            std::vector<int64_t> padding,
                                            std::vector<int64_t> dilation, bool subm) {
  TORCH_CHECK(indices.device().is_cpu() && indices.dtype() == torch::kInt32 &&
              indices.dim() == 2 && indices.size(1) == 4 && indices.is_contiguous());
  const int64_t n = indices.size(0);
  const int64_t K = ksize[0] * ksize[1] * ksize[2];
  const int64_t volume = out_shape[0] * out_shape[1] * out_shape[2];
  auto i32 = torch::dtype(torch::kInt32);
  auto pairs = torch::full({K, 2, n}, -1, i32);
  auto num = torch::zeros({K}, i32);
  auto grid = torch::full({batch_size * volume}, -1, i32);
  if (subm) {
    std::vector<int64_t> one{1, 1, 1}, half{ksize[0] / 2, ksize[1] / 2, ksize[2] / 2};
    spconv::getIndicePairsSubM<int, int, 3>(
        view_of<const int>(indices), view_of<int>(grid), view_of<int>(pairs), view_of<int>(num),
        vec3(ksize).data(), vec3(one).data(), vec3(half).data(), vec3(dilation).data(),
        vec3(out_shape).data());
    return {indices, pairs, num};
  }
  auto outids = torch::zeros({std::max<int64_t>(n * K, 1), 4}, i32);
  int m = spconv::getIndicePairsConv<int, int, 3>(
      view_of<const int>(indices), view_of<int>(outids), view_of<int>(grid), view_of<int>(pairs),
      view_of<int>(num), vec3(ksize).data(), vec3(stride).data(), vec3(padding).data(),
      vec3(dilation).data(), vec3(out_shape).data());
  return {outids.slice(0, 0, m).clone(), pairs, num};
}

// spconv_ops.h:260-361, CPU branch.  gather / scatter-add are the reference's functors, the
// product is torch::mm_out as there; the centre offset of a SubM conv (= argmax of indiceNum,
// first maximum, :272-277) is one full mm without gather (:300-303).
torch::Tensor indice_conv(torch::Tensor features, torch::Tensor filters, torch::Tensor pairs,
                          torch::Tensor num, int64_t n_out, bool inverse, bool subm) {
  TORCH_CHECK(features.dtype() == torch::kFloat32 && features.is_contiguous());
  TORCH_CHECK(pairs.dtype() == torch::kInt32 && pairs.is_contiguous());
  const int64_t K = pairs.size(0), cin = features.size(1);
  const int64_t cout = filters.size(filters.dim() - 1);
  const int *cnt = num.data_ptr<int>();
  const int *mx = std::max_element(cnt, cnt + K);
  const int centre = int(mx - cnt), max_size = *mx;
  auto opt = torch::dtype(torch::kFloat32);
  auto output = torch::zeros({n_out, cout}, opt);
  auto in_buf = torch::zeros({std::max(max_size, 1), cin}, opt);
  auto out_buf = torch::zeros({std::max(max_size, 1), cout}, opt);
  filters = filters.contiguous().view({-1, cin, cout});
  if (subm) torch::mm_out(output, features, filters[centre]);
  spconv::functor::SparseGatherFunctor<tv::CPU, float, int> gather;
  spconv::functor::SparseScatterAddFunctor<tv::CPU, float, int> scatter;
  for (int64_t k = 0; k < K; ++k) {
    const int hot = cnt[k];
    if (hot <= 0 || (subm && k == centre)) continue;
    auto ob = torch::from_blob(out_buf.data_ptr<float>(), {hot, cout}, opt);
    auto ib = torch::from_blob(in_buf.data_ptr<float>(), {hot, cin}, opt);
    gather(tv::CPU(), view_of<float>(in_buf), view_of<const float>(features),
           view_of<const int>(pairs).subview(int(k), int(inverse)), hot);
    torch::mm_out(ob, ib, filters[k]);
    scatter(tv::CPU(), view_of<float>(output), view_of<const float>(out_buf),
            view_of<const int>(pairs).subview(int(k), int(!inverse)), hot, true);
  }
  return output;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("get_indice_pairs", &get_indice_pairs);
  m.def("indice_conv", &indice_conv);
}
