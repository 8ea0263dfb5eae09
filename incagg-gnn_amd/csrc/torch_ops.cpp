// Operator objects that `torch.ops.load_library` can load: the reference package finds
// `_relabel` and `_async` next to its __init__ and loads them
// (torch_geometric_autoscale/__init__.py:8-10); these two objects register the same five
// operators with the same schemas (csrc/async.cpp:44-48, csrc/relabel.cpp:40-42) and forward
// every call to the C ABI of libincagg_gnn.so.  Nothing is computed here: tensors are checked
// the way the reference asserts them (csrc/cuda/async_cuda.cu:20-59,119-137,
// csrc/cpu/relabel_cpu.cpp:9-16) and unwrapped into pointers and sizes.
//
// One source, two objects: -DINCAGG_OPS_RELABEL builds `_relabel`, -DINCAGG_OPS_ASYNC `_async`.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <deque>
#include <mutex>
#include <optional>
#include <tuple>
#include <vector>

#include "../../include/incagg_gnn.h"

namespace {

using at::Tensor;

void check_status(int64_t rc) { TORCH_CHECK(rc >= 0, incagg_last_error()); }

template <typename T>
const T *ptr_or_null(const Tensor &t) {
  return t.numel() ? static_cast<const T *>(t.data_ptr()) : nullptr;
}

#ifdef INCAGG_OPS_RELABEL
using RelabelFn = int (*)(const int64_t *, int64_t, const int64_t *, const void *, int64_t,
                          const int64_t *, int64_t, int, int64_t *, int64_t *, void *, int64_t *,
                          int64_t *);

std::tuple<Tensor, Tensor, std::optional<Tensor>, Tensor> relabel(
    RelabelFn fn, Tensor rowptr, Tensor col, std::optional<Tensor> optional_value, Tensor idx,
    bool bipartite) {
  TORCH_CHECK(!rowptr.is_cuda(), "No GPU version supported: relabelling runs on the host");
  TORCH_CHECK(!col.is_cuda(), "Col tensor must be a CPU tensor");
  TORCH_CHECK(!idx.is_cuda(), "Index tensor must be a CPU tensor");
  Tensor value;
  if (optional_value.has_value()) {
    value = optional_value.value();
    TORCH_CHECK(!value.is_cuda(), "Value tensor must be a CPU tensor");
    TORCH_CHECK(value.dim() == 1, "Value tensor must be one-dimensional");
    value = value.contiguous();
  }
  TORCH_CHECK(rowptr.scalar_type() == at::kLong, "rowptr must be int64");
  TORCH_CHECK(col.scalar_type() == at::kLong, "col must be int64");
  TORCH_CHECK(idx.scalar_type() == at::kLong, "idx must be int64");
  TORCH_CHECK(rowptr.dim() == 1 && rowptr.numel() >= 1, "rowptr must be 1-D and non-empty");
  rowptr = rowptr.contiguous();
  col = col.contiguous();
  idx = idx.contiguous();
  const int64_t n_nodes = rowptr.numel() - 1, n_idx = idx.numel();
  const int64_t ub = incagg_relabel_bound(rowptr.data_ptr<int64_t>(), n_nodes,
                                          ptr_or_null<int64_t>(idx), n_idx);
  check_status(ub);
  Tensor out_rowptr = at::empty({2 * n_idx + 1 + ub}, rowptr.options());
  Tensor out_col = at::empty({ub}, col.options());
  Tensor out_n_id = at::empty({n_idx + ub}, idx.options());
  Tensor out_value;
  if (value.defined()) out_value = at::empty({ub}, value.options());
  int64_t counts[3] = {0, 0, 0};
  check_status(fn(rowptr.data_ptr<int64_t>(), n_nodes, ptr_or_null<int64_t>(col),
                  value.defined() ? ptr_or_null<void>(value) : nullptr,
                  value.defined() ? (int64_t)value.element_size() : 0, ptr_or_null<int64_t>(idx),
                  n_idx, bipartite ? 1 : 0, out_rowptr.data_ptr<int64_t>(),
                  const_cast<int64_t *>(ptr_or_null<int64_t>(out_col)),
                  out_value.defined() ? const_cast<void *>(ptr_or_null<void>(out_value)) : nullptr,
                  const_cast<int64_t *>(ptr_or_null<int64_t>(out_n_id)), counts));
  std::optional<Tensor> ret_value;
  if (out_value.defined()) ret_value = out_value.narrow(0, 0, counts[0]).clone();
  return std::make_tuple(out_rowptr.narrow(0, 0, counts[2]).clone(),
                         out_col.narrow(0, 0, counts[0]).clone(), ret_value,
                         out_n_id.narrow(0, 0, counts[1]).clone());
}

std::tuple<Tensor, Tensor, std::optional<Tensor>, Tensor> relabel_one_hop(
    Tensor rowptr, Tensor col, std::optional<Tensor> value, Tensor idx, bool bipartite) {
  return relabel(&incagg_relabel_one_hop, rowptr, col, value, idx, bipartite);
}

std::tuple<Tensor, Tensor, std::optional<Tensor>, Tensor> relabel_one_hop_within_batch(
    Tensor rowptr, Tensor col, std::optional<Tensor> value, Tensor idx, bool bipartite) {
  return relabel(&incagg_relabel_one_hop_within_batch, rowptr, col, value, idx, bipartite);
}
#endif  // INCAGG_OPS_RELABEL

#ifdef INCAGG_OPS_ASYNC
// Tensors of queued reads stay referenced until the matching synchronize(), which is what
// capturing them by value in the worker's closure does in the reference (async_cuda.cu:63).
std::mutex g_inflight_mutex;
std::deque<std::vector<Tensor>> g_inflight;

Tensor host_i64(const Tensor &t, const char *name) {
  TORCH_CHECK(!t.is_cuda(), name, " tensor must be a CPU tensor");
  TORCH_CHECK(t.dim() == 1, name, " tensor must be one-dimensional");
  return t.to(at::kLong).contiguous();
}

void *non_default_stream(const Tensor &on_device) {
  const auto idx = on_device.get_device();
  auto cur = c10::hip::getCurrentHIPStream(idx);
  TORCH_CHECK(cur != c10::hip::getDefaultHIPStream(idx),
              "Asynchronous read/write requires a non-default stream");
  return static_cast<void *>(cur.stream());
}

int64_t row_elements(const Tensor &t) { return t.numel() / std::max<int64_t>(t.size(0), 1); }

void read_async(Tensor src, std::optional<Tensor> optional_offset,
                std::optional<Tensor> optional_count, Tensor index, Tensor dst, Tensor buffer) {
  TORCH_CHECK(!src.is_cuda(), "Source tensor must be a CPU tensor");
  TORCH_CHECK(!index.is_cuda(), "Index tensor must be a CPU tensor");
  TORCH_CHECK(dst.is_cuda(), "Target tensor must be a GPU tensor");
  TORCH_CHECK(!buffer.is_cuda(), "Buffer tensor must be a CPU tensor");
  TORCH_CHECK(buffer.is_pinned(), "Buffer tensor must be pinned");
  TORCH_CHECK(src.is_contiguous(), "Source tensor must be contiguous");
  TORCH_CHECK(dst.is_contiguous(), "Target tensor must be contiguous");
  TORCH_CHECK(buffer.is_contiguous(), "Buffer tensor must be contiguous");
  TORCH_CHECK(index.dim() == 1, "Index tensor must be one-dimensional");
  TORCH_CHECK(src.dim() >= 1 && dst.dim() >= 1 && buffer.dim() >= 1, "tensors must have rows");
  Tensor offset, count;
  int64_t n_chunks = 0;
  if (optional_offset.has_value()) {
    TORCH_CHECK(src.is_pinned(), "Source tensor must be pinned");
    TORCH_CHECK(optional_count.has_value(), "Count tensor is undefined");
    offset = host_i64(optional_offset.value(), "Offset");
    count = host_i64(optional_count.value(), "Count");
    TORCH_CHECK(offset.numel() == count.numel(), "Size mismatch");
    n_chunks = offset.numel();
  }
  TORCH_CHECK(index.numel() == 0 || index.scalar_type() == at::kLong, "Index tensor must be int64");
  index = index.contiguous();
  TORCH_CHECK(src.scalar_type() == dst.scalar_type() && src.scalar_type() == buffer.scalar_type(),
              "Source, target and buffer must share one dtype");
  const int64_t row_elems = row_elements(src);
  TORCH_CHECK(row_elements(dst) == row_elems && row_elements(buffer) == row_elems,
              "Source, target and buffer rows must have the same size");
  void *stream = non_default_stream(dst);
  c10::DeviceGuard guard(dst.device());
  check_status(incagg_stage_read_async(
      ptr_or_null<void>(src), src.size(0), row_elems * (int64_t)src.element_size(),
      n_chunks ? offset.data_ptr<int64_t>() : nullptr,
      n_chunks ? count.data_ptr<int64_t>() : nullptr, n_chunks, ptr_or_null<int64_t>(index),
      index.numel(), const_cast<void *>(ptr_or_null<void>(dst)), dst.size(0),
      const_cast<void *>(ptr_or_null<void>(buffer)), buffer.size(0), stream));
  std::vector<Tensor> keep{src, index, dst, buffer};
  if (n_chunks) {
    keep.push_back(offset);
    keep.push_back(count);
  }
  std::lock_guard<std::mutex> lock(g_inflight_mutex);
  g_inflight.push_back(std::move(keep));
}

void synchronize() {
  const int rc = incagg_stage_synchronize();
  {
    std::lock_guard<std::mutex> lock(g_inflight_mutex);
    if (!g_inflight.empty()) g_inflight.pop_front();
  }
  check_status(rc);
}

void write_async(Tensor src, Tensor offset, Tensor count, Tensor dst) {
  TORCH_CHECK(src.is_cuda(), "Source tensor must be a GPU tensor");
  TORCH_CHECK(!dst.is_cuda(), "Target tensor must be a CPU tensor");
  TORCH_CHECK(dst.is_pinned(), "Target tensor must be pinned");
  TORCH_CHECK(src.is_contiguous(), "Source tensor must be contiguous");
  TORCH_CHECK(dst.is_contiguous(), "Target tensor must be contiguous");
  TORCH_CHECK(src.dim() >= 1 && dst.dim() >= 1, "tensors must have rows");
  offset = host_i64(offset, "Offset");
  count = host_i64(count, "Count");
  TORCH_CHECK(offset.numel() == count.numel(), "Size mismatch");
  TORCH_CHECK(src.scalar_type() == dst.scalar_type(), "Source and target must share one dtype");
  const int64_t row_elems = row_elements(src);
  TORCH_CHECK(row_elements(dst) == row_elems, "Source and target rows must have the same size");
  void *stream = non_default_stream(src);
  c10::DeviceGuard guard(src.device());
  check_status(incagg_stage_write_async(
      ptr_or_null<void>(src), src.size(0), row_elems * (int64_t)src.element_size(),
      ptr_or_null<int64_t>(offset), ptr_or_null<int64_t>(count), offset.numel(),
      const_cast<void *>(ptr_or_null<void>(dst)), dst.size(0), stream));
}
#endif  // INCAGG_OPS_ASYNC

}  // namespace

// Both objects contribute to one namespace, hence fragments.
TORCH_LIBRARY_FRAGMENT(torch_geometric_autoscale, m) {
#ifdef INCAGG_OPS_RELABEL
  m.def("relabel_one_hop(Tensor rowptr, Tensor col, Tensor? value, Tensor idx, bool bipartite)"
        " -> (Tensor, Tensor, Tensor?, Tensor)", &relabel_one_hop);
  m.def("relabel_one_hop_within_batch(Tensor rowptr, Tensor col, Tensor? value, Tensor idx, "
        "bool bipartite) -> (Tensor, Tensor, Tensor?, Tensor)", &relabel_one_hop_within_batch);
#endif
#ifdef INCAGG_OPS_ASYNC
  m.def("synchronize() -> ()", &synchronize);
  m.def("read_async(Tensor src, Tensor? offset, Tensor? count, Tensor index, Tensor dst, "
        "Tensor buffer) -> ()", &read_async);
  m.def("write_async(Tensor src, Tensor offset, Tensor count, Tensor dst) -> ()", &write_async);
#endif
}
