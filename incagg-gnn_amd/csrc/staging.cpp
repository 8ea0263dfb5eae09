// Host-pinned <-> HBM staging of History rows on hipMemcpyAsync side streams.
//
// Reads are queued to one process-wide worker thread and joined one at a time in
// FIFO order (the contract of csrc/thread.h:10-69 and csrc/cuda/async_cuda.cu:7-12);
// writes are issued from the caller's thread.  Nothing here blocks on a stream:
// the caller orders consumers behind the copies with the stream it passed in.
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace incagg {

struct TaskResult {
  int code = INCAGG_OK;
  std::string message;
};

class StagingWorker {
 public:
  StagingWorker() : stop_(false), worker_([this] { loop(); }) {}
  ~StagingWorker() {
    {
      std::lock_guard<std::mutex> lock(mutex_);
      stop_ = true;
    }
    cv_.notify_all();
    worker_.join();
  }

  void submit(std::function<TaskResult()> fn) {
    std::packaged_task<TaskResult()> task(std::move(fn));
    {
      std::lock_guard<std::mutex> lock(mutex_);
      results_.push_back(task.get_future());
      tasks_.push_back(std::move(task));
    }
    cv_.notify_one();
  }

  // Joins the oldest queued task; false when nothing is queued.
  bool join_one(TaskResult *out) {
    std::future<TaskResult> f;
    {
      std::lock_guard<std::mutex> lock(mutex_);
      if (results_.empty()) return false;
      f = std::move(results_.front());
      results_.pop_front();
    }
    *out = f.get();
    return true;
  }

  int64_t pending() {
    std::lock_guard<std::mutex> lock(mutex_);
    return (int64_t)results_.size();
  }

 private:
  void loop() {
    for (;;) {
      std::packaged_task<TaskResult()> task;
      {
        std::unique_lock<std::mutex> lock(mutex_);
        cv_.wait(lock, [this] { return stop_ || !tasks_.empty(); });
        if (stop_ && tasks_.empty()) return;
        task = std::move(tasks_.front());
        tasks_.pop_front();
      }
      task();
    }
  }

  bool stop_;
  std::mutex mutex_;
  std::condition_variable cv_;
  std::deque<std::packaged_task<TaskResult()>> tasks_;
  std::deque<std::future<TaskResult>> results_;
  std::thread worker_;  // declared last: started after the queues exist
};

static StagingWorker &worker() {
  static StagingWorker w;
  return w;
}

// rows[index[k]] -> bounce[k]; split over a few helper threads when the block is
// large enough for the thread start-up to pay (random 1-KiB rows are
// memory-latency bound on one core).
static void gather_to_bounce(const char *src, int64_t row_bytes, const int64_t *index,
                             int64_t n_index, char *bounce) {
  const int64_t bytes = n_index * row_bytes;
  int helpers = 1;
  if (bytes >= (8ll << 20)) {
    unsigned hc = std::thread::hardware_concurrency();
    helpers = (int)std::min<int64_t>(hc ? std::min(hc, 8u) : 4, bytes >> 21);
  }
  auto part = [&](int64_t lo, int64_t hi) {
    for (int64_t k = lo; k < hi; k++)
      std::memcpy(bounce + k * row_bytes, src + index[k] * row_bytes, (size_t)row_bytes);
  };
  if (helpers <= 1) {
    part(0, n_index);
    return;
  }
  std::vector<std::thread> pool;
  const int64_t step = (n_index + helpers - 1) / helpers;
  for (int h = 1; h < helpers; h++)
    pool.emplace_back(part, std::min<int64_t>(h * step, n_index), std::min<int64_t>((h + 1) * step, n_index));
  part(0, std::min<int64_t>(step, n_index));
  for (auto &t : pool) t.join();
}

}  // namespace incagg

using namespace incagg;

extern "C" int incagg_stage_read_async(const void *src_host, int64_t src_rows, int64_t row_bytes,
                                       const int64_t *offset, const int64_t *count,
                                       int64_t n_chunks, const int64_t *index, int64_t n_index,
                                       void *dst_dev, int64_t dst_rows, void *bounce_host,
                                       int64_t bounce_rows, void *stream) {
  INCAGG_REQUIRE(src_host && dst_dev, "null source or destination");
  INCAGG_REQUIRE(src_rows >= 0 && row_bytes > 0 && n_chunks >= 0 && n_index >= 0, "bad size");
  INCAGG_REQUIRE(n_chunks == 0 || (offset && count), "Count tensor is undefined");
  INCAGG_REQUIRE(n_index == 0 || (index && bounce_host), "null index or bounce buffer");
  INCAGG_REQUIRE(stream != nullptr, "Asynchronous read requires a non-default stream");
  int64_t numel = 0;
  for (int64_t c = 0; c < n_chunks; c++) {
    if (offset[c] < 0 || count[c] < 0 || offset[c] + count[c] > src_rows)
      return fail(INCAGG_E_INDEX, "Invalid index: chunk %lld = [%lld, +%lld) outside %lld rows",
                  (long long)c, (long long)offset[c], (long long)count[c], (long long)src_rows);
    numel += count[c];
  }
  // As csrc/cuda/async_cuda.cu:51-52 the pinned buffer must hold the whole pull
  // (chunks included), whenever one is supplied.
  if ((bounce_host || n_index > 0) && numel + n_index > bounce_rows)
    return fail(INCAGG_E_INVALID, "Buffer tensor size too small");
  if (numel + n_index > dst_rows) return fail(INCAGG_E_INVALID, "Target tensor size too small");

  int dev = 0;
  INCAGG_HIP_TRY(hipGetDevice(&dev));
  std::vector<int64_t> off(offset, offset + n_chunks), cnt(count, count + n_chunks);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char *src = static_cast<const char *>(src_host);
  char *dst = static_cast<char *>(dst_dev);
  char *bounce = static_cast<char *>(bounce_host);

  worker().submit([=]() -> TaskResult {
    TaskResult res;
    auto hip_fail = [&](const char *what, hipError_t e) {
      res.code = INCAGG_E_HIP;
      res.message = std::string(what) + ": " + hipGetErrorString(e);
      return res;
    };
    hipError_t e = hipSetDevice(dev);
    if (e != hipSuccess) return hip_fail("hipSetDevice", e);
    if (n_index > 0) {
      for (int64_t k = 0; k < n_index; k++)
        if (index[k] < 0 || index[k] >= src_rows) {
          res.code = INCAGG_E_INDEX;
          res.message = "Invalid index: gathered row " + std::to_string(index[k]) + " outside " +
                        std::to_string(src_rows) + " rows";
          return res;
        }
      // The bounce buffer is reused from pull to pull of the same slot: an earlier
      // copy out of it (same stream) must have drained before it is overwritten.
      // This wait is on the worker thread; the caller never blocks.
      e = hipStreamSynchronize(s);
      if (e != hipSuccess) return hip_fail("hipStreamSynchronize (before host gather)", e);
      gather_to_bounce(src, row_bytes, index, n_index, bounce);
    }
    int64_t dst_row = 0;
    for (size_t c = 0; c < off.size(); c++) {
      if (cnt[c] == 0) continue;
      e = hipMemcpyAsync(dst + dst_row * row_bytes, src + off[c] * row_bytes,
                         (size_t)(cnt[c] * row_bytes), hipMemcpyHostToDevice, s);
      if (e != hipSuccess) return hip_fail("hipMemcpyAsync (chunk, host to device)", e);
      dst_row += cnt[c];
    }
    if (n_index > 0) {
      e = hipMemcpyAsync(dst + dst_row * row_bytes, bounce, (size_t)(n_index * row_bytes),
                         hipMemcpyHostToDevice, s);
      if (e != hipSuccess) return hip_fail("hipMemcpyAsync (gathered rows, host to device)", e);
    }
    return res;
  });
  return INCAGG_OK;
}

extern "C" int incagg_stage_synchronize(void) {
  TaskResult r;
  if (!worker().join_one(&r)) return INCAGG_OK;
  if (r.code != INCAGG_OK) return fail(r.code, "%s", r.message.c_str());
  return INCAGG_OK;
}

extern "C" int64_t incagg_stage_pending(void) { return worker().pending(); }

extern "C" int incagg_stage_write_async(const void *src_dev, int64_t src_rows, int64_t row_bytes,
                                        const int64_t *offset, const int64_t *count,
                                        int64_t n_chunks, void *dst_host, int64_t dst_rows,
                                        void *stream) {
  INCAGG_REQUIRE(src_dev && dst_host, "null source or destination");
  INCAGG_REQUIRE(src_rows >= 0 && row_bytes > 0 && n_chunks >= 0, "bad size");
  INCAGG_REQUIRE(n_chunks == 0 || (offset && count), "null offset or count");
  INCAGG_REQUIRE(stream != nullptr, "Asynchronous write requires a non-default stream");
  int64_t total = 0;
  for (int64_t c = 0; c < n_chunks; c++) {
    if (offset[c] < 0 || count[c] < 0 || offset[c] + count[c] > dst_rows)
      return fail(INCAGG_E_INDEX, "Invalid index: chunk %lld = [%lld, +%lld) outside %lld rows",
                  (long long)c, (long long)offset[c], (long long)count[c], (long long)dst_rows);
    total += count[c];
  }
  if (total > src_rows)
    return fail(INCAGG_E_INDEX, "Invalid index: chunks hold %lld rows, source has %lld",
                (long long)total, (long long)src_rows);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char *src = static_cast<const char *>(src_dev);
  char *dst = static_cast<char *>(dst_host);
  int64_t src_row = 0;
  for (int64_t c = 0; c < n_chunks; c++) {
    if (count[c] == 0) continue;
    INCAGG_HIP_TRY(hipMemcpyAsync(dst + offset[c] * row_bytes, src + src_row * row_bytes,
                                  (size_t)(count[c] * row_bytes), hipMemcpyDeviceToHost, s));
    src_row += count[c];
  }
  return INCAGG_OK;
}
