// dvm_slam_amd/csrc/host_stage.h -- staging of the host-pointer ("convenience") entry points of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "orb_pipeline.h"   // hip_check / DVM_* status codes

namespace dvm {
// Host-pointer convenience paths.  Every calling thread keeps ONE staging context per device: a page-locked host buffer, a
// device buffer, a MAPPED page-locked buffer (all grow-only) and a stream.  A call packs its copied inputs into the pinned buffer
// and sends them with one asynchronous copy, writes its mapped inputs where the kernels read them in place, launches its kernels,
// fetches the copied outputs with one copy and synchronises once.  The context's stream is a BLOCKING stream: entry points that
// still launch on the legacy default stream are ordered behind the upload and in front of the download by the runtime; the per-frame
// entry points (grid build, window searches, PoseOptimization) launch on stream() itself -- one in-order chain, no cross-stream
// hand-off.  (hipMalloc + a pageable hipMemcpy per array + hipDeviceSynchronize + hipFree per call made
// ORBmatcher::SearchByProjection over 1 100 keypoints a 1.8 ms call next to a 0.44 ms CPU run of the same function; it is 0.10 ms now.)
struct StageCtx {
  int device = -1;
  hipStream_t s = nullptr;
  uint8_t *h = nullptr, *d = nullptr;
  size_t hcap = 0, dcap = 0;
  uint8_t *hm = nullptr, *hm_dev = nullptr;   // mapped pinned buffer (items the kernels read / write in place over PCIe) and its device address
  size_t hmcap = 0;
  void release() {
    if (s) { hipStreamSynchronize(s); hipStreamDestroy(s); }
    if (h) hipHostFree(h);
    if (hm) hipHostFree(hm);
    if (d) hipFree(d);
    s = nullptr; h = d = hm = hm_dev = nullptr; hcap = dcap = hmcap = 0;
  }
  int ensure_mapped(size_t bytes) {
    if (bytes <= hmcap) return DVM_OK;
    if (hm) hipHostFree(hm);
    hm = hm_dev = nullptr; hmcap = 0;
    const size_t cap = std::max<size_t>(bytes * 2, (size_t)1 << 20);
    int rc = hip_check(hipHostMalloc(reinterpret_cast<void**>(&hm), cap, hipHostMallocMapped), "hipHostMalloc");
    if (rc != DVM_OK) return rc;
    rc = hip_check(hipHostGetDevicePointer(reinterpret_cast<void**>(&hm_dev), hm, 0), "hipHostGetDevicePointer");
    if (rc != DVM_OK) return rc;
    hmcap = cap;
    return DVM_OK;
  }
  int ensure(size_t bytes) {
    int dev = 0;
    int rc = hip_check(hipGetDevice(&dev), "hipGetDevice");
    if (rc != DVM_OK) return rc;
    if (dev != device) { release(); device = dev; }
    if (!s) { rc = hip_check(hipStreamCreateWithFlags(&s, hipStreamDefault), "stream"); if (rc != DVM_OK) return rc; }
    rc = hip_check(hipStreamSynchronize(s), "sync");   // the previous call's asynchronous traffic has left the buffers
    if (rc != DVM_OK) return rc;
    if (bytes > hcap) {
      if (h) hipHostFree(h);
      h = nullptr; hcap = 0;
      const size_t cap = std::max<size_t>(bytes * 2, (size_t)1 << 20);
      rc = hip_check(hipHostMalloc(reinterpret_cast<void**>(&h), cap, hipHostMallocDefault), "hipHostMalloc");
      if (rc != DVM_OK) return rc;
      hcap = cap;
    }
    if (bytes > dcap) {
      if (d) hipFree(d);
      d = nullptr; dcap = 0;
      const size_t cap = std::max<size_t>(bytes * 2, (size_t)1 << 20);
      rc = hip_check(hipMalloc(reinterpret_cast<void**>(&d), cap), "hipMalloc");
      if (rc != DVM_OK) return rc;
      dcap = cap;
    }
    return DVM_OK;
  }
  ~StageCtx() { release(); }
};
inline StageCtx& stage_ctx() {
  thread_local StageCtx c;
  return c;
}

// A few persistent host threads for the batch entry points (window tables, staging copies of tens of MB): run(n, fn) calls fn(i) for
// i in [0, n) on the caller and up to `width` - 1 workers and returns when all are done.  Spawning std::threads per call cost 30-50 us
// each -- as much as the work they were given.  One job at a time (callers serialise on a mutex); workers sleep between jobs.
class HostPool {
 public:
  // (never destroyed: its sleeping workers end with the process -- a destructor that joins them would run in every forked child of a
  //  process that had used the pool, where those threads do not exist)
  static HostPool& get() { static HostPool* p = new HostPool; return *p; }
  template <class F> void run(size_t n, int width, F&& fn) {
    if (n == 0) return;
    if (width <= 1 || n == 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
    std::lock_guard<std::mutex> job_lock(job_mtx_);
    std::function<void(size_t)> f = fn;
    {
      std::lock_guard<std::mutex> l(mtx_);
      ensure(std::min<int>(width - 1, kMax));
      fn_ = &f; n_ = n; next_.store(0); active_ = std::min<int>(width - 1, (int)workers_.size()); pending_ = active_; gen_++;
    }
    cv_.notify_all();
    for (;;) { const size_t i = next_.fetch_add(1); if (i >= n) break; f(i); }
    std::unique_lock<std::mutex> l(mtx_);
    done_cv_.wait(l, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }
 private:
  static constexpr int kMax = 32;
  std::mutex job_mtx_, mtx_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> workers_;
  const std::function<void(size_t)>* fn_ = nullptr;
  size_t n_ = 0;
  std::atomic<size_t> next_{0};
  int active_ = 0, pending_ = 0;
  unsigned long gen_ = 0;
  bool stop_ = false;
  void ensure(int want) {
    while ((int)workers_.size() < want) {
      const int id = (int)workers_.size();
      workers_.emplace_back([this, id] {
        unsigned long seen = 0;
        for (;;) {
          std::unique_lock<std::mutex> l(mtx_);
          cv_.wait(l, [&] { return stop_ || (gen_ != seen && id < active_); });
          if (stop_) return;
          seen = gen_;
          const std::function<void(size_t)>* f = fn_;
          const size_t n = n_;
          l.unlock();
          for (;;) { const size_t i = next_.fetch_add(1); if (i >= n) break; (*f)(i); }
          l.lock();
          if (--pending_ == 0) done_cv_.notify_one();
        }
      });
    }
  }
  HostPool() = default;
  ~HostPool() {
    { std::lock_guard<std::mutex> l(mtx_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
};

// ONE persistent helper thread: submit(fn) runs fn on it, wait() blocks until it is done.  For a call that splits its batch in two halves and
// works on both at once (dvm_ba_optimize_windows_fast: the second half's tables and upload under the first half's kernel) -- the helper
// keeps its thread-local staging buffers between calls, which a std::thread per call would allocate and free every time.
class HelperThread {
 public:
  static HelperThread& get() { static HelperThread* h = new HelperThread; return *h; }   // (never destroyed: see HostPool)
  std::mutex use;                    // one user at a time (held by the caller from submit to wait)
  void submit(std::function<void()> fn) {
    { std::lock_guard<std::mutex> l(m_); job_ = std::move(fn); has_ = true; done_ = false; }
    if (!th_.joinable()) th_ = std::thread([this] { loop(); });
    cv_.notify_all();
  }
  void wait() { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return done_; }); }
 private:
  std::mutex m_;
  std::condition_variable cv_;
  std::function<void()> job_;
  bool has_ = false, done_ = true, stop_ = false;
  std::thread th_;
  void loop() {
    for (;;) {
      std::function<void()> f;
      { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return has_ || stop_; }); if (stop_) return; f = std::move(job_); has_ = false; }
      f();
      { std::lock_guard<std::mutex> l(m_); done_ = true; }
      cv_.notify_all();
    }
  }
  HelperThread() = default;
  ~HelperThread() { { std::lock_guard<std::mutex> l(m_); stop_ = true; } cv_.notify_all(); if (th_.joinable()) th_.join(); }
};

// Page-locked host memory a builder packs its tables into (one block per bundle-adjustment window, kept with the pooled builder): the
// block is sent to the device from where it is (Stage::in_pinned).  The staging copy of 30 MB of window tables -- a separate pass over
// data that had left the builders' caches -- was 1.2 ms of a 5.5 ms call.
struct PinnedArena {
  uint8_t* base = nullptr;
  size_t cap = 0, used = 0;
  int reserve(size_t bytes) {
    used = 0; claimed.store(0);
    if (bytes <= cap) return DVM_OK;
    if (base) hipHostFree(base);
    base = nullptr; cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 20);
    int rc = hip_check(hipHostMalloc(reinterpret_cast<void**>(&base), want, hipHostMallocPortable), "hipHostMalloc(window tables)");
    if (rc != DVM_OK) { base = nullptr; return rc; }
    cap = want;
    return DVM_OK;
  }
  // appends `bytes` from src at the next 256-byte boundary; returns the offset (-1: nothing to send)
  ptrdiff_t put(const void* src, size_t bytes) {
    if (!bytes) return -1;
    const size_t off = used;
    std::memcpy(base + off, src, bytes);
    used = off + ((bytes + 255) & ~(size_t)255);
    return (ptrdiff_t)off;
  }
  // several builders, one block (one DMA for all of them: 32 separate 1 MB copies cost 40 us each): claim() hands out a span, -1 when the
  // block is full (the builder then packs into an arena of its own); top() = what has been claimed
  std::atomic<size_t> claimed{0};
  ptrdiff_t claim(size_t bytes) {
    const size_t off = claimed.fetch_add(bytes);
    return off + bytes <= cap ? (ptrdiff_t)off : -1;
  }
  size_t top() const { return std::min(claimed.load(), cap); }
  PinnedArena() = default;
  PinnedArena(const PinnedArena&) = delete;
  PinnedArena& operator=(const PinnedArena&) = delete;
  PinnedArena(PinnedArena&& o) noexcept : base(o.base), cap(o.cap), used(o.used) { o.base = nullptr; o.cap = o.used = 0; }   // (claimed: a moved arena is not in use)
  PinnedArena& operator=(PinnedArena&& o) noexcept { if (this != &o) { if (base) hipHostFree(base); base = o.base; cap = o.cap; used = o.used; o.base = nullptr; o.cap = o.used = 0; } return *this; }
  ~PinnedArena() { if (base) hipHostFree(base); }
};

// Items come in two kinds.  COPIED (in / out / scratch): packed into the pinned buffer, one asynchronous H2D copy, kernels, one D2H
// copy -- for anything a kernel reads more than once or updates in place.  MAPPED (in_mapped / out_mapped): the kernel reads the
// input from / writes the output to page-locked host memory directly -- for arrays touched ONCE per call (a query list, a result
// list), where the copy command in front of / behind the kernel costs more than the kernel: a per-frame call with ~50 KB in and
// ~16 KB out is a chain of latencies, not a bandwidth problem.  A call whose items are all mapped queues no copy at all.
// Kernels of the convenience paths are launched on stream() -- one in-order chain per calling thread.
struct Stage {
  struct Item { const void* src; void* dst; size_t bytes, off; bool mapped; bool direct = false; };
  std::vector<Item> items;
  size_t total = 0, in_bytes = 0, mapped_total = 0;
  uint8_t* d = nullptr;
  StageCtx* ctx = nullptr;
  int add(const void* src, void* dst, size_t bytes, bool mapped = false) {
    items.push_back({src, dst, bytes, 0, mapped, false});
    return (int)items.size() - 1;
  }
  // an input that already lies in page-locked host memory (a PinnedArena): sent from where it is, no copy into the staging buffer
  int in_pinned(const void* src, size_t bytes) { const int i = add(src, nullptr, src ? bytes : 0); items[i].direct = true; return i; }
  int in_mapped(const void* src, size_t bytes) { return add(src, nullptr, src ? bytes : 0, true); }
  int out_mapped(void* dst, size_t bytes) { return add(nullptr, dst, dst ? bytes : 0, true); }
  hipStream_t stream() const { return ctx ? ctx->s : nullptr; }
  int in(const void* src, size_t bytes) { return add(src, nullptr, src ? bytes : 0); }
  int out(void* dst, size_t bytes) { return add(nullptr, dst, dst ? bytes : 0); }
  int scratch(size_t bytes) { return add(nullptr, nullptr, bytes); }   // device-only working memory
  static size_t pad(size_t b) { return (std::max<size_t>(b, 16) + 255) & ~(size_t)255; }
  // inputs first (one contiguous span to send), then outputs and scratch; ptr() is valid from here on.  A caller whose INPUTS hold
  // device addresses of other items (a table of views) calls layout() first, fills them in, then upload().
  int layout() {
    size_t off = 0, moff = 0;
    for (Item& it : items) if (it.src && !it.mapped && !it.direct) { it.off = off; off += pad(it.bytes); }
    in_bytes = off;
    for (Item& it : items) if (it.src && !it.mapped && it.direct) { it.off = off; off += pad(it.bytes); }
    for (Item& it : items) if (!it.src && !it.mapped) { it.off = off; off += pad(it.bytes); }
    total = off;
    for (Item& it : items) if (it.mapped) { it.off = moff; moff += pad(it.bytes); }
    mapped_total = moff;
    ctx = &stage_ctx();
    int rc = ctx->ensure(std::max<size_t>(total, 16));
    if (rc == DVM_OK && mapped_total) rc = ctx->ensure_mapped(mapped_total);
    if (rc != DVM_OK) return rc;
    d = ctx->d;
    return DVM_OK;
  }
  int upload() {
    int rc = d ? DVM_OK : layout();
    if (rc != DVM_OK) return rc;
    for (const Item& it : items)        // (first: their DMA runs under the host copies of the staged items below)
      if (it.direct && it.bytes && (rc = hip_check(hipMemcpyAsync(d + it.off, it.src, it.bytes, hipMemcpyHostToDevice, ctx->s), "upload")) != DVM_OK) return rc;
    if (in_bytes > ((size_t)8 << 20)) {
      // a large batch: the inputs go in four pieces, each piece's host copies (pooled threads) under the previous piece's DMA
      std::vector<size_t> idx;
      for (size_t i = 0; i < items.size(); i++) if (items[i].src && !items[i].mapped && !items[i].direct) idx.push_back(i);
      for (size_t i = 0; i < items.size(); i++) if (items[i].src && items[i].mapped && items[i].bytes) std::memcpy(ctx->hm + items[i].off, items[i].src, items[i].bytes);
      size_t first = 0;
      static const int np = std::getenv("DVM_STAGE_PIECES") ? std::max(1, atoi(std::getenv("DVM_STAGE_PIECES"))) : 4;
      static const int nt = std::getenv("DVM_STAGE_THREADS") ? std::max(1, atoi(std::getenv("DVM_STAGE_THREADS"))) : 8;
      for (int piece = 0; piece < np && first < idx.size(); piece++) {
        const size_t lo = items[idx[first]].off, want = piece == np - 1 ? in_bytes : (in_bytes * (piece + 1)) / np;
        size_t last = first;
        while (last < idx.size() && (piece == np - 1 || items[idx[last]].off + pad(items[idx[last]].bytes) <= want || last == first)) last++;
        HostPool::get().run(last - first, nt, [&](size_t j) { const Item& it = items[idx[first + j]]; if (it.bytes) std::memcpy(ctx->h + it.off, it.src, it.bytes); });
        const size_t hi = last < idx.size() ? items[idx[last]].off : in_bytes;
        rc = hip_check(hipMemcpyAsync(d + lo, ctx->h + lo, hi - lo, hipMemcpyHostToDevice, ctx->s), "upload");
        if (rc != DVM_OK) return rc;
        first = last;
      }
      return rc;
    }
    copy_items(true);
    if (in_bytes) rc = hip_check(hipMemcpyAsync(d, ctx->h, in_bytes, hipMemcpyHostToDevice, ctx->s), "upload");
    return rc;
  }
  // the host-side copies into / out of the pinned buffer; beyond 4 MB (a batch of bundle-adjustment windows: 30 MB) by up to eight threads --
  // one thread moves ~10 GB/s, the DMA engine behind it 55
  void copy_items(bool up) {
    size_t bytes = 0;
    for (const Item& it : items) if ((up ? (const void*)it.src : (const void*)it.dst) && it.bytes && !(up && it.direct)) bytes += it.bytes;
    auto one = [&](const Item& it) {
      uint8_t* stage = (it.mapped ? ctx->hm : ctx->h) + it.off;
      if (up) { if (it.src && it.bytes && !it.direct) std::memcpy(stage, it.src, it.bytes); }
      else if (it.dst && it.bytes) std::memcpy(it.dst, stage, it.bytes);
    };
    if (bytes <= ((size_t)4 << 20)) { for (const Item& it : items) one(it); return; }
    HostPool::get().run(items.size(), 8, [&](size_t i) { one(items[i]); });
  }
  template <class T> T* ptr(int i) const {
    return items[i].bytes ? reinterpret_cast<T*>((items[i].mapped ? ctx->hm_dev : d) + items[i].off) : nullptr;
  }
  int download() {
    size_t lo = total, hi = 0;
    for (const Item& it : items) if (it.dst && it.bytes && !it.mapped) { lo = std::min(lo, it.off); hi = std::max(hi, it.off + it.bytes); }
    int rc = DVM_OK;
    if (hi > lo) rc = hip_check(hipMemcpyAsync(ctx->h + lo, d + lo, hi - lo, hipMemcpyDeviceToHost, ctx->s), "download");
    if (rc == DVM_OK) rc = hip_check(hipStreamSynchronize(ctx->s), "sync");
    if (rc == DVM_OK) copy_items(false);
    return rc;
  }
};
}  // namespace dvm

