// dfx_relation.cpp -- Arrow C stream adapters at the library edge, FilterRelation, ProjectRelation
// and their C-ABI constructors.
#include "dfx_relation.hpp"
#include "dfx_sigs.hpp"

#include <errno.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <stdlib.h>
#include <string.h>

namespace dfx {

Status error_from_ctrl(uint32_t bits) {
  if (bits & 1u) return Status::Err(DFX_ARROW_ERROR, "DivideByZero");  // arrow 0.12 array_ops::divide
  if (bits & 2u) return Status::Err(DFX_INTERNAL_ERROR, "attempt to divide with overflow");
  if (bits & 4u) return Status::Err(DFX_INTERNAL_ERROR, "partitioned aggregation: LDS ring stalled");
  if (bits & 8u) return Status::Err(DFX_INTERNAL_ERROR, "single-pass filter: look-back stalled");
  return Status::OK();
}

// =================================================================================================
// host Arrow stream -> device batches
// =================================================================================================
namespace {

// Host Arrow batches -> HBM (relation.rs:34-54 is the reference's feed).  PCIe Gen5 x16 moves ~57 GB/s out of
// pinned memory here, HBM streams at > 6 TB/s: this relation is bound by the link whatever it does, and the three forms
// below differ by how close they get to it (tools/pin_probe.py, tools/host_stream_probe.py, bench.py
// host_streamed_pcie_inclusive; round 3):
//   * in order (default): copies on the library's stream, one synchronisation per batch, then the producer's array is
//     released.  HIP moves large pageable buffers by pinning them chunk-wise inside the runtime: 53-54 GB/s = 0.85 of the link;
//   * DFX_HOST_PREFETCH=1: batch i + 1 pulled and copied on a second stream while the consumer works on batch i, arrays
//     released on their copy event: 40-43 GB/s -- a pageable copy blocks its caller whichever stream it is queued on;
//   * + DFX_HOST_PIN=1: the producer's buffers page-locked in place (hipHostRegister) so that the DMA engine reads them
//     directly (57 GB/s for the copy alone): 40-46 GB/s end to end -- locking 256 MB costs 2.2 ms of the 4.7 ms its transfer
//     takes and does not overlap the transfer before it.
// In every form the producer's buffers are only read between get_next and release (tests/c_abi/host_stream.c poisons them on
// release), and columns nobody reads downstream never cross the link (require_columns).
class HostStreamRelation : public Relation {
 public:
  explicit HostStreamRelation(struct ArrowArrayStream* s) {
    stream_ = *s;  // move
    memset(s, 0, sizeof(*s));
  }
  ~HostStreamRelation() override {
    drop(&pending_);
    if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);  // (staged pieces still crossing the link read the pinned slots)
    for (hipEvent_t e : slot_event_)
      if (e) (void)hipEventDestroy(e);
    if (batch_event_) (void)hipEventDestroy(batch_event_);
    if (fence_) (void)hipEventDestroy(fence_);
    if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
    if (stream_.release) stream_.release(&stream_);
  }
  RelationKind kind() const override { return REL_HOST_STREAM; }
  void host_stream_options(const HostStreamOptions& o) override {
    if (started_ || opts_set_) return;  // the first operator above decides, before the first batch
    hopt_ = o;
    opts_set_ = true;
  }

  Status init() {
    struct ArrowSchema as;
    memset(&as, 0, sizeof(as));
    const int rc = stream_.get_schema(&stream_, &as);
    if (rc != 0) return stream_error(rc, "get_schema");
    Status st = schema_from_arrow(&as, &schema_);
    if (as.release) as.release(&as);
    return st;
  }

  const SchemaInfo& schema() const override { return schema_; }
  void require_columns(const std::vector<char>& needed) override { needed_ = needed; }
  void explain(std::string* out, int depth) const override {
    int n = 0;
    for (size_t i = 0; i < schema_.fields.size(); ++i) n += (needed_.empty() || needed_[i]) ? 1 : 0;
    explain_line(out, depth, strfmt("HostStream: host Arrow batches, %d of %d columns uploaded per batch (%s%s)", n, (int)schema_.fields.size(),
                                    mode() == 1 ? "pinned staging ring filled by library threads, DMA on a copy stream" :
                                    mode() >= 2 ? "one batch ahead, own copy stream" : "in order on the library's stream",
                                    mode() == 3 ? ", large buffers page-locked in place" : ""));
  }

  Status next(DeviceBatch* out, bool* has) override {
    *has = false;
    DFX_RETURN_IF_ERROR(ensure_init());
    if (!opts_set_) {  // no operator above brought its own option set: the process defaults
      const AggOptions& d = agg_options();
      hopt_.mode = d.host_stream;
      hopt_.threads = d.host_stage_threads;
      hopt_.piece_mb = d.host_stage_mb;
      hopt_.slots = d.host_stage_slots;
      opts_set_ = true;
    }
    prefetch_ = mode() >= 2;
    pin_in_place_ = mode() == 3;
    if (mode() == 1) {
      started_ = true;
      return next_staged(out, has);
    }
    if (!prefetch_) {
      started_ = true;
      return next_in_order(out, has);
    }
    if (!copy_stream_) DFX_HIP(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
    if (!started_) {  // the first batch: nothing to overlap it with yet
      started_ = true;
      DFX_RETURN_IF_ERROR(fetch(&pending_));
    }
    if (!pending_.valid) {
      Status st = pending_.error;  // an error met while prefetching surfaces when ITS batch is asked for
      pending_.error = Status::OK();
      return st;
    }
    InFlight cur;
    std::swap(cur, pending_);
    Status ahead = fetch(&pending_);  // queue the NEXT batch's copies behind this one's before waiting
    if (!ahead.ok()) {
      drop(&pending_);
      pending_.error = ahead;
    }
    Status st = cur.upload;
    if (cur.event) {  // the copies of this batch have read the producer's buffers: only now may they be released
      hipError_t e = hipEventSynchronize(cur.event);
      if (e != hipSuccess && st.ok()) st = Status::Err(DFX_EXECUTION_ERROR, strfmt("HIP error %s after H2D", hipGetErrorString(e)));
    }
    DeviceBatch b = std::move(cur.batch);
    drop(&cur);
    if (!st.ok()) return st;
    *out = std::move(b);
    *has = true;
    return Status::OK();
  }

  // The default: a batch is copied on the library's own stream when it is asked for, the producer's array is released when
  // the stream has passed the copies.  HIP copies large pageable buffers by pinning them chunk-wise inside the runtime:
  // 53-54 GB/s here = 0.85 of the link, which is what this path delivers end to end.
  Status next_in_order(DeviceBatch* out, bool* has) {
    InFlight f;
    copy_stream_in_use_ = ctx().stream;
    Status st = fetch_into(&f, /*fence=*/false);
    if (!st.ok() || !f.valid) {
      drop(&f);
      return st;
    }
    st = f.upload;
    {  // host buffers are borrowed until here -- also when upload failed part-way: earlier columns' copies may be queued
      hipError_t e = hipStreamSynchronize(ctx().stream);
      if (e != hipSuccess && st.ok()) st = Status::Err(DFX_EXECUTION_ERROR, strfmt("HIP error %s after H2D", hipGetErrorString(e)));
    }
    DeviceBatch b = std::move(f.batch);
    drop(&f);
    if (!st.ok()) return st;
    *out = std::move(b);
    *has = true;
    return Status::OK();
  }

  // The staged form (host.stream = 1; NOT the default: measured slower than HIP's own pageable copy on this platform, see
  // HostStreamOptions).  The DMA engine reads PINNED host memory at 57 GB/s and pageable memory not at all: HIP's own
  // copy of a pageable buffer pins it chunk-wise inside the runtime (53-54 GB/s, the calling thread blocked throughout), locking
  // the producer's pages in place costs half the transfer's time (tools/pin_probe.py).  Here the library owns a ring of pinned
  // slots; `threads` library threads copy the producer's buffers into slots piece by piece (one thread fills at ~29 GB/s: it
  // takes two to four to outrun the engine) and queue each slot's DMA on a copy stream as soon as it is full, so the engine
  // drains slot i while slots i + 1 ... are being filled.  The producer's buffers are read by those memcpys only: the array
  // is released when the threads have joined, with the last slots still crossing the link; the consumer's kernels wait for
  // the batch's copy event on the library's stream -- no host synchronisation at all.
  struct InFlight {
    bool valid = false;
    struct ArrowArray arr;        // the producer's batch, borrowed until `event` fires
    DeviceBatch batch;
    hipEvent_t event = nullptr;
    std::vector<void*> registered;  // host ranges page-locked for this batch
    Status upload, error;
    InFlight() { memset(&arr, 0, sizeof(arr)); }
  };
  struct Piece {
    const uint8_t* host;
    uint8_t* dev;
    size_t bytes;
  };
  Status next_staged(DeviceBatch* out, bool* has) {
    if (!copy_stream_) DFX_HIP(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
    InFlight f;
    copy_stream_in_use_ = copy_stream_;
    pieces_.clear();
    staging_ = true;
    Status st = fetch_staged(&f);
    staging_ = false;
    if (!st.ok() || !f.valid) {
      drop(&f);
      return st;
    }
    st = f.upload;
    if (st.ok()) st = run_pieces();
    hipError_t e = hipSuccess;
    if (st.ok()) {  // the consumer's kernels (library stream) start when this batch's last piece has landed
      if (!batch_event_) e = hipEventCreateWithFlags(&batch_event_, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventRecord(batch_event_, copy_stream_);
      if (e == hipSuccess) e = hipStreamWaitEvent(ctx().stream, batch_event_, 0);
      if (e != hipSuccess) st = Status::Err(DFX_EXECUTION_ERROR, strfmt("HIP error %s after H2D", hipGetErrorString(e)));
    } else {
      (void)hipStreamSynchronize(copy_stream_);  // pieces already queued read pinned slots, not the producer: nothing else to wait for
    }
    DeviceBatch b = std::move(f.batch);
    drop(&f);  // every byte of the producer's buffers has been copied out by now
    if (!st.ok()) return st;
    *out = std::move(b);
    *has = true;
    return Status::OK();
  }
  Status fetch_staged(InFlight* f) {
    if (done_) return Status::OK();
    const int rc = stream_.get_next(&stream_, &f->arr);
    if (rc != 0) {
      memset(&f->arr, 0, sizeof(f->arr));
      return stream_error(rc, "get_next");
    }
    if (f->arr.release == nullptr) {
      done_ = true;
      return Status::OK();
    }
    f->valid = true;
    // device buffers come from the pool: a consumer's kernels that still use them may be queued on the library's stream
    if (!fence_) DFX_HIP(hipEventCreateWithFlags(&fence_, hipEventDisableTiming));
    DFX_HIP(hipEventRecord(fence_, ctx().stream));
    DFX_HIP(hipStreamWaitEvent(copy_stream_, fence_, 0));
    f->upload = upload(f->arr, &f->batch, f);  // (h2d only lists the pieces while staging_)
    return Status::OK();
  }
  Status ensure_ring() {
    const size_t piece = (size_t)std::max(1, hopt_.piece_mb) << 20;
    const int slots = std::max(2, std::min(64, hopt_.slots));
    if (ring_ && ring_piece_ == piece && (int)slot_event_.size() == slots) return Status::OK();
    if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);
    Status st;
    ring_ = pinned_alloc(piece * (size_t)slots, &st);
    if (!ring_) return st;
    ring_piece_ = piece;
    for (hipEvent_t e : slot_event_)
      if (e) (void)hipEventDestroy(e);
    slot_event_.assign((size_t)slots, nullptr);
    for (auto& e : slot_event_) DFX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    slot_gen_.assign((size_t)slots, 0);
    return Status::OK();
  }
  Status run_pieces() {
    if (pieces_.empty()) return Status::OK();
    DFX_RETURN_IF_ERROR(ensure_ring());
    const size_t n = pieces_.size(), R = slot_event_.size();
    size_t total = 0;
    for (const Piece& p : pieces_) total += p.bytes;
    std::fill(slot_gen_.begin(), slot_gen_.end(), 0);
    std::atomic<size_t> next{0};
    std::mutex mu;
    std::condition_variable cv;
    Status first_error = Status::OK();
    const int device = ctx().device;
    uint8_t* const ring = (uint8_t*)ring_.get();
    auto worker = [&]() {
      (void)hipSetDevice(device);  // (the current device is a per-thread setting)
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= n) break;
        const size_t sl = i % R, turn = i / R;
        {  // my slot's previous occupant (piece i - R, taken earlier by some thread) has been queued ...
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return slot_gen_[sl] == turn; });
        }
        hipError_t e = hipSuccess;
        if (slot_used_ || turn > 0) e = hipEventSynchronize(slot_event_[sl]);  // ... and has left the slot
        const Piece& p = pieces_[i];
        if (e == hipSuccess) {
          memcpy(ring + sl * ring_piece_, p.host, p.bytes);
          e = hipMemcpyAsync(p.dev, ring + sl * ring_piece_, p.bytes, hipMemcpyHostToDevice, copy_stream_);
        }
        if (e == hipSuccess) e = hipEventRecord(slot_event_[sl], copy_stream_);
        {
          std::lock_guard<std::mutex> lk(mu);
          if (e != hipSuccess && first_error.ok()) first_error = Status::Err(DFX_EXECUTION_ERROR, strfmt("HIP error %s in the staged H2D copy", hipGetErrorString(e)));
          slot_gen_[sl] = turn + 1;
        }
        cv.notify_all();
      }
    };
    // small batches (the reference's 1024-row batches): the calling thread alone -- starting threads costs more than the copy
    const int threads = total < ((size_t)4 << 20) ? 1 : std::max(1, std::min(16, hopt_.threads));
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (std::thread& t : pool) t.join();
    slot_used_ = true;
    counters().h2d_staged_bytes += (long long)total;
    return first_error;
  }

 private:

  void drop(InFlight* f) {  // unpin, hand the array back to the producer
    if (f->event) {
      (void)hipEventSynchronize(f->event);
      (void)hipEventDestroy(f->event);
      f->event = nullptr;
    }
    for (void* p : f->registered) (void)hipHostUnregister(p);
    f->registered.clear();
    if (f->arr.release) f->arr.release(&f->arr);
    memset(&f->arr, 0, sizeof(f->arr));
    f->batch = DeviceBatch();
    f->valid = false;
  }

  Status fetch(InFlight* f) {
    copy_stream_in_use_ = copy_stream_;
    return fetch_into(f, /*fence=*/true);
  }
  // pull one batch from the producer and queue its copies (f->valid stays false at the end of the stream)
  Status fetch_into(InFlight* f, bool fence) {
    if (done_) return Status::OK();
    const int rc = stream_.get_next(&stream_, &f->arr);
    if (rc != 0) {
      memset(&f->arr, 0, sizeof(f->arr));
      return stream_error(rc, "get_next");
    }
    if (f->arr.release == nullptr) {  // end of stream == Ok(None)
      done_ = true;
      return Status::OK();
    }
    f->valid = true;
    if (!fence) {  // copies on the library's own stream: in order with everything else, the caller synchronises it
      f->upload = upload(f->arr, &f->batch, f);
      return Status::OK();
    }
    {  // The device buffers of this batch come from the pool: they may have been handed back by a consumer whose kernels are
       // still queued on the library's stream.  The copies wait for everything that stream holds right now.
      if (!fence_) DFX_HIP(hipEventCreateWithFlags(&fence_, hipEventDisableTiming));
      DFX_HIP(hipEventRecord(fence_, ctx().stream));
      DFX_HIP(hipStreamWaitEvent(copy_stream_, fence_, 0));
    }
    f->upload = upload(f->arr, &f->batch, f);
    hipError_t e = hipEventCreateWithFlags(&f->event, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(f->event, copy_stream_);
    if (e != hipSuccess && f->upload.ok()) f->upload = Status::Err(DFX_EXECUTION_ERROR, strfmt("HIP error %s after H2D", hipGetErrorString(e)));
    // the consumer's kernels run on the library's stream: they start when the copies have landed
    if (f->event) (void)hipStreamWaitEvent(ctx().stream, f->event, 0);
    return Status::OK();
  }

  Status stream_error(int rc, const char* what) {
    const char* m = stream_.get_last_error ? stream_.get_last_error(&stream_) : nullptr;
    // our own streams return a dfx_status; foreign producers an errno
    const int code = (rc > 0 && rc <= DFX_EXECUTION_ERROR) ? rc : DFX_IO_ERROR;
    return Status::Err(code, m ? std::string(m) : strfmt("input stream %s failed with code %d", what, rc));
  }

  Status h2d(const void* host, size_t bytes, std::shared_ptr<void>* dev, InFlight* f) {
    Status st;
    *dev = device_alloc(bytes ? bytes : 8, &st);
    if (!*dev) return st;
    if (staging_) {  // staged form: only list what has to travel (run_pieces moves it)
      for (size_t at = 0; at < bytes; at += ring_piece_bytes()) {
        Piece p;
        p.host = (const uint8_t*)host + at;
        p.dev = (uint8_t*)dev->get() + at;
        p.bytes = std::min(ring_piece_bytes(), bytes - at);
        pieces_.push_back(p);
      }
      counters().h2d_bytes += (long long)bytes;
      return Status::OK();
    }
    if (pin_in_place_ && bytes >= kPinThreshold) {
      if (hipHostRegister(const_cast<void*>(host), bytes, hipHostRegisterDefault) == hipSuccess) f->registered.push_back(const_cast<void*>(host));
      else (void)hipGetLastError();  // not lockable (already registered, overlapping pages ...): the staged copy below still works
    }
    if (bytes) DFX_HIP(hipMemcpyAsync(dev->get(), host, bytes, hipMemcpyHostToDevice, copy_stream_in_use_));
    counters().h2d_bytes += (long long)bytes;
    return Status::OK();
  }

  Status upload(const struct ArrowArray& arr, DeviceBatch* out, InFlight* f) {
    if ((size_t)arr.n_children != schema_.fields.size())
      return Status::Err(DFX_ARROW_ERROR, strfmt("batch has %lld columns, schema has %zu", (long long)arr.n_children, schema_.fields.size()));
    out->num_rows = arr.length;
    out->columns.clear();
    out->columns.resize(schema_.fields.size());
    for (size_t ci = 0; ci < schema_.fields.size(); ++ci) {
      const struct ArrowArray* c = arr.children[ci];
      const int dt = schema_.fields[ci].dtype;
      DeviceColumn& d = out->columns[ci];
      const int64_t off = arr.offset + c->offset;
      const int64_t n = arr.length;
      d.dtype = dt;
      d.length = n;
      if (ci < needed_.size() && !needed_[ci]) {  // projection push-down: never read downstream, so never crosses PCIe
        d.absent = true;
        continue;
      }
      d.bit_offset = off & 7;
      const uint8_t* validity = (c->n_buffers > 0) ? (const uint8_t*)c->buffers[0] : nullptr;
      if (validity && c->null_count != 0) {
        std::shared_ptr<void> dv;
        const int64_t b0 = off >> 3, b1 = (off + n + 7) >> 3;
        DFX_RETURN_IF_ERROR(h2d(validity + b0, (size_t)(b1 - b0), &dv, f));
        d.validity = (const uint8_t*)dv.get();
        d.owners.push_back(dv);
        d.null_count = c->null_count < 0 ? -1 : c->null_count;
      }
      if (dt == DFX_UTF8) {
        if (c->n_buffers < 3) return Status::Err(DFX_ARROW_ERROR, "Utf8 array without 3 buffers");
        // producers may export a zero-length string array with a null (or zero-sized) offsets buffer: offsets = {0}
        static const int32_t kZeroOffset[1] = {0};
        const bool no_offsets = c->buffers[1] == nullptr;
        if (no_offsets && n != 0) return Status::Err(DFX_ARROW_ERROR, "Utf8 array without an offsets buffer");
        const int32_t* offs = no_offsets ? kZeroOffset : (const int32_t*)c->buffers[1] + off;
        const uint8_t* data = (const uint8_t*)c->buffers[2];
        std::shared_ptr<void> doff, ddata;
        DFX_RETURN_IF_ERROR(h2d(offs, sizeof(int32_t) * (size_t)(n + 1), &doff, f));
        const int32_t o0 = offs[0], o1 = offs[n];
        if (o1 < o0 || (o1 > o0 && !data)) return Status::Err(DFX_ARROW_ERROR, "Utf8 array with inconsistent offsets");
        DFX_RETURN_IF_ERROR(h2d(data ? data + o0 : nullptr, (size_t)(o1 - o0), &ddata, f));
        d.offsets = (const int32_t*)doff.get();
        d.data = (const uint8_t*)ddata.get() - o0;  // raw offsets index straight into it
        d.data_bytes = o1 - o0;
        d.owners.push_back(doff);
        d.owners.push_back(ddata);
      } else if (dt == DFX_BOOLEAN) {
        if (c->n_buffers < 2) return Status::Err(DFX_ARROW_ERROR, "Boolean array without 2 buffers");
        std::shared_ptr<void> dv;
        const int64_t b0 = off >> 3, b1 = (off + n + 7) >> 3;
        DFX_RETURN_IF_ERROR(h2d((const uint8_t*)c->buffers[1] + b0, (size_t)(b1 - b0), &dv, f));
        d.values = dv.get();
        d.owners.push_back(dv);
      } else {
        if (c->n_buffers < 2) return Status::Err(DFX_ARROW_ERROR, "primitive array without 2 buffers");
        const int w = dtype_width(dt);
        std::shared_ptr<void> dv;
        DFX_RETURN_IF_ERROR(h2d((const uint8_t*)c->buffers[1] + (size_t)off * w, (size_t)n * w, &dv, f));
        d.values = dv.get();
        d.owners.push_back(dv);
      }
    }
    return Status::OK();
  }

  static constexpr size_t kPinThreshold = (size_t)1 << 20;  // smaller buffers: the staged copy costs less than locking pages
  // DFX_HOST_PIN=1: page-lock the producer's large buffers in place (hipHostRegister) and let the DMA engine read them
  // directly.  Off by default: measured end to end it LOSES (bench.py host_streamed_pcie_inclusive: 46 GB/s against 53-54
  // for HIP's own staged copy of pageable memory) although the copy out of registered memory alone is faster (57 GB/s,
  // tools/pin_probe.py) -- locking 256 MB costs 2.2 ms of the 4.7 ms its transfer takes, and it does not overlap the
  // transfer of the batch before.
  bool pin_in_place_ = false;  // HostStreamOptions::mode == 3 ("host.stream"; the DFX_HOST_PIN environment switch of round 3 is gone)
  // DFX_HOST_PREFETCH=1: batch i + 1 is pulled from the producer and copied on a stream of its own while the consumer works
  // on batch i, the producer's array released on the copy's event (no synchronisation of the compute stream).  Off by
  // default for the same reason: 40-43 GB/s end to end against 53 for the in-order form (tools/host_stream_probe.py) -- a
  // pageable copy blocks the calling thread whichever stream it is queued on, so nothing overlaps, and the second stream
  // costs the runtime's pinned-chunk pipeline its rhythm.
  bool prefetch_ = false;  // HostStreamOptions::mode >= 2 (was DFX_HOST_PREFETCH)
  HostStreamOptions hopt_;
  bool opts_set_ = false;
  int mode() const { return hopt_.mode < 0 || hopt_.mode > 3 ? 0 : hopt_.mode; }
  size_t ring_piece_bytes() const { return (size_t)std::max(1, hopt_.piece_mb) << 20; }
  // staged form
  bool staging_ = false;
  std::vector<Piece> pieces_;
  std::shared_ptr<void> ring_;
  size_t ring_piece_ = 0;
  std::vector<hipEvent_t> slot_event_;
  std::vector<size_t> slot_gen_;
  bool slot_used_ = false;
  hipEvent_t batch_event_ = nullptr;
  hipStream_t copy_stream_in_use_ = nullptr;
  struct ArrowArrayStream stream_;
  SchemaInfo schema_;
  std::vector<char> needed_;
  hipStream_t copy_stream_ = nullptr;
  hipEvent_t fence_ = nullptr;
  InFlight pending_;  // the batch that is crossing PCIe while the consumer works on the one before
  bool started_ = false, done_ = false;
};

// =================================================================================================
// device relation -> host Arrow stream
// =================================================================================================
struct ExportedStream {
  std::unique_ptr<Relation> rel;
  std::string last_error;
};

struct ArrayPriv {
  std::vector<std::shared_ptr<void>> pinned; // large result buffers: pooled pinned memory (fast D2H)
  std::vector<void*> host_buffers;           // malloc'd, 64-byte aligned
  std::vector<const void*> buffer_ptrs;      // this array's buffers
  std::vector<struct ArrowArray> kids;
  std::vector<struct ArrowArray*> kid_ptrs;
};

void release_array(struct ArrowArray* a) {
  if (!a || !a->release) return;
  ArrayPriv* p = (ArrayPriv*)a->private_data;
  for (auto& k : p->kids)
    if (k.release) k.release(&k);
  for (void* b : p->host_buffers) free(b);
  delete p;
  a->release = nullptr;
}

void* host_alloc(size_t bytes) {
  void* p = nullptr;
  const size_t cap = ((bytes ? bytes : 1) + 63) / 64 * 64;  // padded to 64 bytes, tail zeroed
  if (posix_memalign(&p, 64, cap) != 0) return nullptr;
  memset((uint8_t*)p + (cap - 64), 0, 64);
  return p;
}

// result buffer owned by the exported array: pinned (pooled) when large, so the D2H copy runs at
// PCIe speed instead of through a pageable staging copy
void* alloc_result(ArrayPriv* p, size_t bytes) {
  ScopedUs t_alloc(&counters().export_alloc_us);
  if (bytes >= (1u << 16)) {
    Status st;
    std::shared_ptr<void> b = pinned_alloc(bytes + 64, &st);
    if (b) {
      p->pinned.push_back(b);
      return b.get();
    }
  }
  void* raw = host_alloc(bytes);
  if (raw) p->host_buffers.push_back(raw);
  return raw;
}

// move `n` bits starting at src bit `off` to bit 0 of dst (dst pre-zeroed)
void realign_bits(const uint8_t* src, int64_t off, int64_t n, uint8_t* dst) {
  for (int64_t i = 0; i < n; ++i)
    if ((src[(off + i) >> 3] >> ((off + i) & 7)) & 1) dst[i >> 3] |= (uint8_t)(1u << (i & 7));
}

Status download_column(const DeviceColumn& c, struct ArrowArray* out, std::vector<std::function<void()>>* fixups) {
  ArrayPriv* p = new ArrayPriv();
  memset(out, 0, sizeof(*out));
  out->private_data = p;
  out->release = release_array;
  out->length = c.length;
  out->offset = 0;
  const int64_t n = c.length;
  hipStream_t s = ctx().stream;
  // validity
  void* vbuf = nullptr;
  if (c.validity && c.null_count != 0) {
    const size_t bytes = (size_t)((c.bit_offset + n + 7) >> 3);
    void* raw = alloc_result(p, bytes);
    if (!raw) return Status::Err(DFX_EXECUTION_ERROR, "host allocation failed");
    DFX_HIP(hipMemcpyAsync(raw, c.validity, bytes, hipMemcpyDeviceToHost, s));
    vbuf = raw;
    if (c.bit_offset != 0) {
      void* al = alloc_result(p, (size_t)((n + 7) >> 3));
      memset(al, 0, (size_t)((n + 7) >> 3));
      const int64_t bo = c.bit_offset;
      fixups->push_back([raw, al, bo, n]() { realign_bits((const uint8_t*)raw, bo, n, (uint8_t*)al); });
      vbuf = al;
    }
    ArrowArray* oo = out;
    fixups->push_back([oo, vbuf, n]() {  // count nulls once the bits are on the host
      int64_t set = 0;
      const uint8_t* b = (const uint8_t*)vbuf;
      for (int64_t i = 0; i < n; ++i) set += (b[i >> 3] >> (i & 7)) & 1;
      oo->null_count = n - set;
    });
  }
  p->buffer_ptrs.push_back(vbuf);
  if (c.dtype == DFX_UTF8) {
    void* obuf = alloc_result(p, sizeof(int32_t) * (size_t)(n + 1));
    if (!obuf) return Status::Err(DFX_EXECUTION_ERROR, "host allocation failed");
    if (c.offsets) DFX_HIP(hipMemcpyAsync(obuf, c.offsets, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToHost, s));
    else memset(obuf, 0, sizeof(int32_t) * (size_t)(n + 1));
    p->buffer_ptrs.push_back(obuf);
    p->buffer_ptrs.push_back(nullptr);  // data: sized from the offsets once they are on the host
    const uint8_t* dev_data = c.data;
    fixups->push_back([obuf, p, dev_data, n]() {  // fetch the referenced bytes, rebase offsets to 0
      int32_t* o = (int32_t*)obuf;
      const int32_t o0 = o[0];
      const int64_t nbytes = (int64_t)o[n] - o0;
      void* dbuf = alloc_result(p, (size_t)(nbytes > 0 ? nbytes : 1));
      if (nbytes > 0 && dev_data) (void)hipMemcpy(dbuf, dev_data + o0, (size_t)nbytes, hipMemcpyDeviceToHost);
      if (o0 != 0)
        for (int64_t i = 0; i <= n; ++i) o[i] -= o0;
      p->buffer_ptrs[2] = dbuf;
    });
    out->n_buffers = 3;
  } else if (c.dtype == DFX_BOOLEAN) {
    const size_t bytes = (size_t)((c.bit_offset + n + 7) >> 3);
    void* raw = alloc_result(p, bytes);
    if (!raw) return Status::Err(DFX_EXECUTION_ERROR, "host allocation failed");
    if (n) DFX_HIP(hipMemcpyAsync(raw, c.values, bytes, hipMemcpyDeviceToHost, s));
    void* vals = raw;
    if (c.bit_offset != 0) {
      void* al = alloc_result(p, (size_t)((n + 7) >> 3));
      memset(al, 0, (size_t)((n + 7) >> 3));
      const int64_t bo = c.bit_offset;
      fixups->push_back([raw, al, bo, n]() { realign_bits((const uint8_t*)raw, bo, n, (uint8_t*)al); });
      vals = al;
    }
    p->buffer_ptrs.push_back(vals);
    out->n_buffers = 2;
  } else {
    const size_t bytes = (size_t)n * dtype_width(c.dtype);
    if (c.host_values && c.host_values_of == c.values && c.host_bytes == bytes && bytes) {  // already on the host (see DeviceColumn)
      p->pinned.push_back(c.host_values);
      p->buffer_ptrs.push_back(c.host_values.get());
      out->n_buffers = 2;
      out->buffers = p->buffer_ptrs.data();
      out->null_count = 0;
      ++counters().export_host_ready;
      return Status::OK();
    }
    void* raw = alloc_result(p, bytes);
    if (!raw) return Status::Err(DFX_EXECUTION_ERROR, "host allocation failed");
    // large fixed-width result columns (pinned destination): copied by a kernel on the query's stream when the option
    // says so (export.kernel_copy; the copy engines' path has sporadic multi-millisecond stalls on these boxes)
    if (bytes >= (1u << 16) && agg_options().export_kernel_copy && !p->pinned.empty() && p->pinned.back().get() == raw) {
      DFX_HIP(launch_copy_to_host(c.values, raw, bytes, s));
    } else if (bytes) {
      DFX_HIP(hipMemcpyAsync(raw, c.values, bytes, hipMemcpyDeviceToHost, s));
    }
    p->buffer_ptrs.push_back(raw);
    out->n_buffers = 2;
  }
  out->buffers = p->buffer_ptrs.data();
  out->null_count = 0;
  return Status::OK();
}

Status download_batch(const DeviceBatch& b, struct ArrowArray* out) {
  ScopedUs t_export(&counters().export_us);
  ArrayPriv* p = new ArrayPriv();
  memset(out, 0, sizeof(*out));
  out->private_data = p;
  out->release = release_array;
  out->length = b.num_rows;
  p->kids.resize(b.columns.size());
  p->kid_ptrs.resize(b.columns.size());
  std::vector<std::function<void()>> fixups;
  for (size_t i = 0; i < b.columns.size(); ++i) {
    memset(&p->kids[i], 0, sizeof(struct ArrowArray));
    p->kid_ptrs[i] = &p->kids[i];
  }
  Status st;
  for (size_t i = 0; i < b.columns.size() && st.ok(); ++i) st = download_column(b.columns[i], &p->kids[i], &fixups);
  if (st.ok()) {
    hipError_t e = hipStreamSynchronize(ctx().stream);
    if (e != hipSuccess) st = Status::Err(DFX_EXECUTION_ERROR, strfmt("HIP error %s after D2H", hipGetErrorString(e)));
  }
  if (!st.ok()) {
    release_array(out);
    return st;
  }
  for (auto& f : fixups) f();
  p->buffer_ptrs.push_back(nullptr);  // struct validity
  out->n_buffers = 1;
  out->buffers = p->buffer_ptrs.data();
  out->n_children = (int64_t)p->kids.size();
  out->children = p->kid_ptrs.empty() ? nullptr : p->kid_ptrs.data();
  return Status::OK();
}

int exported_get_schema(struct ArrowArrayStream* s, struct ArrowSchema* out) {
  ExportedStream* es = (ExportedStream*)s->private_data;
  schema_to_arrow(es->rel->schema(), out);
  return 0;
}

int exported_get_next(struct ArrowArrayStream* s, struct ArrowArray* out) {
  ExportedStream* es = (ExportedStream*)s->private_data;
  memset(out, 0, sizeof(*out));
  DeviceBatch b;
  bool has = false;
  Status st;
  try {
    st = es->rel->next(&b, &has);
    if (st.ok() && has) st = download_batch(b, out);
  } catch (const std::exception& e) {  // nothing unwinds across the C ABI
    st = Status::Err(DFX_INTERNAL_ERROR, std::string("internal exception: ") + e.what());
  } catch (...) {
    st = Status::Err(DFX_INTERNAL_ERROR, "internal exception");
  }
  if (!st.ok()) {
    es->last_error = st.msg;
    if (out->release) out->release(out);
    memset(out, 0, sizeof(*out));
    return st.code;
  }
  return 0;  // released (zeroed) array == end of stream
}

const char* exported_get_last_error(struct ArrowArrayStream* s) {
  ExportedStream* es = (ExportedStream*)s->private_data;
  return es->last_error.empty() ? nullptr : es->last_error.c_str();
}

void exported_release(struct ArrowArrayStream* s) {
  if (!s || !s->release) return;
  delete (ExportedStream*)s->private_data;
  s->release = nullptr;
  s->private_data = nullptr;
}

}  // namespace

void export_relation(std::unique_ptr<Relation> rel, struct ArrowArrayStream* out) {
  ExportedStream* es = new ExportedStream();
  es->rel = std::move(rel);
  memset(out, 0, sizeof(*out));
  out->get_schema = exported_get_schema;
  out->get_next = exported_get_next;
  out->get_last_error = exported_get_last_error;
  out->release = exported_release;
  out->private_data = es;
}

Relation* peek_exported(struct ArrowArrayStream* s) {
  if (!s || s->release != exported_release) return nullptr;
  return ((ExportedStream*)s->private_data)->rel.get();
}

Status adopt_input_stream(struct ArrowArrayStream* input, std::unique_ptr<Relation>* out) {
  if (!input || !input->release) return Status::Err(DFX_GENERAL, "input stream is null or released");
  if (input->release == exported_release) {  // one of ours: stay on the device
    ExportedStream* es = (ExportedStream*)input->private_data;
    *out = std::move(es->rel);
    delete es;
    memset(input, 0, sizeof(*input));
    return Status::OK();
  }
  std::unique_ptr<HostStreamRelation> h(new HostStreamRelation(input));
  DFX_RETURN_IF_ERROR(h->init());
  *out = std::move(h);
  return Status::OK();
}

// =================================================================================================
// FilterRelation
// =================================================================================================
FilterRelation::FilterRelation(std::unique_ptr<Relation> input, const dfx_runtime_expr& expr, SchemaInfo schema, OptionOverrides options)
    : input_(std::move(input)), expr_(expr), schema_(std::move(schema)) {
  opt_.overrides = std::move(options);
  builder_.reset(new ProgramBuilder(input_->schema()));
  int dt = DFX_TYPE_NONE;
  deferred_ = builder_->add(expr_, expr_.root, &pred_operand_, &dt);
  if (deferred_.ok() && dt != DFX_BOOLEAN)  // filter.rs:64-66
    deferred_ = Status::Err(DFX_EXECUTION_ERROR, "Filter expression did not evaluate to boolean");
  memset(&fast_, 0, sizeof(fast_));
  if (!deferred_.ok() && program_limit_error(deferred_)) deferred_ = build_parts();
  else if (deferred_.ok()) builder_->build_fast(pred_operand_, nullptr, 0, nullptr, 0, &fast_);
}

// the predicate does not fit one fused program: split its top-level AND chain
Status FilterRelation::build_parts() {
  const Status whole = deferred_;
  std::vector<int32_t> conj;  // roots of the conjuncts, left to right
  {
    std::vector<int32_t> stack{expr_.root};
    while (!stack.empty()) {
      const int32_t at = stack.back();
      stack.pop_back();
      if (at < 0 || at >= (int32_t)expr_.nodes.size()) return whole;
      const dfx_expr_node& n = expr_.nodes[(size_t)at];
      if (n.kind == DFX_EXPR_BINARY && n.op == DFX_OP_AND) {
        stack.push_back(n.right);
        stack.push_back(n.left);
      } else {
        conj.push_back(at);
      }
    }
  }
  if (conj.size() < 2) return whole;  // nothing to split (one oversized comparison / OR tree)
  // AND chain over conj[from, to) as an expression of its own (the original nodes plus the new AND nodes)
  auto chain = [&](size_t from, size_t to) {
    dfx_runtime_expr e = expr_;
    int32_t root = conj[from];
    for (size_t i = from + 1; i < to; ++i) {
      dfx_expr_node a;
      memset(&a, 0, sizeof(a));
      a.kind = DFX_EXPR_BINARY;
      a.op = DFX_OP_AND;
      a.dtype = DFX_BOOLEAN;
      a.left = root;
      a.right = conj[i];
      a.column = -1;
      e.nodes.push_back(a);
      e.strings.emplace_back();
      e.has_name.push_back(0);
      root = (int32_t)e.nodes.size() - 1;
    }
    e.root = root;
    e.rebind();
    return e;
  };
  std::vector<Part> parts;
  size_t from = 0;
  while (from < conj.size()) {
    Part best;
    size_t best_to = from;
    for (size_t to = from + 1; to <= conj.size(); ++to) {  // the longest prefix of the remaining conjuncts that fits
      Part p;
      p.builder.reset(new ProgramBuilder(input_->schema()));
      memset(&p.fast, 0, sizeof(p.fast));
      const dfx_runtime_expr e = chain(from, to);
      int dt = DFX_TYPE_NONE;
      Status st = p.builder->add(e, e.root, &p.operand, &dt);
      if (!st.ok()) {
        if (program_limit_error(st) && best_to > from) break;  // the previous prefix is this part
        return st;                                             // a single conjunct that does not fit, or a real error
      }
      if (dt != DFX_BOOLEAN) return Status::Err(DFX_EXECUTION_ERROR, "Filter expression did not evaluate to boolean");
      p.builder->build_fast(p.operand, nullptr, 0, nullptr, 0, &p.fast);
      best = std::move(p);
      best_to = to;
    }
    parts.push_back(std::move(best));
    from = best_to;
  }
  builder_ = std::move(parts[0].builder);
  pred_operand_ = parts[0].operand;
  fast_ = parts[0].fast;
  for (size_t i = 1; i < parts.size(); ++i) more_.push_back(std::move(parts[i]));
  return Status::OK();
}

void FilterRelation::explain(std::string* out, int depth) const {
  if (!deferred_.ok()) {
    explain_line(out, depth, "Filter: error deferred to next(): " + deferred_.msg);
  } else {
    const uint8_t none[kMaxAggs] = {0};
    const DevProgram& P = builder_->program();
    const char* shape = sig_matches<SigPred2F64>(P, fast_, 0, 0, none, none) ? "static shape Pred2F64"
                        : fast_.valid                                       ? "column-op-literal conjunction (FastPolicy; interpreter when a batch has nulls)"
                                                                            : "SSA interpreter";
    int n = 0;
    for (size_t i = 0; i < schema_.fields.size() || i < out_needed_.size(); ++i) n += (out_needed_.empty() || (i < out_needed_.size() && out_needed_[i])) ? 1 : 0;
    const bool single = more_.empty() && opt_.get().filter_single_pass;
    explain_line(out, depth, std::string("Filter: ") + (single ? "single pass (predicate, bitmap, look-back over the tiles' kept counts and compaction of the predicate's own columns in one kernel), "
                                                                 : "mask + scan + compaction (two passes over the predicate's columns), ") +
                                 explain_program(P) + ", " + shape +
                                 (out_needed_.empty() ? std::string(", every column compacted") : strfmt(", %d columns compacted", n)) +
                                 (more_.empty() ? std::string() : strfmt(", conjunction evaluated by %zu fused programs (masks ANDed)", more_.size() + 1)));
  }
  if (input_) input_->explain(out, depth + 1);
}

// the consumer reads only `needed` of the filter's output columns: the input must still deliver the predicate's
// columns, and only the needed ones are compacted
void FilterRelation::require_columns(const std::vector<char>& needed) {
  out_needed_ = needed;
  std::vector<char> in_needed = needed;
  in_needed.resize(input_->schema().fields.size(), 1);
  for (int ci : builder_->columns())
    if (ci >= 0 && ci < (int)in_needed.size()) in_needed[ci] = 1;
  for (const Part& p : more_)
    for (int ci : p.builder->columns())
      if (ci >= 0 && ci < (int)in_needed.size()) in_needed[ci] = 1;
  input_->require_columns(in_needed);
}

static Status alloc_zeroed_ctrl(std::shared_ptr<void>* ctrl) {
  Status st;
  *ctrl = device_alloc(sizeof(uint32_t) * CTRL_WORDS, &st);
  if (!*ctrl) return st;
  DFX_HIP(hipMemsetAsync(ctrl->get(), 0, sizeof(uint32_t) * CTRL_WORDS, ctx().stream));
  return Status::OK();
}

Status FilterRelation::next(DeviceBatch* out, bool* has) {
  *has = false;
  if (!source_told_) {  // this operator's own option set decides how a host source below moves its batches
    source_told_ = true;
    if (!opt_.overrides.empty()) input_->host_stream_options(host_stream_options_of(opt_.get()));
  }
  DeviceBatch in;
  bool got = false;
  DFX_RETURN_IF_ERROR(input_->next(&in, &got));
  if (!got) return Status::OK();
  if (!deferred_.ok()) return deferred_;
  DFX_RETURN_IF_ERROR(ensure_init());
  hipStream_t s = ctx().stream;
  const int64_t n = in.num_rows;
  out->columns.clear();
  out->columns.resize(in.columns.size());
  if (n == 0) {  // zero-row batches are still emitted (filter.rs:55-62)
    for (size_t c = 0; c < in.columns.size(); ++c)  // fn filter matches on the type before it looks at a row
      if (in.columns[c].dtype == DFX_BOOLEAN) return Status::Err(DFX_EXECUTION_ERROR, "filter not supported for Boolean");
    for (size_t c = 0; c < in.columns.size(); ++c) {
      out->columns[c].dtype = in.columns[c].dtype;
      out->columns[c].length = 0;
    }
    out->num_rows = 0;
    *has = true;
    return Status::OK();
  }
  DevProgram prog;
  DevColumns cols;
  DFX_RETURN_IF_ERROR(builder_->bind(in, &prog, &cols));
  if (!ctrl_) DFX_RETURN_IF_ERROR(alloc_zeroed_ctrl(&ctrl_));
  const int64_t n_words = (n + 63) / 64;
  const int64_t n_tiles = (n + kTileRows - 1) / kTileRows;
  Status st;
  auto mask = device_alloc(sizeof(uint64_t) * (size_t)n_words, &st);
  if (!mask) return st;
  auto counts = device_alloc(sizeof(uint32_t) * (size_t)n_tiles, &st);
  if (!counts) return st;
  auto offsets = device_alloc(sizeof(uint64_t) * (size_t)(n_tiles + 1), &st);
  if (!offsets) return st;
  auto tmp = device_alloc(sizeof(uint64_t) * (size_t)(n_tiles / 4096 + 4), &st);
  if (!tmp) return st;
  double in_bytes = (double)n / 8.0;
  for (int ci : builder_->columns()) in_bytes += (double)n * (in.columns[ci].dtype == DFX_BOOLEAN ? 0.125 : dtype_width(in.columns[ci].dtype));
  DevFastPlan fp = fast_;
  if (!opt_.get().fast) fp.valid = 0;
  uint64_t kept = 0;
  uint32_t errbits = 0;
  bool single_pass_done = false;
  // columns the fused kernel compacts itself (index into in.columns -> its output buffer)
  std::vector<std::shared_ptr<void>> fused_vals(in.columns.size());
  bool any_boolean = false;
  for (size_t c = 0; c < in.columns.size(); ++c) any_boolean = any_boolean || in.columns[c].dtype == DFX_BOOLEAN;
  if (more_.empty() && opt_.get().filter_single_pass) {
    // SINGLE PASS: predicate, bitmap, tile offsets (decoupled look-back) and the compaction of up to kFusedOutCols of
    // the predicate's own columns in one kernel -- such a column is read from HBM once (filter.rs:46-110)
    DevFusedOut O;
    memset(&O, 0, sizeof(O));
    double out_bytes = 0;
    // Output buffers sized from the selectivity this stream has shown so far (the first batch: every row): a 2^27-row batch of
    // two Float64 predicate columns pinned 2 GB of HBM however few rows it kept.  A batch that keeps more than its buffers hold
    // has those columns compacted again from the bitmap below (k_compact): a second read of the column, paid only then.
    const uint64_t fused_cap = sel_seen_ < 0.0 ? (uint64_t)n
                                               : std::min<uint64_t>((uint64_t)n, (uint64_t)((double)n * std::min(1.0, 1.5 * sel_seen_ + 0.02)) + 4096);
    O.cap_rows = fused_cap;
    O.dense = opt_.get().filter_dense < 0 ? (sel_seen_ > 0.22 ? 1u : 0u) : (uint32_t)(opt_.get().filter_dense != 0);
    const std::vector<int>& pcols = builder_->columns();
    for (size_t slot = 0; slot < pcols.size() && O.n < kFusedOutCols && !any_boolean; ++slot) {
      const int ci = pcols[slot];
      if (ci < 0 || ci >= (int)in.columns.size()) continue;
      const DeviceColumn& ic = in.columns[ci];
      if (ic.absent || ic.dtype == DFX_UTF8 || ic.dtype == DFX_BOOLEAN) continue;
      if ((size_t)ci < out_needed_.size() && !out_needed_[ci]) continue;  // projection push-down: nobody reads it
      if (fused_vals[ci]) continue;
      const int w = dtype_width(ic.dtype);
      auto vals = device_alloc((size_t)std::max<uint64_t>(fused_cap, 1) * w, &st);
      if (!vals) return st;
      fused_vals[ci] = vals;
      O.slot[O.n] = (uint8_t)slot;
      O.dtype[O.n] = (uint8_t)ic.dtype;
      O.out[O.n] = vals.get();
      ++O.n;
      out_bytes += (double)n * w;  // (upper bound; the profiler's byte count is corrected by the selectivity in bench.py)
    }
    (void)out_bytes;
    const size_t sync_words = filter_fused_sync_words(n);
    auto sync = device_alloc(sizeof(uint64_t) * sync_words, &st);
    if (!sync) return st;
    DFX_HIP(hipMemsetAsync(sync.get(), 0, sizeof(uint64_t) * sync_words, s));
    DFX_HIP(launch_filter_fused(prog, fp, cols, pred_operand_, n, (uint64_t*)mask.get(), (uint64_t*)offsets.get(), (uint64_t*)sync.get(), O,
                                (uint32_t*)ctrl_.get(), in_bytes, s));
    // the kernel leaves the kept count next to the error word of the control block: one 64-byte copy into pinned memory
    // and one synchronisation per batch (two pageable 8-byte copies cost ~30 us of a 2^27-row batch's 340)
    if (!ctrl_host_) {
      ctrl_host_ = pinned_alloc(sizeof(uint32_t) * CTRL_WORDS, &st);
      if (!ctrl_host_) return st;
    }
    // (by a kernel writing the pinned buffer, not by the copy engine: an SDMA device-to-host copy stalls for 6 - 150 ms once in a few
    // hundred calls on these boxes -- tools/d2h_probe.py -- and this one runs once per batch: round 5's bench lines showed one or the other
    // of the dense-filter legs a third slower, never the same one)
    DFX_HIP(launch_copy_to_host(ctrl_.get(), ctrl_host_.get(), sizeof(uint32_t) * CTRL_WORDS, s));
    DFX_HIP(hipStreamSynchronize(s));
    const uint32_t* hc = (const uint32_t*)ctrl_host_.get();
    kept = (uint64_t)hc[CTRL_PASSED_LO] | ((uint64_t)hc[CTRL_PASSED_HI] << 32);
    errbits = hc[CTRL_ERROR];
    single_pass_done = true;
    if (errbits & 8u) {  // (also next to another error bit: the fused outputs sit at wrong offsets, and the second pass reports the real error)
      // The look-back gave up waiting (its grid is sized for an EMPTY device: other work on the GPU -- another process, a
      // multi-rank dry run -- can keep a workgroup from becoming resident).  Not an error of the query: this batch takes the
      // two-pass form, which has no inter-workgroup waits.
      DFX_HIP(hipMemsetAsync(ctrl_.get(), 0, sizeof(uint32_t) * CTRL_WORDS, s));
      for (auto& v : fused_vals) v.reset();
      errbits = 0;
      kept = 0;
      single_pass_done = false;
      ++counters().filter_lookback_fallbacks;
    } else if (errbits == 0) {
      sel_seen_ = std::max(sel_seen_, n > 0 ? (double)kept / (double)n : 0.0);
      if (kept > fused_cap) {  // denser than the stream had been: the kernel stored what fitted; these columns are compacted again below
        for (auto& v : fused_vals) v.reset();
        ++counters().filter_output_regrows;
      }
    }
  }
  if (!single_pass_done) {
    DFX_HIP(launch_predicate_mask(prog, fp, cols, pred_operand_, n, (uint64_t*)mask.get(), (uint32_t*)counts.get(),
                                  (uint32_t*)ctrl_.get(), in_bytes, s));
    if (!more_.empty()) {  // the other conjuncts: their masks are ANDed into the first, the tile counts redone
      auto mask2 = device_alloc(sizeof(uint64_t) * (size_t)n_words, &st);
      if (!mask2) return st;
      for (const Part& p : more_) {
        DevProgram prog2;
        DevColumns cols2;
        DFX_RETURN_IF_ERROR(p.builder->bind(in, &prog2, &cols2));
        double bytes2 = (double)n / 8.0;
        for (int ci : p.builder->columns()) bytes2 += (double)n * (in.columns[ci].dtype == DFX_BOOLEAN ? 0.125 : dtype_width(in.columns[ci].dtype));
        DevFastPlan fp2 = p.fast;
        if (!opt_.get().fast) fp2.valid = 0;
        DFX_HIP(launch_predicate_mask(prog2, fp2, cols2, p.operand, n, (uint64_t*)mask2.get(), nullptr, (uint32_t*)ctrl_.get(), bytes2, s));
        DFX_HIP(launch_mask_and_count((uint64_t*)mask.get(), (const uint64_t*)mask2.get(), (uint32_t*)counts.get(), n, s));
      }
    }
    DFX_HIP(launch_scan_u32((const uint32_t*)counts.get(), (uint64_t*)offsets.get(), n_tiles, (uint64_t*)tmp.get(), s));
    DFX_HIP(hipMemcpyAsync(&kept, (uint64_t*)offsets.get() + n_tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    DFX_HIP(hipMemcpyAsync(&errbits, (uint32_t*)ctrl_.get() + CTRL_ERROR, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    DFX_HIP(hipStreamSynchronize(s));
  }
  if (errbits) {
    DFX_HIP(hipMemsetAsync(ctrl_.get(), 0, sizeof(uint32_t) * CTRL_WORDS, s));
    return error_from_ctrl(errbits);
  }
  if (keep_mask_) {
    last_mask_ = mask;
    last_mask_rows_ = n;
  }
  const int64_t m = (int64_t)kept;
  for (size_t c = 0; c < in.columns.size(); ++c)  // fn filter errs for the batch whatever is projected later
    if (in.columns[c].dtype == DFX_BOOLEAN) return Status::Err(DFX_EXECUTION_ERROR, "filter not supported for Boolean");  // filter.rs:105-108
  for (size_t c = 0; c < in.columns.size(); ++c) {  // fn filter per column (filter.rs:55-57)
    const DeviceColumn& ic = in.columns[c];
    DeviceColumn& oc = out->columns[c];
    oc.dtype = ic.dtype;
    oc.length = m;
    oc.null_count = 0;  // value nulls are ignored: the output is all-valid (filter.rs:83-92)
    if (ic.absent || (c < out_needed_.size() && !out_needed_[c])) {  // projection push-down: nobody reads it
      oc.absent = true;
      continue;
    }
    if (ic.dtype == DFX_UTF8) {
      // lengths + starts -> compact both -> scan lengths -> gather bytes
      auto lens = device_alloc(sizeof(int32_t) * (size_t)n, &st);
      if (!lens) return st;
      auto starts = device_alloc(sizeof(int32_t) * (size_t)n, &st);
      if (!starts) return st;
      auto lens_c = device_alloc(sizeof(int32_t) * (size_t)(m + 1), &st);
      if (!lens_c) return st;
      auto starts_c = device_alloc(sizeof(int32_t) * (size_t)(m + 1), &st);
      if (!starts_c) return st;
      auto offs = device_alloc(sizeof(int32_t) * (size_t)(m + 1), &st);
      if (!offs) return st;
      auto tmp2 = device_alloc(sizeof(uint64_t) * (size_t)(m / 4096 + 4), &st);
      if (!tmp2) return st;
      DFX_HIP(launch_utf8_lengths(ic.offsets, n, (int32_t*)lens.get(), (int32_t*)starts.get(), s));
      DFX_HIP(launch_compact(lens.get(), 4, (const uint64_t*)mask.get(), (const uint64_t*)offsets.get(), n, lens_c.get(), 0, s));
      DFX_HIP(launch_compact(starts.get(), 4, (const uint64_t*)mask.get(), (const uint64_t*)offsets.get(), n, starts_c.get(), 0, s));
      DFX_HIP(launch_scan_i32((const int32_t*)lens_c.get(), (int32_t*)offs.get(), m, (uint64_t*)tmp2.get(), s));
      int32_t total = 0;
      DFX_HIP(hipMemcpyAsync(&total, (int32_t*)offs.get() + m, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      DFX_HIP(hipStreamSynchronize(s));
      auto bytes = device_alloc((size_t)total + 8, &st);
      if (!bytes) return st;
      DFX_HIP(launch_utf8_gather(ic.data, (const int32_t*)starts_c.get(), (const int32_t*)offs.get(), m, (uint8_t*)bytes.get(), s));
      oc.offsets = (const int32_t*)offs.get();
      oc.data = (const uint8_t*)bytes.get();
      oc.data_bytes = total;
      oc.owners = {offs, bytes};
    } else if (ic.dtype == DFX_BOOLEAN) {
      return Status::Err(DFX_EXECUTION_ERROR, "filter not supported for Boolean");  // filter.rs:105-108
    } else if (fused_vals[c]) {  // compacted by the kernel that evaluated the predicate
      oc.values = fused_vals[c].get();
      oc.owners = {fused_vals[c]};
    } else {  // deviation D2: every fixed-width type, not just Float64
      const int w = dtype_width(ic.dtype);
      auto vals = device_alloc((size_t)(m > 0 ? m : 1) * w, &st);
      if (!vals) return st;
      DFX_HIP(launch_compact(ic.values, w, (const uint64_t*)mask.get(), (const uint64_t*)offsets.get(), n, vals.get(),
                             (double)n * w + (double)m * w + (double)n / 8.0, s));
      oc.values = vals.get();
      oc.owners = {vals};
    }
  }
  out->num_rows = m;
  *has = true;
  return Status::OK();
}

// =================================================================================================
// ProjectRelation
// =================================================================================================
ProjectRelation::ProjectRelation(std::unique_ptr<Relation> input, std::vector<dfx_runtime_expr> exprs, SchemaInfo schema)
    : input_(std::move(input)), exprs_(std::move(exprs)), schema_(std::move(schema)) {
  passthrough_.assign(exprs_.size(), -1);
  operands_.assign(exprs_.size(), kNoOperand);
  out_dtype_.assign(exprs_.size(), DFX_TYPE_NONE);
  for (size_t i = 0; i < exprs_.size() && deferred_.ok(); ++i) {
    const dfx_runtime_expr& e = exprs_[i];
    if (e.is_aggregate) {  // RuntimeExpr::get_func() panics on an aggregate (expression.rs:60)
      deferred_ = Status::Err(DFX_INTERNAL_ERROR, "explicit panic: get_func() on an aggregate expression");
      break;
    }
    const dfx_expr_node& root = e.nodes[e.root];
    if (root.kind == DFX_EXPR_COLUMN) {  // Arc clone, zero copy (expression.rs:311-315)
      passthrough_[i] = root.column;
      out_dtype_[i] = input_->schema().fields[root.column].dtype;
      continue;
    }
    int dt = DFX_TYPE_NONE;
    Status st = Status::Err(DFX_NOT_IMPLEMENTED, "");
    if (!groups_.empty() && groups_.back().outputs.size() < (size_t)kMaxOut) {
      // try to extend the current fused program; roll back if it would exceed the device limits
      std::unique_ptr<ProgramBuilder> trial(new ProgramBuilder(*groups_.back().builder));
      st = trial->add(e, e.root, &operands_[i], &dt);
      if (st.ok()) groups_.back().builder = std::move(trial);
    }
    if (!st.ok() && st.code == DFX_NOT_IMPLEMENTED) {
      Group g;
      g.builder.reset(new ProgramBuilder(input_->schema()));
      st = g.builder->add(e, e.root, &operands_[i], &dt);
      if (st.ok()) groups_.push_back(std::move(g));
    }
    if (!st.ok()) {
      deferred_ = st;
      break;
    }
    groups_.back().outputs.push_back(i);
    out_dtype_[i] = dt;
  }
  // the output schema is rebuilt from the expressions (projection.rs:52-57): names from
  // RuntimeExpr::get_name, every field nullable.  Deviation D6: actual array types.
  SchemaInfo derived;
  for (size_t i = 0; i < exprs_.size(); ++i) {
    Field f;
    f.name = exprs_[i].name;
    f.dtype = out_dtype_[i];
    f.nullable = true;
    derived.fields.push_back(f);
  }
  if (schema_.fields.size() != derived.fields.size()) {
    schema_ = derived;
  } else {
    for (size_t i = 0; i < derived.fields.size(); ++i) {
      schema_.fields[i].dtype = derived.fields[i].dtype;
      schema_.fields[i].nullable = true;
      if (schema_.fields[i].name.empty()) schema_.fields[i].name = derived.fields[i].name;
    }
  }
  if (deferred_.ok()) {  // projection push-down: the input only has to produce what the expressions read
    std::vector<char> needed(input_->schema().fields.size(), 0);
    for (int pcol : passthrough_)
      if (pcol >= 0 && pcol < (int)needed.size()) needed[pcol] = 1;
    for (const Group& g : groups_)
      for (int ci : g.builder->columns())
        if (ci >= 0 && ci < (int)needed.size()) needed[ci] = 1;
    input_->require_columns(needed);
  }
}

void ProjectRelation::explain(std::string* out, int depth) const {
  if (!deferred_.ok()) {
    explain_line(out, depth, "Project: error deferred to next(): " + deferred_.msg);
  } else {
    int pass = 0;
    for (int p : passthrough_) pass += p >= 0 ? 1 : 0;
    std::string text = strfmt("Project: %d outputs, %d zero-copy columns, %d fused programs", (int)exprs_.size(), pass, (int)groups_.size());
    for (const Group& g : groups_) text += strfmt(" [%d outputs, %s]", (int)g.outputs.size(), explain_program(g.builder->program()).c_str());
    explain_line(out, depth, text);
  }
  if (input_) input_->explain(out, depth + 1);
}

Status ProjectRelation::next(DeviceBatch* out, bool* has) {
  *has = false;
  DeviceBatch in;
  bool got = false;
  DFX_RETURN_IF_ERROR(input_->next(&in, &got));
  if (!got) return Status::OK();
  if (!deferred_.ok()) return deferred_;
  DFX_RETURN_IF_ERROR(ensure_init());
  hipStream_t s = ctx().stream;
  const int64_t n = in.num_rows;
  out->num_rows = n;
  out->columns.clear();
  out->columns.resize(exprs_.size());
  bool any_computed = false;
  for (size_t i = 0; i < exprs_.size(); ++i) {
    if (passthrough_[i] >= 0) out->columns[i] = in.columns[passthrough_[i]];
    else any_computed = true;
  }
  if (!any_computed || n == 0) {
    for (size_t i = 0; i < exprs_.size(); ++i) {
      if (passthrough_[i] >= 0) continue;
      out->columns[i].dtype = out_dtype_[i];
      out->columns[i].length = 0;
    }
    *has = true;
    return Status::OK();
  }
  if (!ctrl_) DFX_RETURN_IF_ERROR(alloc_zeroed_ctrl(&ctrl_));
  const int64_t n_words = (n + 63) / 64;
  Status st;
  for (const Group& g : groups_) {
    DevProgram prog;
    DevColumns cols;
    DFX_RETURN_IF_ERROR(g.builder->bind(in, &prog, &cols));
    double in_bytes = 0;
    for (int ci : g.builder->columns()) in_bytes += (double)n * (in.columns[ci].dtype == DFX_BOOLEAN ? 0.125 : dtype_width(in.columns[ci].dtype));
    DevProjectPlan plan;
    memset(&plan, 0, sizeof(plan));
    double out_bytes = 0;
    plan.n_out = (int32_t)g.outputs.size();
    for (size_t k = 0; k < g.outputs.size(); ++k) {
      const size_t i = g.outputs[k];
      DeviceColumn& oc = out->columns[i];
      oc.dtype = out_dtype_[i];
      oc.length = n;
      const size_t vbytes = oc.dtype == DFX_BOOLEAN ? sizeof(uint64_t) * (size_t)n_words : (size_t)n * dtype_width(oc.dtype);
      auto vals = device_alloc(vbytes, &st);
      if (!vals) return st;
      oc.values = vals.get();
      oc.owners.push_back(vals);
      plan.out[k] = operands_[i];
      plan.out_dtype[k] = (uint8_t)oc.dtype;
      plan.out_values[k] = vals.get();
      out_bytes += (double)vbytes;
      if (prog.has_nulls) {  // null in => null out (arrow 0.12 math_op / and / or)
        auto vb = device_alloc(sizeof(uint64_t) * (size_t)n_words, &st);
        if (!vb) return st;
        oc.validity = (const uint8_t*)vb.get();
        oc.null_count = -1;
        oc.owners.push_back(vb);
        plan.out_validity[k] = (uint64_t*)vb.get();
        out_bytes += (double)n / 8.0;
      }
    }
    DFX_HIP(launch_project(prog, cols, plan, n, (uint32_t*)ctrl_.get(), in_bytes + out_bytes, s));
  }
  uint32_t errbits = 0;
  DFX_HIP(hipMemcpyAsync(&errbits, (uint32_t*)ctrl_.get() + CTRL_ERROR, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  DFX_HIP(hipStreamSynchronize(s));
  if (errbits) {
    DFX_HIP(hipMemsetAsync(ctrl_.get(), 0, sizeof(uint32_t) * CTRL_WORDS, s));
    return error_from_ctrl(errbits);
  }
  *has = true;
  return Status::OK();
}

}  // namespace dfx

// =================================================================================================
// C ABI
// =================================================================================================
using namespace dfx;

extern "C" {

int32_t dfx_filter_relation_new(struct ArrowArrayStream* input, const dfx_runtime_expr* expr,
                                const struct ArrowSchema* schema, struct ArrowArrayStream* out, char* err,
                                size_t errlen) {
  return dfx_filter_relation_new_with_options(input, expr, schema, nullptr, 0, out, err, errlen);
}

int32_t dfx_filter_relation_new_with_options(struct ArrowArrayStream* input, const dfx_runtime_expr* expr,
                                             const struct ArrowSchema* schema, const dfx_option* options, int32_t n_options,
                                             struct ArrowArrayStream* out, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    if (!expr || !out || (n_options > 0 && !options)) return to_c(Status::Err(DFX_GENERAL, "null argument"), err, errlen);
    OptionOverrides ov;
    {
      AggOptions probe = agg_options();
      for (int i = 0; i < n_options; ++i) {
        if (!options[i].key || !set_option_in(probe, options[i].key, options[i].value))
          return to_c(Status::Err(DFX_GENERAL, std::string("unknown option ") + (options[i].key ? options[i].key : "(null)")), err, errlen);
        ov.emplace_back(options[i].key, options[i].value);
      }
    }
    std::unique_ptr<Relation> in;
    Status st = adopt_input_stream(input, &in);
    if (!st.ok()) return to_c(st, err, errlen);
    SchemaInfo si;
    st = schema_from_arrow(schema, &si);
    if (!st.ok()) return to_c(st, err, errlen);
    si = schema_names_over(si, in->schema());
    if (expr->is_aggregate)
      return to_c(Status::Err(DFX_INTERNAL_ERROR, "explicit panic: get_func() on an aggregate expression"), err, errlen);
    std::unique_ptr<Relation> rel(new FilterRelation(std::move(in), *expr, si, std::move(ov)));
    export_relation(std::move(rel), out);
    return DFX_OK;
  });
}

// Measurement hook: pull every batch of a library stream and leave it ON THE DEVICE (no host RecordBatch is built, no
// D2H copy) -- what an operator stacked on top would see.  bench.py times FilterRelation's mask + compaction kernels
// with it (BASELINE config 2 as written); rows / batches count what came out.
int32_t dfx_relation_drain_device(struct ArrowArrayStream* stream, int64_t* rows, int64_t* batches, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    Relation* r = peek_exported(stream);
    if (!r) return to_c(Status::Err(DFX_GENERAL, "not a stream of this library"), err, errlen);
    int64_t nr = 0, nb = 0;
    for (;;) {
      DeviceBatch b;
      bool has = false;
      Status st = r->next(&b, &has);
      if (!st.ok()) return to_c(st, err, errlen);
      if (!has) break;
      nr += b.num_rows;
      ++nb;
    }
    hipError_t e = hipStreamSynchronize(ctx().stream);
    if (e != hipSuccess) return to_c(Status::Err(DFX_EXECUTION_ERROR, strfmt("HIP error %s", hipGetErrorString(e))), err, errlen);
    if (rows) *rows = nr;
    if (batches) *batches = nb;
    return DFX_OK;
  });
}

// Test hook: the Arrow bitmap FilterRelation computed for its most recent input batch (the BooleanArray of the reference's
// predicate closure, filter.rs:53).  out == NULL switches the keeping on (call before next()); else (rows + 7) / 8 bytes
// are copied to `out`.
int32_t dfx_filter_debug_mask(struct ArrowArrayStream* stream, uint8_t* out, int64_t out_bytes, int64_t* rows, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    Relation* r = peek_exported(stream);
    if (!r || r->kind() != REL_FILTER) return to_c(Status::Err(DFX_GENERAL, "not a FilterRelation of this library"), err, errlen);
    FilterRelation* f = static_cast<FilterRelation*>(r);
    if (!out) {
      f->keep_mask(true);
      return DFX_OK;
    }
    const int64_t n = f->last_mask_rows();
    if (!f->last_mask() || out_bytes < (n + 7) / 8) return to_c(Status::Err(DFX_GENERAL, "no bitmap kept, or the buffer is too small"), err, errlen);
    hipError_t e = hipMemcpy(out, f->last_mask().get(), (size_t)((n + 7) / 8), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return to_c(Status::Err(DFX_EXECUTION_ERROR, strfmt("HIP error %s", hipGetErrorString(e))), err, errlen);
    if (rows) *rows = n;
    return DFX_OK;
  });
}

int64_t dfx_relation_explain(struct ArrowArrayStream* stream, char* buf, size_t buflen) {
  try {
    Relation* r = peek_exported(stream);
    if (!r) return -1;
    std::string text;
    r->explain(&text, 0);
    if (buf && buflen) snprintf(buf, buflen, "%s", text.c_str());
    return (int64_t)text.size();
  } catch (...) {
    return -1;
  }
}

int32_t dfx_project_relation_new(struct ArrowArrayStream* input, const dfx_runtime_expr* const* exprs,
                                 int32_t n_exprs, const struct ArrowSchema* schema, struct ArrowArrayStream* out,
                                 char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    if (!out || (n_exprs > 0 && !exprs)) return to_c(Status::Err(DFX_GENERAL, "null argument"), err, errlen);
    std::unique_ptr<Relation> in;
    Status st = adopt_input_stream(input, &in);
    if (!st.ok()) return to_c(st, err, errlen);
    SchemaInfo si;
    st = schema_from_arrow(schema, &si);
    if (!st.ok()) return to_c(st, err, errlen);
    if (n_exprs < 1)  // RecordBatch::new asserts at least one column
      return to_c(Status::Err(DFX_INTERNAL_ERROR, "assertion failed: record batch needs at least one column"), err, errlen);
    std::vector<dfx_runtime_expr> ev;
    for (int i = 0; i < n_exprs; ++i) ev.push_back(*exprs[i]);
    std::unique_ptr<Relation> rel(new ProjectRelation(std::move(in), std::move(ev), si));
    export_relation(std::move(rel), out);
    return DFX_OK;
  });
}

}  // extern "C"
