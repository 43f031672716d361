// row_store.hpp -- HBM-resident row table shared by the FLAT and HNSW device mirrors.
//
// Layout in HBM: rows[slot][stride_f] f32, row-contiguous, each row zero-padded to a
// multiple of 64 floats (256 B) so a quad of lanes can stream it with aligned 16-B loads
// and the padding reproduces SimSIMD's masked tail (device_common.hpp); labels[slot] u64.
// Host mutations are appended to an op log with their row payload in pinned staging
// memory and applied in order by flush() on the store's stream -- the "publish at the
// write->read phase switch" model that replaces hnswlib storing raw host pointers
// (bruteforce.h:81, hnswalg.h:1576-1577).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

namespace vk {

struct Status {
  int code = 0;  // vk_status
  std::string msg;
  bool ok() const { return code == 0; }
  static Status Ok() { return {}; }
  static Status Err(int c, std::string m) { return Status{c, std::move(m)}; }
};

#define VK_HIP_TRY(expr)                                                                      \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      (void)hipGetLastError(); /* the thread's sticky copy: a later launch check must not find it */ \
      return ::vk::Status::Err(4, std::string(#expr) + ": " + hipGetErrorString(_e));         \
    }                                                                                         \
  } while (0)

#define VK_TRY(expr)                       \
  do {                                     \
    ::vk::Status _s = (expr);              \
    if (!_s.ok()) return _s;               \
  } while (0)

// Row stride in floats: a multiple of 64 (256 B).  16 would be enough for the quad-per-row kernels
// (one SimSIMD chunk); 64 = two MFMA stages of two chunks, so the matrix-core kernel needs no tail
// handling.  Padding is zero: fma(0,0,x) == x, distances are unchanged.
inline uint32_t padded_dim(uint32_t dim) { return (dim + 63u) & ~63u; }

class RowStore {
 public:
  static constexpr uint64_t kRowSlack = 256;   // readable rows past the capacity (see reserve()): one tile of the widest kernel
  RowStore(int device, uint32_t dim, bool bf16 = false);
  ~RowStore();
  RowStore(const RowStore &) = delete;
  RowStore &operator=(const RowStore &) = delete;

  uint32_t dim() const { return dim_; }
  uint32_t stride_f() const { return stride_f_; }
  size_t row_bytes() const { return (size_t)stride_f_ * elem_; }   // stride_f_ counts ELEMENTS (f32 or bf16)
  bool bf16() const { return elem_ == 2; }
  int device() const { return device_; }
  hipStream_t stream() const { return stream_; }
  const void *d_rows() const { return d_rows_; }
  void *d_rows_mut() { return d_rows_; }
  const uint64_t *d_labels() const { return d_labels_; }
  uint64_t alloc_rows() const { return alloc_rows_; }
  size_t staged_ops() const { return ops_.size(); }
  size_t staged_bytes() const { return staged_bytes_; }
  bool dirty() const { return !ops_.empty() || label_dirty_lo_ < label_dirty_hi_; }
  uint64_t device_bytes() const { return alloc_rows_ * (row_bytes() + 8); }
  uint64_t host_bytes() const;

  // host-side staging (caller serialises mutations)
  Status stage_write(uint32_t slot, const float *row, uint64_t label);  // row: dim floats
  // n consecutive slots from `first`, straight from the caller's rows into the table (f32 storage only): pending
  // operations are published first, so the order of effects is the order of calls
  Status bulk_write(uint32_t first, const float *rows, uint64_t n, const uint64_t *labels);
  void stage_move(uint32_t dst, uint32_t src, uint64_t label);          // row[dst] = row[src]
  void stage_label(uint32_t slot, uint64_t label);
  // make sure the device arrays can hold `rows` slots (keeps contents)
  Status reserve(uint64_t rows);
  // apply the log; blocks until the device copy is complete
  Status flush();
  // read one row back (dim floats); flushes first if needed
  Status read_row(uint32_t slot, float *out);
  // labels mirror (authoritative on the host)
  std::vector<uint64_t> &host_labels() { return h_labels_; }
  // slots whose ROW bytes were (re)written on the device since the last call -- what per-row statistics kept beside
  // the table have to be brought up to date for (FlatIndex: the candidate filter's norm / range bounds)
  void note_written(uint64_t lo, uint64_t hi) { written_lo_ = std::min(written_lo_, lo); written_hi_ = std::max(written_hi_, hi); }
  bool take_written(uint64_t *lo, uint64_t *hi) {
    if (written_lo_ >= written_hi_) return false;
    *lo = written_lo_;
    *hi = written_hi_;
    written_lo_ = ~0ull;
    written_hi_ = 0;
    return true;
  }

 private:
  struct Op { uint8_t kind; uint32_t slot; uint32_t src; size_t off; };  // kind 0 write, 1 move
  Status staging_alloc(size_t bytes, size_t *off, char **ptr);

  int device_;
  uint32_t dim_, stride_f_, elem_;
  hipStream_t stream_ = nullptr;
  void *d_rows_ = nullptr;
  uint64_t *d_labels_ = nullptr;
  uint64_t alloc_rows_ = 0;
  std::vector<uint64_t> h_labels_;
  uint64_t label_dirty_lo_ = ~0ull, label_dirty_hi_ = 0;
  uint64_t written_lo_ = ~0ull, written_hi_ = 0;
  std::vector<Op> ops_;
  std::vector<char *> chunks_;   // pinned staging chunks
  size_t chunk_bytes_ = 0, chunk_used_ = 0, cur_chunk_ = 0;
  size_t staged_bytes_ = 0;
};

}  // namespace vk
