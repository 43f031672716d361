// row_store.cc -- see row_store.hpp
#include "row_store.hpp"

#include <string.h>

#include <algorithm>

namespace vk {

namespace {
// f32 -> bf16, round to nearest even (NaN stays NaN): the ingest rule of a bf16 index
inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
}  // namespace

RowStore::RowStore(int device, uint32_t dim, bool bf16)
    : device_(device), dim_(dim), stride_f_(padded_dim(dim)), elem_(bf16 ? 2u : 4u) {
  (void)hipSetDevice(device_);
  (void)hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking);
  size_t rb = row_bytes();
  chunk_bytes_ = std::max<size_t>(rb * 64, (size_t)8 << 20);
  chunk_bytes_ -= chunk_bytes_ % rb;
}

RowStore::~RowStore() {
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  for (char *c : chunks_) (void)hipHostFree(c);
  if (d_rows_) (void)hipFree(d_rows_);
  if (d_labels_) (void)hipFree(d_labels_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

uint64_t RowStore::host_bytes() const {
  return chunks_.size() * chunk_bytes_ + h_labels_.capacity() * 8 + ops_.capacity() * sizeof(Op);
}

Status RowStore::staging_alloc(size_t bytes, size_t *off, char **ptr) {
  if (chunks_.empty() || chunk_used_ + bytes > chunk_bytes_) {
    size_t next = chunks_.empty() ? 0 : cur_chunk_ + 1;
    if (next >= chunks_.size()) {
      (void)hipSetDevice(device_);
      char *p = nullptr;
      VK_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&p), chunk_bytes_, hipHostMallocDefault));
      chunks_.push_back(p);
    }
    cur_chunk_ = next;
    chunk_used_ = 0;
  }
  *off = cur_chunk_ * chunk_bytes_ + chunk_used_;
  *ptr = chunks_[cur_chunk_] + chunk_used_;
  chunk_used_ += bytes;
  staged_bytes_ += bytes;
  return Status::Ok();
}

Status RowStore::stage_write(uint32_t slot, const float *row, uint64_t label) {
  size_t off;
  char *p;
  VK_TRY(staging_alloc(row_bytes(), &off, &p));
  if (elem_ == 4) {
    memcpy(p, row, (size_t)dim_ * 4);
  } else {
    uint16_t *h = reinterpret_cast<uint16_t *>(p);
    for (uint32_t i = 0; i < dim_; ++i) h[i] = f32_to_bf16_rne(row[i]);
  }
  if (stride_f_ > dim_) memset(p + (size_t)dim_ * elem_, 0, (size_t)(stride_f_ - dim_) * elem_);
  ops_.push_back(Op{0, slot, 0, off});
  stage_label(slot, label);
  return Status::Ok();
}

Status RowStore::bulk_write(uint32_t first, const float *rows, uint64_t n, const uint64_t *labels) {
  if (elem_ != 4) return Status::Err(3, "bulk_write: f32 storage only");
  if (n == 0) return Status::Ok();
  VK_TRY(flush());
  (void)hipSetDevice(device_);
  VK_TRY(reserve((uint64_t)first + n));
  const size_t rb = row_bytes(), src_pitch = (size_t)dim_ * 4;
  char *dst = reinterpret_cast<char *>(d_rows_) + (size_t)first * rb;
  if (rb != src_pitch) VK_HIP_TRY(hipMemsetAsync(dst, 0, n * rb, stream_));      // the zero padding of every row
  // the source is the caller's pageable memory: pieces of 64 MiB keep the runtime's staging pipelined
  const uint64_t piece = std::max<uint64_t>(1, ((uint64_t)64 << 20) / src_pitch);
  for (uint64_t at = 0; at < n; at += piece) {
    const uint64_t m = std::min<uint64_t>(piece, n - at);
    VK_HIP_TRY(hipMemcpy2DAsync(dst + at * rb, rb, reinterpret_cast<const char *>(rows) + at * src_pitch, src_pitch, src_pitch, m,
                                hipMemcpyHostToDevice, stream_));
  }
  note_written(first, (uint64_t)first + n);
  if (h_labels_.size() < (size_t)first + n) h_labels_.resize((size_t)first + n, ~0ull);
  memcpy(h_labels_.data() + first, labels, n * 8);
  VK_HIP_TRY(hipMemcpyAsync(d_labels_ + first, labels, n * 8, hipMemcpyHostToDevice, stream_));
  VK_HIP_TRY(hipStreamSynchronize(stream_));
  return Status::Ok();
}

void RowStore::stage_move(uint32_t dst, uint32_t src, uint64_t label) {
  ops_.push_back(Op{1, dst, src, 0});
  stage_label(dst, label);
}

void RowStore::stage_label(uint32_t slot, uint64_t label) {
  if (h_labels_.size() <= slot) h_labels_.resize((size_t)slot + 1, ~0ull);
  h_labels_[slot] = label;
  label_dirty_lo_ = std::min<uint64_t>(label_dirty_lo_, slot);
  label_dirty_hi_ = std::max<uint64_t>(label_dirty_hi_, (uint64_t)slot + 1);
}

Status RowStore::reserve(uint64_t rows) {
  if (rows <= alloc_rows_) return Status::Ok();
  (void)hipSetDevice(device_);
  uint64_t want = std::max<uint64_t>(rows, alloc_rows_ + alloc_rows_ / 2);
  want = std::max<uint64_t>(want, 1024);
  void *nr = nullptr;
  uint64_t *nl = nullptr;
  // kRowSlack rows beyond the capacity stay allocated (and zero): the tiled kernels read whole
  // 128-row tiles and mask the rows past the count afterwards, instead of clamping every address
  hipError_t e = hipMalloc(&nr, (want + kRowSlack) * row_bytes());
  if (e != hipSuccess && want > std::max<uint64_t>(rows, 1024)) {   // no room for half as much again: exactly what is asked for, then
    (void)hipGetLastError();
    want = std::max<uint64_t>(rows, 1024);
    e = hipMalloc(&nr, (want + kRowSlack) * row_bytes());
  }
  VK_HIP_TRY(e);
  e = hipMalloc(reinterpret_cast<void **>(&nl), want * 8);
  // (every failure from here on gives the new arrays back: the store keeps its old ones)
  auto step = [&](hipError_t err, const char *what) -> Status {
    if (err == hipSuccess) return Status::Ok();
    (void)hipGetLastError();   // (the thread's sticky copy of it)
    (void)hipStreamSynchronize(stream_);
    (void)hipFree(nr);
    if (nl) (void)hipFree(nl);
    return Status::Err(4, std::string(what) + ": " + hipGetErrorString(err));
  };
  if (e != hipSuccess) { nl = nullptr; return step(e, "hipMalloc labels"); }
  VK_TRY(step(hipMemsetAsync(static_cast<char *>(nr) + want * row_bytes(), 0, kRowSlack * row_bytes(), stream_), "hipMemsetAsync slack"));
  if (alloc_rows_) {
    VK_TRY(step(hipMemcpyAsync(nr, d_rows_, alloc_rows_ * row_bytes(), hipMemcpyDeviceToDevice, stream_), "hipMemcpyAsync rows"));
    VK_TRY(step(hipMemcpyAsync(nl, d_labels_, alloc_rows_ * 8, hipMemcpyDeviceToDevice, stream_), "hipMemcpyAsync labels"));
  }
  VK_TRY(step(hipMemsetAsync(nl + alloc_rows_, 0xFF, (want - alloc_rows_) * 8, stream_), "hipMemsetAsync labels"));
  VK_TRY(step(hipStreamSynchronize(stream_), "hipStreamSynchronize"));
  if (d_rows_) (void)hipFree(d_rows_);
  if (d_labels_) (void)hipFree(d_labels_);
  d_rows_ = nr;
  d_labels_ = nl;
  alloc_rows_ = want;
  return Status::Ok();
}

Status RowStore::flush() {
  if (!dirty()) return Status::Ok();
  (void)hipSetDevice(device_);
  uint64_t need = h_labels_.size();
  for (const Op &op : ops_) need = std::max<uint64_t>(need, (uint64_t)op.slot + 1);
  VK_TRY(reserve(need));
  const size_t rb = row_bytes();
  for (const Op &op : ops_) note_written(op.slot, (uint64_t)op.slot + 1);
  size_t i = 0;
  while (i < ops_.size()) {
    const Op &op = ops_[i];
    if (op.kind == 0) {
      // merge a run of writes that is contiguous both in slots and in staging
      size_t jn = i + 1;
      while (jn < ops_.size() && ops_[jn].kind == 0 && ops_[jn].slot == ops_[jn - 1].slot + 1 &&
             ops_[jn].off == ops_[jn - 1].off + rb && (ops_[jn].off / chunk_bytes_) == (op.off / chunk_bytes_))
        ++jn;
      const char *src = chunks_[op.off / chunk_bytes_] + (op.off % chunk_bytes_);
      VK_HIP_TRY(hipMemcpyAsync(reinterpret_cast<char *>(d_rows_) + (size_t)op.slot * rb, src, (jn - i) * rb,
                                hipMemcpyHostToDevice, stream_));
      i = jn;
    } else {
      VK_HIP_TRY(hipMemcpyAsync(reinterpret_cast<char *>(d_rows_) + (size_t)op.slot * rb,
                                reinterpret_cast<char *>(d_rows_) + (size_t)op.src * rb, rb,
                                hipMemcpyDeviceToDevice, stream_));
      ++i;
    }
  }
  if (label_dirty_lo_ < label_dirty_hi_) {
    uint64_t hi = std::min<uint64_t>(label_dirty_hi_, h_labels_.size());
    if (hi > label_dirty_lo_)
      VK_HIP_TRY(hipMemcpyAsync(d_labels_ + label_dirty_lo_, h_labels_.data() + label_dirty_lo_,
                                (hi - label_dirty_lo_) * 8, hipMemcpyHostToDevice, stream_));
  }
  VK_HIP_TRY(hipStreamSynchronize(stream_));
  ops_.clear();
  staged_bytes_ = 0;
  chunk_used_ = 0;
  cur_chunk_ = 0;
  while (chunks_.size() > 8) {  // keep at most 8 pinned chunks around
    (void)hipHostFree(chunks_.back());
    chunks_.pop_back();
  }
  label_dirty_lo_ = ~0ull;
  label_dirty_hi_ = 0;
  return Status::Ok();
}

Status RowStore::read_row(uint32_t slot, float *out) {
  VK_TRY(flush());
  (void)hipSetDevice(device_);
  if (slot >= alloc_rows_) return Status::Err(3, "slot out of range");
  if (elem_ == 4) {
    VK_HIP_TRY(hipMemcpy(out, reinterpret_cast<char *>(d_rows_) + (size_t)slot * row_bytes(), (size_t)dim_ * 4,
                         hipMemcpyDeviceToHost));
  } else {
    std::vector<uint16_t> h(dim_);
    VK_HIP_TRY(hipMemcpy(h.data(), reinterpret_cast<char *>(d_rows_) + (size_t)slot * row_bytes(), (size_t)dim_ * 2,
                         hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < dim_; ++i) {
      uint32_t u = (uint32_t)h[i] << 16;
      memcpy(out + i, &u, 4);
    }
  }
  return Status::Ok();
}

}  // namespace vk
