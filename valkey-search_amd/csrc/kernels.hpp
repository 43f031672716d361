// kernels.hpp -- host-visible launch interface of the HIP kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vk {

// K3: FLAT scan over rows [row_begin,row_end) for nq queries.
struct FlatScanArgs {
  const float *rows;          // [cap][row_stride_f], zero padded to a multiple of 16 floats
  const uint64_t *labels;     // [cap]
  const float *queries;       // [nq][q_stride_f], padded like rows
  const uint64_t *allow_bits; // optional bitmap by label
  uint64_t allow_nbits;
  float *part_dist;           // [nq][nrp*4][k] per-wave partial top-k
  uint64_t *part_label;
  uint32_t row_stride_f, q_stride_f;
  uint32_t chunks;            // row_stride_f / 16
  uint32_t row_begin, row_end;
  uint32_t nq, k;
  uint32_t nrp;               // row partitions (blocks along the rows), multiple of 8
  uint32_t nqg;               // query groups = ceil(nq / kQB); grid = nrp * nqg blocks
};

struct MergeArgs {
  const float *in_dist;       // entry (part, q, i) at part*part_stride + q*q_stride + i
  const uint64_t *in_label;
  uint64_t part_stride, q_stride;
  uint32_t parts, per_part;
  uint32_t k;
  float *out_dist;            // [nq][k] ascending by (dist,label)
  uint64_t *out_label;
  uint32_t *out_n;            // [nq]
};

struct GatherArgs {
  const float *rows;
  const float *query;         // padded
  const uint32_t *idx;        // [n] row slots
  float *out;                 // [n]
  uint32_t row_stride_f, chunks, n;
};

int flat_scan_slots_per_lane(uint64_t k);                 // 0 = k too large for the in-register top-k
int flat_scan_pick_qb(uint64_t nq, uint32_t chunks, int e);
hipError_t launch_flat_scan(const FlatScanArgs &a, bool l2, int qb, int e, hipStream_t s);
hipError_t launch_merge_topk(const MergeArgs &a, int e, uint64_t nq, hipStream_t s);
hipError_t launch_gather_distance(const GatherArgs &a, bool l2, hipStream_t s);

}  // namespace vk
