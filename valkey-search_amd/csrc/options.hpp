// options.hpp -- run-time options of an index: vk_index_set_option / vk_index_get_option.
//
// The reference's knobs are module configs changed with `CONFIG SET search.<name>` (src/valkey_search_options.cc:
// hnsw-block-size :74-81, hnsw-allow-replace-deleted / hnsw-validation-enable :150-162, max-query-queue-depth :231-234,
// prefiltering-threshold-ratio :363-390, ...).  The adaptor forwards the ones that reach this library with
// vk_index_set_option(ix, name, value); everything the kernels' host side used to read from the environment is an option
// of the same kind.  An option is a named u64 with a range; reads on the search path are relaxed atomic loads -- no
// getenv(), no lock.
//
// Environment variables remain as DIAGNOSTIC DEFAULTS only: the `env` name of an option, when set, replaces its default
// ONCE, when the index is created (never per call).  Experiment switches whose answers are invalid (ablations, cycle
// counters, the fat-wave kernel) are not options at all: they exist only in the -DVK_EXPERIMENTS build of the library.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <string>

#include "row_store.hpp"   // Status

namespace vk {

enum OptId : uint32_t {
  // ---- serving: the analogues of the reference's module configs -------------------------------------------------------
  kOptCoalesceMaxBatch,     // queries per device batch the dispatcher forms from single-query calls (0 / 1 = off)
  kOptCoalesceMaxWaitUs,    // longest time a queued query waits for company
  kOptMaxQueryQueueDepth,   // max-query-queue-depth (valkey_search_options.cc:231-234): submissions beyond it are rejected
  kOptBatchesInFlight,      // device batches the dispatcher keeps in flight per index (collect N+1 while N runs)
  kOptCompleterThreads,     // threads that hand the answers of a finished batch to the callers' callbacks (0 = the runner does)
  kOptHandoutChunk,         // ... members per piece of that work
  kOptShardEfPct,           // sharded HNSW: per-shard ef as a percentage of the query's ef (vk_index_params.shard_ef_pct)
  kOptShardGather,          // sharded index: 0 = per-shard top-k gathered by peer copies, 1 = by an RCCL all-gather
  kOptFilterCacheEntries,   // vk_index_filter_cache_put: most filters kept (0 = the cache is off) ...
  kOptFilterCacheBytes,     // ... and most device memory they may hold
  kOptKernelTiming,         // 1 = HIP event pairs around the dominant kernel's launches (vk_index_stats.filter_kernel_ns)
  // ---- FLAT: kernel selection and the candidate filter (K4h) ------------------------------------------------------------
  kOptFlatFilter, kOptFilterMinQueries, kOptFilterMinRows, kOptFilterPrepassRows, kOptFilterCap, kOptFilterSpillChunks,
  kOptFilterBDma, kOptFilterRowDma, kOptFilterBf16Mfma, kOptFlatForceScan, kOptFlatFusedRerank, kOptFilterSecondBound, kOptFilterTwoPass, kOptFilterTwoPassMinTiles, kOptFilterEarlyPermille,
  kOptGemmLockstep, kOptGemmPrepassRows, kOptGemmContig, kOptScanMinNrp, kOptUploadParallel,
  // ---- HNSW ----------------------------------------------------------------------------------------------------------------
  kOptHnswStageAdds, kOptHnswStageMax,
  kOptHnswOptimisticTombstones, kOptHnswDeviceBuild, kOptHnswBuildBatch, kOptHnswBuildMinGraph, kOptHnswBuildMinBatch, kOptHnswBuildFrac,
  kOptHnswBuildVerbose, kOptHnswFailpointBatch, kOptHnswPoolFloor, kOptHnswGpoolCap, kOptHnswVisitedHash, kOptHnswHashPerEf, kOptHnswHashLog2, kOptHnswVisitedMode, kOptHnswLdsWork, kOptHnswLdsWorkBig,
  kOptHnswPoolBytes, kOptHnswVisitedBytes, kOptHnswRedoBytes,
  // ---- sharded index ---------------------------------------------------------------------------------------------------
  kOptShardThreads, kOptShardAllowStaged,
  kOptCount
};

struct OptDesc {
  const char *name;   // vk_index_set_option name
  const char *env;    // diagnostic default, read once at index creation (nullptr = none)
  uint64_t dflt, lo, hi;
};

inline const OptDesc &opt_desc(uint32_t id) {
  static const uint64_t kMax = ~0ull;
  static const OptDesc t[kOptCount] = {
      {"coalesce-max-batch", nullptr, 0, 0, 16384},
      {"coalesce-max-wait-us", nullptr, 200, 0, 10000000},
      {"max-query-queue-depth", nullptr, 100000, 0, 0x7FFFFFFF},
      {"batches-in-flight", "VK_BATCHES_IN_FLIGHT", 2, 1, 8},
      {"completer-threads", "VK_COMPLETER_THREADS", 6, 0, 16},
      {"handout-chunk", "VK_HANDOUT_CHUNK", 64, 16, 16384},
      {"shard-ef-pct", nullptr, 100, 1, 1000},
      {"shard-gather", "VK_SHARD_GATHER", 0, 0, 1},
      {"filter-cache-entries", nullptr, 256, 0, 1u << 20},
      {"filter-cache-bytes", nullptr, (uint64_t)1 << 30, 0, kMax},
      {"kernel-timing", "VK_KERNEL_TIMING", 0, 0, 1},
      {"flat-filter", "VK_FLAT_FILTER", 1, 0, 1},
      {"filter-min-queries", "VK_FILTER_MIN_QUERIES", 5, 1, kMax},
      {"filter-min-rows", "VK_FILTER_MIN_ROWS", 262144, 0, kMax},
      {"filter-prepass-rows", "VK_FILTER_PREPASS", 262144, 1024, kMax},
      {"filter-cap", "VK_FILTER_CAP", 8192, 64, 1u << 24},
      {"filter-spill-chunks", "VK_FILTER_SPILL_CHUNKS", 1024, 0, 1u << 20},
      {"filter-bdma", "VK_FILTER_BDMA", 1, 0, 1},
      {"filter-row-dma", "VK_FILTER_DMA", 1, 0, 1},
      {"filter-bf16-mfma", "VK_FILTER_BF16_MFMA", 1, 0, 1},
      {"flat-force-scan", "VK_FLAT_FORCE_SCAN", 0, 0, 1},
      {"flat-fused-rerank", "VK_FLAT_FUSED_RERANK", 1, 0, 1},
      {"filter-second-bound", "VK_FILTER_SECOND_BOUND", 1, 0, 1},
      {"filter-two-pass", "VK_FILTER_TWO_PASS", 1, 0, 1},                         // the rows in an early and a main launch, the main pass's bound from the early one's survivors
      {"filter-two-pass-min-tiles", "VK_FILTER_TWO_PASS_MIN_TILES", 16, 2, 1u << 20},   // ... for indexes of at least this many 128-row tiles per block (CU)
      {"filter-early-permille", "VK_FILTER_EARLY_PERMILLE", 0, 0, 250},          // ... the early pass's share of the rows (0 = sqrt(sample / rows))
      {"gemm-lockstep", "VK_GEMM_LOCKSTEP", 1, 0, 64},
      {"gemm-prepass-rows", "VK_GEMM_PREPASS", 16384, 0, kMax},
      {"gemm-contig", "VK_GEMM_CONTIG", 1, 0, 1},
      {"scan-min-nrp", "VK_SCAN_MIN_NRP", 8, 1, 4096},
      {"upload-parallel", "VK_UPLOAD_PARALLEL", 1, 0, 1},
      {"hnsw-stage-adds", "VK_HNSW_STAGE_ADDS", 1, 0, 1},                         // single vk_index_add calls of new labels are staged and linked in bulk
      {"hnsw-stage-max", "VK_HNSW_STAGE_MAX", 262144, 1, 1u << 26},               // ... at most this many rows wait (the writer that fills it links them)
      {"hnsw-optimistic-tombstones", "VK_HNSW_OPTIMISTIC_TOMBSTONES", 1, 0, 1},     // a few tombstones (<= 1/16 of the nodes), no filter: the LDS-frontier launch + re-run of what outgrows it
      {"hnsw-device-build", "VK_HNSW_DEVICE_BUILD", 1, 0, 1},
      {"hnsw-build-batch", "VK_HNSW_BUILD_BATCH", 8192, 1, 1u << 20},
      {"hnsw-build-min-graph", "VK_HNSW_BUILD_MIN_GRAPH", 16384, 1, kMax},
      {"hnsw-build-min-batch", "VK_HNSW_BUILD_MIN_BATCH", 64, 1, kMax},
      {"hnsw-build-frac", "VK_HNSW_BUILD_FRAC", 32, 1, 1u << 20},
      {"hnsw-build-verbose", "VK_HNSW_BUILD_VERBOSE", 0, 0, 1},
      {"hnsw-failpoint-batch", "VK_HNSW_FAILPOINT_BATCH", 0, 0, kMax},            // TEST failpoint: the n-th device batch of a bulk fails behind its registration (0 = off)
      {"hnsw-pool-floor", "VK_HNSW_POOL_FLOOR", 512, 64, 1u << 20},
      {"hnsw-gpool-cap", "VK_HNSW_GPOOL_CAP", 65536, 128, 65536},
      {"hnsw-visited-hash", "VK_HNSW_VISITED_HASH", 1, 0, 2},
      {"hnsw-hash-per-ef", "VK_HNSW_HASH_PER_EF", 64, 1, 1u << 16},
      {"hnsw-hash-log2", "VK_HNSW_HASH_LOG2", 0, 0, 26},
      {"hnsw-visited-mode", "VK_HNSW_VISITED_MODE", 3, 0, 4},
      {"hnsw-lds-visited-work", "VK_HNSW_LDS_WORK", 17408, 0, 1u << 24},         // mode 3: ef x maxM0 up to here (ef = 544 at M = 16) takes the 12 KB LDS set ...
      {"hnsw-lds-visited-work-big", "VK_HNSW_LDS_WORK_BIG", 0, 0, 1u << 24},      // ... the 32 KB one (off: the small set with spill beats it at every ef measured)
      {"hnsw-pool-bytes", "VK_HNSW_POOL_BYTES", (uint64_t)4 << 30, 1u << 20, kMax},
      {"hnsw-visited-bytes", "VK_HNSW_VISITED_BYTES", (uint64_t)4 << 30, 1u << 20, kMax},
      {"hnsw-redo-bytes", "VK_HNSW_REDO_BYTES", (uint64_t)2 << 30, 1u << 20, kMax},
      {"shard-threads", "VK_SHARD_THREADS", 1, 0, 1},
      {"shard-allow-staged", "VK_SHARD_ALLOW_STAGED", 0, 0, 1},
  };
  return t[id];
}

class Options {
 public:
  Options() {
    for (uint32_t i = 0; i < kOptCount; ++i) {
      const OptDesc &d = opt_desc(i);
      uint64_t v = d.dflt;
      // (index creation is not a search path: the environment is looked at here and nowhere else)
      if (d.env) {
        const char *e = getenv(d.env);
        if (e && *e) {
          const uint64_t x = strtoull(e, nullptr, 10);
          v = x < d.lo ? d.lo : (x > d.hi ? d.hi : x);
        }
      }
      v_[i].store(v, std::memory_order_relaxed);
    }
  }
  Options(const Options &o) { copy_from(o); }
  Options &operator=(const Options &o) { copy_from(o); return *this; }
  uint64_t get(OptId id) const { return v_[id].load(std::memory_order_relaxed); }
  void set(OptId id, uint64_t v) { v_[id].store(v, std::memory_order_relaxed); }
  static int find(const char *name) {
    if (!name) return -1;
    for (uint32_t i = 0; i < kOptCount; ++i)
      if (strcmp(opt_desc(i).name, name) == 0) return (int)i;
    return -1;
  }
  Status set(const char *name, uint64_t v) {
    const int id = find(name);
    if (id < 0) return Status::Err(1, std::string("unknown option: ") + (name ? name : "(null)"));
    const OptDesc &d = opt_desc((uint32_t)id);
    if (v < d.lo || v > d.hi)
      return Status::Err(1, std::string("option ") + name + ": value out of range [" + std::to_string(d.lo) + ", " + std::to_string(d.hi) + "]");
    set((OptId)id, v);
    return Status::Ok();
  }
  Status get(const char *name, uint64_t *out) const {
    const int id = find(name);
    if (id < 0) return Status::Err(1, std::string("unknown option: ") + (name ? name : "(null)"));
    *out = get((OptId)id);
    return Status::Ok();
  }

 private:
  void copy_from(const Options &o) {
    for (uint32_t i = 0; i < kOptCount; ++i) v_[i].store(o.v_[i].load(std::memory_order_relaxed), std::memory_order_relaxed);
  }
  std::atomic<uint64_t> v_[kOptCount];
};

// a member that reads like the plain field it replaces (`filter_cap_`, `visited_hash_`, ...) and is an option underneath
struct OptRef {
  const Options *o;
  OptId id;
  operator uint64_t() const { return o->get(id); }
};

}  // namespace vk
