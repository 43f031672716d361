/*
 * vko.h -- CPU ORACLE for the valkey-search vector-kNN hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke
 * check in __graft_entry__.py and bench.py's `cpu_baseline` leg may link or
 * load it.  The shipped path (valkey-search_amd/) never calls into oracle/.
 *
 * It is a plain-C restatement of the reference's algorithm for the path
 *   src/indexes/vector_{base,flat,hnsw}.cc -> third_party/hnswlib ->
 *   third_party/simsimd (v5.0.1)
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  What pins it:
 *   - distance kernels: bit-compared against the real SimSIMD built from
 *     /root/reference/third_party/simsimd/c/lib.c (oracle/_ref, see
 *     oracle/Makefile) and against the committed fixtures in tests/golden/.
 *   - hnswlib's bruteforce/hnswalg headers need abseil + protobuf-generated
 *     code that this image does not have, so they are UNBUILDABLE here; the
 *     FLAT/HNSW restatement is pinned by the reference's own tests instead
 *     (known-answer score strings, search_test.cc expected key sets,
 *     vector_test.cc recall floor) -- see tests/test_oracle_*.py.
 *     Exact HNSW graph identity against the reference binary: PARITY UNPINNED.
 */
#ifndef VKO_H_
#define VKO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Which SimSIMD kernel's summation order to reproduce
 * (dispatch order: simsimd.h:499-547 -- skylake > haswell > serial on x86). */
typedef enum { VKO_ISA_SERIAL = 0, VKO_ISA_HASWELL = 1, VKO_ISA_SKYLAKE = 2 } vko_isa_t;
/* hnswlib space (vector_base.cc:61-76): COSINE and IP -> InnerProductSpace. */
typedef enum { VKO_SPACE_L2 = 0, VKO_SPACE_IP = 1 } vko_space_t;

/* ---- L1: distance kernels ------------------------------------------------ */
double vko_dot_f32(vko_isa_t isa, const float *a, const float *b, size_t n);
double vko_l2sq_f32(vko_isa_t isa, const float *a, const float *b, size_t n);
/* hnswlib/simsimd.h:16-34 wrappers: IP -> (float)(1.0 - dot); L2 -> (float)l2sq */
float vko_distance(vko_space_t space, vko_isa_t isa, const float *a, const float *b, size_t n);
/* vector_base.cc:112-124 CopyAndNormalizeEmbedding; returns the magnitude */
float vko_normalize(float *dst, const float *src, size_t n);
/* which vectorised clone the running CPU selected (for the bench log) */
const char *vko_cpu_path(void);

/* ---- L2: FLAT (bruteforce.h) ---------------------------------------------- */
typedef struct vko_flat vko_flat;
vko_flat *vko_flat_new(size_t dim, vko_space_t space, vko_isa_t isa, size_t max_elements);
void vko_flat_free(vko_flat *f);
/* 0 ok; 1 = "The number of elements exceeds the specified limit" */
int vko_flat_add(vko_flat *f, const float *row, uint64_t label);
/* like vko_flat_add, but the index keeps the caller's pointer (as the
 * reference does, bruteforce.h:81) instead of copying -- for big baselines */
int vko_flat_add_borrowed(vko_flat *f, const float *row, uint64_t label);
int vko_flat_add_many(vko_flat *f, const float *rows, size_t stride_bytes, const uint64_t *labels, size_t n, int borrowed);
void vko_flat_remove(vko_flat *f, uint64_t label);
void vko_flat_resize(vko_flat *f, size_t new_max);
size_t vko_flat_count(const vko_flat *f);
size_t vko_flat_capacity(const vko_flat *f);
/* distances through an external function (the compiled reference's fstdistfunc_ from oracle/_ref) instead of the restatement */
void vko_flat_set_distfn(vko_flat *f, float (*fn)(const float *, const float *, size_t));
/* searchKnn (bruteforce.h:116-145).  allow_bits: optional bitmap indexed by
 * LABEL (bit set = allowed, labels >= allow_nbits are rejected); cancel_after:
 * isCancelled() returns true from its (cancel_after+1)-th poll on, <0 = never.
 * Output ascending by (dist,label) as VectorBase::CreateReply leaves it
 * (vector_base.cc:258-277).  Returns the number of results. */
size_t vko_flat_search(const vko_flat *f, const float *q, size_t k,
                       const uint64_t *allow_bits, uint64_t allow_nbits, long cancel_after,
                       float *out_dist, uint64_t *out_label);
/* distance of one stored record (vector_flat.cc:256-271); 0 ok, 1 unknown label */
int vko_flat_distance(const vko_flat *f, uint64_t label, const float *q, float *out);

/* ---- pre-filter heap (vector_base.cc:509-530 AddPrefilteredKey) ----------- */
/* exact kNN over an explicit list: strict `<` against the heap top to replace */
size_t vko_prefilter_topk(vko_space_t space, vko_isa_t isa, size_t dim, const float *q,
                          const float *const *rows, const uint64_t *labels, size_t n, size_t k,
                          float *out_dist, uint64_t *out_label);

/* ---- L2: HNSW (hnswalg.h) -------------------------------------------------- */
typedef struct vko_hnsw vko_hnsw;
vko_hnsw *vko_hnsw_new(size_t dim, vko_space_t space, vko_isa_t isa, size_t max_elements,
                       size_t M, size_t ef_construction, size_t random_seed,
                       int allow_replace_deleted);
void vko_hnsw_free(vko_hnsw *h);
void vko_hnsw_set_ef(vko_hnsw *h, size_t ef);
/* addPoint(data,label,replace_deleted=allow_replace_deleted) (hnswalg.h:1278-1340).
 * 0 ok; 1 = exceeds limit; 2 = other runtime_error (message via vko_last_error) */
int vko_hnsw_add(vko_hnsw *h, const float *row, uint64_t label);
/* the same with the tombstoned slot to take over named by the caller (the reference takes
 * *deleted_elements.begin() of an unordered_set, hnswalg.h:1306-1309); 2 if not vacant */
int vko_hnsw_add_into(vko_hnsw *h, const float *row, uint64_t label, uint32_t slot);
size_t vko_hnsw_vacant(const vko_hnsw *h, uint32_t *out, size_t cap);
int vko_hnsw_mark_delete(vko_hnsw *h, uint64_t label); /* 0 ok, 2 error */
void vko_hnsw_resize(vko_hnsw *h, size_t new_max);
size_t vko_hnsw_count(const vko_hnsw *h);
size_t vko_hnsw_deleted_count(const vko_hnsw *h);
size_t vko_hnsw_capacity(const vko_hnsw *h);
int vko_hnsw_max_level(const vko_hnsw *h);
uint32_t vko_hnsw_entry_point(const vko_hnsw *h);
/* searchKnn (hnswalg.h:1659-1725) always through searchBaseLayerST<false>
 * (valkey-search always passes a cancel functor, vector_hnsw.cc:324-326).
 * ef_runtime = 0 -> index default.  n_eval/n_hops (optional) receive the
 * number of layer-0 distance evaluations / expanded nodes. */
size_t vko_hnsw_search(const vko_hnsw *h, const float *q, size_t k, size_t ef_runtime,
                       const uint64_t *allow_bits, uint64_t allow_nbits, long cancel_after,
                       float *out_dist, uint64_t *out_label, uint64_t *n_eval, uint64_t *n_hops);
int vko_hnsw_distance(const vko_hnsw *h, uint64_t label, const float *q, float *out);
/* graph export (for feeding the same graph to the device path in tests) */
int vko_hnsw_level_of(const vko_hnsw *h, uint32_t id);
uint64_t vko_hnsw_label_of(const vko_hnsw *h, uint32_t id);
int vko_hnsw_is_deleted(const vko_hnsw *h, uint32_t id);
/* copies the link list of `id` at `level` into out (capacity cap); returns count */
size_t vko_hnsw_links(const vko_hnsw *h, uint32_t id, int level, uint32_t *out, size_t cap);
const float *vko_hnsw_row(const vko_hnsw *h, uint32_t id);
/* install a graph built elsewhere (see hnsw.c); the index must be empty */
int vko_hnsw_load_graph(vko_hnsw *h, size_t n, const float *rows, const uint64_t *labels,
                        const uint32_t *l0_words, const uint64_t *upper_off, const uint32_t *upper_words,
                        int max_level, uint32_t entry_point);

/* one more searching thread over the same graph (own visited list); free it with vko_hnsw_free before the base */
vko_hnsw *vko_hnsw_view(const vko_hnsw *base);
/* SaveIndex chunk stream (hnswalg.h:808-865) -> oracle index, chunk by chunk: vko_sink_write has the signature of
 * the product's vk_write_chunk_fn */
typedef struct vko_sink vko_sink;
vko_sink *vko_sink_new(size_t dim, vko_space_t space, vko_isa_t isa, size_t M, size_t ef_construction);
int vko_sink_write(void *user, const void *data, uint64_t len);
vko_hnsw *vko_sink_finish(vko_sink *s);
/* the stream of a SHARDED product index (marker chunk + one SaveIndex stream per shard) -> one oracle graph per shard */
typedef struct vko_msink vko_msink;
vko_msink *vko_msink_new(size_t dim, vko_space_t space, vko_isa_t isa, size_t M, size_t ef_construction);
int vko_msink_write(void *user, const void *data, uint64_t len);
uint64_t vko_msink_count(vko_msink *m);
vko_hnsw *vko_msink_take(vko_msink *m, uint64_t i);
void vko_msink_free(vko_msink *m);

/* ---- cluster / shard merge (fanout.cc:162-175 semantics, made total) ------- */
/* k smallest by (dist,label) over `parts` lists of `per` entries each */
size_t vko_merge_topk(const float *dist, const uint64_t *label, const uint32_t *counts,
                      size_t parts, size_t per, size_t k, float *out_dist, uint64_t *out_label);

/* ---- libstdc++ helpers restated (exposed so tests can pin them) ------------- */
/* std::priority_queue<pair<float,uint32>, vector, CompareByFirst> (hnswalg.h:202-208) */
typedef struct { float d; uint32_t id; } vko_pair;
typedef struct { vko_pair *v; size_t n, cap; } vko_heap;
void vko_heap_init(vko_heap *h);
void vko_heap_free(vko_heap *h);
void vko_heap_push(vko_heap *h, float d, uint32_t id);
void vko_heap_pop(vko_heap *h);
/* std::default_random_engine (minstd_rand0) + uniform_real_distribution<double>(0,1) */
typedef struct { uint32_t x; } vko_minstd0;
void vko_minstd0_seed(vko_minstd0 *g, uint32_t seed);
uint32_t vko_minstd0_next(vko_minstd0 *g);
double vko_uniform01_double(vko_minstd0 *g);
float vko_uniform01_float(vko_minstd0 *g);
int vko_random_level(vko_minstd0 *g, double reverse_size); /* hnswalg.h:243-247 */

const char *vko_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* VKO_H_ */
