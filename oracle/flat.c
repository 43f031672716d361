/*
 * flat.c -- ORACLE (test infrastructure only, see vko.h).
 *
 * Restatement of hnswlib::BruteforceSearch<float> as forked by valkey-search
 * (third_party/hnswlib/bruteforce.h) plus the glue around it:
 *   addPoint      bruteforce.h:66-83    removePoint  bruteforce.h:92-113
 *   searchKnn     bruteforce.h:116-145  resizeIndex  bruteforce.h:209-211
 *   AddPrefilteredKey       src/indexes/vector_base.cc:509-530
 *   cluster merge (model)   src/query/fanout.cc:162-175
 * The element table is {row pointer, label}; deleting moves the last element
 * into the hole.  The result heap is std::priority_queue<pair<float,size_t>>,
 * i.e. a max-heap under the lexicographic pair order.
 */
#include <stdlib.h>
#include <string.h>

#include "vko_internal.h"

struct vko_flat {
    size_t dim;
    vko_space_t space;
    vko_isa_t isa;
    size_t cap, count;
    const float **rows; /* element i -> row */
    uint8_t *owned;     /* row storage owned by the oracle (copy) */
    uint64_t *labels;
    vko_map ext2int;    /* dict_external_to_internal */
    /* optional: fstdistfunc_ taken from the COMPILED REFERENCE (oracle/_ref: ref_InnerProductDistanceSimsimd /
     * ref_L2SqrSimsimd, i.e. third_party/hnswlib/simsimd.h:16-34 over SimSIMD 5.0.1 as the reference builds it) instead of
     * the restated kernels -- what bench.py's cpu_baseline times when _ref is present */
    float (*distfn)(const float *, const float *, size_t);
};

void vko_flat_set_distfn(vko_flat *f, float (*fn)(const float *, const float *, size_t)) { f->distfn = fn; }
static inline float flat_dist(const vko_flat *f, const float *q, const float *row) {
    return f->distfn ? f->distfn(q, row, f->dim) : vko_distance(f->space, f->isa, q, row, f->dim);
}

vko_flat *vko_flat_new(size_t dim, vko_space_t space, vko_isa_t isa, size_t max_elements) {
    vko_flat *f = (vko_flat *)calloc(1, sizeof(*f));
    f->dim = dim; f->space = space; f->isa = isa;
    f->cap = max_elements;
    f->rows = (const float **)calloc(max_elements ? max_elements : 1, sizeof(float *));
    f->owned = (uint8_t *)calloc(max_elements ? max_elements : 1, 1);
    f->labels = (uint64_t *)calloc(max_elements ? max_elements : 1, sizeof(uint64_t));
    vko_map_init(&f->ext2int);
    return f;
}

void vko_flat_free(vko_flat *f) {
    if (!f) return;
    for (size_t i = 0; i < f->count; ++i)
        if (f->owned[i]) free((void *)f->rows[i]);
    free(f->rows); free(f->owned); free(f->labels);
    vko_map_free(&f->ext2int);
    free(f);
}

size_t vko_flat_count(const vko_flat *f) { return f->count; }
size_t vko_flat_capacity(const vko_flat *f) { return f->cap; }

void vko_flat_resize(vko_flat *f, size_t new_max) { /* bruteforce.h:209-211 */
    f->rows = (const float **)realloc(f->rows, (new_max ? new_max : 1) * sizeof(float *));
    f->owned = (uint8_t *)realloc(f->owned, new_max ? new_max : 1);
    f->labels = (uint64_t *)realloc(f->labels, (new_max ? new_max : 1) * sizeof(uint64_t));
    f->cap = new_max;
}

static int flat_add(vko_flat *f, const float *row, uint64_t label, int copy) {
    /* bruteforce.h:66-83: an existing label is overwritten in place */
    uint32_t idx;
    if (!vko_map_get(&f->ext2int, label, &idx)) {
        if (f->count >= f->cap) {
            vko_set_error("The number of elements exceeds the specified limit\n");
            return 1;
        }
        idx = (uint32_t)f->count;
        vko_map_put(&f->ext2int, label, idx);
        f->count++;
        f->owned[idx] = 0;
    } else if (f->owned[idx]) {
        free((void *)f->rows[idx]);
        f->owned[idx] = 0;
    }
    f->labels[idx] = label;
    if (copy) {
        float *own = (float *)malloc(f->dim * sizeof(float));
        memcpy(own, row, f->dim * sizeof(float));
        f->rows[idx] = own;
        f->owned[idx] = 1;
    } else {
        f->rows[idx] = row;
    }
    return 0;
}
int vko_flat_add(vko_flat *f, const float *row, uint64_t label) { return flat_add(f, row, label, 1); }
int vko_flat_add_borrowed(vko_flat *f, const float *row, uint64_t label) { return flat_add(f, row, label, 0); }
/* n rows, `stride_bytes` apart, labels NULL = 0..n-1: addPoint in a loop (full-size baselines register 10M rows) */
int vko_flat_add_many(vko_flat *f, const float *rows, size_t stride_bytes, const uint64_t *labels, size_t n, int borrowed) {
    for (size_t i = 0; i < n; ++i) {
        int rc = flat_add(f, (const float *)((const char *)rows + i * stride_bytes), labels ? labels[i] : (uint64_t)i, !borrowed);
        if (rc) return rc;
    }
    return 0;
}

void vko_flat_remove(vko_flat *f, uint64_t label) { /* bruteforce.h:92-113 */
    uint32_t cur;
    if (!vko_map_get(&f->ext2int, label, &cur)) return;
    vko_map_del(&f->ext2int, label);
    if (f->owned[cur]) free((void *)f->rows[cur]);
    size_t last = f->count - 1;
    if (last != cur) {
        vko_map_put(&f->ext2int, f->labels[last], cur);
        f->rows[cur] = f->rows[last];
        f->owned[cur] = f->owned[last];
        f->labels[cur] = f->labels[last];
    }
    f->count--;
}

size_t vko_flat_search(const vko_flat *f, const float *q, size_t k, const uint64_t *allow_bits,
                       uint64_t allow_nbits, long cancel_after, float *out_dist,
                       uint64_t *out_label) {
    /* VectorFlat::Search clamps k (vector_flat.cc:234-236) */
    if (k > f->count) k = f->count;
    vko_dlheap top;
    vko_dlheap_init(&top);
    if (f->count == 0 || k == 0) { vko_dlheap_free(&top); return 0; }
    vko_cancel cancel = {cancel_after, 0};
    /* bruteforce.h:120-127: the first k rows are pushed unconditionally */
    for (size_t i = 0; i < k; i++) {
        float dist = flat_dist(f, q, f->rows[i]);
        if (vko_allowed(allow_bits, allow_nbits, f->labels[i])) vko_dlheap_push(&top, dist, f->labels[i]);
    }
    float lastdist = top.n == 0 ? 3.402823466e+38F : top.v[0].d;
    /* bruteforce.h:129-143 */
    for (size_t i = k; i < f->count && !vko_cancelled(&cancel); i++) {
        float dist = flat_dist(f, q, f->rows[i]);
        if (dist <= lastdist) {
            if (vko_allowed(allow_bits, allow_nbits, f->labels[i])) vko_dlheap_push(&top, dist, f->labels[i]);
            if (top.n > k) vko_dlheap_pop(&top);
            if (top.n != 0) lastdist = top.v[0].d;
        }
    }
    size_t n = vko_dlheap_drain_ascending(&top, out_dist, out_label);
    vko_dlheap_free(&top);
    return n;
}

int vko_flat_distance(const vko_flat *f, uint64_t label, const float *q, float *out) {
    uint32_t idx;
    if (!vko_map_get(&f->ext2int, label, &idx)) return 1;
    *out = vko_distance(f->space, f->isa, q, f->rows[idx], f->dim);
    return 0;
}

/* vector_base.cc:509-530: fill to k, then replace the top only on strict `<`
 * of the distance (ties keep the earlier key). */
size_t vko_prefilter_topk(vko_space_t space, vko_isa_t isa, size_t dim, const float *q,
                          const float *const *rows, const uint64_t *labels, size_t n, size_t k,
                          float *out_dist, uint64_t *out_label) {
    vko_dlheap results;
    vko_dlheap_init(&results);
    for (size_t i = 0; i < n; ++i) {
        float d = vko_distance(space, isa, q, rows[i], dim);
        if (results.n < k) {
            vko_dlheap_push(&results, d, labels[i]);
        } else if (k && d < results.v[0].d) {
            vko_dlheap_pop(&results);
            vko_dlheap_push(&results, d, labels[i]);
        }
    }
    size_t out = vko_dlheap_drain_ascending(&results, out_dist, out_label);
    vko_dlheap_free(&results);
    return out;
}

/* Shard merge.  The reference's cluster merge (fanout.cc:162-175) keeps k by
 * strict `<` on distance in arrival order, which is not reproducible; the
 * multi-GPU path uses the total order (dist,label) of bruteforce.h's heap so an
 * n-shard answer equals the 1-shard answer.  This is that rule. */
size_t vko_merge_topk(const float *dist, const uint64_t *label, const uint32_t *counts,
                      size_t parts, size_t per, size_t k, float *out_dist, uint64_t *out_label) {
    vko_dlheap top;
    vko_dlheap_init(&top);
    for (size_t p = 0; p < parts; ++p)
        for (size_t i = 0; i < counts[p]; ++i) {
            vko_dlheap_push(&top, dist[p * per + i], label[p * per + i]);
            if (top.n > k) vko_dlheap_pop(&top);
        }
    size_t n = vko_dlheap_drain_ascending(&top, out_dist, out_label);
    vko_dlheap_free(&top);
    return n;
}
