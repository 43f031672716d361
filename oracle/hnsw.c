/*
 * hnsw.c -- ORACLE (test infrastructure only, see vko.h).
 *
 * Single-threaded restatement of hnswlib::HierarchicalNSW<float> as forked by
 * valkey-search (third_party/hnswlib/hnswalg.h); locks are omitted, everything
 * else follows the reference step for step so that a single-threaded build from
 * the same inputs walks the same path:
 *   ctor / sizes              hnswalg.h:121-179     getRandomLevel   :243-247
 *   searchBaseLayer (build)   hnswalg.h:255-347     searchBaseLayerST<false> :351-551
 *   getNeighborsByHeuristic2  hnswalg.h:553-594     mutuallyConnectNewElement :613-756
 *   resizeIndex               hnswalg.h:758-777     markDelete / tombstone bit :1173-1270
 *   addPoint(replace_deleted) hnswalg.h:1278-1340   updatePoint :1342-1430
 *   repairConnectionsForUpdate hnswalg.h:1432-1511  addPoint(level) :1523-1650
 *   searchKnn                 hnswalg.h:1659-1725   VisitedList  visited_list_pool.h:9-33
 * Heaps are std::priority_queue with CompareByFirst (distance only), restated in
 * util.c with libstdc++'s sift order, because tie order is heap-defined.
 *
 * Deviations that cannot be pinned from the source (all tie/iteration-order
 * only): std::unordered_set iteration order in updatePoint (:1358-1426) is
 * insertion order here; the vacant slot picked by replace-deleted (:1313-1316,
 * unordered_set::begin) is the most recently tombstoned one here.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vko_internal.h"

#define DELETE_FLAG 0x00010000u /* bit 0 of byte 2 of the count word (hnswalg.h:1259-1262) */

struct vko_hnsw {
    size_t dim;
    vko_space_t space;
    vko_isa_t isa;
    size_t max_elements, count, num_deleted;
    size_t M, maxM, maxM0, efC, ef;
    double mult;
    int maxlevel;
    uint32_t enterpoint; /* 0xFFFFFFFF = none */
    uint32_t *l0;        /* [max][1+maxM0] */
    uint32_t **upper;    /* [max] -> [level][1+maxM] */
    int *levels;
    float **rows;
    uint64_t *labels;
    vko_map label_lookup;
    vko_minstd0 level_gen, update_gen;
    int allow_replace_deleted;
    uint32_t *vacant; size_t n_vacant, cap_vacant; /* deleted_elements */
    uint16_t *visited; uint16_t curV;
    /* layer-0 work counters of the last search */
    uint64_t n_eval, n_hops;
    int is_view;         /* vko_hnsw_view: shares every array of its base except the visited list */
};

static float h_dist(const vko_hnsw *h, const float *a, const float *b) {
    return vko_distance(h->space, h->isa, a, b, h->dim);
}
static uint32_t *ll0(const vko_hnsw *h, uint32_t id) { return h->l0 + (size_t)id * (h->maxM0 + 1); }
static uint32_t *llu(const vko_hnsw *h, uint32_t id, int level) {
    return h->upper[id] + (size_t)(level - 1) * (h->maxM + 1);
}
static uint32_t *ll_at(const vko_hnsw *h, uint32_t id, int level) {
    return level == 0 ? ll0(h, id) : llu(h, id, level);
}
static unsigned list_count(const uint32_t *ll) { return *ll & 0xFFFFu; }
static void set_list_count(uint32_t *ll, unsigned n) { *ll = (*ll & 0xFFFF0000u) | (n & 0xFFFFu); }
static int is_deleted(const vko_hnsw *h, uint32_t id) { return (*ll0(h, id) & DELETE_FLAG) != 0; }

static void visited_alloc(vko_hnsw *h) {
    free(h->visited);
    h->visited = (uint16_t *)malloc((h->max_elements ? h->max_elements : 1) * sizeof(uint16_t));
    h->curV = (uint16_t)-1; /* visited_list_pool.h:15 */
}
static uint16_t visited_next(vko_hnsw *h) { /* VisitedList::reset */
    h->curV++;
    if (h->curV == 0) {
        memset(h->visited, 0, sizeof(uint16_t) * h->max_elements);
        h->curV++;
    }
    return h->curV;
}

vko_hnsw *vko_hnsw_new(size_t dim, vko_space_t space, vko_isa_t isa, size_t max_elements,
                       size_t M, size_t ef_construction, size_t random_seed,
                       int allow_replace_deleted) {
    vko_hnsw *h = (vko_hnsw *)calloc(1, sizeof(*h));
    h->dim = dim; h->space = space; h->isa = isa;
    h->max_elements = max_elements;
    h->M = M <= 10000 ? M : 10000; /* hnswalg.h:130-141 */
    h->maxM = h->M;
    h->maxM0 = h->M * 2;
    h->efC = ef_construction > h->M ? ef_construction : h->M;
    h->ef = 10;
    vko_minstd0_seed(&h->level_gen, (uint32_t)random_seed);
    vko_minstd0_seed(&h->update_gen, (uint32_t)(random_seed + 1));
    h->mult = 1 / log(1.0 * (double)h->M);
    h->maxlevel = -1;
    h->enterpoint = 0xFFFFFFFFu;
    size_t n = max_elements ? max_elements : 1;
    h->l0 = (uint32_t *)calloc(n * (h->maxM0 + 1), sizeof(uint32_t));
    h->upper = (uint32_t **)calloc(n, sizeof(uint32_t *));
    h->levels = (int *)calloc(n, sizeof(int));
    h->rows = (float **)calloc(n, sizeof(float *));
    h->labels = (uint64_t *)calloc(n, sizeof(uint64_t));
    vko_map_init(&h->label_lookup);
    h->allow_replace_deleted = allow_replace_deleted;
    visited_alloc(h);
    return h;
}

void vko_hnsw_free(vko_hnsw *h) {
    if (!h) return;
    if (h->is_view) { free(h->visited); free(h); return; }
    for (size_t i = 0; i < h->count; ++i) { free(h->upper[i]); free(h->rows[i]); }
    free(h->l0); free(h->upper); free(h->levels); free(h->rows); free(h->labels);
    free(h->vacant); free(h->visited);
    vko_map_free(&h->label_lookup);
    free(h);
}

void vko_hnsw_set_ef(vko_hnsw *h, size_t ef) { h->ef = ef; }
size_t vko_hnsw_count(const vko_hnsw *h) { return h->count; }
size_t vko_hnsw_deleted_count(const vko_hnsw *h) { return h->num_deleted; }
size_t vko_hnsw_capacity(const vko_hnsw *h) { return h->max_elements; }
int vko_hnsw_max_level(const vko_hnsw *h) { return h->maxlevel; }
uint32_t vko_hnsw_entry_point(const vko_hnsw *h) { return h->enterpoint; }
int vko_hnsw_level_of(const vko_hnsw *h, uint32_t id) { return h->levels[id]; }
uint64_t vko_hnsw_label_of(const vko_hnsw *h, uint32_t id) { return h->labels[id]; }
int vko_hnsw_is_deleted(const vko_hnsw *h, uint32_t id) { return is_deleted(h, id); }
const float *vko_hnsw_row(const vko_hnsw *h, uint32_t id) { return h->rows[id]; }
size_t vko_hnsw_links(const vko_hnsw *h, uint32_t id, int level, uint32_t *out, size_t cap) {
    if (level > h->levels[id]) return 0;
    const uint32_t *ll = ll_at(h, id, level);
    size_t n = list_count(ll);
    for (size_t i = 0; i < n && i < cap; ++i) out[i] = ll[1 + i];
    return n;
}

void vko_hnsw_resize(vko_hnsw *h, size_t new_max) { /* hnswalg.h:758-777 */
    if (new_max < h->count) { vko_set_error("Cannot resize, max element is less than the current number of elements"); return; }
    size_t old = h->max_elements;
    h->l0 = (uint32_t *)realloc(h->l0, new_max * (h->maxM0 + 1) * sizeof(uint32_t));
    h->upper = (uint32_t **)realloc(h->upper, new_max * sizeof(uint32_t *));
    h->levels = (int *)realloc(h->levels, new_max * sizeof(int));
    h->rows = (float **)realloc(h->rows, new_max * sizeof(float *));
    h->labels = (uint64_t *)realloc(h->labels, new_max * sizeof(uint64_t));
    if (new_max > old) {
        memset(h->l0 + old * (h->maxM0 + 1), 0, (new_max - old) * (h->maxM0 + 1) * sizeof(uint32_t));
        memset(h->upper + old, 0, (new_max - old) * sizeof(uint32_t *));
        memset(h->levels + old, 0, (new_max - old) * sizeof(int));
        memset(h->rows + old, 0, (new_max - old) * sizeof(float *));
    }
    h->max_elements = new_max;
    visited_alloc(h); /* a fresh VisitedListPool */
}

/* ---- build-time beam search: hnswalg.h:255-347 ------------------------------- */
static void search_base_layer(vko_hnsw *h, uint32_t ep_id, const float *q, int layer, vko_heap *top) {
    uint16_t tag = visited_next(h);
    uint16_t *visited = h->visited;
    vko_heap cand;
    vko_heap_init(&cand);
    float lowerBound;
    if (!is_deleted(h, ep_id)) {
        float dist = h_dist(h, q, h->rows[ep_id]);
        vko_heap_push(top, dist, ep_id);
        lowerBound = dist;
        vko_heap_push(&cand, -dist, ep_id);
    } else {
        lowerBound = 3.402823466e+38F;
        vko_heap_push(&cand, -lowerBound, ep_id);
    }
    visited[ep_id] = tag;
    while (cand.n) {
        vko_pair cur = cand.v[0];
        if ((-cur.d) > lowerBound && top->n == h->efC) break;
        vko_heap_pop(&cand);
        const uint32_t *ll = ll_at(h, cur.id, layer);
        size_t size = list_count(ll);
        for (size_t j = 0; j < size; j++) {
            uint32_t cid = ll[1 + j];
            if (visited[cid] == tag) continue;
            visited[cid] = tag;
            float d1 = h_dist(h, q, h->rows[cid]);
            if (top->n < h->efC || lowerBound > d1) {
                vko_heap_push(&cand, -d1, cid);
                if (!is_deleted(h, cid)) vko_heap_push(top, d1, cid);
                if (top->n > h->efC) vko_heap_pop(top);
                if (top->n) lowerBound = top->v[0].d;
            }
        }
    }
    vko_heap_free(&cand);
}

/* ---- heuristic: hnswalg.h:553-594 -------------------------------------------- */
static int cmp_closest(const void *pa, const void *pb) {
    /* pop order of priority_queue<pair<float,uint>> holding (-dist,id):
     * smallest dist first; equal dist -> larger id first */
    const vko_pair *a = (const vko_pair *)pa, *b = (const vko_pair *)pb;
    if (a->d < b->d) return -1;
    if (a->d > b->d) return 1;
    if (a->id > b->id) return -1;
    if (a->id < b->id) return 1;
    return 0;
}
static void neighbors_by_heuristic2(vko_hnsw *h, vko_heap *top, size_t M) {
    if (top->n < M) return;
    size_t n = top->n;
    vko_pair *closest = (vko_pair *)malloc(n * sizeof(vko_pair));
    for (size_t i = 0; i < n; ++i) { closest[i] = top->v[0]; vko_heap_pop(top); }
    qsort(closest, n, sizeof(vko_pair), cmp_closest);
    vko_pair *ret = (vko_pair *)malloc((M ? M : 1) * sizeof(vko_pair));
    size_t nret = 0;
    for (size_t i = 0; i < n; ++i) {
        if (nret >= M) break;
        float dist_to_query = closest[i].d;
        int good = 1;
        for (size_t s = 0; s < nret; ++s) {
            float curdist = h_dist(h, h->rows[ret[s].id], h->rows[closest[i].id]);
            if (curdist < dist_to_query) { good = 0; break; }
        }
        if (good) ret[nret++] = closest[i];
    }
    for (size_t s = 0; s < nret; ++s) vko_heap_push(top, ret[s].d, ret[s].id);
    free(closest); free(ret);
}

/* ---- hnswalg.h:613-756; returns 0xFFFFFFFF on a reference runtime_error -------- */
static uint32_t mutually_connect(vko_hnsw *h, uint32_t cur_c, vko_heap *top, int level, int isUpdate) {
    size_t Mcurmax = level ? h->maxM : h->maxM0;
    neighbors_by_heuristic2(h, top, h->M);
    if (top->n > h->M) { vko_set_error("Should be not be more than M_ candidates returned by the heuristic"); return 0xFFFFFFFFu; }
    size_t nsel = top->n;
    uint32_t *sel = (uint32_t *)malloc((nsel ? nsel : 1) * sizeof(uint32_t));
    for (size_t i = 0; i < nsel; ++i) { sel[i] = top->v[0].id; vko_heap_pop(top); }
    if (nsel == 0) { free(sel); vko_set_error("During insertion, no neighbors found to mutually connect to"); return 0xFFFFFFFFu; }
    uint32_t next_closest = sel[nsel - 1];
    {
        uint32_t *ll_cur = ll_at(h, cur_c, level);
        if (*ll_cur && !isUpdate) { free(sel); vko_set_error("The newly inserted element should have blank link list"); return 0xFFFFFFFFu; }
        set_list_count(ll_cur, (unsigned)nsel);
        for (size_t i = 0; i < nsel; ++i) {
            if (ll_cur[1 + i] && !isUpdate) { free(sel); vko_set_error("Possible memory corruption"); return 0xFFFFFFFFu; }
            if (level > h->levels[sel[i]]) { free(sel); vko_set_error("Trying to make a link on a non-existent level"); return 0xFFFFFFFFu; }
            ll_cur[1 + i] = sel[i];
        }
    }
    for (size_t i = 0; i < nsel; ++i) {
        uint32_t nb = sel[i];
        uint32_t *ll_other = ll_at(h, nb, level);
        size_t sz = list_count(ll_other);
        if (sz > Mcurmax) { free(sel); vko_set_error("Bad value of sz_link_list_other"); return 0xFFFFFFFFu; }
        if (nb == cur_c) { free(sel); vko_set_error("Trying to connect an element to itself"); return 0xFFFFFFFFu; }
        if (level > h->levels[nb]) { free(sel); vko_set_error("Trying to make a link on a non-existent level"); return 0xFFFFFFFFu; }
        uint32_t *data = ll_other + 1;
        int present = 0;
        if (isUpdate)
            for (size_t j = 0; j < sz; j++)
                if (data[j] == cur_c) { present = 1; break; }
        if (present) continue;
        if (sz < Mcurmax) {
            data[sz] = cur_c;
            set_list_count(ll_other, (unsigned)(sz + 1));
        } else {
            float d_max = h_dist(h, h->rows[cur_c], h->rows[nb]);
            vko_heap cands;
            vko_heap_init(&cands);
            vko_heap_push(&cands, d_max, cur_c);
            for (size_t j = 0; j < sz; j++) vko_heap_push(&cands, h_dist(h, h->rows[data[j]], h->rows[nb]), data[j]);
            neighbors_by_heuristic2(h, &cands, Mcurmax);
            unsigned indx = 0;
            while (cands.n) { data[indx++] = cands.v[0].id; vko_heap_pop(&cands); }
            set_list_count(ll_other, indx);
            vko_heap_free(&cands);
        }
    }
    free(sel);
    return next_closest;
}

static void mark_deleted_internal(vko_hnsw *h, uint32_t id, int *err) { /* hnswalg.h:1194-1209 */
    if (!is_deleted(h, id)) {
        *ll0(h, id) |= DELETE_FLAG;
        h->num_deleted += 1;
        if (h->allow_replace_deleted) {
            if (h->n_vacant == h->cap_vacant) {
                h->cap_vacant = h->cap_vacant ? h->cap_vacant * 2 : 16;
                h->vacant = (uint32_t *)realloc(h->vacant, h->cap_vacant * sizeof(uint32_t));
            }
            h->vacant[h->n_vacant++] = id;
        }
    } else {
        vko_set_error("The requested to delete element is already deleted");
        *err = 1;
    }
}
static void vacant_erase(vko_hnsw *h, uint32_t id) {
    for (size_t i = 0; i < h->n_vacant; ++i)
        if (h->vacant[i] == id) { h->vacant[i] = h->vacant[--h->n_vacant]; return; }
}
static void unmark_deleted_internal(vko_hnsw *h, uint32_t id, int *err) { /* hnswalg.h:1236-1251 */
    if (is_deleted(h, id)) {
        *ll0(h, id) &= ~DELETE_FLAG;
        h->num_deleted -= 1;
        if (h->allow_replace_deleted) vacant_erase(h, id);
    } else {
        vko_set_error("The requested to undelete element is not deleted");
        *err = 1;
    }
}

int vko_hnsw_mark_delete(vko_hnsw *h, uint64_t label) { /* hnswalg.h:1173-1187 */
    uint32_t id;
    if (!vko_map_get(&h->label_lookup, label, &id)) { vko_set_error("Label not found"); return 2; }
    int err = 0;
    mark_deleted_internal(h, id, &err);
    return err ? 2 : 0;
}

/* ---- insertion-ordered small set (stands in for std::unordered_set) ------------ */
typedef struct { uint32_t *v; size_t n, cap; } idset;
static int idset_has(const idset *s, uint32_t x) {
    for (size_t i = 0; i < s->n; ++i) if (s->v[i] == x) return 1;
    return 0;
}
static void idset_add(idset *s, uint32_t x) {
    if (idset_has(s, x)) return;
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 64; s->v = (uint32_t *)realloc(s->v, s->cap * sizeof(uint32_t)); }
    s->v[s->n++] = x;
}

static int repair_connections_for_update(vko_hnsw *h, const float *q, uint32_t ep, uint32_t id,
                                         int dataPointLevel, int maxLevel);

/* hnswalg.h:1342-1430 */
static int update_point(vko_hnsw *h, const float *row, uint32_t id, float updateNeighborProbability) {
    memcpy(h->rows[id], row, h->dim * sizeof(float)); /* *data_ptr = dataPoint */
    int maxLevelCopy = h->maxlevel;
    uint32_t entryPointCopy = h->enterpoint;
    if (entryPointCopy == id && h->count == 1) return 0;
    int elemLevel = h->levels[id];
    for (int layer = 0; layer <= elemLevel; layer++) {
        idset sCand = {0, 0, 0}, sNeigh = {0, 0, 0};
        const uint32_t *ll = ll_at(h, id, layer);
        size_t n1 = list_count(ll);
        if (n1 == 0) continue;
        uint32_t *oneHop = (uint32_t *)malloc(n1 * sizeof(uint32_t));
        memcpy(oneHop, ll + 1, n1 * sizeof(uint32_t));
        idset_add(&sCand, id);
        for (size_t a = 0; a < n1; ++a) {
            uint32_t el = oneHop[a];
            idset_add(&sCand, el);
            if (vko_uniform01_float(&h->update_gen) > updateNeighborProbability) continue;
            idset_add(&sNeigh, el);
            const uint32_t *l2 = ll_at(h, el, layer);
            size_t n2 = list_count(l2);
            for (size_t b = 0; b < n2; ++b) idset_add(&sCand, l2[1 + b]);
        }
        for (size_t a = 0; a < sNeigh.n; ++a) {
            uint32_t neigh = sNeigh.v[a];
            vko_heap cands;
            vko_heap_init(&cands);
            size_t size = idset_has(&sCand, neigh) ? sCand.n - 1 : sCand.n;
            size_t keep = h->efC < size ? h->efC : size;
            for (size_t b = 0; b < sCand.n; ++b) {
                uint32_t cand = sCand.v[b];
                if (cand == neigh) continue;
                float distance = h_dist(h, h->rows[neigh], h->rows[cand]);
                if (cands.n < keep) {
                    vko_heap_push(&cands, distance, cand);
                } else if (cands.n && distance < cands.v[0].d) {
                    vko_heap_pop(&cands);
                    vko_heap_push(&cands, distance, cand);
                }
            }
            neighbors_by_heuristic2(h, &cands, layer == 0 ? h->maxM0 : h->maxM);
            uint32_t *ll_cur = ll_at(h, neigh, layer);
            size_t candSize = cands.n;
            set_list_count(ll_cur, (unsigned)candSize);
            for (size_t idx = 0; idx < candSize; idx++) { ll_cur[1 + idx] = cands.v[0].id; vko_heap_pop(&cands); }
            vko_heap_free(&cands);
        }
        free(oneHop); free(sCand.v); free(sNeigh.v);
    }
    return repair_connections_for_update(h, h->rows[id], entryPointCopy, id, elemLevel, maxLevelCopy);
}

/* hnswalg.h:1432-1511 */
static int repair_connections_for_update(vko_hnsw *h, const float *q, uint32_t ep, uint32_t id,
                                         int dataPointLevel, int maxLevel) {
    uint32_t currObj = ep;
    if (dataPointLevel < maxLevel) {
        float curdist = h_dist(h, q, h->rows[currObj]);
        for (int level = maxLevel; level > dataPointLevel; level--) {
            int changed = 1;
            while (changed) {
                changed = 0;
                const uint32_t *ll = ll_at(h, currObj, level);
                int size = (int)list_count(ll);
                for (int i = 0; i < size; i++) {
                    uint32_t cand = ll[1 + i];
                    float d = h_dist(h, q, h->rows[cand]);
                    if (d < curdist) { curdist = d; currObj = cand; changed = 1; }
                }
            }
        }
    }
    if (dataPointLevel > maxLevel) { vko_set_error("Level of item to be updated cannot be bigger than max level"); return 2; }
    for (int level = dataPointLevel; level >= 0; level--) {
        vko_heap topc, filtered;
        vko_heap_init(&topc);
        vko_heap_init(&filtered);
        search_base_layer(h, currObj, q, level, &topc);
        while (topc.n) {
            if (topc.v[0].id != id) vko_heap_push(&filtered, topc.v[0].d, topc.v[0].id);
            vko_heap_pop(&topc);
        }
        if (filtered.n > 0) {
            if (is_deleted(h, ep)) {
                vko_heap_push(&filtered, h_dist(h, q, h->rows[ep]), ep);
                if (filtered.n > h->efC) vko_heap_pop(&filtered);
            }
            currObj = mutually_connect(h, id, &filtered, level, 1);
            if (currObj == 0xFFFFFFFFu) { vko_heap_free(&topc); vko_heap_free(&filtered); return 2; }
        }
        vko_heap_free(&topc);
        vko_heap_free(&filtered);
    }
    return 0;
}

/* hnswalg.h:1523-1650 */
static int add_point_level(vko_hnsw *h, const float *row, uint64_t label, int level_in) {
    uint32_t cur_c;
    uint32_t existing;
    if (vko_map_get(&h->label_lookup, label, &existing)) {
        if (h->allow_replace_deleted && is_deleted(h, existing)) {
            vko_set_error("Can't use addPoint to update deleted elements if replacement of deleted elements is enabled.");
            return 2;
        }
        int err = 0;
        if (is_deleted(h, existing)) unmark_deleted_internal(h, existing, &err);
        return update_point(h, row, existing, 1.0f);
    }
    if (h->count >= h->max_elements) { vko_set_error("The number of elements exceeds the specified limit"); return 1; }
    cur_c = (uint32_t)h->count;
    h->count++;
    vko_map_put(&h->label_lookup, label, cur_c);

    int maxlevelcopy = h->maxlevel;
    int curlevel = vko_random_level(&h->level_gen, h->mult);
    if (level_in > 0) curlevel = level_in;
    h->levels[cur_c] = curlevel;
    uint32_t currObj = h->enterpoint;
    uint32_t enterpoint_copy = h->enterpoint;

    memset(ll0(h, cur_c), 0, (h->maxM0 + 1) * sizeof(uint32_t));
    h->labels[cur_c] = label;
    h->rows[cur_c] = (float *)malloc(h->dim * sizeof(float));
    memcpy(h->rows[cur_c], row, h->dim * sizeof(float));
    free(h->upper[cur_c]);
    h->upper[cur_c] = curlevel ? (uint32_t *)calloc((size_t)curlevel * (h->maxM + 1), sizeof(uint32_t)) : 0;

    if (currObj != 0xFFFFFFFFu) {
        if (curlevel < maxlevelcopy) {
            float curdist = h_dist(h, row, h->rows[currObj]);
            for (int level = maxlevelcopy; level > curlevel; level--) {
                int changed = 1;
                while (changed) {
                    changed = 0;
                    const uint32_t *ll = llu(h, currObj, level);
                    int size = (int)list_count(ll);
                    for (int i = 0; i < size; i++) {
                        uint32_t cand = ll[1 + i];
                        if (cand > h->max_elements) { vko_set_error("cand error"); return 2; }
                        float d = h_dist(h, row, h->rows[cand]);
                        if (d < curdist) { curdist = d; currObj = cand; changed = 1; }
                    }
                }
            }
        }
        int epDeleted = is_deleted(h, enterpoint_copy);
        for (int level = curlevel < maxlevelcopy ? curlevel : maxlevelcopy; level >= 0; level--) {
            vko_heap topc;
            vko_heap_init(&topc);
            search_base_layer(h, currObj, row, level, &topc);
            if (epDeleted) {
                vko_heap_push(&topc, h_dist(h, row, h->rows[enterpoint_copy]), enterpoint_copy);
                if (topc.n > h->efC) vko_heap_pop(&topc);
            }
            currObj = mutually_connect(h, cur_c, &topc, level, 0);
            vko_heap_free(&topc);
            if (currObj == 0xFFFFFFFFu) return 2;
        }
    } else {
        h->enterpoint = 0;
        h->maxlevel = curlevel;
    }
    if (curlevel > maxlevelcopy) {
        h->enterpoint = cur_c;
        h->maxlevel = curlevel;
    }
    return 0;
}

/* hnswalg.h:1306-1338: the new label takes over tombstoned slot `replaced` */
static int replace_vacant(vko_hnsw *h, const float *row, uint64_t label, uint32_t replaced) {
    vacant_erase(h, replaced);
    uint64_t label_replaced = h->labels[replaced];
    h->labels[replaced] = label;
    vko_map_del(&h->label_lookup, label_replaced);
    vko_map_put(&h->label_lookup, label, replaced);
    int err = 0;
    unmark_deleted_internal(h, replaced, &err);
    return update_point(h, row, replaced, 1.0f);
}

/* hnswalg.h:1278-1340 with replace_deleted = allow_replace_deleted_
 * (vector_hnsw.cc:182-183) */
int vko_hnsw_add(vko_hnsw *h, const float *row, uint64_t label) {
    if (!h->allow_replace_deleted) return add_point_level(h, row, label, -1);
    uint32_t existing;
    if (vko_map_get(&h->label_lookup, label, &existing)) {
        int err = 0;
        if (is_deleted(h, existing)) { vacant_erase(h, existing); unmark_deleted_internal(h, existing, &err); }
        return update_point(h, row, existing, 1.0f);
    }
    if (h->n_vacant == 0) return add_point_level(h, row, label, -1);
    return replace_vacant(h, row, label, h->vacant[h->n_vacant - 1]);
}

/* The same, with the tombstoned slot to take over NAMED by the caller.  hnswalg.h:1306-1309
 * takes `*deleted_elements.begin()` of a std::unordered_set<tableint>: which vacant slot that
 * is depends on the container's bucket history and is not part of the algorithm.  A
 * differential test lets the implementation under test choose and replays the choice here;
 * the slot must be vacant (else 2).  With no vacant slot or a known label: vko_hnsw_add. */
int vko_hnsw_add_into(vko_hnsw *h, const float *row, uint64_t label, uint32_t slot) {
    uint32_t existing;
    if (!h->allow_replace_deleted || vko_map_get(&h->label_lookup, label, &existing) || h->n_vacant == 0)
        return vko_hnsw_add(h, row, label);
    for (size_t i = 0; i < h->n_vacant; ++i)
        if (h->vacant[i] == slot) return replace_vacant(h, row, label, slot);
    vko_set_error("the named slot is not vacant");
    return 2;
}

size_t vko_hnsw_vacant(const vko_hnsw *h, uint32_t *out, size_t cap) {
    for (size_t i = 0; i < h->n_vacant && i < cap; ++i) out[i] = h->vacant[i];
    return h->n_vacant;
}

/* ---- query: searchBaseLayerST<false,false>, hnswalg.h:351-551 ------------------ */
static void search_base_layer_st(vko_hnsw *h, uint32_t ep_id, const float *q, size_t ef,
                                 const uint64_t *allow_bits, uint64_t allow_nbits,
                                 vko_cancel *cancel, vko_heap *top) {
    uint16_t tag = visited_next(h);
    uint16_t *visited = h->visited;
    vko_heap cand;
    vko_heap_init(&cand);
    float lowerBound;
    if (!is_deleted(h, ep_id) && vko_allowed(allow_bits, allow_nbits, h->labels[ep_id])) {
        float dist = h_dist(h, q, h->rows[ep_id]);
        h->n_eval++;
        lowerBound = dist;
        vko_heap_push(top, dist, ep_id);
        vko_heap_push(&cand, -dist, ep_id);
    } else {
        lowerBound = 3.402823466e+38F;
        vko_heap_push(&cand, -lowerBound, ep_id);
    }
    visited[ep_id] = tag;
    uint32_t *unvisited = (uint32_t *)malloc((h->maxM0 + 1) * sizeof(uint32_t));
    while (cand.n) {
        vko_pair cur = cand.v[0];
        float candidate_dist = -cur.d;
        int stop;
        if (vko_cancelled(cancel)) stop = 1;
        else stop = candidate_dist > lowerBound && top->n == ef;
        if (stop) break;
        vko_heap_pop(&cand);
        h->n_hops++;
        const uint32_t *ll = ll0(h, cur.id);
        size_t size = list_count(ll);
        /* phase 1 (:453-464): unvisited neighbours in list order */
        size_t nun = 0;
        for (size_t j = 1; j <= size; j++) {
            uint32_t cid = ll[j];
            if (visited[cid] != tag) { visited[cid] = tag; unvisited[nun++] = cid; }
        }
        /* phase 3 (:483-548) */
        for (size_t u = 0; u < nun; u++) {
            uint32_t cid = unvisited[u];
            float dist = h_dist(h, q, h->rows[cid]);
            h->n_eval++;
            int consider = top->n < ef || lowerBound > dist;
            if (consider) {
                vko_heap_push(&cand, -dist, cid);
                if (!is_deleted(h, cid) && vko_allowed(allow_bits, allow_nbits, h->labels[cid]))
                    vko_heap_push(top, dist, cid);
                while (top->n > ef) vko_heap_pop(top);
                if (top->n) lowerBound = top->v[0].d;
            }
        }
    }
    free(unvisited);
    vko_heap_free(&cand);
}

static int cmp_dl(const void *pa, const void *pb) {
    const vko_dl *a = (const vko_dl *)pa, *b = (const vko_dl *)pb;
    if (a->d < b->d) return -1;
    if (a->d > b->d) return 1;
    if (a->label < b->label) return -1;
    if (a->label > b->label) return 1;
    return 0;
}

/* hnswalg.h:1659-1725 + VectorBase::CreateReply ordering */
size_t vko_hnsw_search(const vko_hnsw *hc, const float *q, size_t k, size_t ef_runtime,
                       const uint64_t *allow_bits, uint64_t allow_nbits, long cancel_after,
                       float *out_dist, uint64_t *out_label, uint64_t *n_eval, uint64_t *n_hops) {
    vko_hnsw *h = (vko_hnsw *)hc; /* the visited list and counters are mutable state */
    h->n_eval = h->n_hops = 0;
    if (n_eval) *n_eval = 0;
    if (n_hops) *n_hops = 0;
    if (h->count == 0) return 0;
    vko_cancel cancel = {cancel_after, 0};
    uint32_t currObj = h->enterpoint;
    float curdist = h_dist(h, q, h->rows[h->enterpoint]);
    for (int level = h->maxlevel; level > 0; level--) {
        int changed = 1;
        while (changed) {
            changed = 0;
            const uint32_t *ll = llu(h, currObj, level);
            int size = (int)list_count(ll);
            for (int i = 0; i < size; i++) {
                uint32_t cand = ll[1 + i];
                if (cand > h->max_elements) { vko_set_error("cand error"); return 0; }
                float d = h_dist(h, q, h->rows[cand]);
                if (d < curdist) { curdist = d; currObj = cand; changed = 1; }
            }
        }
    }
    size_t ef = ef_runtime ? ef_runtime : h->ef;
    if (ef < k) ef = k;
    vko_heap top;
    vko_heap_init(&top);
    search_base_layer_st(h, currObj, q, ef, allow_bits, allow_nbits, &cancel, &top);
    while (top.n > k) vko_heap_pop(&top);
    size_t n = top.n;
    vko_dl *res = (vko_dl *)malloc((n ? n : 1) * sizeof(vko_dl));
    for (size_t i = 0; i < n; ++i) { res[i].d = top.v[0].d; res[i].label = h->labels[top.v[0].id]; vko_heap_pop(&top); }
    qsort(res, n, sizeof(vko_dl), cmp_dl);
    for (size_t i = 0; i < n; ++i) { out_dist[i] = res[i].d; out_label[i] = res[i].label; }
    free(res);
    vko_heap_free(&top);
    if (n_eval) *n_eval = h->n_eval;
    if (n_hops) *n_hops = h->n_hops;
    return n;
}

/* vector_hnsw.cc:369-383 (tombstoned labels are "not found", :55-64) */
int vko_hnsw_distance(const vko_hnsw *h, uint64_t label, const float *q, float *out) {
    uint32_t id;
    if (!vko_map_get(&h->label_lookup, label, &id) || is_deleted(h, id)) return 1;
    *out = h_dist(h, q, h->rows[id]);
    return 0;
}

/* ---- bulk load of an existing graph (test infrastructure: lets the oracle SEARCH a graph that
 * was built elsewhere, e.g. by the product's multi-threaded host builder, so GPU and CPU answers
 * can be compared on the very same graph).  Layout mirrors the chunk stream of SaveIndex
 * (hnswalg.h:808-865): per element a level-0 word block [1+2M], a row, a label; then the
 * concatenated upper-level blocks, element i owning words [upper_off[i], upper_off[i+1]). */
int vko_hnsw_load_graph(vko_hnsw *h, size_t n, const float *rows, const uint64_t *labels,
                        const uint32_t *l0_words, const uint64_t *upper_off, const uint32_t *upper_words,
                        int max_level, uint32_t entry_point) {
    if (n > h->max_elements || h->count != 0) { vko_set_error("load_graph: index must be empty and large enough"); return 2; }
    for (size_t i = 0; i < n; ++i) {
        memcpy(ll0(h, (uint32_t)i), l0_words + i * (h->maxM0 + 1), (h->maxM0 + 1) * sizeof(uint32_t));
        h->rows[i] = (float *)malloc(h->dim * sizeof(float));
        memcpy(h->rows[i], rows + i * h->dim, h->dim * sizeof(float));
        h->labels[i] = labels[i];
        size_t nw = (size_t)(upper_off[i + 1] - upper_off[i]);
        if (nw % (h->maxM + 1)) { vko_set_error("load_graph: bad upper block size"); return 2; }
        h->levels[i] = (int)(nw / (h->maxM + 1));
        if (nw) {
            h->upper[i] = (uint32_t *)malloc(nw * sizeof(uint32_t));
            memcpy(h->upper[i], upper_words + upper_off[i], nw * sizeof(uint32_t));
        }
        if (is_deleted(h, (uint32_t)i)) h->num_deleted++;
        else vko_map_put(&h->label_lookup, labels[i], (uint32_t)i);
    }
    h->count = n;
    h->maxlevel = max_level;
    h->enterpoint = entry_point;
    return 0;
}

/* A read-only view of `base` for one more searching thread: hnswlib hands every concurrent search its own
 * VisitedList (visited_list_pool.h:35-77); the restatement keeps one per object, so a thread that wants to search
 * the same graph concurrently takes a view (own visited list and work counters, everything else shared).  The base
 * must not be mutated or freed while views are alive. */
vko_hnsw *vko_hnsw_view(const vko_hnsw *base) {
    vko_hnsw *v = (vko_hnsw *)malloc(sizeof(*v));
    memcpy(v, base, sizeof(*v));
    v->is_view = 1;
    v->visited = NULL;
    visited_alloc(v);
    return v;
}

/* ---- SaveIndex chunk stream -> oracle graph, chunk by chunk (hnswalg.h:808-865) ---------------------------
 * vko_sink_write has the signature of the product's vk_write_chunk_fn, so a test can hand it straight to
 * vk_index_save and the CPU restatement searches the very graph the device searches (10M elements go through
 * without 10M Python objects).  Chunk 0 = HNSWIndexHeader (varint fields: 3 = element count, 7 = max level,
 * 8 = entry point), then one chunk per element [level-0 words | row | label], then per element a u64 size chunk
 * followed, when non-zero, by the block of its upper-level lists. */
struct vko_sink {
    vko_hnsw *h;
    size_t dim, M, efC;
    vko_space_t space;
    vko_isa_t isa;
    size_t n, elem, upper_i;
    int state;          /* 0 header, 1 elements, 2 size chunk, 3 upper block, 4 done, -1 error */
    uint64_t pending;   /* bytes of the upper block announced by the last size chunk */
    int max_level;
    uint32_t entry_point;
};
typedef struct vko_sink vko_sink;

vko_sink *vko_sink_new(size_t dim, vko_space_t space, vko_isa_t isa, size_t M, size_t ef_construction) {
    vko_sink *s = (vko_sink *)calloc(1, sizeof(*s));
    s->dim = dim; s->space = space; s->isa = isa; s->M = M; s->efC = ef_construction;
    return s;
}

static int sink_fail(vko_sink *s, const char *msg) { vko_set_error(msg); s->state = -1; return 1; }

int vko_sink_write(void *user, const void *data, uint64_t len) {
    vko_sink *s = (vko_sink *)user;
    const uint8_t *p = (const uint8_t *)data;
    switch (s->state) {
    case 0: {
        uint64_t f[16] = {0};
        const uint8_t *e = p + len;
        while (p < e) {
            uint8_t key = *p++;
            unsigned field = key >> 3, wire = key & 7;
            if (wire == 0) {
                uint64_t v = 0; int sh = 0;
                while (p < e) { uint8_t b = *p++; v |= (uint64_t)(b & 0x7F) << sh; sh += 7; if (!(b & 0x80)) break; }
                if (field < 16) f[field] = v;
            } else if (wire == 1) p += 8;
            else return sink_fail(s, "sink: unexpected wire type in the header");
        }
        s->n = (size_t)f[3];
        s->max_level = (int)(int32_t)(uint32_t)f[7];
        s->entry_point = (uint32_t)f[8];
        s->h = vko_hnsw_new(s->dim, s->space, s->isa, s->n ? s->n : 1, s->M, s->efC, 100, 0);
        s->state = s->n ? 1 : 4;
        return 0;
    }
    case 1: {
        vko_hnsw *h = s->h;
        const size_t sl0 = (h->maxM0 + 1) * 4, vec = h->dim * 4;
        if (len != sl0 + vec + 8) return sink_fail(s, "sink: element chunk has the wrong size");
        const uint32_t i = (uint32_t)s->elem;
        memcpy(ll0(h, i), p, sl0);
        h->rows[i] = (float *)malloc(vec);
        memcpy(h->rows[i], p + sl0, vec);
        memcpy(&h->labels[i], p + sl0 + vec, 8);
        if (is_deleted(h, i)) h->num_deleted++;
        else vko_map_put(&h->label_lookup, h->labels[i], i);
        h->count = ++s->elem;      /* (free() walks count elements) */
        if (s->elem == s->n) s->state = 2;
        return 0;
    }
    case 2: {
        if (len != 8) return sink_fail(s, "sink: size chunk has the wrong size");
        memcpy(&s->pending, p, 8);
        if (s->pending) { s->state = 3; return 0; }
        if (++s->upper_i == s->n) s->state = 4;
        return 0;
    }
    case 3: {
        vko_hnsw *h = s->h;
        if (len != s->pending || len % ((h->maxM + 1) * 4)) return sink_fail(s, "sink: upper block has the wrong size");
        const uint32_t i = (uint32_t)s->upper_i;
        h->levels[i] = (int)(len / ((h->maxM + 1) * 4));
        h->upper[i] = (uint32_t *)malloc(len);
        memcpy(h->upper[i], p, len);
        s->state = ++s->upper_i == s->n ? 4 : 2;
        return 0;
    }
    default:
        return sink_fail(s, "sink: chunk after the end of the stream");
    }
}

/* the finished oracle index (ownership passes to the caller), NULL when the stream was incomplete; frees the sink */
vko_hnsw *vko_sink_finish(vko_sink *s) {
    vko_hnsw *h = NULL;
    if (s->state == 4 && s->h) {
        h = s->h;
        h->maxlevel = s->n ? s->max_level : -1;
        h->enterpoint = s->n ? s->entry_point : 0xFFFFFFFFu;
    } else {
        if (s->state != -1) vko_set_error("sink: incomplete chunk stream");
        vko_hnsw_free(s->h);
    }
    free(s);
    return h;
}

/* ---- a SHARDED product index's stream: a 16-byte marker chunk ("VKSHARDS", u64 shard count -- the product's own
 * framing, csrc/sharded_index.cc; the reference has one graph per cluster shard and no such stream) followed by the
 * shards' SaveIndex streams back to back.  Splits it into one oracle graph per shard so that a test can search each
 * shard's very graph on the CPU and merge with vko_merge_topk (the role of fanout.cc:162-175). */
#define VKO_MSINK_MAX 16
struct vko_msink {
    size_t dim, M, efC;
    vko_space_t space;
    vko_isa_t isa;
    uint64_t n_shards, cur;
    int started, failed;
    vko_sink *sink;
    vko_hnsw *graphs[VKO_MSINK_MAX];
};
typedef struct vko_msink vko_msink;

vko_msink *vko_msink_new(size_t dim, vko_space_t space, vko_isa_t isa, size_t M, size_t ef_construction) {
    vko_msink *m = (vko_msink *)calloc(1, sizeof(*m));
    m->dim = dim; m->space = space; m->isa = isa; m->M = M; m->efC = ef_construction;
    return m;
}

int vko_msink_write(void *user, const void *data, uint64_t len) {
    vko_msink *m = (vko_msink *)user;
    if (m->failed) return 1;
    if (!m->started) {
        if (len != 16 || memcmp(data, "VKSHARDS", 8) != 0) { vko_set_error("msink: no shard marker chunk"); m->failed = 1; return 1; }
        memcpy(&m->n_shards, (const uint8_t *)data + 8, 8);
        if (m->n_shards == 0 || m->n_shards > VKO_MSINK_MAX) { vko_set_error("msink: shard count out of range"); m->failed = 1; return 1; }
        m->started = 1;
        return 0;
    }
    if (m->cur >= m->n_shards) { vko_set_error("msink: chunk after the last shard"); m->failed = 1; return 1; }
    if (!m->sink) m->sink = vko_sink_new(m->dim, m->space, m->isa, m->M, m->efC);
    if (vko_sink_write(m->sink, data, len)) { m->failed = 1; return 1; }
    if (m->sink->state == 4) {
        m->graphs[m->cur++] = vko_sink_finish(m->sink);
        m->sink = NULL;
    }
    return 0;
}

/* number of complete shard graphs (0 on failure); vko_msink_take hands one over (ownership passes to the caller) */
uint64_t vko_msink_count(vko_msink *m) { return m->failed || m->cur != m->n_shards ? 0 : m->n_shards; }
vko_hnsw *vko_msink_take(vko_msink *m, uint64_t i) {
    if (i >= m->cur) return NULL;
    vko_hnsw *h = m->graphs[i];
    m->graphs[i] = NULL;
    return h;
}
void vko_msink_free(vko_msink *m) {
    if (!m) return;
    if (m->sink) { vko_hnsw *h = vko_sink_finish(m->sink); if (h) vko_hnsw_free(h); }
    for (int i = 0; i < VKO_MSINK_MAX; ++i) if (m->graphs[i]) vko_hnsw_free(m->graphs[i]);
    free(m);
}
