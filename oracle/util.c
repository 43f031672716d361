/*
 * util.c -- ORACLE (test infrastructure only, see vko.h).
 *
 * Small containers plus restatements of the libstdc++ pieces whose exact
 * behaviour shapes hnswlib's results:
 *   - std::push_heap / std::pop_heap as used by std::priority_queue
 *     (bits/stl_heap.h __push_heap / __adjust_heap): decides which of two
 *     equal-distance entries surfaces first under CompareByFirst
 *     (hnswalg.h:202-208).
 *   - std::default_random_engine == minstd_rand0 and
 *     std::uniform_real_distribution via generate_canonical (hnswalg.h:243-247,
 *     :1355).
 * tests/test_oracle_stdlib.py pins both against the real libstdc++ by compiling
 * a tiny C++ program that prints traces.
 */
#include "vko_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[256];
const char *vko_last_error(void) { return g_err; }
void vko_set_error(const char *msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

/* ---- CompareByFirst max-heap, libstdc++ order ------------------------------- */
void vko_heap_init(vko_heap *h) { h->v = 0; h->n = 0; h->cap = 0; }
void vko_heap_free(vko_heap *h) { free(h->v); h->v = 0; h->n = h->cap = 0; }

static void heap_sift_up(vko_pair *first, size_t hole, size_t top, vko_pair value) {
    /* __push_heap: comp(parent, value) == parent.d < value.d */
    while (hole > top) {
        size_t parent = (hole - 1) / 2;
        if (!(first[parent].d < value.d)) break;
        first[hole] = first[parent];
        hole = parent;
    }
    first[hole] = value;
}

void vko_heap_push(vko_heap *h, float d, uint32_t id) {
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 64;
        h->v = (vko_pair *)realloc(h->v, h->cap * sizeof(vko_pair));
    }
    vko_pair value = {d, id};
    h->n++;
    heap_sift_up(h->v, h->n - 1, 0, value);
}

void vko_heap_pop(vko_heap *h) {
    /* pop_heap: value = last; last = first; __adjust_heap(first,0,len-1,value) */
    if (h->n == 0) return;
    if (h->n == 1) { h->n = 0; return; }
    vko_pair *first = h->v;
    vko_pair value = first[h->n - 1];
    size_t len = h->n - 1;
    size_t hole = 0, second = 0;
    while (len > 0 && second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (first[second].d < first[second - 1].d) second--;
        first[hole] = first[second];
        hole = second;
    }
    if ((len & 1) == 0 && len >= 2 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        first[hole] = first[second - 1];
        hole = second - 1;
    }
    heap_sift_up(first, hole, 0, value);
    h->n = len;
}

/* ---- minstd_rand0 + generate_canonical --------------------------------------- */
void vko_minstd0_seed(vko_minstd0 *g, uint32_t seed) {
    /* linear_congruential_engine::seed: x = seed mod m, 0 -> 1 (c == 0) */
    uint32_t x = seed % 2147483647u;
    g->x = x == 0 ? 1u : x;
}
uint32_t vko_minstd0_next(vko_minstd0 *g) {
    g->x = (uint32_t)(((uint64_t)g->x * 16807u) % 2147483647u);
    return g->x;
}
/* generate_canonical<double,53> (bits/random.tcc): R = max-min+1 = 2147483646 (long
 * double), m = max(1, (53 + floor(log2 R) - 1) / floor(log2 R)) = 2 draws; the sum and
 * the scale are accumulated in the RESULT type: sum += T(x - min) * tmp; tmp *= R;
 * a result >= 1 is replaced by nextafter(1,0) */
double vko_uniform01_double(vko_minstd0 *g) {
    const long double R = 2147483646.0L;
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; ++k) {
        sum += (double)(vko_minstd0_next(g) - 1u) * tmp;
        tmp = (double)((long double)tmp * R);
    }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}
/* generate_canonical<float,24>: m = (24 + 30 - 1) / 30 = 1 draw */
float vko_uniform01_float(vko_minstd0 *g) {
    const long double R = 2147483646.0L;
    float sum = 0.0f, tmp = 1.0f;
    sum += (float)(vko_minstd0_next(g) - 1u) * tmp;
    tmp = (float)((long double)tmp * R);
    float ret = sum / tmp;
    if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
    return ret;
}
/* hnswalg.h:243-247 */
int vko_random_level(vko_minstd0 *g, double reverse_size) {
    double r = -log(vko_uniform01_double(g)) * reverse_size;
    return (int)r;
}

/* ---- u64 -> u32 open-addressing map ------------------------------------------ */
static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
void vko_map_init(vko_map *m) {
    m->cap = 64; m->n = 0; m->tomb = 0;
    m->keys = (uint64_t *)calloc(m->cap, sizeof(uint64_t));
    m->vals = (uint32_t *)calloc(m->cap, sizeof(uint32_t));
    m->state = (uint8_t *)calloc(m->cap, 1);
}
void vko_map_free(vko_map *m) { free(m->keys); free(m->vals); free(m->state); }
static void map_rehash(vko_map *m, size_t ncap) {
    vko_map o = *m;
    m->cap = ncap; m->n = 0; m->tomb = 0;
    m->keys = (uint64_t *)calloc(ncap, sizeof(uint64_t));
    m->vals = (uint32_t *)calloc(ncap, sizeof(uint32_t));
    m->state = (uint8_t *)calloc(ncap, 1);
    for (size_t i = 0; i < o.cap; ++i)
        if (o.state[i] == 1) vko_map_put(m, o.keys[i], o.vals[i]);
    free(o.keys); free(o.vals); free(o.state);
}
int vko_map_get(const vko_map *m, uint64_t key, uint32_t *val) {
    size_t i = mix64(key) & (m->cap - 1);
    while (m->state[i]) {
        if (m->state[i] == 1 && m->keys[i] == key) { if (val) *val = m->vals[i]; return 1; }
        i = (i + 1) & (m->cap - 1);
    }
    return 0;
}
void vko_map_put(vko_map *m, uint64_t key, uint32_t val) {
    if ((m->n + m->tomb + 1) * 2 > m->cap) map_rehash(m, (m->n + 1) * 4 > m->cap ? m->cap * 2 : m->cap);
    size_t i = mix64(key) & (m->cap - 1);
    long first_tomb = -1;
    while (m->state[i]) {
        if (m->state[i] == 1 && m->keys[i] == key) { m->vals[i] = val; return; }
        if (m->state[i] == 2 && first_tomb < 0) first_tomb = (long)i;
        i = (i + 1) & (m->cap - 1);
    }
    if (first_tomb >= 0) { i = (size_t)first_tomb; m->tomb--; }
    m->state[i] = 1; m->keys[i] = key; m->vals[i] = val; m->n++;
}
int vko_map_del(vko_map *m, uint64_t key) {
    size_t i = mix64(key) & (m->cap - 1);
    while (m->state[i]) {
        if (m->state[i] == 1 && m->keys[i] == key) { m->state[i] = 2; m->n--; m->tomb++; return 1; }
        i = (i + 1) & (m->cap - 1);
    }
    return 0;
}

/* ---- (dist,label) lexicographic max-heap == priority_queue<pair<float,size_t>> */
static int dl_less(vko_dl a, vko_dl b) { /* std::pair operator< */
    return a.d < b.d || (!(b.d < a.d) && a.label < b.label);
}
void vko_dlheap_init(vko_dlheap *h) { h->v = 0; h->n = h->cap = 0; }
void vko_dlheap_free(vko_dlheap *h) { free(h->v); h->v = 0; h->n = h->cap = 0; }
void vko_dlheap_push(vko_dlheap *h, float d, uint64_t label) {
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 32;
        h->v = (vko_dl *)realloc(h->v, h->cap * sizeof(vko_dl));
    }
    vko_dl value = {d, label};
    size_t hole = h->n++;
    while (hole > 0) {
        size_t parent = (hole - 1) / 2;
        if (!dl_less(h->v[parent], value)) break;
        h->v[hole] = h->v[parent];
        hole = parent;
    }
    h->v[hole] = value;
}
void vko_dlheap_pop(vko_dlheap *h) {
    if (h->n == 0) return;
    if (h->n == 1) { h->n = 0; return; }
    vko_dl *first = h->v;
    vko_dl value = first[h->n - 1];
    size_t len = h->n - 1, hole = 0, second = 0;
    while (len > 0 && second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (dl_less(first[second], first[second - 1])) second--;
        first[hole] = first[second];
        hole = second;
    }
    if ((len & 1) == 0 && len >= 2 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        first[hole] = first[second - 1];
        hole = second - 1;
    }
    while (hole > 0) {
        size_t parent = (hole - 1) / 2;
        if (!dl_less(first[parent], value)) break;
        first[hole] = first[parent];
        hole = parent;
    }
    first[hole] = value;
    h->n = len;
}
/* drain max-first into ascending arrays (CreateReply: pop then reverse) */
size_t vko_dlheap_drain_ascending(vko_dlheap *h, float *out_dist, uint64_t *out_label) {
    size_t n = h->n;
    for (size_t i = n; i-- > 0;) {
        out_dist[i] = h->v[0].d;
        out_label[i] = h->v[0].label;
        vko_dlheap_pop(h);
    }
    return n;
}
