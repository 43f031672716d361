/* vko_internal.h -- ORACLE internals (test infrastructure only, see vko.h). */
#ifndef VKO_INTERNAL_H_
#define VKO_INTERNAL_H_

#include "vko.h"

void vko_set_error(const char *msg);

typedef struct {
    uint64_t *keys;
    uint32_t *vals;
    uint8_t *state; /* 0 empty, 1 used, 2 tombstone */
    size_t cap, n, tomb;
} vko_map;
void vko_map_init(vko_map *m);
void vko_map_free(vko_map *m);
int vko_map_get(const vko_map *m, uint64_t key, uint32_t *val);
void vko_map_put(vko_map *m, uint64_t key, uint32_t val);
int vko_map_del(vko_map *m, uint64_t key);

typedef struct { float d; uint64_t label; } vko_dl;
typedef struct { vko_dl *v; size_t n, cap; } vko_dlheap;
void vko_dlheap_init(vko_dlheap *h);
void vko_dlheap_free(vko_dlheap *h);
void vko_dlheap_push(vko_dlheap *h, float d, uint64_t label);
void vko_dlheap_pop(vko_dlheap *h);
size_t vko_dlheap_drain_ascending(vko_dlheap *h, float *out_dist, uint64_t *out_label);

static inline int vko_allowed(const uint64_t *bits, uint64_t nbits, uint64_t label) {
    if (!bits) return 1;
    if (label >= nbits) return 0;
    return (int)((bits[label >> 6] >> (label & 63)) & 1u);
}

/* emulates cancel::Token polled through BaseCancellationFunctor */
typedef struct { long after; long polls; } vko_cancel;
static inline int vko_cancelled(vko_cancel *c) {
    if (c->after < 0) return 0;
    return c->polls++ >= c->after;
}

#endif
