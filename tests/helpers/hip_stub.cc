// A host-memory stand-in for the dozen HIP runtime entry points csrc/row_store.cc calls, so that the row store's op
// log (staging chunks, merged write runs, swap-delete moves, label ranges, growth) can run under ASAN / UBSAN on a box
// without a GPU: "device" memory is malloc'd (so every out-of-range copy is a sanitizer report), streams are
// synchronous.  TEST INFRASTRUCTURE: linked only into tests/helpers/san_rowstore_main.cc, never into the product.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

extern "C" {
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(0x10); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t i = 0; i < h; ++i) memmove(static_cast<char *>(d) + i * dp, static_cast<const char *>(s) + i * sp, w);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
}
