// host_dist.hpp -- host-side f32 distance used while BUILDING the HNSW graph (the reference
// builds on the CPU too: hnswalg.h:1523-1650).  Same arithmetic as the device kernels
// (device_common.hpp): the 16-accumulator / fixed-tree order of SimSIMD's AVX-512 kernels
// (third_party/simsimd/include/simsimd/dot.h:1183-1204, spatial.h:1131-1153) and the hnswlib
// bridge's (float)(1.0 - dot) (third_party/hnswlib/simsimd.h:16-34), so that the graph the
// host builds and the distances the device evaluates agree bit for bit on every host ISA.
#pragma once
#include <stddef.h>

namespace vk {

using host_dist_fn = float (*)(const float *a, const float *b, size_t n);
// picks the widest clone the running CPU supports (avx512f > avx2+fma > generic)
host_dist_fn host_distance_ip();
host_dist_fn host_distance_l2();
const char *host_distance_path();

}  // namespace vk
