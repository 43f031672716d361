// hip_virtual.hpp -- TEST INFRASTRUCTURE: hooks of the virtual multi-device HIP runtime (hip_virtual.cc).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

namespace hipv {
struct Access { const void *p; size_t bytes; const char *what; };
void set_device_count(int n);
void set_peer_capable(int a, int b, bool ok);      // what hipDeviceCanAccessPeer(a, b) answers (default: every pair)
std::vector<std::string> take_violations();        // what the model objected to since the last call
uint64_t peer_copies();                            // hipMemcpyPeerAsync calls between two DIFFERENT devices so far
uint64_t kernels();
uint64_t collectives();                            // all-gathers the RCCL model ran
int stream_device(hipStream_t s);
int memory_device(const void *p, size_t n);        // >= 0 device, -1 pinned, -2 plain host, -3 past the end of its block
// a stand-in for a kernel launch: body runs in stream order; every buffer named must be local to the stream's device (or pinned)
void launch(hipStream_t s, const char *name, std::vector<Access> touched, std::function<void()> body);
}  // namespace hipv
