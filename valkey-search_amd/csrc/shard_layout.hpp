// shard_layout.hpp -- where the lists of a sharded search travel: shard -> (device group, position) -> send slot -> receive
// offset -> slice of the gathered array.  Pure arithmetic over the index's device list, no HIP: sharded_index.cc builds its
// fan-out on it and tests/helpers/san_sharded_main.cc (check_layout, then whole searches over virtual devices) checks it for 1 to 8
// devices with even and uneven shard counts per device, which no box with one GPU ever exercises.
//
//   devices[s]             the HIP device of shard s (a device may repeat: logical shards)
//   group g                the shards that live on one device, in shard order; groups in order of first appearance, so group 0
//                          is the SERVING device's (devices[0]: queries arrive there, the merged answer is written there)
//   P = max group size     every device contributes P lists to the collective (an all-gather needs equal counts per rank);
//                          a group with fewer shards pads with (+inf, no label) lists
//
// Gather by peer copies (the default):   list of shard s -> entry  s * nk                 of d_all[S][nq][k] on the serving device
// Gather by RCCL all-gather:             list of shard s -> entry  pos(s) * nk            of its device's send buffer [P][nq][k]
//                                        and arrives at            (group(s) * P + pos(s)) * nk  of EVERY rank's receive buffer
//                                        [G][P][nq][k] (rank = group index, ncclAllGather concatenates in rank order);
//                                        the merge reads G * P parts of rank 0's copy.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace vk {

struct ShardLayout {
  std::vector<std::vector<size_t>> groups;   // shards by device, the serving device's group first
  std::vector<int> group_device;             // the device of each group = the RCCL rank order
  std::vector<uint32_t> group_of, pos_of;    // shard -> its group and its place in it
  size_t P = 0;                              // the most shards any device holds

  size_t shards() const { return group_of.size(); }
  size_t G() const { return groups.size(); }

  static ShardLayout from_devices(const std::vector<int> &devices) {
    ShardLayout l;
    const size_t S = devices.size();
    l.group_of.assign(S, 0);
    l.pos_of.assign(S, 0);
    for (size_t s = 0; s < S; ++s) {
      size_t g = 0;
      for (; g < l.groups.size(); ++g)
        if (l.group_device[g] == devices[s]) break;
      if (g == l.groups.size()) {
        l.groups.emplace_back();
        l.group_device.push_back(devices[s]);
      }
      l.group_of[s] = (uint32_t)g;
      l.pos_of[s] = (uint32_t)l.groups[g].size();
      l.groups[g].push_back(s);
      if (l.groups[g].size() > l.P) l.P = l.groups[g].size();
    }
    return l;
  }

  // ---- gather by peer copies: [S][nk] on the serving device
  size_t peer_slice(size_t s, size_t nk) const { return s * nk; }
  size_t peer_parts() const { return shards(); }

  // ---- gather by all-gather
  size_t send_entries(size_t nk) const { return P * nk; }                 // per rank: what ncclAllGather sends (count per rank)
  size_t recv_entries(size_t nk) const { return G() * P * nk; }           // per rank: what it receives
  size_t send_slot(size_t s, size_t nk) const { return (size_t)pos_of[s] * nk; }
  size_t recv_slot(size_t s, size_t nk) const { return ((size_t)group_of[s] * P + pos_of[s]) * nk; }
  size_t rccl_parts() const { return G() * P; }
  // pad slots of group g (positions no shard of that device writes): [pad_begin(g), P)
  size_t pad_begin(size_t g) const { return groups[g].size(); }
  // the stream a device's collective is issued on = the lane of its LAST shard (it waits for the device's other shards)
  size_t collective_lane(size_t g) const { return groups[g].back(); }
};

}  // namespace vk
