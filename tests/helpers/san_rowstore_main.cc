// Sanitizer harness (ASAN + UBSAN) for csrc/row_store.cc over tests/helpers/hip_stub.cc: random writer phases -- single
// staged writes (contiguous runs and scattered slots), swap-delete moves (bruteforce.h:92-113: the last row moves into
// the hole), label-only updates, bulk writes, growth -- each published by flush() and compared with a plain model.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../valkey-search_amd/csrc/row_store.hpp"

int main(int argc, char **argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 60;
  int bad = 0;
  for (int bf16 = 0; bf16 < 2; ++bf16)
    for (uint32_t dim : {5u, 64u, 100u}) {
      vk::RowStore st(0, dim, bf16 != 0);
      std::mt19937 g(dim * 2 + bf16);
      std::normal_distribution<float> nd;
      std::vector<std::vector<float>> model;      // rows as the store should hold them
      std::vector<uint64_t> mlab;
      auto rnd_row = [&] {
        std::vector<float> r(dim);
        for (float &v : r) {
          v = nd(g);
          if (bf16) {                             // values that survive the RNE rounding unchanged
            uint32_t u;
            memcpy(&u, &v, 4);
            u &= 0xFFFF0000u;
            memcpy(&v, &u, 4);
          }
        }
        return r;
      };
      uint64_t next_label = 1;
      for (int ph = 0; ph < phases; ++ph) {
        const int nops = 1 + (int)(g() % 300);
        for (int o = 0; o < nops; ++o) {
          const unsigned kind = g() % 10;
          if (kind < 6 || model.empty()) {                         // append (runs of these merge into one copy)
            auto r = rnd_row();
            if (!st.stage_write((uint32_t)model.size(), r.data(), next_label).ok()) bad++;
            model.push_back(r);
            mlab.push_back(next_label++);
          } else if (kind < 8) {                                   // overwrite in place
            const uint32_t s = g() % model.size();
            auto r = rnd_row();
            if (!st.stage_write(s, r.data(), mlab[s]).ok()) bad++;
            model[s] = r;
          } else if (kind == 8) {                                  // swap-delete
            const uint32_t s = g() % model.size(), last = (uint32_t)model.size() - 1;
            if (s != last) {
              st.stage_move(s, last, mlab[last]);
              model[s] = model[last];
              mlab[s] = mlab[last];
            }
            model.pop_back();
            mlab.pop_back();
          } else {                                                 // relabel
            const uint32_t s = g() % model.size();
            mlab[s] = next_label++;
            st.stage_label(s, mlab[s]);
          }
        }
        if (!bf16 && ph % 9 == 4) {                                // a bulk load behind what is there
          const size_t n = 1 + g() % 2000;
          std::vector<float> rows(n * dim);
          std::vector<uint64_t> labs(n);
          for (size_t i = 0; i < n; ++i) {
            auto r = rnd_row();
            memcpy(rows.data() + i * dim, r.data(), dim * 4);
            labs[i] = next_label++;
            model.push_back(r);
            mlab.push_back(labs[i]);
          }
          if (!st.bulk_write((uint32_t)(model.size() - n), rows.data(), n, labs.data()).ok()) bad++;
        }
        if (!st.flush().ok()) bad++;
        if (st.dirty() || st.staged_ops()) bad++;
        // every row and label against the model (the stub's device memory is host memory)
        const char *dr = static_cast<const char *>(st.d_rows());
        for (size_t s = 0; s < model.size(); ++s) {
          std::vector<float> got(dim);
          if (!st.read_row((uint32_t)s, got.data()).ok() || got != model[s]) { bad++; break; }
          if (st.d_labels()[s] != mlab[s] || st.host_labels()[s] != mlab[s]) { bad++; break; }
          // the padding of the row is zero (it IS SimSIMD's masked tail)
          for (uint32_t e = dim; e < st.stride_f(); ++e) {
            const char *p = dr + s * st.row_bytes() + (size_t)e * (bf16 ? 2 : 4);
            if (p[0] || p[1] || (!bf16 && (p[2] || p[3]))) { bad++; break; }
          }
        }
        if (st.alloc_rows() < model.size()) bad++;
      }
    }
  printf("bad=%d\n", bad);
  return bad ? 1 : 0;
}
