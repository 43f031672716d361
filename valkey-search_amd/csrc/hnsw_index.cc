// placeholder until the HNSW mirror lands
#include "index.hpp"
namespace vk {
Status create_hnsw(const vk_index_params &, std::unique_ptr<Index> *) { return Status::Err(VK_ERR_INTERNAL, "HNSW not built yet"); }
Status load_hnsw(const vk_index_params &, vk_read_chunk_fn, void *, std::unique_ptr<Index> *) { return Status::Err(VK_ERR_INTERNAL, "HNSW not built yet"); }
}
