// gpu_transfer.cpp — see gpu_transfer.hpp.  The reference queues the write in a staging ring and records a
// vkCmdCopyBuffer at the next process_requests() (gpu_transfer.cpp:60-116); here a graph buffer keeps a host shadow
// that is uploaded on the graph's stream before the first task that binds it, so the write lands in the shadow at once.
#include "gpu_transfer.hpp"

#include <cstring>
#include <stdexcept>

namespace gpu_transfer {

static rendergraph::RenderGraph *g_graph = nullptr;

void init(const rendergraph::RenderGraph &graph) { g_graph = const_cast<rendergraph::RenderGraph *>(&graph); }
void close() { g_graph = nullptr; }
void process_requests(rendergraph::RenderGraph &) {}

void write_buffer(rendergraph::BufferResourceId id, uint64_t offset, uint64_t size, const void *data) {
  if (!g_graph) throw std::runtime_error {"gpu_transfer::write_buffer before gpu_transfer::init"};
  if (size > MAX_TRANSFER_SIZE) throw std::runtime_error {"Transfer request is too big"};
  auto &buf = g_graph->get_buffer(id);
  if (offset + size > buf->get_size()) throw std::runtime_error {"gpu_transfer::write_buffer: range outside the buffer"};
  std::memcpy(static_cast<uint8_t *>(buf->get_mapped_ptr()) + offset, data, size);
}

}  // namespace gpu_transfer
