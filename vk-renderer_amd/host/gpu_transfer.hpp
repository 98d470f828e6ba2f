// gpu_transfer.hpp — interface of src/gpu_transfer.hpp:6-15 (the reference's per-frame 1 MiB staging ring ->
// vkCmdCopyBuffer, SURVEY.md section 2 #17: out of scope, hipMemcpyAsync covers it).  Kept so that pass sources that
// upload constants through it (defered_shading.cpp:44) compile unchanged: write_buffer() copies straight into the
// buffer on the graph's stream, ordered before every task recorded afterwards.
#ifndef VKR_HOST_GPU_TRANSFER_HPP_INCLUDED
#define VKR_HOST_GPU_TRANSFER_HPP_INCLUDED
#include "rendergraph/rendergraph.hpp"

namespace gpu_transfer {

constexpr uint64_t MAX_TRANSFER_SIZE = (1 << 20);  // 1 MiB, gpu_transfer.hpp:8

void init(const rendergraph::RenderGraph &graph);
void close();
void process_requests(rendergraph::RenderGraph &graph);
void write_buffer(rendergraph::BufferResourceId id, uint64_t offset, uint64_t size, const void *data);

}  // namespace gpu_transfer

#endif
