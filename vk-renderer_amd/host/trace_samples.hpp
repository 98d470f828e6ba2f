// trace_samples.hpp — the debug sample heat-map of src/trace_samples.hpp:8-33 (imageAtomicAdd counters, enabled by
// GTAO_TRACE_SAMPLES, off in gtao.hpp:8; SURVEY.md section 2 #18: out of scope).  The type exists so that gtao.cpp,
// which includes this header unconditionally, compiles unchanged; the image is created, never written.
#ifndef VKR_HOST_TRACE_SAMPLES_HPP_INCLUDED
#define VKR_HOST_TRACE_SAMPLES_HPP_INCLUDED
#include <memory>
#include "rendergraph/rendergraph.hpp"

struct SamplesMarker {
  static void init(rendergraph::RenderGraph &graph, uint32_t w, uint32_t h) {
    holder().reset(new SamplesMarker {});
    gpu::ImageInfo info {VK_FORMAT_R32_UINT, VK_IMAGE_ASPECT_COLOR_BIT, w, h};
    holder()->handle = graph.create_image(VK_IMAGE_TYPE_2D, info, VK_IMAGE_TILING_OPTIMAL,
                                          VK_IMAGE_USAGE_STORAGE_BIT|VK_IMAGE_USAGE_SAMPLED_BIT|VK_IMAGE_USAGE_TRANSFER_DST_BIT);
  }
  static void clear(rendergraph::RenderGraph &) {}
  static rendergraph::ImageResourceId get_image() { return holder()->handle; }

private:
  rendergraph::ImageResourceId handle;
  static std::unique_ptr<SamplesMarker> &holder() { static std::unique_ptr<SamplesMarker> p; return p; }
};

#endif
