// pass_recorder.hpp — declarative recording of one rendergraph task.
//
// Every pass of this path has the same shape: declare which views it samples / stores / renders to,
// then at execution time write those views into a descriptor set in binding order, attach a uniform
// block and push constants, and issue one dispatch (or one full-screen triangle).  Instead of
// spelling the two callbacks of RenderGraph::add_task out per pass, a pass lists its bindings once:
//
//   rec::compute(graph, "TAA", pipeline,
//                {rec::sampled(0, history, s), ..., rec::storage(5, target), rec::uniform(6, consts)},
//                rec::no_push(), rec::Grid{target, 8, 8, rec::Ceil});
//
// The recorder declares the usages on the builder (so the graph's usage checks and error messages
// apply unchanged) and replays the list inside the run callback, which captures everything by value.
#ifndef PASS_RECORDER_HPP_INCLUDED
#define PASS_RECORDER_HPP_INCLUDED

#include <cstring>
#include <string>
#include <vector>

#include "rendergraph/rendergraph.hpp"

namespace rec {

using rendergraph::BufferResourceId;
using rendergraph::ImageResourceId;
using rendergraph::ImageViewId;

struct Binding {
  enum Kind { Sampled, SampledWhole, Storage, StorageArray, Color, Depth, UniformBlock, UniformBuffer, StorageBuffer, GraphUniformBuffer } kind;
  uint32_t slot = 0;
  ImageResourceId image;
  BufferResourceId graph_buffer;
  gpu::BufferPtr buffer;
  VkImageAspectFlags aspect = 0;
  uint32_t base_mip = 0, mip_count = 1;  // Sampled: view range; Storage / Color / Depth: base_mip = the mip
  VkSampler sampler = nullptr;
  bool writable = true;
  std::vector<uint8_t> bytes;  // UniformBlock payload
};

// texture(): all mips and layers of the image
inline Binding sampled(uint32_t slot, ImageResourceId img, VkSampler s, VkImageAspectFlags aspect = 0) {
  Binding b; b.kind = Binding::SampledWhole; b.slot = slot; b.image = img; b.sampler = s; b.aspect = aspect; return b;
}
// texture() through a view of mips [base_mip, base_mip + count)
inline Binding sampled_mips(uint32_t slot, ImageResourceId img, VkSampler s, VkImageAspectFlags aspect, uint32_t base_mip, uint32_t count) {
  Binding b; b.kind = Binding::Sampled; b.slot = slot; b.image = img; b.sampler = s; b.aspect = aspect; b.base_mip = base_mip; b.mip_count = count; return b;
}
inline Binding storage(uint32_t slot, ImageResourceId img, uint32_t mip = 0) {
  Binding b; b.kind = Binding::Storage; b.slot = slot; b.image = img; b.base_mip = mip; return b;
}
inline Binding storage_array(uint32_t slot, ImageResourceId img) {
  Binding b; b.kind = Binding::StorageArray; b.slot = slot; b.image = img; return b;
}
inline Binding color_target(ImageResourceId img, uint32_t mip = 0) {
  Binding b; b.kind = Binding::Color; b.image = img; b.base_mip = mip; return b;
}
inline Binding depth_target(ImageResourceId img, uint32_t mip = 0) {
  Binding b; b.kind = Binding::Depth; b.image = img; b.base_mip = mip; return b;
}
// a block of the per-frame uniform ring holding a copy of `value`
template <typename T> Binding uniform(uint32_t slot, const T &value) {
  Binding b; b.kind = Binding::UniformBlock; b.slot = slot; b.bytes.resize(sizeof(T)); std::memcpy(b.bytes.data(), &value, sizeof(T)); return b;
}
inline Binding uniform_buffer(uint32_t slot, const gpu::BufferPtr &buf) {
  Binding b; b.kind = Binding::UniformBuffer; b.slot = slot; b.buffer = buf; return b;
}
inline Binding uniform_buffer(uint32_t slot, BufferResourceId buf) {
  Binding b; b.kind = Binding::GraphUniformBuffer; b.slot = slot; b.graph_buffer = buf; return b;
}
inline Binding storage_buffer(uint32_t slot, BufferResourceId buf, bool writable = true) {
  Binding b; b.kind = Binding::StorageBuffer; b.slot = slot; b.graph_buffer = buf; b.writable = writable; return b;
}

using Push = std::vector<uint8_t>;
inline Push no_push() { return {}; }
template <typename T> Push push(const T &value) { Push p(sizeof(T)); std::memcpy(p.data(), &value, sizeof(T)); return p; }

enum Rounding { Floor, Ceil };
// group counts = extent of `sized_by` (mip 0) divided by the workgroup size
struct Grid { ImageResourceId sized_by; uint32_t group_w, group_h; Rounding rounding; };

// what the run callback needs of a binding once the builder has turned images into views
struct Bound { Binding b; ImageViewId view; };

inline std::vector<Bound> declare(const std::vector<Binding> &bindings, rendergraph::RenderGraphBuilder &builder, VkShaderStageFlags stage) {
  std::vector<Bound> out;
  for (const auto &b : bindings) {
    Bound r {b, {}};
    switch (b.kind) {
      case Binding::SampledWhole: r.view = builder.sample_image(b.image, stage, b.aspect); break;
      case Binding::Sampled: r.view = builder.sample_image(b.image, stage, b.aspect, b.base_mip, b.mip_count, 0, 1); break;
      case Binding::Storage: r.view = builder.use_storage_image(b.image, stage, b.base_mip, 0); break;
      case Binding::StorageArray: r.view = builder.use_storage_image_array(b.image, stage); break;
      case Binding::Color: r.view = builder.use_color_attachment(b.image, b.base_mip, 0); break;
      case Binding::Depth: r.view = builder.use_depth_attachment(b.image, b.base_mip, 0); break;
      case Binding::StorageBuffer: builder.use_storage_buffer(b.graph_buffer, stage, b.writable); break;
      case Binding::GraphUniformBuffer: builder.use_uniform_buffer(b.graph_buffer, stage); break;
      default: break;
    }
    out.push_back(r);
  }
  return out;
}

// writes every descriptor binding of `bound` into a fresh set; returns it
inline VkDescriptorSet write(const std::vector<Bound> &bound, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
  VkDescriptorSet set = cmd.allocate_set();
  for (const auto &r : bound) {
    const Binding &b = r.b;
    switch (b.kind) {
      case Binding::SampledWhole:
      case Binding::Sampled: gpu::write_set(set, gpu::TextureBinding {b.slot, resources.get_view(r.view), b.sampler}); break;
      case Binding::Storage:
      case Binding::StorageArray: gpu::write_set(set, gpu::StorageTextureBinding {b.slot, resources.get_view(r.view)}); break;
      case Binding::UniformBlock: {
        auto blk = cmd.get_ubo_pool().allocate_bytes(b.bytes.size());
        std::memcpy(blk.ptr, b.bytes.data(), b.bytes.size());
        gpu::write_set(set, gpu::UBOBinding {b.slot, blk.ptr, b.bytes.size()});
        break;
      }
      case Binding::UniformBuffer: gpu::write_set(set, gpu::UBOBinding {b.slot, b.buffer}); break;
      case Binding::GraphUniformBuffer: gpu::write_set(set, gpu::UBOBinding {b.slot, resources.get_buffer(b.graph_buffer)}); break;
      case Binding::StorageBuffer: gpu::write_set(set, gpu::SSBOBinding {b.slot, resources.get_buffer(b.graph_buffer)}); break;
      default: break;
    }
  }
  return set;
}

// one compute dispatch
inline void compute(rendergraph::RenderGraph &graph, const std::string &task, const gpu::ComputePipeline &pipeline,
                    const std::vector<Binding> &bindings, const Push &push_constants, const Grid &grid)
{
  struct Data { std::vector<Bound> bound; };
  graph.add_task<Data>(task,
    [&](Data &d, rendergraph::RenderGraphBuilder &builder) { d.bound = declare(bindings, builder, VK_SHADER_STAGE_COMPUTE_BIT); },
    [=](Data &d, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      VkDescriptorSet set = write(d.bound, resources, cmd);
      cmd.bind_pipeline(pipeline);
      cmd.bind_descriptors_compute(0, {set});
      if (!push_constants.empty()) cmd.push_constants_compute(0, uint32_t(push_constants.size()), push_constants.data());
      const auto extent = resources.get_image(grid.sized_by)->get_extent();
      const uint32_t round_w = grid.rounding == Ceil ? grid.group_w - 1 : 0, round_h = grid.rounding == Ceil ? grid.group_h - 1 : 0;
      cmd.dispatch((extent.width + round_w)/grid.group_w, (extent.height + round_h)/grid.group_h, 1);
    });
}

// one full-screen triangle into the Color / Depth bindings (in list order: colours first, depth last),
// viewport = (width, height)
inline void fullscreen(rendergraph::RenderGraph &graph, const std::string &task, const gpu::GraphicsPipeline &pipeline,
                       const std::vector<Binding> &bindings, const Push &push_constants, uint32_t width, uint32_t height)
{
  struct Data { std::vector<Bound> bound; };
  graph.add_task<Data>(task,
    [&](Data &d, rendergraph::RenderGraphBuilder &builder) { d.bound = declare(bindings, builder, VK_SHADER_STAGE_FRAGMENT_BIT); },
    [=](Data &d, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      VkDescriptorSet set = write(d.bound, resources, cmd);
      std::vector<gpu::ImageViewObject> targets;
      for (const auto &r : d.bound)
        if (r.b.kind == Binding::Color) targets.push_back(resources.get_image_range(r.view));
      for (const auto &r : d.bound)
        if (r.b.kind == Binding::Depth) targets.push_back(resources.get_image_range(r.view));
      cmd.set_framebuffer(width, height, targets);
      cmd.bind_pipeline(pipeline);
      cmd.bind_viewport(0.f, 0.f, float(width), float(height), 0.f, 1.f);
      cmd.bind_scissors(0, 0, width, height);
      cmd.bind_descriptors_graphics(0, {set});
      if (!push_constants.empty()) cmd.push_constants_graphics(VK_SHADER_STAGE_FRAGMENT_BIT, 0, uint32_t(push_constants.size()), push_constants.data());
      cmd.draw(3, 1, 0, 0);
      cmd.end_renderpass();
    });
}

}  // namespace rec

#endif
