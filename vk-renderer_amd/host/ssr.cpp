// ssr.cpp — records the simple SSR pass.  Follows src/ssr.cpp: RGBA8_UNORM target :5-8, depth read
// through a NEAREST sampler with U/W clamp-to-border :21-28, bindings {0 normal, 1 depth, 2 colour,
// 3 SSRParams, 4 material} :55-60, full-screen draw into `out` :62-72.
#include "ssr.hpp"

rendergraph::ImageResourceId create_ssr_tex(rendergraph::RenderGraph &graph, uint32_t w, uint32_t h) {
  const gpu::ImageInfo info {VK_FORMAT_R8G8B8A8_UNORM, VK_IMAGE_ASPECT_COLOR_BIT, w, h};
  return graph.create_image(VK_IMAGE_TYPE_2D, info, VK_IMAGE_TILING_OPTIMAL,
    VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT|VK_IMAGE_USAGE_SAMPLED_BIT|VK_IMAGE_USAGE_STORAGE_BIT);
}

void add_ssr_pass(rendergraph::RenderGraph &graph, rendergraph::ImageResourceId depth, rendergraph::ImageResourceId normal,
  rendergraph::ImageResourceId color, rendergraph::ImageResourceId material, rendergraph::ImageResourceId out, const SSRParams &params)
{
  static_assert(sizeof(SSRParams) == sizeof(vkr_ssr_params), "SSRParams must match the C-ABI");
  const auto sampler = gpu::create_sampler(gpu::DEFAULT_SAMPLER);

  auto depth_sampler_info = gpu::DEFAULT_SAMPLER;
  depth_sampler_info.minFilter = VK_FILTER_NEAREST;
  depth_sampler_info.magFilter = VK_FILTER_NEAREST;
  depth_sampler_info.mipmapMode = VK_SAMPLER_MIPMAP_MODE_NEAREST;
  depth_sampler_info.addressModeU = VK_SAMPLER_ADDRESS_MODE_CLAMP_TO_BORDER;
  depth_sampler_info.addressModeW = VK_SAMPLER_ADDRESS_MODE_CLAMP_TO_BORDER;
  const auto depth_sampler = gpu::create_sampler(depth_sampler_info);

  auto pipeline = gpu::create_graphics_pipeline();
  pipeline.set_program("ssr");
  pipeline.set_registers({});
  pipeline.set_vertex_input({});
  pipeline.set_rendersubpass({false, {graph.get_descriptor(out).format}});

  struct Input { rendergraph::ImageViewId depth, normal, color, material, rt; };

  graph.add_task<Input>("SSR",
    [&](Input &in, rendergraph::RenderGraphBuilder &builder) {
      const auto fs = VK_SHADER_STAGE_FRAGMENT_BIT;
      in.depth = builder.sample_image(depth, fs, VK_IMAGE_ASPECT_DEPTH_BIT);
      in.normal = builder.sample_image(normal, fs);
      in.color = builder.sample_image(color, fs);
      in.material = builder.sample_image(material, fs);
      in.rt = builder.use_color_attachment(out, 0, 0);
    },
    [=](Input &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto block = cmd.allocate_ubo<SSRParams>();
      *block.ptr = params;

      auto set = resources.allocate_set(pipeline, 0);
      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(in.normal), sampler},
        gpu::TextureBinding {1, resources.get_view(in.depth), depth_sampler},
        gpu::TextureBinding {2, resources.get_view(in.color), sampler},
        gpu::UBOBinding {3, cmd.get_ubo_pool(), block},
        gpu::TextureBinding {4, resources.get_view(in.material), sampler});

      const auto extent = resources.get_image(in.rt)->get_extent();
      cmd.set_framebuffer(extent.width, extent.height, {resources.get_image_range(in.rt)});
      cmd.bind_pipeline(pipeline);
      cmd.bind_viewport(0.f, 0.f, float(extent.width), float(extent.height), 0.f, 1.f);
      cmd.bind_scissors(0, 0, extent.width, extent.height);
      cmd.bind_descriptors_graphics(0, {set}, {block.offset});
      cmd.draw(3, 1, 0, 0);
      cmd.end_renderpass();
    });
}
