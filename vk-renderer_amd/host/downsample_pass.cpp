// downsample_pass.cpp — records the Hi-Z build.  Behaviour follows src/downsample_pass.cpp:
// ctor :3-23 (two full-screen programs, depth test ALWAYS + depth write), gbuffer step :25-92
// (size checks and their messages :37-50), one task per further mip :94-131, run :133-143.
#include "downsample_pass.hpp"

#include <algorithm>
#include <stdexcept>
#include <vector>

DownsamplePass::DownsamplePass() : sampler {gpu::create_sampler(gpu::DEFAULT_SAMPLER)} {
  gpu::Registers always_write {};
  always_write.depth_stencil.depthTestEnable = VK_TRUE;
  always_write.depth_stencil.depthCompareOp = VK_COMPARE_OP_ALWAYS;
  always_write.depth_stencil.depthWriteEnable = VK_TRUE;

  for (auto *p : {&downsample_gbuffer, &downsample_depth}) {
    *p = gpu::create_graphics_pipeline();
    p->set_registers(always_write);
    p->set_vertex_input({});
  }
  downsample_gbuffer.set_program("downsample_gbuffer");
  downsample_depth.set_program("depth_mips");
}

void DownsamplePass::run_downsample_gbuff(rendergraph::RenderGraph &graph, rendergraph::ImageResourceId src_normals,
  rendergraph::ImageResourceId src_velocity, rendergraph::ImageResourceId depth, rendergraph::ImageResourceId out_normal,
  rendergraph::ImageResourceId out_velocity)
{
  const auto depth_desc = graph.get_descriptor(depth);
  const auto norm_desc = graph.get_descriptor(out_normal);
  const auto vel_desc = graph.get_descriptor(out_velocity);

  if (depth_desc.mip_levels < 2)
    throw std::runtime_error {"Can't downsample depth texture with 1 mip level"};

  const uint32_t half_w = std::max(1u, depth_desc.width/2), half_h = std::max(1u, depth_desc.height/2);
  const bool same = half_w == norm_desc.width && half_h == norm_desc.height && vel_desc.width == norm_desc.width && vel_desc.height == norm_desc.height;
  if (!same)
    throw std::runtime_error {"Output textures have different sizes"};

  downsample_gbuffer.set_rendersubpass({true, {norm_desc.format, vel_desc.format, depth_desc.format}});

  struct Input {
    rendergraph::ImageViewId gbuffer_depth, gbuffer_normal, gbuffer_velocity;
    rendergraph::ImageViewId out_depth, out_normal, out_velocity;
  };

  graph.add_task<Input>("DownsampleGbuffer",
    [&](Input &in, rendergraph::RenderGraphBuilder &builder) {
      const auto frag = VK_SHADER_STAGE_FRAGMENT_BIT;
      in.gbuffer_depth = builder.sample_image(depth, frag, VK_IMAGE_ASPECT_DEPTH_BIT, 0, 1, 0, 1);
      in.gbuffer_normal = builder.sample_image(src_normals, frag, VK_IMAGE_ASPECT_COLOR_BIT, 0, 1, 0, 1);
      in.gbuffer_velocity = builder.sample_image(src_velocity, frag, VK_IMAGE_ASPECT_COLOR_BIT, 0, 1, 0, 1);
      in.out_depth = builder.use_depth_attachment(depth, 1, 0);
      in.out_normal = builder.use_color_attachment(out_normal, 0, 0);
      in.out_velocity = builder.use_color_attachment(out_velocity, 0, 0);
    },
    [=](Input &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(downsample_gbuffer, 0);
      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(in.gbuffer_depth), sampler},
        gpu::TextureBinding {1, resources.get_view(in.gbuffer_normal), sampler},
        gpu::TextureBinding {2, resources.get_view(in.gbuffer_velocity), sampler});

      cmd.set_framebuffer(half_w, half_h, {
        resources.get_image_range(in.out_normal),
        resources.get_image_range(in.out_velocity),
        resources.get_image_range(in.out_depth)});
      cmd.bind_pipeline(downsample_gbuffer);
      cmd.bind_descriptors_graphics(0, {set});
      cmd.bind_viewport(0.f, 0.f, float(half_w), float(half_h), 0.f, 1.f);
      cmd.bind_scissors(0, 0, half_w, half_h);
      cmd.draw(3, 1, 0, 0);
      cmd.end_renderpass();
    });
}

// downsample_pass.cpp:94-131 records one "DownsampleDepth" draw per mip (L-2 dependent passes).
// MI355X-first: the same chain is one task whose attachments are all remaining mips; the bound
// program reduces five levels per workgroup through LDS.  Each mip is still the 2x2 min of its
// parent with extent max(1, W >> i) (:118-120).
void DownsamplePass::run_downsample_depth(rendergraph::RenderGraph &graph, rendergraph::ImageResourceId depth, uint32_t src_mip) {
  const auto desc = graph.get_descriptor(depth);
  if (src_mip + 1 >= desc.mip_levels) return;
  downsample_depth.set_rendersubpass({true, {desc.format}});

  struct Input {
    rendergraph::ImageViewId depth_tex;
    std::vector<rendergraph::ImageViewId> depth_rt;
  };

  graph.add_task<Input>("DownsampleDepth",
    [&](Input &in, rendergraph::RenderGraphBuilder &builder) {
      in.depth_tex = builder.sample_image(depth, VK_SHADER_STAGE_FRAGMENT_BIT, VK_IMAGE_ASPECT_DEPTH_BIT, src_mip, 1, 0, 1);
      for (uint32_t mip = src_mip + 1; mip < desc.mip_levels; mip++)
        in.depth_rt.push_back(builder.use_depth_attachment(depth, mip, 0));
    },
    [=](Input &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(downsample_depth, 0);
      gpu::write_set(set, gpu::TextureBinding {0, resources.get_view(in.depth_tex), sampler});

      std::vector<gpu::ImageViewObject> targets;
      for (const auto &rt : in.depth_rt) targets.push_back(resources.get_image_range(rt));
      const uint32_t w = std::max(desc.width >> (src_mip + 1), 1u), h = std::max(desc.height >> (src_mip + 1), 1u);
      cmd.set_framebuffer(w, h, targets);
      cmd.bind_pipeline(downsample_depth);
      cmd.bind_descriptors_graphics(0, {set});
      cmd.bind_viewport(0.f, 0.f, float(w), float(h), 0.f, 1.f);
      cmd.bind_scissors(0, 0, w, h);
      cmd.draw(3, 1, 0, 0);
      cmd.end_renderpass();
    });
}

void DownsamplePass::run(rendergraph::RenderGraph &graph, rendergraph::ImageResourceId src_normals, rendergraph::ImageResourceId src_velocity,
  rendergraph::ImageResourceId depth, rendergraph::ImageResourceId out_normals, rendergraph::ImageResourceId out_velocity)
{
  run_downsample_gbuff(graph, src_normals, src_velocity, depth, out_normals, out_velocity);
  run_downsample_depth(graph, depth, 1);
}
