#include "synthetic_gbuffer.hpp"

#include <cstring>

SyntheticGbuffer::SyntheticGbuffer(uint32_t s) : seed {s} {
  pipeline = gpu::create_graphics_pipeline();
  pipeline.set_program("synthetic_gbuffer");
  pipeline.set_vertex_input({});
}

static vkr_synth_params make_params(const glm::mat4 &camera, const glm::mat4 &mvp, const glm::mat4 &prev_mvp, const glm::vec4 &fazz, uint32_t seed, uint32_t flags) {
  vkr_synth_params p {};
  const glm::mat4 c2w = glm::inverse(camera);
  std::memcpy(p.camera_to_world.m, &c2w, sizeof(float) * 16);
  std::memcpy(p.prev_mvp.m, &prev_mvp, sizeof(float) * 16);
  std::memcpy(p.mvp.m, &mvp, sizeof(float) * 16);
  p.fovy = fazz.x; p.aspect = fazz.y; p.znear = fazz.z; p.zfar = fazz.w;
  p.seed = seed;
  p.flags = flags;
  return p;
}

void SyntheticGbuffer::draw_taa(rendergraph::RenderGraph &graph, const Gbuffer &gbuffer, const DrawTAAParams &params) {
  struct Data { rendergraph::ImageViewId albedo, normal, material, velocity, depth; };
  const vkr_synth_params consts = make_params(params.camera, params.mvp, params.prev_mvp, params.fovy_aspect_znear_zfar, seed, 0);
  const uint32_t w = gbuffer.w, h = gbuffer.h;

  graph.add_task<Data>("GbufferPass",
    [&](Data &d, rendergraph::RenderGraphBuilder &builder) {
      d.albedo = builder.use_color_attachment(gbuffer.albedo, 0, 0);
      d.normal = builder.use_color_attachment(gbuffer.normal, 0, 0);
      d.material = builder.use_color_attachment(gbuffer.material, 0, 0);
      d.velocity = builder.use_color_attachment(gbuffer.velocity_vectors, 0, 0);
      d.depth = builder.use_depth_attachment(gbuffer.depth, 0, 0);
    },
    [=](Data &d, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto blk = cmd.allocate_ubo<vkr_synth_params>();
      *blk.ptr = consts;
      auto set = resources.allocate_set(pipeline, 0);
      gpu::write_set(set, gpu::UBOBinding {0, cmd.get_ubo_pool(), blk});
      cmd.set_framebuffer(w, h, {
        resources.get_image_range(d.albedo), resources.get_image_range(d.normal), resources.get_image_range(d.material),
        resources.get_image_range(d.velocity), resources.get_image_range(d.depth)});
      cmd.bind_pipeline(pipeline);
      cmd.bind_descriptors_graphics(0, {set}, {blk.offset});
      cmd.draw(3, 1, 0, 0);
      cmd.end_renderpass();
    });
}

void SyntheticGbuffer::draw_depth(rendergraph::RenderGraph &graph, rendergraph::ImageResourceId depth_target, const glm::mat4 &camera,
  const glm::mat4 &mvp, const glm::vec4 &fazz)
{
  struct Data { rendergraph::ImageViewId depth; };
  const vkr_synth_params consts = make_params(camera, mvp, mvp, fazz, seed, VKR_SYNTH_DEPTH_ONLY);
  const auto desc = graph.get_descriptor(depth_target);

  graph.add_task<Data>("GbufferDepthOnly",
    [&](Data &d, rendergraph::RenderGraphBuilder &builder) {
      d.depth = builder.use_depth_attachment(depth_target, 0, 0);
    },
    [=](Data &d, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto blk = cmd.allocate_ubo<vkr_synth_params>();
      *blk.ptr = consts;
      auto set = resources.allocate_set(pipeline, 0);
      gpu::write_set(set, gpu::UBOBinding {0, cmd.get_ubo_pool(), blk});
      cmd.set_framebuffer(desc.width, desc.height, {resources.get_image_range(d.depth)});
      cmd.bind_pipeline(pipeline);
      cmd.bind_descriptors_graphics(0, {set}, {blk.offset});
      cmd.draw(3, 1, 0, 0);
      cmd.end_renderpass();
    });
}
