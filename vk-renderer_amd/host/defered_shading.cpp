// defered_shading.cpp — records the deferred-shading composite.  Follows src/defered_shading.cpp:
// ShaderConstants :4-12, GPU-side constant buffer written by update_params :33-45, bindings 0-8 :93-102
// (the shadow map, binding 5, is bound by the reference but never read by the shader; it may be
// left unbound here), push constants {vec2 min_max_roughness, uint show_ao} :68-72, full-screen draw.
#include "defered_shading.hpp"

#include <cstring>

struct ShaderConstants {
  glm::mat4 inverse_camera;
  glm::mat4 camera;
  glm::mat4 shadow_mvp;
  float fovy;
  float aspect;
  float znear;
  float zfar;
};
static_assert(sizeof(ShaderConstants) == sizeof(vkr_shading_params), "ShaderConstants must match the C-ABI");

DeferedShadingPass::DeferedShadingPass(rendergraph::RenderGraph &graph, SDL_Window *) {
  pipeline = gpu::create_graphics_pipeline();
  pipeline.set_program("defered_shading");
  pipeline.set_registers({});
  pipeline.set_vertex_input({});
  sampler = gpu::create_sampler(gpu::DEFAULT_SAMPLER);
  ubo_consts = graph.create_buffer(VMA_MEMORY_USAGE_GPU_ONLY, sizeof(ShaderConstants), VK_BUFFER_USAGE_TRANSFER_DST_BIT|VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT);
  graph_ref = &graph;
}

void DeferedShadingPass::update_params(const glm::mat4 &camera, const glm::mat4 &shadow, float fovy, float aspect, float znear, float zfar) {
  const ShaderConstants consts {glm::inverse(camera), camera, shadow, fovy, aspect, znear, zfar};
  // gpu_transfer::write_buffer in the reference; here the buffer keeps a host shadow the program reads
  std::memcpy(graph_ref->get_buffer(ubo_consts)->get_mapped_ptr(), &consts, sizeof(consts));
}

void DeferedShadingPass::draw(rendergraph::RenderGraph &graph, const Gbuffer &gbuffer, rendergraph::ImageResourceId shadow,
  rendergraph::ImageResourceId ssao, rendergraph::ImageResourceId brdf_tex, rendergraph::ImageResourceId reflections,
  rendergraph::ImageResourceId out_image)
{
  struct PassData {
    rendergraph::ImageViewId albedo, normal, material, depth, rt, shadow, ssao, ssr, brdf;
    rendergraph::BufferResourceId ubo;
    bool has_shadow;
  };
  struct PushConsts {
    glm::vec2 min_max_roughness;
    uint32_t show_ao;
  };
  static_assert(sizeof(PushConsts) == sizeof(vkr_shading_push), "push constants must match the C-ABI");
  const PushConsts pc {min_max_roughness, only_ao? 1u : 0u};
  pipeline.set_rendersubpass({false, {graph.get_descriptor(out_image).format}});
  const bool has_shadow = shadow.get_index() != ~0u;

  graph.add_task<PassData>("DeferedShading",
    [&](PassData &in, rendergraph::RenderGraphBuilder &builder) {
      const auto fs = VK_SHADER_STAGE_FRAGMENT_BIT;
      in.albedo = builder.sample_image(gbuffer.albedo, fs);
      in.normal = builder.sample_image(gbuffer.normal, fs);
      in.material = builder.sample_image(gbuffer.material, fs);
      in.depth = builder.sample_image(gbuffer.depth, fs, VK_IMAGE_ASPECT_DEPTH_BIT);
      in.rt = builder.use_color_attachment(out_image, 0, 0);
      in.has_shadow = has_shadow;
      if (has_shadow) in.shadow = builder.sample_image(shadow, fs, VK_IMAGE_ASPECT_DEPTH_BIT, 0, 1, 0, 1);
      in.ssao = builder.sample_image(ssao, fs);
      in.ssr = builder.sample_image(reflections, fs);
      in.brdf = builder.sample_image(brdf_tex, fs);
      in.ubo = ubo_consts;
      builder.use_uniform_buffer(in.ubo, VK_SHADER_STAGE_VERTEX_BIT);
    },
    [=](PassData &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(pipeline, 0);
      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(in.albedo), sampler},
        gpu::TextureBinding {1, resources.get_view(in.normal), sampler},
        gpu::TextureBinding {2, resources.get_view(in.material), sampler},
        gpu::TextureBinding {3, resources.get_view(in.depth), sampler},
        gpu::UBOBinding {4, resources.get_buffer(in.ubo)},
        gpu::TextureBinding {6, resources.get_view(in.ssao), sampler},
        gpu::TextureBinding {7, resources.get_view(in.brdf), sampler},
        gpu::TextureBinding {8, resources.get_view(in.ssr), sampler});
      if (in.has_shadow) gpu::write_set(set, gpu::TextureBinding {5, resources.get_view(in.shadow), sampler});

      const auto extent = resources.get_image(in.rt)->get_extent();
      cmd.set_framebuffer(extent.width, extent.height, {resources.get_image_range(in.rt)});
      cmd.bind_pipeline(pipeline);
      cmd.bind_viewport(0.f, 0.f, float(extent.width), float(extent.height), 0.f, 1.f);
      cmd.bind_scissors(0, 0, extent.width, extent.height);
      cmd.bind_descriptors_graphics(0, {set}, {0});
      cmd.push_constants_graphics(VK_SHADER_STAGE_FRAGMENT_BIT, 0, sizeof(pc), &pc);
      cmd.draw(3, 1, 0, 0);
      cmd.end_renderpass();
    });
}
