// gtao.cpp — records the GTAO passes.  Follows src/gtao.cpp: resources :17-47, main pass :84-148
// (12-entry angle table + random jitter :109-111, floor dispatch :145), filter :198-239,
// accumulate :286-347 (AccumConstants :300-305, clear_history consumed once :313-315).
#include "gtao.hpp"

#include <cmath>
#include <cstdlib>
#include <limits>
#include <stdexcept>

rendergraph::ImageResourceId create_gtao_texture(rendergraph::RenderGraph &graph, uint32_t width, uint32_t height) {
  gpu::ImageInfo info {VK_FORMAT_R8_UNORM, VK_IMAGE_ASPECT_COLOR_BIT, width, height};
  return graph.create_image(VK_IMAGE_TYPE_2D, info, VK_IMAGE_TILING_OPTIMAL, VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT|VK_IMAGE_USAGE_SAMPLED_BIT);
}

GTAO::GTAO(rendergraph::RenderGraph &graph, uint32_t width, uint32_t height, bool use_ray_query, bool half_res, int pattern_n)
  : deinterleave_n {pattern_n}, pinned_jitter {std::numeric_limits<float>::quiet_NaN()}
{
  if (use_ray_query)
    throw std::runtime_error {"GTAO: ray-query AO needs an acceleration structure; not available on the HIP path"};

  if (half_res) {
    width /= 2;
    height /= 2;
    depth_lod = 1;
  }

  const auto usage = VK_IMAGE_USAGE_STORAGE_BIT|VK_IMAGE_USAGE_SAMPLED_BIT;
  auto make = [&](VkFormat fmt, uint32_t w, uint32_t h, uint32_t layers, VkImageUsageFlags extra) {
    gpu::ImageInfo info {fmt, VK_IMAGE_ASPECT_COLOR_BIT, w, h, 1, 1, layers};
    return graph.create_image(VK_IMAGE_TYPE_2D, info, VK_IMAGE_TILING_OPTIMAL, usage|extra);
  };
  raw = make(VK_FORMAT_R16G16B16A16_SFLOAT, width, height, 1, VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT);
  filtered = make(VK_FORMAT_R16_SFLOAT, width, height, 1, 0);
  prev_frame = make(VK_FORMAT_R16_SFLOAT, width, height, 1, 0);
  output = make(VK_FORMAT_R16_SFLOAT, width, height, 1, 0);
  accumulated_ao = make(VK_FORMAT_R16G16_SFLOAT, width, height, 1, 0);
  accumulated_history = make(VK_FORMAT_R16G16_SFLOAT, width, height, 1, 0);

  const uint32_t pattern_step = 1u << (uint32_t)pattern_n;
  deinterleaved_depth = make(VK_FORMAT_R32_SFLOAT, width/pattern_step, height/pattern_step, pattern_step * pattern_step, 0);

  main_pipeline = gpu::create_compute_pipeline("gtao_compute_main");
  reproject_pipeline = gpu::create_compute_pipeline("gtao_reproject");
  deinterleave_pipeline = gpu::create_compute_pipeline("deinterleave_depth");
  main_deinterleaved_pipeline = gpu::create_compute_pipeline("main_deinterleaved");
  main_pipeline_gfx = gpu::create_graphics_pipeline();
  main_pipeline_gfx.set_program("gtao_main");
  main_pipeline_gfx.set_registers({});
  main_pipeline_gfx.set_vertex_input({});
  main_pipeline_gfx.set_rendersubpass({false, {graph.get_descriptor(raw).format}});
  filter_pipeline = gpu::create_compute_pipeline("gtao_filter");
  accumulate_pipeline = gpu::create_compute_pipeline("gtao_accumulate");
  sampler = gpu::create_sampler(gpu::DEFAULT_SAMPLER);
}

void GTAO::add_main_pass(rendergraph::RenderGraph &graph, const GTAOParams &params, rendergraph::ImageResourceId depth,
  rendergraph::ImageResourceId normal, rendergraph::ImageResourceId material, rendergraph::ImageResourceId preintegrated_pdf)
{
  struct PassData { rendergraph::ImageViewId out, depth, norm, material, pdf; };
  struct PushConsts {
    float base_angle;
    float weight_ratio;
    uint32_t use_mis;
    uint32_t two_directions;
    uint32_t reflections_only;
  };
  static_assert(sizeof(PushConsts) == sizeof(vkr_gtao_push), "push constants must match the C-ABI");
  static_assert(sizeof(GTAOParams) == sizeof(vkr_gtao_params), "GTAOParams must match the C-ABI");

  const float base_angle = next_base_angle();

  const PushConsts push_consts {base_angle, weight_ratio, mis_gtao, two_directions? 255u : 0u, only_reflections? 255u : 0u};
  const auto lod = depth_lod;

  graph.add_task<PassData>("GTAO_main",
    [&](PassData &in, rendergraph::RenderGraphBuilder &builder) {
      const auto cs = VK_SHADER_STAGE_COMPUTE_BIT;
      in.depth = builder.sample_image(depth, cs, VK_IMAGE_ASPECT_DEPTH_BIT, lod, 1, 0, 1);
      in.norm = builder.sample_image(normal, cs);
      in.material = builder.sample_image(material, cs);
      in.pdf = builder.sample_image(preintegrated_pdf, cs);
      in.out = builder.use_storage_image(raw, cs, 0, 0);
    },
    [=](PassData &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto block = cmd.allocate_ubo<GTAOParams>();
      *block.ptr = params;

      auto set = resources.allocate_set(main_pipeline, 0);
      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(in.depth), sampler},
        gpu::UBOBinding {1, cmd.get_ubo_pool(), block},
        gpu::TextureBinding {2, resources.get_view(in.norm), sampler},
        gpu::TextureBinding {3, resources.get_view(in.material), sampler},
        gpu::TextureBinding {4, resources.get_view(in.pdf), sampler},
        gpu::StorageTextureBinding {5, resources.get_view(in.out)});

      const auto extent = resources.get_image(in.out)->get_extent();
      cmd.bind_pipeline(main_pipeline);
      cmd.bind_descriptors_compute(0, {set}, {block.offset});
      cmd.push_constants_compute(0, sizeof(push_consts), &push_consts);
      cmd.dispatch(extent.width/8, extent.height/4, 1);
    });
}

void GTAO::add_filter_pass(rendergraph::RenderGraph &graph, const GTAOParams &params, rendergraph::ImageResourceId depth) {
  struct PassData { rendergraph::ImageViewId out, depth, raw_gtao; };
  struct FilterData { float znear, zfar; };
  const FilterData filter_params {params.znear, params.zfar};
  const auto lod = depth_lod;

  graph.add_task<PassData>("GTAO_filter",
    [&](PassData &in, rendergraph::RenderGraphBuilder &builder) {
      in.depth = builder.sample_image(depth, VK_SHADER_STAGE_COMPUTE_BIT, VK_IMAGE_ASPECT_DEPTH_BIT, lod, 1, 0, 1);
      in.raw_gtao = builder.sample_image(raw, VK_SHADER_STAGE_COMPUTE_BIT);
      in.out = builder.use_storage_image(filtered, VK_SHADER_STAGE_COMPUTE_BIT, 0, 0);
    },
    [=](PassData &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(filter_pipeline, 0);
      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(in.depth), sampler},
        gpu::TextureBinding {1, resources.get_view(in.raw_gtao), sampler},
        gpu::StorageTextureBinding {2, resources.get_view(in.out)});

      const auto extent = resources.get_image(in.out)->get_extent();
      cmd.bind_pipeline(filter_pipeline);
      cmd.bind_descriptors_compute(0, {set}, {});
      cmd.push_constants_compute(0, sizeof(filter_params), &filter_params);
      cmd.dispatch(extent.width/8, extent.height/4, 1);
    });
}

void GTAO::add_accumulate_pass(rendergraph::RenderGraph &graph, const DrawTAAParams &params, const Gbuffer &gbuffer) {
  struct PassData { rendergraph::ImageViewId depth, prev_depth, gtao, accumulated_ao, velocity, history; };
  struct AccumConstants {
    glm::mat4 inverse_camera;
    glm::mat4 prev_inverse_camera;
    glm::mat4 mvp;
    glm::vec4 fovy_aspect_znear_zfar;
  };
  static_assert(sizeof(AccumConstants) == sizeof(vkr_gtao_accum_params), "AccumConstants must match the C-ABI");
  struct PushConstants { uint32_t clear_history; };

  const AccumConstants constants {glm::inverse(params.camera), glm::inverse(params.prev_camera), params.mvp, params.fovy_aspect_znear_zfar};
  const PushConstants pc {clear_history? 1u : 0u};
  clear_history = false;
  const auto lod = depth_lod;

  graph.add_task<PassData>("GTAO_accumulate",
    [&](PassData &in, rendergraph::RenderGraphBuilder &builder) {
      const auto cs = VK_SHADER_STAGE_COMPUTE_BIT;
      in.depth = builder.sample_image(gbuffer.depth, cs, VK_IMAGE_ASPECT_DEPTH_BIT, lod, 1, 0, 1);
      in.prev_depth = builder.sample_image(gbuffer.prev_depth, cs, VK_IMAGE_ASPECT_DEPTH_BIT, lod, 1, 0, 1);
      in.gtao = builder.sample_image(filtered, cs);
      in.accumulated_ao = builder.use_storage_image(accumulated_ao, cs, 0, 0);
      in.velocity = builder.sample_image(gbuffer.downsampled_velocity_vectors, cs);
      in.history = builder.sample_image(accumulated_history, cs);
    },
    [=](PassData &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(accumulate_pipeline, 0);
      auto blk = cmd.allocate_ubo<AccumConstants>();
      *blk.ptr = constants;

      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(in.depth), sampler},
        gpu::TextureBinding {1, resources.get_view(in.prev_depth), sampler},
        gpu::TextureBinding {2, resources.get_view(in.gtao), sampler},
        gpu::StorageTextureBinding {3, resources.get_view(in.accumulated_ao)},
        gpu::TextureBinding {4, resources.get_view(in.velocity), sampler},
        gpu::TextureBinding {5, resources.get_view(in.history), sampler},
        gpu::UBOBinding {6, cmd.get_ubo_pool(), blk});

      const auto extent = resources.get_image(in.accumulated_ao)->get_extent();
      cmd.bind_pipeline(accumulate_pipeline);
      cmd.bind_descriptors_compute(0, {set}, {blk.offset});
      cmd.push_constants_compute(0, sizeof(pc), &pc);
      cmd.dispatch((extent.width + 7)/8, (extent.height + 3)/4, 1);
    });
}

// ---- variants the reference's frame loop never records -------------------
// Shared by the three main-pass flavours: 12-entry angle table + jitter (gtao.cpp:109-111,362-364,487-489).
float GTAO::next_base_angle() {
  static const float table[12] {60.f, 300.f, 180.f, 240.f, 120.f, 0.f, 300.f, 60.f, 180.f, 120.f, 240.f, 0.f};
  const float jitter = std::isnan(pinned_jitter)? (rand()/float(RAND_MAX) - 0.5f) : pinned_jitter;
  return table[frame_count++ % 12]/360.f + jitter;
}

// gtao.cpp:349-413: full-screen triangle into `raw`, fragment program "gtao_main".
void GTAO::add_main_pass_graphics(rendergraph::RenderGraph &graph, const GTAOParams &params,
  rendergraph::ImageResourceId depth, rendergraph::ImageResourceId normal)
{
  struct Views { rendergraph::ImageViewId rt, depth, norm; };
  const vkr_gtao_gfx_push pc {next_base_angle()};
  const auto lod = depth_lod;

  graph.add_task<Views>("GTAO",
    [&](Views &v, rendergraph::RenderGraphBuilder &builder) {
      const auto fs = VK_SHADER_STAGE_FRAGMENT_BIT;
      v.depth = builder.sample_image(depth, fs, VK_IMAGE_ASPECT_DEPTH_BIT, lod, 1, 0, 1);
      v.norm = builder.sample_image(normal, fs);
      v.rt = builder.use_color_attachment(raw, 0, 0);
    },
    [=](Views &v, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto ubo = cmd.allocate_ubo<GTAOParams>();
      *ubo.ptr = params;
      auto set = resources.allocate_set(main_pipeline_gfx, 0);
      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(v.depth), sampler},
        gpu::UBOBinding {1, cmd.get_ubo_pool(), ubo},
        gpu::TextureBinding {2, resources.get_view(v.norm), sampler});

      const auto ext = resources.get_image(v.rt)->get_extent();
      cmd.set_framebuffer(ext.width, ext.height, {resources.get_image_range(v.rt)});
      cmd.bind_pipeline(main_pipeline_gfx);
      cmd.bind_viewport(0.f, 0.f, float(ext.width), float(ext.height), 0.f, 1.f);
      cmd.bind_scissors(0, 0, ext.width, ext.height);
      cmd.bind_descriptors_graphics(0, {set}, {ubo.offset});
      cmd.push_constants_graphics(VK_SHADER_STAGE_FRAGMENT_BIT, 0, sizeof(pc), &pc);
      cmd.draw(3, 1, 0, 0);
      cmd.end_renderpass();
    });
}

// gtao.cpp:241-284: filtered + prev_frame -> output, program "gtao_reproject".
void GTAO::add_reprojection_pass(rendergraph::RenderGraph &graph, const GTAOReprojection &params,
  rendergraph::ImageResourceId depth, rendergraph::ImageResourceId prev_depth)
{
  static_assert(sizeof(GTAOReprojection) == sizeof(vkr_gtao_reprojection), "GTAOReprojection must match the C-ABI");
  struct Views { rendergraph::ImageViewId out, ao, prev_ao, depth, prev_depth; };
  const auto lod = depth_lod;

  graph.add_task<Views>("GTAO_reproject",
    [&](Views &v, rendergraph::RenderGraphBuilder &builder) {
      const auto cs = VK_SHADER_STAGE_COMPUTE_BIT;
      v.depth = builder.sample_image(depth, cs, VK_IMAGE_ASPECT_DEPTH_BIT, lod, 1, 0, 1);
      v.prev_depth = builder.sample_image(prev_depth, cs, VK_IMAGE_ASPECT_DEPTH_BIT, lod, 1, 0, 1);
      v.ao = builder.sample_image(filtered, cs);
      v.prev_ao = builder.sample_image(prev_frame, cs);
      v.out = builder.use_storage_image(output, cs, 0, 0);
    },
    [=](Views &v, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto ubo = cmd.allocate_ubo<GTAOReprojection>();
      *ubo.ptr = params;
      auto set = resources.allocate_set(reproject_pipeline, 0);
      gpu::write_set(set,
        gpu::UBOBinding {0, cmd.get_ubo_pool(), ubo},
        gpu::TextureBinding {1, resources.get_view(v.depth), sampler},
        gpu::TextureBinding {2, resources.get_view(v.prev_depth), sampler},
        gpu::TextureBinding {3, resources.get_view(v.ao), sampler},
        gpu::TextureBinding {4, resources.get_view(v.prev_ao), sampler},
        gpu::StorageTextureBinding {5, resources.get_view(v.out)});

      const auto ext = resources.get_image(v.out)->get_extent();
      cmd.bind_pipeline(reproject_pipeline);
      cmd.bind_descriptors_compute(0, {set}, {ubo.offset});
      cmd.dispatch(ext.width/8, ext.height/4, 1);
    });
}

// gtao.cpp:445-470: depth (lod view) -> R32F array, program "deinterleave_depth".  The dispatch is
// sized by the *array* extent, as in the reference.
void GTAO::deinterleave_depth(rendergraph::RenderGraph &graph, rendergraph::ImageResourceId depth) {
  struct Views { rendergraph::ImageViewId depth, out; };
  const auto lod = depth_lod;
  const vkr_deinterleave_push pc {deinterleave_n};

  graph.add_task<Views>("GTAO_deinterleave",
    [&](Views &v, rendergraph::RenderGraphBuilder &builder) {
      v.depth = builder.sample_image(depth, VK_SHADER_STAGE_COMPUTE_BIT, VK_IMAGE_ASPECT_DEPTH_BIT, lod, 1, 0, 1);
      v.out = builder.use_storage_image_array(deinterleaved_depth, VK_SHADER_STAGE_COMPUTE_BIT);
    },
    [=](Views &v, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(deinterleave_pipeline, 0);
      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(v.depth), sampler},
        gpu::StorageTextureBinding {1, resources.get_view(v.out)});

      const auto ext = resources.get_image(v.out)->get_extent();
      cmd.bind_pipeline(deinterleave_pipeline);
      cmd.bind_descriptors_compute(0, {set}, {});
      cmd.push_constants_compute(0, sizeof(pc), &pc);
      cmd.dispatch(ext.width/8, ext.height/4, 1);
    });
}

// gtao.cpp:472-526: program "main_deinterleaved", one dispatch per array layer of the *output* image
// (raw has one), exactly as the reference loops.
void GTAO::add_main_pass_deinterleaved(rendergraph::RenderGraph &graph, const GTAOParams &params, rendergraph::ImageResourceId normal) {
  struct Views { rendergraph::ImageViewId out, depth, norm; };
  const float base_angle = next_base_angle();
  const int pattern = deinterleave_n;

  graph.add_task<Views>("GTAO_deinterleaved",
    [&](Views &v, rendergraph::RenderGraphBuilder &builder) {
      const auto cs = VK_SHADER_STAGE_COMPUTE_BIT;
      v.depth = builder.sample_image(deinterleaved_depth, cs);
      v.norm = builder.sample_image(normal, cs);
      v.out = builder.use_storage_image(raw, cs, 0, 0);
    },
    [=](Views &v, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto ubo = cmd.allocate_ubo<GTAOParams>();
      *ubo.ptr = params;
      const auto &target = resources.get_image(v.out);
      const auto ext = target->get_extent();
      for (uint32_t layer = 0; layer < target->get_array_layers(); layer++) {
        auto set = resources.allocate_set(main_deinterleaved_pipeline, 0);
        gpu::write_set(set,
          gpu::TextureBinding {0, resources.get_view(v.depth), sampler},
          gpu::UBOBinding {1, cmd.get_ubo_pool(), ubo},
          gpu::TextureBinding {2, resources.get_view(v.norm), sampler},
          gpu::StorageTextureBinding {3, resources.get_view(v.out)});
        const vkr_gtao_deinterleaved_push pc {pattern, layer, base_angle};
        cmd.bind_pipeline(main_deinterleaved_pipeline);
        cmd.bind_descriptors_compute(0, {set}, {ubo.offset});
        cmd.push_constants_compute(0, sizeof(pc), &pc);
        cmd.dispatch(ext.width/8, ext.height/4, 1);
      }
    });
}
