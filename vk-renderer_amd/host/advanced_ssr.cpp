// advanced_ssr.cpp — records the SSR passes.  Follows src/advanced_ssr.cpp: Halton table :8-34,
// resources :36-93, LUT :95-114, trace :147-214 (frame counter cycling :168-171, depth view =
// mips 1..L-1 :186), filter :308-369 (flags :331-338, depth view mips 0..9 :342), blur :371-438.
#include "advanced_ssr.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

#define HALTON_SEQ_SIZE 128
#define NORMALIZE_REFLECTIONS 1
#define ACCUMULATE_REFLECTIONS 2
#define BILATERAL_FILTER 4

// radical inverse of `index` in `base`, with the reference's float-floor division
// (advanced_ssr.cpp:16) kept as is
static float halton_elem(uint32_t index, uint32_t base) {
  float scale = 1.f, result = 0.f;
  for (uint32_t current = index;;) {
    scale = scale / float(base);
    result = result + scale * float(current % base);
    current = uint32_t(std::floor(float(current) / float(base)));
    if (current == 0) break;
  }
  return result;
}

std::vector<glm::vec4> halton23_seq(uint32_t count) {
  std::vector<glm::vec4> seq(count);
  for (uint32_t i = 0; i < count; i++)
    seq[i] = glm::vec4 {halton_elem(i + 1, 2), halton_elem(i + 1, 3), 0.f, 0.f};
  return seq;
}

AdvancedSSR::AdvancedSSR(rendergraph::RenderGraph &graph, uint32_t w, uint32_t h) {
  trace_pass = gpu::create_compute_pipeline("sssr_trace");
  filter_pass = gpu::create_compute_pipeline("sssr_filter");
  blur_pass = gpu::create_compute_pipeline("sssr_blur");
  preintegrate_pass = gpu::create_compute_pipeline("pdf_preintegrate");
  preintegrate_brdf_pass = gpu::create_compute_pipeline("brdf_preintegrate");
  classification_pass = gpu::create_compute_pipeline("sssr_classification");
  trace_indirect_pass = gpu::create_compute_pipeline("sssr_trace_indirect");

  // xy: the reference's table (advanced_ssr.cpp:22-34); zw: cos/sin of 2*PI*y evaluated on the host
  // for the HIP trace kernel (vkr_halton23_fill, include/vkr_postfx.h) — zero in the reference
  auto halton_samples = halton23_seq(HALTON_SEQ_SIZE);
  {
    std::vector<float> filled(4 * HALTON_SEQ_SIZE);
    vkr_halton23_fill(filled.data(), HALTON_SEQ_SIZE);
    for (uint32_t i = 0; i < HALTON_SEQ_SIZE; i++) {
      if (filled[4 * i] != halton_samples[i].x || filled[4 * i + 1] != halton_samples[i].y)
        throw std::runtime_error {"Halton table mismatch between host pass and C-ABI helper"};
      halton_samples[i].z = filled[4 * i + 2];
      halton_samples[i].w = filled[4 * i + 3];
    }
  }
  const uint64_t bytes = sizeof(halton_samples[0]) * HALTON_SEQ_SIZE;
  halton_buffer = gpu::create_buffer(VMA_MEMORY_USAGE_CPU_TO_GPU, bytes, VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT);
  std::memcpy(halton_buffer->get_mapped_ptr(), halton_samples.data(), bytes);

  const auto usage = VK_IMAGE_USAGE_SAMPLED_BIT|VK_IMAGE_USAGE_STORAGE_BIT;
  auto make = [&](VkFormat fmt, uint32_t iw, uint32_t ih) {
    return graph.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo {fmt, VK_IMAGE_ASPECT_COLOR_BIT, iw, ih}, VK_IMAGE_TILING_OPTIMAL, usage);
  };
  rays = make(VK_FORMAT_R16G16B16A16_UNORM, w/2, h/2);
  rays_occlusion = make(VK_FORMAT_R16_SFLOAT, w/2, h/2);
  reflections = make(VK_FORMAT_R8G8B8A8_UNORM, w/2, h/2);
  blurred_reflection = make(VK_FORMAT_R8G8B8A8_UNORM, w/2, h/2);
  blurred_reflection_history = make(VK_FORMAT_R8G8B8A8_UNORM, w/2, h/2);
  preintegrated_pdf = make(VK_FORMAT_R32_SFLOAT, 1024, 1024);
  preintegrated_brdf = make(VK_FORMAT_R16G16_SFLOAT, 1024, 1024);

  sampler = gpu::create_sampler(gpu::DEFAULT_SAMPLER);

  // advanced_ssr.cpp:77-83: indirect arguments and one tile-index slot per 8x8 block of the full-res frame
  const auto indirect_usage = VK_BUFFER_USAGE_STORAGE_BUFFER_BIT|VK_BUFFER_USAGE_INDIRECT_BUFFER_BIT|VK_BUFFER_USAGE_TRANSFER_DST_BIT;
  reflective_indirect = graph.create_buffer(VMA_MEMORY_USAGE_GPU_ONLY, sizeof(VkDispatchIndirectCommand), indirect_usage);
  glossy_indirect = graph.create_buffer(VMA_MEMORY_USAGE_GPU_ONLY, sizeof(VkDispatchIndirectCommand), indirect_usage);
  const uint64_t tile_bytes = sizeof(int) * std::max<uint64_t>(1, uint64_t(w) * h/64);
  reflective_tiles = graph.create_buffer(VMA_MEMORY_USAGE_GPU_ONLY, tile_bytes, VK_BUFFER_USAGE_STORAGE_BUFFER_BIT);
  glossy_tiles = graph.create_buffer(VMA_MEMORY_USAGE_GPU_ONLY, tile_bytes, VK_BUFFER_USAGE_STORAGE_BUFFER_BIT);
}

void AdvancedSSR::preintegrate_pdf(rendergraph::RenderGraph &graph) {
  struct Input { rendergraph::ImageViewId out_pdf; };
  graph.add_task<Input>("SSR_preintegrate",
    [&](Input &in, rendergraph::RenderGraphBuilder &builder) {
      in.out_pdf = builder.use_storage_image(preintegrated_pdf, VK_SHADER_STAGE_COMPUTE_BIT, 0, 0);
    },
    [=](Input &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(preintegrate_pass, 0);
      gpu::write_set(set, gpu::StorageTextureBinding {0, resources.get_view(in.out_pdf)});
      const auto extent = resources.get_image(in.out_pdf)->get_extent();
      cmd.bind_pipeline(preintegrate_pass);
      cmd.bind_descriptors_compute(0, {set});
      cmd.dispatch((extent.width + 7)/8, (extent.height + 3)/4, 1);
    });
}

// advanced_ssr.cpp:116-136
void AdvancedSSR::preintegrate_brdf(rendergraph::RenderGraph &graph) {
  struct Input { rendergraph::ImageViewId out_brdf; };
  graph.add_task<Input>("BRDF_preintegrate",
    [&](Input &in, rendergraph::RenderGraphBuilder &builder) {
      in.out_brdf = builder.use_storage_image(preintegrated_brdf, VK_SHADER_STAGE_COMPUTE_BIT, 0, 0);
    },
    [=](Input &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(preintegrate_brdf_pass, 0);
      gpu::write_set(set,
        gpu::UBOBinding {0, halton_buffer},
        gpu::StorageTextureBinding {1, resources.get_view(in.out_brdf)});
      const auto extent = resources.get_image(in.out_brdf)->get_extent();
      cmd.bind_pipeline(preintegrate_brdf_pass);
      cmd.bind_descriptors_compute(0, {set}, {0});
      cmd.dispatch((extent.width + 7)/8, (extent.height + 3)/4, 1);
    });
}

struct TraceParams {
  glm::mat4 normal_mat;
  uint32_t frame_random;
  float fovy;
  float aspect;
  float znear;
  float zfar;
};
static_assert(sizeof(TraceParams) == sizeof(vkr_trace_params), "TraceParams must match the C-ABI");

void AdvancedSSR::run_trace_pass(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff, rendergraph::ImageResourceId ssr_occlusion) {
  const TraceParams config {params.normal_mat, counter, params.fovy, params.aspect, params.znear, params.zfar};
  struct PushConstants { float max_roughness; };
  const PushConstants push_consts {settings.max_rougness};

  if (settings.update_random) {
    counter++;
    counter = counter % settings.max_accumulated_rays;
  }

  struct Input { rendergraph::ImageViewId depth, normal, material, out, occlusion, preintegrated_pdf; };

  // single GPU: the march reads image mips 1..L-1 of gbuff.depth; tiled: the gathered whole-frame pyramid
  const bool tiled = gbuff.tiled;
  const auto hiz = tiled? gbuff.frame_hiz : gbuff.depth;
  const auto normals = tiled? gbuff.frame_normals : gbuff.downsampled_normals;
  const auto mips_count = graph.get_descriptor(hiz).mip_levels;

  graph.add_task<Input>("SSSR_trace",
    [&](Input &in, rendergraph::RenderGraphBuilder &builder) {
      const auto cs = VK_SHADER_STAGE_COMPUTE_BIT;
      in.depth = tiled? builder.sample_image(hiz, cs, VK_IMAGE_ASPECT_DEPTH_BIT, 0, mips_count, 0, 1)
                      : builder.sample_image(hiz, cs, VK_IMAGE_ASPECT_DEPTH_BIT, 1, mips_count - 1, 0, 1);
      in.normal = builder.sample_image(normals, cs);
      in.material = builder.sample_image(gbuff.material, cs);
      in.out = builder.use_storage_image(rays, cs, 0, 0);
      in.occlusion = builder.use_storage_image(ssr_occlusion, cs, 0, 0);
      in.preintegrated_pdf = builder.sample_image(preintegrated_pdf, cs);
    },
    [=](Input &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(trace_pass, 0);
      auto blk = cmd.allocate_ubo<TraceParams>();
      *blk.ptr = config;

      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(in.depth), sampler},
        gpu::TextureBinding {1, resources.get_view(in.normal), sampler},
        gpu::TextureBinding {2, resources.get_view(in.material), sampler},
        gpu::UBOBinding {3, cmd.get_ubo_pool(), blk},
        gpu::UBOBinding {4, halton_buffer},
        gpu::StorageTextureBinding {5, resources.get_view(in.out)},
        gpu::StorageTextureBinding {6, resources.get_view(in.occlusion)},
        gpu::TextureBinding {7, resources.get_view(in.preintegrated_pdf), sampler});

      const auto ext = resources.get_image(in.out)->get_extent();
      cmd.bind_pipeline(trace_pass);
      cmd.bind_descriptors_compute(0, {set}, {blk.offset, 0});
      cmd.push_constants_compute(0, sizeof(push_consts), &push_consts);
      cmd.dispatch((ext.width + 7)/8, (ext.height + 7)/8, 1);
    });
}

void AdvancedSSR::run_filter_pass(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff) {
  const TraceParams config {params.normal_mat, counter, params.fovy, params.aspect, params.znear, params.zfar};
  struct Input { rendergraph::ImageViewId depth, normal, albedo, material, rays, reflection; };
  struct PushConstants { uint32_t render_flags; };

  PushConstants pc {0u};
  pc.render_flags |= settings.normalize_reflections? NORMALIZE_REFLECTIONS : 0;
  pc.render_flags |= settings.accumulate_reflections? ACCUMULATE_REFLECTIONS : 0;
  pc.render_flags |= settings.bilateral_filter? BILATERAL_FILTER : 0;

  const uint32_t depth_mips = std::min(10u, graph.get_descriptor(gbuff.depth).mip_levels);
  const auto albedo = gbuff.tiled? gbuff.frame_albedo : gbuff.albedo;  // hit colour: unbounded reach

  graph.add_task<Input>("SSSR_filter",
    [&](Input &in, rendergraph::RenderGraphBuilder &builder) {
      const auto cs = VK_SHADER_STAGE_COMPUTE_BIT;
      in.depth = builder.sample_image(gbuff.depth, cs, VK_IMAGE_ASPECT_DEPTH_BIT, 0, depth_mips, 0, 1);
      in.normal = builder.sample_image(gbuff.normal, cs);
      in.albedo = builder.sample_image(albedo, cs);
      in.rays = builder.sample_image(rays, cs);
      in.material = builder.sample_image(gbuff.material, cs);
      in.reflection = builder.use_storage_image(reflections, cs, 0, 0);
    },
    [=](Input &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(filter_pass, 0);
      auto blk = cmd.allocate_ubo<TraceParams>();
      *blk.ptr = config;

      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(in.rays), sampler},
        gpu::TextureBinding {1, resources.get_view(in.depth), sampler},
        gpu::TextureBinding {2, resources.get_view(in.albedo), sampler},
        gpu::TextureBinding {3, resources.get_view(in.normal), sampler},
        gpu::TextureBinding {4, resources.get_view(in.material), sampler},
        gpu::StorageTextureBinding {5, resources.get_view(in.reflection)},
        gpu::UBOBinding {6, cmd.get_ubo_pool(), blk});

      const auto ext = resources.get_image(in.reflection)->get_extent();
      cmd.bind_pipeline(filter_pass);
      cmd.bind_descriptors_compute(0, {set}, {blk.offset});
      cmd.push_constants_compute(0, sizeof(pc), &pc);
      cmd.dispatch((ext.width + 7)/8, (ext.height + 7)/8, 1);
    });
}

void AdvancedSSR::run_blur_pass(rendergraph::RenderGraph &graph, const AdvancedSSRParams &, const DrawTAAParams &taa_params, const Gbuffer &gbuff) {
  struct Input { rendergraph::ImageViewId depth, normal, material, reflections, history, velocity, history_depth, result; };
  struct PushConstants {
    float max_roughness;
    uint32_t accumulate;
    uint32_t disable_blur;
  };
  struct Params {
    glm::mat4 inverse_camera;
    glm::mat4 prev_inverse_camera;
    glm::vec4 fovy_aspect_znear_zfar;
  };
  static_assert(sizeof(Params) == sizeof(vkr_reproject_params), "ReprojectConsts must match the C-ABI");

  const PushConstants pc {settings.max_rougness, settings.accumulate_reflections, !settings.use_blur};
  const Params buf {glm::inverse(taa_params.camera), glm::inverse(taa_params.prev_camera), taa_params.fovy_aspect_znear_zfar};
  const uint32_t depth_mips = std::min(10u, graph.get_descriptor(gbuff.depth).mip_levels);

  graph.add_task<Input>("SSSR_blur",
    [&](Input &in, rendergraph::RenderGraphBuilder &builder) {
      const auto cs = VK_SHADER_STAGE_COMPUTE_BIT;
      in.depth = builder.sample_image(gbuff.depth, cs, VK_IMAGE_ASPECT_DEPTH_BIT, 0, depth_mips, 0, 1);
      in.normal = builder.sample_image(gbuff.normal, cs);
      in.reflections = builder.sample_image(reflections, cs);
      in.material = builder.sample_image(gbuff.material, cs);
      in.history = builder.sample_image(blurred_reflection_history, cs);
      in.velocity = builder.sample_image(gbuff.downsampled_velocity_vectors, cs);
      in.history_depth = builder.sample_image(gbuff.prev_depth, cs, VK_IMAGE_ASPECT_DEPTH_BIT, 0, depth_mips, 0, 1);
      in.result = builder.use_storage_image(blurred_reflection, cs, 0, 0);
    },
    [=](Input &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto blk = cmd.allocate_ubo<Params>();
      *blk.ptr = buf;

      auto set = resources.allocate_set(blur_pass, 0);
      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(in.depth), sampler},
        gpu::TextureBinding {1, resources.get_view(in.normal), sampler},
        gpu::TextureBinding {2, resources.get_view(in.reflections), sampler},
        gpu::TextureBinding {3, resources.get_view(in.material), sampler},
        gpu::TextureBinding {4, resources.get_view(in.history), sampler},
        gpu::TextureBinding {5, resources.get_view(in.velocity), sampler},
        gpu::TextureBinding {6, resources.get_view(in.history_depth), sampler},
        gpu::StorageTextureBinding {7, resources.get_view(in.result)},
        gpu::UBOBinding {8, cmd.get_ubo_pool(), blk});

      const auto ext = resources.get_image(in.result)->get_extent();
      cmd.bind_pipeline(blur_pass);
      cmd.bind_descriptors_compute(0, {set}, {blk.offset});
      cmd.push_constants_compute(0, sizeof(pc), &pc);
      cmd.dispatch((ext.width + 7)/8, (ext.height + 7)/8, 1);
    });
}

// advanced_ssr.cpp:440-452
void AdvancedSSR::clear_indirect_params(rendergraph::RenderGraph &graph) {
  struct Nothing {};
  graph.add_task<Nothing>("SSSR_Clear",
    [&](Nothing &, rendergraph::RenderGraphBuilder &builder) {
      builder.transfer_write(reflective_indirect);
      builder.transfer_write(glossy_indirect);
    },
    [=](Nothing &, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      const VkDispatchIndirectCommand none {0, 1, 1};
      for (auto id : {reflective_indirect, glossy_indirect})
        cmd.update_buffer(resources.get_buffer(id)->api_buffer(), 0, none);
    });
}

// advanced_ssr.cpp:454-495
void AdvancedSSR::run_classification_pass(rendergraph::RenderGraph &graph, const AdvancedSSRParams &, const Gbuffer &gbuff) {
  struct Input { rendergraph::ImageViewId material_tex; };
  const auto extent = graph.get_descriptor(rays).extent2D();
  const vkr_classification_push pc {int(extent.width), int(extent.height), settings.max_rougness, settings.glossy_roughness_value};

  graph.add_task<Input>("SSSR_Classification",
    [&](Input &in, rendergraph::RenderGraphBuilder &builder) {
      const auto cs = VK_SHADER_STAGE_COMPUTE_BIT;
      in.material_tex = builder.sample_image(gbuff.material, cs);
      for (auto id : {reflective_indirect, glossy_indirect, reflective_tiles, glossy_tiles})
        builder.use_storage_buffer(id, cs, false);
    },
    [=](Input &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(classification_pass, 0);
      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(in.material_tex), sampler},
        gpu::SSBOBinding {1, resources.get_buffer(reflective_tiles)},
        gpu::SSBOBinding {2, resources.get_buffer(glossy_tiles)},
        gpu::SSBOBinding {3, resources.get_buffer(reflective_indirect)},
        gpu::SSBOBinding {4, resources.get_buffer(glossy_indirect)});

      cmd.bind_pipeline(classification_pass);
      cmd.bind_descriptors_compute(0, {set});
      cmd.push_constants_compute(0, sizeof(pc), &pc);
      cmd.dispatch((extent.width + 7)/8, (extent.height + 7)/8, 1);
    });
}

// advanced_ssr.cpp:216-302: two indirect dispatches of one program, mirror tiles then glossy tiles
void AdvancedSSR::run_trace_indirect_pass(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff) {
  const TraceParams config {params.normal_mat, counter, params.fovy, params.aspect, params.znear, params.zfar};
  const float max_roughness = settings.max_rougness;

  if (settings.update_random) {
    counter++;
    counter = counter % settings.max_accumulated_rays;
  }

  struct Input { rendergraph::ImageViewId depth, normal, material, out; };
  const bool tiled = gbuff.tiled;
  const auto hiz = tiled? gbuff.frame_hiz : gbuff.depth;
  const auto normals = tiled? gbuff.frame_normals : gbuff.downsampled_normals;
  const auto mips_count = graph.get_descriptor(hiz).mip_levels;

  graph.add_task<Input>("SSSR_trace",
    [&](Input &in, rendergraph::RenderGraphBuilder &builder) {
      const auto cs = VK_SHADER_STAGE_COMPUTE_BIT;
      in.depth = tiled? builder.sample_image(hiz, cs, VK_IMAGE_ASPECT_DEPTH_BIT, 0, mips_count, 0, 1)
                      : builder.sample_image(hiz, cs, VK_IMAGE_ASPECT_DEPTH_BIT, 1, mips_count - 1, 0, 1);
      in.normal = builder.sample_image(normals, cs);
      in.material = builder.sample_image(gbuff.material, cs);
      in.out = builder.use_storage_image(rays, cs, 0, 0);
      builder.use_indirect_buffer(reflective_indirect);
      builder.use_indirect_buffer(glossy_indirect);
      builder.use_storage_buffer(reflective_tiles, cs);
      builder.use_storage_buffer(glossy_tiles, cs);
    },
    [=](Input &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto blk = cmd.allocate_ubo<TraceParams>();
      *blk.ptr = config;
      cmd.bind_pipeline(trace_indirect_pass);

      const rendergraph::BufferResourceId lists[2] {reflective_tiles, glossy_tiles};
      const rendergraph::BufferResourceId arguments[2] {reflective_indirect, glossy_indirect};
      for (uint32_t kind = 0; kind < 2; kind++) {  // 0 - mirror, 1 - glossy
        auto set = resources.allocate_set(trace_indirect_pass, 0);
        gpu::write_set(set,
          gpu::TextureBinding {0, resources.get_view(in.depth), sampler},
          gpu::TextureBinding {1, resources.get_view(in.normal), sampler},
          gpu::TextureBinding {2, resources.get_view(in.material), sampler},
          gpu::UBOBinding {3, cmd.get_ubo_pool(), blk},
          gpu::UBOBinding {4, halton_buffer},
          gpu::StorageTextureBinding {5, resources.get_view(in.out)},
          gpu::SSBOBinding {6, resources.get_buffer(lists[kind])});

        const vkr_trace_indirect_push pc {kind, max_roughness};
        cmd.bind_descriptors_compute(0, {set}, {blk.offset, 0});
        cmd.push_constants_compute(0, sizeof(pc), &pc);
        cmd.dispatch_indirect(resources.get_buffer(arguments[kind])->api_buffer());
      }
    });
}

void AdvancedSSR::run_trace(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const Gbuffer &gbuff, rendergraph::ImageResourceId ssr_occlusion) {
  if (settings.use_tile_classification) {  // the path advanced_ssr.cpp:547-550 keeps commented out
    clear_indirect_params(graph);
    run_classification_pass(graph, params, gbuff);
    run_trace_indirect_pass(graph, params, gbuff);
  } else {
    run_trace_pass(graph, params, gbuff, ssr_occlusion);
  }
}

void AdvancedSSR::run_resolve(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const DrawTAAParams &taa_params, const Gbuffer &gbuff) {
  run_filter_pass(graph, params, gbuff);
  run_blur_pass(graph, params, taa_params, gbuff);
}

void AdvancedSSR::run(rendergraph::RenderGraph &graph, const AdvancedSSRParams &params, const DrawTAAParams &taa_params,
  const Gbuffer &gbuff, rendergraph::ImageResourceId ssr_occlusion)
{
  run_trace(graph, params, gbuff, ssr_occlusion);
  run_resolve(graph, params, taa_params, gbuff);
}
