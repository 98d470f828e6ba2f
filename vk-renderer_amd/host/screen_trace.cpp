// screen_trace.cpp — records the ScreenSpaceTrace passes.  Follows src/screen_trace.cpp: three
// full-res RGBA16F targets :3-21, trace :23-95 (angle table + jitter :47-53, dispatch w/8 x h/8 :93),
// filter :97-140 (push {znear, zfar}), accumulate :142-181 (push {fovy, aspect, znear, zfar}).
#include "screen_trace.hpp"

#include <cmath>

namespace {
constexpr auto CS = VK_SHADER_STAGE_COMPUTE_BIT;
// depth is bound as a one-mip view of mip 0 everywhere in this pass family
rendergraph::ImageViewId depth_mip0(rendergraph::RenderGraphBuilder &builder, rendergraph::ImageResourceId depth) {
  return builder.sample_image(depth, CS, VK_IMAGE_ASPECT_DEPTH_BIT, 0, 1, 0, 1);
}
}

ScreenSpaceTrace::ScreenSpaceTrace(rendergraph::RenderGraph &graph, uint32_t width, uint32_t height) {
  const gpu::ImageInfo info {VK_FORMAT_R16G16B16A16_SFLOAT, VK_IMAGE_ASPECT_COLOR_BIT, width, height};
  const auto usage = VK_IMAGE_USAGE_STORAGE_BIT|VK_IMAGE_USAGE_SAMPLED_BIT;
  for (auto *id : {&raw, &filtered, &accumulated})
    *id = graph.create_image(VK_IMAGE_TYPE_2D, info, VK_IMAGE_TILING_OPTIMAL, usage);

  trace_pipeline = gpu::create_compute_pipeline("screen_trace_main");
  filter_pipeline = gpu::create_compute_pipeline("screen_trace_filter");
  accum_pipeline = gpu::create_compute_pipeline("screen_trace_accumulate");
  sampler = gpu::create_sampler(gpu::DEFAULT_SAMPLER);
}

void ScreenSpaceTrace::add_main_pass(rendergraph::RenderGraph &graph, const ScreenTraceParams &params,
  rendergraph::ImageResourceId depth, rendergraph::ImageResourceId normal, rendergraph::ImageResourceId color,
  rendergraph::ImageResourceId material)
{
  struct Views { rendergraph::ImageViewId out, depth, norm, color, material; };

  static const float table[12] {60.f, 300.f, 180.f, 240.f, 120.f, 0.f, 300.f, 60.f, 180.f, 120.f, 240.f, 0.f};
  // the reference always draws both randoms, in this order
  const float drawn_jitter = random_floats(generator) - 0.5f;
  const float drawn_offset = random_floats(generator);

  vkr_screen_trace_params ubo_data {};
  static_assert(sizeof(params.normal_mat) == sizeof(ubo_data.normal_mat), "mat4 layout");
  std::memcpy(&ubo_data.normal_mat, &params.normal_mat, sizeof(ubo_data.normal_mat));
  ubo_data.angle_offset = table[frame_count++ % 12]/360.f + (std::isnan(pinned_jitter)? drawn_jitter : pinned_jitter);
  ubo_data.random_offset = std::isnan(pinned_offset)? drawn_offset : pinned_offset;
  ubo_data.fovy = params.fovy;
  ubo_data.aspect = params.aspect;
  ubo_data.znear = params.znear;
  ubo_data.zfar = params.zfar;

  graph.add_task<Views>("ScreenTrace",
    [&](Views &v, rendergraph::RenderGraphBuilder &builder) {
      v.depth = depth_mip0(builder, depth);
      v.norm = builder.sample_image(normal, CS);
      v.color = builder.sample_image(color, CS);
      v.material = builder.sample_image(material, CS);
      v.out = builder.use_storage_image(raw, CS, 0, 0);
    },
    [=](Views &v, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto ubo = cmd.allocate_ubo<vkr_screen_trace_params>();
      *ubo.ptr = ubo_data;
      auto set = resources.allocate_set(trace_pipeline, 0);
      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(v.depth), sampler},
        gpu::TextureBinding {1, resources.get_view(v.norm), sampler},
        gpu::TextureBinding {2, resources.get_view(v.color), sampler},
        gpu::TextureBinding {3, resources.get_view(v.material), sampler},
        gpu::StorageTextureBinding {4, resources.get_view(v.out)},
        gpu::UBOBinding {5, cmd.get_ubo_pool(), ubo});

      const auto ext = resources.get_image(v.out)->get_extent();
      cmd.bind_pipeline(trace_pipeline);
      cmd.bind_descriptors_compute(0, {set}, {ubo.offset});
      cmd.dispatch(ext.width/8, ext.height/8, 1);
    });
}

void ScreenSpaceTrace::add_filter_pass(rendergraph::RenderGraph &graph, const ScreenTraceParams &params, rendergraph::ImageResourceId depth) {
  struct Views { rendergraph::ImageViewId depth, raw, filtered; };
  const vkr_screen_trace_filter_push pc {params.znear, params.zfar};

  graph.add_task<Views>("ScreenTraceFilter",
    [&](Views &v, rendergraph::RenderGraphBuilder &builder) {
      v.depth = depth_mip0(builder, depth);
      v.raw = builder.sample_image(raw, CS);
      v.filtered = builder.use_storage_image(filtered, CS, 0, 0);
    },
    [=](Views &v, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(filter_pipeline, 0);
      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(v.raw), sampler},
        gpu::TextureBinding {1, resources.get_view(v.depth), sampler},
        gpu::StorageTextureBinding {2, resources.get_view(v.filtered)});

      const auto ext = resources.get_image(v.filtered)->get_extent();
      cmd.bind_pipeline(filter_pipeline);
      cmd.bind_descriptors_compute(0, {set}, {});
      cmd.push_constants_compute(0, sizeof(pc), &pc);
      cmd.dispatch(ext.width/8, ext.height/4, 1);
    });
}

void ScreenSpaceTrace::add_accumulate_pass(rendergraph::RenderGraph &graph, const ScreenTraceParams &params,
  rendergraph::ImageResourceId depth, rendergraph::ImageResourceId prev_depth)
{
  struct Views { rendergraph::ImageViewId depth, prev_depth, filtered, accum; };
  const vkr_screen_trace_accum_push pc {params.fovy, params.aspect, params.znear, params.zfar};

  graph.add_task<Views>("ScreenTraceAccumulate",
    [&](Views &v, rendergraph::RenderGraphBuilder &builder) {
      v.depth = depth_mip0(builder, depth);
      v.prev_depth = depth_mip0(builder, prev_depth);
      v.filtered = builder.sample_image(filtered, CS);
      v.accum = builder.use_storage_image(accumulated, CS, 0, 0);
    },
    [=](Views &v, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      auto set = resources.allocate_set(accum_pipeline, 0);
      gpu::write_set(set,
        gpu::TextureBinding {0, resources.get_view(v.depth), sampler},
        gpu::TextureBinding {1, resources.get_view(v.prev_depth), sampler},
        gpu::TextureBinding {2, resources.get_view(v.filtered), sampler},
        gpu::StorageTextureBinding {3, resources.get_view(v.accum)});

      const auto ext = resources.get_image(v.accum)->get_extent();
      cmd.bind_pipeline(accum_pipeline);
      cmd.bind_descriptors_compute(0, {set}, {});
      cmd.push_constants_compute(0, sizeof(pc), &pc);
      cmd.dispatch(ext.width/8, ext.height/4, 1);
    });
}
