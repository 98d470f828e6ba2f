// gpu/gpu.cpp — implementation of the HIP-backed gpu:: layer and the program table that maps
// the reference's program names (src/shaders/config.json) onto C-ABI entry points.
#include "gpu.hpp"
#include <dlfcn.h>

#include <hip/hip_runtime_api.h>

#include <cmath>

#include <map>
#include <mutex>

namespace gpu {

// ---- allocation --------------------------------------------------------------------------------
namespace {
void* default_alloc(size_t bytes, void*) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) throw std::runtime_error{std::string{"hipMalloc failed: "} + hipGetErrorString(e)};
  return p;
}
void default_free(void* p, void*) { (void)hipFree(p); }
AllocFn g_alloc = default_alloc;
FreeFn g_free = default_free;
void* g_alloc_user = nullptr;
}  // namespace

void set_device_allocator(AllocFn alloc, FreeFn free_fn, void* user) {
  g_alloc = alloc ? alloc : default_alloc;
  g_free = free_fn ? free_fn : default_free;
  g_alloc_user = user;
}
void* device_alloc(size_t bytes) {
  void* p = g_alloc(bytes, g_alloc_user);
  if (!p) throw std::runtime_error{"device allocation failed"};
  return p;
}
void device_free(void* ptr) { if (ptr) g_free(ptr, g_alloc_user); }

namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    // rocprofv3 --marker-trace records the ranges of rocprofiler-sdk's ROCTx; libroctx64 is roctracer's (rocprof v1 / v2)
    for (const char* n : {"librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so.4"}) {
      if (void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) {
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (push && pop) return;
        push = nullptr; pop = nullptr;
      }
    }
  }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
void trace_push(const char* name) { if (roctx().push) roctx().push(name); }
void trace_pop() { if (roctx().pop) roctx().pop(); }

void check_status(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error{std::string{what} + ": " + vkr_last_error() + " (code " + std::to_string(rc) + ")"};
}

// ---- images ----------------------------------------------------------------------------------------
uint32_t to_vkr_format(VkFormat fmt) {
  switch (fmt) {
    case VK_FORMAT_D24_UNORM_S8_UINT: return VKR_FMT_D24_UNORM_S8;
    case VK_FORMAT_R16G16_UNORM: return VKR_FMT_RG16_UNORM;
    case VK_FORMAT_R16G16_SFLOAT: return VKR_FMT_RG16_SFLOAT;
    case VK_FORMAT_R8G8B8A8_SRGB: return VKR_FMT_RGBA8_SRGB;
    case VK_FORMAT_R8G8B8A8_UNORM: return VKR_FMT_RGBA8_UNORM;
    case VK_FORMAT_R16G16B16A16_UNORM: return VKR_FMT_RGBA16_UNORM;
    case VK_FORMAT_R16G16B16A16_SFLOAT: return VKR_FMT_RGBA16_SFLOAT;
    case VK_FORMAT_R16_SFLOAT: return VKR_FMT_R16_SFLOAT;
    case VK_FORMAT_R32_SFLOAT: return VKR_FMT_R32_SFLOAT;
    case VK_FORMAT_R8_UNORM: return VKR_FMT_R8_UNORM;
    case VK_FORMAT_R32G32B32A32_SFLOAT: return VKR_FMT_RGBA32_SFLOAT;
    default: throw std::runtime_error{"Unsupported image format on the post-process path"};
  }
}

static inline uint32_t mip_dim(uint32_t v, uint32_t i) { uint32_t r = v >> i; return r ? r : 1u; }

Image::Image(const ImageInfo& i, const FrameWindow& w) : info{i}, window{w} {
  if (info.mip_levels == 0 || info.mip_levels > VKR_MAX_MIPS) throw std::runtime_error{"Image: bad mip count"};
  if (info.width == 0 || info.height == 0) throw std::runtime_error{"Image: empty extent"};
  const uint32_t bpp = vkr_format_bytes(to_vkr_format(info.format));
  uint64_t off = 0;
  const uint32_t layers = info.array_layers ? info.array_layers : 1;
  for (uint32_t m = 0; m < info.mip_levels; m++) {
    pitch[m] = (mip_dim(info.width, m) * bpp + 255u) & ~255u;  // rows 256-B aligned
    offset[m] = off;
    off += (uint64_t(pitch[m]) * mip_dim(info.height, m) * layers + 255u) & ~uint64_t(255);
  }
  bytes = off;
  base = device_alloc(bytes);
  if (window.full_width == 0) { window.full_width = info.width; window.full_height = info.height; }
}
Image::~Image() { device_free(base); }

vkr_img Image::describe(uint32_t base_mip, uint32_t count) const {
  if (base_mip + count > info.mip_levels || count == 0) throw std::runtime_error{"Image view outside the mip chain"};
  vkr_img d{};
  d.base = (uint8_t*)base + offset[base_mip];
  d.format = to_vkr_format(info.format);
  d.mip_count = count;
  d.width = mip_dim(info.width, base_mip);
  d.height = mip_dim(info.height, base_mip);
  d.full_width = mip_dim(window.full_width, base_mip);
  d.full_height = mip_dim(window.full_height, base_mip);
  d.origin_x = window.origin_x >> base_mip;
  d.origin_y = window.origin_y >> base_mip;
  for (uint32_t m = 0; m < count; m++) {
    d.pitch_bytes[m] = pitch[base_mip + m];
    d.mip_offset[m] = offset[base_mip + m] - offset[base_mip];
  }
  return d;
}

void Image::set_store_rows(uint32_t row0, uint32_t rows) {
  if (rows && (info.mip_levels != 1 || get_array_layers() != 1 || uint64_t(row0) + rows > info.height))
    throw std::runtime_error{"Image::set_store_rows: rows outside the window, or not a single-mip image"};
  store_row0 = rows ? row0 : 0;
  store_rows = rows;
}
vkr_img Image::describe_store(uint32_t base_mip, uint32_t count) const {
  vkr_img d = describe(base_mip, count);
  if (store_rows) {  // (single-mip image: base_mip = 0, count = 1)
    d.base = (uint8_t*)d.base + uint64_t(store_row0) * d.pitch_bytes[0];
    d.origin_y += (int32_t)store_row0;
    d.height = store_rows;
  }
  return d;
}

vkr_img Image::describe_layer(uint32_t layer) const {
  if (layer >= get_array_layers() || info.mip_levels != 1) throw std::runtime_error{"Image layer view outside the array"};
  vkr_img d = describe(0, 1);
  d.base = (uint8_t*)d.base + uint64_t(pitch[0]) * info.height * layer;
  return d;
}

void Image::upload_mip(uint32_t mip, const void* rows) {
  if (mip >= info.mip_levels) throw std::runtime_error{"Image upload outside the mip chain"};
  const uint32_t bpp = vkr_format_bytes(to_vkr_format(info.format));
  const uint32_t w = mip_dim(info.width, mip), h = mip_dim(info.height, mip);
  hipError_t e = hipMemcpy2D((uint8_t*)base + offset[mip], pitch[mip], rows, size_t(w) * bpp, size_t(w) * bpp, h, hipMemcpyHostToDevice);
  if (e != hipSuccess) throw std::runtime_error{std::string{"image upload failed: "} + hipGetErrorString(e)};
}

Buffer::Buffer(VmaMemoryUsage memory, uint64_t sz, VkBufferUsageFlags usage) : size{sz} {
  dev = device_alloc((sz + 255) & ~uint64_t(255));
  // host shadow: mapped (CPU_TO_GPU) buffers, and uniform buffers — their contents become kernel
  // arguments of the C-ABI call, so the program reads them on the host
  if (memory != VMA_MEMORY_USAGE_GPU_ONLY || (usage & VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT)) shadow.resize(sz);
}
Buffer::~Buffer() { device_free(dev); }
void* Buffer::device_ptr(void* stream) {
  if (dirty && !shadow.empty()) {
    hipError_t e = hipMemcpyAsync(dev, shadow.data(), size, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) throw std::runtime_error{std::string{"buffer upload failed: "} + hipGetErrorString(e)};
    e = hipStreamSynchronize((hipStream_t)stream);  // the shadow is pageable; happens once per write
    if (e != hipSuccess) throw std::runtime_error{std::string{"buffer upload failed: "} + hipGetErrorString(e)};
    dirty = false;
  }
  return dev;
}
BufferPtr create_buffer(VmaMemoryUsage memory, uint64_t size, VkBufferUsageFlags usage) { return std::make_shared<Buffer>(memory, size, usage); }

// ---- samplers ----------------------------------------------------------------------------------------
namespace { std::vector<std::unique_ptr<VkSamplerCreateInfo>> g_samplers; std::mutex g_sampler_lock; }
VkSampler create_sampler(const VkSamplerCreateInfo& info) {
  std::lock_guard<std::mutex> lock{g_sampler_lock};
  for (auto& s : g_samplers)
    if (std::memcmp(s.get(), &info, sizeof(info)) == 0) return (VkSampler)s.get();
  g_samplers.emplace_back(new VkSamplerCreateInfo(info));
  return (VkSampler)g_samplers.back().get();
}
const VkSamplerCreateInfo& sampler_info(VkSampler s) { return *(const VkSamplerCreateInfo*)s; }

// ---- descriptor writes -----------------------------------------------------------------------------------
static SetSlot& slot_of(VkDescriptorSet set, uint32_t binding) {
  if (!set) throw std::runtime_error{"write_set: null descriptor set"};
  auto* obj = (DescriptorSetObject*)set;
  if (binding >= obj->slots.size()) throw std::runtime_error{"write_set: binding out of range"};
  return obj->slots[binding];
}
static const ImageViewObject& view_of(VkImageView v) {
  if (!v) throw std::runtime_error{"write_set: null image view"};
  return *(const ImageViewObject*)v;
}
void write_binding(VkDescriptorSet, const AccelerationStructBinding&) {
  throw std::runtime_error{"Acceleration structures are not supported on the post-process path (ray-query AO is out of scope)"};
}
void write_binding(VkDescriptorSet set, const TextureBinding& b) {
  auto& s = slot_of(set, b.binding);
  s = SetSlot{};
  s.kind = SetSlot::Texture; s.view = view_of(b.view); s.sampler = b.sampler;
}
void write_binding(VkDescriptorSet set, const StorageTextureBinding& b) {
  auto& s = slot_of(set, b.binding);
  s = SetSlot{};
  s.kind = SetSlot::StorageTexture; s.view = view_of(b.view);
}
void write_binding(VkDescriptorSet set, const UBOBinding& b) {
  auto& s = slot_of(set, b.binding);
  s = SetSlot{};
  s.kind = SetSlot::Ubo; s.host_data = b.host; s.host_size = b.size; s.buffer = b.buffer;
}
void write_binding(VkDescriptorSet set, const SSBOBinding& b) {
  auto& s = slot_of(set, b.binding);
  s = SetSlot{};
  s.kind = SetSlot::Ssbo; s.buffer = b.buffer;
}

namespace { std::vector<std::unique_ptr<DescriptorSetObject>> g_long_lived_sets; std::mutex g_set_lock; }
VkDescriptorSet allocate_descriptor_set(VkDescriptorSetLayout, const std::initializer_list<uint32_t>&) {
  std::lock_guard<std::mutex> lock{g_set_lock};
  g_long_lived_sets.emplace_back(new DescriptorSetObject{});
  return (VkDescriptorSet)g_long_lived_sets.back().get();
}
void write_binding(VkDescriptorSet set, const ArrayOfImagesBinding& b) {
  if (!set) throw std::runtime_error{"write_set: null descriptor set"};
  auto* obj = (DescriptorSetObject*)set;
  obj->image_array.clear();
  for (const auto& it : b.images) obj->image_array.push_back(view_of(it.first));
}

// ---- program table ----------------------------------------------------------------------------------------
namespace {
std::map<std::string, ProgramFn>& programs() { static std::map<std::string, ProgramFn> p; return p; }

// `nearest_border`: the one non-default sampler of the path (ssr.cpp:21-28: NEAREST, U clamp-to-border)
vkr_img tex(const LaunchState& st, uint32_t slot, SetSlot::Kind kind, const char* prog, bool nearest_border = false) {
  const SetSlot& s = st.set ? st.set->slots[slot] : SetSlot{};
  if (!st.set || s.kind != kind || !s.view.image)
    throw std::runtime_error{std::string{prog} + ": binding " + std::to_string(slot) + " is not bound as expected"};
  if (kind == SetSlot::Texture) {
    if (!s.sampler) throw std::runtime_error{std::string{prog} + ": binding " + std::to_string(slot) + " has no sampler"};
    const auto& si = sampler_info(s.sampler);
    const bool is_default = si.magFilter == VK_FILTER_LINEAR && si.addressModeU == VK_SAMPLER_ADDRESS_MODE_CLAMP_TO_EDGE;
    const bool is_nearest_border = si.magFilter == VK_FILTER_NEAREST && si.minFilter == VK_FILTER_NEAREST &&
                                   si.addressModeU == VK_SAMPLER_ADDRESS_MODE_CLAMP_TO_BORDER && si.addressModeV == VK_SAMPLER_ADDRESS_MODE_CLAMP_TO_EDGE;
    if (nearest_border ? !is_nearest_border : !is_default)
      throw std::runtime_error{std::string{prog} + ": binding " + std::to_string(slot) + ": sampler not implemented on this path"};
  }
  if (kind == SetSlot::StorageTexture) return s.view.image->describe_store(s.view.range.base_mip, s.view.range.mips_count);
  return s.view.image->describe(s.view.range.base_mip, s.view.range.mips_count);
}
// The Halton(2,3) UBO of AdvancedSSR (advanced_ssr.cpp:54-58: 128 x vec4, xy filled, zw = 0).  The HIP programs want
// cos / sin(2 PI y) in zw (evaluated on the host once instead of per ray, vkr_halton23_fill); a table that arrives as
// the reference builds it is completed here, in the buffer's host shadow, before it is uploaded.
const float* halton_table(Buffer* b, void* stream) {
  const float* h = (const float*)b->host_data();
  const size_t n = b->get_size() / 16;
  bool raw = h != nullptr && n > 0;
  for (size_t i = 0; raw && i < n; i++) raw = h[4 * i + 2] == 0.0f && h[4 * i + 3] == 0.0f;
  if (raw) {
    float* w = (float*)b->get_mapped_ptr();  // marks the shadow dirty: re-uploaded by device_ptr() below
    const float PI = 3.1415926535897932384626433832795f;
    for (size_t i = 0; i < n; i++) {
      const float phi = (2.0f * PI) * w[4 * i + 1];
      w[4 * i + 2] = (float)std::cos((double)phi);
      w[4 * i + 3] = (float)std::sin((double)phi);
    }
  }
  return (const float*)b->device_ptr(stream);
}

template <typename T> const T* ubo(const LaunchState& st, uint32_t slot, const char* prog) {
  const SetSlot& s = st.set ? st.set->slots[slot] : SetSlot{};
  if (!st.set || s.kind != SetSlot::Ubo) throw std::runtime_error{std::string{prog} + ": uniform block " + std::to_string(slot) + " is not bound"};
  const void* data = s.host_data ? s.host_data : (s.buffer ? s.buffer->host_data() : nullptr);
  const uint64_t size = s.host_data ? s.host_size : (s.buffer ? s.buffer->get_size() : 0);
  if (!data || size < sizeof(T)) throw std::runtime_error{std::string{prog} + ": uniform block " + std::to_string(slot) + " is not bound"};
  return (const T*)data;
}
template <typename T> const T* push(const LaunchState& st, const char* prog) {
  if (st.push_size < sizeof(T)) throw std::runtime_error{std::string{prog} + ": push constants missing"};
  return (const T*)st.push;
}
}  // namespace

void create_program(const std::string& name, ProgramFn fn) { programs()[name] = std::move(fn); }
bool has_program(const std::string& name) { return programs().count(name) != 0; }

void register_hot_path_programs() {
  static std::once_flag once;
  std::call_once(once, [] {
    const auto T = SetSlot::Texture, S = SetSlot::StorageTexture;
    // advanced_ssr/downsample_gbuffer.frag: set {0 depth, 1 normal, 2 velocity}; attachments {normal, velocity, depth mip+1}
    create_program("downsample_gbuffer", [=](LaunchState& st) {
      const char* P = "downsample_gbuffer";
      const SetSlot& ds = st.set->slots[0];
      if (st.attachments.size() != 3) throw std::runtime_error{"downsample_gbuffer: expects 2 colour attachments + depth"};
      const ImageViewObject& dout = st.attachments[2];
      if (ds.kind != T || ds.view.image != dout.image || dout.range.base_mip != ds.view.range.base_mip + 1)
        throw std::runtime_error{"downsample_gbuffer: depth attachment must be the next mip of the sampled depth"};
      vkr_img depth = ds.view.image->describe(ds.view.range.base_mip, 2);
      vkr_img n = tex(st, 1, T, P), v = tex(st, 2, T, P);
      vkr_img on = st.attachments[0].image->describe(st.attachments[0].range.base_mip, 1);
      vkr_img ov = st.attachments[1].image->describe(st.attachments[1].range.base_mip, 1);
      return vkr_downsample_gbuffer(&depth, &n, &v, &on, &ov, st.stream);
    });
    // advanced_ssr/depth_mips.frag: set {0 depth mip i-1}; attachments {depth mip i, i+1, ...}.
    // The reference draws once per mip; bound with several consecutive mips the fused kernel
    // builds the whole run in one go (each mip still the 2x2 min of its parent).
    create_program("depth_mips", [=](LaunchState& st) {
      const SetSlot& ds = st.set->slots[0];
      if (st.attachments.empty()) throw std::runtime_error{"depth_mips: expects at least one depth attachment"};
      if (ds.kind != T) throw std::runtime_error{"depth_mips: binding 0 is not bound as expected"};
      for (size_t i = 0; i < st.attachments.size(); i++) {
        const ImageViewObject& dout = st.attachments[i];
        if (ds.view.image != dout.image || dout.range.base_mip != ds.view.range.base_mip + 1 + i)
          throw std::runtime_error{"depth_mips: attachments must be the consecutive mips after the sampled depth"};
      }
      vkr_img depth = ds.view.image->describe(ds.view.range.base_mip, 1 + (uint32_t)st.attachments.size());
      return vkr_depth_mips(&depth, 0, st.stream);
    });
    // Programs the reference's constructors name but this path does not implement (out of scope):
    // known to the table, so that constructing the pass works as it does in the reference; launching one throws.
    for (const char* name : {"tile_regression", "gtao_rt_main"})
      create_program(name, [name](LaunchState&) -> int { throw std::runtime_error{std::string{name} + ": program is not implemented on the post-process path"}; });
    create_program("pdf_preintegrate", [=](LaunchState& st) {
      vkr_img out = tex(st, 0, S, "pdf_preintegrate");
      return vkr_pdf_preintegrate(&out, st.stream);
    });
    create_program("sssr_trace", [=](LaunchState& st) {
      const char* P = "sssr_trace";
      vkr_img depth = tex(st, 0, T, P), normal = tex(st, 1, T, P), material = tex(st, 2, T, P);
      vkr_img rays = tex(st, 5, S, P), occ = tex(st, 6, S, P), pdf = tex(st, 7, T, P);
      const SetSlot& h = st.set->slots[4];
      if (h.kind != SetSlot::Ubo || !h.buffer) throw std::runtime_error{"sssr_trace: Halton buffer (binding 4) is not bound"};
      // Two launches where they pay: the rays still marching after step 48 (two of the four compacted rounds: a tenth of them) are
      // parked in the context's workspace and finished by the resume launch, 256 rays of many tiles per block.  Same images bit
      // for bit.  Measured on one MI355X (tools/trace_split_probe.py): 3840x2160 0.259 against 0.272 ms (park after one round
      // 0.280, after three 0.268); 1920x1080 0.105 against 0.094 (a second launch and a memset on a 0.1 ms pass);
      // 7680x4320 1.008 against 1.004; 15360x8640 4.196 against 4.007 — the 80-byte records of the parked rays (17 MB at 4K,
      // 270 MB at 15360x8640) must stay in the caches between the two launches for the resume launch to be cheap.
      const uint64_t n_rays = uint64_t(rays.width) * rays.height;
      if ((vkr_get_switches() & VKR_SWITCH_TRACE_ONE_LAUNCH) || n_rays < (1ull << 20) || n_rays > (6ull << 20))
        return vkr_sssr_trace(&depth, &normal, &material, ubo<vkr_trace_params>(st, 3, P), halton_table(h.buffer.get(), st.stream),
                              &rays, &occ, &pdf, push<vkr_trace_push>(st, P), st.stream);
      const uint64_t bytes = vkr_sssr_trace_workspace_bytes(rays.width, rays.height);
      return vkr_sssr_trace_split(&depth, &normal, &material, ubo<vkr_trace_params>(st, 3, P), halton_table(h.buffer.get(), st.stream),
                                  &rays, &occ, &pdf, push<vkr_trace_push>(st, P), st.require_workspace(bytes), bytes, 2u, st.stream);
    });
    create_program("sssr_trace_windowed", [=](LaunchState& st) {  // multi-GPU variant (include/vkr_postfx.h), bindings 0..9
      const char* P = "sssr_trace_windowed";
      vkr_img depth = tex(st, 0, T, P), normal = tex(st, 1, T, P), material = tex(st, 2, T, P);
      vkr_img rays = tex(st, 5, S, P), occ = tex(st, 6, S, P), pdf = tex(st, 7, T, P), mask = tex(st, 8, S, P), data = tex(st, 9, S, P);
      const SetSlot& h = st.set->slots[4];
      if (h.kind != SetSlot::Ubo || !h.buffer) throw std::runtime_error{"sssr_trace_windowed: Halton buffer (binding 4) is not bound"};
      return vkr_sssr_trace_windowed(&depth, &normal, &material, ubo<vkr_trace_params>(st, 3, P), halton_table(h.buffer.get(), st.stream),
                                     &rays, &occ, &pdf, &mask, &data, push<vkr_trace_window_push>(st, P), st.stream);
    });
    // the windowed trace in two tasks (include/vkr_postfx.h): bindings of sssr_trace_windowed, the head with the window's own
    // pyramid levels at 0 and the whole-frame pyramid (extents only) at 10; both use the context's workspace
    create_program("sssr_trace_windowed_head", [=](LaunchState& st) {
      const char* P = "sssr_trace_windowed_head";
      vkr_img local = tex(st, 0, T, P), frame = tex(st, 10, T, P), normal = tex(st, 1, T, P), material = tex(st, 2, T, P);
      vkr_img rays = tex(st, 5, S, P), occ = tex(st, 6, S, P), pdf = tex(st, 7, T, P), mask = tex(st, 8, S, P), data = tex(st, 9, S, P);
      const SetSlot& h = st.set->slots[4];
      if (h.kind != SetSlot::Ubo || !h.buffer) throw std::runtime_error{"sssr_trace_windowed_head: Halton buffer (binding 4) is not bound"};
      const uint64_t bytes = vkr_sssr_trace_workspace_bytes(rays.width, rays.height);
      return vkr_sssr_trace_windowed_head(&local, &frame, &normal, &material, ubo<vkr_trace_params>(st, 3, P), halton_table(h.buffer.get(), st.stream),
                                          &rays, &occ, &pdf, &mask, &data, push<vkr_trace_window_push>(st, P), st.require_workspace(bytes), bytes, 2u, st.stream);
    });
    create_program("sssr_trace_windowed_resume", [=](LaunchState& st) {
      const char* P = "sssr_trace_windowed_resume";
      vkr_img depth = tex(st, 0, T, P), normal = tex(st, 1, T, P), material = tex(st, 2, T, P);
      vkr_img rays = tex(st, 5, S, P), occ = tex(st, 6, S, P), pdf = tex(st, 7, T, P), mask = tex(st, 8, S, P), data = tex(st, 9, S, P);
      const SetSlot& h = st.set->slots[4];
      if (h.kind != SetSlot::Ubo || !h.buffer) throw std::runtime_error{"sssr_trace_windowed_resume: Halton buffer (binding 4) is not bound"};
      const uint64_t bytes = vkr_sssr_trace_workspace_bytes(rays.width, rays.height);
      if (st.workspace_bytes < bytes) throw std::runtime_error{"sssr_trace_windowed_resume: no head launch has filled the workspace"};
      return vkr_sssr_trace_windowed_resume(&depth, &normal, &material, ubo<vkr_trace_params>(st, 3, P), halton_table(h.buffer.get(), st.stream),
                                            &rays, &occ, &pdf, &mask, &data, push<vkr_trace_window_push>(st, P), st.workspace, bytes, st.stream);
    });
    create_program("sssr_filter", [=](LaunchState& st) {
      const char* P = "sssr_filter";
      vkr_img rays = tex(st, 0, T, P), depth = tex(st, 1, T, P), albedo = tex(st, 2, T, P), normal = tex(st, 3, T, P);
      vkr_img material = tex(st, 4, T, P), out = tex(st, 5, S, P);
      return vkr_sssr_filter(&rays, &depth, &albedo, &normal, &material, &out, ubo<vkr_trace_params>(st, 6, P),
                             push<vkr_filter_push>(st, P), st.stream);
    });
    create_program("sssr_blur", [=](LaunchState& st) {
      const char* P = "sssr_blur";
      vkr_img depth = tex(st, 0, T, P), normal = tex(st, 1, T, P), refl = tex(st, 2, T, P), material = tex(st, 3, T, P);
      vkr_img history = tex(st, 4, T, P), velocity = tex(st, 5, T, P), hdepth = tex(st, 6, T, P), out = tex(st, 7, S, P);
      return vkr_sssr_blur(&depth, &normal, &refl, &material, &history, &velocity, &hdepth, &out,
                           ubo<vkr_reproject_params>(st, 8, P), push<vkr_blur_push>(st, P), st.stream);
    });
    create_program("gtao_compute_main", [=](LaunchState& st) {
      const char* P = "gtao_compute_main";
      vkr_img depth = tex(st, 0, T, P), normal = tex(st, 2, T, P), material = tex(st, 3, T, P), pdf = tex(st, 4, T, P);
      vkr_img out = tex(st, 5, S, P);
      return vkr_gtao_main(&depth, ubo<vkr_gtao_params>(st, 1, P), &normal, &material, &pdf, &out, push<vkr_gtao_push>(st, P), st.stream);
    });
    create_program("gtao_filter", [=](LaunchState& st) {
      const char* P = "gtao_filter";
      vkr_img depth = tex(st, 0, T, P), raw = tex(st, 1, T, P), out = tex(st, 2, S, P);
      return vkr_gtao_filter(&depth, &raw, &out, push<vkr_gtao_filter_push>(st, P), st.stream);
    });
    create_program("gtao_accumulate", [=](LaunchState& st) {
      const char* P = "gtao_accumulate";
      vkr_img depth = tex(st, 0, T, P), pdepth = tex(st, 1, T, P), ao = tex(st, 2, T, P), out = tex(st, 3, S, P);
      vkr_img velocity = tex(st, 4, T, P), history = tex(st, 5, T, P);
      return vkr_gtao_accumulate(&depth, &pdepth, &ao, &out, &velocity, &history, ubo<vkr_gtao_accum_params>(st, 6, P),
                                 push<vkr_gtao_accum_push>(st, P), st.stream);
    });
    create_program("taa_resolve", [=](LaunchState& st) {
      const char* P = "taa_resolve";
      vkr_img hist = tex(st, 0, T, P), hdepth = tex(st, 1, T, P), depth = tex(st, 2, T, P), velocity = tex(st, 3, T, P);
      vkr_img color = tex(st, 4, T, P), out = tex(st, 5, S, P);
      return vkr_taa_resolve(&hist, &hdepth, &depth, &velocity, &color, &out, ubo<vkr_reproject_params>(st, 6, P), st.stream);
    });
    // ssr/shader.frag: set {0 normal, 1 depth (nearest/border sampler), 2 frame, 3 SSRParams, 4 material}; attachment {out}
    create_program("ssr", [=](LaunchState& st) {
      const char* P = "ssr";
      if (st.attachments.size() != 1) throw std::runtime_error{"ssr: expects one colour attachment"};
      vkr_img normal = tex(st, 0, T, P), depth = tex(st, 1, T, P, true), frame = tex(st, 2, T, P), material = tex(st, 4, T, P);
      vkr_img out = st.attachments[0].image->describe(st.attachments[0].range.base_mip, 1);
      return vkr_ssr(&normal, &depth, &frame, ubo<vkr_ssr_params>(st, 3, P), &material, &out, st.stream);
    });
    create_program("brdf_preintegrate", [=](LaunchState& st) {
      const SetSlot& h = st.set->slots[0];
      if (h.kind != SetSlot::Ubo || !h.buffer) throw std::runtime_error{"brdf_preintegrate: Halton buffer (binding 0) is not bound"};
      vkr_img out = tex(st, 1, S, "brdf_preintegrate");
      return vkr_brdf_preintegrate(halton_table(h.buffer.get(), st.stream), &out, st.stream);
    });
    // defered_shading/shader.frag: set {0 albedo, 1 normal, 2 material, 3 depth, 4 Constants, 5 shadow (unused), 6 occlusion, 7 brdf, 8 reflections}
    create_program("defered_shading", [=](LaunchState& st) {
      const char* P = "defered_shading";
      if (st.attachments.size() != 1) throw std::runtime_error{"defered_shading: expects one colour attachment"};
      vkr_img albedo = tex(st, 0, T, P), normal = tex(st, 1, T, P), material = tex(st, 2, T, P), depth = tex(st, 3, T, P);
      vkr_img occlusion = tex(st, 6, T, P), brdf = tex(st, 7, T, P), refl = tex(st, 8, T, P);
      vkr_img out = st.attachments[0].image->describe(st.attachments[0].range.base_mip, 1);
      return vkr_defered_shading(&albedo, &normal, &material, &depth, ubo<vkr_shading_params>(st, 4, P), &occlusion, &brdf, &refl, &out,
                                 push<vkr_shading_push>(st, P), st.stream);
    });
    // ---- tile-classified trace ----
    auto ssbo = [](const LaunchState& st, uint32_t slot, const char* prog) -> Buffer* {
      const SetSlot& s = st.set ? st.set->slots[slot] : SetSlot{};
      if (!st.set || s.kind != SetSlot::Ssbo || !s.buffer) throw std::runtime_error{std::string{prog} + ": storage buffer " + std::to_string(slot) + " is not bound"};
      return s.buffer.get();
    };
    create_program("sssr_classification", [=](LaunchState& st) {
      const char* P = "sssr_classification";
      vkr_img material = tex(st, 0, T, P);
      return vkr_sssr_classification(&material, (int32_t*)ssbo(st, 1, P)->device_ptr(st.stream), (int32_t*)ssbo(st, 2, P)->device_ptr(st.stream),
                                     (uint32_t*)ssbo(st, 3, P)->device_ptr(st.stream), (uint32_t*)ssbo(st, 4, P)->device_ptr(st.stream),
                                     push<vkr_classification_push>(st, P), st.stream);
    });
    create_program("sssr_trace_indirect", [=](LaunchState& st) {
      const char* P = "sssr_trace_indirect";
      if (!st.indirect) throw std::runtime_error{"sssr_trace_indirect: must be launched with dispatch_indirect"};
      vkr_img depth = tex(st, 0, T, P), normal = tex(st, 1, T, P), material = tex(st, 2, T, P), rays = tex(st, 5, S, P);
      const SetSlot& h = st.set->slots[4];
      if (h.kind != SetSlot::Ubo || !h.buffer) throw std::runtime_error{"sssr_trace_indirect: Halton buffer (binding 4) is not bound"};
      Buffer* tiles = ssbo(st, 6, P);
      return vkr_sssr_trace_indirect(&depth, &normal, &material, ubo<vkr_trace_params>(st, 3, P), halton_table(h.buffer.get(), st.stream), &rays,
                                     (const int32_t*)tiles->device_ptr(st.stream), (const uint32_t*)st.indirect->device_ptr(st.stream),
                                     (uint32_t)(tiles->get_size() / sizeof(int32_t)), push<vkr_trace_indirect_push>(st, P), st.stream);
    });
    // ---- G-buffer raster stage: gbuf/opaque_taa.{vert,frag} ----
    // set 0 {0 GbufConst, 1 transforms (pairs of mat4: model, normal)}, set 1 {0 material textures[]}; vertex + index
    // buffers; one recorded draw_indexed per primitive with PushData {transform, albedo, mr, flags}; attachments
    // {albedo, normal, material, velocity, depth}, cleared.
    create_program("gbuf_opaque_taa", [=](LaunchState& st) {
      const char* P = "gbuf_opaque_taa";
      if (st.attachments.size() != 5) throw std::runtime_error{"gbuf_opaque_taa: expects 4 colour attachments + depth"};
      if (!st.cleared_color || !st.cleared_depth) throw std::runtime_error{"gbuf_opaque_taa: attachments must be cleared (colour 0, depth 1)"};
      if (!st.vertex_buffer || !st.index_buffer) throw std::runtime_error{"gbuf_opaque_taa: vertex / index buffer not bound"};
      Buffer* transforms = ssbo(st, 1, P);
      if (!transforms->host_data()) throw std::runtime_error{"gbuf_opaque_taa: the transform buffer must be host-visible on this path"};
      auto att = [&](size_t i) { return st.attachments[i].image->describe(st.attachments[i].range.base_mip, 1); };
      vkr_img albedo = att(0), normal = att(1), material = att(2), velocity = att(3), depth = att(4);
      std::vector<vkr_img> textures;
      if (st.set1)
        for (const auto& v : st.set1->image_array) textures.push_back(v.image->describe(v.range.base_mip, v.range.mips_count));
      std::vector<vkr_raster_draw> draws;
      for (const auto& d : st.indexed_draws) {
        if (d.push.size() < 16) throw std::runtime_error{"gbuf_opaque_taa: push constants missing"};
        vkr_raster_draw r{};
        std::memcpy(&r, d.push.data(), 16);
        r.index_offset = d.first_index; r.index_count = d.index_count; r.vertex_offset = (uint32_t)d.vertex_offset;
        if (st.set1 && r.albedo_index < st.set1->image_array.size() && st.set1->image_array[r.albedo_index].image->alpha_never_zero)
          r.reserved |= VKR_RASTER_DRAW_OPAQUE_ALBEDO;  // the discard of opaque_taa.frag:32-34 cannot fire
        draws.push_back(r);
      }
      vkr_raster_scene scene{};
      scene.vertices = (const vkr_raster_vertex*)st.vertex_buffer->device_ptr(st.stream);
      scene.vertex_count = (uint32_t)(st.vertex_buffer->get_size() / sizeof(vkr_raster_vertex));
      scene.indices = (const uint32_t*)st.index_buffer->device_ptr(st.stream);
      scene.index_count = (uint32_t)(st.index_buffer->get_size() / sizeof(uint32_t));
      scene.transforms = (const vkr_raster_transform*)transforms->host_data();
      scene.transform_count = (uint32_t)(transforms->get_size() / sizeof(vkr_raster_transform));
      scene.draws = draws.data(); scene.draw_count = (uint32_t)draws.size();
      scene.textures = textures.data(); scene.texture_count = (uint32_t)textures.size();
      return vkr_raster_gbuffer(&scene, ubo<vkr_gbuf_const>(st, 0, P), &albedo, &normal, &material, &velocity, &depth,
                                st.scratch, st.scratch_bytes, st.stream);
    });
    // ---- dormant GTAO variants ----
    // gtao/main.frag: set {0 depth, 1 GTAOParams, 2 normal}; colour attachment raw
    create_program("gtao_main", [=](LaunchState& st) {
      const char* P = "gtao_main";
      if (st.attachments.size() != 1) throw std::runtime_error{"gtao_main: expects one colour attachment"};
      vkr_img depth = tex(st, 0, T, P), normal = tex(st, 2, T, P);
      vkr_img out = st.attachments[0].image->describe(st.attachments[0].range.base_mip, 1);
      return vkr_gtao_main_graphics(&depth, ubo<vkr_gtao_params>(st, 1, P), &normal, &out, push<vkr_gtao_gfx_push>(st, P), st.stream);
    });
    create_program("gtao_reproject", [=](LaunchState& st) {
      const char* P = "gtao_reproject";
      vkr_img depth = tex(st, 1, T, P), pdepth = tex(st, 2, T, P), ao = tex(st, 3, T, P), pao = tex(st, 4, T, P), out = tex(st, 5, S, P);
      return vkr_gtao_reproject(ubo<vkr_gtao_reprojection>(st, 0, P), &depth, &pdepth, &ao, &pao, &out, st.stream);
    });
    auto layers_of = [](const LaunchState& st, uint32_t slot, SetSlot::Kind kind, const char* prog) {
      const SetSlot& s = st.set ? st.set->slots[slot] : SetSlot{};
      if (!st.set || s.kind != kind || !s.view.image)
        throw std::runtime_error{std::string{prog} + ": binding " + std::to_string(slot) + " is not bound as expected"};
      std::vector<vkr_img> layers;
      for (uint32_t l = 0; l < s.view.image->get_array_layers(); l++) layers.push_back(s.view.image->describe_layer(l));
      return layers;
    };
    create_program("deinterleave_depth", [=](LaunchState& st) {
      const char* P = "deinterleave_depth";
      vkr_img depth = tex(st, 0, T, P);
      auto layers = layers_of(st, 1, S, P);
      return vkr_deinterleave_depth(&depth, layers.data(), (uint32_t)layers.size(), push<vkr_deinterleave_push>(st, P), st.stream);
    });
    create_program("main_deinterleaved", [=](LaunchState& st) {
      const char* P = "main_deinterleaved";
      auto layers = layers_of(st, 0, T, P);
      vkr_img normal = tex(st, 2, T, P), out = tex(st, 3, S, P);
      return vkr_gtao_main_deinterleaved(layers.data(), (uint32_t)layers.size(), ubo<vkr_gtao_params>(st, 1, P), &normal, &out,
                                         push<vkr_gtao_deinterleaved_push>(st, P), st.stream);
    });
    // ---- ScreenSpaceTrace (row R2) ----
    create_program("screen_trace_main", [=](LaunchState& st) {
      const char* P = "screen_trace_main";
      vkr_img depth = tex(st, 0, T, P), normal = tex(st, 1, T, P), color = tex(st, 2, T, P), material = tex(st, 3, T, P), out = tex(st, 4, S, P);
      return vkr_screen_trace_main(&depth, &normal, &color, &material, &out, ubo<vkr_screen_trace_params>(st, 5, P), st.stream);
    });
    create_program("screen_trace_filter", [=](LaunchState& st) {
      const char* P = "screen_trace_filter";
      vkr_img raw = tex(st, 0, T, P), depth = tex(st, 1, T, P), out = tex(st, 2, S, P);
      return vkr_screen_trace_filter(&raw, &depth, &out, push<vkr_screen_trace_filter_push>(st, P), st.stream);
    });
    create_program("screen_trace_accumulate", [=](LaunchState& st) {
      const char* P = "screen_trace_accumulate";
      vkr_img depth = tex(st, 0, T, P), pdepth = tex(st, 1, T, P), cur = tex(st, 2, T, P), acc = tex(st, 3, S, P);
      return vkr_screen_trace_accumulate(&depth, &pdepth, &cur, &acc, push<vkr_screen_trace_accum_push>(st, P), st.stream);
    });
    // synthetic G-buffer "raster" program: attachments {albedo, normal, material, velocity, depth} or {depth}
    create_program("synthetic_gbuffer", [=](LaunchState& st) {
      const vkr_synth_params* p = ubo<vkr_synth_params>(st, 0, "synthetic_gbuffer");
      auto att = [&](size_t i) { return st.attachments[i].image->describe(st.attachments[i].range.base_mip, 1); };
      if (st.attachments.size() == 1) {
        vkr_img d = att(0);
        return vkr_synth_gbuffer(&d, nullptr, nullptr, nullptr, nullptr, p, st.stream);
      }
      if (st.attachments.size() != 5) throw std::runtime_error{"synthetic_gbuffer: expects 4 colour attachments + depth"};
      vkr_img a = att(0), n = att(1), m = att(2), v = att(3), d = att(4);
      return vkr_synth_gbuffer(&d, &n, &a, &m, &v, p, st.stream);
    });
  });
}

void BasePipeline::set_program(const std::string& name) {
  register_hot_path_programs();
  if (!gpu::has_program(name)) throw std::runtime_error{"Program not found"};
  program = name;
}
ComputePipeline create_compute_pipeline() { return ComputePipeline{}; }
ComputePipeline create_compute_pipeline(const char* name) { ComputePipeline p; p.set_program(name); return p; }
GraphicsPipeline create_graphics_pipeline() { return GraphicsPipeline{}; }

// ---- command context ---------------------------------------------------------------------------------------
void CmdContext::begin() {
  if (!staged_updates.empty()) {  // their async copies must have drained before the storage goes away
    (void)hipStreamSynchronize((hipStream_t)stream);
    staged_updates.clear();
  }
  ubo_pool.reset();
  sets.clear();
}
VkDescriptorSet CmdContext::allocate_set() {
  sets.emplace_back(new DescriptorSetObject{});
  return (VkDescriptorSet)sets.back().get();
}
void CmdContext::bind_pipeline(const ComputePipeline& p) {
  if (!p.has_program()) throw std::runtime_error{"Pipeline without program"};
  bound_program = p.program_name();
}
void CmdContext::bind_pipeline(const GraphicsPipeline& p) {
  if (!p.has_program()) throw std::runtime_error{"Pipeline without program"};
  bound_program = p.program_name();
}
void CmdContext::bind_sets(uint32_t first_set, const std::initializer_list<VkDescriptorSet>& s) {
  if (first_set > 1 || s.size() != 1) throw std::runtime_error{"Only descriptor sets 0 and 1 are used on this path"};
  (first_set == 0 ? state.set : state.set1) = (const DescriptorSetObject*)*s.begin();
}
CmdContext::~CmdContext() { device_free(state.scratch); device_free(state.workspace); }
void* LaunchState::require_workspace(uint64_t bytes) {
  if (bytes > workspace_bytes) {
    if (workspace) { (void)hipStreamSynchronize((hipStream_t)stream); device_free(workspace); }
    workspace = device_alloc(bytes);
    workspace_bytes = bytes;
  }
  return workspace;
}
void* CmdContext::require_scratch(uint64_t bytes) {
  if (bytes > state.scratch_bytes) {
    if (state.scratch) { (void)hipStreamSynchronize((hipStream_t)stream); device_free(state.scratch); }
    state.scratch = device_alloc(bytes);
    state.scratch_bytes = bytes;
  }
  return state.scratch;
}
void CmdContext::clear_color_attachments(float r, float g, float b, float a) {
  if (r != 0.f || g != 0.f || b != 0.f || a != 0.f) throw std::runtime_error{"Only a clear to 0 is implemented on this path"};
  state.cleared_color = true;
}
void CmdContext::clear_depth_attachment(float depth) {
  if (depth != 1.f) throw std::runtime_error{"Only a depth clear to 1 is implemented on this path"};
  state.cleared_depth = true;
}
void CmdContext::bind_vertex_buffers(uint32_t first, const std::initializer_list<VkBuffer>& buffers, const std::initializer_list<uint64_t>& offsets) {
  if (first != 0 || buffers.size() != 1 || (offsets.size() && *offsets.begin() != 0)) throw std::runtime_error{"One vertex buffer at offset 0 is implemented on this path"};
  state.vertex_buffer = (Buffer*)*buffers.begin();
}
void CmdContext::bind_index_buffer(VkBuffer buffer, uint64_t offset, VkIndexType type) {
  if (offset != 0 || type != VK_INDEX_TYPE_UINT32) throw std::runtime_error{"A uint32 index buffer at offset 0 is implemented on this path"};
  state.index_buffer = (Buffer*)buffer;
}
void CmdContext::draw_indexed(uint32_t index_count, uint32_t instance_count, uint32_t first_index, int32_t vertex_offset, uint32_t) {
  if (instance_count != 1) throw std::runtime_error{"Instanced draws are not implemented on this path"};
  state.indexed_draws.push_back(LaunchState::IndexedDraw{push_data, index_count, first_index, vertex_offset});
}
void CmdContext::end_renderpass() {
  if (!state.indexed_draws.empty()) {  // the recorded geometry is one pass of the bound raster program
    uint32_t triangles = 0;
    for (const auto& d : state.indexed_draws) triangles += d.index_count / 3u;
    require_scratch(vkr_raster_scratch_bytes(state.fb_width, state.fb_height, triangles));
    launch();
  }
  state.indexed_draws.clear();
  state.attachments.clear();
  state.set1 = nullptr;
  state.vertex_buffer = state.index_buffer = nullptr;
  state.cleared_color = state.cleared_depth = false;
}
void CmdContext::push_constants_compute(uint32_t offset, uint32_t size, const void* constants) {
  if (push_data.size() < offset + size) push_data.resize(offset + size);
  std::memcpy(push_data.data() + offset, constants, size);
}
void CmdContext::set_framebuffer(uint32_t width, uint32_t height, const std::initializer_list<ImageViewObject>& attachments) {
  state.fb_width = width; state.fb_height = height;
  state.attachments.assign(attachments.begin(), attachments.end());
}
void CmdContext::set_framebuffer(uint32_t width, uint32_t height, const std::vector<ImageViewObject>& attachments) {
  state.fb_width = width; state.fb_height = height;
  state.attachments = attachments;
}
void CmdContext::launch() {
  if (!bound_program) throw std::runtime_error{"No pipeline bound"};
  auto it = programs().find(*bound_program);
  if (it == programs().end()) throw std::runtime_error{"Program not found"};
  state.push = push_data.data();
  state.push_size = (uint32_t)push_data.size();
  state.stream = stream;
  int rc = it->second(state);
  push_data.clear();
  state.set = nullptr;
  check_status(rc, bound_program->c_str());
}
void CmdContext::dispatch(uint32_t x, uint32_t y, uint32_t z) {
  state.groups[0] = x; state.groups[1] = y; state.groups[2] = z;
  launch();
}
void CmdContext::dispatch_indirect(VkBuffer arguments) {
  if (!arguments) throw std::runtime_error{"dispatch_indirect: null argument buffer"};
  state.groups[0] = state.groups[1] = state.groups[2] = 0;
  state.indirect = (Buffer*)arguments;
  launch();
  state.indirect = nullptr;
}
void CmdContext::update_buffer_bytes(VkBuffer dst, uint64_t offset, const void* data, uint64_t size) {
  auto* b = (Buffer*)dst;
  if (!b || offset + size > b->get_size() || size > 65536) throw std::runtime_error{"update_buffer: range outside the buffer"};
  // like vkCmdUpdateBuffer the value is captured at record time: stage it in memory that outlives the copy
  staged_updates.emplace_back((const uint8_t*)data, (const uint8_t*)data + size);
  hipError_t e = hipMemcpyAsync((uint8_t*)b->device_ptr(stream) + offset, staged_updates.back().data(), size, hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e != hipSuccess) throw std::runtime_error{std::string{"update_buffer failed: "} + hipGetErrorString(e)};
}
void CmdContext::draw(uint32_t vertex_count, uint32_t, uint32_t, uint32_t) {
  if (vertex_count != 3) throw std::runtime_error{"Only the full-screen triangle draw is implemented on this path"};
  launch();
}

}  // namespace gpu
