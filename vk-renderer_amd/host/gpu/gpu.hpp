// gpu/gpu.hpp — HIP-backed stand-in for the reference's Vulkan wrapper (src/gpu/*), reduced to
// what the hot-path passes touch: images / buffers, views, samplers, pipelines looked up by
// *program name*, descriptor sets written slot by slot, and a CmdContext that records onto one
// HIP stream.  `dispatch()` / `draw()` hand the bound state to the C-ABI entry point registered
// for the program (include/vkr_postfx.h) — there is no shader, no descriptor pool, no barrier.
//
// Mirrors (names, argument meaning, exceptions): gpu/gpu.hpp:19-51, gpu/cmd_buffers.hpp:162-247,
// gpu/descriptors.hpp:31-117, gpu/resources.hpp:22-42, gpu/samplers.hpp:36-55, gpu/dynbuffer.hpp.
#ifndef VKR_HOST_GPU_HPP_INCLUDED
#define VKR_HOST_GPU_HPP_INCLUDED

#include <array>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../vk_compat.hpp"
#include "../../../include/vkr_postfx.h"

namespace gpu {

// ---- device memory -------------------------------------------------------------------------
// The graph owns its images (reference: VMA allocations, managed_resources.hpp).  Allocation goes
// through a replaceable hook so a launcher can back every image with memory it can also hand to
// RCCL (bench.py backs them with torch tensors); default hipMalloc / hipFree.
using AllocFn = void* (*)(size_t bytes, void* user);
using FreeFn = void (*)(void* ptr, void* user);
void set_device_allocator(AllocFn alloc, FreeFn free_fn, void* user);
void* device_alloc(size_t bytes);
// roctx ranges (libroctx64, loaded on first use; no-ops when it is absent)
void trace_push(const char* name);
void trace_pop();
struct TraceRange { explicit TraceRange(const char* name) { trace_push(name); } ~TraceRange() { trace_pop(); } TraceRange(const TraceRange&) = delete; };
void device_free(void* ptr);

// ---- images --------------------------------------------------------------------------------------
struct ImageInfo {  // gpu/resources.hpp:22-42
  VkFormat format = VK_FORMAT_UNDEFINED;
  VkImageAspectFlags aspect = 0;
  uint32_t width = 0, height = 0, depth = 1, mip_levels = 1, array_layers = 1;
  ImageInfo() {}
  ImageInfo(VkFormat fmt, VkImageAspectFlags aspect_flags, uint32_t w, uint32_t h) : format{fmt}, aspect{aspect_flags}, width{w}, height{h} {}
  ImageInfo(VkFormat fmt, VkImageAspectFlags aspect_flags, uint32_t w, uint32_t h, uint32_t d, uint32_t mips, uint32_t layers)
      : format{fmt}, aspect{aspect_flags}, width{w}, height{h}, depth{d}, mip_levels{mips}, array_layers{layers} {}
  VkExtent3D extent3D() const { return {width, height, depth}; }
  VkExtent2D extent2D() const { return {width, height}; }
};

struct ImageViewRange {
  VkImageViewType type = VK_IMAGE_VIEW_TYPE_2D;
  VkImageAspectFlags aspect = 0;
  uint32_t base_mip = 0, mips_count = 1, base_layer = 0, layers_count = 1;
};

uint32_t to_vkr_format(VkFormat fmt);  // throws for formats the hot path never stores

// Where an image sits inside the whole frame (multi-GPU tiling; identity on one GPU).
struct FrameWindow { uint32_t full_width = 0, full_height = 0; int32_t origin_x = 0, origin_y = 0; };

struct Image {
  Image(const ImageInfo& info, const FrameWindow& window);
  ~Image();
  Image(const Image&) = delete;
  Image& operator=(const Image&) = delete;

  const ImageInfo& get_info() const { return info; }
  VkExtent3D get_extent() const { return info.extent3D(); }
  uint32_t get_mip_levels() const { return info.mip_levels; }
  uint32_t get_array_layers() const { return info.array_layers ? info.array_layers : 1; }
  void* device_ptr() const { return base; }
  size_t size_bytes() const { return bytes; }
  // C-ABI view of mips [base_mip, base_mip + count)
  vkr_img describe(uint32_t base_mip, uint32_t count) const;
  // one layer of a (single-mip) array image
  vkr_img describe_layer(uint32_t layer) const;
  // Multi-GPU strips (host/frame.cpp): of a window image that a pass writes, only rows [row0, row0 + rows) of mip 0 are ever
  // read — by the passes downstream in the same frame, or as the rank's own share of a history whose halo rows arrive from the
  // neighbours.  A program that binds the image as its storage OUTPUT gets the view describe_store() returns: the same memory,
  // window origin and height narrowed to those rows, so its launch covers nothing else.  rows = 0: the whole window (default).
  void set_store_rows(uint32_t row0, uint32_t rows);
  vkr_img describe_store(uint32_t base_mip, uint32_t count) const;
  // tightly packed rows of one mip -> device (asset upload; synchronous)
  void upload_mip(uint32_t mip, const void* rows);
  // asset knowledge the raster stage can use: no texel of any mip level has alpha 0 (set by the scene loader, which
  // sees the host bytes), so opaque_taa.frag:32-34 can never discard a fragment textured with this image
  bool alpha_never_zero = false;

 private:
  ImageInfo info;
  FrameWindow window;
  void* base = nullptr;
  size_t bytes = 0;
  std::array<uint32_t, VKR_MAX_MIPS> pitch{};
  std::array<uint64_t, VKR_MAX_MIPS> offset{};
  uint32_t store_row0 = 0, store_rows = 0;
};
using ImagePtr = std::shared_ptr<Image>;

// what a VkImageView handle points at
struct ImageViewObject { const Image* image; ImageViewRange range; };

struct Buffer {
  Buffer(VmaMemoryUsage memory, uint64_t size, VkBufferUsageFlags usage);
  ~Buffer();
  uint64_t get_size() const { return size; }
  // CPU_TO_GPU buffers: host shadow, uploaded to the device copy when next bound
  void* get_mapped_ptr() { dirty = true; return shadow.data(); }
  const void* host_data() const { return shadow.empty() ? nullptr : shadow.data(); }
  void* device_ptr(void* stream);
  // the handle cmd.dispatch_indirect / cmd.update_buffer take (gpu/resources.hpp api_buffer())
  VkBuffer api_buffer() { return (VkBuffer)this; }

 private:
  uint64_t size;
  void* dev = nullptr;
  std::vector<uint8_t> shadow;
  bool dirty = false;
};
using BufferPtr = std::shared_ptr<Buffer>;
BufferPtr create_buffer(VmaMemoryUsage memory, uint64_t size, VkBufferUsageFlags usage);

// ---- samplers (gpu/samplers.hpp:36-55) -----------------------------------------------------------
// Every hot-path pass uses DEFAULT_SAMPLER: bilinear, clamp-to-edge, LOD 0..10; the kernels
// implement exactly that, so the handle only records that the default was asked for.
constexpr VkSamplerCreateInfo DEFAULT_SAMPLER{VK_FILTER_LINEAR, VK_FILTER_LINEAR, VK_SAMPLER_MIPMAP_MODE_LINEAR,
                                               VK_SAMPLER_ADDRESS_MODE_CLAMP_TO_EDGE, VK_SAMPLER_ADDRESS_MODE_CLAMP_TO_EDGE,
                                               VK_SAMPLER_ADDRESS_MODE_CLAMP_TO_EDGE, 0.f, 10.f};
VkSampler create_sampler(const VkSamplerCreateInfo& info);
const VkSamplerCreateInfo& sampler_info(VkSampler s);

// ---- uniform ring (gpu/dynbuffer.hpp, cmd_buffers.hpp:129) -----------------------------------------
constexpr uint64_t UBO_POOL_SIZE = 16 * (1 << 10);
template <typename T> struct UboBlock { T* ptr; uint32_t offset; };
struct UniformBufferPool {
  UniformBufferPool() : storage(UBO_POOL_SIZE) {}
  void reset() { top = 0; }
  template <typename T> UboBlock<T> allocate_ubo() {
    uint64_t at = (top + 255) & ~uint64_t(255);
    if (at + sizeof(T) > storage.size()) throw std::runtime_error{"Not enough space in uniform buffer"};
    top = at + sizeof(T);
    return UboBlock<T>{reinterpret_cast<T*>(storage.data() + at), (uint32_t)at};
  }
  // untyped block (pass_recorder.hpp)
  UboBlock<uint8_t> allocate_bytes(uint64_t bytes) {
    uint64_t at = (top + 255) & ~uint64_t(255);
    if (at + bytes > storage.size()) throw std::runtime_error{"Not enough space in uniform buffer"};
    top = at + bytes;
    return UboBlock<uint8_t>{storage.data() + at, (uint32_t)at};
  }
  const uint8_t* data() const { return storage.data(); }
 private:
  std::vector<uint8_t> storage;
  uint64_t top = 0;
};

// ---- descriptor sets ---------------------------------------------------------------------------------
struct SetSlot {
  enum Kind { Empty, Texture, StorageTexture, Ubo, Ssbo } kind = Empty;
  ImageViewObject view{nullptr, {}};
  VkSampler sampler = nullptr;
  const void* host_data = nullptr;  // UBO living in the per-frame ring
  uint64_t host_size = 0;
  BufferPtr buffer;                 // UBO / SSBO living in a device buffer
};
struct DescriptorSetObject {
  std::array<SetSlot, 16> slots;
  std::vector<ImageViewObject> image_array;  // ArrayOfImagesBinding (bindless material textures)
};
// gpu/descriptors.hpp: a long-lived set outside the per-frame pool (scene_renderer.cpp:98)
VkDescriptorSet allocate_descriptor_set(VkDescriptorSetLayout layout, const std::initializer_list<uint32_t>& variable_counts);
struct ArrayOfImagesBinding {
  uint32_t binding;
  const std::vector<std::pair<VkImageView, VkSampler>>& images;
};
void write_binding(VkDescriptorSet set, const ArrayOfImagesBinding& b);

struct BaseBinding { uint32_t binding; };
struct TextureBinding : BaseBinding {  // gpu/descriptors.hpp:89-101
  TextureBinding(uint32_t b, VkImageView v, VkSampler s) : BaseBinding{b}, view{v}, sampler{s} {}
  VkImageView view; VkSampler sampler;
};
struct StorageTextureBinding : BaseBinding {  // gpu/descriptors.hpp:103-115
  StorageTextureBinding(uint32_t b, VkImageView v) : BaseBinding{b}, view{v} {}
  VkImageView view;
};
struct UBOBinding : BaseBinding {  // gpu/descriptors.hpp:31-63
  template <typename T> UBOBinding(uint32_t b, const UniformBufferPool&, const UboBlock<T>& blk) : BaseBinding{b}, host{blk.ptr}, size{sizeof(T)} {}
  UBOBinding(uint32_t b, const BufferPtr& buf) : BaseBinding{b}, buffer{buf}, size{buf->get_size()} {}
  UBOBinding(uint32_t b, const void* ring_memory, uint64_t bytes) : BaseBinding{b}, host{ring_memory}, size{bytes} {}
  const void* host = nullptr; BufferPtr buffer; uint64_t size;
};
struct SSBOBinding : BaseBinding {
  SSBOBinding(uint32_t b, const BufferPtr& buf) : BaseBinding{b}, buffer{buf} {}
  BufferPtr buffer;
};
// gpu/descriptors.hpp:168-183: named by GTAO::add_main_rt_pass only.  No program on this path can read an
// acceleration structure, so writing one into a set is an error, not a silent no-op.
struct AccelerationStructBinding : BaseBinding {
  AccelerationStructBinding(uint32_t b, VkAccelerationStructureKHR t) : BaseBinding{b}, tlas{t} {}
  VkAccelerationStructureKHR tlas;
};
void write_binding(VkDescriptorSet set, const AccelerationStructBinding& b);
void write_binding(VkDescriptorSet set, const TextureBinding& b);
void write_binding(VkDescriptorSet set, const StorageTextureBinding& b);
void write_binding(VkDescriptorSet set, const UBOBinding& b);
void write_binding(VkDescriptorSet set, const SSBOBinding& b);
template <typename... Bindings> void write_set(VkDescriptorSet set, const Bindings&... bindings) { (write_binding(set, bindings), ...); }

// ---- programs & pipelines ------------------------------------------------------------------------------
struct CmdContext;
// what a program sees when it is dispatched / drawn
struct LaunchState {
  const DescriptorSetObject* set = nullptr;
  const uint8_t* push = nullptr;
  uint32_t push_size = 0;
  uint32_t groups[3] = {0, 0, 0};
  uint32_t fb_width = 0, fb_height = 0;
  std::vector<ImageViewObject> attachments;  // graphics programs: colour..., depth last
  Buffer* indirect = nullptr;                // dispatch_indirect: VkDispatchIndirectCommand on the device
  // indexed geometry recorded between set_framebuffer and end_renderpass (raster programs)
  struct IndexedDraw { std::vector<uint8_t> push; uint32_t index_count, first_index; int32_t vertex_offset; };
  std::vector<IndexedDraw> indexed_draws;
  const DescriptorSetObject* set1 = nullptr;  // descriptor set 1 (bindless textures)
  Buffer* vertex_buffer = nullptr;
  Buffer* index_buffer = nullptr;
  bool cleared_color = false, cleared_depth = false;
  void* scratch = nullptr;                    // device scratch owned by the command context
  uint64_t scratch_bytes = 0;
  void* stream = nullptr;
  // a second device buffer of the command context, for programs that need room between their own launches (sssr_trace: the
  // frame-wide queue of parked rays); grows on demand, lives as long as the context
  void* workspace = nullptr;
  uint64_t workspace_bytes = 0;
  void* require_workspace(uint64_t bytes);
};
using ProgramFn = std::function<int(LaunchState&)>;
// Registers `name` -> C-ABI thunk.  The hot-path programs of src/shaders/config.json are
// registered by register_hot_path_programs(); anything else: "Program not found".
void create_program(const std::string& name, ProgramFn fn);
void register_hot_path_programs();
bool has_program(const std::string& name);

struct BasePipeline {
  void set_program(const std::string& name);  // throws std::runtime_error{"Program not found"} (gpu/shader_program.cpp:197)
  bool has_program() const { return program.has_value(); }
  const std::string& program_name() const { return *program; }
  VkDescriptorSetLayout get_layout(uint32_t index) const { return (VkDescriptorSetLayout)(uintptr_t)(index + 1); }
 protected:
  std::optional<std::string> program;
};
struct ComputePipeline : BasePipeline {};
struct RenderSubpassDesc {  // gpu/pipelines.hpp: {use_depth, {formats}}; the reference writes set_rendersubpass({false, {format}})
  RenderSubpassDesc() {}
  RenderSubpassDesc(bool depth, std::initializer_list<VkFormat> f) : use_depth{depth}, formats{f} {}
  RenderSubpassDesc(bool depth, std::vector<VkFormat> f) : use_depth{depth}, formats{std::move(f)} {}
  bool use_depth = false;
  std::vector<VkFormat> formats{};
};
struct VertexInput {};
struct Registers {
  struct { VkBool32 depthTestEnable = VK_FALSE; VkCompareOp depthCompareOp = VK_COMPARE_OP_LESS; VkBool32 depthWriteEnable = VK_FALSE; } depth_stencil;
};
struct GraphicsPipeline : BasePipeline {
  void set_vertex_input(const VertexInput&) {}
  void set_registers(const Registers& r) { regs = r; }
  void set_rendersubpass(const RenderSubpassDesc& d) { subpass = d; }
  const RenderSubpassDesc& get_renderpass_desc() const { return subpass; }
  VkRenderPass get_renderpass() const { return nullptr; }  // no render-pass objects on this path (only the UI consumed it)
 private:
  Registers regs;
  RenderSubpassDesc subpass;
};
ComputePipeline create_compute_pipeline();
ComputePipeline create_compute_pipeline(const char* name);
GraphicsPipeline create_graphics_pipeline();

// ---- command context (gpu/cmd_buffers.hpp:162-247) --------------------------------------------------------
struct CmdContext {
  explicit CmdContext(void* hip_stream = nullptr) : stream{hip_stream} {}
  void begin();
  void set_stream(void* s) { stream = s; }
  void* get_stream() const { return stream; }

  VkDescriptorSet allocate_set();
  void bind_pipeline(const ComputePipeline& p);
  void bind_pipeline(const GraphicsPipeline& p);
  void bind_descriptors_compute(uint32_t first_set, const std::initializer_list<VkDescriptorSet>& s, const std::initializer_list<uint32_t> = {}) { bind_sets(first_set, s); }
  void bind_descriptors_graphics(uint32_t first_set, const std::initializer_list<VkDescriptorSet>& s, const std::initializer_list<uint32_t> = {}) { bind_sets(first_set, s); }
  void push_constants_compute(uint32_t offset, uint32_t size, const void* constants);
  void push_constants_graphics(VkShaderStageFlags, uint32_t offset, uint32_t size, const void* constants) { push_constants_compute(offset, size, constants); }
  void set_framebuffer(uint32_t width, uint32_t height, const std::initializer_list<ImageViewObject>& attachments);
  void set_framebuffer(uint32_t width, uint32_t height, const std::vector<ImageViewObject>& attachments);
  void bind_viewport(float, float, float, float, float, float) {}
  void bind_scissors(int32_t, int32_t, uint32_t, uint32_t) {}
  void end_renderpass();
  void clear_color_attachments(float r, float g, float b, float a);
  void clear_depth_attachment(float depth);
  void bind_vertex_buffers(uint32_t first, const std::initializer_list<VkBuffer>& buffers, const std::initializer_list<uint64_t>& offsets);
  void bind_index_buffer(VkBuffer buffer, uint64_t offset, VkIndexType type);
  // recorded, executed as one pass by end_renderpass()
  void draw_indexed(uint32_t index_count, uint32_t instance_count, uint32_t first_index, int32_t vertex_offset, uint32_t first_instance);
  // grows the context's device scratch (raster visibility buffer)
  void* require_scratch(uint64_t bytes);
  ~CmdContext();
  void dispatch(uint32_t groups_x, uint32_t groups_y, uint32_t groups_z);
  void draw(uint32_t vertex_count, uint32_t instance_count, uint32_t first_vertex, uint32_t first_instance);
  // group counts come from a device buffer; the program bounds its launch itself (no host read-back)
  void dispatch_indirect(VkBuffer arguments);
  // vkCmdUpdateBuffer of a small POD value, ordered on the stream
  void update_buffer_bytes(VkBuffer dst, uint64_t offset, const void* data, uint64_t size);
  template <typename T> void update_buffer(VkBuffer dst, uint64_t offset, const T& value) { update_buffer_bytes(dst, offset, &value, sizeof(T)); }
  // cmd_buffers.hpp:212-213 (vkCmdBeginDebugUtilsLabelEXT): here a roctx range, so that a rocprofv3 --marker-trace timeline
  // names every rendergraph task (rendergraph.cpp:289-304 labels each task) and every phase of the tiled frame
  void push_label(const char* name) { trace_push(name); }
  void pop_label() { trace_pop(); }

  UniformBufferPool& get_ubo_pool() { return ubo_pool; }
  template <typename T> UboBlock<T> allocate_ubo() { return ubo_pool.allocate_ubo<T>(); }

 private:
  void bind_sets(uint32_t first_set, const std::initializer_list<VkDescriptorSet>& s);
  void launch();
  void* stream;
  UniformBufferPool ubo_pool;
  std::vector<std::unique_ptr<DescriptorSetObject>> sets;
  std::optional<std::string> bound_program;
  std::vector<uint8_t> push_data;
  std::vector<std::vector<uint8_t>> staged_updates;  // update_buffer payloads, released by begin()
  LaunchState state;
};

// VKCHECK analogue (gpu/common.cpp:6-12): a non-zero C-ABI status becomes std::runtime_error
void check_status(int rc, const char* what);

}  // namespace gpu
#endif
