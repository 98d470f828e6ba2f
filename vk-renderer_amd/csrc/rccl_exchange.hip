// rccl_exchange.hip — the wire of the multi-GPU frame (SURVEY.md 8(b) "vkr_halo_exchange", 8(e); no reference
// counterpart: the reference is single-queue, gpu/driver.hpp:84).  Host code only.
//
//   vkr_all_gather     one grouped ncclAllGather launch for several surfaces (the Hi-Z mips + downsampled normals, or
//                      the albedo): recv = [rank][bytes], out of place, so a strip's rows land in the whole-frame image
//   vkr_all_gather_v   the same for shares of different sizes (cost-balanced strips): a grouped ncclBroadcast per share
//   vkr_halo_exchange  one grouped ncclSend / ncclRecv launch refreshing the halo rings of a history surface
//
// Both enqueue on the caller's stream and return; nothing blocks the host.  RCCL is loaded with dlopen the first
// time a communicator is made, so the library has no link-time dependency on it and a single-GPU process never
// touches RCCL.  dlopen resolves by soname: in a process that has already loaded a librccl.so.1 (PyTorch brings its
// own under torch/lib) that copy is the one used; a plain C++ integrator gets the ROCm installation's.
// VKR_RCCL_LIBRARY names a specific file instead.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "vkr_host.hpp"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_once;
const char* g_load_error = nullptr;

void load_rccl() {
  static const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
  const char* forced = getenv("VKR_RCCL_LIBRARY");
  void* h = forced ? dlopen(forced, RTLD_NOW | RTLD_LOCAL) : nullptr;
  for (size_t i = 0; !h && !forced && i < sizeof(names) / sizeof(names[0]); i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!h) { g_load_error = "librccl.so.1 not found (set VKR_RCCL_LIBRARY)"; return; }
  g_rccl.handle = h;
#define VKR_SYM(field, name) \
  *(void**)(&g_rccl.field) = dlsym(h, name); \
  if (!g_rccl.field) { g_load_error = "RCCL symbol missing: " name; return; }
  VKR_SYM(GetUniqueId, "ncclGetUniqueId")
  VKR_SYM(CommInitRank, "ncclCommInitRank")
  VKR_SYM(CommDestroy, "ncclCommDestroy")
  VKR_SYM(AllGather, "ncclAllGather")
  VKR_SYM(Broadcast, "ncclBroadcast")
  VKR_SYM(Send, "ncclSend")
  VKR_SYM(Recv, "ncclRecv")
  VKR_SYM(GroupStart, "ncclGroupStart")
  VKR_SYM(GroupEnd, "ncclGroupEnd")
  VKR_SYM(GetErrorString, "ncclGetErrorString")
#undef VKR_SYM
}

bool rccl_ready(const char* what) {
  std::call_once(g_once, load_rccl);
  if (g_load_error) { vkr::set_error("%s: %s", what, g_load_error); return false; }
  return true;
}

int fail(const char* what, ncclResult_t r) {
  vkr::set_error("%s: RCCL error %d: %s", what, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return 2000 + (int)r;
}

}  // namespace

struct vkr_comm {
  ncclComm_t comm;
  int rank, world;
  bool emulated = false;      // vkr_comm_create_emulated: no bytes move, the stream is held for the time the wire would take
  float link_gbps = 0.0f, launch_us = 0.0f;
};

namespace {
// one wave that holds its stream for `ticks` of the 100 MHz wall clock (the time passes whatever else happens: every wave exits)
__global__ void k_wire_delay(uint64_t ticks) {
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
int emulate_wire(const vkr_comm* comm, uint64_t busiest_link_bytes, void* stream, const char* what) {
  const double us = (double)comm->launch_us + (double)busiest_link_bytes / ((double)comm->link_gbps * 1e3);
  hipLaunchKernelGGL(k_wire_delay, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint64_t)(us * 100.0));
  return vkr::launch_status(what);
}
}  // namespace

static_assert(sizeof(ncclUniqueId) == VKR_COMM_ID_BYTES, "VKR_COMM_ID_BYTES must hold an ncclUniqueId");

extern "C" int vkr_comm_unique_id(uint8_t* id_bytes) {
  if (!id_bytes) { vkr::set_error("comm_unique_id: NULL argument"); return vkr::VKR_ERR_NULL; }
  if (!rccl_ready("comm_unique_id")) return vkr::VKR_ERR_LAYOUT;
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail("comm_unique_id", r);
  std::memcpy(id_bytes, &id, sizeof(id));
  return vkr::VKR_OK;
}

extern "C" int vkr_comm_create(const uint8_t* id_bytes, int rank, int world, vkr_comm** out) {
  if (!id_bytes || !out) { vkr::set_error("comm_create: NULL argument"); return vkr::VKR_ERR_NULL; }
  if (world < 1 || rank < 0 || rank >= world) { vkr::set_error("comm_create: rank %d of %d", rank, world); return vkr::VKR_ERR_EXTENT; }
  if (!rccl_ready("comm_create")) return vkr::VKR_ERR_LAYOUT;
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t r = g_rccl.CommInitRank(&c, world, id, rank);  // collective over the current device of every rank
  if (r != ncclSuccess) return fail("comm_create", r);
  *out = new vkr_comm{c, rank, world};
  return vkr::VKR_OK;
}

extern "C" int vkr_comm_create_emulated(int rank, int world, float link_gbps, float launch_us, vkr_comm** out) {
  if (!out) { vkr::set_error("comm_create_emulated: NULL argument"); return vkr::VKR_ERR_NULL; }
  if (world < 1 || rank < 0 || rank >= world || !(link_gbps > 0.0f) || launch_us < 0.0f) {
    vkr::set_error("comm_create_emulated: rank %d of %d, %g GB/s per link, %g us per launch", rank, world, (double)link_gbps, (double)launch_us);
    return vkr::VKR_ERR_EXTENT;
  }
  vkr_comm* c = new vkr_comm{nullptr, rank, world};
  c->emulated = true; c->link_gbps = link_gbps; c->launch_us = launch_us;
  *out = c;
  return vkr::VKR_OK;
}

extern "C" int vkr_comm_destroy(vkr_comm* comm) {
  if (!comm) return vkr::VKR_OK;
  if (comm->emulated) { delete comm; return vkr::VKR_OK; }
  const ncclResult_t r = g_rccl.CommDestroy(comm->comm);
  delete comm;
  return r == ncclSuccess ? vkr::VKR_OK : fail("comm_destroy", r);
}

extern "C" int vkr_comm_rank(const vkr_comm* comm, int* rank, int* world) {
  if (!comm) { vkr::set_error("comm_rank: NULL communicator"); return vkr::VKR_ERR_NULL; }
  if (rank) *rank = comm->rank;
  if (world) *world = comm->world;
  return vkr::VKR_OK;
}

extern "C" int vkr_comm_available(void) { return rccl_ready("comm_available") ? vkr::VKR_OK : vkr::VKR_ERR_LAYOUT; }

extern "C" int vkr_all_gather(vkr_comm* comm, const vkr_gather_part* parts, uint32_t count, void* stream) {
  if (count == 0) return vkr::VKR_OK;
  if (!comm || !parts) { vkr::set_error("all_gather: NULL argument"); return vkr::VKR_ERR_NULL; }
  for (uint32_t i = 0; i < count; i++)
    if (!parts[i].send || !parts[i].recv || parts[i].bytes == 0) { vkr::set_error("all_gather: part %u is empty", i); return vkr::VKR_ERR_NULL; }
  if (comm->emulated) {  // every peer's share arrives on its own link, the parts one after the other
    uint64_t link = 0;
    for (uint32_t i = 0; i < count; i++) {
      void* dst = (uint8_t*)parts[i].recv + uint64_t(comm->rank) * parts[i].bytes;
      if (parts[i].send != dst && hipMemcpyAsync(dst, parts[i].send, (size_t)parts[i].bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
        vkr::set_error("all_gather (emulated): copy of the own share failed"); return vkr::VKR_ERR_LAYOUT;
      }
      link += parts[i].bytes;
    }
    return emulate_wire(comm, comm->world > 1 ? link : 0, stream, "all_gather (emulated)");
  }
  ncclResult_t r = g_rccl.GroupStart();
  for (uint32_t i = 0; r == ncclSuccess && i < count; i++)
    r = g_rccl.AllGather(parts[i].send, parts[i].recv, (size_t)parts[i].bytes, ncclUint8, comm->comm, (hipStream_t)stream);
  const ncclResult_t e = g_rccl.GroupEnd();
  if (r != ncclSuccess) return fail("all_gather", r);
  return e == ncclSuccess ? vkr::VKR_OK : fail("all_gather", e);
}

extern "C" int vkr_all_gather_v(vkr_comm* comm, const vkr_gather_v_part* parts, uint32_t count, void* stream) {
  if (count == 0) return vkr::VKR_OK;
  if (!comm || !parts) { vkr::set_error("all_gather_v: NULL argument"); return vkr::VKR_ERR_NULL; }
  for (uint32_t i = 0; i < count; i++) {
    const vkr_gather_v_part& p = parts[i];
    if (!p.recv || !p.offsets) { vkr::set_error("all_gather_v: part %u has a NULL buffer", i); return vkr::VKR_ERR_NULL; }
    for (int r = 0; r < comm->world; r++)
      if (p.offsets[r + 1] < p.offsets[r]) { vkr::set_error("all_gather_v: part %u: offsets must not decrease", i); return vkr::VKR_ERR_LAYOUT; }
    if (p.offsets[comm->rank + 1] > p.offsets[comm->rank] && !p.send) { vkr::set_error("all_gather_v: part %u has no send buffer", i); return vkr::VKR_ERR_NULL; }
  }
  // Shares of different sizes.  Default: every rank sends its share of every surface straight to each peer and receives
  // theirs into place, all in one group — on a fully connected xGMI node each share crosses one link once, and RCCL fuses
  // the point-to-point operations of a group into one launch (the same shape as an all-to-all-v).  VKR_GATHER_V_BROADCAST=1
  // selects the textbook form instead, one ncclBroadcast per surface and owner (world x surfaces collectives in the group).
  static const bool by_broadcast = getenv("VKR_GATHER_V_BROADCAST") != nullptr;
  if (comm->emulated) {  // peer r's shares of all parts arrive on the link from r: the busiest link carries the tallest strip
    uint64_t link = 0;
    for (int peer = 0; peer < comm->world; peer++) {
      uint64_t b = 0;
      for (uint32_t i = 0; i < count; i++) b += parts[i].offsets[peer + 1] - parts[i].offsets[peer];
      if (peer == comm->rank) {  // what this rank sends to every peer leaves on as many links, full duplex: its own share counts once
        for (uint32_t i = 0; i < count; i++) {
          const vkr_gather_v_part& p = parts[i];
          const uint64_t n = p.offsets[peer + 1] - p.offsets[peer];
          void* dst = (uint8_t*)p.recv + p.offsets[peer];
          if (n && p.send != dst && hipMemcpyAsync(dst, p.send, (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
            vkr::set_error("all_gather_v (emulated): copy of the own share failed"); return vkr::VKR_ERR_LAYOUT;
          }
        }
      }
      if (comm->world > 1 && b > link) link = b;
    }
    return emulate_wire(comm, link, stream, "all_gather_v (emulated)");
  }
  if (!by_broadcast) {
    for (uint32_t i = 0; i < count; i++) {  // my own share into place (what the root's broadcast would have copied)
      const vkr_gather_v_part& p = parts[i];
      const uint64_t n = p.offsets[comm->rank + 1] - p.offsets[comm->rank];
      void* dst = (uint8_t*)p.recv + p.offsets[comm->rank];
      if (n && p.send != dst) {
        const hipError_t ce = hipMemcpyAsync(dst, p.send, (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (ce != hipSuccess) { vkr::set_error("all_gather_v: copy of the own share failed: %s", hipGetErrorString(ce)); return (int)ce; }
      }
    }
    ncclResult_t r = g_rccl.GroupStart();
    for (uint32_t i = 0; r == ncclSuccess && i < count; i++) {
      const vkr_gather_v_part& p = parts[i];
      const uint64_t mine = p.offsets[comm->rank + 1] - p.offsets[comm->rank];
      for (int peer = 0; r == ncclSuccess && peer < comm->world; peer++) {
        if (peer == comm->rank) continue;
        const uint64_t theirs = p.offsets[peer + 1] - p.offsets[peer];
        if (mine) r = g_rccl.Send(p.send, (size_t)mine, ncclUint8, peer, comm->comm, (hipStream_t)stream);
        if (r == ncclSuccess && theirs) r = g_rccl.Recv((uint8_t*)p.recv + p.offsets[peer], (size_t)theirs, ncclUint8, peer, comm->comm, (hipStream_t)stream);
      }
    }
    const ncclResult_t e = g_rccl.GroupEnd();
    if (r != ncclSuccess) return fail("all_gather_v", r);
    return e == ncclSuccess ? vkr::VKR_OK : fail("all_gather_v", e);
  }
  ncclResult_t r = g_rccl.GroupStart();
  for (uint32_t i = 0; r == ncclSuccess && i < count; i++) {
    const vkr_gather_v_part& p = parts[i];
    for (int root = 0; r == ncclSuccess && root < comm->world; root++) {
      const uint64_t n = p.offsets[root + 1] - p.offsets[root];
      if (n == 0) continue;
      void* dst = (uint8_t*)p.recv + p.offsets[root];
      r = g_rccl.Broadcast(root == comm->rank ? p.send : dst, dst, (size_t)n, ncclUint8, root, comm->comm, (hipStream_t)stream);
    }
  }
  const ncclResult_t e = g_rccl.GroupEnd();
  if (r != ncclSuccess) return fail("all_gather_v", r);
  return e == ncclSuccess ? vkr::VKR_OK : fail("all_gather_v", e);
}

extern "C" int vkr_halo_exchange(vkr_comm* comm, const vkr_halo_peer* peers, uint32_t count, void* stream) {
  if (count == 0) return vkr::VKR_OK;
  if (!comm || !peers) { vkr::set_error("halo_exchange: NULL argument"); return vkr::VKR_ERR_NULL; }
  for (uint32_t i = 0; i < count; i++) {
    const vkr_halo_peer& p = peers[i];
    if (p.peer < 0 || p.peer >= comm->world || p.peer == comm->rank) { vkr::set_error("halo_exchange: peer %d of rank %d / %d", p.peer, comm->rank, comm->world); return vkr::VKR_ERR_EXTENT; }
    if ((p.send_bytes && !p.send) || (p.recv_bytes && !p.recv)) { vkr::set_error("halo_exchange: peer %d has a NULL buffer", p.peer); return vkr::VKR_ERR_NULL; }
  }
  if (comm->emulated) {  // one link per peer, full duplex: the busiest link carries max(send, recv) of that peer's messages
    uint64_t link = 0;
    for (int peer = 0; peer < comm->world; peer++) {
      uint64_t out_b = 0, in_b = 0;
      for (uint32_t i = 0; i < count; i++)
        if (peers[i].peer == peer) { out_b += peers[i].send_bytes; in_b += peers[i].recv_bytes; }
      const uint64_t b = out_b > in_b ? out_b : in_b;
      if (b > link) link = b;
    }
    return emulate_wire(comm, link, stream, "halo_exchange (emulated)");
  }
  // every send and receive of the surface in one group: RCCL fuses them into one launch and cannot deadlock on their order
  ncclResult_t r = g_rccl.GroupStart();
  for (uint32_t i = 0; r == ncclSuccess && i < count; i++) {
    const vkr_halo_peer& p = peers[i];
    if (p.send_bytes) r = g_rccl.Send(p.send, (size_t)p.send_bytes, ncclUint8, p.peer, comm->comm, (hipStream_t)stream);
    if (r == ncclSuccess && p.recv_bytes) r = g_rccl.Recv(p.recv, (size_t)p.recv_bytes, ncclUint8, p.peer, comm->comm, (hipStream_t)stream);
  }
  const ncclResult_t e = g_rccl.GroupEnd();
  if (r != ncclSuccess) return fail("halo_exchange", r);
  return e == ncclSuccess ? vkr::VKR_OK : fail("halo_exchange", e);
}

// ---- start-up self check -------------------------------------------------------------------------------------------
// Shares of SELF_UNIT * (r + 1) bytes for the all-gather-v (so every offset differs), SELF_UNIT for the uniform gather and
// the halo trade; byte k of what rank r contributes to test t is pattern(t, r, k).
namespace {
constexpr uint64_t SELF_UNIT = 4096;
inline uint8_t self_pattern(int test, int rank, uint64_t k) { return (uint8_t)(0x5Au ^ (uint32_t)(test * 61 + rank * 37) ^ (uint32_t)(k * 7u + (k >> 8))); }
uint64_t self_v_offset(int r) { return SELF_UNIT * (uint64_t)r * (uint64_t)(r + 1) / 2; }  // sum of (i + 1) for i < r
}  // namespace

extern "C" uint64_t vkr_comm_selfcheck_bytes(int world) {
  if (world < 1) world = 1;
  // [gather send | gather recv world] [gather_v frame (in place)] [halo: 2 send, 2 recv]
  return SELF_UNIT * (1 + (uint64_t)world) + self_v_offset(world) + 4 * SELF_UNIT;
}

extern "C" int vkr_comm_selfcheck(vkr_comm* comm, void* scratch, void* stream) {
  using vkr::VKR_OK;
  if (!comm || !scratch) { vkr::set_error("comm_selfcheck: NULL argument"); return vkr::VKR_ERR_NULL; }
  if (comm->emulated) { vkr::set_error("comm_selfcheck: an emulated communicator moves no bytes"); return vkr::VKR_ERR_LAYOUT; }
  const int rank = comm->rank, world = comm->world;
  const uint64_t total = vkr_comm_selfcheck_bytes(world);
  std::vector<uint8_t> host(total, 0xEE);
  uint8_t* const dev = (uint8_t*)scratch;
  const uint64_t g_send = 0, g_recv = SELF_UNIT, v_frame = g_recv + SELF_UNIT * world, h_base = v_frame + self_v_offset(world);
  for (uint64_t k = 0; k < SELF_UNIT; k++) host[g_send + k] = self_pattern(0, rank, k);
  for (uint64_t k = 0; k < SELF_UNIT * (rank + 1); k++) host[v_frame + self_v_offset(rank) + k] = self_pattern(1, rank, k);  // my share, in place
  for (int nb = 0; nb < 2; nb++)
    for (uint64_t k = 0; k < SELF_UNIT; k++) host[h_base + nb * SELF_UNIT + k] = self_pattern(2 + nb, rank, k);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemcpyAsync(dev, host.data(), total, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) { vkr::set_error("comm_selfcheck: %s", hipGetErrorString(e)); return (int)e; }
  const vkr_gather_part gp {dev + g_send, dev + g_recv, SELF_UNIT};
  VKR_TRY(vkr_all_gather(comm, &gp, 1, stream));
  std::vector<uint64_t> offs(world + 1);
  for (int r = 0; r <= world; r++) offs[r] = self_v_offset(r);
  const vkr_gather_v_part vp {dev + v_frame + offs[rank], dev + v_frame, offs.data()};
  VKR_TRY(vkr_all_gather_v(comm, &vp, 1, stream));
  vkr_halo_peer peers[2];
  uint32_t np = 0;
  // slot 0 travels up (to rank - 1), slot 1 down (to rank + 1); what arrives from above was that rank's slot 1
  if (rank > 0) peers[np++] = vkr_halo_peer {rank - 1, 0u, dev + h_base, SELF_UNIT, dev + h_base + 2 * SELF_UNIT, SELF_UNIT};
  if (rank + 1 < world) peers[np++] = vkr_halo_peer {rank + 1, 0u, dev + h_base + SELF_UNIT, SELF_UNIT, dev + h_base + 3 * SELF_UNIT, SELF_UNIT};
  VKR_TRY(vkr_halo_exchange(comm, peers, np, stream));
  e = hipMemcpyAsync(host.data(), dev, total, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { vkr::set_error("comm_selfcheck: %s", hipGetErrorString(e)); return (int)e; }
  for (int r = 0; r < world; r++) {
    for (uint64_t k = 0; k < SELF_UNIT; k++)
      if (host[g_recv + SELF_UNIT * r + k] != self_pattern(0, r, k)) { vkr::set_error("comm_selfcheck: all_gather: share of rank %d wrong at byte %llu on rank %d", r, (unsigned long long)k, rank); return vkr::VKR_ERR_EXTENT; }
    for (uint64_t k = 0; k < SELF_UNIT * (r + 1); k++)
      if (host[v_frame + offs[r] + k] != self_pattern(1, r, k)) { vkr::set_error("comm_selfcheck: all_gather_v: share of rank %d wrong at byte %llu on rank %d", r, (unsigned long long)k, rank); return vkr::VKR_ERR_EXTENT; }
  }
  for (uint64_t k = 0; k < SELF_UNIT; k++) {
    if (rank > 0 && host[h_base + 2 * SELF_UNIT + k] != self_pattern(3, rank - 1, k)) { vkr::set_error("comm_selfcheck: halo from rank %d wrong at byte %llu", rank - 1, (unsigned long long)k); return vkr::VKR_ERR_EXTENT; }
    if (rank + 1 < world && host[h_base + 3 * SELF_UNIT + k] != self_pattern(2, rank + 1, k)) { vkr::set_error("comm_selfcheck: halo from rank %d wrong at byte %llu", rank + 1, (unsigned long long)k); return vkr::VKR_ERR_EXTENT; }
  }
  return vkr::VKR_OK;
}
