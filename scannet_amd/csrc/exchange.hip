// exchange.hip -- the one exchange step of a scan partitioned over several GPUs (BASELINE configs[4]), device to device, for host programs
// that have no process group: bin/depthsensing --ranks N starts N copies of itself and each copy calls this.
//
// What travels: before meshing, rank r hands the lowest block layer of each of its stripes (sf_fuser_export_boundary: coords n x 3 int32, voxels
// n x 4096 bytes, written into device memory) to rank r - 1 and receives rank r + 1's; sf_fuser_import_ghosts keeps the blocks that sit right above
// one of its own layers.  Two transports, both from device memory to device memory:
//   rccl  ranks on distinct GPUs: ncclSend / ncclRecv on a communicator of the N ranks (the north star's "RCCL ... of boundary blocks over xGMI"; every
//         pair of GPUs of a node has its own link, so the ring shift is one hop per rank and nothing is all-gathered that only one neighbour needs).
//         librccl is loaded with dlopen when the first exchange is created: programs that never partition a scan do not pay for it.
//   ipc   ranks that share a device (RCCL refuses two ranks on one GPU; `--share-gpu` is how a one-GPU box tests the control flow) or a node without
//         RCCL: the owner publishes a hipIpc handle of its packed boundary, the neighbour maps it and imports straight from the mapping.
// The rendezvous (who sits on which device, the ncclUniqueId, the 64-byte ipc handles) goes through small files in the run's exchange directory; the
// blocks themselves never touch host memory.  The tool keeps its /dev/shm file exchange as the fallback when neither transport comes up.
// scannet_amd/partition.py is the same exchange for callers that do have a process group (torch.distributed over RCCL).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "scanfuse.h"

namespace {

// ---- the few RCCL entry points this file uses, resolved at run time (rccl/rccl.h: ncclUniqueId is 128 opaque bytes, ncclUint8 == 1, ncclSuccess == 0)
struct NcclId { char internal[128]; };
typedef void* NcclComm;
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok() const { return so != nullptr; }
};
constexpr int NCCL_UINT8 = 1;

Rccl load_rccl() {
  Rccl r;
  void* so = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (so) break;
  }
  if (!so) return r;
  r.GetUniqueId = (int (*)(NcclId*))dlsym(so, "ncclGetUniqueId");
  r.CommInitRank = (int (*)(NcclComm*, int, NcclId, int))dlsym(so, "ncclCommInitRank");
  r.CommDestroy = (int (*)(NcclComm))dlsym(so, "ncclCommDestroy");
  r.Send = (int (*)(const void*, size_t, int, int, NcclComm, hipStream_t))dlsym(so, "ncclSend");
  r.Recv = (int (*)(void*, size_t, int, int, NcclComm, hipStream_t))dlsym(so, "ncclRecv");
  r.GroupStart = (int (*)())dlsym(so, "ncclGroupStart");
  r.GroupEnd = (int (*)())dlsym(so, "ncclGroupEnd");
  r.GetErrorString = (const char* (*)(int))dlsym(so, "ncclGetErrorString");
  if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Send && r.Recv && r.GroupStart && r.GroupEnd) r.so = so;
  else dlclose(so);
  return r;
}

// ---- rendezvous files: written under a temporary name and renamed, so a file that exists is complete
bool put_file(const std::string& path, const void* data, size_t n) {
  const std::string tmp = path + ".tmp";
  FILE* fp = std::fopen(tmp.c_str(), "wb");
  if (!fp) return false;
  bool ok = n == 0 || std::fwrite(data, 1, n, fp) == n;
  ok = (std::fclose(fp) == 0) && ok;
  if (ok) ok = std::rename(tmp.c_str(), path.c_str()) == 0;
  if (!ok) std::remove(tmp.c_str());
  return ok;
}
bool there(const std::string& path) {
  struct stat st;
  return ::stat(path.c_str(), &st) == 0;
}

constexpr size_t PAD = 256;
inline size_t padded(size_t n) { return (n + PAD - 1) & ~(PAD - 1); }

struct IpcNote {   // what the owner of a boundary publishes for its neighbour
  hipIpcMemHandle_t handle;
  uint64_t blocks, bytes;
  int32_t device;
  int32_t pad;
};

}  // namespace

struct sf_exchange {
  int rank = 0, ranks = 1, device = 0;
  int transport = 0;   // SF_EXCHANGE_RCCL / SF_EXCHANGE_IPC
  std::string dir;
  double timeout_s = 600.0;
  Rccl rccl;
  NcclComm comm = nullptr;
  hipStream_t stream = nullptr;
  int round = 0;       // exchanges made so far (file names of the ipc notes carry it)
  char what[96] = "";

  // wait for a rendezvous file; gives up when <dir>/abort appears (another rank failed: the parent says so) or after timeout_s
  int wait_for(const std::string& path) const {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      if (there(path)) return SF_OK;
      if (there(dir + "/abort")) return sf::fail(SF_ERR_IO, "exchange aborted (another rank failed) while waiting for %s", path.c_str());
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
        return sf::fail(SF_ERR_IO, "exchange: gave up waiting for %s after %.0f s", path.c_str(), timeout_s);
      std::this_thread::sleep_for(std::chrono::microseconds(300));
    }
  }
  int read_file(const std::string& path, void* dst, size_t n) const {
    const int rc = wait_for(path);
    if (rc != SF_OK) return rc;
    FILE* fp = std::fopen(path.c_str(), "rb");
    const bool ok = fp && std::fread(dst, 1, n, fp) == n;
    if (fp) std::fclose(fp);
    return ok ? SF_OK : sf::fail(SF_ERR_IO, "exchange: could not read %s", path.c_str());
  }
};

#define SFX_HIP(call)                                                                                              \
  do {                                                                                                             \
    const hipError_t e_ = (call);                                                                                  \
    if (e_ != hipSuccess) return sf::fail(SF_ERR_DEVICE, "exchange: %s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)
#define SFX_NCCL(x, call)                                                                                                                        \
  do {                                                                                                                                           \
    const int r_ = (call);                                                                                                                       \
    if (r_ != 0) return sf::fail(SF_ERR_DEVICE, "exchange: %s failed: %s", #call, (x)->rccl.GetErrorString ? (x)->rccl.GetErrorString(r_) : "?"); \
  } while (0)

SF_API int sf_exchange_create(const char* rendezvous_dir, int rank, int ranks, int device, int transport, sf_exchange** out) {
  if (!rendezvous_dir || !out || ranks < 1 || rank < 0 || rank >= ranks || transport < SF_EXCHANGE_AUTO || transport > SF_EXCHANGE_IPC)
    return sf::fail(SF_ERR_INVALID_ARG, "sf_exchange_create: bad argument");
  try {
    sf_exchange* x = new sf_exchange();
    x->rank = rank; x->ranks = ranks; x->device = device; x->dir = rendezvous_dir;
    auto fail_with = [&](int rc) { sf_exchange_destroy(x); return rc; };
    if (hipSetDevice(device) != hipSuccess) return fail_with(sf::fail(SF_ERR_DEVICE, "exchange: no HIP device %d", device));
    if (hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking) != hipSuccess) return fail_with(sf::fail(SF_ERR_DEVICE, "exchange: no stream"));
    // who sits where and what it can do: every rank publishes its device's PCI bus id, whether librccl loads and whether hipIpc hands out a handle; every
    // rank reads the same notes and comes to the same decision -- two ranks on one device rule RCCL out, and a transport is only taken when ALL ranks have it
    // (a rank that fell back on its own would leave the others waiting in a collective)
    struct Note { char bus[64]; int32_t rccl_ok, ipc_ok; } me;
    std::memset(&me, 0, sizeof me);
    if (hipDeviceGetPCIBusId(me.bus, sizeof me.bus - 1, device) != hipSuccess) std::snprintf(me.bus, sizeof me.bus, "device-%d", device);
    if (transport != SF_EXCHANGE_IPC) {
      x->rccl = load_rccl();
      me.rccl_ok = x->rccl.ok() ? 1 : 0;
    }
    {
      void* probe = nullptr;
      hipIpcMemHandle_t h;
      if (hipMalloc(&probe, PAD) == hipSuccess) {
        me.ipc_ok = hipIpcGetMemHandle(&h, probe) == hipSuccess ? 1 : 0;
        (void)hipFree(probe);
      }
      (void)hipGetLastError();
    }
    if (!put_file(x->dir + "/dev" + std::to_string(rank), &me, sizeof me)) return fail_with(sf::fail(SF_ERR_IO, "exchange: could not write into %s", x->dir.c_str()));
    bool shared = false, all_rccl = true, all_ipc = true;
    std::vector<std::string> buses((size_t)ranks);
    for (int r = 0; r < ranks; r++) {
      Note o;
      const int rc = x->read_file(x->dir + "/dev" + std::to_string(r), &o, sizeof o);
      if (rc != SF_OK) return fail_with(rc);
      o.bus[63] = 0;
      buses[(size_t)r] = o.bus;
      all_rccl = all_rccl && o.rccl_ok;
      all_ipc = all_ipc && o.ipc_ok;
      for (int q = 0; q < r; q++) shared = shared || buses[(size_t)q] == buses[(size_t)r];
    }
    if (transport == SF_EXCHANGE_RCCL && shared && ranks > 1)
      return fail_with(sf::fail(SF_ERR_INVALID_ARG, "exchange: RCCL asked for, but two ranks share a device (RCCL refuses that)"));
    if (transport == SF_EXCHANGE_RCCL && !all_rccl) return fail_with(sf::fail(SF_ERR_UNSUPPORTED, "exchange: RCCL asked for, but librccl.so does not load on every rank"));
    if (transport == SF_EXCHANGE_AUTO) transport = ((shared && ranks > 1) || !all_rccl) ? SF_EXCHANGE_IPC : SF_EXCHANGE_RCCL;
    if (transport == SF_EXCHANGE_IPC && !all_ipc && ranks > 1)
      return fail_with(sf::fail(SF_ERR_UNSUPPORTED, "exchange: no device-to-device transport on every rank (hipIpcGetMemHandle failed: HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)"));
    if (transport == SF_EXCHANGE_RCCL) {
      NcclId id;
      std::memset(&id, 0, sizeof id);
      if (rank == 0) {
        const int r = x->rccl.GetUniqueId(&id);
        if (r != 0) return fail_with(sf::fail(SF_ERR_DEVICE, "exchange: ncclGetUniqueId failed: %s", x->rccl.GetErrorString ? x->rccl.GetErrorString(r) : "?"));
        if (!put_file(x->dir + "/nccl.id", &id, sizeof id)) return fail_with(sf::fail(SF_ERR_IO, "exchange: could not publish the RCCL id"));
      } else {
        const int rc = x->read_file(x->dir + "/nccl.id", &id, sizeof id);
        if (rc != SF_OK) return fail_with(rc);
      }
      const int r = x->rccl.CommInitRank(&x->comm, ranks, id, rank);
      if (r != 0) { x->comm = nullptr; return fail_with(sf::fail(SF_ERR_DEVICE, "exchange: ncclCommInitRank failed: %s", x->rccl.GetErrorString ? x->rccl.GetErrorString(r) : "?")); }
      std::snprintf(x->what, sizeof x->what, "rccl (ncclSend / ncclRecv, %d ranks, device to device)", ranks);
    } else {
      std::snprintf(x->what, sizeof x->what, "hipIpc (the neighbour imports from a mapping of the owner's device buffer)");
    }
    x->transport = transport;
    *out = x;
    return SF_OK;
  } catch (...) { return sf::fail(SF_ERR_IO, "exchange: out of memory"); }
}

SF_API const char* sf_exchange_transport(const sf_exchange* x) { return x ? x->what : ""; }

SF_API void sf_exchange_destroy(sf_exchange* x) {
  if (!x) return;
  if (x->comm && x->rccl.ok()) (void)x->rccl.CommDestroy(x->comm);
  if (x->stream) (void)hipStreamDestroy(x->stream);
  // the RCCL library stays loaded: its own threads may outlive the communicator
  delete x;
}

// One exchange: my boundary to rank - 1, rank + 1's boundary to me.  Counts per call: blocks sent, blocks received, blocks kept as ghosts.
SF_API int sf_exchange_boundary(sf_exchange* x, sf_fuser* f, uint64_t* sent, uint64_t* received, uint64_t* kept) {
  if (!x || !f) return sf::fail(SF_ERR_INVALID_ARG, "NULL argument");
  try {
    SFX_HIP(hipSetDevice(x->device));
    const int below = (x->rank + x->ranks - 1) % x->ranks, above = (x->rank + 1) % x->ranks;
    uint64_t n = 0, m = 0, k = 0, got = 0;
    int rc = sf_fuser_export_boundary(f, nullptr, nullptr, 0, &n, 0);
    if (rc != SF_OK) return rc;
    // one device buffer: [coords n x 12 | pad][voxels n x 4096]
    const size_t coord_b = padded((size_t)n * 12), bytes = coord_b + (size_t)n * 4096;
    uint8_t* mine = nullptr;
    SFX_HIP(hipMalloc((void**)&mine, bytes ? bytes : PAD));
    struct Free { void* p; ~Free() { if (p) (void)hipFree(p); } } free_mine{mine};
    if (n) {
      rc = sf_fuser_export_boundary(f, reinterpret_cast<int32_t*>(mine), mine + coord_b, n, &m, 1);
      if (rc != SF_OK) return rc;
      if (m != n) return sf::fail(SF_ERR_DEVICE, "exchange: the boundary export wrote %llu of %llu blocks", (unsigned long long)m, (unsigned long long)n);
    }
    if (x->transport == SF_EXCHANGE_RCCL) {
      // counts first (8 bytes each way), then the payloads; a group makes the send and the receive of a rank one operation (no ordering between ranks to get wrong)
      uint64_t* d_cnt = nullptr;
      SFX_HIP(hipMalloc((void**)&d_cnt, 2 * PAD));
      Free free_cnt{d_cnt};
      SFX_HIP(hipMemcpyAsync(d_cnt, &n, 8, hipMemcpyHostToDevice, x->stream));
      SFX_NCCL(x, x->rccl.GroupStart());
      SFX_NCCL(x, x->rccl.Send(d_cnt, 8, NCCL_UINT8, below, x->comm, x->stream));
      SFX_NCCL(x, x->rccl.Recv(reinterpret_cast<uint8_t*>(d_cnt) + PAD, 8, NCCL_UINT8, above, x->comm, x->stream));
      SFX_NCCL(x, x->rccl.GroupEnd());
      SFX_HIP(hipMemcpyAsync(&k, reinterpret_cast<uint8_t*>(d_cnt) + PAD, 8, hipMemcpyDeviceToHost, x->stream));
      SFX_HIP(hipStreamSynchronize(x->stream));
      if (k > (1ull << 26)) return sf::fail(SF_ERR_BOUNDS, "exchange: rank %d announces %llu boundary blocks", above, (unsigned long long)k);
      const size_t kcoord_b = padded((size_t)k * 12), kbytes = kcoord_b + (size_t)k * 4096;
      uint8_t* theirs = nullptr;
      SFX_HIP(hipMalloc((void**)&theirs, kbytes ? kbytes : PAD));
      Free free_theirs{theirs};
      if (n || k) {
        SFX_NCCL(x, x->rccl.GroupStart());
        if (n) SFX_NCCL(x, x->rccl.Send(mine, bytes, NCCL_UINT8, below, x->comm, x->stream));
        if (k) SFX_NCCL(x, x->rccl.Recv(theirs, kbytes, NCCL_UINT8, above, x->comm, x->stream));
        SFX_NCCL(x, x->rccl.GroupEnd());
        SFX_HIP(hipStreamSynchronize(x->stream));
      }
      if (k) {
        rc = sf_fuser_import_ghosts(f, reinterpret_cast<const int32_t*>(theirs), theirs + kcoord_b, k, 1, &got);
        if (rc != SF_OK) return rc;
      }
    } else {
      // publish a handle of my buffer for the rank below; map the buffer of the rank above and import straight from the mapping; tell its owner I am done
      const std::string tag = std::to_string(x->round);
      IpcNote note;
      std::memset(&note, 0, sizeof note);
      SFX_HIP(hipDeviceSynchronize());   // the export kernels have written the buffer before anybody maps it
      SFX_HIP(hipIpcGetMemHandle(&note.handle, mine));
      note.blocks = n; note.bytes = bytes; note.device = x->device;
      if (!put_file(x->dir + "/ipc" + tag + "_" + std::to_string(x->rank), &note, sizeof note)) return sf::fail(SF_ERR_IO, "exchange: could not publish the ipc handle");
      IpcNote theirs;
      rc = x->read_file(x->dir + "/ipc" + tag + "_" + std::to_string(above), &theirs, sizeof theirs);
      if (rc != SF_OK) return rc;
      k = theirs.blocks;
      if (k > (1ull << 26) || theirs.bytes != padded((size_t)k * 12) + (size_t)k * 4096) return sf::fail(SF_ERR_FORMAT, "exchange: rank %d's ipc note is inconsistent", above);
      if (k && above == x->rank) {   // one rank: its own buffer needs no mapping
        rc = sf_fuser_import_ghosts(f, reinterpret_cast<const int32_t*>(mine), mine + coord_b, k, 1, &got);
        if (rc != SF_OK) return rc;
      } else if (k) {
        void* map = nullptr;
        SFX_HIP(hipIpcOpenMemHandle(&map, theirs.handle, hipIpcMemLazyEnablePeerAccess));
        rc = sf_fuser_import_ghosts(f, reinterpret_cast<const int32_t*>(map), reinterpret_cast<const uint8_t*>(map) + padded((size_t)k * 12), k, 1, &got);
        const hipError_t se = hipDeviceSynchronize();   // the import kernels have read the mapping before it goes away
        (void)hipIpcCloseMemHandle(map);
        if (rc != SF_OK) return rc;
        if (se != hipSuccess) return sf::fail(SF_ERR_DEVICE, "exchange: import from the ipc mapping failed: %s", hipGetErrorString(se));
      }
      if (!put_file(x->dir + "/done" + tag + "_" + std::to_string(x->rank), "", 0)) return sf::fail(SF_ERR_IO, "exchange: could not write the completion note");
      // my buffer may go once the rank below has read it
      rc = x->wait_for(x->dir + "/done" + tag + "_" + std::to_string(below));
      if (rc != SF_OK) return rc;
    }
    x->round++;
    if (sent) *sent = n;
    if (received) *received = k;
    if (kept) *kept = got;
    return SF_OK;
  } catch (...) { return sf::fail(SF_ERR_IO, "exchange: out of memory"); }
}
