// tool_depthsensing.cpp -- drop-in for the `improve` stage executable of the reference pipeline:
//     DepthSensing.exe zParametersScanNet.txt zParametersTrackingDefault.txt <abs path>.sens
// (Server/scan_processor.py:34-35,137-138; cwd = tool directory, Server/util.py:34-38).  Product: <dir>/<id>_vh.ply
// next to the .sens file (scan_processor.py:141; Server/config/scan_stages.json:33-42 checks existence only).
// Protocol kept: progress on stdout (captured into process.log, util.py:38-41), NOTHING on stderr on success
// (util.py:42-44 logs any stderr as an error), non-zero exit + stderr message on failure.
// Thin C++ host over libscanfuse.so's C ABI; device selected with SF_DEVICE (one process per GPU).
//
// --ranks N (not an argument of the tool this replaces): ONE scan over N GPUs, BASELINE configs[4] -- this process starts N copies of itself, one per
// GPU (SF_DEVICE + rank).  Every rank reads the whole file and fuses the stripes of the block space it owns (sf_fuser_set_stripes: 16 block layers
// along x dealt round-robin); before meshing, rank r hands the lowest layer of each of its stripes to rank r - 1 -- the only exchange of the path, device
// to device: RCCL send / recv between the GPUs, a hipIpc mapping between ranks that share one (csrc/exchange.hip; files in the exchange directory only
// when neither comes up; scannet_amd/partition.py is the same exchange for callers that have a torch.distributed process group); every rank meshes its
// own blocks and the parent merges the parts by key (sf_mesh_merge_parts) into the mesh one GPU would have written.
#include <signal.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "scanfuse.h"

extern char** environ;

namespace {

const int STRIPE_BLOCKS = 16;   // scannet_amd/partition.py STRIPE_BLOCKS: 0.5 m at 4 mm voxels, a fraction of the view frustum
char g_prefix[32] = "";         // "[rank r/N] " in front of every line a rank of a partitioned run prints

int die(const char* what) {
  std::fprintf(stderr, "%s%s: %s\n", g_prefix, what, sf_last_error());
  return 1;
}
int die_msg(const char* fmt, ...) {
  std::fprintf(stderr, "%s", g_prefix);
  va_list ap;
  va_start(ap, fmt);
  std::vfprintf(stderr, fmt, ap);
  va_end(ap);
  std::fprintf(stderr, "\n");
  return 1;
}
void say(const char* fmt, ...) {
  std::fputs(g_prefix, stdout);
  va_list ap;
  va_start(ap, fmt);
  std::vprintf(fmt, ap);
  va_end(ap);
}

struct Args {
  int upstream = 0;
  int ranks = 1, rank = -1;      // rank >= 0: this process is one rank of a partitioned run (started by the parent below)
  bool share_gpu = false;        // testing aid: every rank on the same device
  std::string ipc;               // the directory the ranks exchange through
  const char* pos[8];
  int n_pos = 0;
};

// ---- files of the exchange directory: written under a temporary name and renamed, so a file that exists is complete --------------------------------
bool write_file(const std::string& path, const std::vector<std::pair<const void*, size_t>>& pieces) {
  const std::string tmp = path + ".tmp";
  FILE* fp = std::fopen(tmp.c_str(), "wb");
  if (!fp) return false;
  bool ok = true;
  for (const auto& pc : pieces)
    if (pc.second && std::fwrite(pc.first, 1, pc.second, fp) != pc.second) { ok = false; break; }
  ok = (std::fclose(fp) == 0) && ok;
  if (ok) ok = std::rename(tmp.c_str(), path.c_str()) == 0;
  if (!ok) std::remove(tmp.c_str());
  return ok;
}
bool exists(const std::string& path) {
  struct stat st;
  return ::stat(path.c_str(), &st) == 0;
}
bool read_exact(FILE* fp, void* dst, size_t n) { return n == 0 || std::fread(dst, 1, n, fp) == n; }

// A rank waits for its right neighbour's boundary file; it gives up when the parent says so (another rank failed) or is gone.
bool wait_for(const std::string& path, const std::string& abort_flag, pid_t parent) {
  for (;;) {
    if (exists(path)) return true;
    if (exists(abort_flag) || getppid() != parent) return false;
    usleep(500);
  }
}

// A rank's mesh as it travels to the parent: {nv, nf} then the arrays, widest element type first so that every array is aligned in the mapping --
// keys nv x u64, face keys nf x u64, xyz nv x 3 f32, tris nf x 3 u32, rgba nv x 4 u8.  The parent maps the file and hands the pointers to
// sf_mesh_create_keyed (one copy, into the handle).
struct Part {
  uint64_t nv = 0, nf = 0;
  void* map = nullptr;
  size_t bytes = 0;
  const uint64_t *keys = nullptr, *fkeys = nullptr;
  const float* xyz = nullptr;
  const uint32_t* tris = nullptr;
  const uint8_t* rgba = nullptr;
  void unmap() {
    if (map) ::munmap(map, bytes);
    map = nullptr;
  }
};
bool map_part(const std::string& path, Part* p) {
  const int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  struct stat st;
  bool ok = ::fstat(fd, &st) == 0 && st.st_size >= 16;
  if (ok) {
    p->bytes = (size_t)st.st_size;
    p->map = ::mmap(nullptr, p->bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    ok = p->map != MAP_FAILED;
    if (!ok) p->map = nullptr;
  }
  ::close(fd);
  if (!ok) return false;
  const uint8_t* at = (const uint8_t*)p->map;
  std::memcpy(&p->nv, at, 8);
  std::memcpy(&p->nf, at + 8, 8);
  if (p->nv > (1ull << 32) || p->nf > (1ull << 34) || p->bytes != 16 + p->nv * 24 + p->nf * 20) { p->unmap(); return false; }
  at += 16;
  p->keys = (const uint64_t*)at;  at += p->nv * 8;
  p->fkeys = (const uint64_t*)at; at += p->nf * 8;
  p->xyz = (const float*)at;      at += p->nv * 12;
  p->tris = (const uint32_t*)at;  at += p->nf * 12;
  p->rgba = at;
  return true;
}

std::string output_path(const Args& a) {
  if (a.n_pos > 4) return a.pos[4];
  std::string out = a.pos[3];
  const size_t dot = out.find_last_of('.');
  if (dot != std::string::npos) out = out.substr(0, dot);
  return out + "_vh.ply";
}

// ---- one process, one GPU: the whole scan (ranks == 1) or this rank's stripes of it ------------------------------------------------------------------
int fuse_scan(const Args& a) {
  const bool part = a.rank >= 0;
  const char* sens_path = a.pos[3];
  sf_params p;
  sf_params_default(&p);
  if (a.upstream && sf_params_upstream_preset(&p, a.upstream) != SF_OK) return die("preset");
  if (sf_params_load_file(a.pos[1], &p) != SF_OK) return die("parameter file");   // s_scanfuse* keys of the file override the preset
  if (p.frustum_mode | p.colour_round | p.colour_first | p.weight_mode | p.weight_wrap)
    say("Upstream-conformance switches: frustum_mode %d, colour_round %d, colour_first %d, weight_mode %d, weight_wrap %d\n", p.frustum_mode, p.colour_round,
        p.colour_first, p.weight_mode, p.weight_wrap);
  // the second parameter file holds tracking settings only; it must exist (the reference tool reads it) but nothing in it concerns fusion
  if (FILE* fp = std::fopen(a.pos[2], "r")) std::fclose(fp);
  else return die_msg("could not open parameter file %s", a.pos[2]);
  sf_sens* sens = nullptr;
  if (sf_sens_open(sens_path, &sens) != SF_OK) return die("sens");
  sf_sens_info info;
  sf_sens_get_info(sens, &info);
  say("Loaded %s: %llu frames, depth %ux%u, color %ux%u, sensor '%s'\n", sens_path, (unsigned long long)info.num_frames, info.depth_width, info.depth_height,
      info.color_width, info.color_height, info.sensor_name);
  // the frames come at the file's depth resolution with the file's calibration (row-major intrinsic, sensorData.h:305-312); the fuser
  // resamples them to s_integrationWidth x s_integrationHeight when the parameter file asks for another size (zParametersScanNet.txt:20-21)
  p.depth_width = (int32_t)info.depth_width;
  p.depth_height = (int32_t)info.depth_height;
  if (p.integration_width > 0 && p.integration_height > 0 && (p.integration_width != p.depth_width || p.integration_height != p.depth_height))
    say("Depth resampled to s_integrationWidth x s_integrationHeight = %d x %d\n", p.integration_width, p.integration_height);
  p.fx = info.depth_intrinsic[0]; p.fy = info.depth_intrinsic[5]; p.mx = info.depth_intrinsic[2]; p.my = info.depth_intrinsic[6];
  p.depth_shift = info.depth_shift;
  if ((info.color_width != info.depth_width || info.color_height != info.depth_height) && info.color_width > 0 && info.color_height > 0 &&
      (info.color_compression == 0 || info.color_compression == 2) && info.color_intrinsic[0] > 0) {
    // real ScanNet scans: 1296x968 colour over 640x480 depth -- sample the colour under each depth pixel's ray
    p.color_width = (int32_t)info.color_width; p.color_height = (int32_t)info.color_height;
    p.cfx = info.color_intrinsic[0]; p.cfy = info.color_intrinsic[5]; p.cmx = info.color_intrinsic[2]; p.cmy = info.color_intrinsic[6];
  }
  const int device = std::getenv("SF_DEVICE") ? std::atoi(std::getenv("SF_DEVICE")) : 0;
  (void)sf_fuse_run_prepare(sens, &p, device);   // the run's streams and rings, sized for this file, made beside the fuser's own allocations (an optimisation: failures surface in sf_fuser_create)
  sf_fuser* fuser = nullptr;
  if (sf_fuser_create(&p, device, &fuser) != SF_OK) return die("fuser");
  if (part && sf_fuser_set_stripes(fuser, 0, 0, STRIPE_BLOCKS, a.ranks, a.rank) != SF_OK) return die("stripes");
  sf_run_stats rs;
  // decode threads: the library's default (one per usable core) for one process; the ranks of a partitioned run share the host's cores
  const int cores = (int)std::thread::hardware_concurrency();
  const int decode_threads = part ? std::max(4, cores / std::max(1, a.ranks)) : 0;
  if (sf_fuse_run(fuser, sens, 0, 0, decode_threads, &rs) != SF_OK) return die("fuse");
  sf_stats st;
  sf_fuser_stats(fuser, &st);
  say("Integrated %llu frames (%llu skipped: invalid pose) in %.3f s = %.1f frames/s with %u decode threads; %u SDF blocks, heapFreeCount = %u\n",
      (unsigned long long)rs.frames_integrated, (unsigned long long)rs.frames_skipped, rs.seconds_total,
      rs.seconds_total > 0 ? (double)rs.frames_total / rs.seconds_total : 0.0, rs.decode_threads, st.blocks_allocated, st.heap_free);
  if (st.alloc_failures) say("WARNING: %u block allocations failed (s_hashNumSDFBlocks / s_hashNumBuckets too small)\n", st.alloc_failures);
  if (p.gc_enabled) {
    uint32_t freed = 0;
    if (sf_fuser_garbage_collect(fuser, &freed) != SF_OK) return die("garbage collection");
    say("Garbage collection freed %u blocks\n", freed);
  }
  if (part) {
    // the exchange step: my stripes' lowest layers out, my right neighbour's in (rank r's stripes lie right above rank r - 1's; ghosts are read as
    // neighbours by marching cubes and never fused or meshed).  Device to device (sf_exchange_*: RCCL send / recv between GPUs, a hipIpc mapping
    // between ranks that share one); SF_EXCHANGE=file|ipc|rccl forces a transport, and files in the exchange directory are what is left when no
    // device-to-device transport comes up on every rank (the decision is the same on every rank: sf_exchange_create).
    const char* want = std::getenv("SF_EXCHANGE");
    const bool files_only = want && !std::strcmp(want, "file");
    const int transport = want && !std::strcmp(want, "rccl") ? SF_EXCHANGE_RCCL : (want && !std::strcmp(want, "ipc") ? SF_EXCHANGE_IPC : SF_EXCHANGE_AUTO);
    if (want && !files_only && transport == SF_EXCHANGE_AUTO && std::strcmp(want, "auto")) return die_msg("SF_EXCHANGE=%s: expected file, ipc, rccl or auto", want);
    sf_exchange* xch = nullptr;
    int xrc = files_only ? SF_ERR_UNSUPPORTED : sf_exchange_create(a.ipc.c_str(), a.rank, a.ranks, device, transport, &xch);
    if (xrc != SF_OK && xrc != SF_ERR_UNSUPPORTED) return die("exchange set-up");
    if (xrc != SF_OK && !files_only && transport != SF_EXCHANGE_AUTO) return die("exchange set-up");   // a transport that was asked for by name is not replaced silently
    if (xrc == SF_OK) {
      uint64_t n = 0, k = 0, got = 0;
      if (sf_exchange_boundary(xch, fuser, &n, &k, &got) != SF_OK) return die("exchange");
      say("Exchange: %llu boundary blocks sent to rank %d, %llu of rank %d's %llu kept as ghosts -- over %s\n", (unsigned long long)n,
          (a.rank + a.ranks - 1) % a.ranks, (unsigned long long)got, (a.rank + 1) % a.ranks, (unsigned long long)k, sf_exchange_transport(xch));
      sf_exchange_destroy(xch);
    } else {
    const pid_t parent = getppid();
    uint64_t n = 0, got = 0;
    if (sf_fuser_export_boundary(fuser, nullptr, nullptr, 0, &n, 0) != SF_OK) return die("boundary count");
    std::vector<int32_t> coords(n * 3);
    std::vector<uint8_t> voxels(n * 4096);
    uint64_t m = 0;
    if (n && sf_fuser_export_boundary(fuser, coords.data(), voxels.data(), n, &m, 0) != SF_OK) return die("boundary export");
    if (m != n) return die_msg("boundary export wrote %llu of %llu blocks", (unsigned long long)m, (unsigned long long)n);
    if (!write_file(a.ipc + "/b" + std::to_string(a.rank) + ".bin", {{&n, 8}, {coords.data(), n * 12}, {voxels.data(), n * 4096}}))
      return die_msg("could not write the boundary layers into %s", a.ipc.c_str());
    const std::string from = a.ipc + "/b" + std::to_string((a.rank + 1) % a.ranks) + ".bin";
    if (!wait_for(from, a.ipc + "/abort", parent)) return die_msg("gave up waiting for rank %d (another rank failed)", (a.rank + 1) % a.ranks);
    FILE* fp = std::fopen(from.c_str(), "rb");
    uint64_t k = 0;
    bool ok = fp && read_exact(fp, &k, 8);
    if (ok) {
      coords.resize(k * 3);
      voxels.resize(k * 4096);
      ok = read_exact(fp, coords.data(), k * 12) && read_exact(fp, voxels.data(), k * 4096);
    }
    if (fp) std::fclose(fp);
    if (!ok) return die_msg("could not read %s", from.c_str());
    if (k && sf_fuser_import_ghosts(fuser, coords.data(), voxels.data(), k, 0, &got) != SF_OK) return die("ghost import");
    say("Exchange: %llu boundary blocks sent to rank %d, %llu of rank %d's %llu kept as ghosts -- over files in %s (no device-to-device transport)\n", (unsigned long long)n,
        (a.rank + a.ranks - 1) % a.ranks, (unsigned long long)got, (a.rank + 1) % a.ranks, (unsigned long long)k, a.ipc.c_str());
    }
  }
  sf_mesh* mesh = nullptr;
  if (sf_fuser_extract_mesh(fuser, &mesh) != SF_OK) return die("marching cubes");
  uint64_t nv = 0, nf = 0;
  sf_mesh_counts(mesh, &nv, &nf);
  if (part) {
    std::vector<float> xyz(nv * 3);
    std::vector<uint8_t> rgba(nv * 4);
    std::vector<uint64_t> keys(nv), fkeys(nf);
    std::vector<uint32_t> tris(nf * 3);
    if (nv && sf_mesh_copy(mesh, xyz.data(), rgba.data(), tris.data(), keys.data()) != SF_OK) return die("mesh copy");
    if (nf && sf_mesh_copy_face_keys(mesh, fkeys.data()) != SF_OK) return die("face keys");
    const uint64_t hdr[2] = {nv, nf};
    if (!write_file(a.ipc + "/m" + std::to_string(a.rank) + ".bin",
                    {{hdr, 16}, {keys.data(), nv * 8}, {fkeys.data(), nf * 8}, {xyz.data(), nv * 12}, {tris.data(), nf * 12}, {rgba.data(), nv * 4}}))
      return die_msg("could not write the mesh part into %s", a.ipc.c_str());
    say("Mesh part with %llu vertices, %llu faces handed to the parent\n", (unsigned long long)nv, (unsigned long long)nf);
  } else {
    const std::string out = output_path(a);
    if (sf_mesh_write_ply(mesh, out.c_str()) != SF_OK) return die("ply");
    say("Mesh with %llu vertices, %llu faces written to %s\n", (unsigned long long)nv, (unsigned long long)nf, out.c_str());
  }
  sf_mesh_free(mesh);
  sf_fuser_destroy(fuser);
  sf_sens_close(sens);
  return 0;
}

// ---- the parent of a partitioned run: starts the ranks, waits, merges -------------------------------------------------------------------------------
volatile sig_atomic_t g_stop = 0;   // SIGTERM / SIGINT reached the parent: stop the ranks, remove the exchange directory, fail
void on_stop(int) { g_stop = 1; }
void remove_exchange_dir(const std::string& dir, int ranks) {
  for (int r = 0; r < ranks; r++) {
    for (const char* stem : {"b", "m"})
      for (const char* ext : {".bin", ".bin.tmp"}) std::remove((dir + "/" + stem + std::to_string(r) + ext).c_str());
    for (const char* stem : {"dev", "ipc0_", "done0_"})   // the rendezvous notes of sf_exchange_* (one exchange per run: round 0)
      for (const char* ext : {"", ".tmp"}) std::remove((dir + "/" + stem + std::to_string(r) + ext).c_str());
  }
  for (const char* name : {"abort", "nccl.id", "nccl.id.tmp"}) std::remove((dir + "/" + name).c_str());
  ::rmdir(dir.c_str());
}

int run_ranks(const Args& a, int argc_in, const char** argv_in) {
  // fail before anything is started when the file or the parameters are not there: one message instead of N
  for (int i = 1; i <= 3; i++)
    if (FILE* fp = std::fopen(a.pos[i], "r")) std::fclose(fp);
    else return die_msg("could not open %s", a.pos[i]);
  const int base = std::getenv("SF_DEVICE") ? std::atoi(std::getenv("SF_DEVICE")) : 0;
  if (!a.share_gpu) {   // with --share-gpu the ranks find out for themselves (and say "no CPU fallback" when there is no device at all)
    int ndev = 0;
    if (sf_device_count(&ndev) != SF_OK) ndev = 0;   // no driver, no device: the message below says what is missing
    if (base + a.ranks > ndev)
      return die_msg("--ranks %d from device %d needs %d GPUs, %d visible (--share-gpu puts every rank on device %d: a test of the control flow, not a speed-up)",
                     a.ranks, base, base + a.ranks, ndev, base);
  }
  char exe[4096];
  const ssize_t len = ::readlink("/proc/self/exe", exe, sizeof exe - 1);
  if (len <= 0) return die_msg("could not find this executable (/proc/self/exe)");
  exe[len] = 0;
  // the exchange directory: memory-backed /dev/shm when it has room for a scan-sized exchange (boundary layers ~1/16 of the blocks x 4 KiB, then the
  // mesh parts: a few GB for a 50 000-frame scan; containers often give /dev/shm 64 MB), else $TMPDIR or /tmp (short-lived files: page cache)
  struct statvfs vfs;
  const bool shm_has_room = ::statvfs("/dev/shm", &vfs) == 0 && (double)vfs.f_bavail * (double)vfs.f_frsize >= 8e9;
  const char* tmpdir = std::getenv("TMPDIR");
  std::string shm = "/dev/shm/sf_ranks_XXXXXX", tmp = std::string(tmpdir && *tmpdir ? tmpdir : "/tmp") + "/sf_ranks_XXXXXX";
  const char* made = shm_has_room ? ::mkdtemp(&shm[0]) : nullptr;
  if (!made) made = ::mkdtemp(&tmp[0]);
  if (!made) return die_msg("could not create the exchange directory in /dev/shm or %s", tmp.c_str());
  const std::string dir = made;
  say("Partitioned run: %d ranks%s, stripes of %d block layers along x, exchange through %s\n", a.ranks, a.share_gpu ? " sharing one device" : "", STRIPE_BLOCKS, dir.c_str());
  std::fflush(stdout);   // before the ranks write to the same stream
  std::vector<pid_t> pids(a.ranks, -1);
  int failed = 0;
  struct sigaction sa;
  std::memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_stop;   // no SA_RESTART: the waitpid below returns with EINTR
  sigaction(SIGTERM, &sa, nullptr);
  sigaction(SIGINT, &sa, nullptr);
  for (int r = 0; r < a.ranks && !failed; r++) {
    std::vector<std::string> args(argv_in, argv_in + argc_in);   // the command line as given (its --ranks included) ...
    args[0] = exe;
    args.push_back("--rank-of=" + std::to_string(r));            // ... plus what makes the copy a rank
    args.push_back("--exchange-dir=" + dir);
    std::vector<std::string> env;
    for (char** e = environ; *e; e++)
      if (std::strncmp(*e, "SF_DEVICE=", 10) != 0) env.push_back(*e);
    env.push_back("SF_DEVICE=" + std::to_string(a.share_gpu ? base : base + r));
    std::vector<char*> av, ev;
    for (auto& s : args) av.push_back(&s[0]);
    for (auto& s : env) ev.push_back(&s[0]);
    av.push_back(nullptr);
    ev.push_back(nullptr);
    if (::posix_spawn(&pids[r], exe, nullptr, nullptr, av.data(), ev.data()) != 0) { pids[r] = -1; failed = 1; }
  }
  // wait for every rank; the first one that fails stops the others (they may be waiting for its boundary layers)
  int alive = 0;
  for (pid_t p : pids) alive += p > 0;
  while (alive > 0) {
    if (failed == 1) {
      if (FILE* fp = std::fopen((dir + "/abort").c_str(), "w")) std::fclose(fp);
      for (pid_t p : pids)
        if (p > 0) ::kill(p, SIGTERM);
      failed = 2;
    }
    int status = 0;
    const pid_t done = ::waitpid(-1, &status, 0);
    if (done < 0 && errno == EINTR) {
      if (g_stop && !failed) failed = 1;
      continue;
    }
    if (done < 0) break;
    for (pid_t& p : pids)
      if (p == done) {
        p = -1;
        alive--;
        if (!(WIFEXITED(status) && WEXITSTATUS(status) == 0) && !failed) failed = 1;
      }
  }
  if (failed) {
    remove_exchange_dir(dir, a.ranks);
    return die_msg(g_stop ? "stopped by a signal; nothing written" : "a rank of the partitioned run failed (its message is above); nothing written");
  }
  std::vector<sf_mesh*> meshes(a.ranks, nullptr);
  int rc = 0;
  for (int r = 0; r < a.ranks && !rc; r++) {
    Part p;
    if (!map_part(dir + "/m" + std::to_string(r) + ".bin", &p)) { rc = die_msg("could not read the mesh part of rank %d", r); break; }
    if (sf_mesh_create_keyed(p.xyz, p.rgba, p.keys, p.nv, p.tris, p.fkeys, p.nf, &meshes[r]) != SF_OK) rc = die("mesh part");
    p.unmap();   // the handle holds its own copy
  }
  remove_exchange_dir(dir, a.ranks);
  sf_mesh* merged = nullptr;
  if (!rc && sf_mesh_merge_parts(meshes.data(), a.ranks, &merged) != SF_OK) rc = die("merge");
  for (sf_mesh* m : meshes) sf_mesh_free(m);
  if (rc) return rc;
  uint64_t nv = 0, nf = 0;
  sf_mesh_counts(merged, &nv, &nf);
  const std::string out = output_path(a);
  if (sf_mesh_write_ply(merged, out.c_str()) != SF_OK) return die("ply");
  say("Mesh with %llu vertices, %llu faces written to %s\n", (unsigned long long)nv, (unsigned long long)nf, out.c_str());
  sf_mesh_free(merged);
  return 0;
}

}  // namespace

int main(int argc, const char** argv_in) {
  // This PROCESS drives sf_fuse_run's seven streams: ask the HIP runtime for a hardware queue each before its first call (default 4: kernels of
  // streams that share a queue run one after the other).  The application's decision, not the library's; a value the user exported wins.
  (void)setenv("GPU_MAX_HW_QUEUES", "16", 0);
  // ranks of a partitioned run hand device memory to each other (RCCL, hipIpc): the host driver of this platform supports dmabuf IPC only
  (void)setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
  // --upstream[=voxelhashing|bundlefusion]: the upstream-conformance preset (scanfuse.h sf_params_upstream_preset; default: SURVEY App. C).  Not an
  // argument of the tool this replaces -- the pipeline's command line (scan_processor.py:138) stays valid -- and it may stand anywhere.  So may
  // --ranks N / --ranks=N and --share-gpu (see the top of this file); --rank-of= and --exchange-dir= are what the parent of a partitioned run adds.
  Args a;
  bool bad = false;
  for (int i = 0; i < argc; i++) {
    const char* s = argv_in[i];
    if (!std::strcmp(s, "--upstream") || !std::strcmp(s, "--upstream=voxelhashing")) a.upstream = 1;
    else if (!std::strcmp(s, "--upstream=bundlefusion")) a.upstream = 2;
    else if (!std::strcmp(s, "--ranks") && i + 1 < argc) a.ranks = std::atoi(argv_in[++i]);
    else if (!std::strncmp(s, "--ranks=", 8)) a.ranks = std::atoi(s + 8);
    else if (!std::strcmp(s, "--share-gpu")) a.share_gpu = true;
    else if (!std::strncmp(s, "--rank-of=", 10)) a.rank = std::atoi(s + 10);
    else if (!std::strncmp(s, "--exchange-dir=", 15)) a.ipc = s + 15;
    else if (i > 0 && !std::strncmp(s, "--", 2)) bad = true;
    else if (a.n_pos < 8) a.pos[a.n_pos++] = s;
  }
  if (a.n_pos < 4 || bad || a.ranks < 1 || a.ranks > 64 || (a.rank >= 0 && (a.rank >= a.ranks || a.ipc.empty()))) {
    std::printf("Usage: depthsensing [--upstream[=voxelhashing|bundlefusion]] [--ranks N [--share-gpu]] <zParameters.txt> <zParametersTracking.txt> <scan.sens> [out.ply]\n");
    return 255;
  }
  if (a.rank >= 0) {
    std::snprintf(g_prefix, sizeof g_prefix, "[rank %d/%d] ", a.rank, a.ranks);
    std::setvbuf(stdout, nullptr, _IOLBF, 0);   // whole lines into the stream the ranks share
    return fuse_scan(a);
  }
  if (a.ranks > 1) return run_ranks(a, argc, argv_in);
  return fuse_scan(a);
}
