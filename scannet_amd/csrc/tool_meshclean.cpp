// tool_meshclean.cpp -- drop-in for the meshlabserver invocations of the `improve` stage that only clean:
//     meshlabserver -i <in.ply> -o <out.ply> -m vc -s <dir>/clean.mlx          (Server/scan_processor.py:143)
//     meshlabserver -i <in.ply> -o <in.ply>  -m vc -s <dir>/cleanLoRes.mlx     (Server/scan_processor.py:134)
// Point cfg.MESHLAB_BIN (Server/config.py:19) at this executable.  Same flags: -i input, -o output, -m save-mask
// tokens (vc = vertex colours: always written), -s filter script.  Progress on stdout, nothing on stderr on success,
// non-zero exit + stderr message on failure (Server/util.py:38-50).  The same executable serves the decimate stage:
//     meshlabserver -i X_vh_clean.ply -o X_vh_clean_1.ply -m vc -s <dir>/simplify.mlx   (Server/scan_processor.py:144-145)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "scanfuse.h"

static int die(const char* what) {
  std::fprintf(stderr, "%s: %s\n", what, sf_last_error());
  return 1;
}

int main(int argc, const char** argv) {
  const char *in = nullptr, *out = nullptr, *script = nullptr;
  int gpu = -1;   // --gpu [device]: the quadric collapse of simplify.mlx runs as rounds of independent collapses on that MI355X
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "-i" && i + 1 < argc) in = argv[++i];
    else if (a == "-o" && i + 1 < argc) out = argv[++i];
    else if (a == "-s" && i + 1 < argc) script = argv[++i];
    else if (a == "-m") { while (i + 1 < argc && argv[i + 1][0] != '-') i++; }  // vc vn fc ... : colours are always saved
    else if (a == "--gpu") { gpu = 0; if (i + 1 < argc && argv[i + 1][0] >= '0' && argv[i + 1][0] <= '9') gpu = std::atoi(argv[++i]); }  // not a meshlabserver flag: opt-in
    else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 255; }
  }
  if (!in || !out || !script) {
    std::printf("Usage: meshclean -i input.ply -o output.ply [-m vc] -s clean.mlx|simplify.mlx [--gpu [device]]\n");
    return 255;
  }
  sf_clean_script sc;
  if (sf_mlx_load(script, &sc) != SF_OK) return die("filter script");
  sc.simplify_device = gpu;
  sc.clean_device = gpu;
  sf_mesh* m = nullptr;
  if (sf_ply_read(in, &m) != SF_OK) return die("input mesh");
  uint64_t nv = 0, nf = 0;
  sf_mesh_counts(m, &nv, &nf);
  std::printf("Mesh %s loaded has %llu vn %llu fn\n", in, (unsigned long long)nv, (unsigned long long)nf);
  sf_mesh* c = nullptr;
  sf_clean_stats st;
  if (sf_mesh_clean_script(m, &sc, &c, &st) != SF_OK) return die("clean");
  if (sc.simplify) {
    const sf_simplify_stats& ss = sc.simplify_stats;
    std::printf("Quadric Edge Collapse Decimation: %llu -> %llu faces (target %llu), %llu collapses, %llu vertices left\n",
                (unsigned long long)ss.faces_in, (unsigned long long)ss.faces_out, (unsigned long long)ss.target_faces,
                (unsigned long long)ss.collapses, (unsigned long long)ss.vertices_out);
  }
  std::printf("Merge Close Vertices (%g): merged %llu vertices, removed %llu degenerate faces\n", (double)sc.merge_distance,
              (unsigned long long)st.vertices_merged, (unsigned long long)st.faces_degenerate);
  std::printf("Remove Duplicate Faces: removed %llu faces\n", (unsigned long long)st.faces_duplicate);
  std::printf("Remove Isolated pieces (< %u faces): removed %llu connected components out of %llu (%llu faces)\n", sc.min_component_faces,
              (unsigned long long)st.components_removed, (unsigned long long)st.components_in, (unsigned long long)st.faces_small_component);
  std::printf("Remove Unreferenced Vertex: removed %llu vertices\n", (unsigned long long)st.vertices_unreferenced);
  if (sf_mesh_write_ply(c, out) != SF_OK) return die("output mesh");
  std::printf("Mesh saved as %s (%llu vn %llu fn)\n", out, (unsigned long long)st.vertices_out, (unsigned long long)st.faces_out);
  sf_mesh_free(c);
  sf_mesh_free(m);
  return 0;
}
