// mesh.h -- internal mesh container behind the sf_mesh handle (PLY I/O, marching-cubes output, Segmentator input)
#pragma once
#include <cstdint>
#include <memory>
#include <new>
#include <utility>
#include <vector>

#include "common.h"

// The arrays of a scan-sized mesh are hundreds of MB that are about to be overwritten (a download, a file read, a filter's output): resize()
// must not zero-fill them first -- one thread touching 245 MB twice was a third of a marching-cubes extraction.  Default-initialising allocator:
// value-initialisation becomes default-initialisation (a no-op for the arithmetic types used here); everything else is std::allocator.
namespace sf {
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
    else ::new ((void*)p) U(std::forward<A>(a)...);
  }
};
template <class T> using mesh_vec = std::vector<T, NoInitAlloc<T>>;
}  // namespace sf

struct sf_mesh {
  sf::mesh_vec<float> pos;      // 3 per vertex
  sf::mesh_vec<uint8_t> col;    // 4 per vertex (r,g,b,a); empty if the source had no colour
  sf::mesh_vec<uint32_t> tri;   // 3 per face
  sf::mesh_vec<uint64_t> keys;  // optional: canonical edge key per vertex (marching-cubes output)
  sf::mesh_vec<uint64_t> tkeys; // optional: cube key per face, ascending (marching-cubes output): merging the meshes of a partitioned scan
};
