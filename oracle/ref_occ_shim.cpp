// oracle/ref_occ_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" wrapper around the REFERENCE's own Occipital depth codec and shift table, the header-only
// /root/reference/ScannerApp/depth2pgm/uplinksimple_image-codecs.h and uplinksimple_shift2depth.h, compiled from where they
// lie (oracle/Makefile, -I path); no reference source is copied into this repository.  oracle/_ref/libref_occ.so is the
// black-box oracle for sf_occ_decode / sf_occ_encode / sf_occ_shift2depth and the "reference" CPU decode baseline.
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "uplinksimple_image-codecs.h"
#include "uplinksimple_shift2depth.h"

extern "C" {
// the caller pads the stream (the reference reads up to two bytes past a stream's last code)
void ref_occ_decode(const uint8_t* stream, uint32_t stream_bytes, int num_elements, uint16_t* out) {
  uplinksimple::decode(stream, stream_bytes, num_elements, out);
}
uint32_t ref_occ_encode(const uint16_t* in, int num_elements, uint8_t* out, uint32_t cap) {
  return uplinksimple::encode(in, num_elements, out, cap);
}
uint16_t ref_occ_shift2depth(uint16_t s) { return uplinksimple::shift2depth(s); }
// decode + shift2depth + invalid -> 0, the frame loop of ScannerApp/depth2pgm/depth2pgm.cpp:57-65 and Converter/main.cpp:85-93
void ref_occ_frame(const uint8_t* stream, uint32_t stream_bytes, int num_elements, uint16_t* out) {
  uplinksimple::decode(stream, stream_bytes, num_elements, out);
  uplinksimple::shift2depth(out, (size_t)num_elements);
  for (int i = 0; i < num_elements; i++)
    if (out[i] >= uplinksimple::shift2depth(0xffff)) out[i] = 0;
}
}
