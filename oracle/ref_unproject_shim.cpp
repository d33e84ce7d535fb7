// oracle/ref_unproject_shim.cpp -- TEST INFRASTRUCTURE.  The reference's own depth unprojection, SURVEY 8a row a6, as host code:
//   * the matrix arithmetic is the REFERENCE's -- float4x4, its operator*(float4) and getInverse() come from
//     AnnotationTools/Filter2dAnnotations/cuda_SimpleMatrixUtil.h, compiled where it lies (oracle/Makefile streams it into the compiler behind
//     oracle/cuda_shim/; this file is appended to that stream, which is why it includes nothing of the reference itself);
//   * the per-pixel statement is the body of convertDepthFloatToCameraSpaceFloat4Device, filter.cu:74-91, restated -- a __global__ kernel whose
//     file also holds <<< >>> launches cannot be compiled by a host compiler:
//         float4 cameraSpace(intrinsicsInv * make_float4((float)x*depth, (float)y*depth, depth, depth));
//         d_output[y*width+x] = make_float4(cameraSpace.x, cameraSpace.y, cameraSpace.w, 1.0f);
// tests/test_oracle_tsdf.py bounds the difference between this form, K^-1 . (x d, y d, d), and the ray-slope form the fusion kernels and
// oracle/tsdf_oracle.c use, ((x - mx) / fx) d, over every pixel of a 640x480 image and the sensor's depth range.
extern "C" {

// K16: the 4x4 intrinsic matrix, row-major (fx 0 mx 0 / 0 fy my 0 / 0 0 1 0 / 0 0 0 1); depth: width*height metres, -inf = invalid;
// out_xyz: width*height*3
void ref_unproject(const float* K16, unsigned width, unsigned height, const float* depth, float* out_xyz) {
  const float4x4 K(K16);
  const float4x4 intrinsicsInv = K.getInverse();   // cuda_SimpleMatrixUtil.h:944
  for (unsigned y = 0; y < height; y++)
    for (unsigned x = 0; x < width; x++) {
      float* o = out_xyz + 3 * ((size_t)y * width + x);
      o[0] = o[1] = o[2] = MINF;
      const float d = depth[(size_t)y * width + x];
      if (d != MINF) {
        const float4 cameraSpace(intrinsicsInv * make_float4((float)x * d, (float)y * d, d, d));
        o[0] = cameraSpace.x; o[1] = cameraSpace.y; o[2] = cameraSpace.w;
      }
    }
}

// the inverse the reference's class computes, for the record (16 floats, row-major)
void ref_intrinsics_inverse(const float* K16, float* out16) {
  const float4x4 inv = float4x4(K16).getInverse();
  for (int i = 0; i < 16; i++) out16[i] = inv.ptr()[i];
}

}  // extern "C"
