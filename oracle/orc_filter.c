/* oracle/orc_filter.c -- CPU restatement of the 5x5x5 bilateral volume filter.  TEST INFRASTRUCTURE ONLY
 * (see orc.h): nothing under cl_volume_renderer_amd/ may call or link this.
 *
 * Follows  opencl_kernels/utility_filter.cl:38-62 (bilateral_kernel)
 *          opencl_kernels/volume_filter.cl:5-11   (bilateral_filter)
 * Parity unpinned: the reference holds no fixture for this kernel and needs an OpenCL runtime to run.
 *
 * Fixed semantics (same conventions as orc_render.c): read_imagei outside the image returns 0 (border of
 * CLK_ADDRESS_CLAMP), integer coordinates address the texel of that index; float arithmetic is IEEE binary32
 * without contraction; pow and exp are evaluated in binary64 by libm and rounded once; the float -> short
 * conversion of the return value truncates; write_imagei saturates to the CL_SIGNED_INT16 range. */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

static inline int32_t fetch0(const int16_t *v, int32_t X, int32_t Y, int32_t Z, int32_t x, int32_t y, int32_t z) {
  if (x < 0 || y < 0 || z < 0 || x >= X || y >= Y || z >= Z) return 0;
  return v[((size_t)z * (size_t)Y + (size_t)y) * (size_t)X + (size_t)x];
}

void orc_bilateral_filter(const int16_t *vol, int32_t X, int32_t Y, int32_t Z, int16_t *out) {
  const float sigmas = 0.6f, sigmar = 1.0f; /* utility_filter.cl:43-44 */
  const int radius = 2;
#pragma omp parallel for collapse(2) schedule(static)
  for (int32_t pz = 0; pz < Z; ++pz)
    for (int32_t py = 0; py < Y; ++py)
      for (int32_t px = 0; px < X; ++px) {
        float out_colour = 0;
        const float mid_colour = (float)fetch0(vol, X, Y, Z, px, py, pz);
        float wp = 0;
        for (int z = -radius; z <= radius; ++z)
          for (int y = -radius; y <= radius; ++y)
            for (int x = -radius; x <= radius; ++x) {
              const float local_colour = (float)fetch0(vol, X, Y, Z, px + x, py + y, pz + z);
              const float posd = ((float)(x * x + y * y + z * z)) / (2 * sigmas * sigmas);
              const float diff = mid_colour - local_colour;
              const float cold = ((float)pow((double)diff, 2.0)) / (2 * sigmar * sigmar);
              const float w = (float)exp((double)(-posd - cold));
              wp += w;
              out_colour += local_colour * w;
            }
        const float q = out_colour / wp;
        int32_t s = (int32_t)q; /* short bilateral_kernel(...): float -> short truncates */
        if (s > 32767) s = 32767;
        if (s < -32768) s = -32768;
        out[((size_t)pz * (size_t)Y + (size_t)py) * (size_t)X + (size_t)px] = (int16_t)s;
      }
}
