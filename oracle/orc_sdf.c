/*
 * oracle/orc_sdf.c -- CPU restatement of the reference's SDF builder.
 * TEST INFRASTRUCTURE ONLY (see orc.h).  PINNED: reproduces the reference's golden vector
 * tests/sdf/values.x for tests/sdf/testdata.nrrd bit-exactly (tests/test_oracle_sdf.py).
 *
 * Follows app/signed_distance_field.cpp:7-35 (host loop) and
 * opencl_kernels/signed_distance_field.cl:6-54 (create_base_image), :56-87
 * (neightbour_distance_calc), :89-112 (create_signed_distance_field).
 */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

typedef struct {
  const int16_t *vol;
  int32_t X, Y, Z;
  const orc_tf *tf;
  int uses_gradient;
} sdf_ctx;

static inline int32_t vol_at(const sdf_ctx *c, int32_t x, int32_t y, int32_t z) {
  /* read_imagei with CLK_ADDRESS_CLAMP / int coords: out of range -> border 0 */
  if (x < 0 || y < 0 || z < 0 || x >= c->X || y >= c->Y || z >= c->Z) return 0;
  return c->vol[((int64_t)z * c->Y + y) * c->X + x];
}

static inline int32_t f2i_sat(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return INT32_MAX;
  if (v <= -2147483648.0f) return INT32_MIN;
  return (int32_t)v;
}

/* is_event_gen(value, length(gradient_prewitt_nn(volume, make_float(pos))), &color)
 * signed_distance_field.cl:13-20, 36-38 ; utility_filter.cl:2-35 */
static inline int event_at(const sdf_ctx *c, int32_t x, int32_t y, int32_t z) {
  int32_t value = vol_at(c, x, y, z);
  int32_t gradient = 0;
  if (c->uses_gradient) {
    float dx = (float)(vol_at(c, x + 1, y, z) - vol_at(c, x - 1, y, z));
    float dy = (float)(vol_at(c, x, y + 1, z) - vol_at(c, x, y - 1, z));
    float dz = (float)(vol_at(c, x, y, z + 1) - vol_at(c, x, y, z - 1));
    float len = sqrtf((dx * dx + dy * dy) + dz * dz);
    gradient = (int16_t)f2i_sat(len);
  }
  int32_t color[4] = {0, 0, 0, 0};
  return orc_tf_eval(c->tf, value, gradient, color);
}

/* signed_distance_field.cl:6-54 */
static void create_base_image(const sdf_ctx *c, int8_t *ping, int8_t *pong, uint32_t max_iterations) {
  for (int32_t z = 0; z < c->Z; ++z)
    for (int32_t y = 0; y < c->Y; ++y)
      for (int32_t x = 0; x < c->X; ++x) {
        int is_event_result = event_at(c, x, y, z);
        int32_t resulting_value = is_event_result ? -1 : 1;
        int homogenous = 1;
        for (int ox = -1; ox < 2; ox++)
          for (int oy = -1; oy < 2; oy++)
            for (int oz = -1; oz < 2; oz++)
              if (ox != 0 && oy != 0 && oz != 0) { /* the 8 corner neighbours only */
                int32_t nx = clampi(x + ox, 0, c->X - 1);
                int32_t ny = clampi(y + oy, 0, c->Y - 1);
                int32_t nz = clampi(z + oz, 0, c->Z - 1);
                homogenous &= (event_at(c, nx, ny, nz) == is_event_result);
              }
        if (homogenous) resulting_value *= (int32_t)max_iterations;
        int64_t i = ((int64_t)z * c->Y + y) * c->X + x;
        /* write_imagei on CL_SIGNED_INT8 saturates; |value| <= 127 here */
        ping[i] = (int8_t)resulting_value;
        pong[i] = (int8_t)resulting_value;
      }
}

/* signed_distance_field.cl:56-87 */
static inline int32_t neighbour_distance_calc(const int8_t *img, int32_t X, int32_t Y, int32_t Z,
                                              int32_t x, int32_t y, int32_t z) {
  int8_t neighbour_distance = 127; /* CHAR_MAX */
  int32_t abs_added = 0, added = 0;
  for (int ox = -1; ox < 2; ox++)
    for (int oy = -1; oy < 2; oy++)
      for (int oz = -1; oz < 2; oz++)
        if (ox != 0 && oy != 0 && oz != 0) {
          int32_t nx = clampi(x + ox, 0, X - 1);
          int32_t ny = clampi(y + oy, 0, Y - 1);
          int32_t nz = clampi(z + oz, 0, Z - 1);
          int32_t v = img[((int64_t)nz * Y + ny) * X + nx];
          int8_t tmp = (int8_t)abs(v);
          abs_added += tmp;
          added += v;
          if (tmp < neighbour_distance) neighbour_distance = tmp;
        }
  if (abs(added) == abs_added) return neighbour_distance;
  return 0;
}

/* signed_distance_field.cl:89-112, one launch over the whole volume */
static int32_t sdf_layer(const int8_t *in, int8_t *out, int32_t X, int32_t Y, int32_t Z,
                         int32_t iteration, int32_t max_iterations) {
  int32_t counter = 0;
  for (int32_t z = 0; z < Z; ++z)
    for (int32_t y = 0; y < Y; ++y)
      for (int32_t x = 0; x < X; ++x) {
        int64_t i = ((int64_t)z * Y + y) * X + x;
        int32_t local_value = in[i];
        int32_t absolut_current_value = abs(local_value);
        if (absolut_current_value < iteration) continue;
        if (absolut_current_value > iteration) {
          int8_t mul = local_value < 0 ? -1 : 1;
          int8_t nd = (int8_t)neighbour_distance_calc(in, X, Y, Z, x, y, z);
          if (nd != 0 && nd == iteration) {
            absolut_current_value = iteration + 1;
            local_value = absolut_current_value * mul;
          }
        }
        if (absolut_current_value < max_iterations) {
          out[i] = (int8_t)local_value;
          counter++;
        }
      }
  return counter;
}

int orc_sdf_build(const int16_t *volume, int32_t X, int32_t Y, int32_t Z, const orc_tf *tf,
                  int8_t *sdf_out, int32_t *n_launches, int32_t *layer_counts) {
  if (!volume || !tf || !sdf_out || X <= 0 || Y <= 0 || Z <= 0) return -1;
  sdf_ctx c = {volume, X, Y, Z, tf, 0};
  for (int k = 0; k < tf->n; ++k)
    if (tf->rules[k].use_gradient) c.uses_gradient = 1;

  /* signed_distance_field.cpp:11 */
  int32_t maxdim = X > Y ? X : Y;
  if (Z > maxdim) maxdim = Z;
  uint32_t max_iterations = (uint32_t)(maxdim / 2);
  if (max_iterations > 127u) max_iterations = 127u;

  const int64_t n = (int64_t)X * Y * Z;
  int8_t *pong_buf = (int8_t *)calloc((size_t)n, 1);
  if (!pong_buf) return -2;
  memset(sdf_out, 0, (size_t)n);
  create_base_image(&c, sdf_out, pong_buf, max_iterations);

  int8_t *ping = sdf_out, *pong = pong_buf;
  int32_t launches = 0;
  /* signed_distance_field.cpp:22-32 */
  for (uint32_t i = 1; i <= max_iterations + (max_iterations % 2) + 1; ++i) {
    int32_t counter = sdf_layer(ping, pong, X, Y, Z, (int32_t)i, (int32_t)max_iterations);
    if (layer_counts) layer_counts[launches] = counter;
    launches++;
    int8_t *t = ping; ping = pong; pong = t;
    if (counter == 0 && (i % 2) == 1) break;
  }
  /* After the loop the reference returns the member `sdf` (= sdf_out here) whatever `ping` is.
   * When the loop ends by the odd-iteration break, the freshly written image is `pong` after
   * the swap == sdf_out.  If it runs to the bound instead, the member is still what is returned. */
  if (n_launches) *n_launches = launches;
  free(pong_buf);
  return 0;
}
