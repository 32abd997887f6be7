"""Helpers shared by the -m gpu parity tests: upload a scene through the C ABI and run the same
passes on the HIP path and on the oracle."""
import numpy as np

from cl_volume_renderer_amd import ffi, scene


class GpuScene:
    def __init__(self, ctx, vol, sdf, env, tf_source, frame_wh, launch_wh=None, world=1, padded_cache=True):
        self.ctx = ctx
        Z, Y, X = vol.shape
        self.dims = (X, Y, Z)
        self.frame_w, self.frame_h = frame_wh
        self.launch_w, self.launch_h = launch_wh or frame_wh
        self.volume = ctx.image_from(vol.astype(np.int16))
        self.sdf = ctx.image_from(sdf.astype(np.int8)) if sdf is not None else ctx.image([X, Y, Z], 1, np.int8, (Z, Y, X))
        self.env = ctx.image_from(env.astype(np.uint8), channels=4)
        self.frame = ctx.image([self.frame_w, self.frame_h], 4, np.uint8, (self.frame_h, self.frame_w, 4))
        self.frame.push(np.zeros((self.frame_h, self.frame_w, 4), np.uint8))
        n_cache = ffi.cache_len(X, Y, Z) if padded_cache else X * Y * Z * 4
        self.cache = ctx.buffer(n_cache * 2, np.uint16)
        ctx.buffer_reset(self.cache)
        npx = self.launch_w * self.launch_h
        self.world = world
        self.accum = [ctx.buffer(ffi.accum_len(self.launch_w, self.launch_h, world) * 16, np.float32)
                      for _ in range(world)]
        for a in self.accum:
            ctx.buffer_reset(a)
        self.hit_index = ctx.buffer(npx * 8, np.int64)
        self.contrib = ctx.buffer(npx * 16, np.uint32, (npx, 4))
        self.kernel = ctx.kernel("ray_marching.cl", "render", tf_source)
        self.tf_source = tf_source

    def render(self, pos, d, seed, mode=ffi.ACCUM_VOXEL_CACHE, rank=0, write_frame=True, debug=True, seeds=None,
               shading=ffi.SHADE_LIGHT):
        """seed: one pass; seeds: several passes in one launch (then no per-pixel contribution output)"""
        self.kernel.render(frame=self.frame, volume=self.volume, sdf=self.sdf, env=self.env,
                           buffer_volume=self.cache, accum=self.accum[rank], cam_pos=pos, cam_dir=d,
                           seed=seed if seeds is None else 0, seeds=seeds,
                           width=self.launch_w, height=self.launch_h, mode=mode, tile_rank=rank,
                           tile_world=self.world, write_frame=write_frame,
                           hit_index=self.hit_index if debug else None,
                           contrib=self.contrib if (debug and seeds is None) else None, shading=shading)

    def accum_row_major(self, rank=0):
        """tile-major float4 accumulation -> [h][w][4] with zeros for tiles the rank does not own."""
        a = self.accum[rank].pull(np.float32).reshape(-1, 64, 4)
        w, h, W = self.launch_w, self.launch_h, self.world
        tiles_x, tiles_y = w // 8, h // 8
        per_row = (tiles_x + W - 1) // W
        out = np.zeros((h, w, 4), np.float32)
        for ty in range(tiles_y):
            for tx in range(tiles_x):
                if (tx + ty) % W != rank:
                    continue
                slot = ty * per_row + tx // W
                out[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8] = a[slot].reshape(8, 8, 4)
        return out

    def release(self):
        for m in [self.volume, self.sdf, self.env, self.frame, self.cache, self.hit_index, self.contrib] + self.accum:
            m.release()
        self.kernel.release()


def small_scene(orc, n=64, dims=None, tf_source=None, env_wh=(256, 128)):
    tf_source = tf_source or scene.tf_default_source()
    vol = scene.phantom(n, dims=dims)
    sdf, _, _ = orc.sdf_build(vol, orc.parse_tf(tf_source))
    env = scene.env_map(*env_wh)
    return vol, sdf, env, tf_source


def look_at_centre(vol, pos):
    Z, Y, X = vol.shape
    pos = np.asarray(pos, np.float32)
    d = np.array([X / 2, Y / 2, Z / 2], np.float32) - pos
    return pos, (d / np.linalg.norm(d)).astype(np.float32)
