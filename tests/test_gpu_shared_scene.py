"""-m gpu: one copy of the derived scene data per device however many contexts render it (VERDICT r2 "What's weak" 4,
ADVICE r2 items 1 and 4), the shared content version of aliased device memory, and a clwh_render that never waits for
the GPU (the reference's single in-order queue has no readback between camera and pass,
app/renderer.cpp:145-150)."""
import numpy as np
import pytest

from cl_volume_renderer_amd import ffi, scene
from tests.gpu_util import GpuScene, look_at_centre, small_scene

pytestmark = pytest.mark.gpu


def _lane(ctx, g0, tf, wh):
    """a second frame lane as bench.py builds it: its own context and stream over the SAME volume / SDF / env-map memory"""
    X, Y, Z = g0.dims
    w, h = wh
    vol = ctx.image_wrap(g0.volume.device_ptr, [X, Y, Z], 1, np.int16, (Z, Y, X))
    sdf = ctx.image_wrap(g0.sdf.device_ptr, [X, Y, Z], 1, np.int8, (Z, Y, X))
    env = ctx.image_wrap(g0.env.device_ptr, [g0.env.shape[1], g0.env.shape[0]], 4, np.uint8, g0.env.shape)
    accum = ctx.buffer(ffi.accum_len(w, h, 1) * 16, np.float32)
    ctx.buffer_reset(accum)
    k = ctx.kernel("ray_marching.cl", "render", tf)
    return dict(ctx=ctx, vol=vol, sdf=sdf, env=env, accum=accum, k=k)


def _render_lane(ln, pos, d, seeds, wh):
    ln["k"].render(frame=None, volume=ln["vol"], sdf=ln["sdf"], env=ln["env"], accum=ln["accum"], cam_pos=pos, cam_dir=d,
                   seed=0, seeds=seeds, width=wh[0], height=wh[1], mode=ffi.ACCUM_IMAGE_SPACE, write_frame=False)


def test_contexts_share_one_copy_of_the_derived_scene(orc):
    vol, sdf, env, tf = small_scene(orc, 64)
    pos, d = look_at_centre(vol, [-25, 50, -30])
    wh = (128, 96)
    seeds = scene.glibc_rand(8)
    ctx0, ctx1 = ffi.Context(0), ffi.Context(0)
    g0 = GpuScene(ctx0, vol, sdf, env, tf, wh)
    ln = _lane(ctx1, g0, tf, wh)
    g0.render(pos, d, None, mode=ffi.ACCUM_IMAGE_SPACE, seeds=seeds, debug=False, write_frame=False)
    id0, bytes0, holders0 = ctx0.scene_info()
    assert id0 != 0 and bytes0 >= 9 * 64 ** 3 and holders0 == 1
    _render_lane(ln, pos, d, seeds, wh)
    id1, bytes1, holders1 = ctx1.scene_info()
    assert (id1, bytes1) == (id0, bytes0) and holders1 == 2          # adopted, not rebuilt
    ctx0.finish()
    ctx1.finish()
    want = g0.accum[0].pull(np.float32)
    assert want.reshape(-1, 4)[:, 3].max() == 8.0
    assert np.array_equal(ln["accum"].pull(np.float32), want)

    # lane 0 rebuilds the SDF in place (a transfer-function flush: another TF, same memory): the alias in lane 1 sees the new
    # content version and must not keep rendering from the old step bytes
    tf2 = scene.tf_rect_source([(30.0, 1200.0, 0.0, 4000.0, (1.0, 0.5, 0.25, 0.5))])
    ctx0.sdf_build(g0.volume, tf2, g0.sdf)
    ctx0.finish()
    k0 = ctx0.kernel("ray_marching.cl", "render", tf2)
    ln["k"].release()
    ln["k"] = ctx1.kernel("ray_marching.cl", "render", tf2)
    ctx1.buffer_reset(ln["accum"])
    _render_lane(ln, pos, d, seeds, wh)                                  # lane 1 first this time: it builds, lane 0 adopts
    ctx1.finish()
    id1b, _, _ = ctx1.scene_info()
    assert id1b != id0
    ctx0.buffer_reset(g0.accum[0])
    k0.render(frame=None, volume=g0.volume, sdf=g0.sdf, env=g0.env, accum=g0.accum[0], cam_pos=pos, cam_dir=d, seed=0, seeds=seeds,
              width=wh[0], height=wh[1], mode=ffi.ACCUM_IMAGE_SPACE, write_frame=False)
    ctx0.finish()
    id0b, _, holders = ctx0.scene_info()
    assert id0b == id1b and holders == 2
    got0, got1 = g0.accum[0].pull(np.float32), ln["accum"].pull(np.float32)
    assert np.array_equal(got0, got1)
    # ... and it is the right content: the oracle with the new TF on the new SDF
    sdf2, _, _ = orc.sdf_build(vol, orc.parse_tf(tf2))
    assert np.array_equal(g0.sdf.pull(), sdf2)
    o = orc.Scene(vol, sdf2, env, orc.parse_tf(tf2), wh, mode=orc.MODE_IMAGE_SPACE)
    for s in seeds:
        o.render(pos, d, s)
    acc = got0.reshape(wh[1] // 8, wh[0] // 8, 8, 8, 4).transpose(0, 2, 1, 3, 4).reshape(wh[1], wh[0], 4)
    assert np.array_equal(acc, o.accum)
    assert not np.array_equal(got0, want)

    # mark_dirty through the alias invalidates the owner's context too; invalidate_derived(scene) rebuilds for one context only
    ffi._check(ffi.lib().clwh_mem_mark_dirty(ln["vol"].h), "clwh_mem_mark_dirty")
    k0.render(frame=None, volume=g0.volume, sdf=g0.sdf, env=g0.env, accum=g0.accum[0], cam_pos=pos, cam_dir=d, seed=0, seeds=seeds[:1],
              width=wh[0], height=wh[1], mode=ffi.ACCUM_IMAGE_SPACE, write_frame=False)
    ctx0.finish()
    assert ctx0.scene_info()[0] not in (id0, id0b)
    for m in (ln["vol"], ln["sdf"], ln["env"], ln["accum"]):
        m.release()
    ln["k"].release()
    k0.release()
    g0.release()
    ctx1.destroy()
    ctx0.destroy()


def test_render_does_not_wait_for_the_gpu(orc):
    """A multi-seed image-space launch right after a camera move used to copy the hit count to the host and wait for it.
    Now every call returns while the stream still holds earlier work: a long run of memsets is queued first, then two
    cameras' worth of launches, and the stream must still be busy when the calls have returned.  The results are the
    oracle's (work buffers were sized from an ESTIMATE of the hit count)."""
    import torch

    vol, sdf, env, tf = small_scene(orc, 48)
    wh = (256, 192)
    stream = torch.cuda.Stream()
    ctx = ffi.Context(0, stream=stream.cuda_stream)
    g = GpuScene(ctx, vol, sdf, env, tf, wh)
    seeds = scene.glibc_rand(12)
    cams = [look_at_centre(vol, p) for p in ([-20, 40, -30], [70, 35, -25], [-22, 41, -30])]
    g.render(*cams[0], None, mode=ffi.ACCUM_IMAGE_SPACE, seeds=seeds, debug=False, write_frame=False)   # warm-up: allocations
    ctx.finish()
    n_acc = ffi.accum_len(*wh) * 4
    with torch.cuda.stream(stream):
        ballast = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        accs = [torch.zeros(n_acc, dtype=torch.float32, device="cuda") for _ in cams[1:]]
    torch.cuda.synchronize()
    mems = [ctx.wrap(a.data_ptr(), n_acc * 4) for a in accs]
    with torch.cuda.stream(stream):
        for _ in range(60):        # 60 GiB of memsets: tens of milliseconds of queued work
            ballast.zero_()
        for (pos, d), m_acc in zip(cams[1:], mems):
            for k in range(0, 12, 4):
                g.kernel.render(frame=None, volume=g.volume, sdf=g.sdf, env=g.env, accum=m_acc, cam_pos=pos, cam_dir=d, seed=0,
                                seeds=seeds[k:k + 4], width=wh[0], height=wh[1], mode=ffi.ACCUM_IMAGE_SPACE, write_frame=False)
        busy = not stream.query()
    assert busy, "clwh_render waited for the GPU somewhere: the stream drained before the calls returned"
    stream.synchronize()
    ctx.finish()   # raises if a fix-up buffer sized from the estimated hit count overflowed
    for (pos, d), a, m in zip(cams[1:], accs, mems):
        o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), wh, mode=orc.MODE_IMAGE_SPACE)
        for s in seeds:
            o.render(pos, d, s)
        acc = a.cpu().numpy().reshape(wh[1] // 8, wh[0] // 8, 8, 8, 4).transpose(0, 2, 1, 3, 4).reshape(wh[1], wh[0], 4)
        assert np.array_equal(acc, o.accum)
        assert o.accum[..., 3].max() == 12.0
        m.release()
    g.release()
    ctx.destroy()
