"""GPU parity tests (-m gpu) of the round-3 features, all through the C ABI, bit-exact: progressive passes over several devices
(mp_render_pass_multi, mp_untile_preview), object groups that outlive their members' handles."""
import numpy as np
import pytest

import minipath_amd as mp
from tests import meshes
from tests.conftest import TEAPOT

pytestmark = pytest.mark.gpu
SEED = 0x5EED


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return mp.Context(0)


@pytest.mark.parametrize("n,chunked", [(2, False), (3, True)])
def test_progressive_passes_over_several_devices(oracle, teapot_oracle_bvh, n, chunked):
    """BASELINE configs[4]'s real shape (VERDICT r2 #4b; SURVEY 8e: "accumulators stay sharded; gather only per displayed pass /
    at end"): mp_render_pass_multi adds ragged passes to the shards that stay on their devices; a gather after the last pass is
    the single-launch frame bit for bit (= the oracle's), a gather before it is the preview mp_untile_preview defines -- checked
    against the same formula applied to a one-device FrameRenderer's running state; passes that do not continue the state are
    refused.  The n contexts share this box's GPU."""
    import torch

    ctxs = [mp.Context(0) for _ in range(n)]
    scenes = [mp.Scene(mp.TriangleBvh.with_obj(TEAPOT, c)) for c in ctxs]
    cam = mp.Camera.teapot_view()
    res, spp, ts = (200, 136), 600 if chunked else 24, 16
    depth = 3
    st = mp.RenderSettings(ts, spp, res, seed=SEED, max_depth=depth, chunked_sum=chunked)
    passes = (300, 13, 0) if chunked else (5, 12, 0)   # the chunked case cuts a 256-sample chunk in the middle
    # the expected frame: one launch on one device (itself pinned against the oracle in the round-1/2 suites; and here)
    one = mp.FrameRenderer(scenes[0], cam, st)
    one.render()
    full, _ = one.untile()
    torch.cuda.synchronize()
    full = full.cpu().numpy()
    if not chunked:
        of, _, _, _ = teapot_oracle_bvh.render_image_paths_mt(oracle.build_sampler(oracle.teapot_camera(), *res), res[0], res[1], spp, SEED, depth, ts, 8)
        assert np.array_equal(bits(full), bits(of))
    # the running state after each pass on one device -> expected previews
    ref = mp.FrameRenderer(scenes[0], cam, st)
    mf = mp.MultiDeviceFrame(scenes, cam, st)
    nxt = 0
    for k, count in enumerate(passes):
        nxt_ref = ref.render_pass(nxt, count)
        torch.cuda.synchronize()
        if k == 1:  # a pass that stays on the devices: nothing gathered
            assert mf.render_pass(nxt, count, gather=False) == nxt_ref
            nxt = nxt_ref
            continue
        nxt2, img, img8 = mf.render_pass(nxt, count)
        torch.cuda.synchronize()
        assert nxt2 == nxt_ref
        nxt = nxt2
        got = img.cpu().numpy()
        if nxt == spp:
            assert np.array_equal(bits(got), bits(full))
            ref_img, ref_u8 = ref.untile()
        else:
            ref_img, ref_u8 = ref.untile(preview_samples=nxt)
            state = ref.tile_buf.cpu().numpy()  # [tile, y, x, 4] running state of the one-device render
            torch.cuda.synchronize()
            t0 = ref.tiles[0]
            s = state[0, :t0.height(), :t0.width()]
            if chunked:
                tot = np.ascontiguousarray(s[..., 2:4]).view(np.float64)[..., 0] + s[..., 0].astype(np.float64)
                exp = (tot * (1.0 / np.float64(nxt))).astype(np.float32)
                exp_a = (s[..., 1].astype(np.float64) * (1.0 / np.float64(nxt))).astype(np.float32)
            else:
                inv = np.float32(1.0) / np.float32(nxt)
                exp, exp_a = s[..., 0] * inv, s[..., 3] * inv
            blk = got[t0.min_y:t0.max_y, t0.min_x:t0.max_x]
            assert np.array_equal(bits(blk[..., 0]), bits(exp)) and np.array_equal(bits(blk[..., 3]), bits(exp_a))
            assert np.array_equal(bits(blk[..., 1]), bits(exp)) and np.array_equal(bits(blk[..., 2]), bits(exp))
        torch.cuda.synchronize()
        assert np.array_equal(bits(got), bits(ref_img.cpu().numpy())) and np.array_equal(img8.cpu().numpy(), ref_u8.cpu().numpy())
    assert nxt == spp
    # a pass that does not continue the shards' state is refused; a pass from 0 starts over
    with pytest.raises(mp.MinipathError):
        mf.render_pass(7, 3)
    assert mf.render_pass(0, 4, gather=False) == 4
    with pytest.raises(mp.MinipathError):
        mf.render_pass(5, 3)
    with pytest.raises(mp.MinipathError):
        mp.MultiDeviceFrame(list(reversed(scenes)), cam, st).render_pass(4, 3)   # other ranks: not the state those shards hold
    assert mf.render_pass(4, 0)[0] == spp
    torch.cuda.synchronize()
    assert np.array_equal(bits(mf.image.cpu().numpy()), bits(full))
    # whole frames through the same buffers afterwards
    img, _ = mf.render()
    torch.cuda.synchronize()
    assert np.array_equal(bits(img.cpu().numpy()), bits(full))


def test_group_outlives_its_members_handles(ctx):
    """an object group borrows its members' device arrays; it now holds a reference on each member, so closing the
    members first must leave the group renderable (before: kernels read freed device memory)."""
    import torch

    base = mp.TriangleBvh.with_obj(TEAPOT, ctx)
    tr = np.array([[0, 0, 0], [7.5, 0, -3]], np.float32)
    inst = mp.Instances(base, tr)
    cam = mp.Camera.teapot_view()
    st = mp.RenderSettings(32, 4, (160, 96), seed=SEED)
    fr = mp.FrameRenderer(mp.Scene(inst), cam, st)
    fr.render()
    before, _ = fr.untile()
    torch.cuda.synchronize()
    before = before.cpu().numpy()
    base.close()                       # the caller's handle on the member goes first
    junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(16)]  # churn the allocator over whatever was freed
    del junk
    fr2 = mp.FrameRenderer(mp.Scene(inst), cam, st)
    fr2.render()
    after, _ = fr2.untile()
    torch.cuda.synchronize()
    assert np.array_equal(bits(after.cpu().numpy()), bits(before))
    assert inst.info().triangle_count == 2256  # info() reads the member's host tree: still there
    inst.close()
