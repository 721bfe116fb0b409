"""The x2 synthesis engine's sampled error monitor (round 5): every guarded x2 forward re-evaluates ~32 of its 128-pixel tiles per
image on the bf16 x3 engine (h3d_synthesis_x3_tiles), compares (h3d_synthesis_check) and raises the ITEM's device flag (round 6:
one flag per batch item) when a sampled pixel leaves the tolerance -- the x3 engine behind it then redoes that item, and only
that item.  The tolerance is the 6e-4 an image may carry divided by the measured sampling factor 1.7 (SynthesisPlan).  Reference
semantics: lib/components/map3d_layers.py:176-238 (fp32 convolutions, nothing to monitor)."""
import ctypes
import importlib

import pytest
import torch

from conftest import rel_err
from test_gpu_x2_guard import make, oracle_rgb, run

pytestmark = pytest.mark.gpu
_lib = importlib.import_module("3dhumangan_amd._lib")
DEV = "cuda"


def test_monitor_is_quiet_on_the_default_arithmetic_and_reports_the_sampled_error():
    G, meta, sd = make(256, 128, 128, 24, 24, seed=1)
    plan = G.synthesis_plan(DEV)
    assert plan.engine == "f16x2" and plan.x2_guard and plan.x2_monitor
    assert abs(plan.x2_monitor_tol - plan.X2_MONITOR_BUDGET / plan.X2_SAMPLING_FACTOR) < 1e-12 and 3e-4 < plan.x2_monitor_tol < 4e-4
    plan.x2_monitor_tol = 1e-3                    # this test wants the x2 image itself whatever this random network's error is
    B = 3
    fmap, style = torch.randn(B, 576, 256), torch.randn(B, 256)
    out = run(G, meta, fmap, style)
    assert not plan.x2_fell_back()
    err = plan.x2_monitor_errors().cpu()
    assert err.shape == (B,) and float(err.max()) < 1e-3 and float(err.min()) > 1e-6      # a real, non-trivial measurement
    # the same numbers from the two full images
    plan.engine = "bf16x3"
    ref = run(G, meta, fmap, style)
    plan.engine = "f16x2"
    first, step = plan.monitor_tiles(128, 128)
    tiles = torch.arange(first, 128 * 128 // 128, step)
    assert 24 <= len(tiles) <= 40 and step % 2 == 1 and len(tiles) == plan.monitor_tile_count(128, 128)
    px = (tiles[:, None] * 128 + torch.arange(128)[None, :]).flatten().to(DEV)
    a, b = out.flatten(2)[:, :, px], ref.flatten(2)[:, :, px]
    want = ((a - b).abs().amax(2) / out.flatten(2).abs().amax(2)).amax(1).cpu()        # relative to the WHOLE image's channel maximum
    assert torch.allclose(err, want, rtol=1e-5, atol=0)
    # switching the monitor off changes nothing in the image
    plan.x2_monitor = False
    assert torch.equal(run(G, meta, fmap, style), out)


def test_sampled_pixels_outside_the_tolerance_redo_every_item_on_x3():
    G, meta, sd = make(256, 128, 128, 24, 24, seed=2)
    plan = G.synthesis_plan(DEV)
    fmap, style = torch.randn(2, 576, 256), torch.randn(2, 256)
    plan.x2_monitor_tol = 1e-3
    x2 = run(G, meta, fmap, style)
    assert not plan.x2_fell_back()
    plan.engine = "bf16x3"
    x3 = run(G, meta, fmap, style)
    plan.engine = "f16x2"
    assert not torch.equal(x2, x3)
    plan.x2_monitor_tol = 1e-7                    # tighter than the x2 arithmetic can be: every forward must fall back
    out = run(G, meta, fmap, style)
    assert plan.x2_fell_back() and plan.x2_fallback_items() == [0, 1]
    assert torch.equal(out, x3)                   # the image IS the x3 engine's
    assert rel_err(out.cpu(), oracle_rgb(sd, meta, fmap, style)) < 1e-4


def test_only_the_items_over_the_tolerance_are_redone():
    """Round 6: one flag per batch item.  With the tolerance set between the items' sampled errors exactly the items above it come
    from the x3 engine (bit-identical to the x3 image), the others keep their x2 pixels (bit-identical to the x2 image): an image
    never depends on its batch mates."""
    G, meta, sd = make(256, 128, 128, 24, 24, seed=5)
    plan = G.synthesis_plan(DEV)
    B = 6
    fmap, style = torch.randn(B, 576, 256), torch.randn(B, 256)
    plan.x2_monitor_tol = 1.0
    x2 = run(G, meta, fmap, style)
    assert plan.x2_fallback_items() == []
    err = plan.x2_monitor_errors().cpu()
    plan.engine = "bf16x3"
    x3 = run(G, meta, fmap, style)
    plan.engine = "f16x2"
    srt = torch.sort(err).values
    plan.x2_monitor_tol = float((srt[B // 2 - 1] + srt[B // 2]) / 2)          # half of the items above, half below
    want = [b for b in range(B) if float(err[b]) > plan.x2_monitor_tol]
    assert 0 < len(want) < B
    out = run(G, meta, fmap, style)
    assert plan.x2_fallback_items() == want and plan.x2_fell_back()
    for b in range(B):
        assert torch.equal(out[b], x3[b] if b in want else x2[b]), b
    assert not torch.equal(x2[want[0]], x3[want[0]])
    # the same item alone: the same decision, the same pixels
    b = want[0]
    one = run(G, meta, fmap[b:b + 1], style[b:b + 1])
    assert plan.x2_fallback_items() == [0] and torch.equal(one[0], x3[b])


def test_tile_subset_launch_writes_the_sampled_tiles_only():
    G, meta, sd = make(128, 64, 128, 12, 24, seed=3)
    plan = G.synthesis_plan(DEV)
    B, H, W, Hr, Wr = 2, 64, 128, 12, 24
    fmap, style = torch.randn(B, Hr * Wr, 128).to(DEV), torch.randn(B, 128).to(DEV)
    plan.engine = "bf16x3"
    full = plan.run(fmap, style, (Hr, Wr), (H, W))
    seg = plan.build_x3(False)["segments"][0]
    Gt, cst, ab = plan.x3_forward_tables(fmap.float(), style.float(), False)
    scratch = torch.full_like(full, -7.0)
    first, step = 3, 5
    rc = _lib.load().h3d_synthesis_x3_tiles(_lib.ptr(seg["stream"]), seg["stages"], _lib.ptr(seg["tables"]), seg["tables"].numel(),
                                            ctypes.byref(seg["desc"]), _lib.ptr(Gt), plan.g_channels, Hr, Wr, _lib.ptr(cst),
                                            len(plan.pixel_ids), _lib.ptr(ab), len(plan.const_ids), _lib.ptr(scratch), B, H, W,
                                            first, step, _lib.stream_handle())
    _lib.check(rc, "h3d_synthesis_x3_tiles")
    n_tiles = H * W // 128
    sampled = torch.zeros(n_tiles, dtype=torch.bool)
    sampled[first::step] = True
    mask = sampled[:, None].expand(n_tiles, 128).reshape(H * W).to(DEV)
    s, f = scratch.flatten(2), full.flatten(2)
    assert torch.equal(s[:, :, mask], f[:, :, mask])                     # same kernel, same arithmetic: bit-identical pixels
    assert bool((s[:, :, ~mask] == -7.0).all())                          # nothing else touched
    # the check kernel: identical images -> 0, no flag; one perturbed sampled pixel -> that error, flag; non-finite -> flag
    flag = torch.zeros(B, dtype=torch.int32, device=DEV)             # one flag per item
    err, work = torch.zeros(B, device=DEV), torch.full((6 * B,), 9.0, device=DEV)       # the call zeroes its scratch itself
    chk = lambda img, tol: _lib.check(_lib.load().h3d_synthesis_check(_lib.ptr(img), _lib.ptr(scratch), B, H, W, first, step, tol,
                                                                      _lib.ptr(flag), _lib.ptr(err), _lib.ptr(work),
                                                                      _lib.stream_handle()), "check")
    chk(full, 1e-3)
    assert flag.tolist() == [0, 0] and float(err.max()) == 0.0
    bad = full.clone()
    p = first * 128 + 17
    bad.flatten(2)[1, 2, p] += 0.5
    chk(bad, 1e-3)
    want = 0.5 / float(bad.flatten(2)[1, 2].abs().max())
    assert flag.tolist() == [0, 1] and abs(float(err[1]) - want) < 1e-5 * want and float(err[0]) == 0.0
    flag.zero_()
    bad.flatten(2)[1, 2, p + 128] += 0.5                                 # a pixel OUTSIDE the sample: not seen (that is what sampling means)
    bad.flatten(2)[1, 2, p] = full.flatten(2)[1, 2, p]
    chk(bad, 1e-3)
    assert flag.tolist() == [0, 0]
    bad.flatten(2)[0, 0, p] = float("nan")
    chk(bad, 1e-3)
    assert flag.tolist() == [1, 0] and not torch.isfinite(err[0])
    # a tile step beyond the tile count samples the first tile alone (and does not overflow the tile index)
    flag.zero_()
    _lib.check(_lib.load().h3d_synthesis_check(_lib.ptr(full), _lib.ptr(scratch), B, H, W, first, 2 ** 31 - 1, 1e-3, _lib.ptr(flag),
                                               _lib.ptr(err), _lib.ptr(work), _lib.stream_handle()), "check")
    assert flag.tolist() == [0, 0] and float(err.max()) == 0.0


@pytest.mark.parametrize("width", [128, 256])
def test_torgb_head_tiles_give_the_image_of_the_riding_torgb(width):
    """x2 plans fold the ToRGB of every skip block into a ninth output tile of that block's second convolution
    (SynthesisPlan._torgb_heads: sum_k Wr_k x_k = (sum_k Wr_k) x_f + sum_j (V_j W1_j) y_j).  The plan with heads and the plan
    without (per-block ToRGB riding in the next producer) are the same function: equal within the x2 arithmetic's own noise, and
    both inside the budget of the oracle (lib/generators/map3d_generator.py:82-86, map3d_layers.py:346-352)."""
    G, meta, sd = make(width, 128, 128, 24, 24, seed=4)
    plan = G.synthesis_plan(DEV)
    plan.x2_monitor_tol = 1e-3                    # the x2 images themselves are compared
    assert plan.engine == "f16x2" and plan.X2_HEADS
    fmap, style = torch.randn(2, 576, width), torch.randn(2, width)
    want = oracle_rgb(sd, meta, fmap, style)
    with_heads = run(G, meta, fmap, style)
    desc = plan.build_x3(True)["segments"][0]["desc"]
    assert any(desc.block[k].spade[1].b_conv >= 0 for k in range(desc.n_blocks))          # the head tables are in the plan
    assert not plan.x2_fell_back()
    plan.X2_HEADS = False
    plan._x2 = None                                                                       # rebuild without heads
    try:
        without = run(G, meta, fmap, style)
        desc = plan.build_x3(True)["segments"][0]["desc"]
        assert all(desc.block[k].spade[1].b_conv < 0 for k in range(desc.n_blocks))
        assert not plan.x2_fell_back()
    finally:
        plan.X2_HEADS = True
        plan._x2 = None
    den = want.abs().amax(dim=(0, 2, 3), keepdim=True)
    e = lambda a: float(((a.cpu() - want).abs() / den).max())
    d = float(((with_heads - without).abs().cpu() / den).max())
    print(f"width {width}: heads {e(with_heads):.2e}, riding {e(without):.2e}, heads vs riding {d:.2e}")
    assert e(with_heads) < 1e-3 and e(without) < 1e-3 and 0 < d < 6e-4
