"""Stress test of the LDS weight ring of the register-resident x3 engines (csrc/x3_common.hpp: WeightRing::acquire).

The ring's write-after-read hazard (a refill landing in a buffer whose last fragment reads are still in flight) is closed by
construction since round 6: every wave waits for its LDS reads (lgkmcnt(0)) before the stage barrier behind which the buffer is
refilled (rounds 2-5: by a distance argument, which failed in conv_x3.hip once several workgroups shared a CU --
tests/test_gpu_conv.py::test_runs_are_bit_identical_with_several_workgroups_per_cu).  A violation would show up as a corrupted
weight fragment in some workgroup, i.e. as a different image.  Here the two kernels run 200 times back to back on the bench
geometry (small batch), alone and next to a second stream that thrashes HBM / L2 (which stretches LDS-DMA latencies and perturbs
the relative timing of the waves), and every output must be BIT-identical to the first one; the result is also checked against the
strict fp32-MFMA engines, which have no ring."""
import importlib

import pytest
import torch


pytestmark = pytest.mark.gpu
gens = importlib.import_module("3dhumangan_amd.lib.generators")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
synthetic = importlib.import_module("3dhumangan_amd.synthetic")
configs = importlib.import_module("3dhumangan_amd.configs")
DEV = "cuda"


def _setup(B=2):
    cfg = {k: v for k, v in configs.MAP3DBN512.items() if isinstance(k, str)}
    cfg.update(gen_height=512, gen_width=512, render_height=96, render_width=96, num_steps=64, dataset_length=4, nerf_noise=0,
               last_back=True)
    cfg["neural_field_cls"] = impl.COORDCONCATSIREN
    torch.manual_seed(77)
    G = gens.Map3DGenerator(**cfg).to(DEV).eval()
    G.set_device(DEV)
    g = torch.Generator().manual_seed(77)
    cond = {k: v.to(DEV) for k, v in synthetic.make_conditions(B, 6890, seed=77).items()}
    z = torch.randn(B, cfg["latent_dim"], generator=g).to(DEV)
    jit = torch.rand(B, 96 * 96, 64, 1, generator=g).to(DEV)
    return G, cfg, z, cond, jit


@pytest.mark.parametrize("engines", [("f16x2", "f16x2"), ("f16x3", "bf16x3")])
@pytest.mark.parametrize("thrash", [False, True])
def test_x3_kernels_are_bit_reproducible_over_200_launches(thrash, engines):
    G, cfg, z, cond, jit = _setup()
    assert G.neural_field.precision == "f16x2" and G.synthesis_plan(DEV).engine == "f16x2"
    G.neural_field.precision, G.synthesis_plan(DEV).engine = engines
    first = G.forward(z, cond, jitter=jit, **cfg)
    ref_rgb, ref_ren = first["rgbs"].clone(), first["rgbs_render"].clone()
    side = torch.cuda.Stream()
    junk = torch.empty(2, 1 << 28, dtype=torch.uint8, device=DEV) if thrash else None       # 2 x 256 MB: beyond the caches
    bad = 0
    for it in range(200):
        if thrash:
            with torch.cuda.stream(side):
                for _ in range(4):
                    junk[1].copy_(junk[0], non_blocking=True)
                    junk[0].add_(1)
        out = G.forward(z, cond, jitter=jit, **cfg)
        bad += int(not torch.equal(out["rgbs"], ref_rgb)) + int(not torch.equal(out["rgbs_render"], ref_ren))
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of 400 outputs differ from the first launch"
    # and the ring-less fp32-MFMA engines agree with it to rounding -- except on rays whose LAST sample has a density within
    # rounding of zero: the reference gives that sample delta = 1e9 (lib/generators/volume_rendering.py:21), so its alpha is 0 or 1 by the SIGN
    # of the density, and with white_back the background term 1 - sum(w) flips between "all of the remaining transmittance" and
    # 0 -- the same shift in every channel.  With this random-initialised field (densities centred on zero) that happens on about
    # one ray in 20 000 (round 4: 1 of 18 432 once the fused render builds its own geometry features, which differ from
    # h3d_geo_features' in the last bit).  Such rays are identified by that signature, counted, and left out of the tolerance.
    G.neural_field.precision = "f32"
    G.synthesis_plan(DEV).engine = "f32"
    strict = G.forward(z, cond, jitter=jit, **cfg)
    d = (ref_ren - strict["rgbs_render"]).cpu()                        # [B, 3, Hr, Wr]
    flipped = d.abs().amax(1) > 1e-2
    assert float(flipped.float().mean()) < 5e-4, f"{int(flipped.sum())} rays differ by more than 1e-2"
    if flipped.any():
        spread = (d.amax(1) - d.amin(1))[flipped]
        assert float(spread.max()) < 1e-3, "a ray differs by more than the background term of the last-sample discontinuity"
    keep = ~flipped
    ren_err = float((d.abs().amax(1) * keep).max() / strict["rgbs_render"].abs().max())
    assert ren_err < 2e-4, ren_err
    # the image: leave out the footprint of those rays (the feature map is resized bilinearly: a flipped ray reaches 2 rays far)
    far = torch.nn.functional.max_pool2d(flipped.float()[:, None], 5, 1, 2)
    keep_px = torch.nn.functional.interpolate(1.0 - far, size=ref_rgb.shape[-2:], mode="nearest")
    img_err = float(((ref_rgb - strict["rgbs"]).cpu().abs() * keep_px).max() / strict["rgbs"].abs().max())
    assert img_err < (2e-4 if engines[1] == "bf16x3" else 1e-3), img_err
