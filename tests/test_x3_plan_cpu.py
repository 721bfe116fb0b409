"""Host logic of the split-bf16 synthesis plan (lib/generators/synthesis_pack.py: build_x3 / x3_forward_tables) checked on
the CPU: the packed weight stream (consumption order, accumulator-order K permutation, hi + lo halves), the folded conv
biases / ToRGB biases and the per-forward tables are decoded and run through a plain float64 restatement of what
csrc/synthesis_x3.hip computes, and the image must match the oracle's SynthesisNetwork.  No HIP call is made."""
import importlib

import pytest
import torch

import h3d_oracle as O
from conftest import load_golden, rel_err

gens = importlib.import_module("3dhumangan_amd.lib.generators")
sp = importlib.import_module("3dhumangan_amd.lib.generators.synthesis_pack")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")


def acc_k(ks, h, e):
    return 32 * (ks // 2) + (e & 3) + 8 * (2 * (ks & 1) + (e >> 2)) + 4 * h


def decode_matrix(stream_i16, stage0, KS, NT):
    """stages [KS][NT][2][64][8] of bf16 bit patterns -> dense W [32*NT, 16*KS] (hi + lo), K back in natural order."""
    n = KS * NT * 2 * 64 * 8
    t = stream_i16[stage0 * NT * 2 * 64 * 8: stage0 * NT * 2 * 64 * 8 + n].view(torch.bfloat16).double().view(KS, NT, 2, 64, 8)
    t = t[:, :, 0] + t[:, :, 1]                                       # [KS, NT, 64, 8]
    W = torch.zeros(32 * NT, 16 * KS, dtype=torch.float64)
    for ks in range(KS):
        for h in range(2):
            for e in range(8):
                W[:, acc_k(ks, h, e)] = t[ks, :, 32 * h: 32 * h + 32, e].reshape(-1)
    return W


CODES = torch.tensor([0, .125, .25, .375, .5, .625, .75, .875, 1, 1.125, 1.25, 1.375, 1.5, 1.625, 1.75, 1.875,
                      2, 2.25, 2.5, 2.75, 3, 3.25, 3.5, 3.75, 4, 4.5, 5, 5.5, 6, 6.5, 7, 7.5], dtype=torch.float64)


def decode_x2(stream_i16, stage0, KS, NT, order=None, dense=True):
    """x2 stages [KS][NT][1 KiB f16 hi fragment | 1 KiB record halves] -> (Whi [32 NT, 16 KS] in feature order, groups for
    x2_emulation.x2_operands_matmul: (features [16], codes_hi [N, 16], codes_lo [N, 16], block scale [N])).
    order(ks, lane half, element) -> feature (default: accumulator-register order)."""
    order = order or acc_k
    n = KS * NT * 1024                                                # int16 elements
    st = stream_i16[stage0 * NT * 1024: stage0 * NT * 1024 + n].view(KS, NT, 2, 512)
    hi = st[:, :, 0].contiguous().view(torch.float16).double().view(KS, NT, 64, 8)
    Whi = torch.zeros(32 * NT, 16 * KS, dtype=torch.float64)
    for ks in range(KS):
        for h in range(2):
            for e in range(8):
                Whi[:, order(ks, h, e)] = hi[ks, :, 32 * h: 32 * h + 32, e].reshape(-1)
    half = st[:, :, 1].contiguous().view(torch.uint8).view(KS, NT, 1024).to(torch.int64)
    groups = []
    for T in range(KS // 2):
        # even stage: code dwords 0-3 at 16 B per lane; odd stage DENSE: code dwords 4-5 as [64 lanes][8 B], scale dwords as
        # [64 lanes][4 B], 256 B of zeros (pack_stream_x2(dense=True): conflict-free 64- / 32-bit LDS reads)
        ev, od = half[2 * T].view(NT, 64, 16), half[2 * T + 1]
        if dense:
            assert not od[:, 768:].any()
            sd = od[:, 512:768].view(NT, 64, 4)
            rec = torch.cat([ev, od[:, :512].view(NT, 64, 8), sd, sd], dim=-1)      # [NT, 64, 32 bytes] (scale dword twice)
        else:                                                                       # the LDS-resident engine: 16 B per lane in both stages
            rec = torch.cat([ev, od.view(NT, 64, 16)], dim=-1)
        b = rec[..., :24].reshape(NT, 64, 8, 3)
        c = torch.stack([b[..., 0] & 63, (b[..., 0] >> 6) | ((b[..., 1] & 15) << 2), (b[..., 1] >> 4) | ((b[..., 2] & 3) << 4),
                         b[..., 2] >> 2], dim=-1).reshape(NT, 64, 32)
        vals = CODES[c & 31] * torch.where((c & 32) > 0, -1.0, 1.0)
        assert torch.equal(rec[..., 24:32], rec[..., 24:25].expand(-1, -1, 8))          # the scale byte fills dwords 6 and 7
        scale = torch.exp2((rec[..., 24] - 127).double())
        for h in range(2):
            feats = torch.tensor([order(2 * T + j, h, e) for j in range(2) for e in range(8)])
            v = vals[:, 32 * h: 32 * h + 32].reshape(32 * NT, 32)
            groups.append((feats, v[:, :16], v[:, 16:], scale[:, 32 * h: 32 * h + 32].reshape(32 * NT)))
    return Whi, groups


def emulate(plan, fmap_lowres, fixed_style, Hr, Wr, H, W, x2=False):
    """float64 restatement of synthesis_x3_kernel on the plan's own tables / stream (x2: the x2 arithmetic on the decoded f16
    fragments and fp6 records, with the per-pixel activation scales of csrc/synthesis_x3.hip)."""
    x3 = plan.build_x3(x2)
    assert len(x3["segments"]) == 1
    seg, NT, HdP = x3["segments"][0], x3["NT"], x3["HdP"]
    desc, tab = seg["desc"], seg["tables"].double()
    stream = seg["stream"]
    G, cst, ab = plan.x3_forward_tables(fmap_lowres.float(), fixed_style.float(), x2)

    def decode(stream_, stage0, KS, NT_):
        return decode_x2(stream_, stage0, KS, NT_) if x2 else decode_matrix(stream_, stage0, KS, NT_)

    def mm(y, ops):                                                   # y @ W.t() in the engine's arithmetic
        if not x2:
            return y @ ops.t()
        from x2_emulation import x2_operands_matmul
        yp = torch.nn.functional.pad(y, (0, ops[0].shape[1] - y.shape[-1]))
        return x2_operands_matmul(yp, ops[0], ops[1], dynamic=True)
    B = fixed_style.shape[0]
    vec = lambda off, n=HdP: tab[off: off + n]
    ii = torch.linspace(-1, 1, H, dtype=torch.float64).view(H, 1).expand(H, W).reshape(-1)
    jj = torch.linspace(-1, 1, W, dtype=torch.float64).view(1, W).expand(H, W).reshape(-1)
    x = torch.sin(ii[:, None] * vec(desc.w_in) + jj[:, None] * vec(desc.w_in + HdP) + vec(desc.b_in))   # [HW, HdP]
    x = x.unsqueeze(0).repeat(B, 1, 1)
    if G is not None:
        Gmap = G.double().view(B, Hr, Wr, -1).permute(0, 3, 1, 2)
        Gup = torch.nn.functional.interpolate(Gmap, (H, W), mode="bilinear").permute(0, 2, 3, 1).reshape(B, H * W, -1)
    rgb = torch.zeros(B, H * W, 3, dtype=torch.float64)
    stage, heads = 0, 0
    mid_x3 = [0]
    lrelu = lambda v: torch.maximum(v, 0.2 * v)
    for k in range(desc.n_blocks):
        bk = desc.block[k]
        x_in = x
        for s in range(2):
            d = bk.spade[s]
            if d.pixel_style:
                a = torch.relu(Gup[:, :, d.g_offset: d.g_offset + 128] + cst[:, d.cst_index].double()[:, None, :])
                Wg = decode(stream, stage, 8, NT); stage += 8
                Wb = decode(stream, stage, 8, NT); stage += 8
                g1 = vec(d.vec) + mm(a, Wg)
                y = lrelu((x * vec(d.vec + 2 * HdP) + vec(d.vec + 3 * HdP)) * g1 + vec(d.vec + HdP) + mm(a, Wb))
            else:
                t4 = ab[:, d.ab_index].double()                       # [B, HdP/2, 2 (sc | sh), 2 (channel pair)]
                sc = t4[:, :, 0, :].reshape(B, 1, HdP)
                sh = t4[:, :, 1, :].reshape(B, 1, HdP)
                u = x * sc + sh                                      # the tables carry 0.4 * (sc, sh): lrelu(t) = 1.5 u + |u|
                y = 1.5 * u + u.abs()
            if x2 and not d.pixel_style and d.g_offset == 1:
                # round 6 (MIDX3): this convolution travels in the x3 format and runs on three bf16 products inside the x2 kernel
                assert not bk.skip
                Wc = decode_matrix(stream, stage, 2 * NT, NT); stage += 2 * NT
                mid_x3[0] += 1
                x = y @ Wc.t()
                continue
            Wc = decode(stream, stage, 2 * NT, NT); stage += 2 * NT
            x = mm(y, Wc) + (x_in if (s == 1 and bk.skip) else 0.0)
            if x2 and s == 1 and d.b_conv >= 0:
                # ToRGB head table of a skip block (round 5): the 8 lanes (rows 0-3, both halves) of a one-tile x2 stream of
                # M_j = V_j W1_j -- scattered back into a full tile and decoded like any other x2 matrix; rows 0-2 are r, g, b
                heads += 1
                t8 = seg["tables"][d.b_conv: d.b_conv + 2 * NT * 64].contiguous().view(torch.int16).view(2 * NT, 2, 8, 8)
                full = torch.zeros(2 * NT, 1, 2, 64, 8, dtype=torch.int16)
                full[:, 0][:, :, [0, 1, 2, 3, 32, 33, 34, 35]] = t8
                rgb = rgb + mm(y, decode_x2(full.flatten(), 0, 2 * NT, 1, dense=False))[..., :3]
        if bk.to_rgb:
            wr = torch.stack([vec(bk.w_rgb), vec(bk.w_rgb + HdP), vec(bk.w_rgb + 2 * HdP)])          # [3, HdP]
            rgb = rgb + x @ wr.t() + vec(bk.w_rgb + 3 * HdP, 3)
    assert stage == seg["stages"]
    if x2:      # the constant-style blocks in front of the first skip block (one in the shipped layouts): both convolutions in the x3 format
        n_mid = sum(2 for k in range(desc.n_blocks) if not desc.block[k].skip and not desc.block[k].spade[0].pixel_style
                    and not any(desc.block[q].skip for q in range(k)))
        assert mid_x3[0] == (n_mid if plan.X2_MID_X3 else 0)
    if x2 and plan.X2_HEADS and any(desc.block[k].skip for k in range(desc.n_blocks)):
        assert heads == sum(1 for k in range(desc.n_blocks) if desc.block[k].skip)      # every skip block of an x2 plan carries one
        assert not any(desc.block[k].to_rgb for k in range(desc.n_blocks) if desc.block[k].skip)
    return rgb.view(B, H, W, 3).permute(0, 3, 1, 2)


@pytest.mark.parametrize("x2", [False, True])
@pytest.mark.parametrize("mode,width", [("mixed", 32), ("isolated", 40)])
def test_x3_plan_matches_oracle(mode, width, x2):
    meta = dict(load_golden("gen_tiny_mixed")["meta"])
    meta.update(map3d_mode=mode, hidden_dim=width, latent_dim=width, feature_dim=width, gen_height=12, gen_width=8,
                render_height=5, render_width=4)
    meta["neural_field_cls"] = impl.COORDCONCATSIREN
    torch.manual_seed(3 + width)
    Gn = gens.Map3DGenerator(**meta).eval()
    with torch.no_grad():                                             # non-trivial BN statistics, biases, spectral u/v
        for n, p in Gn.named_parameters():
            if n.endswith("bias"):
                p.add_(0.1 * torch.randn_like(p))
        for n, b in Gn.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(0.2 * torch.randn_like(b))
            if n.endswith("running_var"):
                b.copy_(0.5 + torch.rand_like(b))
    sd = {k: v.detach().clone() for k, v in Gn.state_dict().items()}
    plan = sp.SynthesisPlan(sd, "synthesis_network", "synthesis_input", meta["synthesis_blocks"], tuple(meta["mod_blocks"]), mode,
                            torch.device("cpu"))
    assert plan.x3_supported()
    B, Hr, Wr, H, W = 2, 5, 4, 12, 8
    fmap = torch.randn(B, Hr * Wr, width)
    style = torch.randn(B, width)
    got = emulate(plan, fmap, style, Hr, Wr, H, W, x2)
    fm = fmap.view(B, Hr, Wr, width).permute(0, 3, 1, 2)
    fm_up = torch.nn.functional.interpolate(fm, (H, W), mode="bilinear")
    x0 = O.synthesis_input(sd, B, H, W)
    ref = O.synthesis_network(sd, x0, fm_up, style.view(B, 1, width), mode, tuple(meta["mod_blocks"]), meta["synthesis_blocks"])["final"]
    # bf16 hi + lo carries 16 significant bits of every weight: 1e-4 covers it comfortably
    assert rel_err(got.float(), ref) < 1e-4


def test_planner_uses_the_kernels_own_lds_count():
    """x3_supported / x2_supported ask the library how much LDS the kernel needs (tables + per-sample tables + descriptor copy +
    weight ring) instead of restating the formula: the x2 kernel needs exactly one more ring buffer, and a table set that leaves
    no room for it makes the plan fall back to the x3 engine, not fail at launch."""
    lib = importlib.import_module("3dhumangan_amd._lib").load()
    base = lib.h3d_synthesis_x3_lds_bytes(11544, 12, 6, 256, 0)
    assert lib.h3d_synthesis_x3_lds_bytes(11544, 12, 6, 256, 1) - base == lib.h3d_synthesis_x2_extra_lds(256) == 8 * 2048
    assert lib.h3d_synthesis_x3_lds_bytes(11544 + 4, 12, 6, 256, 0) - base == 16          # 4 more table floats
    assert lib.h3d_synthesis_x3_lds_bytes(11544, 13, 6, 256, 0) - base == 2 * 256 * 4     # one more constant-style SPADE
    assert lib.h3d_synthesis_x3_lds_bytes(11544, 12, 6, 256, 1) <= 160 * 1024             # MAP3DBN512 (the bench workload) fits x2
    assert lib.h3d_synthesis_x3_lds_bytes(11544, 12, 6, 256, 1) - lib.h3d_synthesis_x3_lds_bytes(11544, 12, 6, 256, 3) == 3 * 256 * 4   # heads: no zero table
    assert lib.h3d_synthesis_x3_lds_bytes(12804, 12, 6, 256, 3) <= 160 * 1024 < lib.h3d_synthesis_x3_lds_bytes(12804, 12, 6, 256, 1)    # ... which is what makes the head tables fit
    assert lib.h3d_synthesis_x3_lds_bytes(11544, 16, 6, 256, 1) > 160 * 1024 >= lib.h3d_synthesis_x3_lds_bytes(11544, 16, 6, 256, 0)


def test_head_tables_that_do_not_fit_the_lds_leave_the_plan_on_x2_with_the_riding_torgb():
    """MAP3DBN512's widths with a tenth block: 14 constant-style SPADEs leave 112 B of the 160 KB after the x2 ring -- the ToRGB
    head tables (4 KB per skip block for 3 KB of zero table) do not fit, and the plan keeps the x2 engine with the per-block ToRGB
    instead of failing at launch or dropping to the x3 engine.  With 9 blocks (the shipped config) the heads fit."""
    lib = importlib.import_module("3dhumangan_amd._lib").load()
    for n_blocks, want_heads in ((9, True), (10, False)):
        meta = dict(load_golden("gen_tiny_mixed")["meta"])
        meta.update(map3d_mode="mixed", hidden_dim=256, latent_dim=256, feature_dim=256, gen_height=32, gen_width=32, render_height=8,
                    render_width=8, synthesis_blocks=n_blocks, mod_blocks=[0, 1, 2])
        meta["neural_field_cls"] = impl.COORDCONCATSIREN
        torch.manual_seed(0)
        sd = {k: v.detach().clone() for k, v in gens.Map3DGenerator(**meta).eval().state_dict().items()}
        plan = sp.SynthesisPlan(sd, "synthesis_network", "synthesis_input", n_blocks, (0, 1, 2), "mixed", torch.device("cpu"))
        seg = plan.build_x3(True)["segments"][0]
        heads = any(seg["desc"].block[j].spade[1].b_conv >= 0 for j in range(seg["desc"].n_blocks))
        assert heads == want_heads and plan.X2_HEADS == want_heads
        assert plan._x3_fits(True)
        assert lib.h3d_synthesis_x3_lds_bytes(seg["tables"].numel(), len(plan.const_ids), len(plan.pixel_ids), 256, 3 if heads else 1) <= 160 * 1024
        if not heads:          # every block that feeds ToRGB kept its own table
            assert all(seg["desc"].block[j].w_rgb >= 0 for j in range(seg["desc"].n_blocks) if seg["desc"].block[j].to_rgb)
            assert sum(int(seg["desc"].block[j].to_rgb) for j in range(seg["desc"].n_blocks)) > 0
