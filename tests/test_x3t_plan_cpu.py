"""Host logic of the LDS-resident split-bf16 synthesis plan (lib/generators/synthesis_pack.py: build_x3t) checked on the
CPU: the tile-major fragment blob (accumulator-order K for the convs, natural order for gamma / beta, hi + lo halves), the
fp32 tables and the per-forward tables are decoded and run through a plain float64 restatement of what
csrc/synthesis_x3t.hip computes; the image must match the oracle's SynthesisNetwork.  Widths 384 / 420 are the ones the
engine exists for (here at a tiny image).  No kernel launch (only the HOST helper h3d_synthesis_x3t_tiles)."""
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


def decode_matrix(wblob_i16, byte_off, KS, NT, acc_order):
    """A fragments [NT][KS][2][64][8] of bf16 bit patterns -> dense W [32*NT, 16*KS] (hi + lo), K back in natural order."""
    n = NT * KS * 2 * 64 * 8
    assert byte_off % 16 == 0
    t = wblob_i16[byte_off // 2: byte_off // 2 + n].view(torch.bfloat16).double().view(NT, KS, 2, 64, 8)
    t = t[:, :, 0] + t[:, :, 1]                                       # [NT, KS, 64, 8]
    W = torch.zeros(32 * NT, 16 * KS, dtype=torch.float64)
    for ks in range(KS):
        for h in range(2):
            for e in range(8):
                k = acc_k(ks, h, e) if acc_order else 16 * ks + 8 * h + e
                W[:, k] = t[:, ks, 32 * h: 32 * h + 32, e].reshape(-1)
    return W


def decode_x2(wblob_i16, byte_off, KS, NT, acc_order):
    """x2c blob (round 6) [NT][K-tile][hi fragment 2T | lo record | hi fragment 2T + 1] -> (Whi, groups) of
    tests/x2_emulation.x2_operands_matmul.  The hi codes are not stored: the kernel converts them from the hi fragments with the
    lane's block scale (v_cvt_scalef32_pk32_fp6_f16) -- restated here with the packer's own e2m3 rounding -- and the stream decoder
    of the register engine's tests takes the re-assembled full records."""
    from test_x3_plan_cpu import decode_x2 as decode_stream
    T = KS // 2
    n = NT * T * 3 * 512                                                           # int16 elements
    t = wblob_i16[byte_off // 2: byte_off // 2 + n].view(NT, T, 3, 512)
    hi = torch.stack([t[:, :, 0], t[:, :, 2]], dim=2)                              # [NT, T, 2, 512] hi fragments of k-steps 2T, 2T + 1
    lorec = t[:, :, 1].contiguous().view(torch.uint8).view(NT, T, 64, 16).to(torch.int64)
    scale_byte = lorec[..., 12]
    assert not lorec[..., 13:16].any()                                              # the scale byte alone in its dword
    alpha = torch.exp2((127 - scale_byte).double())                                # [NT, T, 64]
    hv = hi.contiguous().view(torch.float16).double().view(NT, T, 2, 64, 8).permute(0, 1, 3, 2, 4).reshape(NT, T, 64, 16)
    codes = sp.SynthesisPlan.e2m3_codes((hv * alpha.unsqueeze(-1)).float())      # slots 0-15: 8 j + e  [NT, T, 64, 16]
    c = codes.view(NT, T, 64, 4, 4)
    b0 = c[..., 0] | ((c[..., 1] & 3) << 6)
    b1 = (c[..., 1] >> 2) | ((c[..., 2] & 15) << 4)
    b2 = (c[..., 2] >> 4) | (c[..., 3] << 2)
    hib = torch.stack([b0, b1, b2], dim=-1).reshape(NT, T, 64, 12)                 # dwords 0-2 of the record
    sc4 = scale_byte.unsqueeze(-1).expand(-1, -1, -1, 4)
    full = torch.cat([hib, lorec[..., 0:12], sc4, sc4], dim=-1).to(torch.uint8)     # [NT, T, 64, 32 B]: dwords 0-2 | 3-5 | 6 | 7
    st = torch.zeros(KS, NT, 2, 1024, dtype=torch.uint8)
    for j in range(2):
        st[j::2, :, 0] = hi[:, :, j].contiguous().view(torch.uint8).view(NT, T, 1024).transpose(0, 1)
        st[j::2, :, 1] = full[..., 16 * j: 16 * j + 16].reshape(NT, T, 1024).transpose(0, 1)
    return decode_stream(st.view(torch.int16).flatten(), 0, KS, NT, order=None if acc_order else (lambda ks, h, e: 16 * ks + 8 * h + e),
                         dense=False)


def emulate(plan, fmap_lowres, fixed_style, Hr, Wr, H, W, x2=False):
    x3t = plan.build_x3t(torch.float16, "x2") if x2 else plan.build_x3t()
    NT, HdP = x3t["NT"], x3t["HdP"]
    desc, tab, wb = x3t["desc"], x3t["tables"].double(), x3t["wblob"]
    G, cst, ab = plan.per_forward_tables(fmap_lowres.float(), fixed_style.float(), HdP)
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
    lrelu = lambda v: torch.maximum(v, 0.2 * v)
    decode = decode_x2 if x2 else decode_matrix

    def mm(y, ops):                                                   # y @ W.t() in the engine's arithmetic
        if not x2:
            return y @ ops.t()
        from x2_emulation import x2_operands_matmul
        return x2_operands_matmul(y.float(), ops[0], ops[1], dynamic=True)       # activations are fp32 in the kernel

    for k in range(desc.n_blocks):
        bk = desc.block[k]
        x_in = x
        for s in range(2):
            d = bk.spade[s]
            if d.pixel_style:
                assert not bk.skip
                a = torch.relu(Gup[:, :, d.g_offset: d.g_offset + 128] + cst[:, d.cst_index].double()[:, None, :])
                g1 = vec(d.vec) + mm(a, decode(wb, d.w_gamma, 8, NT, False))
                y = lrelu((x * vec(d.vec + 2 * HdP) + vec(d.vec + 3 * HdP)) * g1 + vec(d.vec + HdP)
                          + mm(a, decode(wb, d.w_beta, 8, NT, False)))
            else:
                t2 = ab[:, d.ab_index].double()                       # [B, 2, HdP]
                y = lrelu(x * t2[:, 0:1] + t2[:, 1:2])
            x = mm(y, decode(wb, d.w_conv, 2 * NT, NT, True)) + vec(d.b_conv) + (x_in if (s == 1 and bk.skip) else 0.0)
        if bk.to_rgb:
            wr = torch.stack([vec(bk.w_rgb), vec(bk.w_rgb + HdP), vec(bk.w_rgb + 2 * HdP)])          # [3, HdP]
            rgb = rgb + x @ wr.t() + vec(bk.w_rgb + 3 * HdP, 3)
    return rgb.view(B, H, W, 3).permute(0, 3, 1, 2)


@pytest.mark.parametrize("mode,width,x2", [("mixed", 40, False), ("isolated", 384, False), ("mixed", 420, False), ("mixed", 40, True),
                                           ("mixed", 420, True)])
def test_x3t_plan_matches_oracle(mode, width, x2):
    meta = dict(load_golden("gen_tiny_mixed")["meta"])
    meta.update(map3d_mode=mode, hidden_dim=width, latent_dim=width, feature_dim=width, gen_height=6, gen_width=4,
                render_height=3, render_width=2)
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
    assert plan.x3t_supported()
    assert plan.engine == ("f16x2" if width <= 256 else "f16x2t")         # the default follows the width
    B, Hr, Wr, H, W = 2, 3, 2, 6, 4
    fmap = torch.randn(B, Hr * Wr, width)
    style = torch.randn(B, width)
    got = emulate(plan, fmap, style, Hr, Wr, H, W, x2)
    fm = fmap.view(B, Hr, Wr, width).permute(0, 3, 1, 2)
    fm_up = torch.nn.functional.interpolate(fm, (H, W), mode="bilinear")
    x0 = O.synthesis_input(sd, B, H, W)
    ref = O.synthesis_network(sd, x0, fm_up, style.view(B, 1, width), mode, tuple(meta["mod_blocks"]), meta["synthesis_blocks"])["final"]
    # bf16 hi + lo carries 16 significant bits of every weight: 1e-4 covers it comfortably; x2 (f16 hi + fp6 cross terms): the
    # engine's tolerance of the GPU tests
    assert rel_err(got.float(), ref) < (1e-3 if x2 else 1e-4)


def test_x3t_plan_refuses_what_the_kernel_cannot_run():
    meta = dict(load_golden("gen_tiny_mixed")["meta"])
    meta.update(map3d_mode="all", hidden_dim=300, latent_dim=300, feature_dim=300)        # per-pixel styles in skip blocks
    meta["neural_field_cls"] = impl.COORDCONCATSIREN
    Gn = gens.Map3DGenerator(**meta).eval()
    plan = sp.SynthesisPlan(Gn.state_dict(), "synthesis_network", "synthesis_input", meta["synthesis_blocks"],
                            tuple(meta["mod_blocks"]), "all", torch.device("cpu"))
    assert not plan.x3t_supported() and plan.engine == "f32"
