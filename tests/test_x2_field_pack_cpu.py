"""Host-side packing of the x2 field engine (h3d_field_pack_x2, a HOST function of libh3d.so) checked on the CPU: the f16 hi
fragments and the fp6 records (32 six-bit e2m3 codes + block scale per lane and K-tile, split over the even / odd k-step's
stage) are decoded through h3d_field_x2_layout, and the arithmetic of csrc/x3_common.hpp (gemm_x2_roll) restated in float64
on the decoded data must equal tests/x2_emulation.py -- the model whose error against the reference's vectors is bounded in
tests/test_x2_error_model_cpu.py.  No GPU, no kernel launch."""
import ctypes
import importlib

import numpy as np
import pytest
import torch

from x2_emulation import acc_k, f16, q_e2m3, x2_matmul

L = importlib.import_module("3dhumangan_amd._lib")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
W_NAMES = ["coord", "f0a", "geo", "f0b", "f1", "f2", "f3", "color", "feat"]
CODES = torch.tensor([0, .125, .25, .375, .5, .625, .75, .875, 1, 1.125, 1.25, 1.375, 1.5, 1.625, 1.75, 1.875,
                      2, 2.25, 2.5, 2.75, 3, 3.25, 3.5, 3.75, 4, 4.5, 5, 5.5, 6, 6.5, 7, 7.5], dtype=torch.float64)


def pack(net, Hd, F):
    lib = L.load()
    lins = net._params_for_pack()
    host = [(l.weight.detach().float().contiguous(), l.bias.detach().float().contiguous()) for l in lins]
    P = L.FieldParams()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    P.w_coord, P.b_coord = vp(host[0][0]), vp(host[0][1])
    P.w_geo, P.b_geo = vp(host[1][0]), vp(host[1][1])
    for k in range(4):
        P.w_film[k], P.b_film[k] = host[2 + k][0].data_ptr(), host[2 + k][1].data_ptr()
    P.w_sigma, P.b_sigma = vp(host[6][0]), vp(host[6][1])
    P.w_color, P.b_color = vp(host[7][0]), vp(host[7][1])
    P.w_rgb, P.b_rgb = vp(host[8][0]), vp(host[8][1])
    P.w_feat, P.b_feat = vp(host[9][0]), vp(host[9][1])
    nbytes = lib.h3d_field_pack_x2_size(Hd, F)
    blob = torch.zeros(nbytes, dtype=torch.uint8)
    L.check(lib.h3d_field_pack_x2(ctypes.byref(P), Hd, F, ctypes.c_void_p(blob.data_ptr())), "h3d_field_pack_x2")
    lay = (ctypes.c_int64 * 20)()
    L.check(lib.h3d_field_x2_layout(Hd, F, lay, 20), "h3d_field_x2_layout")
    return blob, list(lay), host


def decode_x2(blob, off, KS, NT):
    """-> (Whi [32 NT, 16 KS] f16 values in feature order,
           per K-tile T, output row n, lane half h: codes_hi [16], codes_lo [16] (values), scale 2^(byte - 127))"""
    st = blob[off: off + KS * NT * 2048].view(KS, NT, 2, 1024)
    Whi = torch.zeros(32 * NT, 16 * KS, dtype=torch.float64)
    hi = st[:, :, 0].contiguous().view(torch.float16).double().view(KS, NT, 64, 8)
    for ks in range(KS):
        for h in range(2):
            for e in range(8):
                Whi[:, acc_k(ks, h, e)] = hi[ks, :, 32 * h: 32 * h + 32, e].reshape(-1)
    recs = {}
    half = st[:, :, 1].contiguous().view(torch.int32).view(KS, NT, 256)
    for T in range(KS // 2):
        # even stage: dwords 0-3 at 16 B per lane; odd stage DENSE: dwords 4-5 as [64][2], scale dwords as [64], 64 zero dwords
        ev, od = half[2 * T].view(NT, 64, 4), half[2 * T + 1]
        assert not od[:, 192:].any()
        rec = torch.cat([ev, od[:, :128].view(NT, 64, 2), od[:, 128:192].view(NT, 64, 1)], dim=-1).numpy().astype(np.uint32)  # [NT, 64, 7]
        bits = np.zeros((NT, 64, 32), dtype=np.int64)
        for s in range(32):
            b = 6 * s
            v = rec[..., b // 32].astype(np.uint64) >> np.uint64(b & 31)
            if (b & 31) > 26:
                v |= rec[..., b // 32 + 1].astype(np.uint64) << np.uint64(32 - (b & 31))
            bits[..., s] = (v & np.uint64(63)).astype(np.int64)
        vals = CODES[torch.from_numpy(bits & 31)] * torch.where(torch.from_numpy(bits & 32) > 0, -1.0, 1.0)
        sb = rec[..., 6]
        assert np.all(sb == (sb & 255) * 0x01010101)                                      # the e8m0 byte fills the scale dword
        scale = torch.from_numpy(np.ldexp(1.0, (sb & 255).astype(np.int64) - 127))
        recs[T] = (vals, scale)                        # vals [NT, 64, 32]; lanes = 32 * h + (row % 32)
    return Whi, recs


@pytest.mark.parametrize("Hd", [64, 256])
def test_field_x2_pack_is_the_emulated_arithmetic(Hd):
    F = Hd
    torch.manual_seed(Hd + 1)
    net = impl.COORDCONCATSIREN(input_dim=3, latent_dim=Hd, hidden_dim=Hd, geo_feature_dim=31, output_dim=F + 4, feature_dim=F,
                                num_blocks=4)
    blob, lay, host = pack(net, Hd, F)
    NT, KS, HdP, n_stages = lay[0:4]
    woff = dict(zip(W_NAMES, lay[4:13]))
    inv = dict(zip(W_NAMES, blob[lay[13]: lay[13] + 36].view(torch.float32).double()))
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(23, Hd, generator=g, dtype=torch.float64) * 2 - 1).float().double()          # sine outputs: |x| <= 1
    for name, Wsrc in (("f1", host[3][0]), ("f3", host[5][0]), ("feat", host[9][0]), ("color", host[7][0][:, 3:])):
        Whi, recs = decode_x2(blob, woff[name], KS, NT)
        sc = 1.0 / float(inv[name])                                                            # activation scale kSA = 1
        # hi plane = f16(W * sc) in accumulator order, zero padding outside the matrix
        want = torch.zeros(32 * NT, 16 * KS, dtype=torch.float64)
        want[:Wsrc.shape[0], :Hd] = f16(Wsrc.double() * sc)
        assert torch.equal(Whi, want), name
        # the kernel's arithmetic on the decoded operands
        xp = torch.nn.functional.pad(x, (0, 16 * KS - Hd))
        xh = f16(xp)
        xl = xp - xh
        Bh, Bl = q_e2m3(xh * 4.0), q_e2m3(f16(xl * 4096.0) * 4.0)
        y = xh @ Whi.t()
        for T, (vals, scale) in recs.items():
            for h in range(2):
                feats = torch.tensor([acc_k(2 * T + j, h, e) for j in range(2) for e in range(8)])
                a = vals[:, 32 * h: 32 * h + 32].reshape(32 * NT, 32)                           # rows n = 32 * nt + (lane & 31)
                s = scale[:, 32 * h: 32 * h + 32].reshape(32 * NT)
                cross = Bl[:, feats] @ a[:, :16].t() + Bh[:, feats] @ a[:, 16:].t()
                y = y + cross * s * 2.0 ** -14
        y = (y / sc)[:, :Wsrc.shape[0]]
        emu = x2_matmul(x, Wsrc)
        assert float((y - emu).abs().max() / emu.abs().max()) < 1e-12, name
        exact = x @ Wsrc.double().t()
        assert float((y - exact).abs().max() / exact.abs().max()) < 2e-5, name
    # heads: third plane = hi * 2^-12
    hw = blob[lay[16]: lay[16] + 2 * 4 * 3 * KS * 16].view(torch.float16).double().view(4, 3, KS, 2, 8)
    assert torch.equal(hw[:, 2], f16(hw[:, 0] / 4096.0))


def decode_x2t(blob, off, KStot, ks0, KS, NT):
    """Tile-major x2c tiles of the LDS-resident engine (round 6, csrc/x3t_common.hpp): per K-tile 3 KiB = [hi fragment of k-step 2T |
    lo record: 12 B of lo codes + the scale byte in a dword | hi fragment of k-step 2T + 1]; a trailing odd k-step of the matrix (the
    colour layer's view-direction k-step) keeps the 2 KiB x3 format behind the K-tiles -> as decode_x2.  The hi codes are not stored:
    the kernel converts them from the hi fragments with the lane's scale (v_cvt_scalef32_pk32_fp6_f16); restated here with q_e2m3."""
    tile_bytes = (KStot // 2) * 3072 + (KStot & 1) * 2048
    assert ks0 % 2 == 0 and KS % 2 == 0
    tiles = blob[off: off + NT * tile_bytes].view(NT, tile_bytes)
    kt = tiles[:, (ks0 // 2) * 3072: (ks0 // 2 + KS // 2) * 3072].reshape(NT, KS // 2, 3, 1024)
    Whi = torch.zeros(32 * NT, 16 * KS, dtype=torch.float64)
    hi = torch.stack([kt[:, :, 0], kt[:, :, 2]], dim=2).contiguous().view(torch.float16).double().view(NT, KS, 64, 8)
    for ks in range(KS):
        for h in range(2):
            for e in range(8):
                Whi[:, acc_k(ks, h, e)] = hi[:, ks, 32 * h: 32 * h + 32, e].reshape(-1)
    recs = {}
    lorec = kt[:, :, 1].contiguous().view(torch.int32).view(NT, KS // 2, 64, 4)
    for T in range(KS // 2):
        rec = lorec[:, T].numpy().astype(np.uint32)                          # [NT, 64, 4]: code dwords 3, 4, 5 and the scale dword
        sb = rec[..., 3]
        assert np.all(sb == (sb & 255))                                      # the scale byte alone
        scale = torch.from_numpy(np.ldexp(1.0, sb.astype(np.int64) - 127))   # 1 / alpha
        bits = np.zeros((NT, 64, 16), dtype=np.int64)
        for s_ in range(16):
            b_ = 6 * s_
            v = rec[..., b_ // 32].astype(np.uint64) >> np.uint64(b_ & 31)
            if (b_ & 31) > 26:
                v |= rec[..., b_ // 32 + 1].astype(np.uint64) << np.uint64(32 - (b_ & 31))
            bits[..., s_] = (v & np.uint64(63)).astype(np.int64)
        lo_vals = CODES[torch.from_numpy(bits & 31)] * torch.where(torch.from_numpy(bits & 32) > 0, -1.0, 1.0)     # [NT, 64, 16]
        hv = hi[:, 2 * T: 2 * T + 2].permute(0, 2, 1, 3).reshape(NT, 64, 16)                # slot 8 j + e = element e of k-step 2T + j
        hi_vals = q_e2m3(hv / scale.unsqueeze(-1))                                          # what the conversion makes of the fragments
        recs[T] = (torch.cat([hi_vals, lo_vals], dim=-1), scale)
    return Whi, recs


@pytest.mark.parametrize("Hd", [64, 420])
def test_field_x2t_pack_is_the_emulated_arithmetic(Hd):
    """h3d_field_pack_x2t (the x2 tier of the LDS-resident engine: the matrix offsets of h3d_field_pack_x3t, tiles in the x2c format)."""
    F = Hd
    torch.manual_seed(Hd + 2)
    net = impl.COORDCONCATSIREN(input_dim=3, latent_dim=Hd, hidden_dim=Hd, geo_feature_dim=31, output_dim=F + 4, feature_dim=F,
                                num_blocks=4)
    lib = L.load()
    lins = net._params_for_pack()
    host = [(l.weight.detach().float().contiguous(), l.bias.detach().float().contiguous()) for l in lins]
    P = L.FieldParams()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    P.w_coord, P.b_coord = vp(host[0][0]), vp(host[0][1])
    P.w_geo, P.b_geo = vp(host[1][0]), vp(host[1][1])
    for k in range(4):
        P.w_film[k], P.b_film[k] = host[2 + k][0].data_ptr(), host[2 + k][1].data_ptr()
    P.w_sigma, P.b_sigma = vp(host[6][0]), vp(host[6][1])
    P.w_color, P.b_color = vp(host[7][0]), vp(host[7][1])
    P.w_rgb, P.b_rgb = vp(host[8][0]), vp(host[8][1])
    P.w_feat, P.b_feat = vp(host[9][0]), vp(host[9][1])
    blob = torch.zeros(lib.h3d_field_pack_x3t_size(Hd, F), dtype=torch.uint8)
    L.check(lib.h3d_field_pack_x2t(ctypes.byref(P), Hd, F, ctypes.c_void_p(blob.data_ptr())), "h3d_field_pack_x2t")
    lay = (ctypes.c_int64 * 18)()
    L.check(lib.h3d_field_x3t_layout(Hd, F, lay, 18), "h3d_field_x3t_layout")
    NT, KS, HdP = lay[0:3]
    woff = list(lay[3:11])                                    # coord, geo, f0, f1, f2, f3, color, feat
    inv = blob[lay[11]: lay[11] + 32].view(torch.float32).double()
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(19, Hd, generator=g, dtype=torch.float64) * 2 - 1).float().double()
    cases = [("f1", woff[3], KS, 0, host[3][0], 3), ("f3", woff[5], KS, 0, host[5][0], 5), ("feat", woff[7], KS, 0, host[9][0], 7),
             ("color", woff[6], KS + 1, 0, host[7][0][:, 3:], 6), ("f0 geometry half", woff[2], 2 * KS, KS, host[2][0][:, Hd:], 2)]
    for name, off, KStot, ks0, Wsrc, wi in cases:
        Whi, recs = decode_x2t(blob, off, KStot, ks0, KS, NT)
        sc = 1.0 / float(inv[wi])
        want = torch.zeros(32 * NT, 16 * KS, dtype=torch.float64)
        want[:Wsrc.shape[0], :Hd] = f16(Wsrc.double() * sc)
        assert torch.equal(Whi, want), name
        xp = torch.nn.functional.pad(x, (0, 16 * KS - Hd))
        xh = f16(xp)
        Bh, Bl = q_e2m3(xh * 4.0), q_e2m3(f16((xp - xh) * 4096.0) * 4.0)
        y = xh @ Whi.t()
        for T, (vals, scale) in recs.items():
            for h in range(2):
                feats = torch.tensor([acc_k(2 * T + j, h, e) for j in range(2) for e in range(8)])
                a = vals[:, 32 * h: 32 * h + 32].reshape(32 * NT, 32)
                s = scale[:, 32 * h: 32 * h + 32].reshape(32 * NT)
                y = y + (Bl[:, feats] @ a[:, :16].t() + Bh[:, feats] @ a[:, 16:].t()) * s * 2.0 ** -14
        y = (y / sc)[:, :Wsrc.shape[0]]
        if name.startswith("f0"):
            # FiLM 0's two halves share one matrix scale (the largest weight of the WHOLE matrix): emulate with it
            from x2_emulation import weight_alpha, slot_groups
            Ws = Wsrc.double() * sc
            Wh = f16(Ws)
            emu = f16(x) @ Wh.t()
            for gi in slot_groups(Hd):
                al = weight_alpha(Wh[:, gi], (Ws - Wh)[:, gi])
                emu = emu + (q_e2m3(f16((x - f16(x)) * 4096.0) * 4.0)[:, gi] @ q_e2m3(Wh[:, gi] * al[:, None]).t()
                             + q_e2m3(f16(x) * 4.0)[:, gi] @ q_e2m3((Ws - Wh)[:, gi] * 4096.0 * al[:, None]).t()) / (al * 4.0 * 4096.0)
            emu = emu / sc
        else:
            emu = x2_matmul(x, Wsrc)
        assert float((y - emu).abs().max() / emu.abs().max()) < 1e-12, name
