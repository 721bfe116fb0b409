"""Host-side packing of the split-f16 field engine (h3d_field_pack_x3, a HOST function of libh3d.so) checked on the CPU:
the blob is decoded through h3d_field_x3_layout -- weight stream in consumption order, accumulator-order K permutation,
power-of-two scales, hi + lo halves, biases, head rows -- and a float64 restatement of the kernel's algebra on the decoded
data must reproduce the oracle's COORDCONCATSIREN.  No GPU, no kernel launch."""
import ctypes
import importlib

import pytest
import torch

import h3d_oracle as O
from conftest import rel_err

L = importlib.import_module("3dhumangan_amd._lib")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")

K_SIN, K_SA = 64.0, 1.0        # input / activation scales of csrc/field_x3.hip
W_NAMES = ["coord", "f0a", "geo", "f0b", "f1", "f2", "f3", "color", "feat"]


def acc_k(ks, h, e):
    return 32 * (ks // 2) + (e & 3) + 8 * (2 * (ks & 1) + (e >> 2)) + 4 * h


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
    nbytes = lib.h3d_field_pack_x3_size(Hd, F)
    blob = torch.zeros(nbytes, dtype=torch.uint8)
    L.check(lib.h3d_field_pack_x3(ctypes.byref(P), Hd, F, ctypes.c_void_p(blob.data_ptr())), "h3d_field_pack_x3")
    lay = (ctypes.c_int64 * 20)()
    L.check(lib.h3d_field_x3_layout(Hd, F, lay, 20), "h3d_field_x3_layout")
    return blob, list(lay), host


def stages(blob, off, n_stages, NT, acc_order):
    """-> scaled dense matrix [32*NT, 16*n_stages] (hi + lo), K in natural feature order."""
    n = n_stages * NT * 2 * 64 * 8
    t = blob[off: off + 2 * n].view(torch.float16).double().view(n_stages, NT, 2, 64, 8)
    t = t[:, :, 0] + t[:, :, 1]
    W = torch.zeros(32 * NT, 16 * n_stages, dtype=torch.float64)
    for ks in range(n_stages):
        for h in range(2):
            for e in range(8):
                k = acc_k(ks, h, e) if acc_order else 16 * ks + 8 * h + e
                W[:, k] = t[ks, :, 32 * h: 32 * h + 32, e].reshape(-1)
    return W


@pytest.mark.parametrize("Hd", [64, 40, 256])
def test_field_x3_pack_decodes_to_the_reference_network(Hd):
    F = Hd
    torch.manual_seed(Hd)
    net = impl.COORDCONCATSIREN(input_dim=3, latent_dim=Hd, hidden_dim=Hd, geo_feature_dim=31, output_dim=F + 4, feature_dim=F,
                                num_blocks=4)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    blob, lay, host = pack(net, Hd, F)
    NT, KS, HdP, n_stages = lay[0:4]
    woff = dict(zip(W_NAMES, lay[4:13]))
    inv_off, bias_off, bfeat_off, headw_off, headinv_off, headb_off, total = lay[13:20]
    assert total == blob.numel() and n_stages == 1 + 2 + 7 * KS + 1
    f32 = lambda off, n: blob[off: off + 4 * n].view(torch.float32).double()
    inv = dict(zip(W_NAMES, f32(inv_off, 9)))
    bias = f32(bias_off, 7 * HdP).view(7, HdP)          # coord, geo, film0..3, colour
    W = {name: stages(blob, woff[name], ks, NT, name not in ("coord", "geo")) for name, ks in
         [("coord", 1), ("f0a", KS), ("geo", 2), ("f0b", KS), ("f1", KS), ("f2", KS), ("f3", KS), ("feat", KS)]}
    Wcol = stages(blob, woff["color"], KS, NT, True)
    Wdir = stages(blob, woff["color"] + KS * NT * 2 * 64 * 8 * 2, 1, NT, False)

    N = 37
    g = torch.Generator().manual_seed(1)
    pts, geo = torch.rand(1, N, 3, generator=g) * 2 - 1, torch.rand(1, N, 31, generator=g) * 2 - 1
    freq, phase = torch.randn(1, 4 * Hd, generator=g) * 0.5, torch.randn(1, 4 * Hd, generator=g)
    scaler = 0.7
    f = (freq[0].double() * 15 + 30).view(4, Hd)
    ph = phase[0].double().view(4, Hd)
    padk = lambda x, K: torch.nn.functional.pad(x, (0, K - x.shape[1]))
    padn = lambda v: torch.nn.functional.pad(v, (0, HdP - Hd))

    def film(pre_acc, inv_s, b, fr, p):                 # what FilmProducer computes, on the padded width
        return torch.sin(padn(fr) * (pre_acc * inv_s + b) + padn(p))

    ones, zeros = torch.full((Hd,), 30.0, dtype=torch.float64), torch.zeros(Hd, dtype=torch.float64)
    xc = padk(pts[0].double() * scaler * K_SIN, 16)
    a_c = film(xc @ W["coord"].t(), inv["coord"], bias[0], ones, zeros)
    xg = padk(geo[0].double() * K_SIN, 32)
    a_g = film(xg @ W["geo"].t(), inv["geo"], bias[1], ones, zeros)
    for t in (a_c, a_g):
        assert float(t[:, Hd:].abs().max() if HdP > Hd else 0.0) == 0.0       # padding channels stay exactly zero
    x = film((a_c * K_SA) @ W["f0a"].t() + (a_g * K_SA) @ W["f0b"].t(), inv["f0a"], bias[2], f[0], ph[0])
    assert float(inv["f0a"]) == float(inv["f0b"])
    for l, name in ((1, "f1"), (2, "f2"), (3, "f3")):
        x = film((x * K_SA) @ W[name].t(), inv[name], bias[2 + l], f[l], ph[l])
    d = torch.zeros(N, 16, dtype=torch.float64)
    d[:, 2] = -1.0                                                             # lock_view_dependence
    c = film((x * K_SA) @ Wcol.t() + (d * K_SA) @ Wdir.t(), inv["color"], bias[6], f[3], ph[3])
    # heads: [head][hi|lo][KS][half][8] f16, accumulator-order K
    hw = blob[headw_off: headw_off + 2 * 4 * 2 * KS * 16].view(torch.float16).double().view(4, 2, KS, 2, 8).sum(1)
    hv = torch.zeros(4, HdP, dtype=torch.float64)
    for ks in range(KS):
        for h in range(2):
            for e in range(8):
                hv[:, acc_k(ks, h, e)] = hw[:, ks, h, e]
    hinv, hb = f32(headinv_off, 4), f32(headb_off, 4)
    sigma = (x * K_SA) @ hv[0] * hinv[0] + hb[0]
    rgb = torch.sigmoid((c * K_SA) @ hv[1:4].t() * hinv[1:4] + hb[1:4])
    feat = ((c * K_SA) @ W["feat"].t() * inv["feat"])[:, :F] + f32(bfeat_off, HdP)[:F]
    got = torch.cat([rgb, feat, sigma[:, None]], dim=1)

    sd = {"neural_field." + k: v.detach() for k, v in net.state_dict().items()}
    dirs = torch.zeros(1, N, 3)
    dirs[..., 2] = -1
    ref = O.neural_field({k: v.double() for k, v in sd.items()}, pts.double(), freq.double(), phase.double(), geo.double(),
                         dirs.double(), input_scaler=scaler)[0]
    # f16 hi + lo carries 22 significant bits of every (scaled) weight; activations are exact here
    assert rel_err(got, ref) < 1e-5
