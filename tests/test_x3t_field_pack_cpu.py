"""Host-side packing of the LDS-resident split-f16 field engine (h3d_field_pack_x3t, a HOST function of libh3d.so) checked
on the CPU: the blob is decoded through h3d_field_x3t_layout -- tile-major A fragments, accumulator-order K permutation,
power-of-two scales, hi + lo halves, biases, fragment-order head rows -- and a float64 restatement of the kernel's algebra
on the decoded data must reproduce the oracle's COORDCONCATSIREN.  Widths 384 / 420 are the ones this engine exists for.
No GPU, no kernel launch."""
import ctypes
import importlib

import pytest
import torch

import h3d_oracle as O
from conftest import rel_err

L = importlib.import_module("3dhumangan_amd._lib")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")

K_SIN = 64.0        # input scale of csrc/field_x3t.hip (coordinates, geometry features)
W_NAMES = ["coord", "geo", "f0", "f1", "f2", "f3", "color", "feat"]


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
    nbytes = lib.h3d_field_pack_x3t_size(Hd, F)
    blob = torch.zeros(nbytes, dtype=torch.uint8)
    L.check(lib.h3d_field_pack_x3t(ctypes.byref(P), Hd, F, ctypes.c_void_p(blob.data_ptr())), "h3d_field_pack_x3t")
    lay = (ctypes.c_int64 * 18)()
    L.check(lib.h3d_field_x3t_layout(Hd, F, lay, 18), "h3d_field_x3t_layout")
    return blob, list(lay)


def matrix(blob, off, NT, kstot, ks0, n_ks, acc_order):
    """A fragments [NT][kstot][2][64][8] -> scaled dense matrix [32*NT, 16*n_ks] over k-steps ks0.., natural feature order."""
    n = NT * kstot * 2 * 64 * 8
    t = blob[off: off + 2 * n].view(torch.float16).double().view(NT, kstot, 2, 64, 8)
    t = t[:, ks0: ks0 + n_ks, 0] + t[:, ks0: ks0 + n_ks, 1]                      # [NT, n_ks, 64, 8]
    W = torch.zeros(32 * NT, 16 * n_ks, dtype=torch.float64)
    for ks in range(n_ks):
        for h in range(2):
            for e in range(8):
                k = acc_k(ks, h, e) if acc_order else 16 * ks + 8 * h + e
                W[:, k] = t[:, ks, 32 * h: 32 * h + 32, e].reshape(-1)
    return W


@pytest.mark.parametrize("Hd", [40, 200, 384, 420])
def test_field_x3t_pack_decodes_to_the_reference_network(Hd):
    F = Hd
    torch.manual_seed(Hd)
    net = impl.COORDCONCATSIREN(input_dim=3, latent_dim=Hd, hidden_dim=Hd, geo_feature_dim=31, output_dim=F + 4, feature_dim=F,
                                num_blocks=4)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    blob, lay = pack(net, Hd, F)
    NT, KS, HdP = lay[0:3]
    assert NT % 2 == 0 and NT >= 4 and KS == 2 * NT and HdP == 32 * NT and HdP >= Hd
    woff = dict(zip(W_NAMES, lay[3:11]))
    inv_off, bias_off, bfeat_off, headw_off, headinv_off, headb_off, total = lay[11:18]
    assert total == blob.numel()
    f32 = lambda off, n: blob[off: off + 4 * n].view(torch.float32).double()
    inv = dict(zip(W_NAMES, f32(inv_off, 8)))
    bias = f32(bias_off, 7 * HdP).view(7, HdP)          # coord, geo, film0..3, colour
    Wc = matrix(blob, woff["coord"], NT, 1, 0, 1, False)
    Wg = matrix(blob, woff["geo"], NT, 2, 0, 2, False)
    W0a = matrix(blob, woff["f0"], NT, 2 * KS, 0, KS, True)
    W0b = matrix(blob, woff["f0"], NT, 2 * KS, KS, KS, True)
    Wl = {l: matrix(blob, woff[f"f{l}"], NT, KS, 0, KS, True) for l in (1, 2, 3)}
    Wcol = matrix(blob, woff["color"], NT, KS + 1, 0, KS, True)
    Wdir = matrix(blob, woff["color"], NT, KS + 1, KS, 1, False)
    Wf = matrix(blob, woff["feat"], NT, KS, 0, KS, True)

    N = 29
    g = torch.Generator().manual_seed(1)
    pts, geo = torch.rand(1, N, 3, generator=g) * 2 - 1, torch.rand(1, N, 31, generator=g) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(1, N, 3, generator=g), dim=-1)
    freq, phase = torch.randn(1, 4 * Hd, generator=g) * 0.5, torch.randn(1, 4 * Hd, generator=g)
    scaler = 0.7
    f = (freq[0].double() * 15 + 30).view(4, Hd)
    ph = phase[0].double().view(4, Hd)
    padk = lambda x, K: torch.nn.functional.pad(x, (0, K - x.shape[1]))
    padn = lambda v: torch.nn.functional.pad(v, (0, HdP - Hd))

    def film(pre_acc, inv_s, b, fr, p):                 # the kernel's sine epilogue, on the padded width
        return torch.sin(padn(fr) * (pre_acc * inv_s + b) + padn(p))

    thirty, zeros = torch.full((Hd,), 30.0, dtype=torch.float64), torch.zeros(Hd, dtype=torch.float64)
    a_c = film(padk(pts[0].double() * scaler * K_SIN, 16) @ Wc.t(), inv["coord"], bias[0], thirty, zeros)
    a_g = film(padk(geo[0].double() * K_SIN, 32) @ Wg.t(), inv["geo"], bias[1], thirty, zeros)
    for t in (a_c, a_g):
        assert float(t[:, Hd:].abs().max() if HdP > Hd else 0.0) == 0.0       # padding channels stay exactly zero
    x = film(a_c @ W0a.t() + a_g @ W0b.t(), inv["f0"], bias[2], f[0], ph[0])
    for l in (1, 2, 3):
        x = film(x @ Wl[l].t(), inv[f"f{l}"], bias[2 + l], f[l], ph[l])
    c = film(x @ Wcol.t() + padk(dirs[0].double(), 16) @ Wdir.t(), inv["color"], bias[6], f[3], ph[3])
    # heads: one A tile [KS][hi|lo][64 lanes][8] f16 whose rows 0..3 are sigma, r, g, b (rows 4..31 zero), accumulator-order K
    ht = blob[headw_off: headw_off + KS * 2048].view(torch.float16).double().view(KS, 2, 64, 8).sum(1)      # hi + lo
    assert float(ht.view(KS, 2, 32, 8)[:, :, 4:].abs().max()) == 0.0
    hv = torch.zeros(4, HdP, dtype=torch.float64)
    for ks in range(KS):
        for h in range(2):
            for e in range(8):
                hv[:, acc_k(ks, h, e)] = ht[ks, 32 * h: 32 * h + 4, e]
    hinv, hb = f32(headinv_off, 4), f32(headb_off, 4)
    sigma = x @ hv[0] * hinv[0] + hb[0]
    rgb = torch.sigmoid(c @ hv[1:4].t() * hinv[1:4] + hb[1:4])
    feat = (c @ Wf.t() * inv["feat"])[:, :F] + f32(bfeat_off, HdP)[:F]
    got = torch.cat([rgb, feat, sigma[:, None]], dim=1)

    sd = {"neural_field." + k: v.detach() for k, v in net.state_dict().items()}
    ref = O.neural_field({k: v.double() for k, v in sd.items()}, pts.double(), freq.double(), phase.double(), geo.double(),
                         dirs.double(), input_scaler=scaler)[0]
    # f16 hi + lo carries 22 significant bits of every (scaled) weight; activations are exact here
    assert rel_err(got, ref) < 1e-5
