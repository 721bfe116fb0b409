"""CPU-side checks of the generator mirror: state_dict schema, packing helpers (no GPU compute)."""
import ctypes
import importlib

import torch

from conftest import load_golden

gens = importlib.import_module("3dhumangan_amd.lib.generators")
impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
pack = importlib.import_module("3dhumangan_amd.lib.generators.synthesis_pack")
h3dlib = importlib.import_module("3dhumangan_amd._lib")
configs = importlib.import_module("3dhumangan_amd.configs")


def build(meta):
    cfg = dict(meta)
    cfg["neural_field_cls"] = impl.COORDCONCATSIREN
    return gens.Map3DGenerator(**cfg)


def test_state_dict_schema_matches_reference_checkpoint_layout():
    for name in ("gen_tiny_mixed", "gen_tiny_isolated_legacy"):
        g = load_golden(name)
        G = build(g["meta"])
        mine = {k: tuple(v.shape) for k, v in G.state_dict().items()}
        ref = {k: tuple(v.shape) for k, v in g["state"].items()}
        assert mine == ref
        G.load_state_dict(g["state"], strict=True)


def test_full_size_schema_counts():
    cfg = {k: v for k, v in configs.MAP3DBN512L.items() if isinstance(k, str)}
    cfg["neural_field_cls"] = impl.COORDCONCATSIREN
    cfg["dataset_length"] = 10
    with torch.device("meta"):
        G = gens.Map3DGenerator(**cfg)
    sd = G.state_dict()
    assert len(sd) == 343
    assert sum(v.numel() for v in sd.values()) > 11_000_000
    names = [n for n, _ in G.named_parameters()]
    for sub in ("neural_field_mapping_network", "synthesis_mapping_network", "latent_pool", "neural_field."):
        assert any(sub in n for n in names)               # LR groups of the reference trainer key on these


def test_torch_packing_equals_c_helper():
    lib = h3dlib.load()
    g = torch.Generator().manual_seed(0)
    for n_out, n_in, KB, NT in [(40, 31, 4, 2), (256, 128, 16, 8), (3, 3, 1, 1), (420, 420, 56, 14)]:
        w = torch.randn(n_out, n_in, generator=g)
        a = pack.pack_matrix(w, KB, NT)
        b = torch.empty(NT * KB * 256)
        rc = lib.h3d_pack_matrix(ctypes.c_void_p(w.data_ptr()), n_in, 0, n_in, n_out, KB, NT, ctypes.c_void_p(b.data_ptr()))
        assert rc == 0
        assert torch.equal(a, b)
        # spot-check the documented layout
        nt, kb, lane, e = NT - 1, KB - 1, 37, 2
        k, n = 8 * kb + 4 * (lane >> 5) + e, 32 * nt + (lane & 31)
        want = w[n, k] if (k < n_in and n < n_out) else 0.0
        assert float(b[((nt * KB + kb) * 64 + lane) * 4 + e]) == float(want)


def test_field_pack_size_matches_layout():
    lib = h3dlib.load()
    assert lib.h3d_field_pack_size(0, 4) == -1
    for Hd, F in [(32, 32), (256, 256), (420, 420), (64, 40)]:
        HdP, FP = (Hd + 31) // 32 * 32, (F + 31) // 32 * 32
        NT, NTF, KBH = HdP // 32, FP // 32, HdP // 8
        mats = NT * 256 * (2 + 4 + 2 * KBH + 3 * KBH + 2 + KBH) + NTF * KBH * 256
        vecs = HdP * (1 + 1 + 4 + 1 + 3 + 1 + 3) + FP + 4
        assert lib.h3d_field_pack_size(Hd, F) == 4 * (mats + vecs)
