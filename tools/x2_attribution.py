"""CPU only: which contractions of the synthesis network carry the x2 engine's error on the bench workload.

The float64 restatement of synthesis_x3_kernel on the plan's own tables / weight streams (tests/test_x3_plan_cpu.py: emulate)
is run on a pixel subset of one item of bench.py's batch (MAP3DBN512, 512x512, seed 1234) with the rendered feature maps of the
CPU oracle, once with every contraction exact (the bf16 hi + lo stream, 16 significant bits), once with every contraction in
the x2 arithmetic (tests/x2_emulation.py on the decoded f16 fragments / fp6 records), and once per contraction with ONLY that
one in x2.  Error = per-channel max |rgb - rgb_exact| / max |rgb_exact| over the subset, the measure of bench.py's `checked`.

usage: python tools/x2_attribution.py [item=14] [n_cells=8] [x3_list="29,28"]   (x3_list: contractions kept exact in a mixed plan)
"""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import h3d_oracle as O  # noqa: E402
from test_x3_plan_cpu import decode_matrix, decode_x2  # noqa: E402
from x2_emulation import x2_operands_matmul  # noqa: E402

sp = importlib.import_module("3dhumangan_amd.lib.generators.synthesis_pack")


def contraction_names(desc):
    names = []
    for k in range(desc.n_blocks):
        for s in range(2):
            if desc.block[k].spade[s].pixel_style:
                names += [f"b{k}.spade{s}.gamma", f"b{k}.spade{s}.beta"]
            names.append(f"b{k}.conv{s}")
    return names


def emulate_subset(plan, G_rays, taps, wts, ii, jj, fixed_style, fmap_rays, x2_set):
    """x2_set: set of contraction indices evaluated in the x2 arithmetic (the rest exact) -> rgb [P, 3] float64."""
    x3e, x3a = plan.build_x3(False), plan.build_x3(True)
    seg_e, seg_a = x3e["segments"][0], x3a["segments"][0]
    NT, HdP = x3a["NT"], x3a["HdP"]
    desc, tab = seg_a["desc"], seg_a["tables"].double()
    _, cst, ab = plan.x3_forward_tables(fmap_rays.float(), fixed_style.float(), True)
    vec = lambda off, n=HdP: tab[off: off + n]
    x = torch.sin(ii[:, None] * vec(desc.w_in) + jj[:, None] * vec(desc.w_in + HdP) + vec(desc.b_in))        # [P, HdP]
    Gup = None
    if G_rays is not None:
        Gd = G_rays.double()[0]                                                                                # [Rs, 128 np]
        Gup = sum(wts[t][:, None] * Gd[taps[t]] for t in range(4))                                            # [P, 128 np]
    rgb = torch.zeros(x.shape[0], 3, dtype=torch.float64)
    lrelu = lambda v: torch.maximum(v, 0.2 * v)
    stage, gi = 0, 0

    def mm(y, KS):
        nonlocal stage, gi
        if gi in x2_set:
            ops = decode_x2(seg_a["stream"], stage, KS, NT)
            yp = torch.nn.functional.pad(y, (0, ops[0].shape[1] - y.shape[-1]))
            out = x2_operands_matmul(yp, ops[0], ops[1], dynamic=True)
        else:
            W = decode_matrix(seg_e["stream"], stage, KS, NT)
            out = torch.nn.functional.pad(y, (0, W.shape[1] - y.shape[-1])) @ W.t()
        stage += KS
        gi += 1
        return out

    for k in range(desc.n_blocks):
        bk = desc.block[k]
        x_in = x
        for s in range(2):
            d = bk.spade[s]
            if d.pixel_style:
                a = torch.relu(Gup[:, d.g_offset: d.g_offset + 128] + cst[0, d.cst_index].double()[None, :])
                g1 = vec(d.vec) + mm(a, 8)
                bt = mm(a, 8)
                y = lrelu((x * vec(d.vec + 2 * HdP) + vec(d.vec + 3 * HdP)) * g1 + vec(d.vec + HdP) + bt)
            else:
                t4 = ab[0, d.ab_index].double()
                sc, sh = t4[:, 0, :].reshape(1, HdP), t4[:, 1, :].reshape(1, HdP)
                u = x * sc + sh
                y = 1.5 * u + u.abs()
            x = mm(y, 2 * NT) + (x_in if (s == 1 and bk.skip) else 0.0)
        if bk.to_rgb:
            wr = torch.stack([vec(bk.w_rgb), vec(bk.w_rgb + HdP), vec(bk.w_rgb + 2 * HdP)])
            rgb = rgb + x @ wr.t() + vec(bk.w_rgb + 3 * HdP, 3)
    assert stage == seg_a["stages"]
    return rgb


def main():
    item = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    n_cells = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    keep_exact = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 and sys.argv[3] else []
    configs = importlib.import_module("3dhumangan_amd.configs")
    gens = importlib.import_module("3dhumangan_amd.lib.generators")
    impl = importlib.import_module("3dhumangan_amd.lib.implicit_funcitions")
    cfg = {k: v for k, v in configs.MAP3DBN512.items() if isinstance(k, str)}
    cfg.update(gen_height=512, gen_width=512, render_height=96, render_width=96, num_steps=64, dataset_length=4, nerf_noise=0,
               last_back=cfg["eval_last_back"])
    torch.manual_seed(1234)
    Gn = gens.Map3DGenerator(**dict(cfg, neural_field_cls=impl.COORDCONCATSIREN)).eval()
    sd = {k: v.detach().clone() for k, v in Gn.state_dict().items()}
    z, cond, jitter = bench.make_inputs(cfg, 16, "cpu", seed=1234)
    Hr = Wr = 96
    H = W = 512
    g = torch.Generator().manual_seed(5)
    cells = [(0, 0), (Hr - 1, Wr - 1), (Hr // 2, Wr // 2)] + list(zip(torch.randint(0, Hr, (n_cells,), generator=g).tolist(),
                                                                      torch.randint(0, Wr, (n_cells,), generator=g).tolist()))
    pix = O.pixels_of_cells(cells, (H, W), (Hr, Wr))
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = O.generator_forward_subset(sd, cfg, z[item:item + 1], {k: v[item:item + 1] for k, v in cond.items()},
                                         jitter[item:item + 1], pix)
    plan = sp.SynthesisPlan(sd, "synthesis_network", "synthesis_input", cfg["synthesis_blocks"], tuple(cfg["mod_blocks"]),
                            cfg["map3d_mode"], torch.device("cpu"))
    plan.X2_HEADS = False          # per-block ToRGB tables (the ToRGB head tiles of the shipped x2 plan are one more contraction each)
    plan.X2_MID_X3 = False         # the all-x2 stream: the attribution is what decided to put block 3 on three products (round 6)
    plan._x2 = None                # (the constructor's fit test has built and cached the default stream)
    fm = ref["feature_maps"][0].t().unsqueeze(0).contiguous()                     # [1, Rs, F] channels last
    G_rays, _, _ = plan.x3_forward_tables(fm.float(), ref["styles"].reshape(1, -1).float(), True)
    Y, X = pix // W, pix % W
    y0, y1, ty = O._resize_axis(Hr, H, torch.float64)
    x0, x1, tx = O._resize_axis(Wr, W, torch.float64)
    txp, typ = tx[X], ty[Y]
    wts = [(1 - txp) * (1 - typ), txp * (1 - typ), (1 - txp) * typ, txp * typ]
    ii = torch.linspace(-1, 1, H, dtype=torch.float64)[Y]
    jj = torch.linspace(-1, 1, W, dtype=torch.float64)[X]
    style = ref["styles"].reshape(1, -1)
    names = contraction_names(plan.build_x3(True)["segments"][0]["desc"])
    n = len(names)
    run = lambda s: emulate_subset(plan, G_rays, ref["taps"], wts, ii, jj, style, fm, s)
    exact = run(set())
    orc = ref["rgbs"][0].double().t()                                             # [P, 3]
    den = exact.abs().amax(dim=0)
    err = lambda a: float(((a - exact).abs().amax(dim=0) / den).max())
    out = dict(item=item, pixels=int(len(pix)), contractions=n,
               plumbing_exact_vs_oracle=float(((exact - orc).abs().amax(dim=0) / orc.abs().amax(dim=0)).max()),
               all_x2=err(run(set(range(n)))), only={})
    print(json.dumps({k: v for k, v in out.items() if k != "only"}), flush=True)
    for i in range(n):
        out["only"][names[i]] = err(run({i}))
        print(f"  only {i:2d} {names[i]:20s} {out['only'][names[i]]:.3e}", flush=True)
    ranked = sorted(range(n), key=lambda i: -out["only"][names[i]])
    out["ranked"] = [(i, names[i], out["only"][names[i]]) for i in ranked[:8]]
    for keep in ([ranked[:1], ranked[:2], ranked[:4]] + ([keep_exact] if keep_exact else [])):
        out[f"x2_except_{','.join(str(i) for i in keep)}"] = err(run(set(range(n)) - set(keep)))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
