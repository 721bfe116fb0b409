"""Config registry with the reference's API (configs/__init__.py): get_config(opt), extract_metadata(curriculum,
step) and the up-sample step helpers.  Class names in the dicts are resolved against this package's mirrors
(lib.implicit_funcitions), exactly where the reference resolves them."""
from ..lib import implicit_funcitions
from .map3d import MAP3DBN, MAP3DBN512, MAP3DBN512L  # noqa: F401

_REGISTRY = {"MAP3DBN": MAP3DBN, "MAP3DBN512": MAP3DBN512, "MAP3DBN512L": MAP3DBN512L}


def _int_steps(curriculum):
    return sorted(k for k in curriculum if type(k) == int)


def extract_metadata(curriculum, current_step):
    """Static keys + the entries of the latest curriculum step <= current_step (reference :37-46)."""
    out = {}
    active = [s for s in _int_steps(curriculum) if s <= current_step]
    if active:
        out.update(curriculum[active[-1]])
    out.update({k: v for k, v in curriculum.items() if type(k) != int})
    return out


def _size(meta_or_step, default_w, default_h):
    return max(meta_or_step.get("render_width", default_w), meta_or_step.get("render_height", default_h))


def next_upsample_step(curriculum, current_step):
    """First curriculum step after current_step that raises the render size (reference :5-15)."""
    meta = extract_metadata(curriculum, current_step)
    cur = _size(meta, meta["gen_width"], meta["gen_height"])
    for s in _int_steps(curriculum):
        if s > current_step and _size(curriculum[s], 512, 512) > cur:
            return s
    return float("Inf")


def last_upsample_step(curriculum, current_step):
    """Start step of the current resolution stage (reference :17-28)."""
    meta = extract_metadata(curriculum, current_step)
    cur = max(meta.get("render_height", meta["gen_width"]), meta.get("render_width", meta["gen_height"]))
    for s in _int_steps(curriculum):
        if s <= current_step and _size(curriculum[s], meta["gen_width"], meta["gen_height"]) == cur:
            return s
    return 0


def get_current_step(curriculum, epoch):
    return sum(1 for e in curriculum["update_epochs"] if epoch >= e)


def get_config(opt):
    """opt.config in {MAP3DBN, MAP3DBN512, MAP3DBN512L}; opt.tune in {'', 'lr', 'map3d_mode'}; opt.variant index."""
    config = _REGISTRY[opt.config] if opt.config in _REGISTRY else globals()[opt.config]
    if isinstance(config["neural_field_cls"], str):
        config["neural_field_cls"] = getattr(implicit_funcitions, config["neural_field_cls"])
    tune = getattr(opt, "tune", "")
    if len(tune) == 0:
        return config
    if tune == "lr":
        gen_lr, disc_lr = [(1e-4, 4e-4), (2e-4, 2e-4), (1e-4, 2e-4), (1e-4, 1e-4)][opt.variant]
        for k in _int_steps(config):
            config[k]["gen_lr"], config[k]["disc_lr"] = gen_lr, disc_lr
        config["name"] = "{}_G_lr={}_D_lr={}".format(config["name"], gen_lr, disc_lr)
    elif tune == "map3d_mode":
        mode = ["isolated", "mixed", "all"][opt.variant]
        config["map3d_mode"] = mode
        config["name"] = "{}_map3d_mode={}".format(config["name"], mode)
    else:
        raise NotImplementedError
    return config
