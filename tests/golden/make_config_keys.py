"""Captures the OPERATOR CONTRACT of the reference as data (run in the development container only; reads
/root/reference, writes tests/golden/config_keys.json):

  * the `net:` / `loss:` / `solver:` / `optimizer:` / `scheduler:` blocks of configs/model/anomaly_clip_*.yaml with
    their `${data.*}` interpolations resolved against configs/data/*.yaml (keys and scalar values only),
  * the datamodule hparams the LightningModule reaches into,
  * configs/trainer/ddp.yaml,
  * the argument names of the hooks `pytorch_lightning.Trainer` calls on `AnomalyCLIPModule` (parsed with `ast`, names
    only).

No reference source text is stored."""
import ast
import json
import os
import re

import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_keys.json")
PAIRS = {"ucf": ("anomaly_clip_ucfcrime.yaml", "ucfcrime.yaml"), "sht": ("anomaly_clip_shanghaitech.yaml", "shanghaitech.yaml"),
         "xd": ("anomaly_clip_xdviolence.yaml", "xdviolence.yaml")}


def resolve(v, data):
    if isinstance(v, str):
        m = re.fullmatch(r"\$\{data\.(\w+)\}", v)
        if m:
            return data[m.group(1)]
    return v


def main():
    out = {"configs": {}}
    for key, (mf, df) in PAIRS.items():
        model = yaml.safe_load(open(os.path.join(REF, "configs", "model", mf)))
        data = yaml.safe_load(open(os.path.join(REF, "configs", "data", df)))
        ent = {"num_classes": resolve(model["num_classes"], data)}
        for blk in ("net", "loss", "solver", "optimizer", "scheduler"):
            ent[blk] = {k: resolve(v, data) for k, v in model[blk].items()}
        ent["data"] = {k: data[k] for k in ("num_segments", "seg_length", "batch_size", "batch_size_test", "num_classes",
                                            "load_from_features", "normal_id", "stride", "ncrops", "labels_file", "visualize")}
        out["configs"][key] = ent
    out["trainer_ddp"] = yaml.safe_load(open(os.path.join(REF, "configs", "trainer", "ddp.yaml")))
    tree = ast.parse(open(os.path.join(REF, "src", "models", "anomaly_clip_module.py")).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "AnomalyCLIPModule")
    hooks = {}
    for n in cls.body:
        if isinstance(n, ast.FunctionDef):
            hooks[n.name] = {"args": [a.arg for a in n.args.args], "kwarg": n.args.kwarg.arg if n.args.kwarg else None,
                             "decorators": [d.id for d in n.decorator_list if isinstance(d, ast.Name)]}
    out["module_hooks"] = hooks
    for name, path in (("AnomalyCLIP.forward", "src/models/components/anomaly_clip.py"), ("ComputeLoss.__call__", "src/models/components/loss.py")):
        t = ast.parse(open(os.path.join(REF, path)).read())
        cname, fname = name.split(".")
        c = next(n for n in t.body if isinstance(n, ast.ClassDef) and n.name == cname)
        f = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == fname)
        hooks[name] = {"args": [a.arg for a in f.args.args], "kwarg": None, "decorators": []}
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
