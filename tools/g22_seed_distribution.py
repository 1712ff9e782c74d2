#!/usr/bin/env python3
"""Outcome distribution of golden G22's short schedule through the product's train CLI: final validation table and per-epoch train PSNR per seed
(a schedule this short is chaotic: see docs/history/round6.md section 2).  ~5 s per seed on an MI355X.
   SEEDS=0,1,2,...  python tools/g22_seed_distribution.py "batch_size=256" ["key=value;key=value" ...]     (one line per seed and setting)
   CPU_RNG=1|perm|jit: patch the pixel order / the jitter to torch's global CPU generator (the experiment of round 6)"""
import importlib.util, os, sys, json, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import make_synthetic_mos as gen
def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod); return mod
train = load(os.path.join(REPO, "trainer", "train_panopli_tensorf.py"), "cli")
sc = json.load(open(os.path.join(REPO, "tests/golden/g22_short_schedule.json")))["schedule"]
tmp = tempfile.mkdtemp(); os.chdir(tmp)
scene_dir = gen.make_scene(os.path.join(tmp, "data", "synth_scene"), n_frames=sc["n_frames"], size=sc["image_dim"], seed=sc["scene_seed"])
base = [f"+experiment={sc['experiment']}", f"dataset_root={scene_dir}", f"image_dim={sc['image_dim']}", f"min_grid_dim={sc['min_grid_dim']}", f"max_grid_dim={sc['max_grid_dim']}",
        f"max_epoch={sc['max_epoch']}", f"batch_size={sc['batch_size']}", f"chunk={sc['chunk']}", f"max_depth={sc['max_depth']}", f"max_rays_instances={sc['max_rays_instances']}",
        f"decay_step={sc['decay_step']}", "seed=0"]
import io, contextlib
if os.environ.get("CPU_RNG"):            # the reference's random sources: CPU generator for the pixel order and the jitter
    import torch
    from contrastive_lift_amd.data.mos import SceneTables
    from contrastive_lift_amd.trainer import HotPathTrainer
    if os.environ["CPU_RNG"] in ("1", "perm"):
        SceneTables.epoch_order = lambda self, seed, epoch, rank=0, world=1: torch.randperm(self.tables["rays"].shape[0]).to(self.device)[rank::world]
    _mp = HotPathTrainer.main_pass
    def main_pass(self, batch, jitter=None, **k):
        if jitter is None and os.environ["CPU_RNG"] in ("1", "jit"):
            jitter = torch.rand(batch["rays"].shape[0], 1).reshape(-1).to(self.device)
        return _mp(self, batch, jitter=jitter, **k)
    HotPathTrainer.main_pass = main_pass
SEEDS = [int(x) for x in os.environ.get("SEEDS", "0").split(",")]
for name, extra in [(f"{a}|seed={sd}", a.split(";") + [f"seed={sd}"]) for a in sys.argv[1:] for sd in SEEDS] if os.environ.get("SEEDS") else [("default", []), ("fp32", ["mlp_dtype=fp32"]), ("noshards", ["grad_shards=false"]), ("noskip", ["skip_discarded_instance_heads=false"])] if len(sys.argv) < 2 else [(a, a.split(",")) for a in sys.argv[1:]]:
    os.environ["experiment"] = "v_" + name.replace("=", "_").replace(",", "_").replace("|", "_").replace("[", "").replace("]", "")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        train.main(list(base) + [e for e in extra if e])
    alll = [l for l in buf.getvalue().splitlines() if l.startswith("epoch") and " it " in l]
    lines = [l for i, l in enumerate(alll) if i + 1 == len(alll) or alll[i + 1].split()[1] != l.split()[1]]
    vals = [l for l in buf.getvalue().splitlines() if l.startswith("│") and "loss_rgb" not in l]
    if not os.environ.get("SEEDS"): print("   val rows:", [" ".join(x.strip()[:6] for x in v.split("│")[1:4]) for v in vals])
    grids = [l.split("grid=")[1] + " S=" + l.split("S=")[1].split()[0] for l in lines]
    print("   grids:", grids)
    print(name, "| train psnr per epoch:", [l.split("(psnr ")[1].split(")")[0] for l in lines], "| val:", {k: round(v, 3) for k, v in train.main.last_validation.items() if k in ("psnr", "iou", "pq")}, flush=True)
